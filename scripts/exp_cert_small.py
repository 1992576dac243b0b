"""Experiment (GPU box): the certified walk on one scene (bunny, sponza, sanmiguel) - how many rays take the second walk, for
rays spread through the scene box and for rays from one eye point, and what the passes cost.  usage: exp_cert_small.py [scene]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt
from pathtracer_amd import core
scene = sys.argv[1] if len(sys.argv) > 1 else "bunny"
if scene == "bunny":
    pos, aux, idx = pt.load_ply(os.path.join(ROOT, "tests", "golden", "models", "CBbunny.ply")); eye = (0.0, 0.75, 3.0)
elif scene == "sponza":
    pos, aux, idx = pt.scenes.sponza_like(); eye = None
else:
    pos, aux, idx = pt.scenes.san_miguel_like(); eye = None
nodes, ridx, _ = pt.build_bvh(pos, idx, 16)
ctx = pt.PathTracer(0)
ctx.upload_scene(pos, aux, ridx, nodes, pt.default_lights(), pt.load_blue_noise())
n = 1 << 21
rng = np.random.default_rng(5)
lo, hi = nodes[0]["a"], nodes[0]["b"]
def rays_from(o, target):
    r = np.zeros(n, dtype=pt.RAY_DT)
    r["o"] = o
    d = target - r["o"]
    r["d"] = d / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-20)
    r["min_t"], r["max_t"] = 0.001, 1e5
    return r
box = lambda: (lo + (hi - lo) * rng.random((n, 3), dtype=np.float32)).astype(np.float32)
if eye is None: eye = tuple(0.5 * (lo + hi) + np.array([0.0, 0.1, 0.0], dtype=np.float32) * (hi - lo))
sets = {"eye": rays_from(np.array(eye, dtype=np.float32), box()), "spread": rays_from(box(), box())}
for name, rays in sets.items():
    d_rays, d_hits = ctx.alloc(n * 32), ctx.alloc(n * 16)
    d_rays.from_host(rays)
    for trav in (2, 3, 1):
        ctx.set_option(core.OPT_TRAVERSAL, trav)
        st = ctx.trace_rays_device(d_rays, n, d_hits, want_stats=True)
        ctx.trace_rays_device(d_rays, n, d_hits); ctx.synchronize()
        ms = []
        for _ in range(5):
            ctx.trace_rays_device(d_rays, n, d_hits); ms.append(ctx.last_kernel_ms())
        print(f"{name}: traversal {trav}: {min(ms):.3f} ms kernel(s) for 2^21 rays, records/ray {st[2] / n:.1f}, rays walked {int(st[3])} ({int(st[3]) - n} second walks)", flush=True)
