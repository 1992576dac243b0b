"""Experiment: the certified walk's second pass, a ray per lane against a ray per wavefront: batches and frames."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracer_amd as pt
from pathtracer_amd import core

tris = int(sys.argv[1]) if len(sys.argv) > 1 else 5_800_000
pos, aux, idx = pt.scenes.san_miguel_like(tris)
nodes, ridx, depth = pt.build_bvh(pos, idx, 16)
ctx = pt.PathTracer(0)
ctx.upload_scene(pos, aux, ridx, nodes, pt.default_lights(), pt.load_blue_noise())
n = 1 << 26
d_rays, d_hits = ctx.alloc(n * 32), ctx.alloc(n * 16)
ctx.generate_rays(d_rays, n)
ctx.set_option(core.OPT_TRAVERSAL, 1)
m = 1 << 22
ctx.trace_rays_device(d_rays, m, d_hits)
ref = d_hits.to_host(pt.HIT_DT, m)
ctx.set_option(core.OPT_TRAVERSAL, 3)
for second in (0, 1):
    ctx.set_option(core.OPT_SECOND_WALK, second)
    st = ctx.trace_rays_device(d_rays, m, d_hits, want_stats=True)
    got = d_hits.to_host(pt.HIT_DT, m)
    print(f"second walk {second}: {int((got.view(np.uint32).reshape(-1, 4) != ref.view(np.uint32).reshape(-1, 4)).any(axis=1).sum())} of {m} hits differ from the reference-order walk; "
          f"{int(st[3]) - m} rays took the second walk", flush=True)
    for _ in range(2):
        ctx.trace_rays_device(d_rays, n, d_hits)
    ctx.synchronize()
    t = time.perf_counter()
    for _ in range(4):
        ctx.trace_rays_device(d_rays, n, d_hits)
        ms = ctx.last_kernel_ms()
    ctx.synchronize()
    print(f"   batch of 2^26: {(time.perf_counter() - t) / 4 * 1e3:.2f} ms per call, both passes {ms:.2f} ms", flush=True)
d_rays.free(); d_hits.free()
W, H, NR = 1920, 1080, 8
ctx.resize(W, H, NR)
ctx.set_constants(pt.fill_constants(W, H, NR, ridx.shape[0], nodes.shape[0], **pt.scenes.COURTYARD_CAMERA))
acc = {}
for trav, second in ((2, 1), (3, 0), (3, 1), (1, 1)):
    ctx.set_option(core.OPT_TRAVERSAL, trav); ctx.set_option(core.OPT_SECOND_WALK, second)
    ctx.clear_accumulation()
    ctx.render_samples(1, 6)
    acc[(trav, second)] = ctx.readback_accumulation()
    ctx.synchronize()
    t = time.perf_counter()
    ctx.render_samples(7, 24)
    ctx.synchronize()
    dt = time.perf_counter() - t
    print(f"frames 1080p x 8, traversal {trav}, second walk {second}: {24 / dt:.1f} spp/s", flush=True)
d = lambda a, b: int((a.view(np.uint32) != b.view(np.uint32)).any(axis=2).sum())
print(f"pixels differing from the reference-order frames after 6 frames: certified/lane {d(acc[(3, 0)], acc[(1, 1)])}, certified/wave {d(acc[(3, 1)], acc[(1, 1)])}, near-first {d(acc[(2, 1)], acc[(1, 1)])}")
ctx.close()
