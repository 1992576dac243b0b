"""Experiment: what the closest-hit rays of one frame cost when each bounce's rays are traced as a batch with the whole GPU
to themselves (GPU box).  Renders one frame, reads the ray stack back, rebuilds the rays of every depth from their slots and
traces them with pt_trace_rays in pixel order, with counters."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt
from pathtracer_amd import core
scene = sys.argv[1] if len(sys.argv) > 1 else "sanmiguel"
if scene == "sanmiguel":
    pos, aux, idx = pt.scenes.san_miguel_like(); cam = dict(pt.scenes.COURTYARD_CAMERA)
else:
    pos, aux, idx = pt.scenes.sponza_like(); cam = dict(pt.scenes.SPONZA_CAMERA)
nodes, ridx, _ = pt.build_bvh(pos, idx, 16)
W, H, NR = 1920, 1080, 8
ctx = pt.PathTracer(0)
ctx.upload_scene(pos, aux, ridx, nodes, pt.default_lights(), pt.load_blue_noise())
ctx.resize(W, H, NR)
ctx.set_constants(pt.fill_constants(W, H, NR, ridx.shape[0], nodes.shape[0], **cam))
ctx.set_option(core.OPT_FRAMES_IN_FLIGHT, 1)
rc = ctx.render_sample(1, want_ray_counts=True)
slots, _ = ctx.readback_raystack()
slots = slots.reshape(W * H, NR)
ctx.set_option(core.OPT_SORT_RAYS, 0)
tot_ms = 0.0
for d in range(NR):
    s = slots[:, d]
    live = (np.abs(s["rayDirection"]).sum(axis=1) > 0)
    n = int(live.sum())
    rays = np.zeros(n, dtype=pt.RAY_DT)
    rays["o"] = s["rayOrigin"][live]; rays["d"] = s["rayDirection"][live]; rays["min_t"] = 0.001; rays["max_t"] = 100000.0
    d_rays, d_hits = ctx.alloc(max(n, 1) * 32), ctx.alloc(max(n, 1) * 16)
    d_rays.from_host(rays)
    st = ctx.trace_rays_device(d_rays, n, d_hits, want_stats=True)
    best = 1e9
    for rep in range(5):
        ctx.trace_rays_device(d_rays, n, d_hits); best = min(best, ctx.last_kernel_ms())
    hits = d_hits.to_host(pt.HIT_DT, n)
    tot_ms += best
    print(f"depth {d}: {n} rays (frame counted {int(rc[2*d])}), hit rate {(hits['i1'] >= 0).mean():.3f}, records/ray {st[2]/max(n,1):.1f}, alone {best:.3f} ms = {n/best/1e3:.0f} Mrays/s", flush=True)
    d_rays.free(); d_hits.free()
print(f"closest-hit rays of the frame as batches, one after the other: {tot_ms:.2f} ms")
ctx.close()
