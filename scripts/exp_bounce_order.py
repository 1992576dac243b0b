"""Experiment (GPU box): would tracing a frame's bounce rays in origin-cell order pay?

Takes the closest-hit rays of depth 1..3 of a real 1080p frame (read back from the ray stack between dispatch-site
calls), and traces each set as a batch through pt_trace_rays three ways: in job-list order (what the frame does),
in origin-cell order, and in random order (the microbench's situation).  Kernel times only.
"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt
from pathtracer_amd import core

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="sanmiguel")
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
pos, aux, idx = pt.scenes.san_miguel_like() if args.scene == "sanmiguel" else pt.scenes.sponza_like()
nodes, ridx, _ = pt.build_bvh(pos, idx, 16)
ctx = pt.PathTracer(0)
ctx.upload_scene(pos, aux, ridx, nodes, pt.default_lights(), pt.load_blue_noise())
W, H, NR = 1920, 1080, 8
ctx.resize(W, H, NR)
ctx.set_constants(pt.fill_constants(W, H, NR, ridx.shape[0], nodes.shape[0], frame_index=1, **pt.scenes.COURTYARD_CAMERA))
ctx.launch()


def timed(rays, sort):
    n = rays.shape[0]
    d_rays, d_hits = ctx.alloc(n * 32), ctx.alloc(n * 16)
    d_rays.from_host(rays)
    ctx.set_option(core.OPT_SORT_RAYS, sort)
    ms = []
    for _ in range(args.reps):
        ctx.trace_rays_device(d_rays, n, d_hits)
        ctx.synchronize()
        ms.append(ctx.last_kernel_ms())
    d_rays.free(); d_hits.free()
    return min(ms)


rng = np.random.default_rng(1)
for depth in range(4):
    slots, jobs = ctx.readback_raystack()
    live = jobs[jobs != 0xFFFFFFFF]
    s = slots[live]
    rays = np.zeros(live.shape[0], dtype=pt.RAY_DT)
    rays["o"], rays["d"], rays["min_t"], rays["max_t"] = s["rayOrigin"], s["rayDirection"], 0.001, 1e5
    n = rays.shape[0]
    t_job, t_cell, t_rand = timed(rays, 0), timed(rays, 1), timed(rays[rng.permutation(n)], 0)
    print(f"depth {depth}: {n} rays; job order {t_job:.3f} ms ({n/t_job/1e3:.0f} Mrays/s), origin-cell order {t_cell:.3f} ms "
          f"({n/t_cell/1e3:.0f}), random order {t_rand:.3f} ms ({n/t_rand/1e3:.0f})", flush=True)
    ctx.trace(depth)
