"""Experiment: the 128 B treelet layout against the 64 B records on the bench scene: hits bit for bit, time per launch."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracer_amd as pt
from pathtracer_amd import core

ap = argparse.ArgumentParser()
ap.add_argument("--tris", type=int, default=5_800_000)
ap.add_argument("--rays", type=int, default=1 << 26)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--modes", type=str, default="2,1")
ap.add_argument("--blocks", type=str, default="0")
ap.add_argument("--refill", type=str, default="0")
ap.add_argument("--layouts", type=str, default="0,2")
ap.add_argument("--sort", type=int, default=2)
ap.add_argument("--anyhit", type=int, default=1)
args = ap.parse_args()

pos, aux, idx = pt.scenes.san_miguel_like(args.tris)
nodes, ridx, depth = pt.build_bvh(pos, idx, 16)
print(f"scene: {ridx.shape[0]} triangles, depth {depth}", flush=True)
n = args.rays
ref = {}
for layout in [int(x) for x in args.layouts.split(",")]:
    ctx = pt.PathTracer(0)
    ctx.set_option(core.OPT_RECORD_LAYOUT, layout)
    t = time.time()
    ctx.upload_scene(pos, aux, ridx, nodes)
    print(f"layout {layout}: flatten+upload {time.time()-t:.1f}s", flush=True)
    d_rays, d_hits = ctx.alloc(n * 32), ctx.alloc(n * 16)
    ctx.generate_rays(d_rays, n)
    ctx.set_option(core.OPT_SORT_RAYS, args.sort)
    for mode in [int(m) for m in args.modes.split(",")]:
        ctx.set_option(core.OPT_TRAVERSAL, mode)
        for any_hit in ([False, True] if args.anyhit else [False]):
            nn = n if not any_hit else min(n, 1 << 24)
            st = ctx.trace_rays_device(d_rays, min(nn, 1 << 22), d_hits, any_hit=any_hit, want_stats=True)
            m = min(nn, 1 << 22)
            ctx.trace_rays_device(d_rays, nn, d_hits, any_hit=any_hit)
            hits = d_hits.to_host(pt.HIT_DT, nn)
            key = (mode, any_hit)
            if key not in ref:
                ref[key] = hits
                cmp = "reference for this mode"
            else:
                r = ref[key]
                if any_hit and mode == 2:
                    cmp = f"hit/miss verdicts differ on {int(((hits['i1'] >= 0) != (r['i1'] >= 0)).sum())} rays"
                else:
                    cmp = f"{int((hits.view(np.uint32).reshape(-1, 4) != r.view(np.uint32).reshape(-1, 4)).any(axis=1).sum())} of {nn} hits differ bitwise from the first layout"
            print(f"layout {layout} mode {mode} any_hit {int(any_hit)}: box/ray {st[0]/m:.1f} tri/ray {st[1]/m:.2f} units/ray {st[2]/m:.1f}; {cmp}", flush=True)
            if any_hit:
                continue
            for blocks in [int(b) for b in args.blocks.split(",")]:
                for refill in [int(r) for r in args.refill.split(",")]:
                    if blocks:
                        ctx.set_option(core.OPT_PERSIST_BLOCKS, blocks)
                    if refill:
                        ctx.set_option(core.OPT_REFILL_MIN, refill)
                    ms = []
                    for _ in range(args.reps):
                        ctx.trace_rays_device(d_rays, n, d_hits)
                        ms.append(ctx.last_kernel_ms())
                    print(f"   blocks/CU {blocks} refill {refill}: {min(ms):.2f} ms  {n/min(ms)/1e3:.1f} Mrays/s  (all: {', '.join('%.2f' % x for x in ms)})", flush=True)
    d_rays.free(); d_hits.free()
    ctx.close()
