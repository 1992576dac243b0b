"""Experiment: traversal throughput and visit statistics per layout on a stand-in scene (GPU box)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracer_amd as pt
from pathtracer_amd import core

ap = argparse.ArgumentParser()
ap.add_argument("--tris", type=int, default=1_000_000)
ap.add_argument("--rays", type=int, default=1 << 24)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--blocks", type=str, default="0", help="0 = library default")
ap.add_argument("--refill", type=str, default="0", help="0 = library default")
ap.add_argument("--modes", type=str, default="0,1,2")
ap.add_argument("--chunk", type=str, default="0")
ap.add_argument("--sort", type=int, default=2)
ap.add_argument("--bits", type=str, default="3")
ap.add_argument("--layout", type=int, default=0, help="0 = 64 B records, 1 = 48 B records when encodable")
ap.add_argument("--scene", type=str, default="courtyard")
args = ap.parse_args()

t = time.time()
if args.scene == "bunny":
    pos, aux, idx = pt.load_ply(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "models", "CBbunny.ply"))
elif args.scene == "sanmiguel":
    pos, aux, idx = pt.scenes.san_miguel_like(args.tris)
else:
    pos, aux, idx = pt.scenes.courtyard(args.tris, seed=13)
print(f"scene: {idx.shape[0]} tris, {pos.shape[0]} verts, gen {time.time()-t:.1f}s", flush=True)
t = time.time()
nodes, ridx, depth = pt.build_bvh(pos, idx, 16)
print(f"bvh: {nodes.shape[0]} nodes depth {depth} in {time.time()-t:.1f}s", flush=True)
ctx = pt.PathTracer(0)
ctx.set_option(core.OPT_RECORD_LAYOUT, args.layout)
t = time.time()
ctx.upload_scene(pos, aux, ridx, nodes)
print(f"upload+flatten {time.time()-t:.1f}s", flush=True)
n = args.rays
d_rays, d_hits = ctx.alloc(n * 32), ctx.alloc(n * 16)
ctx.generate_rays(d_rays, n)
ctx.synchronize()
ref_hits = None
for mode in [int(m) for m in args.modes.split(",")]:
    ctx.set_option(core.OPT_TRAVERSAL, mode)
    st = ctx.trace_rays_device(d_rays, n, d_hits, want_stats=True)
    hits = d_hits.to_host(pt.HIT_DT)
    if ref_hits is None:
        ref_hits = hits
    diff = ((hits["i1"] != ref_hits["i1"]) | (hits["i2"] != ref_hits["i2"]) | (hits["i3"] != ref_hits["i3"])).mean()
    tdiff = int((hits["t"].view(np.uint32) != ref_hits["t"].view(np.uint32)).sum())
    print(f"mode {mode}: box/ray {st[0]/n:.1f} tri/ray {st[1]/n:.2f} rec/ray {st[2]/n:.1f} hit rate {(hits['i1']>=0).mean():.3f} diff vs first mode {diff:.2e} (t differs bitwise on {tdiff} rays; rays walked / rays {st[3]/n:.5f})", flush=True)
    for blocks in [int(b) for b in args.blocks.split(",")]:
        for refill in [int(r) for r in args.refill.split(",")]:
          for chunk in [int(c) for c in args.chunk.split(",")]:
           for bits in [int(c) for c in args.bits.split(",")]:
            ctx.set_option(core.OPT_SORT_BITS, bits)
            ctx.set_option(core.OPT_CLAIM_CHUNK, chunk)
            ctx.set_option(core.OPT_SORT_RAYS, args.sort)
            if blocks:
                ctx.set_option(core.OPT_PERSIST_BLOCKS, blocks)
            if refill:
                ctx.set_option(core.OPT_REFILL_MIN, refill)
            ms = []
            ctx.synchronize()
            tw = time.perf_counter()
            for _ in range(args.reps):
                ctx.trace_rays_device(d_rays, n, d_hits)
                ms.append(ctx.last_kernel_ms())
            ctx.synchronize()
            wall = (time.perf_counter() - tw) / args.reps * 1e3
            best = min(ms)
            print(f"   blocks/CU {blocks} refill {refill} chunk {chunk} bits {bits} wall/launch {wall:.2f} ms: {best:.2f} ms  {n/best/1e3:.1f} Mrays/s  (all: {', '.join('%.2f'%m for m in ms)})", flush=True)
d_rays.free(); d_hits.free()
ctx.close()
