"""Experiment (diagnostic build with in-kernel cycle stamps): where a wavefront's cycles go."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt
from pathtracer_amd import core
tris = int(sys.argv[1]) if len(sys.argv) > 1 else 5_800_000
pos, aux, idx = pt.scenes.courtyard(tris, seed=13)
nodes, ridx, depth = pt.build_bvh(pos, idx, 16)
ctx = pt.PathTracer(0)
ctx.upload_scene(pos, aux, ridx, nodes)
n = 1 << 25
d_rays, d_hits = ctx.alloc(n * 32), ctx.alloc(n * 16)
ctx.generate_rays(d_rays, n)
for blocks in (2, 4, 6, 8):
    ctx.set_option(core.OPT_PERSIST_BLOCKS, blocks)
    ctx.set_option(core.OPT_TRAVERSAL, 2)
    # the stats pointer is only zeroed/copied when stats are requested; the timing build writes cycle sums there
    st = ctx.trace_rays_device(d_rays, n, d_hits, want_stats=True)   # counting kernel variant (also timed build)
    ms = ctx.last_kernel_ms()
    waves = 256 * blocks * 4
    tFetch, tTotal, nIter, nLane = [float(x) for x in st]
    print(f"blocks/CU {blocks}: {ms:.2f} ms, {n/ms/1e3:.0f} Mrays/s | per wave: {tTotal/waves/1e6:.2f} Mticks, iterations {nIter/waves:.0f}, "
          f"fetch wait {tFetch/max(nIter,1):.1f} ticks/iter, total {tTotal/max(nIter,1):.1f} ticks/iter, lanes active/iter {nLane/max(nIter,1):.1f}")
ctx.close()
