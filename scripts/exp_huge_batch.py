"""Experiment: one pt_trace_rays call over 2^31 rays (64 GiB of rays, 32 GiB of hits) - index arithmetic at the top of the
32-bit range (GPU box).  Hits of sub-ranges at the start, in the middle and at the very end must equal those of the same rays
traced as small batches of their own (ray i depends on seed and i only)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt
from pathtracer_amd import core
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 31
n = 1 << logn if logn < 32 else 0xFE000000          # 32: the most pt_trace_rays takes in one call
modes = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2, 3]
pos, aux, idx = pt.scenes.courtyard(1_000_000, seed=3)
nodes, ridx, _ = pt.build_bvh(pos, idx, 16)
ctx = pt.PathTracer(0)
ctx.upload_scene(pos, aux, ridx, nodes)
for trav in modes:
    ctx.set_option(core.OPT_TRAVERSAL, trav)
    d_rays, d_hits = ctx.alloc(n * 32), ctx.alloc(n * 16)
    ctx.generate_rays(d_rays, n, seed=77)
    ctx.synchronize()
    t = time.perf_counter()
    ctx.trace_rays_device(d_rays, n, d_hits)
    ctx.synchronize()
    dt = time.perf_counter() - t
    print(f"traversal {trav}: {n} rays in {dt*1e3:.0f} ms = {n/dt/1e6:.0f} Mrays/s", flush=True)
    m = 1 << 18
    small_r, small_h = ctx.alloc(m * 32), ctx.alloc(m * 16)
    for first in (0, n // 2 - m // 2, n - m):
        big = np.zeros(m, dtype=pt.HIT_DT)
        ctx._chk(ctx._L.pt_copy_to_host(ctx._h, big.ctypes.data, d_hits.ptr + first * 16, m * 16))
        ctx.generate_rays(small_r, m, seed=77, first=first)
        ctx.trace_rays_device(small_r, m, small_h)
        small = small_h.to_host(pt.HIT_DT, m)
        same = np.array_equal(big.view(np.uint8), small.view(np.uint8))
        print(f"   rays [{first}, {first + m}): {'identical' if same else 'DIFFER'}; hit rate {(big['i1'] >= 0).mean():.3f}", flush=True)
        assert same
    small_r.free(); small_h.free(); d_rays.free(); d_hits.free()
ctx.close()
print("huge batch ok")
