"""Experiment: how much faster the traversal kernel runs when incoherent rays arrive sorted by origin cell (GPU box)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt
from pathtracer_amd import core

def part1by2(x):
    x = x.astype(np.uint64) & 0x3FF
    x = (x | (x << 16)) & 0x30000FF
    x = (x | (x << 8)) & 0x300F00F
    x = (x | (x << 4)) & 0x30C30C3
    x = (x | (x << 2)) & 0x9249249
    return x

pos, aux, idx = pt.scenes.san_miguel_like()
nodes, ridx, _ = pt.build_bvh(pos, idx, 16)
ctx = pt.PathTracer(0)
ctx.upload_scene(pos, aux, ridx, nodes)
ctx.set_option(core.OPT_SORT_RAYS, 0)
ctx.set_option(core.OPT_PERSIST_BLOCKS, 6)
ctx.set_option(core.OPT_REFILL_MIN, 24)
n = 1 << 24
d_rays, d_hits = ctx.alloc(n * 32), ctx.alloc(n * 16)
ctx.generate_rays(d_rays, n)
rays = d_rays.to_host(pt.RAY_DT)
lo, hi = nodes[0]["a"], nodes[0]["b"]
def run(label, r):
    d_rays.from_host(r)
    ms = []
    for _ in range(4):
        st = ctx.trace_rays_device(d_rays, n, d_hits)
        ms.append(ctx.last_kernel_ms())
    print(f"{label}: {min(ms):.2f} ms  {n/min(ms)/1e3:.0f} Mrays/s", flush=True)
run("unsorted", rays)
for bits in (1, 2, 3, 4, 5):
    q = np.clip(((rays["o"] - lo) / (hi - lo) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    key = part1by2(q[:, 0]) | (part1by2(q[:, 1]) << 1) | (part1by2(q[:, 2]) << 2)
    run(f"sorted by origin cell, {bits} bits/axis", rays[np.argsort(key, kind="stable")])
    if bits < 3:
        continue
    octant = ((rays["d"][:, 0] > 0).astype(np.uint64) | ((rays["d"][:, 1] > 0).astype(np.uint64) << 1) | ((rays["d"][:, 2] > 0).astype(np.uint64) << 2))
    run(f"  + direction octant", rays[np.argsort((key << 3) | octant, kind="stable")])
    run(f"  direction octant first, then cell", rays[np.argsort((octant << np.uint64(3 * bits)) | key, kind="stable")])
    # direction quantised finer: 6 faces x 2x2 of the direction cube (24 classes)
    d = rays["d"]; ax = np.argmax(np.abs(d), axis=1); sgn = (np.take_along_axis(d, ax[:, None], 1)[:, 0] > 0)
    u = np.take_along_axis(d, ((ax + 1) % 3)[:, None], 1)[:, 0] > 0; v = np.take_along_axis(d, ((ax + 2) % 3)[:, None], 1)[:, 0] > 0
    dclass = (ax.astype(np.uint64) * 2 + sgn) * 4 + u * 2 + v
    run(f"  + 24 direction classes", rays[np.argsort((key << np.uint64(5)) | dclass.astype(np.uint64), kind="stable")])
