"""Experiment: how the near-child-first walk differs from the reference-order walk (GPU box)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt
from pathtracer_amd import core
tris = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
pos, aux, idx = pt.scenes.courtyard(tris, seed=13)
nodes, ridx, depth = pt.build_bvh(pos, idx, 16)
ctx = pt.PathTracer(0)
ctx.upload_scene(pos, aux, ridx, nodes)
n = 1 << 22
d_rays, d_hits = ctx.alloc(n * 32), ctx.alloc(n * 16)
ctx.generate_rays(d_rays, n)
res = {}
for mode in (1, 2):
    ctx.set_option(core.OPT_TRAVERSAL, mode)
    ctx.trace_rays_device(d_rays, n, d_hits)
    res[mode] = d_hits.to_host(pt.HIT_DT)
a, b = res[1], res[2]
diff = (a["i1"] != b["i1"]) | (a["i2"] != b["i2"]) | (a["i3"] != b["i3"])
print(f"tris {ridx.shape[0]}: mismatch {diff.mean():.3e}")
ha, hb = a["i1"] >= 0, b["i1"] >= 0
print("  ref hit / ordered miss:", (diff & ha & ~hb).sum(), " ref miss / ordered hit:", (diff & ~ha & hb).sum(), " both hit:", (diff & ha & hb).sum())
both = diff & ha & hb
dt = b["t"][both] - a["t"][both]
print("  ordered closer:", (dt < 0).sum(), " ordered farther:", (dt > 0).sum(), " equal t:", (dt == 0).sum())
ext = float((nodes["b"][0] - nodes["a"][0]).max())
for q in (50, 90, 99, 99.9, 100):
    print(f"  |dt| percentile {q}: {np.percentile(np.abs(dt), q):.3e}  (scene extent {ext:.2f})")
rel = np.abs(dt) / np.maximum(a["t"][both], 1e-9)
for q in (50, 90, 99, 100):
    print(f"  |dt|/t percentile {q}: {np.percentile(rel, q):.3e}")
rays_h = d_rays.to_host(pt.RAY_DT)
sel = np.nonzero(both)[0][:2000]
np.savez_compressed(os.path.join(ROOT, 'gpurun_out', f'mismatch_{tris}.npz'), rays=rays_h[sel].view(np.uint8), ref=a[sel].view(np.uint8), ordered=b[sel].view(np.uint8))
ctx.close()
