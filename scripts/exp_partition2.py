"""Diagnostic 2: is a frame rendered alone reproducible, and where do back-to-back frames differ?"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracer_amd as pt
from pathtracer_amd import core

W, H = 3840, 2160
NR = int(sys.argv[1]) if len(sys.argv) > 1 else 8
trav = int(sys.argv[2]) if len(sys.argv) > 2 else 2
pos, aux, idx = pt.scenes.san_miguel_like(int(sys.argv[3]) if len(sys.argv) > 3 else 5_800_000)
nodes, ridx, _ = pt.build_bvh(pos, idx, 16)
c = pt.PathTracer(0)
c.upload_scene(pos, aux, ridx, nodes, pt.default_lights(), pt.load_blue_noise())
c.resize(W, H, NR)
c.set_option(core.OPT_RESEED, 1); c.set_option(core.OPT_FRAMES_IN_FLIGHT, 1); c.set_option(core.OPT_TRAVERSAL, trav)
c.set_constants(pt.fill_constants(W, H, NR, ridx.shape[0], nodes.shape[0], **pt.scenes.COURTYARD_CAMERA))

def where(a, b, name):
    d = (a.view(np.uint32) != b.view(np.uint32)).any(axis=2)
    ys, xs = np.nonzero(d)
    print(f"{name}: {int(d.sum())} differ", end="")
    if len(ys):
        print(f"; rows {ys.min()}..{ys.max()}, cols {xs.min()}..{xs.max()}; first pixel index {int(ys[0]) * W + int(xs[0])}; per-row-band counts {np.histogram(ys, bins=8, range=(0, H))[0].tolist()}", end="")
    print(flush=True)

def alone(f):
    c.clear_accumulation(); rc = c.render_samples(f, 1, want_ray_counts=True); return c.readback_accumulation(), rc

a1, rc1 = alone(1); a1b, rc1b = alone(1); a3, rc3 = alone(3); a1c, rc1c = alone(1)
where(a1, a1b, "frame 1 alone, twice"); where(a1, a1c, "frame 1 alone, again after frame 3")
print("ray counts frame 1:", rc1.tolist(), rc1b.tolist())
c.clear_accumulation(); c.render_samples(1, 1); c.render_samples(3, 1); both = c.readback_accumulation()
where(both, a1 + a3, "1 then 3 back to back vs sum")
c.clear_accumulation(); c.render_samples(1, 1); c.synchronize(); c.render_samples(3, 1); both2 = c.readback_accumulation()
where(both2, a1 + a3, "1, sync, 3 vs sum")
cur3 = c.readback_current()
c.clear_accumulation(); c.render_samples(3, 1); cur3b = c.readback_current()
where(cur3.astype(np.float32), cur3b.astype(np.float32), "current image of frame 3: after frame 1 vs alone")
c.close()
