"""Diagnostic: where does a sample-partitioned render differ from the single-context render?"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracer_amd as pt
from pathtracer_amd import core

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
tris = int(sys.argv[3]) if len(sys.argv) > 3 else 5_800_000
NR, NF = 8, 4
pos, aux, idx = pt.scenes.san_miguel_like(tris)
nodes, ridx, _ = pt.build_bvh(pos, idx, 16)

def ctx_for(in_flight):
    c = pt.PathTracer(0)
    c.upload_scene(pos, aux, ridx, nodes, pt.default_lights(), pt.load_blue_noise())
    c.resize(W, H, NR)
    c.set_option(core.OPT_RESEED, 1); c.set_option(core.OPT_FRAMES_IN_FLIGHT, in_flight)
    c.set_constants(pt.fill_constants(W, H, NR, ridx.shape[0], nodes.shape[0], **pt.scenes.COURTYARD_CAMERA))
    return c

def diff(a, b, name):
    d = (a.view(np.uint32) != b.view(np.uint32)).any(axis=2)
    print(f"{name}: {int(d.sum())} of {d.size} pixels differ; max abs {np.abs(a - b).max():.4g}", flush=True)

c = ctx_for(1)
per_frame = []
for f in range(1, NF + 1):
    c.clear_accumulation(); c.render_samples(f, 1); per_frame.append(c.readback_accumulation())
c.clear_accumulation(); c.render_samples(1, NF); seq = c.readback_accumulation()
diff(seq, ((per_frame[0] + per_frame[1]) + per_frame[2]) + per_frame[3], "K=1 four frames vs sum of single frames")
c.clear_accumulation(); c.render_samples(1, 1); c.render_samples(3, 1); r0 = c.readback_accumulation()
diff(r0, per_frame[0] + per_frame[2], "K=1 frames 1,3 back to back vs sum")
c.close()
c = ctx_for(2)
c.render_samples(1, NF); k2 = c.readback_accumulation()
diff(k2, seq, "K=2 vs K=1")
c.close()
a, b = ctx_for(2), ctx_for(2)
a.render_samples(1, 1); a.render_samples(3, 1); b.render_samples(2, 1); b.render_samples(4, 1)
ra, rb = a.readback_accumulation(), b.readback_accumulation()
diff(ra, per_frame[0] + per_frame[2], "rank 0 (concurrent) vs sum of frames 1,3")
diff(rb, per_frame[1] + per_frame[3], "rank 1 (concurrent) vs sum of frames 2,4")
a.close(); b.close()
