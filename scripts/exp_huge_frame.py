"""Experiment: the frame pipelines at 7680 x 4320 x 8 slots - 265 M slots per ray stack, ring-wide slot indices up to 2.4e9 (GPU
box).  The frame train's accumulation must equal the per-frame pipeline's bit for bit."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt
from pathtracer_amd import core
W, H, NR, NF = 7680, 4320, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 10
pos, aux, idx = pt.scenes.sponza_like()
nodes, ridx, _ = pt.build_bvh(pos, idx, 16)
out = {}
for train in (1, 0):
    ctx = pt.PathTracer(0)
    ctx.upload_scene(pos, aux, ridx, nodes, pt.default_lights(), pt.load_blue_noise())
    ctx.resize(W, H, NR)
    ctx.set_constants(pt.fill_constants(W, H, NR, ridx.shape[0], nodes.shape[0], **pt.scenes.SPONZA_CAMERA))
    ctx.set_option(core.OPT_FRAME_TRAIN, train); ctx.set_option(core.OPT_FRAMES_IN_FLIGHT, 1)
    t = time.perf_counter()
    rc = ctx.render_samples(1, NF, want_ray_counts=True)
    ctx.synchronize()
    dt = time.perf_counter() - t
    acc = ctx.readback_accumulation()
    print(f"train {train}: {NF} frames of {W} x {H} x {NR} in {dt:.2f} s ({NF/dt:.2f} spp/s, {rc[:2*NR].sum()/dt/1e6:.0f} Mrays/s); mean radiance {acc[..., :3].mean() / NF:.4f}", flush=True)
    out[train] = (acc, rc)
    ctx.close()
assert (out[1][0][..., 3] == NF).all()
assert np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32)) and np.array_equal(out[0][1], out[1][1])
print("huge frame ok: identical")
