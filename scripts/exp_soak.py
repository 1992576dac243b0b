"""Experiment: a long render through the frame train against the same frames rendered one per pipe, one at a time (GPU box):
accumulation, current image and per-depth ray counts must be equal bit for bit, for every traversal."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt
from pathtracer_amd import core
W, H, NR = 480, 270, 8
NF = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
pos, aux, idx = pt.scenes.courtyard(200_000, seed=5)
nodes, ridx, _ = pt.build_bvh(pos, idx, 16)
cam = dict(pt.scenes.COURTYARD_CAMERA)
for trav in (2, 3, 1):
    out = {}
    for train, parts in ((0, 1), (1, 2), (1, 3)):
        ctx = pt.PathTracer(0)
        ctx.upload_scene(pos, aux, ridx, nodes, pt.default_lights(), pt.load_blue_noise())
        ctx.set_option(core.OPT_TRAVERSAL, trav)
        ctx.resize(W, H, NR)
        ctx.set_constants(pt.fill_constants(W, H, NR, ridx.shape[0], nodes.shape[0], **cam))
        ctx.set_option(core.OPT_FRAME_TRAIN, train); ctx.set_option(core.OPT_TRAIN_PARTS, parts); ctx.set_option(core.OPT_FRAMES_IN_FLIGHT, 1)
        t = time.perf_counter()
        rc = np.zeros(2 * NR, np.uint64)
        done = 0
        for n in (1, 7, 130, NF - 138):               # calls of different lengths
            rc += ctx.render_samples(1 + done, n, want_ray_counts=True)[:2 * NR]; done += n
        ctx.synchronize()
        dt = time.perf_counter() - t
        out[(train, parts)] = (ctx.readback_accumulation(), ctx.readback_current(), rc)
        print(f"traversal {trav} train {train} bands {parts}: {NF} frames in {dt:.2f} s", flush=True)
        ctx.close()
    # the train kept between calls: one frame per call, the accumulation read back (without emptying the train) every 97 frames
    ctx = pt.PathTracer(0)
    ctx.upload_scene(pos, aux, ridx, nodes, pt.default_lights(), pt.load_blue_noise())
    ctx.set_option(core.OPT_TRAVERSAL, trav)
    ctx.resize(W, H, NR)
    consts = pt.fill_constants(W, H, NR, ridx.shape[0], nodes.shape[0], **cam)
    ctx.set_option(core.OPT_FRAME_TRAIN, 1); ctx.set_option(core.OPT_TRAIN_KEEP, 1)
    t = time.perf_counter()
    for f in range(1, NF + 1):
        consts["frameIndex"] = f
        ctx.set_constants(consts)
        ctx.render_sample(f)
        if f % 97 == 0:
            w = ctx.readback_accumulation()[..., 3]
            assert (w == f - NR).all(), (f, float(w.min()), float(w.max()))
    ctx.render_flush(); ctx.synchronize()
    dt = time.perf_counter() - t
    out[("kept", 2)] = (ctx.readback_accumulation(), ctx.readback_current(), out[(0, 1)][2])
    print(f"traversal {trav} train kept, one frame per call: {NF} frames in {dt:.2f} s", flush=True)
    ctx.close()
    ref = out[(0, 1)]
    for key, got in out.items():
        same = all(np.array_equal(np.asarray(ref[k]).view(np.uint8), np.asarray(got[k]).view(np.uint8)) for k in range(3))
        print(f"   {key}: {'identical' if same else 'DIFFERS'}", flush=True)
        assert same
print("soak ok")
