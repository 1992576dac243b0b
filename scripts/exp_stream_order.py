"""Experiment: does the frame train's rate depend on how many other streams the context created and used before it (GPU box)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt
from pathtracer_amd import core
pos, aux, idx = pt.scenes.san_miguel_like(); cam = dict(pt.scenes.COURTYARD_CAMERA)
nodes, ridx, _ = pt.build_bvh(pos, idx, 16)
W, H, NR = 1920, 1080, 8
print("hardware queues:", os.environ.get("GPU_MAX_HW_QUEUES"), flush=True)
for pipes_first in (0, 1, 2, 3, 4, 5, 8):
    ctx = pt.PathTracer(0)
    ctx.upload_scene(pos, aux, ridx, nodes, pt.default_lights(), pt.load_blue_noise())
    ctx.resize(W, H, NR)
    ctx.set_constants(pt.fill_constants(W, H, NR, ridx.shape[0], nodes.shape[0], **cam))
    if pipes_first:
        ctx.set_option(core.OPT_FRAME_TRAIN, 0); ctx.set_option(core.OPT_MERGE_PASSES, 0); ctx.set_option(core.OPT_FRAMES_IN_FLIGHT, pipes_first)
        ctx.render_samples(1, 2 * pipes_first); ctx.synchronize()
        ctx.set_option(core.OPT_MERGE_PASSES, 1)
    ctx.set_option(core.OPT_FRAME_TRAIN, 1)
    ctx.render_samples(50, 12); ctx.synchronize()
    t = time.perf_counter()
    ctx.render_samples(100, 120); ctx.synchronize()
    dt = time.perf_counter() - t
    print(f"{pipes_first} pipes ({2 * pipes_first} streams) used before the train: {120 / dt:.1f} spp/s", flush=True)
    ctx.close()
