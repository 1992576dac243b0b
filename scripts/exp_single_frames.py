"""Experiment: latency of pt_render_sample, one frame per call, as an interactive host would call it (GPU box)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt
from pathtracer_amd import core
ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="sanmiguel")
ap.add_argument("--frames", type=int, default=40)
ap.add_argument("--train", type=int, default=2)
ap.add_argument("--parts", type=int, default=2)
ap.add_argument("--slots", type=int, default=8)
args = ap.parse_args()
if args.scene == "sanmiguel":
    pos, aux, idx = pt.scenes.san_miguel_like(); cam = dict(pt.scenes.COURTYARD_CAMERA)
else:
    pos, aux, idx = pt.scenes.sponza_like(); cam = dict(pt.scenes.SPONZA_CAMERA)
nodes, ridx, _ = pt.build_bvh(pos, idx, 16)
W, H, NR = 1920, 1080, args.slots
ctx = pt.PathTracer(0)
ctx.upload_scene(pos, aux, ridx, nodes, pt.default_lights(), pt.load_blue_noise())
ctx.resize(W, H, NR)
ctx.set_constants(pt.fill_constants(W, H, NR, ridx.shape[0], nodes.shape[0], **cam))
ctx.set_option(core.OPT_FRAME_TRAIN, args.train); ctx.set_option(core.OPT_TRAIN_PARTS, args.parts)
for f in range(1, 4):
    ctx.render_sample(f); ctx.synchronize()
t = time.perf_counter()
for f in range(4, 4 + args.frames):
    ctx.render_sample(f); ctx.synchronize()           # the host shows every frame before it asks for the next
dt = time.perf_counter() - t
print(f"{args.scene} train {args.train} bands {args.parts} slots {NR}: {dt / args.frames * 1e3:.2f} ms per frame, one frame per call ({args.frames / dt:.1f} spp/s)")
ctx.close()
