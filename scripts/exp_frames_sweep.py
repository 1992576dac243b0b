"""Experiment: frames per second of pt_render_samples over frames in flight, launch width and refill threshold (GPU box).
One scene build, many option settings; the number of hardware queues is the process's (GPU_MAX_HW_QUEUES)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt
from pathtracer_amd import core
ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="sanmiguel")
ap.add_argument("--traversal", type=int, default=2)
ap.add_argument("--frames", type=int, default=36)
ap.add_argument("--inflight", default="1,2,3,4,6,8")
ap.add_argument("--blocks", default="0")
ap.add_argument("--refill", default="0")
ap.add_argument("--merge", default="1")
ap.add_argument("--layout", type=int, default=0)
ap.add_argument("--train", default="1")
ap.add_argument("--parts", default="2")
args = ap.parse_args()
if args.scene == "sanmiguel":
    pos, aux, idx = pt.scenes.san_miguel_like(); cam = dict(pt.scenes.COURTYARD_CAMERA)
else:
    pos, aux, idx = pt.scenes.sponza_like(); cam = dict(pt.scenes.SPONZA_CAMERA)
nodes, ridx, _ = pt.build_bvh(pos, idx, 16)
W, H, NR = 1920, 1080, 8
print("hardware queues:", os.environ.get("GPU_MAX_HW_QUEUES"), flush=True)
for parts in [int(v) for v in args.parts.split(",")]:
 for train in [int(v) for v in args.train.split(",")]:
  for merge in [int(v) for v in args.merge.split(",")]:
    for blocks in [int(v) for v in args.blocks.split(",")]:
        for refill in [int(v) for v in args.refill.split(",")]:
            for inflight in [int(v) for v in args.inflight.split(",")]:
                ctx = pt.PathTracer(0)
                ctx.set_option(core.OPT_RECORD_LAYOUT, args.layout)
                ctx.upload_scene(pos, aux, ridx, nodes, pt.default_lights(), pt.load_blue_noise())
                ctx.set_option(core.OPT_TRAVERSAL, args.traversal)
                ctx.resize(W, H, NR)
                ctx.set_constants(pt.fill_constants(W, H, NR, ridx.shape[0], nodes.shape[0], **cam))
                ctx.set_option(core.OPT_FRAMES_IN_FLIGHT, inflight)
                ctx.set_option(core.OPT_MERGE_PASSES, merge)
                ctx.set_option(core.OPT_FRAME_TRAIN, train)
                ctx.set_option(core.OPT_TRAIN_PARTS, parts)
                if blocks: ctx.set_option(core.OPT_PERSIST_BLOCKS, blocks)
                if refill: ctx.set_option(core.OPT_REFILL_MIN, refill)
                ctx.render_samples(1, 2 * inflight); ctx.synchronize()
                t = time.perf_counter()
                ctx.render_samples(100, args.frames); ctx.synchronize()
                dt = time.perf_counter() - t
                print(f"parts {parts} train {train} layout {args.layout} {args.scene} traversal {args.traversal} merge {merge} blocks {blocks} refill {refill} in flight {inflight}: {args.frames / dt:.1f} spp/s", flush=True)
                ctx.close()
