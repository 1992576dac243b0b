"""Experiment: frames per second of pt_render_samples over the options of the frame pipeline (GPU box).  One scene build,
many option settings; the number of hardware queues is the process's (GPU_MAX_HW_QUEUES)."""
import argparse, itertools, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt
from pathtracer_amd import core
ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="sanmiguel")
ap.add_argument("--traversal", default="2")
ap.add_argument("--frames", type=int, default=120)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--layout", type=int, default=0)
ap.add_argument("--slots", default="8")
ap.add_argument("--bits", type=int, default=4)
ap.add_argument("--second", type=int, default=-1, help="PT_OPT_SECOND_WALK (traversal 3): 0 = one lane per listed ray, 1 = one wavefront per listed ray; -1 = leave")
ap.add_argument("--keep", type=int, default=0, help="PT_OPT_TRAIN_KEEP: the train is kept between calls (flushed inside the timed region at the end)")
ap.add_argument("--percall", type=int, default=0, help="frames per pt_render_samples call (a synchronize after each); 0 = all in one call")
lists = ("train", "parts", "merge", "blocks", "refill", "inflight")
defaults = dict(train="1", parts="2", merge="1", blocks="0", refill="0", inflight="3")
for name in lists:
    ap.add_argument("--" + name, default=defaults[name])
args = ap.parse_args()
if args.scene == "bunny":
    import numpy as np
    pos, aux, idx = pt.load_ply(os.path.join(ROOT, "tests", "golden", "models", "CBbunny.ply"))
    cam = dict(orbit_angle=-np.pi / 2, orbit_height=0.75, focus_center=(0.0, 0.75, 0.0))
elif args.scene == "sanmiguel":
    pos, aux, idx = pt.scenes.san_miguel_like(); cam = dict(pt.scenes.COURTYARD_CAMERA)
else:
    pos, aux, idx = pt.scenes.sponza_like(); cam = dict(pt.scenes.SPONZA_CAMERA)
nodes, ridx, _ = pt.build_bvh(pos, idx, 16)
W, H = args.width, args.height
print("hardware queues:", os.environ.get("GPU_MAX_HW_QUEUES"), flush=True)
grid = [[int(v) for v in getattr(args, name).split(",")] for name in lists]
for trav, NR in itertools.product([int(v) for v in args.traversal.split(",")], [int(v) for v in args.slots.split(",")]):
    for train, parts, merge, blocks, refill, inflight in itertools.product(*grid):
        ctx = pt.PathTracer(0)
        ctx.set_option(core.OPT_RECORD_LAYOUT, args.layout)
        ctx.upload_scene(pos, aux, ridx, nodes, pt.default_lights(), pt.load_blue_noise())
        ctx.set_option(core.OPT_TRAVERSAL, trav)
        ctx.resize(W, H, NR)
        ctx.set_constants(pt.fill_constants(W, H, NR, ridx.shape[0], nodes.shape[0], **cam))
        for o, v in ((core.OPT_FRAMES_IN_FLIGHT, inflight), (core.OPT_MERGE_PASSES, merge), (core.OPT_FRAME_TRAIN, train), (core.OPT_TRAIN_PARTS, parts)):
            ctx.set_option(o, v)
        ctx.set_option(core.OPT_TRAIN_KEEP, args.keep)
        if args.second >= 0: ctx.set_option(core.OPT_SECOND_WALK, args.second)
        if blocks: ctx.set_option(core.OPT_PERSIST_BLOCKS, blocks)
        if refill: ctx.set_option(core.OPT_REFILL_MIN, refill)
        ctx.render_samples(1, 12); ctx.synchronize()
        t = time.perf_counter()
        if args.percall:
            for first in range(100, 100 + args.frames, args.percall):
                ctx.render_samples(first, min(args.percall, 100 + args.frames - first))
                if not args.keep: ctx.synchronize()
            ctx.render_flush(); ctx.synchronize()
        else:
            ctx.render_samples(100, args.frames); ctx.synchronize()
        dt = time.perf_counter() - t
        print(f"{args.scene} {W}x{H} slots {NR} traversal {trav} train {train} parts {parts} merge {merge} blocks {blocks} refill {refill} in flight {inflight}: "
              f"{args.frames / dt:.1f} spp/s ({args.frames} frames in {args.frames + NR if train else args.frames} steps{f', {args.percall} per call' if args.percall else ''}{', train kept' if args.keep else ''})", flush=True)
        ctx.close()
