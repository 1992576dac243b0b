"""Experiment: wavefront path tracer throughput at 1080p (GPU box)."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt
from pathtracer_amd import core
ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="sponza")
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--slots", type=int, default=8)
ap.add_argument("--frames", type=int, default=20)
ap.add_argument("--traversal", type=int, default=2)
ap.add_argument("--inflight", type=int, default=3)
ap.add_argument("--reseed", type=int, default=0)
ap.add_argument("--blocks", type=int, default=0, help="0 = library default")
ap.add_argument("--refill", type=int, default=0, help="0 = library default")
ap.add_argument("--shadow-stream", type=int, default=2)
ap.add_argument("--merge", type=int, default=1)
ap.add_argument("--train", type=int, default=1)
ap.add_argument("--parts", type=int, default=2)
args = ap.parse_args()
if args.scene == "bunny":
    pos, aux, idx = pt.load_ply(os.path.join(ROOT, "tests", "golden", "models", "CBbunny.ply"))
    cam = dict(orbit_angle=-np.pi / 2, orbit_height=0.75, focus_center=(0.0, 0.75, 0.0))
elif args.scene == "sanmiguel":
    pos, aux, idx = pt.scenes.san_miguel_like()
    cam = dict(pt.scenes.COURTYARD_CAMERA)
else:
    pos, aux, idx = pt.scenes.sponza_like()
    cam = dict(pt.scenes.SPONZA_CAMERA)
nodes, ridx, _ = pt.build_bvh(pos, idx, 16)
ctx = pt.PathTracer(0)
ctx.upload_scene(pos, aux, ridx, nodes, pt.default_lights(), pt.load_blue_noise())
ctx.set_option(core.OPT_TRAVERSAL, args.traversal)
W, H, NR = args.width, args.height, args.slots
ctx.resize(W, H, NR)
ctx.set_constants(pt.fill_constants(W, H, NR, ridx.shape[0], nodes.shape[0], **cam))
ctx.set_option(core.OPT_FRAMES_IN_FLIGHT, args.inflight)
ctx.set_option(core.OPT_RESEED, args.reseed)
ctx.set_option(core.OPT_SHADOW_STREAM, args.shadow_stream)
ctx.set_option(core.OPT_MERGE_PASSES, args.merge)
ctx.set_option(core.OPT_FRAME_TRAIN, args.train)
ctx.set_option(core.OPT_TRAIN_PARTS, args.parts)
if args.blocks:
    ctx.set_option(core.OPT_PERSIST_BLOCKS, args.blocks)
if args.refill:
    ctx.set_option(core.OPT_REFILL_MIN, args.refill)
ctx.render_samples(1, 9)
ctx.synchronize()
t = time.perf_counter()
ctx.render_samples(20, args.frames)
tq = time.perf_counter() - t
ctx.synchronize()
dt = time.perf_counter() - t
print(f"host returned after {tq*1e3:.1f} ms of {dt*1e3:.1f} ms")
rc = ctx.render_sample(20 + args.frames, want_ray_counts=True)
rays = int(rc[:2 * NR].sum())
print(f"{args.scene} {W}x{H} slots {NR} traversal {args.traversal} in flight {args.inflight} blocks {args.blocks} refill {args.refill} reseed {args.reseed} shadow stream {args.shadow_stream} merge {args.merge}: {args.frames/dt:.2f} spp/s, {dt/args.frames*1e3:.2f} ms/frame, rays/sample {rays}, {rays*args.frames/dt/1e6:.0f} Mrays/s")
print("ray counts (closest, shadow) per depth:", rc[:2 * NR].reshape(-1, 2).tolist())
acc = ctx.readback_accumulation()
print("mean radiance", (acc[..., :3] / acc[..., 3:4]).mean(axis=(0, 1)))
ctx.close()
