"""Renders a scene to a PNG/PPM with the tone map of composite.frag (GPU box)."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt
ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="bunny")
ap.add_argument("--width", type=int, default=640)
ap.add_argument("--height", type=int, default=360)
ap.add_argument("--slots", type=int, default=5)
ap.add_argument("--spp", type=int, default=64)
ap.add_argument("--out", default="gpurun_out/render.png")
ap.add_argument("--angle", type=float, default=None)
ap.add_argument("--radius", type=float, default=None)
ap.add_argument("--cheight", type=float, default=None)
args = ap.parse_args()
cam = {}
if args.scene in ("bunny", "monkey"):
    pos, aux, idx = pt.load_ply(os.path.join(ROOT, "tests", "golden", "models", "CB%s.ply" % args.scene))
    if args.scene == "bunny":
        cam = dict(orbit_angle=-np.pi / 2, orbit_height=0.75, focus_center=(0.0, 0.75, 0.0))
elif args.scene == "cornell":
    pos, aux, idx = pt.scenes.cornell_box()
elif args.scene == "sanmiguel":
    pos, aux, idx = pt.scenes.san_miguel_like()
    cam = dict(pt.scenes.COURTYARD_CAMERA)
else:
    pos, aux, idx = pt.scenes.sponza_like()
    cam = dict(pt.scenes.SPONZA_CAMERA)
if args.angle is not None: cam["orbit_angle"] = args.angle
if args.radius is not None: cam["orbit_radius"] = args.radius
if args.cheight is not None: cam["orbit_height"] = args.cheight
print("bounds", pos[:, :3].min(axis=0), pos[:, :3].max(axis=0), "tris", idx.shape[0])
nodes, ridx, _ = pt.build_bvh(pos, idx, 16)
ctx = pt.PathTracer(0)
ctx.upload_scene(pos, aux, ridx, nodes, pt.default_lights(), pt.load_blue_noise())
W, H, NR = args.width, args.height, args.slots
ctx.resize(W, H, NR)
ctx.set_constants(pt.fill_constants(W, H, NR, ridx.shape[0], nodes.shape[0], **cam))
ctx.render_samples(1, args.spp)
img = ctx.tonemap()
acc = ctx.readback_accumulation()
print("mean radiance", (acc[..., :3] / acc[..., 3:4]).mean(axis=(0, 1)), "mean display", img[..., :3].mean(axis=(0, 1)))
if args.out.endswith(".ppm"):
    pt.write_ppm(args.out, img)
else:
    from PIL import Image
    Image.fromarray(img[..., :3]).save(args.out)
ctx.close()
