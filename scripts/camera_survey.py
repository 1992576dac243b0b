"""Renders a contact sheet of candidate cameras for a stand-in scene (GPU box)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt
from PIL import Image
scene = sys.argv[1]
pos, aux, idx = pt.scenes.san_miguel_like() if scene == "sanmiguel" else pt.scenes.sponza_like()
nodes, ridx, _ = pt.build_bvh(pos, idx, 16)
ctx = pt.PathTracer(0)
ctx.upload_scene(pos, aux, ridx, nodes, pt.default_lights(), pt.load_blue_noise())
W, H, NR = 320, 180, 5
ctx.resize(W, H, NR)
cams = [dict(orbit_angle=a, orbit_radius=r, orbit_height=h, focus_center=(0.0, fy, 0.0))
        for (a, r, h, fy) in [(0.0, 1.9, 0.5, 0.4), (1.57, 1.9, 0.5, 0.4), (0.8, 2.5, 1.2, 0.3), (3.14, 1.9, 0.9, 0.3),
                              (0.0, 0.3, 0.4, 0.4), (2.4, 2.6, 1.4, 0.2), (0.0, 1.0, 2.5, 0.0), (4.0, 1.2, 0.3, 0.6)]]
sheet = Image.new("RGB", (4 * W, 2 * H))
for k, cam in enumerate(cams):
    ctx.clear_accumulation()
    ctx.set_constants(pt.fill_constants(W, H, NR, ridx.shape[0], nodes.shape[0], **cam))
    ctx.render_samples(1, 48)
    img = ctx.tonemap()
    print(k, cam, "mean display", img[..., :3].mean())
    sheet.paste(Image.fromarray(img[..., :3]), ((k % 4) * W, (k // 4) * H))
sheet.save(os.path.join(ROOT, "gpurun_out", f"survey_{scene}.png"))
ctx.close()
