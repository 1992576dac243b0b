"""Experiment (GPU box): frames with PT_OPT_TRAVERSAL = 3 against = 1 on the stand-in scenes at 1080p: throughput, and
whether accumulation, current image and ray stack are the same bytes."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt
from pathtracer_amd import core
W, H, NR, NF = 1920, 1080, 8, 12
for scene in ("sponza", "sanmiguel"):
    pos, aux, idx = pt.scenes.sponza_like() if scene == "sponza" else pt.scenes.san_miguel_like()
    nodes, ridx, _ = pt.build_bvh(pos, idx, 16)
    out = {}
    for trav in (1, 3, 2):
        ctx = pt.PathTracer(0)
        ctx.upload_scene(pos, aux, ridx, nodes, pt.default_lights(), pt.load_blue_noise())
        ctx.set_option(core.OPT_TRAVERSAL, trav)
        ctx.resize(W, H, NR)
        ctx.set_constants(pt.fill_constants(W, H, NR, ridx.shape[0], nodes.shape[0], **pt.scenes.COURTYARD_CAMERA))
        ctx.render_samples(1, 3); ctx.synchronize()
        t = time.perf_counter(); ctx.render_samples(4, NF); ctx.synchronize(); dt = time.perf_counter() - t
        out[trav] = (ctx.readback_accumulation(), ctx.readback_current())
        print(f"{scene} traversal {trav}: {NF / dt:.1f} spp/s", flush=True)
        ctx.close()
    for trav in (3, 2):
        same = all(np.array_equal(a.view(np.uint8), b.view(np.uint8)) for a, b in zip(out[trav], out[1]))
        npix = int((np.abs(out[trav][0] - out[1][0]).max(axis=2) > 0).sum())
        print(f"{scene}: traversal {trav} against 1 after {NF + 3} frames: {'identical' if same else 'differs'} ({npix} of {W * H} pixels differ)", flush=True)
