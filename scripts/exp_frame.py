"""Experiment: one frame on the GPU vs the oracle, slot by slot (GPU box)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
import oracle
import pathtracer_amd as pt
from pathtracer_amd import core

W, H, NR = 256, 144, 4
name = sys.argv[1] if len(sys.argv) > 1 else "CBbunny"
pos, aux, idx = pt.load_ply(os.path.join(ROOT, "tests", "golden", "models", name + ".ply"))
nodes, ridx, _ = pt.build_bvh(pos, idx)
model = dict(positions=pos, aux=aux, indices=ridx, nodes=nodes)
lights, bn = pt.default_lights(), pt.load_blue_noise()
ctx = pt.PathTracer(0)
helpers.upload(ctx, model, lights, bn)
ctx.resize(W, H, NR)
osc = helpers.oracle_scene(model, lights, bn)
fr = oracle.Frame(W, H, NR)
for frame in (1, 2, 3):
    cam = dict(orbit_angle=-np.pi / 2, orbit_height=0.75, focus_center=(0.0, 0.75, 0.0)) if name == "CBbunny" else {}
    c = pt.fill_constants(W, H, NR, ridx.shape[0], nodes.shape[0], frame_index=frame, **cam)
    ctx.set_constants(c)
    rc = ctx.render_sample(frame, want_ray_counts=True)
    t = time.time()
    oracle.render_frame(osc, fr, c, subgroup0=0, ray_sort=True, mode=oracle.MODE_ZERO_NEXT_SLOT | oracle.MODE_KILL_ON_MISS)
    dt = time.time() - t
    slots, jobs = ctx.readback_raystack()
    acc = ctx.readback_accumulation()
    cur = ctx.readback_current()
    oslots = fr.stack
    print(f"frame {frame}: gpu ray counts {rc[:2*NR]}  oracle counts {fr.ray_counts()} ({dt:.1f}s)")
    for fld in ("rayDirection", "randState", "rayOrigin", "currentDepth", "hitAlbedo", "wIn", "prob"):
        a, b = slots[fld].reshape(W * H, NR, -1), oslots[fld].reshape(W * H, NR, -1)
        neq = (a != b).any(axis=2)
        print(f"   {fld:14s} slots differing per depth: {neq.sum(axis=0)}")
    oacc = fr.accumulation
    d = acc - oacc
    print(f"   accumulation: exact pixels {(np.abs(d).max(axis=2) == 0).mean():.5f}  rmse {np.sqrt((d[..., :3] ** 2).mean()):.3e}  mean {oacc[..., :3].mean():.4f}")
    print(f"   current image identical: {(cur == fr.current).all(axis=2).mean():.5f}")
ctx.close()
