"""Produces the golden vectors of tests/golden from the CPU oracle.

The reference ships no tests or fixtures (parity unpinned), so these vectors pin the oracle itself
against regressions: inputs are committed model files and seeded rays, outputs are whatever the
oracle returned when this script was run.  Run from the repository root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
import oracle  # noqa: E402
import pathtracer_amd as pt  # noqa: E402

GOLD = os.path.dirname(os.path.abspath(__file__))


def load(name):
    pos, aux, idx = pt.load_ply(os.path.join(GOLD, "models", name))
    nodes, ridx, depth = oracle.build_bvh(pos, idx)
    return dict(positions=pos, aux=aux, indices=ridx, nodes=nodes)


bunny = load("CBbunny.ply")
osc = helpers.oracle_scene(bunny)
rays = np.concatenate([helpers.random_rays(bunny["nodes"][0]["a"], bunny["nodes"][0]["b"], 3000, seed=42),
                       helpers.camera_rays(64, 36, eye=(0.0, 0.75, -3.0), target=(0.0, 0.75, 0.0))])
np.savez_compressed(os.path.join(GOLD, "bunny_trace_golden.npz"), rays=rays.view(np.uint8),
                    closest=osc.trace(rays).view(np.uint8), anyhit=osc.trace(rays, any_hit=True).view(np.uint8))

monkey = load("CBmonkey.ply")
W, H, NR = 64, 36, 4
osc = helpers.oracle_scene(monkey, pt.default_lights(), pt.load_blue_noise())
fr = oracle.Frame(W, H, NR)
for frame in (1, 2):
    c = pt.fill_constants(W, H, NR, monkey["indices"].shape[0], monkey["nodes"].shape[0], frame_index=frame)
    oracle.render_frame(osc, fr, c)
np.savez_compressed(os.path.join(GOLD, "monkey_frame_golden.npz"), shape=np.array([W, H, NR]), accumulation=fr.accumulation.copy(),
                    stack=fr.stack.view(np.uint8).copy())
print("golden vectors written")
