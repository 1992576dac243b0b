"""Experiment (GPU box): the certified walk against the reference-order walk on geometry and rays chosen to hurt -
sliver and flat soups, rays lying almost in triangle planes, rays through vertices, tiny and huge scales."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pathtracer_amd as pt
from pathtracer_amd import core
import helpers

ctx = pt.PathTracer(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 123)
total = bad = 0
for kind in ("sliver", "flat", "unit", "offset", "huge"):
    for ntri in (2000, 40000):
        for scale in (1.0, 1e-3, 300.0):
            pos, idx = helpers.triangle_soup(rng, ntri, kind)
            pos = pos.copy(); pos[:, :3] *= np.float32(scale)
            if idx.shape[0] < 2:
                continue
            nodes, ridx, _ = pt.build_bvh(pos, idx, 8)
            ctx.upload_scene(pos, np.zeros(pos.shape[0], dtype=pt.AUX_DT), ridx, nodes)
            lo, hi = nodes[0]["a"], nodes[0]["b"]
            n = 1 << 20
            rays = helpers.random_rays(lo, hi, n, seed=int(rng.integers(1 << 30)))
            # a third of the rays lie almost in the plane of some triangle, a third pass through a vertex or an edge point
            k = n // 3
            tri = rng.integers(0, ridx.shape[0], k)
            p0, p1, p2 = (pos[ridx[tri, j], :3] for j in range(3))
            w = rng.random((k, 2)).astype(np.float32)
            inplane = p0 + (p1 - p0) * w[:, :1] + (p2 - p0) * w[:, 1:]
            start = p0 + (p1 - p0) * (w[:, 1:] * 3 - 1) + (p2 - p0) * (w[:, :1] * 3 - 1)
            nrm = np.cross(p1 - p0, p2 - p0); nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)
            start = start + nrm * (rng.normal(0, 1e-4, (k, 1)) * np.linalg.norm(p1 - p0, axis=1, keepdims=True)).astype(np.float32)
            d = inplane - start
            rays["o"][:k] = start; rays["d"][:k] = d / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30)
            tri = rng.integers(0, ridx.shape[0], k)
            q0, q1 = pos[ridx[tri, 0], :3], pos[ridx[tri, 1], :3]
            w = rng.random((k, 1)).astype(np.float32); w[: k // 2] = 0
            target = q0 * (1 - w) + q1 * w
            d = target - rays["o"][k:2 * k]
            rays["d"][k:2 * k] = d / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30)
            ctx.set_option(core.OPT_TRAVERSAL, 1); ref = ctx.trace_rays(rays)
            ctx.set_option(core.OPT_TRAVERSAL, 3); got, st = ctx.trace_rays(rays, want_stats=True)
            ctx.set_option(core.OPT_TRAVERSAL, 2); fast = ctx.trace_rays(rays)
            diff = int((got.view(np.uint32).reshape(-1, 4) != ref.view(np.uint32).reshape(-1, 4)).any(axis=1).sum())
            dfast = int((fast.view(np.uint32).reshape(-1, 4) != ref.view(np.uint32).reshape(-1, 4)).any(axis=1).sum())
            total += n; bad += diff
            print(f"{kind:7s} T={ridx.shape[0]:6d} scale {scale:g}: certified differs on {diff}, near-first on {dfast}, second walks {int(st[3]) - n}, "
                  f"hit rate {(ref['i1'] >= 0).mean():.3f}", flush=True)
print(f"TOTAL: {bad} of {total} rays differ")
