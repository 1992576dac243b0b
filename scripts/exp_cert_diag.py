"""Experiment (GPU box): what kind of ray the certified walk gets wrong on sliver soups."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pathtracer_amd as pt
from pathtracer_amd import core
import helpers
ctx = pt.PathTracer(0)
rng = np.random.default_rng(123)
for kind, ntri in (("sliver", 2000), ("unit", 40000)):
    pos, idx = helpers.triangle_soup(rng, ntri, kind)
    nodes, ridx, _ = pt.build_bvh(pos, idx, 8)
    ctx.upload_scene(pos, np.zeros(pos.shape[0], dtype=pt.AUX_DT), ridx, nodes)
    lo, hi = nodes[0]["a"], nodes[0]["b"]
    n = 1 << 22
    rays = helpers.random_rays(lo, hi, n, seed=7)
    k = n // 3
    tri = rng.integers(0, ridx.shape[0], k)
    p0, p1, p2 = (pos[ridx[tri, j], :3] for j in range(3))
    # class A, rays [0, k): lying almost in the plane of a triangle
    w = rng.random((k, 2)).astype(np.float32)
    inplane = p0 + (p1 - p0) * w[:, :1] + (p2 - p0) * w[:, 1:]
    start = p0 + (p1 - p0) * (w[:, 1:] * 3 - 1) + (p2 - p0) * (w[:, :1] * 3 - 1)
    nrm = np.cross(p1 - p0, p2 - p0); nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)
    start = start + nrm * (rng.normal(0, 1e-4, (k, 1)) * np.linalg.norm(p1 - p0, axis=1, keepdims=True)).astype(np.float32)
    d = inplane - start
    rays["o"][:k] = start
    rays["d"][:k] = d / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30)
    # class B, rays [k, 2k): through a vertex or a point of an edge
    tri = rng.integers(0, ridx.shape[0], k)
    p0, p1 = pos[ridx[tri, 0], :3], pos[ridx[tri, 1], :3]
    w = rng.random((k, 1)).astype(np.float32); w[: k // 2] = 0
    d = p0 * (1 - w) + p1 * w - rays["o"][k:2 * k]
    rays["d"][k:2 * k] = d / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30)
    ctx.set_option(core.OPT_TRAVERSAL, 1); ref = ctx.trace_rays(rays)
    ctx.set_option(core.OPT_TRAVERSAL, 3); got = ctx.trace_rays(rays)
    bad = np.nonzero((got.view(np.uint32).reshape(-1, 4) != ref.view(np.uint32).reshape(-1, 4)).any(axis=1))[0]
    print(kind, "differ:", len(bad), "in-plane class:", int((bad < k).sum()), "vertex/edge class:", int(((bad >= k) & (bad < 2 * k)).sum()), "random:", int((bad >= 2 * k).sum()))
    for i in bad[:12]:
        r, g = ref[i], got[i]
        def tri_info(h):
            if h["i1"] < 0: return "miss"
            a, b, c = pos[h["i1"], :3].astype(np.float64), pos[h["i2"], :3].astype(np.float64), pos[h["i3"], :3].astype(np.float64)
            nn = np.cross(b - a, c - a); sinphi = np.linalg.norm(nn) / (np.linalg.norm(b - a) * np.linalg.norm(c - a))
            cosang = abs(np.dot(nn / np.linalg.norm(nn), rays["d"][i].astype(np.float64)))
            o, d = rays["o"][i].astype(np.float64), rays["d"][i].astype(np.float64)
            t_true = np.dot(a - o, nn) / np.dot(d, nn)
            return f"t {h['t']:.7g} (fp64 plane t {t_true:.7g}) ids {h['i1']},{h['i2']},{h['i3']} sin(apex) {sinphi:.2e} |cos(ray,normal)| {cosang:.2e}"
        print(f" ray {i}: ref {tri_info(r)}\n          cert {tri_info(g)}")
