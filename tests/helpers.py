"""Shared helpers of the test-suite: synthetic rays, oracle scene construction."""
import numpy as np

import oracle
import pathtracer_amd as pt


def random_rays(root_a, root_b, n, seed, inside=True):
    """Incoherent rays: origin uniform in the box, direction uniform on the sphere (numpy twin of pt_generate_rays)."""
    rng = np.random.default_rng(seed)
    rays = np.zeros(n, dtype=pt.RAY_DT)
    a, b = np.asarray(root_a, dtype=np.float32), np.asarray(root_b, dtype=np.float32)
    rays["o"] = a + (b - a) * rng.random((n, 3), dtype=np.float32)
    if not inside:
        rays["o"] = (a + b) * 0.5 + (rays["o"] - (a + b) * 0.5) * 2.5
    z = 1 - 2 * rng.random(n); phi = 2 * np.pi * rng.random(n); r = np.sqrt(np.maximum(0, 1 - z * z))
    rays["d"] = np.stack([r * np.cos(phi), r * np.sin(phi), z], 1).astype(np.float32)
    rays["min_t"] = 0.001
    rays["max_t"] = 1e5
    return rays


def camera_rays(W, H, eye=(3.0, 0.5, 0.0), target=(0.0, 0.0, 0.0), fov_deg=60.0):
    """Primary-like rays from the reference's default camera position (PathTracer.cpp:109-112)."""
    eye = np.asarray(eye, dtype=np.float64); target = np.asarray(target, dtype=np.float64)
    f = target - eye; f /= np.linalg.norm(f)
    s = np.cross(f, [0, 1, 0]); s /= np.linalg.norm(s)
    u = np.cross(s, f)
    th = np.tan(np.radians(fov_deg) / 2)
    xs = ((np.arange(W) + 0.5) / W * 2 - 1) * th * (W / H)
    ys = ((np.arange(H) + 0.5) / H * 2 - 1) * th
    X, Y = np.meshgrid(xs, ys)
    d = f[None, None] + X[..., None] * s + Y[..., None] * u
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    rays = np.zeros(W * H, dtype=pt.RAY_DT)
    rays["o"] = eye.astype(np.float32)
    rays["d"] = d.reshape(-1, 3).astype(np.float32)
    rays["min_t"] = 0.001
    rays["max_t"] = 1e5
    return rays


def oracle_scene(model, lights=None, blue_noise=None):
    return oracle.Scene(model["positions"], model["aux"], model["indices"], model["nodes"], lights, blue_noise)


def upload(ctx, model, lights=None, blue_noise=None):
    ctx.upload_scene(model["positions"], model["aux"], model["indices"], model["nodes"], lights, blue_noise)


def triangle_soup(rng, n, kind):
    """Random triangle soups that stress the arithmetic: sizes from 2 triangles up, flat axis-aligned triangles (boxes
    of zero thickness: only the builder's 6e-5 pad, an fp16 subnormal, keeps them open), slivers, shared vertices,
    and coordinates far from the origin or beyond the fp16 range (the box digest then holds infinities)."""
    if kind == "unit":
        v = rng.uniform(-1, 1, (3 * n, 3))
    elif kind == "flat":
        v = rng.uniform(-1, 1, (3 * n, 3))
        axis = rng.integers(0, 3, n)
        for t in range(n):
            v[3 * t:3 * t + 3, axis[t]] = np.round(rng.uniform(-1, 1), 2)
    elif kind == "sliver":
        a = rng.uniform(-1, 1, (n, 3)); d = rng.normal(0, 1, (n, 3))
        v = np.stack([a, a + d, a + d * (1 + 1e-4) + rng.normal(0, 1e-5, (n, 3))], 1).reshape(-1, 3)
    elif kind == "offset":
        v = rng.uniform(-1, 1, (3 * n, 3)) * 0.25 + np.array([900.0, -350.0, 1200.0])
    else:   # "huge": extents beyond 65504, the largest fp16
        v = rng.uniform(-1, 1, (3 * n, 3)) * 2.0e5
    pos = np.ones((3 * n, 4), np.float32)
    pos[:, :3] = v.astype(np.float32)
    idx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    if n > 4:                                   # let some triangles share vertices
        idx[1, 0] = idx[0, 0]; idx[2, 1] = idx[0, 2]
    a, b, c = pos[idx[:, 0], :3], pos[idx[:, 1], :3], pos[idx[:, 2], :3]
    keep = np.linalg.norm(np.cross(b - a, c - a), axis=1) > 0      # zero-area triangles: the documented NaN deviation
    return pos, np.ascontiguousarray(idx[keep])
