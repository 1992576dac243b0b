"""Procedural stand-in scenes.

The reference's cornellBox.ply, sponza.ply and SanMiguel.ply are Git-LFS pointer stubs
(Models/*.ply:1-3), so their geometry is not available.  These generators build scenes of the
same character and a stated triangle count, scaled to a few units like the models that are real
(CBbunny spans [-1,1] x [0,1.5] x [-1,1]): the reference's fp16 box test recentres each box in
half precision and degrades on scenes tens of units across.

Every generator returns (positions [V,4] f32 with w = 1, aux [V] of (normal f32x3, RGBA8),
indices [T,3] u32) and is deterministic for a given seed.
"""
import numpy as np

AUX_DT = np.dtype([("normal", "<f4", 3), ("color", "u1", 4)])


class _Mesh:
    def __init__(self):
        self.pos, self.nrm, self.col, self.idx = [], [], [], []
        self.nv = 0

    def add(self, pos, nrm, col, idx):
        pos = np.asarray(pos, dtype=np.float32).reshape(-1, 3)
        nrm = np.asarray(nrm, dtype=np.float32).reshape(-1, 3)
        col = np.broadcast_to(np.asarray(col, dtype=np.uint8), (pos.shape[0], 4)) if np.ndim(col) == 1 else np.asarray(col, dtype=np.uint8)
        self.pos.append(pos); self.nrm.append(nrm); self.col.append(col)
        self.idx.append(np.asarray(idx, dtype=np.int64).reshape(-1, 3) + self.nv)
        self.nv += pos.shape[0]

    def num_triangles(self):
        return sum(i.shape[0] for i in self.idx)

    def finish(self):
        pos = np.concatenate(self.pos); nrm = np.concatenate(self.nrm); col = np.concatenate(self.col)
        idx = np.concatenate(self.idx).astype(np.uint32)
        p4 = np.ones((pos.shape[0], 4), dtype=np.float32)
        p4[:, :3] = pos
        ln = np.linalg.norm(nrm, axis=1, keepdims=True)
        nrm = np.where(ln > 0, nrm / np.maximum(ln, 1e-30), np.array([0, 1, 0], dtype=np.float32)).astype(np.float32)
        aux = np.zeros(pos.shape[0], dtype=AUX_DT)
        aux["normal"], aux["color"] = nrm, col
        # drop zero-area triangles: the reference's intersect() accepts them with a NaN t
        a, b, c = pos[idx[:, 0]], pos[idx[:, 1]], pos[idx[:, 2]]
        area = np.linalg.norm(np.cross(b - a, c - a), axis=1)
        idx = idx[area > 1e-12]
        return p4, aux, np.ascontiguousarray(idx)


def _quad(m, p0, eu, ev, nu, nv, col, height=None, rng=None):
    """A tessellated parallelogram p0 + s*eu + t*ev with nu x nv cells; optional displacement along the normal."""
    p0, eu, ev = (np.asarray(x, dtype=np.float64) for x in (p0, eu, ev))
    s, t = np.meshgrid(np.linspace(0, 1, nu + 1), np.linspace(0, 1, nv + 1), indexing="xy")
    n = np.cross(eu, ev); n /= np.linalg.norm(n)
    pts = p0 + s[..., None] * eu + t[..., None] * ev
    if height is not None:
        pts = pts + (height(s, t))[..., None] * n
    i = (np.arange(nv)[:, None] * (nu + 1) + np.arange(nu)[None, :]).reshape(-1)
    tri = np.concatenate([np.stack([i, i + 1, i + nu + 2], 1), np.stack([i, i + nu + 2, i + nu + 1], 1)])
    m.add(pts.reshape(-1, 3), np.broadcast_to(n, (pts.size // 3, 3)), col, tri)


def _boxes(m, centres, half, col):
    """Batch of axis-aligned boxes: centres [M,3], half extents [M,3]; 12 triangles each, flat normals."""
    centres = np.asarray(centres, dtype=np.float64).reshape(-1, 3)
    half = np.broadcast_to(np.asarray(half, dtype=np.float64), centres.shape)
    M = centres.shape[0]
    faces = []
    for axis in range(3):
        for sgn in (-1.0, 1.0):
            u, v = (axis + 1) % 3, (axis + 2) % 3
            corner = np.zeros((M, 4, 3))
            for k, (su, sv) in enumerate(((-1, -1), (1, -1), (1, 1), (-1, 1))):
                corner[:, k, axis] = centres[:, axis] + sgn * half[:, axis]
                corner[:, k, u] = centres[:, u] + su * half[:, u]
                corner[:, k, v] = centres[:, v] + sv * half[:, v]
            if sgn < 0:
                corner = corner[:, ::-1]
            n = np.zeros(3); n[axis] = sgn
            faces.append((corner, n))
    for corner, n in faces:
        base = np.arange(M) * 4
        tri = np.concatenate([np.stack([base, base + 1, base + 2], 1), np.stack([base, base + 2, base + 3], 1)])
        c = np.repeat(np.asarray(col, dtype=np.uint8).reshape(-1, 4), 4, axis=0) if np.ndim(col) == 2 else col
        m.add(corner.reshape(-1, 3), np.broadcast_to(n, (M * 4, 3)), c, tri)


def _cylinders(m, base, radius, height, seg, col):
    """Batch of vertical cylinders (side walls only): base [M,3], radius [M], height [M]; 2*seg triangles each."""
    base = np.asarray(base, dtype=np.float64).reshape(-1, 3)
    M = base.shape[0]
    radius = np.broadcast_to(np.asarray(radius, dtype=np.float64), (M,))
    height = np.broadcast_to(np.asarray(height, dtype=np.float64), (M,))
    ang = np.linspace(0, 2 * np.pi, seg, endpoint=False)
    ring = np.stack([np.cos(ang), np.zeros(seg), np.sin(ang)], 1)
    lo = base[:, None, :] + radius[:, None, None] * ring[None]
    hi = lo + np.array([0, 1.0, 0])[None, None] * height[:, None, None]
    pts = np.concatenate([lo, hi], axis=1)                      # [M, 2*seg, 3]
    nrm = np.broadcast_to(np.concatenate([ring, ring])[None], pts.shape)
    k = np.arange(seg); k1 = (k + 1) % seg
    t = np.concatenate([np.stack([k, k1, k1 + seg], 1), np.stack([k, k1 + seg, k + seg], 1)])   # [2*seg,3]
    tri = (np.arange(M)[:, None, None] * (2 * seg) + t[None]).reshape(-1, 3)
    c = np.repeat(np.asarray(col, dtype=np.uint8).reshape(-1, 4), 2 * seg, axis=0) if np.ndim(col) == 2 else col
    m.add(pts.reshape(-1, 3), nrm.reshape(-1, 3), c, tri)


def _spheres(m, centre, radius, nlat, nlon, col):
    """Batch of uv-spheres: 2*nlon*(nlat-1) triangles each."""
    centre = np.asarray(centre, dtype=np.float64).reshape(-1, 3)
    M = centre.shape[0]
    radius = np.broadcast_to(np.asarray(radius, dtype=np.float64), (M,))
    th = np.linspace(0, np.pi, nlat + 1)[1:-1]
    ph = np.linspace(0, 2 * np.pi, nlon, endpoint=False)
    T, P = np.meshgrid(th, ph, indexing="ij")
    unit = np.stack([np.sin(T) * np.cos(P), np.cos(T), np.sin(T) * np.sin(P)], -1).reshape(-1, 3)
    unit = np.concatenate([unit, [[0, 1, 0], [0, -1, 0]]])
    nv = unit.shape[0]
    top, bot = nv - 2, nv - 1
    tris = []
    for i in range(nlat - 2):
        a = i * nlon + np.arange(nlon); b = i * nlon + (np.arange(nlon) + 1) % nlon
        tris += [np.stack([a, b, b + nlon], 1), np.stack([a, b + nlon, a + nlon], 1)]
    a = np.arange(nlon); b = (np.arange(nlon) + 1) % nlon
    tris.append(np.stack([np.full(nlon, top), b, a], 1))
    l = (nlat - 2) * nlon
    tris.append(np.stack([np.full(nlon, bot), l + a, l + b], 1))
    t = np.concatenate(tris)
    pts = centre[:, None, :] + radius[:, None, None] * unit[None]
    tri = (np.arange(M)[:, None, None] * nv + t[None]).reshape(-1, 3)
    c = np.repeat(np.asarray(col, dtype=np.uint8).reshape(-1, 4), nv, axis=0) if np.ndim(col) == 2 else col
    m.add(pts.reshape(-1, 3), np.broadcast_to(unit[None], pts.shape).reshape(-1, 3), c, tri)


def _leaves(m, rng, centres, spread, size, per, col):
    """Foliage: `per` small randomly oriented triangles around each centre (unshared vertices)."""
    centres = np.asarray(centres, dtype=np.float64).reshape(-1, 3)
    M = centres.shape[0]
    c0 = centres[:, None, :] + rng.normal(0, 1, (M, per, 3)) * spread
    e1 = rng.normal(0, 1, (M, per, 3)); e1 /= np.linalg.norm(e1, axis=-1, keepdims=True)
    e2 = rng.normal(0, 1, (M, per, 3)); e2 -= (e2 * e1).sum(-1, keepdims=True) * e1
    e2 /= np.maximum(np.linalg.norm(e2, axis=-1, keepdims=True), 1e-9)
    s = size * (0.5 + rng.random((M, per, 1)))
    pts = np.stack([c0, c0 + e1 * s, c0 + e2 * s], axis=2).reshape(-1, 3)
    n = np.repeat(np.cross(e1, e2).reshape(-1, 3), 3, axis=0)
    shade = (90 + rng.integers(0, 100, (M * per, 1))).astype(np.uint8)
    cc = np.concatenate([shade // 3, shade, shade // 4, np.full_like(shade, 255)], 1)
    cc = np.repeat(cc, 3, axis=0) if col is None else col
    m.add(pts, n, cc, np.arange(M * per * 3).reshape(-1, 3))


def cornell_box():
    """Cornell-box stand-in for cornellBox.ply: five walls and two boxes, 34 triangles, spanning
    [-1,1] x [0,2] x [-1,1], open towards +x where the reference's default camera sits."""
    m = _Mesh()
    white, red, green = (242, 242, 242, 255), (200, 30, 30, 255), (30, 200, 30, 255)
    _quad(m, (-1, 0, -1), (0, 0, 2), (2, 0, 0), 1, 1, white)        # floor (normal +y)
    _quad(m, (-1, 2, -1), (2, 0, 0), (0, 0, 2), 1, 1, white)        # ceiling (normal -y)
    _quad(m, (-1, 0, -1), (0, 2, 0), (0, 0, 2), 1, 1, white)        # back wall x=-1 (normal +x)
    _quad(m, (-1, 0, -1), (2, 0, 0), (0, 2, 0), 1, 1, red)          # z=-1 (normal +z)
    _quad(m, (-1, 0, 1), (0, 2, 0), (2, 0, 0), 1, 1, green)         # z=+1 (normal -z)
    _boxes(m, [(-0.3, 0.3, -0.35), (0.1, 0.6, 0.4)], [(0.3, 0.3, 0.3), (0.3, 0.6, 0.3)], white)
    return m.finish()


def courtyard(num_triangles, seed=7, extent=2.0):
    """A walled courtyard with arcades, furniture and trees: a mix of large flat surfaces, regular
    mid-size detail and clouds of tiny foliage triangles, as in architectural scenes of the
    sponza / San Miguel kind.  The triangle count lands within about 1 % of `num_triangles`."""
    rng = np.random.default_rng(seed)
    m = _Mesh()
    E = extent
    budget = int(num_triangles)
    stone, plaster, wood, cloth = (205, 195, 175, 255), (230, 225, 210, 255), (120, 80, 50, 255), (170, 40, 40, 255)

    # ground: bumpy paving, ~18 % of the budget
    g = max(2, int(np.sqrt(0.18 * budget / 2)))
    fx, fz = rng.random(4) * 9 + 3, rng.random(4) * 9 + 3
    _quad(m, (-E, 0, -E), (0, 0, 2 * E), (2 * E, 0, 0), g, g, stone,
          height=lambda s, t: 0.004 * sum(np.sin(fx[k] * s * 6.28) * np.cos(fz[k] * t * 6.28) for k in range(4)))

    # four walls, two storeys, ~10 %
    w = max(1, int(np.sqrt(0.10 * budget / 8 / 2)))
    H = 0.75 * E
    for p0, eu, ev in (((-E, 0, -E), (2 * E, 0, 0), (0, H, 0)), ((E, 0, E), (-2 * E, 0, 0), (0, H, 0)),
                       ((-E, 0, E), (0, 0, -2 * E), (0, H, 0)), ((E, 0, -E), (0, 0, 2 * E), (0, H, 0))):
        _quad(m, p0, eu, ev, 2 * w, w, plaster)

    # arcade columns + arches along two sides, ~12 %
    ncol = 24
    seg = max(6, int(0.12 * budget / (2 * ncol * 2)))
    seg = min(seg, 4096)
    xs = np.linspace(-0.85 * E, 0.85 * E, ncol // 2)
    base = np.concatenate([np.stack([xs, np.zeros_like(xs), np.full_like(xs, -0.7 * E)], 1),
                           np.stack([xs, np.zeros_like(xs), np.full_like(xs, 0.7 * E)], 1)])
    _cylinders(m, base, 0.035 * E, 0.45 * E, seg, stone)
    _boxes(m, base + np.array([0, 0.47 * E, 0]), (0.05 * E, 0.02 * E, 0.05 * E), stone)

    # furniture: tables with a bowl each, ~20 %.  Tables sit on a jittered grid so that no two overlap, and each
    # has its own height: coplanar overlapping faces would make exact t-ties, which real scenes mostly avoid.
    nobj = max(1, int(0.20 * budget / (12 * 5 + 2 * 12 * 10)))
    side = int(np.ceil(np.sqrt(nobj)))
    cell = 1.2 * E / side
    hw = min(0.16, 0.38 * cell)                                                            # half width of a table top
    gx, gz = np.meshgrid(np.arange(side), np.arange(side), indexing="xy")
    keep = rng.permutation(side * side)[:nobj]
    jit = (rng.random((nobj, 2)) - 0.5) * (cell - 2.1 * hw)
    c = np.stack([-0.6 * E + (gx.reshape(-1)[keep] + 0.5) * cell + jit[:, 0], 0.15 + 0.08 * rng.random(nobj),
                  -0.6 * E + (gz.reshape(-1)[keep] + 0.5) * cell + jit[:, 1]], 1)          # c.y = height of the table top
    up = np.array([0.0, 1.0, 0.0])
    _boxes(m, c, (hw, 0.008, hw), wood)                                                    # table top
    leg_h = (c[:, 1] - 0.008) * 0.5
    for dx, dz in ((-0.85, -0.85), (0.85, -0.85), (0.85, 0.85), (-0.85, 0.85)):
        lc = c.copy(); lc[:, 0] += dx * hw; lc[:, 2] += dz * hw; lc[:, 1] = leg_h
        _boxes(m, lc, np.stack([np.full(nobj, 0.05 * hw), leg_h, np.full(nobj, 0.05 * hw)], 1), wood)   # legs
    _spheres(m, c + up * (0.008 + 0.19 * hw), 0.19 * hw, 12, 10, cloth)                    # a bowl on each table

    # trees: trunk + foliage clouds; foliage takes what is left of the budget
    ntree = max(4, int(np.sqrt(budget) / 40))
    tc = np.stack([rng.uniform(-0.75 * E, 0.75 * E, ntree), np.zeros(ntree), rng.uniform(-0.5 * E, 0.5 * E, ntree)], 1)
    _cylinders(m, tc, 0.03 * E, 0.35 * E, 24, wood)
    left = budget - m.num_triangles()
    if left > 0:
        ncl = ntree * 12
        per = max(1, left // ncl)
        cc = np.repeat(tc, 12, axis=0) + np.array([0, 0.45 * E, 0]) + rng.normal(0, 0.09 * E, (ncl, 3))
        _leaves(m, rng, cc, 0.05 * E, 0.012 * E, per, None)
    return m.finish()


# A camera for the courtyard stand-ins (keyword arguments of fill_constants): in a corner above the tree crowns, looking
# down into the yard, so that the picture holds walls, arcade, furniture and foliage instead of the inside of a crown.
COURTYARD_CAMERA = dict(orbit_angle=2.4, orbit_radius=2.6, orbit_height=1.4, focus_center=(0.0, 0.2, 0.0))


def sponza_like(num_triangles=262_144, seed=11):
    """Stand-in for sponza.ply (a 16.8 MB PLY whose geometry is not in the reference checkout): an atrium of the
    Sponza kind, different in character from the courtyard stand-in - a long nave between two-storey colonnades, the
    nave open to the sky, a few large occluders (walls, gallery slabs, columns) and rows of draped cloth
    that carry a third of the triangles; no foliage soup.  Spans [-2,2] x [0,1.7] x [-0.9,0.9]."""
    rng = np.random.default_rng(seed)
    m = _Mesh()
    budget = int(num_triangles)
    stone, plaster, floor_c = (200, 190, 170, 255), (225, 218, 200, 255), (150, 140, 125, 255)
    cloths = [(170, 40, 40, 255), (40, 70, 150, 255), (50, 130, 60, 255), (190, 160, 60, 255)]
    LX, HY, HZ = 2.0, 1.7, 0.9

    # floor: worn paving, ~14 %
    g = max(2, int(np.sqrt(0.14 * budget / 2 / 2.2)))
    fx, fz = rng.random(3) * 7 + 2, rng.random(3) * 7 + 2
    _quad(m, (-LX, 0, -HZ), (0, 0, 2 * HZ), (2 * LX, 0, 0), g, int(2.2 * g), floor_c,
          height=lambda s_, t_: 0.003 * sum(np.sin(fx[k] * s_ * 6.28) * np.cos(fz[k] * t_ * 6.28) for k in range(3)))
    # long walls, end walls, and eaves over the galleries: the nave is open to the sky, ~12 %
    w = max(1, int(np.sqrt(0.12 * budget / 2 / 12)))
    _quad(m, (-LX, 0, -HZ), (2 * LX, 0, 0), (0, HY, 0), 4 * w, w, plaster)                  # z = -HZ, normal +z
    _quad(m, (LX, 0, HZ), (-2 * LX, 0, 0), (0, HY, 0), 4 * w, w, plaster)                   # z = +HZ, normal -z
    _quad(m, (-LX, 0, HZ), (0, 0, -2 * HZ), (0, HY, 0), 2 * w, w, plaster)                  # x = -LX, normal +x
    _quad(m, (LX, 0, -HZ), (0, 0, 2 * HZ), (0, HY, 0), 2 * w, w, plaster)                   # x = +LX, normal -x
    _quad(m, (-LX, HY, -HZ), (2 * LX, 0, 0), (0, 0, 0.30 * HZ), 2 * w, max(1, w // 3), plaster)     # eaves over the -z gallery (normal -y)
    _quad(m, (LX, HY, HZ), (-2 * LX, 0, 0), (0, 0, -0.30 * HZ), 2 * w, max(1, w // 3), plaster)     # eaves over the +z gallery

    # colonnades: two rows of columns on two storeys, lintels and gallery slabs, ~26 %
    ncol = 14
    seg = int(min(2048, max(8, 0.26 * budget * 0.8 / (4 * ncol * 2))))
    xs = np.linspace(-0.9 * LX, 0.9 * LX, ncol)
    for zrow in (-0.52 * HZ, 0.52 * HZ):
        for y0, hgt, rad in ((0.0, 0.72, 0.045), (0.80, 0.62, 0.035)):
            base = np.stack([xs, np.full_like(xs, y0), np.full_like(xs, zrow)], 1)
            _cylinders(m, base, rad, hgt, seg, stone)
            _boxes(m, base + np.array([0, hgt + 0.02, 0]), (0.07, 0.02, 0.07), stone)       # capitals
        sign = 1.0 if zrow > 0 else -1.0
        _boxes(m, [(0.0, 0.77, zrow + sign * 0.24 * HZ)], [(LX, 0.03, 0.24 * HZ + 0.05)], plaster)  # gallery floor slab
        _boxes(m, [(0.0, 0.745, zrow)], [(LX, 0.025, 0.05)], stone)                          # lintel over the ground-floor columns
        _boxes(m, [(0.0, 1.465, zrow)], [(LX, 0.025, 0.05)], stone)                          # and over the gallery's

    # vases along the nave, ~4 %
    nv = 8
    lat = max(6, int(np.sqrt(0.04 * budget / nv / 2)))
    vc = np.stack([np.linspace(-0.8 * LX, 0.8 * LX, nv), np.full(nv, 0.09), np.where(np.arange(nv) % 2 == 0, -0.25, 0.25) * HZ], 1)
    _spheres(m, vc, 0.09, lat, lat, stone)

    # draped cloth: hanging banners across the nave and curtains between gallery columns; they take what is left
    left = budget - m.num_triangles()
    nb = 10
    bays = [(zrow, k) for zrow in (-0.52 * HZ, 0.52 * HZ) for k in range(ncol - 1) if (k + (zrow > 0)) % 3 != 0]   # every third bay stays open
    per = max(8, left // (nb + len(bays)))                                   # triangles per piece of cloth
    bu = max(2, int(round(np.sqrt(per / 2 / 1.6)))); bv = max(2, int(round(per / 2 / bu)))
    for k, x in enumerate(np.linspace(-0.7 * LX, 0.55 * LX, nb)):
        ph = rng.random(3) * 6.28
        _quad(m, (x, 1.50, -0.30 * HZ), (0, 0, 0.60 * HZ), (0, -0.55, 0), bu, bv, cloths[k % 4],
              height=lambda s_, t_, ph=ph: 0.035 * np.sin(5.0 * s_ * 6.28 + ph[0]) * (0.3 + t_) + 0.02 * np.sin(2.0 * t_ * 6.28 + ph[1]))
    cu = max(2, int(round(np.sqrt(per / 2)))); cv = max(2, int(round(per / 2 / cu)))
    for zrow, k in bays:
        ph = rng.random(2) * 6.28
        _quad(m, (xs[k] + 0.04, 1.40, zrow), (xs[k + 1] - xs[k] - 0.08, 0, 0), (0, -0.52, 0), cu, cv, cloths[(k + 1) % 4],
              height=lambda s_, t_, ph=ph: 0.03 * np.sin(4.0 * s_ * 6.28 + ph[0]) * (0.2 + t_) + 0.015 * np.sin(3.0 * t_ * 6.28 + ph[1]))
    return m.finish()


# A camera for the atrium stand-in: at one end of the nave, a little above head height, looking down its length.
SPONZA_CAMERA = dict(orbit_angle=0.0, orbit_radius=1.75, orbit_height=0.15, focus_center=(0.0, 0.55, 0.0))


def san_miguel_like(num_triangles=5_800_000, seed=13):
    """Stand-in for SanMiguel.ply.  The real file is 156.6 MB; at 28 B per vertex and 13 B per
    triangle with about two triangles per vertex that is roughly 5.8 M triangles, the default here."""
    return courtyard(num_triangles, seed=seed, extent=2.0)
