"""ctypes binding of oracle/libpt_oracle.so (the CPU restatement; test infrastructure only)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(ROOT, "oracle", "libpt_oracle.so")

NODE_DT = np.dtype([("a", "<f4", 3), ("next", "<i4"), ("b", "<f4", 3), ("right", "<i4")])
RAY_DT = np.dtype([("o", "<f4", 3), ("min_t", "<f4"), ("d", "<f4", 3), ("max_t", "<f4")])
HIT_DT = np.dtype([("t", "<f4"), ("i1", "<i4"), ("i2", "<i4"), ("i3", "<i4")])
LIGHT_DT = np.dtype([("pos", "<f4", 3), ("_p0", "<f4"), ("radiance", "<f4", 3), ("_p1", "<f4")])
AUX_DT = np.dtype([("normal", "<f4", 3), ("color", "u1", 4)])
SLOT_DT = np.dtype([("rayDirection", "<f4", 3), ("randState", "<u4"), ("rayOrigin", "<f4", 3), ("currentDepth", "<u4"),
                    ("hitAlbedo", "<u2", 4), ("wIn", "<u2", 3), ("prob", "<u2")])
CONST_DT = np.dtype([("viewMtx", "<f4", 16), ("projMtx", "<f4", 16), ("viewInvMtx", "<f4", 16), ("projInvMtx", "<f4", 16),
                     ("viewportSize", "<f4", 2), ("viewportBase", "<f4", 2), ("numLights", "<u4"), ("numTriangles", "<u4"),
                     ("frameIndex", "<u4"), ("numRays", "<u4"), ("numBVHNodes", "<u4"), ("_pad0", "<u4", 3),
                     ("ambientRadiance", "<f4", 3), ("_pad1", "<u4")])
assert NODE_DT.itemsize == 32 and SLOT_DT.itemsize == 48 and CONST_DT.itemsize == 320 and AUX_DT.itemsize == 16

FLAG_REJECT_NAN = 1
FLAG_SKIP_ODD_INTERIOR = 2      # analysis only (scripts/exp_wide_nodes.py)
MODE_ZERO_NEXT_SLOT, MODE_RESEED_EVERY_FRAME, MODE_KILL_ON_MISS = 1, 2, 4
MODE_PLAIN_SHADOW_WALK = 8       # analysis only: the SPECULATIVE build without the vote in its shadow walk

_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB_PATH)
        vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
        L.po_build_bvh.argtypes = [vp, vp, u32, vp, u32, C.POINTER(u32), C.POINTER(u32)]
        L.po_build_bvh.restype = i32
        L.po_scene_create.argtypes = [vp, u32, vp, vp, u32, vp, u32, vp, u32, vp]
        L.po_scene_create.restype = vp
        L.po_scene_destroy.argtypes = [vp]
        L.po_trace_rays.argtypes = [vp, vp, u64, i32, i32, vp, vp, vp, vp, i32]
        L.po_trace_rays_group.argtypes = [vp, vp, u64, i32, i32, i32, i32, vp]
        L.po_frame_create.argtypes = [u32, u32, u32]
        L.po_frame_create.restype = vp
        L.po_frame_destroy.argtypes = [vp]
        for n in ("po_frame_stack", "po_frame_jobs", "po_frame_accumulation", "po_frame_current"):
            getattr(L, n).argtypes = [vp]
            getattr(L, n).restype = vp
        L.po_frame_clear_accumulation.argtypes = [vp]
        L.po_frame_ray_counts.argtypes = [vp, vp]
        L.po_launch.argtypes = [vp, vp, vp, i32]
        L.po_trace.argtypes = [vp, vp, vp, i32, i32, i32]
        L.po_raysort.argtypes = [vp, vp]
        L.po_composite.argtypes = [vp, vp]
        L.po_tonemap.argtypes = [vp, C.c_uint64, vp]
        L.po_render_frame.argtypes = [vp, vp, vp, i32, i32, i32, i32]
        L.po_render_frame_mt.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32]
        L.po_intersect_bbox.argtypes = [vp, vp, vp]
        L.po_intersect_bbox.restype = i32
        L.po_weyl.argtypes = [u32, vp]
        L.po_cosine_hemisphere.argtypes = [vp, vp]
        L.po_to_coord_space.argtypes = [vp, vp, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def build_bvh(positions, indices):
    """positions [V,4] f32, indices [T,3] u32 -> (nodes, reordered indices, max_depth)."""
    idx = np.ascontiguousarray(indices, dtype=np.uint32).copy()
    T = idx.shape[0]
    nodes = np.zeros(2 * T - 1, dtype=NODE_DT)
    nn, md = C.c_uint32(0), C.c_uint32(0)
    rc = lib().po_build_bvh(_p(np.ascontiguousarray(positions, dtype=np.float32)), _p(idx), 3 * T, _p(nodes), nodes.shape[0],
                            C.byref(nn), C.byref(md))
    assert rc == 0, rc
    assert nn.value == 2 * T - 1
    return nodes, idx, md.value


class Scene:
    def __init__(self, positions, aux, indices, nodes, lights=None, blue_noise=None):
        self.positions = np.ascontiguousarray(positions, dtype=np.float32)
        self.aux = np.ascontiguousarray(aux) if aux is not None else None
        self.indices = np.ascontiguousarray(indices, dtype=np.uint32)
        self.nodes = np.ascontiguousarray(nodes)
        self.lights = np.ascontiguousarray(lights) if lights is not None else None
        self.blue_noise = np.ascontiguousarray(blue_noise, dtype=np.uint16) if blue_noise is not None else None
        self.h = lib().po_scene_create(_p(self.positions), self.positions.shape[0], _p(self.aux), _p(self.indices),
                                       self.indices.shape[0], _p(self.nodes), self.nodes.shape[0], _p(self.lights),
                                       0 if self.lights is None else self.lights.shape[0], _p(self.blue_noise))

    def __del__(self):
        if getattr(self, "h", None):
            lib().po_scene_destroy(self.h)
            self.h = None

    def trace(self, rays, any_hit=False, flags=FLAG_REJECT_NAN, threads=0, want_bary=False, want_counters=False, want_ranoff=False):
        rays = np.ascontiguousarray(rays)
        n = rays.shape[0]
        hits = np.zeros(n, dtype=HIT_DT)
        bary = np.zeros((n, 3), dtype=np.uint16) if want_bary else None
        ran = np.zeros(n, dtype=np.uint32) if want_ranoff else None
        cnt = np.zeros(2, dtype=np.uint64) if want_counters else None
        lib().po_trace_rays(self.h, _p(rays), n, int(any_hit), flags, _p(hits), _p(bary), _p(ran), _p(cnt), threads)
        out = [hits]
        if want_bary:
            out.append(bary)
        if want_ranoff:
            out.append(ran)
        if want_counters:
            out.append(cnt)
        return out[0] if len(out) == 1 else tuple(out)

    def trace_group(self, rays, group, speculative, any_hit=False, flags=FLAG_REJECT_NAN):
        rays = np.ascontiguousarray(rays)
        hits = np.zeros(rays.shape[0], dtype=HIT_DT)
        lib().po_trace_rays_group(self.h, _p(rays), rays.shape[0], int(any_hit), flags, group, int(speculative), _p(hits))
        return hits


class Frame:
    def __init__(self, W, H, slots):
        self.W, self.H, self.slots = W, H, slots
        self.h = lib().po_frame_create(W, H, slots)

    def __del__(self):
        if getattr(self, "h", None):
            lib().po_frame_destroy(self.h)
            self.h = None

    def _view(self, fn, dtype, count):
        ptr = getattr(lib(), fn)(self.h)
        buf = (C.c_char * (np.dtype(dtype).itemsize * count)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype, count=count)

    @property
    def stack(self):
        return self._view("po_frame_stack", SLOT_DT, self.W * self.H * self.slots)

    @property
    def jobs(self):
        return self._view("po_frame_jobs", np.uint32, self.W * self.H)

    @property
    def accumulation(self):
        return self._view("po_frame_accumulation", np.float32, self.W * self.H * 4).reshape(self.H, self.W, 4)

    @property
    def current(self):
        return self._view("po_frame_current", np.uint16, self.W * self.H * 4).reshape(self.H, self.W, 4)

    def clear_accumulation(self):
        lib().po_frame_clear_accumulation(self.h)

    def ray_counts(self):
        out = np.zeros(4, dtype=np.uint64)
        lib().po_frame_ray_counts(self.h, _p(out))
        return out


def launch(scene, frame, consts, mode=MODE_ZERO_NEXT_SLOT):
    lib().po_launch(scene.h, frame.h, _p(consts), mode)


def trace(scene, frame, consts, subgroup=0, mode=MODE_ZERO_NEXT_SLOT, flags=FLAG_REJECT_NAN):
    lib().po_trace(scene.h, frame.h, _p(consts), subgroup, mode, flags)


def raysort(frame, consts):
    lib().po_raysort(frame.h, _p(consts))


def composite(frame, consts):
    lib().po_composite(frame.h, _p(consts))


def tonemap(accumulation):
    """(H, W, 4) float32 accumulation -> (H, W, 3) float32 display colour (composite.frag:18-26,57)."""
    acc = np.ascontiguousarray(accumulation, dtype=np.float32)
    out = np.zeros(acc.shape[:2] + (3,), dtype=np.float32)
    lib().po_tonemap(_p(acc), acc.shape[0] * acc.shape[1], _p(out))
    return out


def render_frame(scene, frame, consts, subgroup0=0, ray_sort=True, mode=MODE_ZERO_NEXT_SLOT, flags=FLAG_REJECT_NAN, threads=1):
    if threads == 1:
        lib().po_render_frame(scene.h, frame.h, _p(consts), subgroup0, int(ray_sort), mode, flags)
    else:
        lib().po_render_frame_mt(scene.h, frame.h, _p(consts), subgroup0, int(ray_sort), mode, flags, threads)
