"""ctypes host layer over libptcore.so.

Names follow the reference: `build_bvh` is BuildBVH (BVH.cpp:83), `PathTracer.launch / trace /
compact / resolve_accumulate` are the dispatch sites of PathTracer.cpp:776-838, `fill_constants`
fills the block of PathTracer.cpp:731-750, `upload_scene` / `resize` create the buffers of
PathTracer.cpp:272-321 and :378-459.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

NODE_DT = np.dtype([("a", "<f4", 3), ("next", "<i4"), ("b", "<f4", 3), ("right", "<i4")])
NODE2_DT = np.dtype([("c", "<f4", (3, 2)), ("h", "<u2", (6, 2)), ("left", "<u4"), ("right", "<u4"),
                     ("refIndex", "<u4"), ("wmax", "<u2", 2)])
TRI_DT = np.dtype([("p1", "<f4", 3), ("i1", "<i4"), ("e1", "<f4", 3), ("i2", "<i4"), ("e2", "<f4", 3), ("i3", "<i4"),
                   ("refLeaf", "<u4"), ("flags", "<u4"), ("_pad", "<u4", 2)])
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
NODE2C_DT = np.dtype([("c", "<f4", (3, 2)), ("hA", "<u2", (3, 2)), ("left", "<u4"), ("right", "<u4"), ("delta", "<u4")])
TRIC_DT = np.dtype([("p1", "<f4", 3), ("i1", "<i4"), ("e1", "<f4", 3), ("i2", "<i4"), ("e2", "<f4", 3), ("i3", "<i4")])
assert (NODE_DT.itemsize, NODE2_DT.itemsize, TRI_DT.itemsize, RAY_DT.itemsize, HIT_DT.itemsize) == (32, 64, 64, 32, 16)
PAIR_DT = np.dtype([("c", "<f4", (3, 2)), ("hA", "<u2", (3, 2)), ("delta", "<u4")])
TREELET_DT = np.dtype([("pair", PAIR_DT, 3), ("link", "<u4", 2)])
assert (NODE2C_DT.itemsize, TRIC_DT.itemsize) == (48, 48) and (PAIR_DT.itemsize, TREELET_DT.itemsize) == (40, 128)
assert (LIGHT_DT.itemsize, AUX_DT.itemsize, SLOT_DT.itemsize, CONST_DT.itemsize) == (32, 16, 48, 320)

OPT_TRAVERSAL, OPT_STALE_SLOTS, OPT_RESEED, OPT_PERSIST_BLOCKS, OPT_REFILL_MIN, OPT_CLAIM_CHUNK = 1, 2, 3, 4, 5, 6
OPT_FRAMES_IN_FLIGHT = 7
OPT_SORT_RAYS = 9               # 0 = never, 1 = always, 2 = batches of 2^20 rays or more (default)
OPT_SORT_BITS = 10
OPT_SHADOW_STREAM = 12
OPT_MERGE_PASSES = 13
OPT_TRAIN_MAX_MB = 16            # most device memory (MiB) the frame train may take; 0 = no limit
OPT_TRAIN_KEEP = 17              # 1: the frame train is kept between render calls (frames stay in flight; render_flush finishes them)
OPT_TIME_TRAVERSAL = 18          # 1: the frame train times its traversal launches with HIP events (traversal_ms reads the sums)
OPT_TRAIN_PARTS = 15             # 1..4: bands of rows the frame train renders side by side
OPT_FRAME_TRAIN = 14             # pt_render_samples steps its frames together, one traversal launch per step: 0 never, 1 always, 2 = for numRays >= 3 or scenes of more than 8 MiB of records (default)
OPT_SECOND_WALK = 11            # certified walk's second pass: 0 = a ray per lane, 1 = a ray per wavefront (default)
OPT_RECORD_LAYOUT = 8           # 0 = 64 B records, 1 = 48 B records, 2 = 128 B treelet records beside the 64 B ones
LAYOUT_RECORDS64, LAYOUT_RECORDS48, LAYOUT_TREELETS = 0, 1, 2
TRAVERSAL_REFERENCE_LAYOUT, TRAVERSAL_FLAT, TRAVERSAL_FLAT_ORDERED, TRAVERSAL_CERTIFIED = 0, 1, 2, 3
TRAVERSAL_DEFAULT = TRAVERSAL_CERTIFIED          # what a new context starts with


class PtError(RuntimeError):
    pass


def lib_path():
    """The product library, or the experiment build PTCORE_LIB names (scripts/exp_variant.sh; never set by tests or bench)."""
    return os.environ.get("PTCORE_LIB") or os.path.join(_HERE, "libptcore.so")


_lib = None


def _load():
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own ROCm runtime libraries under the same sonames as /opt/rocm's; whichever set is
    # loaded first serves the whole process.  Loading torch's first keeps one consistent set when torch is
    # used for torch.distributed later (a mixed set aborts at process exit).  PTCORE_NO_TORCH=1 skips this.
    # pt_render_samples keeps three frames in flight on six streams; with HIP's default of four hardware queues per
    # process the streams share queues and wait on each other (measured: 183 against 199 spp/s at 1080p).  The
    # setting is read when the HIP runtime starts, so it only takes effect if nothing has used the GPU yet.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    if os.environ.get("PTCORE_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    path = lib_path()
    if not os.path.exists(path):
        raise PtError(f"{path} is missing: build it with `python pathtracer_amd/build.py` (there is no CPU fallback)")
    L = C.CDLL(path)
    vp, u32, u64, i32, sz, cp = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_size_t, C.c_char_p
    pu32 = C.POINTER(u32)
    sig = {
        "pt_ply_load": (i32, [cp, pu32, pu32, vp, vp, vp, cp, sz]),
        "pt_build_bvh": (i32, [vp, u32, vp, u32, vp, pu32, i32, cp, sz]),
        "pt_flatten_bvh": (i32, [vp, u32, vp, vp, u32, vp, vp, vp, pu32, cp, sz]),
        "pt_flatten_bvh_compact": (i32, [vp, u32, vp, vp, u32, vp, vp, vp, vp, pu32, cp, sz]),
        "pt_flatten_bvh_treelets": (i32, [vp, u32, vp, vp, u32, vp, u64, C.POINTER(u64), vp, pu32, cp, sz]),
        "pt_fill_constants": (None, [vp, u32, u32, vp, C.c_float, C.c_float, C.c_float, u32, u32, u32, u32, u32, vp]),
        "pt_ctx_create": (i32, [i32, C.POINTER(vp)]),
        "pt_ctx_destroy": (None, [vp]),
        "pt_last_error": (cp, [vp]),
        "pt_synchronize": (i32, [vp]),
        "pt_stream": (vp, [vp]),
        "pt_set_option": (i32, [vp, i32, i32]),
        "pt_get_option": (i32, [vp, i32, C.POINTER(i32)]),
        "pt_scene_upload": (i32, [vp, vp, u32, vp, vp, u32, vp, u32, vp, u32, vp]),
        "pt_resize": (i32, [vp, u32, u32, u32]),
        "pt_selftest_entry_bound": (i32, [vp, u64, u64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "pt_set_constants": (i32, [vp, vp]),
        "pt_clear_accumulation": (i32, [vp]),
        "pt_launch": (i32, [vp]),
        "pt_trace": (i32, [vp, u32, i32]),
        "pt_compact": (i32, [vp]),
        "pt_resolve_accumulate": (i32, [vp]),
        "pt_render_sample": (i32, [vp, u32, i32, vp]),
        "pt_render_samples": (i32, [vp, u32, u32, i32, vp]),
        "pt_render_flush": (i32, [vp]),
        "pt_readback_current": (i32, [vp, vp]),
        "pt_readback_accumulation": (i32, [vp, vp]),
        "pt_readback_raystack": (i32, [vp, vp, vp]),
        "pt_upload_raystack": (i32, [vp, vp, vp]),
        "pt_accumulation_device_ptr": (vp, [vp]),
        "pt_dump_current": (i32, [vp, cp, u32]),
        "pt_frame_dump_write": (i32, [cp, u32, u32, u32, vp]),
        "pt_frame_dump_read": (i32, [cp, u32, pu32, pu32, vp, u64]),
        "pt_tonemap": (i32, [vp, vp, vp]),
        "pt_write_ppm": (i32, [cp, u32, u32, vp]),
        "pt_trace_rays": (i32, [vp, vp, u64, i32, vp, vp]),
        "pt_trace_rays_noalloc": (i32, [vp, vp, u64, i32, vp]),
        "pt_trace_rays_host": (i32, [vp, vp, u64, i32, vp, vp]),
        "pt_generate_rays": (i32, [vp, u64, u64, u64, vp]),
        "pt_device_alloc": (vp, [vp, sz]),
        "pt_device_free": (i32, [vp, vp]),
        "pt_copy_to_host": (i32, [vp, vp, vp, sz]),
        "pt_copy_to_device": (i32, [vp, vp, vp, sz]),
        "pt_last_kernel_ms": (C.c_float, [vp]),
        "pt_traversal_ms": (i32, [vp, vp, u32, C.POINTER(C.c_uint32)]),
        "pt_comm_unique_id": (i32, [vp]),
        "pt_comm_init": (i32, [vp, i32, i32, vp]),
        "pt_allreduce_accumulation": (i32, [vp]),
        "pt_version": (cp, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    L._pt_signatures = sig
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


# ---------------------------------------------------------------------------------------------
# host-side scene preparation (runs without a GPU)
# ---------------------------------------------------------------------------------------------
def load_ply(path):
    """HimaliaPlyModel::LoadFile + BuildVertices + BuildIndices (PathTracer.cpp:246-262).
    Returns (positions [V,4] f32, aux [V] AUX_DT, indices [T,3] u32)."""
    L = _load()
    nv, nt = C.c_uint32(0), C.c_uint32(0)
    err = C.create_string_buffer(256)
    if L.pt_ply_load(path.encode(), C.byref(nv), C.byref(nt), None, None, None, err, 256) != 0:
        raise PtError(err.value.decode())
    pos = np.zeros((nv.value, 4), dtype=np.float32)
    aux = np.zeros(nv.value, dtype=AUX_DT)
    idx = np.zeros((nt.value, 3), dtype=np.uint32)
    if L.pt_ply_load(path.encode(), C.byref(nv), C.byref(nt), _p(pos), _p(aux), _p(idx), err, 256) != 0:
        raise PtError(err.value.decode())
    return pos, aux, idx


def build_bvh(positions, indices, num_threads=0):
    """BuildBVH (BVH.cpp:83-309).  Returns (nodes [2T-1] NODE_DT, reordered indices [T,3], max_depth)."""
    L = _load()
    pos = np.ascontiguousarray(positions, dtype=np.float32)
    idx = np.ascontiguousarray(indices, dtype=np.uint32).copy()
    if pos.ndim != 2 or pos.shape[1] != 4 or idx.ndim != 2 or idx.shape[1] != 3:
        raise PtError("build_bvh: positions must be [V,4] and indices [T,3]")
    T = idx.shape[0]
    nodes = np.zeros(max(2 * T - 1, 0), dtype=NODE_DT)
    md = C.c_uint32(0)
    err = C.create_string_buffer(256)
    if L.pt_build_bvh(_p(pos), pos.shape[0], _p(idx), T, _p(nodes), C.byref(md), num_threads, err, 256) != 0:
        raise PtError(err.value.decode())
    return nodes, idx, md.value


def flatten_bvh(nodes, positions, indices):
    """The flattening pass.  Returns (nodes2 [T-1] NODE2_DT, tris [T] TRI_DT, root_box [6], max_stack)."""
    L = _load()
    nodes = np.ascontiguousarray(nodes)
    pos = np.ascontiguousarray(positions, dtype=np.float32)
    idx = np.ascontiguousarray(indices, dtype=np.uint32)
    T = idx.shape[0]
    n2 = np.zeros(max(T - 1, 1), dtype=NODE2_DT)
    tr = np.zeros(T, dtype=TRI_DT)
    rb = np.zeros(6, dtype=np.float32)
    ms = C.c_uint32(0)
    err = C.create_string_buffer(256)
    if L.pt_flatten_bvh(_p(nodes), nodes.shape[0], _p(pos), _p(idx), T, _p(n2), _p(tr), _p(rb), C.byref(ms), err, 256) != 0:
        raise PtError(err.value.decode())
    return n2[:max(T - 1, 0)], tr, rb, ms.value


def flatten_bvh_compact(nodes, positions, indices):
    """The flattening pass with 48 B records.  Returns (nodes2 [T-1] NODE2C_DT, tris [T] TRIC_DT, leaf_ref [T] u32,
    root_box [6], max_stack), or None if some box does not fit the encoding."""
    L = _load()
    nodes = np.ascontiguousarray(nodes)
    pos = np.ascontiguousarray(positions, dtype=np.float32)
    idx = np.ascontiguousarray(indices, dtype=np.uint32)
    T = idx.shape[0]
    n2 = np.zeros(max(T - 1, 1), dtype=NODE2C_DT)
    tr = np.zeros(T, dtype=TRIC_DT)
    lr = np.zeros(T, dtype=np.uint32)
    rb = np.zeros(6, dtype=np.float32)
    ms = C.c_uint32(0)
    err = C.create_string_buffer(256)
    rc = L.pt_flatten_bvh_compact(_p(nodes), nodes.shape[0], _p(pos), _p(idx), T, _p(n2), _p(tr), _p(lr), _p(rb), C.byref(ms), err, 256)
    if rc == 1:
        return None
    if rc != 0:
        raise PtError(err.value.decode())
    return n2[:max(T - 1, 0)], tr, lr, rb, ms.value


def flatten_bvh_treelets(nodes, positions, indices):
    """The flattening pass with 128 B treelet records.  Returns (units [n, 16] u32: the depth-first stream of 64 B
    units holding TREELET_DT records (two units each) and TRI_DT leaves, root_box [6], max_stack), or None if some
    box does not fit the encoding."""
    L = _load()
    nodes = np.ascontiguousarray(nodes)
    pos = np.ascontiguousarray(positions, dtype=np.float32)
    idx = np.ascontiguousarray(indices, dtype=np.uint32)
    T = idx.shape[0]
    rb = np.zeros(6, dtype=np.float32)
    ms, nu = C.c_uint32(0), C.c_uint64(0)
    err = C.create_string_buffer(256)
    rc = L.pt_flatten_bvh_treelets(_p(nodes), nodes.shape[0], _p(pos), _p(idx), T, None, 0, C.byref(nu), _p(rb), C.byref(ms), err, 256)
    if rc < 0:
        raise PtError(err.value.decode())
    units = np.zeros((nu.value, 16), dtype=np.uint32)
    rc = L.pt_flatten_bvh_treelets(_p(nodes), nodes.shape[0], _p(pos), _p(idx), T, _p(units), nu.value, C.byref(nu), _p(rb), C.byref(ms), err, 256)
    if rc == 1:
        return None
    if rc != 0:
        raise PtError(err.value.decode())
    return units, rb, ms.value


def fill_constants(width, height, num_rays, num_triangles, num_bvh_nodes, frame_index=1, focus_center=(0.0, 0.0, 0.0),
                   orbit_height=0.5, orbit_radius=3.0, orbit_angle=0.0, num_lights=1, ambient=(0.4, 0.5, 0.7)):
    """The constants block of PathTracer.cpp:731-750 with the reference's defaults (PathTracer.cpp:96-114)."""
    L = _load()
    c = np.zeros(1, dtype=CONST_DT)
    fc = np.asarray(focus_center, dtype=np.float32)
    am = np.asarray(ambient, dtype=np.float32)
    L.pt_fill_constants(_p(c), width, height, _p(fc), orbit_height, orbit_radius, orbit_angle, num_lights, num_triangles,
                        frame_index, num_rays, num_bvh_nodes, _p(am))
    return c


def default_lights():
    """The single point light of PathTracer.cpp:41-43."""
    l = np.zeros(1, dtype=LIGHT_DT)
    l["pos"][0] = (0.0, 1.4, 0.0)
    l["radiance"][0] = (1.0, 1.0, 1.0)
    return l


def load_blue_noise():
    """The 65536-entry table the reference compiles in (blueNoise.h), shipped as little-endian u16."""
    path = os.path.join(_HERE, "data", "blue_noise_u16le.bin")
    bn = np.fromfile(path, dtype="<u2")
    if bn.shape[0] != 65536:
        raise PtError(f"{path}: expected 65536 entries")
    return bn


# ---------------------------------------------------------------------------------------------
# device context
# ---------------------------------------------------------------------------------------------
class DeviceBuffer:
    def __init__(self, pt, nbytes):
        self.pt, self.nbytes = pt, nbytes
        self.ptr = pt._L.pt_device_alloc(pt._h, nbytes)
        if not self.ptr:
            raise PtError(pt._err())

    def free(self):
        if self.ptr:
            self.pt._L.pt_device_free(self.pt._h, self.ptr)
            self.ptr = None

    def to_host(self, dtype, count=None):
        dtype = np.dtype(dtype)
        count = self.nbytes // dtype.itemsize if count is None else count
        out = np.zeros(count, dtype=dtype)
        self.pt._chk(self.pt._L.pt_copy_to_host(self.pt._h, _p(out), self.ptr, count * dtype.itemsize))
        return out

    def from_host(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.pt._chk(self.pt._L.pt_copy_to_device(self.pt._h, self.ptr, _p(arr), arr.nbytes))


class PathTracer:
    """One device context (one GPU, one in-order stream)."""

    def __init__(self, device_id=0):
        self._L = _load()
        h = C.c_void_p()
        rc = self._L.pt_ctx_create(device_id, C.byref(h))
        if rc != 0:
            raise PtError(f"pt_ctx_create failed ({rc}): {self._L.pt_last_error(None).decode()}")
        self._h = h
        self.device_id = device_id
        self.W = self.H = self.slots = 0
        self.num_triangles = self.num_nodes = 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.pt_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _err(self):
        return self._L.pt_last_error(self._h).decode()

    def _chk(self, rc):
        if rc != 0:
            raise PtError(f"ptcore error {rc}: {self._err()}")

    def set_option(self, opt, value):
        self._chk(self._L.pt_set_option(self._h, opt, int(value)))

    def get_option(self, opt):
        v = C.c_int(0)
        self._chk(self._L.pt_get_option(self._h, opt, C.byref(v)))
        return v.value

    def synchronize(self):
        self._chk(self._L.pt_synchronize(self._h))

    # scene + frame resources ------------------------------------------------------------------
    def upload_scene(self, positions, aux, indices, nodes, lights=None, blue_noise=None):
        pos = np.ascontiguousarray(positions, dtype=np.float32)
        idx = np.ascontiguousarray(indices, dtype=np.uint32)
        nodes = np.ascontiguousarray(nodes)
        aux = np.ascontiguousarray(aux) if aux is not None else None
        lights = np.ascontiguousarray(lights) if lights is not None else None
        bn = np.ascontiguousarray(blue_noise, dtype=np.uint16) if blue_noise is not None else None
        self._chk(self._L.pt_scene_upload(self._h, _p(pos), pos.shape[0], _p(aux), _p(idx), idx.shape[0], _p(nodes), nodes.shape[0],
                                          _p(lights), 0 if lights is None else lights.shape[0], _p(bn)))
        self.num_triangles, self.num_nodes = idx.shape[0], nodes.shape[0]

    def resize(self, width, height, max_slots):
        self._chk(self._L.pt_resize(self._h, width, height, max_slots))
        self.W, self.H, self.slots = width, height, max_slots

    def set_constants(self, consts):
        consts = np.ascontiguousarray(consts)
        assert consts.dtype == CONST_DT and consts.size == 1
        self._chk(self._L.pt_set_constants(self._h, _p(consts)))

    def clear_accumulation(self):
        self._chk(self._L.pt_clear_accumulation(self._h))

    # dispatch sites ---------------------------------------------------------------------------
    def launch(self):
        self._chk(self._L.pt_launch(self._h))

    def trace(self, depth, speculative=False):
        self._chk(self._L.pt_trace(self._h, depth, int(speculative)))

    def compact(self):
        self._chk(self._L.pt_compact(self._h))

    def resolve_accumulate(self):
        self._chk(self._L.pt_resolve_accumulate(self._h))

    def render_sample(self, frame_index, speculative_depth0=False, want_ray_counts=False):
        rc = np.zeros(2 * max(self.slots, 1), dtype=np.uint64) if want_ray_counts else None
        self._chk(self._L.pt_render_sample(self._h, frame_index, int(speculative_depth0), _p(rc)))
        return rc

    def render_samples(self, first_frame_index, num_frames, speculative_depth0=False, want_ray_counts=False):
        """num_frames consecutive frames, several in flight; same results as that many render_sample calls."""
        rc = np.zeros(2 * max(self.slots, 1), dtype=np.uint64) if want_ray_counts else None
        self._chk(self._L.pt_render_samples(self._h, first_frame_index, num_frames, int(speculative_depth0), _p(rc)))
        return rc

    def render_flush(self):
        """Finishes the frames a kept frame train still holds (OPT_TRAIN_KEEP = 1); nothing to do otherwise."""
        self._chk(self._L.pt_render_flush(self._h))

    # read-back --------------------------------------------------------------------------------
    def readback_accumulation(self):
        out = np.zeros((self.H, self.W, 4), dtype=np.float32)
        self._chk(self._L.pt_readback_accumulation(self._h, _p(out)))
        return out

    def upload_accumulation(self, acc):
        """Overwrites the RGBA32F accumulation with a host image (H, W, 4) float32."""
        acc = np.ascontiguousarray(acc, dtype=np.float32)
        assert acc.shape == (self.H, self.W, 4)
        self._chk(self._L.pt_copy_to_device(self._h, self.accumulation_device_ptr(), _p(acc), acc.nbytes))

    def readback_current(self):
        out = np.zeros((self.H, self.W, 4), dtype=np.uint16)
        self._chk(self._L.pt_readback_current(self._h, _p(out)))
        return out

    def readback_raystack(self):
        slots = np.zeros(self.W * self.H * self.slots, dtype=SLOT_DT)
        jobs = np.zeros(self.W * self.H, dtype=np.uint32)
        self._chk(self._L.pt_readback_raystack(self._h, _p(slots), _p(jobs)))
        return slots, jobs

    def upload_raystack(self, slots=None, jobs=None):
        slots = np.ascontiguousarray(slots) if slots is not None else None
        jobs = np.ascontiguousarray(jobs, dtype=np.uint32) if jobs is not None else None
        self._chk(self._L.pt_upload_raystack(self._h, _p(slots), _p(jobs)))

    def accumulation_device_ptr(self):
        return self._L.pt_accumulation_device_ptr(self._h)

    def dump_current(self, directory, index):
        """Writes RayStackBuffer_<index>.bin.z (+ .txt) under directory (PathTracer.cpp:180-202, 840-861)."""
        self._chk(self._L.pt_dump_current(self._h, os.fsencode(directory), index))

    def tonemap(self, want_float=False):
        """Display colour of composite.frag:57 as (H, W, 4) uint8, or (H, W, 3) float32 with want_float."""
        if want_float:
            out = np.zeros((self.H, self.W, 3), dtype=np.float32)
            self._chk(self._L.pt_tonemap(self._h, _p(out), None))
        else:
            out = np.zeros((self.H, self.W, 4), dtype=np.uint8)
            self._chk(self._L.pt_tonemap(self._h, None, _p(out)))
        return out

    # traversal microbench ---------------------------------------------------------------------
    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def generate_rays(self, d_rays, n, seed=20261111, first=0):
        self._chk(self._L.pt_generate_rays(self._h, seed, first, n, d_rays.ptr))

    def trace_rays_device(self, d_rays, n, d_hits, any_hit=False, want_stats=False):
        st = np.zeros(4, dtype=np.uint64) if want_stats else None
        self._chk(self._L.pt_trace_rays(self._h, d_rays.ptr, n, int(any_hit), d_hits.ptr, _p(st)))
        return st

    def trace_rays_device_noalloc(self, d_rays, n, d_hits, any_hit=False):
        """pt_trace_rays_noalloc: the same batch through the static scratch array, nothing allocated."""
        self._chk(self._L.pt_trace_rays_noalloc(self._h, d_rays.ptr, n, int(any_hit), d_hits.ptr))

    def trace_rays(self, rays, any_hit=False, want_stats=False):
        rays = np.ascontiguousarray(rays)
        assert rays.dtype == RAY_DT
        hits = np.zeros(rays.shape[0], dtype=HIT_DT)
        st = np.zeros(4, dtype=np.uint64) if want_stats else None
        self._chk(self._L.pt_trace_rays_host(self._h, _p(rays), rays.shape[0], int(any_hit), _p(hits), _p(st)))
        return (hits, st) if want_stats else hits

    def selftest_entry_bound(self, num_cases, seed=1):
        """(cases in which the ray reaches the box, violations of bound <= true entry distance); see ptcore.h."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._chk(self._L.pt_selftest_entry_bound(self._h, seed, num_cases, C.byref(a), C.byref(b)))
        return a.value, b.value

    def traversal_ms(self):
        """Durations (ms, float32 array, launch order) of the frame train's traversal launches since the last reading
        (OPT_TIME_TRAVERSAL = 1); the record starts over."""
        out = np.zeros(4096, dtype=np.float32)
        n = C.c_uint32(0)
        self._chk(self._L.pt_traversal_ms(self._h, out.ctypes.data_as(C.c_void_p), out.size, C.byref(n)))
        return out[:min(int(n.value), out.size)].copy()

    def last_kernel_ms(self):
        return float(self._L.pt_last_kernel_ms(self._h))

    # multi-GPU --------------------------------------------------------------------------------
    def comm_init(self, rank, num_ranks, unique_id):
        uid = np.ascontiguousarray(unique_id, dtype=np.uint8)
        assert uid.shape[0] == 128
        self._chk(self._L.pt_comm_init(self._h, rank, num_ranks, _p(uid)))

    def allreduce_accumulation(self):
        self._chk(self._L.pt_allreduce_accumulation(self._h))


def comm_unique_id():
    L = _load()
    uid = np.zeros(128, dtype=np.uint8)
    if L.pt_comm_unique_id(_p(uid)) != 0:
        raise PtError(L.pt_last_error(None).decode())
    return uid


_HOST_CODES = {-1: "bad argument", -4: "file I/O error", -5: "zlib error", -6: "malformed size file or stream length",
               -7: "destination too small"}


def write_frame_dump(directory, index, rgba16f):
    """Host half of the frame dump: rgba16f is (H, W, 4) uint16 (fp16 bit patterns) or float16."""
    a = np.ascontiguousarray(rgba16f)
    if a.dtype == np.float16:
        a = a.view(np.uint16)
    if a.dtype != np.uint16 or a.ndim != 3 or a.shape[2] != 4:
        raise ValueError("rgba16f must be (H, W, 4) uint16 or float16")
    rc = _load().pt_frame_dump_write(os.fsencode(directory), index, a.shape[1], a.shape[0], _p(a))
    if rc:
        raise PtError(f"pt_frame_dump_write: {_HOST_CODES.get(rc, rc)}")


def read_frame_dump(directory, index):
    """Reads a dump pair (this library's or the reference's) back as (H, W, 4) uint16."""
    L = _load()
    w, h = C.c_uint32(0), C.c_uint32(0)
    rc = L.pt_frame_dump_read(os.fsencode(directory), index, C.byref(w), C.byref(h), None, 0)
    if rc:
        raise PtError(f"pt_frame_dump_read: {_HOST_CODES.get(rc, rc)}")
    out = np.zeros((h.value, w.value, 4), dtype=np.uint16)
    rc = L.pt_frame_dump_read(os.fsencode(directory), index, C.byref(w), C.byref(h), _p(out), out.size // 4)
    if rc:
        raise PtError(f"pt_frame_dump_read: {_HOST_CODES.get(rc, rc)}")
    return out


def write_ppm(path, rgba8):
    a = np.ascontiguousarray(rgba8, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] != 4:
        raise ValueError("rgba8 must be (H, W, 4) uint8")
    rc = _load().pt_write_ppm(os.fsencode(path), a.shape[1], a.shape[0], _p(a))
    if rc:
        raise PtError(f"pt_write_ppm: {_HOST_CODES.get(rc, rc)}")
