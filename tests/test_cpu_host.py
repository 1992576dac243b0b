"""CPU-side checks (no GPU): the C ABI loads and exports what include/ptcore.h declares, the PLY
reader, the parallel BVH builder against the oracle's line-by-line restatement of BuildBVH, the
flattening pass, and the constants block."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import plyio
import pathtracer_amd as pt
from pathtracer_amd import core

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODELS = os.path.join(ROOT, "tests", "golden", "models")


def test_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "ptcore.h")).read()
    declared = set(re.findall(r"PT_API\s+[\w\s\*]+?\b(pt_\w+)\s*\(", hdr))
    assert len(declared) >= 30
    lib = C.CDLL(pt.lib_path())
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} is declared in ptcore.h but not exported"
    assert declared == set(core._load()._pt_signatures), "python binding and header disagree"
    assert b"gfx950" in core._load().pt_version()


def test_option_numbers_of_the_binding_are_the_header_s():
    hdr = open(os.path.join(ROOT, "include", "ptcore.h")).read()
    in_header = {name: int(val) for name, val in re.findall(r"^\s+PT_OPT_(\w+)\s*=\s*(\d+)", hdr, re.M)}      # the enum's own lines
    assert len(in_header) >= 17 and len(set(in_header.values())) == len(in_header)
    in_binding = {k[4:]: v for k, v in vars(core).items() if k.startswith("OPT_") and isinstance(v, int)}
    assert in_binding == in_header


def test_context_needs_a_gpu_and_says_so():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pt.PtError) as e:
        pt.PathTracer(0)
    assert "no CPU path" in str(e.value) or "HIP" in str(e.value)


def test_struct_layouts_match_reference_contract():
    # sizes and offsets of the reference's data contract (ShaderData.h:9-47, structures.glsl:35-72)
    assert core.SLOT_DT.fields["randState"][1] == 12 and core.SLOT_DT.fields["rayOrigin"][1] == 16
    assert core.SLOT_DT.fields["currentDepth"][1] == 28 and core.SLOT_DT.fields["hitAlbedo"][1] == 32
    assert core.SLOT_DT.fields["wIn"][1] == 40 and core.SLOT_DT.fields["prob"][1] == 46
    assert core.CONST_DT.fields["viewportSize"][1] == 256 and core.CONST_DT.fields["numLights"][1] == 272
    assert core.CONST_DT.fields["numBVHNodes"][1] == 288 and core.CONST_DT.fields["ambientRadiance"][1] == 304
    assert core.NODE_DT.fields["next"][1] == 12 and core.NODE_DT.fields["right"][1] == 28


@pytest.mark.parametrize("name,nv,nt", [("CBbunny.ply", 14454, 28588), ("CBmonkey.ply", 11828, 5818)])
def test_ply_reader_matches_numpy_reader(name, nv, nt):
    pos, aux, idx = pt.load_ply(os.path.join(MODELS, name))
    ref = plyio.read_ply(os.path.join(MODELS, name))
    assert pos.shape == (nv, 4) and idx.shape == (nt, 3)
    assert np.array_equal(pos, ref["positions"]) and np.array_equal(idx, ref["indices"])
    assert np.array_equal(aux["normal"], ref["normals"]) and np.array_equal(aux["color"], ref["colors"])
    assert (pos[:, 3] == 1).all()


def test_ply_reader_ascii_and_errors(tmp_path):
    p = tmp_path / "quad.ply"
    p.write_text("ply\nformat ascii 1.0\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\n"
                 "element face 1\nproperty list uchar uint vertex_indices\nend_header\n"
                 "0 0 0\n1 0 0\n1 1 0\n0 1 0\n4 0 1 2 3\n")
    pos, aux, idx = pt.load_ply(str(p))
    assert idx.tolist() == [[0, 1, 2], [0, 2, 3]]          # fan triangulation
    assert (aux["color"] == 255).all()
    stub = tmp_path / "stub.ply"
    stub.write_text("version https://git-lfs.github.com/spec/v1\noid sha256:00\nsize 1\n")
    with pytest.raises(pt.PtError):
        pt.load_ply(str(stub))
    with pytest.raises(pt.PtError):
        pt.load_ply(str(tmp_path / "missing.ply"))
    bad = tmp_path / "bad.ply"
    bad.write_text("ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nproperty float y\nproperty float z\n"
                   "element face 1\nproperty list uchar uint vertex_indices\nend_header\n0 0 0\n3 0 1 2\n")
    with pytest.raises(pt.PtError):
        pt.load_ply(str(bad))


def test_ply_reader_refuses_hostile_files(tmp_path):
    """Files a fuzzer made the reader trip over once: each must come back as an error (or load), never as a crash, a
    hang or a buffer written past its end."""
    head = ("ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\n"
            "element face 1\nproperty list uchar int vertex_indices\nend_header\n0 0 0\n1 0 0\n0 1 0\n")
    cases = {
        "negative_list_length": head + "-3 0 1 2\n",
        "huge_list_length": head + "99999999999 0 1 2\n",
        "negative_index": head + "3 0 1 -2\n",
        "index_past_uint32": head + "3 0 1 99999999999\n",
        "not_a_number": head + "3 a b c\n",
        "short_polygon_then_end": head + "2 0 1\n",
        "second_vertex_element": head.replace("element face 1", "element vertex 1\nproperty float x\nproperty float y\n"
                                              "property float z\nelement face 1") + "5 5 5\n3 0 0 0\n",
        "vertex_count_past_uint32": head.replace("element vertex 3", "element vertex 99999999999"),
        "binary_truncated": None,
    }
    for name, text in cases.items():
        f = tmp_path / (name + ".ply")
        if text is None:
            src = open(os.path.join(os.path.dirname(__file__), "golden", "models", "CBbunny.ply"), "rb").read()
            f.write_bytes(src[: len(src) // 2])
        else:
            f.write_text(text)
        if name == "short_polygon_then_end":          # a two-corner polygon makes no triangle: a valid, empty model
            pos, aux, idx = pt.load_ply(str(f))
            assert idx.shape[0] == 0 and pos.shape[0] == 3
            continue
        with pytest.raises(pt.PtError):
            pt.load_ply(str(f))
    # elements without properties carry no data, whatever their count says: must not loop 2^64 times
    f = tmp_path / "empty_elements.ply"
    f.write_text("ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\n"
                 "element foo 18446744073709551615\nelement face 18446744073709551615\nend_header\n0 0 0\n1 0 0\n0 1 0\n")
    pos, aux, idx = pt.load_ply(str(f))
    assert pos.shape[0] == 3 and idx.shape[0] == 0
    # colours out of range saturate instead of wrapping
    f = tmp_path / "colours.ply"
    f.write_text("ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nproperty float y\nproperty float z\n"
                 "property int red\nproperty int green\nproperty int blue\nend_header\n0 0 0 300 -5 17\n")
    pos, aux, idx = pt.load_ply(str(f))
    assert aux["color"][0].tolist() == [255, 0, 17, 255]


def _check_bvh_invariants(nodes, ridx, pos):
    T = ridx.shape[0]
    N = nodes.shape[0]
    assert N == 2 * T - 1
    right, nxt = nodes["right"], nodes["next"]
    leaves = right <= 0
    assert leaves.sum() == T
    # every triangle in exactly one leaf
    assert sorted((-right[leaves] // 3).tolist()) == list(range(T))
    assert nxt[0] == 0
    # pre-order + skip links: walking with every box missed visits root, then ends
    # children inside parents, leaf boxes contain their triangle (with the 6e-5 pad)
    stack = [(0, 0)]
    seen = 0
    while stack:
        i, nx = stack.pop()
        seen += 1
        assert nxt[i] == nx
        if right[i] > 0:
            l, r = i + 1, right[i]
            for c in (l, r):
                assert (nodes["a"][c] >= nodes["a"][i] - 1e-7).all() and (nodes["b"][c] <= nodes["b"][i] + 1e-7).all()
            stack.append((r, nx)); stack.append((l, r))
        else:
            tri = pos[ridx[-right[i] // 3], :3]
            assert (tri.min(0) - 6.1e-5 <= nodes["a"][i]).all() and (tri.min(0) >= nodes["a"][i]).all()
            assert (tri.max(0) <= nodes["b"][i]).all()
    assert seen == N


@pytest.mark.parametrize("fixture", ["bunny", "monkey", "courtyard_small"])
def test_bvh_builder_bit_identical_to_oracle(request, oracle_lib, fixture):
    m = request.getfixturevalue(fixture)
    n_o, i_o, d_o = oracle_lib.build_bvh(m["positions"], m["indices_in"])
    assert n_o.tobytes() == m["nodes"].tobytes()
    assert np.array_equal(i_o, m["indices"]) and d_o == m["depth"]
    for threads in (1, 3):
        n_p, i_p, _ = pt.build_bvh(m["positions"], m["indices_in"], threads)
        assert n_p.tobytes() == n_o.tobytes() and np.array_equal(i_p, i_o)
    _check_bvh_invariants(m["nodes"], m["indices"], m["positions"])
    # the same set of triangles, only reordered
    assert sorted(map(tuple, m["indices"].tolist())) == sorted(map(tuple, m["indices_in"].tolist()))


def test_bvh_builder_small_and_degenerate_inputs(oracle_lib):
    pos, aux, idx = pt.scenes.cornell_box()
    n_o, i_o, _ = oracle_lib.build_bvh(pos, idx)
    n_p, i_p, _ = pt.build_bvh(pos, idx)
    assert n_p.tobytes() == n_o.tobytes() and np.array_equal(i_p, i_o)
    # one triangle: a single leaf; coincident centroids: the median fallback (BVH.cpp:269-275)
    one_p = np.array([[0, 0, 0, 1], [1, 0, 0, 1], [0, 1, 0, 1]], dtype=np.float32)
    n1, _, _ = pt.build_bvh(one_p, np.array([[0, 1, 2]], dtype=np.uint32))
    assert n1.shape[0] == 1 and n1["right"][0] == 0 and n1["next"][0] == 0
    same = np.tile(np.array([[0, 1, 2]], dtype=np.uint32), (9, 1))
    n_o, i_o, _ = oracle_lib.build_bvh(one_p, same)
    n_p, i_p, _ = pt.build_bvh(one_p, same)
    assert n_p.tobytes() == n_o.tobytes() and n_p.shape[0] == 17
    with pytest.raises(pt.PtError):
        pt.build_bvh(one_p, np.array([[0, 1, 7]], dtype=np.uint32))
    with pytest.raises(pt.PtError):
        pt.build_bvh(one_p, np.zeros((0, 3), dtype=np.uint32))


def test_flatten_pass(bunny):
    n2, tr, rb, ms = pt.flatten_bvh(bunny["nodes"], bunny["positions"], bunny["indices"])
    nodes, ridx, pos = bunny["nodes"], bunny["indices"], bunny["positions"]
    T = ridx.shape[0]
    assert n2.shape[0] == T - 1 and tr.shape[0] == T and ms == bunny["depth"]
    assert np.array_equal(rb[:3], nodes["a"][0]) and np.array_equal(rb[3:], nodes["b"][0])
    interior = np.nonzero(nodes["right"] > 0)[0]
    assert np.array_equal(n2["refIndex"], interior)
    # boxes are the reference's, pre-digested: c = (a+b)*0.5f, h = f16(a-c), f16(b-c)
    for k, (side, child) in enumerate((("L", interior + 1), ("R", nodes["right"][interior]))):
        a, b = nodes["a"][child], nodes["b"][child]
        c = ((a + b) * np.float32(0.5)).astype(np.float32)
        assert np.array_equal(n2["c"][:, :, k], c)
        h = np.concatenate([(a - c).astype(np.float16), (b - c).astype(np.float16)], axis=1).view(np.uint16)
        assert np.array_equal(n2["h"][:, :, k], h)
    # triangles inline, edges as fp32 differences
    assert np.array_equal(tr["i1"], ridx[:, 0].astype(np.int32))
    assert np.array_equal(tr["p1"], pos[ridx[:, 0], :3])
    assert np.array_equal(tr["e1"], pos[ridx[:, 1], :3] - pos[ridx[:, 0], :3])
    assert np.array_equal(tr["e2"], pos[ridx[:, 2], :3] - pos[ridx[:, 0], :3])
    leaf_nodes = np.nonzero(nodes["right"] <= 0)[0]
    assert np.array_equal(np.sort(tr["refLeaf"]), leaf_nodes)
    assert tr["flags"].sum() == 1 and tr["refLeaf"][tr["flags"] == 1][0] == nodes.shape[0] - 1
    # child references resolve to the same subtrees
    lref = n2["left"]
    is_leaf = (lref & 0x80000000) != 0
    assert np.array_equal(is_leaf, nodes["right"][interior + 1] <= 0)
    with pytest.raises(pt.PtError):
        pt.flatten_bvh(nodes[:-1], pos, ridx)


def test_flatten_pass_half_extent_bound(bunny):
    """PtNode2.wmax, the per-box constant of the certified walk's entry bound: an fp16 value not below the largest magnitude
    among the box's six digested half extents plus 2^-12, and the next fp16 value down is below it (rounded up, no further);
    +inf for a child whose subtree holds a needle triangle (none in CBbunny; a sliver added to a small soup shows it)."""
    n2, _, _, _ = pt.flatten_bvh(bunny["nodes"], bunny["positions"], bunny["indices"])
    h = (n2["h"] & 0x7FFF).view(np.float16).astype(np.float64)           # (records, 6, 2)
    need = h.max(axis=1) + 2.0 ** -12
    w = n2["wmax"].view(np.float16).astype(np.float64)
    assert np.isfinite(w).all() and (w >= need).all()
    below = (n2["wmax"] - 1).view(np.float16).astype(np.float64)
    assert (below < need).all()
    # a needle: two edges at an angle of 1e-6 make every box above it one that is never skipped on distance
    pos = np.ones((9, 4), np.float32)
    pos[:6, :3] = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [2, 0, 0], [3, 0, 0], [2, 1, 0]], np.float32)
    pos[6:, :3] = np.array([[5, 0, 0], [6, 0, 0], [7, 1e-6, 0]], np.float32)
    idx = np.arange(9, dtype=np.uint32).reshape(3, 3)
    nodes, ridx, _ = pt.build_bvh(pos, idx, 1)
    m2, tr, _, _ = pt.flatten_bvh(nodes, pos, ridx)
    thin_leaf = int(np.nonzero(tr["i1"] == 6)[0][0]) | 0x80000000
    marked = m2["wmax"] == 0x7C00
    assert marked.any()
    # the marks are exactly the children whose subtree contains the needle's leaf: its own reference and every ancestor's
    holds = {}
    def subtree_has(ref):
        if ref & 0x80000000:
            return ref == thin_leaf
        if ref not in holds:
            holds[ref] = subtree_has(int(m2["left"][ref])) or subtree_has(int(m2["right"][ref]))
        return holds[ref]
    for r in range(m2.shape[0]):
        assert bool(marked[r, 0]) == subtree_has(int(m2["left"][r])) and bool(marked[r, 1]) == subtree_has(int(m2["right"][r]))


def test_flatten_pass_refuses_a_malformed_tree(bunny):
    """A node array that is not the builder's pre-order tree would send a walk round in circles on the device (or the
    flattening pass itself): a right child that points back up, a skip link into the middle of a subtree, a leaf
    listed twice.  Each is refused with a reason; the intact array passes."""
    nodes, pos, idx = bunny["nodes"], bunny["positions"], bunny["indices"]
    pt.flatten_bvh(nodes, pos, idx)
    interior = np.nonzero(nodes["right"] > 0)[0]
    leaves = np.nonzero(nodes["right"] <= 0)[0]
    deep = interior[len(interior) // 2]

    def broken(edit):
        bad = nodes.copy()
        edit(bad)
        for fn in (pt.flatten_bvh, pt.flatten_bvh_compact):
            with pytest.raises(pt.PtError):
                fn(bad, pos, idx)

    broken(lambda b: b["right"].__setitem__(deep, 1))                             # right child = an ancestor's left child
    broken(lambda b: b["right"].__setitem__(deep, int(deep) + 1))                 # right child = own left child
    broken(lambda b: b["next"].__setitem__(deep + 1, int(b["next"][deep + 1]) + 1))   # skip link past the subtree's end
    broken(lambda b: b["next"].__setitem__(0, 5))                                 # the root's skip link must end the walk
    broken(lambda b: b["right"].__setitem__(leaves[3], int(b["right"][leaves[7]])))   # one triangle in two leaves
    broken(lambda b: b["right"].__setitem__(leaves[3], -1))                       # leaf offset not a multiple of 3


def test_fill_constants():
    c = pt.fill_constants(1920, 1080, 5, 100, 199, frame_index=7)
    assert c["numRays"][0] == 5 and c["frameIndex"][0] == 7 and c["numBVHNodes"][0] == 199
    assert tuple(c["viewportSize"][0]) == (1920.0, 1080.0) and tuple(c["viewportBase"][0]) == (0.0, 0.0)
    v, p = c["viewMtx"][0].reshape(4, 4).T, c["projMtx"][0].reshape(4, 4).T       # column-major -> row-major
    vi, pi = c["viewInvMtx"][0].reshape(4, 4).T, c["projInvMtx"][0].reshape(4, 4).T
    assert np.allclose(v @ vi, np.eye(4), atol=1e-5) and np.allclose(p @ pi, np.eye(4), atol=1e-4)
    eye = vi[:3, 3]
    assert np.allclose(eye, (3.0, 0.5, 0.0), atol=1e-6)                            # PathTracer.cpp:109-112,734
    assert p[1, 1] < 0                                                             # the y flip of PathTracer.cpp:737
    assert np.isclose(-p[1, 1], 1 / np.tan(np.radians(30)), rtol=1e-6)
    assert np.allclose(c["ambientRadiance"][0], (0.4, 0.5, 0.7))


def test_blue_noise_table_is_the_reference_table():
    import hashlib
    bn = pt.load_blue_noise()
    assert hashlib.sha256(bn.astype("<u2").tobytes()).hexdigest() == "4762b68dd49315e4c17cde4abc0f143cde425ed3a361182e71191239e94685b8"
    assert sorted(bn.tolist()) == list(range(65536))


def test_procedural_scenes_are_deterministic_and_clean():
    for gen, kw in ((pt.scenes.cornell_box, {}), (pt.scenes.courtyard, dict(num_triangles=20000, seed=5))):
        p1, a1, i1 = gen(**kw)
        p2, a2, i2 = gen(**kw)
        assert np.array_equal(p1, p2) and np.array_equal(i1, i2) and a1.tobytes() == a2.tobytes()
        tri = p1[i1][:, :, :3]
        area = np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
        assert (area > 0).all()
        assert np.allclose(np.linalg.norm(a1["normal"], axis=1), 1, atol=1e-5)
    p, a, i = pt.scenes.courtyard(50000)
    assert abs(i.shape[0] - 50000) < 1500


# ---- frame-dump interchange (PathTracer.cpp:180-202) ----------------------------------------------

def test_frame_dump_pair_round_trip_and_python_zlib(tmp_path):
    """The writer produces the reference's file pair: a zlib stream of W*H RGBA16F texels that python's zlib
    inflates to the same bytes, and "W,H" in the .txt; the reader accepts a pair made without the library
    (what a miniz-written dump looks like to it)."""
    import zlib
    rng = np.random.default_rng(5)
    W, H = 37, 19
    img = rng.standard_normal((H, W, 4)).astype(np.float16)
    img[..., 3] = 1.0
    pt.write_frame_dump(tmp_path, 7, img)
    z = (tmp_path / "RayStackBuffer_7.bin.z").read_bytes()
    assert zlib.decompress(z) == img.tobytes()
    assert (tmp_path / "RayStackBuffer_7.bin.z.txt").read_text() == f"{W},{H}\n"
    back = pt.read_frame_dump(tmp_path, 7)
    assert back.shape == (H, W, 4) and back.tobytes() == img.tobytes()

    # a pair written by another zlib implementation / level, CRLF in the size file
    img2 = rng.standard_normal((H, W, 4)).astype(np.float16)
    (tmp_path / "RayStackBuffer_8.bin.z").write_bytes(zlib.compress(img2.tobytes(), 1))
    (tmp_path / "RayStackBuffer_8.bin.z.txt").write_text(f"{W},{H}\r\n")
    assert pt.read_frame_dump(tmp_path, 8).tobytes() == img2.tobytes()


def test_frame_dump_errors(tmp_path):
    import zlib
    with pytest.raises(pt.PtError, match="file I/O"):
        pt.read_frame_dump(tmp_path, 1)
    (tmp_path / "RayStackBuffer_2.bin.z.txt").write_text("4,4\n")
    (tmp_path / "RayStackBuffer_2.bin.z").write_bytes(b"not a zlib stream")
    with pytest.raises(pt.PtError, match="zlib"):
        pt.read_frame_dump(tmp_path, 2)
    (tmp_path / "RayStackBuffer_3.bin.z.txt").write_text("4,4\n")
    (tmp_path / "RayStackBuffer_3.bin.z").write_bytes(zlib.compress(b"\0" * 64))      # 8 texels, size file says 16
    with pytest.raises(pt.PtError, match="malformed"):
        pt.read_frame_dump(tmp_path, 3)
    (tmp_path / "RayStackBuffer_4.bin.z.txt").write_text("banana\n")
    with pytest.raises(pt.PtError, match="malformed"):
        pt.read_frame_dump(tmp_path, 4)
    with pytest.raises(pt.PtError, match="file I/O"):
        pt.write_frame_dump(tmp_path / "no_such_dir", 1, np.zeros((2, 2, 4), np.uint16))


def test_write_ppm(tmp_path):
    img = (np.arange(5 * 3 * 4) % 251).astype(np.uint8).reshape(3, 5, 4)
    pt.write_ppm(tmp_path / "a.ppm", img)
    raw = (tmp_path / "a.ppm").read_bytes()
    assert raw.startswith(b"P6\n5 3\n255\n")
    assert raw[len(b"P6\n5 3\n255\n"):] == img[..., :3].tobytes()


@pytest.mark.parametrize("fixture", ["bunny", "monkey", "courtyard_small"])
def test_flatten_pass_compact_records_decode_to_the_wide_ones(request, fixture):
    """The 48 B records hold what the 64 B records hold: f16(b-c) rebuilt from f16(a-c) and the stored difference the
    way the kernel does it equals the stored halves of the 64 B record, and every other field is the same."""
    m = request.getfixturevalue(fixture)
    n2, tr, rb, ms = pt.flatten_bvh(m["nodes"], m["positions"], m["indices"])
    out = pt.flatten_bvh_compact(m["nodes"], m["positions"], m["indices"])
    assert out is not None, "a box of this model does not fit the 48 B encoding"
    c2, ct, leaf_ref, rb2, ms2 = out
    assert ms2 == ms and np.array_equal(rb, rb2)
    assert np.array_equal(c2["c"], n2["c"]) and np.array_equal(c2["left"], n2["left"]) and np.array_equal(c2["right"], n2["right"])
    assert np.array_equal(c2["hA"], n2["h"][:, :3, :])
    a_pair = c2["hA"].astype(np.uint32)
    a_pair = a_pair[:, :, 0] | (a_pair[:, :, 1] << 16)                                    # [n, axis] (left, right) halves
    for k in range(3):
        b_pair = ((a_pair[:, k] & 0x7FFF7FFF) + ((c2["delta"] >> (5 * k)) & 0x001F001F) + np.uint32(0xFFF0FFF1)).astype(np.uint32)
        assert np.array_equal(b_pair & 0xFFFF, n2["h"][:, 3 + k, 0])
        assert np.array_equal(b_pair >> 16, n2["h"][:, 3 + k, 1])
    for f in ("p1", "e1", "e2", "i1", "i2", "i3"):
        assert np.array_equal(ct[f], tr[f])
    assert np.array_equal(leaf_ref, tr["refLeaf"])


def test_flatten_pass_compact_refuses_a_box_it_cannot_encode(bunny):
    """A box far from the origin and thin against its distance: the rounding of the centre moves f16(a-c) and f16(b-c)
    apart by more than the 5-bit field holds.  The pass must say so instead of writing a wrong record."""
    nodes = bunny["nodes"].copy()
    k = int(np.nonzero(nodes["right"] > 0)[0][3]) + 1            # a left child somewhere in the tree
    nodes["a"][k] = (4096.0, 0.0, 0.0)
    nodes["b"][k] = (4096.0 + 2.0 ** -11 * 3, 1.0, 1.0)
    assert pt.flatten_bvh_compact(nodes, bunny["positions"], bunny["indices"]) is None
    pt.flatten_bvh(nodes, bunny["positions"], bunny["indices"])  # the 64 B pass takes any box


@pytest.mark.parametrize("kind", ["unit", "flat", "sliver", "offset", "huge"])
def test_bvh_builder_bit_identical_on_random_soups(oracle_lib, kind):
    """The parallel builder against the oracle's restatement of BVH.cpp:83-309 on awkward random inputs: every node
    byte and the reordered index array are the same."""
    import helpers
    import oracle
    rng = np.random.default_rng(7)
    for n in (2, 3, 5, 17, 64, 257, 2000):
        pos, idx = helpers.triangle_soup(rng, n, kind)
        for threads in (1, 4):
            nodes, ridx, depth = pt.build_bvh(pos, idx, threads)
            onodes, oridx, odepth = oracle.build_bvh(pos, idx)
            assert np.array_equal(nodes.view(np.uint8), onodes.view(np.uint8)), (kind, n, threads)
            assert np.array_equal(ridx, oridx) and depth == odepth


def _decode_treelet_stream(units, num_nodes):
    """Walks the depth-first stream of 64 B units from unit 0 the way the kernel's references do (first target behind
    the record, the other three through the link words) and returns the nodes it finds in the reference tree's
    pre-order - ("box", (c[3] f32, hA[3] u16, hB[3] u16 rebuilt from the 5-bit differences), is_leaf) per node below
    the root, ("leaf", 64 B record) after each leaf's box - plus the deepest level of records on any path."""
    flat = units.reshape(-1)
    out, deepest = [], [0]

    def half(pair, s):
        hA = pair["hA"][:, s].copy()
        hB = np.array([(int(hA[k]) & 0x7FFF) + ((int(pair["delta"]) >> (5 * k + 16 * s)) & 31) - 15 for k in range(3)], dtype=np.uint16)
        return pair["c"][:, s].copy(), hA, hB

    def target(is_leaf, u, level):
        if is_leaf:
            out.append(("leaf", flat[u * 16:u * 16 + 16].view(core.TRI_DT)[0]))
        else:
            visit(u, level)

    def visit(u, level):
        assert u % 2 == 0, "a record does not start on a 128 B line"
        deepest[0] = max(deepest[0], level)
        r = flat[u * 16:u * 16 + 32].view(core.TREELET_DT)[0]
        link = int(r["link"][0]) | (int(r["link"][1]) << 32)
        sh = 5 if link >> 63 else 0
        starts = [u + 2] + [u + (((link >> (21 * k)) & 0x1FFFFF) << sh) for k in range(3)]
        d = [int(r["pair"][k]["delta"]) for k in range(3)]
        for side, bit, pair, st in ((0, 0x8000, 1, starts[0:2]), (1, 0x80000000, 2, starts[2:4])):
            half_is_leaf = bool(d[0] & bit)
            out.append(("box", half(r["pair"][0], side), half_is_leaf))
            if half_is_leaf:
                target(True, st[0], level + 1)
            else:
                for s2, bit2 in ((0, 0x8000), (1, 0x80000000)):
                    child_is_leaf = bool(d[pair] & bit2)
                    out.append(("box", half(r["pair"][pair], s2), child_is_leaf))
                    target(child_is_leaf, st[s2], level + 1)

    visit(0, 1)
    return out, deepest[0]


@pytest.mark.parametrize("fixture", ["monkey", "bunny"])
def test_treelet_stream_is_the_reference_tree(request, fixture):
    """The 128 B treelet layout holds the reference tree, box for box and leaf for leaf: walking the stream through its
    own links visits the nodes in the reference's pre-order; every box decodes (centre, f16(a-c), f16(b-c) rebuilt from
    the 5-bit difference as the kernel rebuilds it) to the digest of that reference node's box; every leaf unit is the
    64 B leaf record of the two-box layout; records sit on 128 B lines; and the stack bound covers the deepest path."""
    m = request.getfixturevalue(fixture)
    nodes = m["nodes"]
    out = pt.flatten_bvh_treelets(nodes, m["positions"], m["indices"])
    assert out is not None
    units, rb, max_stack = out
    n2, tr, rb2, ms2 = pt.flatten_bvh(nodes, m["positions"], m["indices"])
    assert np.array_equal(rb, rb2)
    walk, deepest = _decode_treelet_stream(units, nodes.shape[0])
    assert max_stack == 3 * deepest and max_stack >= ms2
    # the walk lists every node but the root: a "box" entry per node in pre-order, a "leaf" entry after each leaf's box
    boxes = [w for w in walk if w[0] == "box"]
    leaves = [w[1] for w in walk if w[0] == "leaf"]
    assert len(boxes) == nodes.shape[0] - 1 and len(leaves) == m["indices"].shape[0]
    c = (nodes["a"] + nodes["b"]) * np.float32(0.5)
    hA = (nodes["a"] - c).astype(np.float16).view(np.uint16)
    hB = (nodes["b"] - c).astype(np.float16).view(np.uint16)
    for i, (_, (bc, ba, bb), is_leaf) in enumerate(boxes, start=1):
        assert np.array_equal(bc, c[i]) and np.array_equal(ba, hA[i]) and np.array_equal(bb, hB[i]), f"box of node {i}"
        assert is_leaf == (nodes["right"][i] <= 0)
    order = [int(-nodes["right"][i]) // 3 for i in range(nodes.shape[0]) if nodes["right"][i] <= 0]      # leaves in pre-order
    for leaf, t in zip(leaves, order):
        assert leaf.tobytes() == tr[t].tobytes()
    assert units.shape[0] <= 3 * nodes.shape[0]


def test_treelet_pass_small_trees_and_refusals(bunny):
    """One, two and three triangles (a root that is a leaf, a record with two leaf halves, a record with one interior
    half), a box the 5-bit difference cannot hold, and a malformed tree."""
    for n in (1, 2, 3, 4, 5):
        pos = np.ones((3 * n, 4), np.float32)
        pos[:, :3] = np.random.default_rng(n).uniform(-1, 1, (3 * n, 3))
        idx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
        nodes, ridx, _ = pt.build_bvh(pos, idx, 1)
        out = pt.flatten_bvh_treelets(nodes, pos, ridx)
        assert out is not None
        units, _, ms = out
        walk, deepest = _decode_treelet_stream(units, nodes.shape[0]) if n > 1 else ([], 0)
        if n == 1:
            assert units.shape[0] == 1 and ms == 0
        else:
            assert sum(1 for w in walk if w[0] == "leaf") == n and ms == 3 * deepest
    nodes = bunny["nodes"].copy()
    k = int(np.nonzero(nodes["right"] > 0)[0][3]) + 1
    nodes["a"][k] = (4096.0, 0.0, 0.0)
    nodes["b"][k] = (4096.0 + 2.0 ** -11 * 3, 1.0, 1.0)
    assert pt.flatten_bvh_treelets(nodes, bunny["positions"], bunny["indices"]) is None
    bad = bunny["nodes"].copy(); bad["next"][5] = 3
    with pytest.raises(pt.PtError):
        pt.flatten_bvh_treelets(bad, bunny["positions"], bunny["indices"])
