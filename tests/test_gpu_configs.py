"""The two large rendering configurations, on stand-in scenes (the reference's sponza.ply and SanMiguel.ply are
Git-LFS pointer stubs, Models/*.ply:1-3):

  * sponza-class scene, 1920 x 1080, 8 bounces with next-event-estimation shadow rays (the reference's frame graph,
    PathTracer.cpp:776-838, with numRays raised past its cap of 5, PathTracer.cpp:96,448);
  * San-Miguel-class scene, 3840 x 2160 x 8 bounces, samples partitioned over ranks and the RGBA32F accumulations
    summed (.w, the sample count, included).

Byte parity with the CPU oracle at a size the oracle serves in seconds (256 x 144), on the reference-order walk and
on the certified walk; at full size the size-independent properties: sample counts, finiteness, the ray-count
bookkeeping closest[d+1] <= shadow[d] = hits[d] <= closest[d], run-to-run identical bytes, and the certified walk's
image equal to the reference-order walk's byte for byte.  The partition check plays two ranks on one GPU with two
contexts and sums their accumulations on the host.
"""
import numpy as np
import pytest

import helpers
import oracle
import pathtracer_amd as pt
from pathtracer_amd import core, distributed as ptd
from test_gpu_frame import _compare, _consts

pytestmark = pytest.mark.gpu

CAM = pt.scenes.COURTYARD_CAMERA
SPONZA_CAM = pt.scenes.SPONZA_CAMERA


def _model(pos, aux, idx, threads=16):
    nodes, ridx, depth = pt.build_bvh(pos, idx, threads)
    return dict(positions=pos, aux=aux, indices=ridx, nodes=nodes, depth=depth)


@pytest.fixture(scope="module")
def sponza():
    return _model(*pt.scenes.sponza_like())


@pytest.fixture(scope="module")
def san_miguel():
    return _model(*pt.scenes.san_miguel_like())


@pytest.fixture()
def ctx():
    c = pt.PathTracer(0)
    yield c
    c.close()


def _prepare(ctx, model, W, H, NR, trav, reseed=0, in_flight=3, cam=CAM):
    helpers.upload(ctx, model, pt.default_lights(), pt.load_blue_noise())
    ctx.resize(W, H, NR)
    for o, v in ((core.OPT_TRAVERSAL, trav), (core.OPT_STALE_SLOTS, 0), (core.OPT_RESEED, reseed), (core.OPT_FRAMES_IN_FLIGHT, in_flight)):
        ctx.set_option(o, v)
    ctx.set_constants(_consts(model, W, H, NR, 1, cam))


def _check_counts(rc, NR, W, H, frames=1):
    closest, shadow = rc[0:2 * NR:2].astype(np.int64), rc[1:2 * NR:2].astype(np.int64)
    assert closest[0] == frames * W * H
    assert (shadow <= closest).all()                      # one shadow ray per hit (trace.comp:123)
    assert (closest[1:] <= shadow[:-1]).all()             # a path goes on only from a hit (trace.comp:212-227)
    assert closest[-1] > 0                                # some paths do reach the last bounce


def _check_image(acc, frames):
    assert (acc[..., 3] == frames).all()
    assert np.isfinite(acc).all() and (acc[..., :3] >= 0).all()
    assert acc[..., :3].mean() > 1e-3 * frames            # the picture is not black


@pytest.mark.parametrize("trav", [core.TRAVERSAL_FLAT, core.TRAVERSAL_CERTIFIED])
def test_sponza_8_bounces_nee_oracle_parity(ctx, oracle_lib, sponza, trav):
    """Every byte of the ray stack, the current image and the accumulation equals the oracle's after each of two
    frames of 8 bounces with shadow rays on the 262 k-triangle sponza stand-in."""
    W, H, NR = 256, 144, 8
    _prepare(ctx, sponza, W, H, NR, trav, in_flight=1, cam=SPONZA_CAM)
    osc = helpers.oracle_scene(sponza, pt.default_lights(), pt.load_blue_noise())
    fr = oracle.Frame(W, H, NR)
    for frame in (1, 2):
        c = _consts(sponza, W, H, NR, frame, SPONZA_CAM)
        ctx.set_constants(c)
        rc = ctx.render_sample(frame, want_ray_counts=True)
        before = fr.ray_counts().copy()
        oracle.render_frame(osc, fr, c, mode=oracle.MODE_ZERO_NEXT_SLOT | oracle.MODE_KILL_ON_MISS)
        after = fr.ray_counts()
        _compare(ctx, fr)
        assert int(rc[0:2 * NR:2].sum()) == int(after[0] - before[0])
        assert int(rc[1:2 * NR:2].sum()) == int(after[1] - before[1])
        _check_counts(rc, NR, W, H)
    assert float(rc[2 * (NR - 1)]) > 0.01 * W * H          # the eighth bounce is not a formality on this camera


def test_sponza_1080p_8_bounces_properties(ctx, sponza):
    """BASELINE configuration: 1920 x 1080, 8 bounces + next-event estimation, frames in flight as the library
    defaults.  The default (near-child-first) walk by its properties and twice for determinism; the certified walk
    against the reference-order walk byte for byte."""
    W, H, NR, NF = 1920, 1080, 8, 4
    _prepare(ctx, sponza, W, H, NR, core.TRAVERSAL_FLAT_ORDERED, cam=SPONZA_CAM)
    images = []
    for rep in range(2):
        ctx.clear_accumulation()
        rc = ctx.render_samples(1, NF, want_ray_counts=True)
        _check_counts(rc, NR, W, H, NF)
        acc = ctx.readback_accumulation()
        _check_image(acc, NF)
        images.append(acc)
    assert np.array_equal(images[0], images[1]), "two runs of the same frames differ"
    exact = {}
    for trav in (core.TRAVERSAL_FLAT, core.TRAVERSAL_CERTIFIED):
        ctx.set_option(core.OPT_TRAVERSAL, trav)
        ctx.clear_accumulation()
        ctx.render_samples(1, NF)
        exact[trav] = (ctx.readback_accumulation(), ctx.readback_current())
    a, b = exact[core.TRAVERSAL_FLAT], exact[core.TRAVERSAL_CERTIFIED]
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1], b[1])
    # the default walk's image is the reference-order image up to the few near-tie rays: a pixel differs only where one
    # of its paths met such a ray (at most 1e-4 of rays do, and a path has up to 8 closest-hit rays per frame), and the
    # RMSE over the whole image stays within 5 % of the mean radiance at these 4 spp
    differ = (images[0] != a[0]).any(axis=2).mean()
    rmse = np.sqrt(((images[0] - a[0])[..., :3] ** 2).mean())
    print(f"near-first against reference order: {differ:.2e} of pixels differ, RMSE {rmse:.3e} at mean {a[0][..., :3].mean():.3e}")
    assert differ <= 1e-4 * NR * NF and rmse <= 0.05 * a[0][..., :3].mean()


def test_san_miguel_4k_8_bounces_properties(ctx, san_miguel):
    """BASELINE configuration, one rank's work: 3840 x 2160 x 8 bounces on the 5.8 M-triangle stand-in."""
    W, H, NR, NF = 3840, 2160, 8, 2
    _prepare(ctx, san_miguel, W, H, NR, core.TRAVERSAL_FLAT_ORDERED, reseed=1, in_flight=2)
    rc = ctx.render_samples(1, NF, want_ray_counts=True)
    _check_counts(rc, NR, W, H, NF)
    acc = ctx.readback_accumulation()
    _check_image(acc, NF)
    first = acc.copy()
    ctx.clear_accumulation()
    ctx.render_samples(1, NF)
    assert np.array_equal(ctx.readback_accumulation(), first), "two runs of the same frames differ"


@pytest.mark.parametrize("size", [(3840, 2160), (480, 272)])
def test_sample_partition_sums_to_the_single_context_image(san_miguel, size):
    """Two contexts play ranks 0 and 1 of a sample-partitioned render (frames dealt round-robin, per-frame reseed);
    their accumulations summed on the host - what the all-reduce computes - equal one context rendering all frames:
    .w exactly, radiance to the fp32 rounding of sums taken in a different order (rtol 1e-5)."""
    W, H = size
    NR, NF, world = 8, 4, 2
    ranks = [pt.PathTracer(0) for _ in range(world)]
    single = pt.PathTracer(0)
    try:
        parts = []
        for rank, c in enumerate(ranks):
            _prepare(c, san_miguel, W, H, NR, core.TRAVERSAL_FLAT_ORDERED, reseed=1, in_flight=2)
            n = ptd.render_partitioned(lambda f, c=c: c.render_samples(f, 1), 1, NF, rank, world)      # queued, not waited for
            assert n == NF // world
        for c in ranks:
            parts.append(c.readback_accumulation())
        # the same frames dealt in blocks, one pt_render_samples call per rank (what the frame train likes): the same sum
        blocks = []
        for rank, c in enumerate(ranks):
            c.clear_accumulation()
            assert ptd.render_block(c, 1, NF, rank, world) == NF // world
        for c in ranks:
            blocks.append(c.readback_accumulation())
            c.close()
        _prepare(single, san_miguel, W, H, NR, core.TRAVERSAL_FLAT_ORDERED, reseed=1, in_flight=2)
        single.render_samples(1, NF)
        whole = single.readback_accumulation()
        total = parts[0] + parts[1]
        assert (parts[0][..., 3] == NF // world).all() and (parts[1][..., 3] == NF // world).all()
        assert np.array_equal(total[..., 3], whole[..., 3]) and (whole[..., 3] == NF).all()
        assert np.allclose(total, whole, rtol=1e-5, atol=1e-6)
        assert np.array_equal((blocks[0] + blocks[1])[..., 3], whole[..., 3]) and np.allclose(blocks[0] + blocks[1], whole, rtol=1e-5, atol=1e-6)
    finally:
        for c in ranks:
            c.close()
        single.close()


def test_treelet_layout_equals_two_box_layout_at_scale(san_miguel):
    """The 128 B two-level treelet records against the 64 B two-box records on the 5.8 M-triangle stand-in, 2^24
    incoherent rays: hits (ids and t) byte for byte in the reference-order walk and in the near-child-first walk,
    closest-hit and any-hit (the walk visits the same leaves in the same order with the same max_t in both layouts)."""
    n = 1 << 24
    out = {}
    for layout in (core.LAYOUT_RECORDS64, core.LAYOUT_TREELETS):
        c = pt.PathTracer(0)
        try:
            c.set_option(core.OPT_RECORD_LAYOUT, layout)
            helpers.upload(c, san_miguel)
            d_rays, d_hits = c.alloc(n * 32), c.alloc(n * 16)
            c.generate_rays(d_rays, n, seed=5)
            for trav in (core.TRAVERSAL_FLAT, core.TRAVERSAL_FLAT_ORDERED):
                c.set_option(core.OPT_TRAVERSAL, trav)
                for any_hit in (False, True):
                    m = n if trav == core.TRAVERSAL_FLAT_ORDERED else n // 4        # the reference order is the slow one
                    c.trace_rays_device(d_rays, m, d_hits, any_hit=any_hit)
                    out[(layout, trav, any_hit)] = d_hits.to_host(pt.HIT_DT, m)
            d_rays.free(); d_hits.free()
        finally:
            c.close()
    for trav in (core.TRAVERSAL_FLAT, core.TRAVERSAL_FLAT_ORDERED):
        for any_hit in (False, True):
            a, b = out[(core.LAYOUT_RECORDS64, trav, any_hit)], out[(core.LAYOUT_TREELETS, trav, any_hit)]
            assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), f"traversal {trav}, any_hit {any_hit}: {int((a != b).sum())} hits differ"
            assert 0.5 < (a["i1"] >= 0).mean() < 0.95


def test_scene_past_two_gib_of_records(oracle_lib):
    """A 21 M-triangle scene: 2.7 GB of traversal records, past the point where a signed 32-bit byte offset or buffer size
    would wrap (the record buffer is addressed through one 4 GB buffer descriptor with unsigned 32-bit offsets; uploads past
    that are refused).  The reference-order and certified walks equal the oracle's walk over the same tree bit for bit on a
    sample of rays, the near-first walk within its bar, and on 2^22 rays on the device the certified walk equals the
    reference-order walk; the treelet layout agrees as well."""
    pos, aux, idx = pt.scenes.courtyard(21_000_000, seed=17)
    model = _model(pos, aux, idx)
    T = model["indices"].shape[0]
    assert 64 * (2 * T - 1) > (1 << 31)
    lo, hi = model["nodes"][0]["a"], model["nodes"][0]["b"]
    rays = helpers.random_rays(lo, hi, 30_000, seed=5)
    osc = helpers.oracle_scene(model)
    want = osc.trace(rays)
    n = 1 << 22
    dev = {}
    for layout in (core.LAYOUT_RECORDS64, core.LAYOUT_TREELETS):
        c = pt.PathTracer(0)
        try:
            c.set_option(core.OPT_RECORD_LAYOUT, layout)
            helpers.upload(c, model)
            for trav in (core.TRAVERSAL_FLAT, core.TRAVERSAL_CERTIFIED, core.TRAVERSAL_FLAT_ORDERED):
                c.set_option(core.OPT_TRAVERSAL, trav)
                got = c.trace_rays(rays)
                if trav == core.TRAVERSAL_FLAT_ORDERED:
                    assert (got["i1"] != want["i1"]).mean() <= 1e-4
                else:
                    assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), f"layout {layout}, traversal {trav}"
            d_rays, d_hits = c.alloc(n * 32), c.alloc(n * 16)
            c.generate_rays(d_rays, n, seed=9)
            for trav in (core.TRAVERSAL_FLAT, core.TRAVERSAL_CERTIFIED):
                c.set_option(core.OPT_TRAVERSAL, trav)
                c.trace_rays_device(d_rays, n, d_hits)
                dev[(layout, trav)] = d_hits.to_host(pt.HIT_DT, n)
            d_rays.free(); d_hits.free()
        finally:
            c.close()
    ref = dev[(core.LAYOUT_RECORDS64, core.TRAVERSAL_FLAT)]
    assert 0.5 < (ref["i1"] >= 0).mean() < 0.99
    for key, got in dev.items():
        assert np.array_equal(got.view(np.uint8), ref.view(np.uint8)), f"{key} differs from the reference-order walk on 64 B records"
