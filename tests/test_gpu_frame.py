"""Parity of the wavefront stages (launch, closest-hit, shade + NEE, shadow, compaction, resolve +
accumulate) with the CPU oracle, through the C ABI dispatch sites.

Bar: every byte of the ray stack, the RGBA16F current image and the RGBA32F accumulation equal the
oracle's for the reference-order traversal.  The only arithmetic not guaranteed identical is the
fp32 pow/sin/cos under the fp16 shading (libm on the CPU, ocml on the GPU, both rounded to fp16);
a differing pixel would show up here as an inequality and is bounded at 1e-3 of pixels and an RMSE
of 1e-3 x mean radiance."""
import numpy as np
import pytest

import helpers
import oracle
import pathtracer_amd as pt
from pathtracer_amd import core

pytestmark = pytest.mark.gpu

BUNNY_CAM = dict(orbit_angle=-np.pi / 2, orbit_height=0.75, focus_center=(0.0, 0.75, 0.0))


def _setup(ctx, model, W, H, NR):
    lights, bn = pt.default_lights(), pt.load_blue_noise()
    helpers.upload(ctx, model, lights, bn)
    ctx.resize(W, H, NR)
    for o, v in ((core.OPT_TRAVERSAL, core.TRAVERSAL_FLAT), (core.OPT_STALE_SLOTS, 0), (core.OPT_RESEED, 0), (core.OPT_FRAMES_IN_FLIGHT, 2),
                 (core.OPT_FRAME_TRAIN, 1), (core.OPT_TRAIN_PARTS, 2), (core.OPT_TRAIN_MAX_MB, 0), (core.OPT_TRAIN_KEEP, 0), (core.OPT_MERGE_PASSES, 1), (core.OPT_SHADOW_STREAM, 2)):
        ctx.set_option(o, v)
    return helpers.oracle_scene(model, lights, bn), oracle.Frame(W, H, NR)


def _consts(model, W, H, NR, frame, cam):
    return pt.fill_constants(W, H, NR, model["indices"].shape[0], model["nodes"].shape[0], frame_index=frame, **cam)


def _compare(ctx, fr, strict=True, live_slots_only=False):
    slots, _ = ctx.readback_raystack()
    acc, cur = ctx.readback_accumulation(), ctx.readback_current()
    if strict:
        if live_slots_only:
            # slots the frame's paths reached; what lies beyond a path's end is left over from whichever earlier
            # frame used this stack and never enters an image (the build zeroes prob of the slot after the end)
            nr = fr.stack.shape[0] // (cur.shape[0] * cur.shape[1])
            a, b = slots.reshape(-1, nr), fr.stack.reshape(-1, nr)
            prob = b["prob"].view(np.float16).astype(np.float32)
            reached = np.ones(prob.shape, bool)
            reached[:, 1:] = np.cumprod(prob[:, :-1] > 0, axis=1).astype(bool)
            assert np.array_equal(a["prob"][reached], b["prob"][reached])
            full = reached & (prob > 0)
            assert np.array_equal(a[full].view(np.uint8), b[full].view(np.uint8)), "ray stack differs in a live slot"
        else:
            assert np.array_equal(slots.view(np.uint8), fr.stack.view(np.uint8)), "ray stack differs"
        assert np.array_equal(cur, fr.current)
        assert np.array_equal(acc, fr.accumulation)
    else:
        bad = (np.abs(acc - fr.accumulation).max(axis=2) > 0).mean()
        rmse = np.sqrt(((acc - fr.accumulation)[..., :3] ** 2).mean())
        assert bad <= 1e-3 and rmse <= 1e-3 * max(fr.accumulation[..., :3].mean(), 1e-6)


@pytest.mark.parametrize("name,cam", [("bunny", BUNNY_CAM), ("monkey", {})])
@pytest.mark.parametrize("trav", [core.TRAVERSAL_FLAT, core.TRAVERSAL_REFERENCE_LAYOUT, core.TRAVERSAL_CERTIFIED])
def test_frames_match_oracle(gpu_ctx, oracle_lib, request, name, cam, trav):
    model = request.getfixturevalue(name)
    W, H, NR = 256, 144, 4
    osc, fr = _setup(gpu_ctx, model, W, H, NR)
    gpu_ctx.set_option(core.OPT_TRAVERSAL, trav)
    for frame in (1, 2, 3):
        c = _consts(model, W, H, NR, frame, cam)
        gpu_ctx.set_constants(c)
        rc = gpu_ctx.render_sample(frame, want_ray_counts=True)
        before = fr.ray_counts().copy()
        oracle.render_frame(osc, fr, c, mode=oracle.MODE_ZERO_NEXT_SLOT | oracle.MODE_KILL_ON_MISS)
        after = fr.ray_counts()
        _compare(gpu_ctx, fr)
        assert int(rc[0::2][:NR].sum()) == int(after[0] - before[0])       # closest-hit rays
        assert int(rc[1::2][:NR].sum()) == int(after[1] - before[1])       # shadow rays
        assert int(rc[0]) == W * H
    assert (gpu_ctx.readback_accumulation()[..., 3] == 3).all()


def test_dispatch_sites_one_by_one(gpu_ctx, oracle_lib, monkey):
    """launch / trace / compact / resolve called individually, with and without compaction, equal render_sample."""
    W, H, NR = 128, 72, 3
    osc, fr = _setup(gpu_ctx, monkey, W, H, NR)
    c = _consts(monkey, W, H, NR, 1, {})
    gpu_ctx.set_constants(c)
    gpu_ctx.launch()
    slots, jobs = gpu_ctx.readback_raystack()
    oracle.launch(osc, fr, c)
    assert np.array_equal(slots.view(np.uint8), fr.stack.view(np.uint8)) and np.array_equal(jobs, fr.jobs)
    for d in range(NR):
        gpu_ctx.trace(d)
        oracle.trace(osc, fr, c, mode=oracle.MODE_ZERO_NEXT_SLOT | oracle.MODE_KILL_ON_MISS)
        slots, jobs = gpu_ctx.readback_raystack()
        assert np.array_equal(slots.view(np.uint8), fr.stack.view(np.uint8)), f"depth {d}"
        live_g = np.sort(jobs[jobs != 0xFFFFFFFF]); live_o = np.sort(fr.jobs[fr.jobs != 0xFFFFFFFF])
        assert np.array_equal(live_g, live_o)
        if d == 0:      # compaction keeps exactly the live jobs, densely packed (raysort.comp:23-55 keeps them per segment)
            gpu_ctx.compact()
            oracle.raysort(fr, c)
            _, jobs = gpu_ctx.readback_raystack()
            assert np.array_equal(np.sort(jobs[:len(live_o)]), live_o)
    gpu_ctx.resolve_accumulate()
    oracle.composite(fr, c)
    _compare(gpu_ctx, fr)


def test_speculative_depth0_matches_subgroup64(gpu_ctx, oracle_lib, monkey):
    """The SPECULATIVE build (subgroup vote) at depth 0 with 64-lane subgroups on 8x8 tiles: the vote governs the
    closest-hit walk and the shadow walk shadeHit makes (the whole of trace.comp is compiled with it,
    CMakeLists.txt:33-36; intersections.glsl:73-77 via trace.comp:123), among the lanes whose closest-hit walk hit.
    That the shadow walk's vote matters on this scene is checked against an oracle run without it."""
    W, H, NR = 128, 72, 3
    osc, fr = _setup(gpu_ctx, monkey, W, H, NR)
    plain = oracle.Frame(W, H, NR)
    mode = oracle.MODE_ZERO_NEXT_SLOT | oracle.MODE_KILL_ON_MISS
    for frame in (1, 2):
        c = _consts(monkey, W, H, NR, frame, {})
        gpu_ctx.set_constants(c)
        gpu_ctx.render_sample(frame, speculative_depth0=True)
        oracle.render_frame(osc, fr, c, subgroup0=64, mode=mode)
        oracle.render_frame(osc, plain, c, subgroup0=64, mode=mode | oracle.MODE_PLAIN_SHADOW_WALK)
        _compare(gpu_ctx, fr)
    differ = int((fr.accumulation != plain.accumulation).any(axis=2).sum())
    print(f"the vote in the shadow walk changes {differ} of {W * H} pixels")
    assert differ > 0


def test_stale_slot_and_reseed_options(gpu_ctx, oracle_lib, monkey):
    W, H, NR = 96, 64, 4
    osc, fr = _setup(gpu_ctx, monkey, W, H, NR)
    gpu_ctx.set_option(core.OPT_STALE_SLOTS, 1)
    gpu_ctx.set_option(core.OPT_RESEED, 1)
    for frame in (1, 2, 3):
        c = _consts(monkey, W, H, NR, frame, {})
        gpu_ctx.set_constants(c)
        gpu_ctx.render_sample(frame)
        oracle.render_frame(osc, fr, c, mode=oracle.MODE_RESEED_EVERY_FRAME | oracle.MODE_KILL_ON_MISS)
        _compare(gpu_ctx, fr)
    gpu_ctx.set_option(core.OPT_STALE_SLOTS, 0)
    gpu_ctx.set_option(core.OPT_RESEED, 0)


def test_eight_slots_and_clear(gpu_ctx, oracle_lib, bunny):
    """numRays = 8 (beyond the reference's cap of 5, PathTracer.cpp:96,448) and the accumulation clear."""
    W, H, NR = 160, 96, 8
    osc, fr = _setup(gpu_ctx, bunny, W, H, NR)
    for frame in (1, 2):
        c = _consts(bunny, W, H, NR, frame, BUNNY_CAM)
        gpu_ctx.set_constants(c)
        gpu_ctx.render_sample(frame)
        oracle.render_frame(osc, fr, c, mode=oracle.MODE_ZERO_NEXT_SLOT | oracle.MODE_KILL_ON_MISS)
    _compare(gpu_ctx, fr)
    gpu_ctx.clear_accumulation()
    fr.clear_accumulation()
    c = _consts(bunny, W, H, NR, 3, BUNNY_CAM)
    gpu_ctx.set_constants(c)
    gpu_ctx.render_sample(3)
    oracle.render_frame(osc, fr, c, mode=oracle.MODE_ZERO_NEXT_SLOT | oracle.MODE_KILL_ON_MISS)
    _compare(gpu_ctx, fr)          # launch reseeds after the clear (launch.comp:49-50)


def test_full_size_properties_1080p(gpu_ctx, bunny):
    """BASELINE config: CBbunny 1080p, primary + one bounce (numRays = 2).  Too large for the scalar oracle in a
    test; checked through properties: sample counts, finiteness, ray-count bookkeeping, determinism."""
    W, H, NR = 1920, 1080, 2
    lights, bn = pt.default_lights(), pt.load_blue_noise()
    helpers.upload(gpu_ctx, bunny, lights, bn)
    gpu_ctx.resize(W, H, NR)
    gpu_ctx.set_option(core.OPT_TRAVERSAL, core.TRAVERSAL_FLAT)
    c = _consts(bunny, W, H, NR, 1, BUNNY_CAM)
    gpu_ctx.set_constants(c)
    images = []
    for rep in range(2):
        gpu_ctx.clear_accumulation()
        tot = np.zeros(2 * NR, dtype=np.uint64)
        for frame in (1, 2, 3, 4):
            rc = gpu_ctx.render_sample(frame, want_ray_counts=True)
            assert rc[0] == W * H and rc[2] <= rc[1] and rc[3] <= rc[2]      # closest[d+1] <= shadow[d] = hits[d]
            tot += rc[:2 * NR]
        acc = gpu_ctx.readback_accumulation()
        assert (acc[..., 3] == 4).all() and np.isfinite(acc).all() and (acc[..., :3] >= 0).all()
        assert acc[..., :3].mean() > 1e-3
        images.append(acc)
    assert np.array_equal(images[0], images[1])


def test_ordered_traversal_image_rmse(gpu_ctx, oracle_lib, monkey):
    """Near-child-first traversal changes a small fraction of hits (fp16 box-test edge cases); the image stays
    within an RMSE of 2 % of the mean radiance after 4 samples at equal seed."""
    W, H, NR = 128, 72, 4
    osc, fr = _setup(gpu_ctx, monkey, W, H, NR)
    gpu_ctx.set_option(core.OPT_TRAVERSAL, core.TRAVERSAL_FLAT_ORDERED)
    gpu_ctx.set_option(core.OPT_RESEED, 1)
    for frame in (1, 2, 3, 4):
        c = _consts(monkey, W, H, NR, frame, {})
        gpu_ctx.set_constants(c)
        gpu_ctx.render_sample(frame)
        oracle.render_frame(osc, fr, c, mode=oracle.MODE_ZERO_NEXT_SLOT | oracle.MODE_RESEED_EVERY_FRAME)
    acc = gpu_ctx.readback_accumulation()
    rmse = np.sqrt(((acc - fr.accumulation)[..., :3] ** 2).mean()) / 4
    mean = fr.accumulation[..., :3].mean() / 4
    print(f"ordered traversal: image rmse {rmse:.3e} vs mean radiance {mean:.3e}")
    assert rmse <= 0.02 * mean
    gpu_ctx.set_option(core.OPT_TRAVERSAL, core.TRAVERSAL_FLAT)
    gpu_ctx.set_option(core.OPT_RESEED, 0)


def test_error_behaviour(oracle_lib, bunny):
    ctx = pt.PathTracer(0)
    try:
        # a new context starts with the walk whose results are the reference's
        assert ctx.get_option(core.OPT_TRAVERSAL) == core.TRAVERSAL_DEFAULT == core.TRAVERSAL_CERTIFIED
        assert ctx.get_option(core.OPT_SECOND_WALK) == 2
        with pytest.raises(pt.PtError):
            ctx.launch()                                   # nothing uploaded
        with pytest.raises(pt.PtError):
            ctx.trace_rays(np.zeros(4, dtype=pt.RAY_DT))
        bad = bunny["nodes"].copy(); bad["right"][0] = 10 ** 9
        with pytest.raises(pt.PtError):
            ctx.upload_scene(bunny["positions"], bunny["aux"], bunny["indices"], bad)
        helpers.upload(ctx, bunny, pt.default_lights(), pt.load_blue_noise())
        with pytest.raises(pt.PtError):
            ctx.launch()                                   # no resize yet
        ctx.resize(64, 64, 2)
        with pytest.raises(pt.PtError):
            ctx.set_constants(pt.fill_constants(64, 64, 5, bunny["indices"].shape[0], bunny["nodes"].shape[0]))   # numRays > slots
        with pytest.raises(pt.PtError):
            ctx.set_constants(pt.fill_constants(32, 64, 2, bunny["indices"].shape[0], bunny["nodes"].shape[0]))   # wrong viewport
        with pytest.raises(pt.PtError):
            ctx.set_option(core.OPT_TRAVERSAL, 9)
        # every option: a value in range reads back, a value out of range is refused and leaves the option as it was
        for opt, good, bad in ((core.OPT_TRAVERSAL, 1, 4), (core.OPT_PERSIST_BLOCKS, 4, 0), (core.OPT_REFILL_MIN, 20, 65),
                               (core.OPT_CLAIM_CHUNK, 256, 100), (core.OPT_FRAMES_IN_FLIGHT, 4, 9), (core.OPT_RECORD_LAYOUT, 1, 3),
                               (core.OPT_SORT_RAYS, 1, 3), (core.OPT_SORT_BITS, 2, 5), (core.OPT_FRAME_TRAIN, 1, 3), (core.OPT_TRAIN_PARTS, 3, 5),
                               (core.OPT_TRAIN_MAX_MB, 64, -1), (core.OPT_SHADOW_STREAM, 1, 3), (core.OPT_SECOND_WALK, 0, 3)):
            ctx.set_option(opt, good)
            assert ctx.get_option(opt) == good
            with pytest.raises(pt.PtError):
                ctx.set_option(opt, bad)
            assert ctx.get_option(opt) == good
        with pytest.raises(pt.PtError):
            ctx.set_option(99, 0)
        ctx.set_option(core.OPT_TRAVERSAL, core.TRAVERSAL_DEFAULT)
        with pytest.raises(pt.PtError):
            ctx.set_constants(pt.fill_constants(64, 64, 2, bunny["indices"].shape[0], bunny["nodes"].shape[0] + 2))   # wrong node count
        ctx.set_constants(pt.fill_constants(64, 64, 2, bunny["indices"].shape[0], bunny["nodes"].shape[0]))
        ctx.render_sample(1)
        assert (ctx.readback_accumulation()[..., 3] == 1).all()
        ctx.resize(32, 48, 2)                               # the 64 x 64 constants do not fit this size: dropped
        with pytest.raises(pt.PtError):
            ctx.launch()
        ctx.set_constants(pt.fill_constants(32, 48, 2, bunny["indices"].shape[0], bunny["nodes"].shape[0]))
        ctx.render_sample(2)
        assert (ctx.readback_accumulation()[..., 3] == 1).all()
    finally:
        ctx.close()


def test_allreduce_single_rank(gpu_ctx, monkey):
    """RCCL path with one rank: the sum over one rank leaves the buffer as it is, through both routes."""
    from pathtracer_amd import distributed as ptd
    W, H, NR = 64, 48, 2
    helpers.upload(gpu_ctx, monkey, pt.default_lights(), pt.load_blue_noise())
    gpu_ctx.resize(W, H, NR)
    gpu_ctx.set_constants(_consts(monkey, W, H, NR, 1, {}))
    gpu_ctx.render_sample(1)
    before = gpu_ctx.readback_accumulation()
    gpu_ctx.comm_init(0, 1, core.comm_unique_id())
    gpu_ctx.allreduce_accumulation()
    gpu_ctx.synchronize()
    assert np.array_equal(gpu_ctx.readback_accumulation(), before)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29655", rank=0, world_size=1)
    try:
        t = ptd.allreduce_accumulation_torch(gpu_ctx)
        assert t.data_ptr() == gpu_ctx.accumulation_device_ptr()
        assert np.array_equal(gpu_ctx.readback_accumulation(), before)
    finally:
        dist.destroy_process_group()


def test_tonemap_and_frame_dump(gpu_ctx, oracle_lib, monkey, tmp_path):
    """Display colour (composite.frag:18-26,57) against the oracle's restatement on the same accumulation, and the
    frame-dump pair (PathTracer.cpp:180-202) holding exactly the RGBA16F current image.
    Tolerance: the colour is fp32 with one powf (ocml on the GPU, libm in the oracle): 1e-5 relative to the
    [0,1] range; the 8-bit image may differ by one step where the colour sits on a rounding boundary."""
    import zlib
    W, H, NR = 192, 104, 4
    _setup(gpu_ctx, monkey, W, H, NR)
    for frame in (1, 2, 3, 4):
        gpu_ctx.set_constants(_consts(monkey, W, H, NR, frame, {}))
        gpu_ctx.render_sample(frame)
    acc, cur = gpu_ctx.readback_accumulation(), gpu_ctx.readback_current()
    ref = oracle.tonemap(acc)
    col = gpu_ctx.tonemap(want_float=True)
    assert np.abs(col - ref).max() <= 1e-5
    img = gpu_ctx.tonemap()
    assert (img[..., 3] == 255).all()
    step = np.abs(img[..., :3].astype(np.int32) - np.rint(ref * 255.0).astype(np.int32))
    assert step.max() <= 1 and (step > 0).mean() <= 1e-3
    assert img[..., :3].max() > 64          # the picture is not black

    gpu_ctx.dump_current(tmp_path, 4)
    assert zlib.decompress((tmp_path / "RayStackBuffer_4.bin.z").read_bytes()) == cur.tobytes()
    assert (tmp_path / "RayStackBuffer_4.bin.z.txt").read_text() == f"{W},{H}\n"
    assert np.array_equal(pt.read_frame_dump(tmp_path, 4), cur)
    pt.write_ppm(tmp_path / "monkey.ppm", img)
    assert (tmp_path / "monkey.ppm").stat().st_size == len(f"P6\n{W} {H}\n255\n") + W * H * 3


@pytest.mark.parametrize("trav", [core.TRAVERSAL_FLAT, core.TRAVERSAL_FLAT_ORDERED])
@pytest.mark.parametrize("shadow_stream", [0, 1])
def test_merged_passes_and_stream_options_change_nothing(gpu_ctx, oracle_lib, monkey, trav, shadow_stream):
    """PT_OPT_MERGE_PASSES (the shadow rays of bounce d traced by the launch that traces the closest-hit rays of bounce
    d + 1) and PT_OPT_SHADOW_STREAM: every buffer is what the separate passes leave - against the oracle in reference
    order, against the separate passes byte for byte in the near-first order."""
    W, H, NR, NF = 160, 88, 5, 5
    osc, fr = _setup(gpu_ctx, monkey, W, H, NR)
    gpu_ctx.set_option(core.OPT_TRAVERSAL, trav)
    gpu_ctx.set_option(core.OPT_FRAMES_IN_FLIGHT, 3)
    gpu_ctx.set_option(core.OPT_SHADOW_STREAM, shadow_stream)
    gpu_ctx.set_option(core.OPT_FRAME_TRAIN, 0)            # these options are about frames that run one per pipe
    try:
        out = {}
        for merge in (0, 1):
            gpu_ctx.set_option(core.OPT_MERGE_PASSES, merge)
            gpu_ctx.resize(W, H, NR)                       # fresh stacks and accumulation
            gpu_ctx.set_constants(_consts(monkey, W, H, NR, 1, {}))
            rc = gpu_ctx.render_samples(1, NF, want_ray_counts=True)
            out[merge] = (gpu_ctx.readback_accumulation(), gpu_ctx.readback_current(), gpu_ctx.readback_raystack()[0], rc)
        for k in range(3):
            assert np.array_equal(np.asarray(out[0][k]).view(np.uint8), np.asarray(out[1][k]).view(np.uint8)), f"buffer {k} differs with merged passes"
        assert np.array_equal(out[0][3], out[1][3])
        if trav == core.TRAVERSAL_FLAT:
            for frame in range(1, NF + 1):
                oracle.render_frame(osc, fr, _consts(monkey, W, H, NR, frame, {}), mode=oracle.MODE_ZERO_NEXT_SLOT | oracle.MODE_KILL_ON_MISS)
            _compare(gpu_ctx, fr, live_slots_only=True)
    finally:
        gpu_ctx.set_option(core.OPT_MERGE_PASSES, 1)
        gpu_ctx.set_option(core.OPT_SHADOW_STREAM, 2)
        gpu_ctx.set_option(core.OPT_FRAME_TRAIN, 1)
        gpu_ctx.set_option(core.OPT_TRAVERSAL, core.TRAVERSAL_FLAT)
        gpu_ctx.set_option(core.OPT_FRAMES_IN_FLIGHT, 2)


@pytest.mark.parametrize("in_flight", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("reseed", [0, 1])
def test_frames_in_flight_equal_sequential(gpu_ctx, oracle_lib, monkey, in_flight, reseed):
    """pt_render_samples with several frames in flight, one per pipe, leaves the ray stack, the current image and the
    accumulation of the same frames rendered one after the other (the oracle does exactly that), and a pt_render_sample
    that follows continues the RNG chain from the batch's last frame."""
    W, H, NR, NF = 160, 88, 4, 7
    osc, fr = _setup(gpu_ctx, monkey, W, H, NR)
    gpu_ctx.set_option(core.OPT_FRAME_TRAIN, 0)
    gpu_ctx.set_option(core.OPT_RESEED, reseed)
    gpu_ctx.set_option(core.OPT_FRAMES_IN_FLIGHT, in_flight)
    mode = oracle.MODE_ZERO_NEXT_SLOT | oracle.MODE_KILL_ON_MISS | (oracle.MODE_RESEED_EVERY_FRAME if reseed else 0)
    c = _consts(monkey, W, H, NR, 1, {})
    gpu_ctx.set_constants(c)
    rc = gpu_ctx.render_samples(1, NF, want_ray_counts=True)
    for frame in range(1, NF + 1):
        c["frameIndex"] = frame
        oracle.render_frame(osc, fr, c, mode=mode)
    _compare(gpu_ctx, fr, live_slots_only=in_flight > 1)
    counts = fr.ray_counts()
    assert int(rc[0::2][:NR].sum()) == int(counts[0]) and int(rc[1::2][:NR].sum()) == int(counts[1])
    gpu_ctx.render_sample(NF + 1)
    c["frameIndex"] = NF + 1
    oracle.render_frame(osc, fr, c, mode=mode)
    _compare(gpu_ctx, fr, live_slots_only=in_flight > 1)
    gpu_ctx.set_option(core.OPT_FRAMES_IN_FLIGHT, 2)
    gpu_ctx.set_option(core.OPT_FRAME_TRAIN, 1)


@pytest.mark.parametrize("shape", [(100, 50, 3, 3), (70, 44, 2, 4), (33, 9, 5, 5), (64, 40, 1, 2), (48, 24, 12, 12)])
def test_frame_train_odd_shapes(gpu_ctx, monkey, shape):
    """Widths that are no multiple of the launch pass's 32 x 8 tiles, heights that do not split into equal bands (or are too
    small for the bands asked for), fewer bounces than slots: the train's accumulation, current image and per-depth ray
    counts are those of the per-frame pipeline, byte for byte, for every number of bands."""
    W, H, NR, slots = shape
    helpers.upload(gpu_ctx, monkey, pt.default_lights(), pt.load_blue_noise())
    out = {}
    for train, parts in ((0, 1), (1, 1), (1, 2), (1, 3), (1, 4)):
        gpu_ctx.resize(W, H, slots)
        for o, v in ((core.OPT_TRAVERSAL, core.TRAVERSAL_FLAT_ORDERED), (core.OPT_STALE_SLOTS, 0), (core.OPT_RESEED, 0), (core.OPT_FRAMES_IN_FLIGHT, 2),
                     (core.OPT_FRAME_TRAIN, train), (core.OPT_TRAIN_PARTS, parts), (core.OPT_MERGE_PASSES, 1), (core.OPT_SHADOW_STREAM, 2)):
            gpu_ctx.set_option(o, v)
        gpu_ctx.set_constants(_consts(monkey, W, H, NR, 1, {}))
        rc = gpu_ctx.render_samples(1, 9, want_ray_counts=True)
        rc2 = gpu_ctx.render_samples(10, 2, want_ray_counts=True)
        out[(train, parts)] = (gpu_ctx.readback_accumulation(), gpu_ctx.readback_current(), rc, rc2)
    ref = out[(0, 1)]
    assert int(ref[2][0]) == 9 * W * H and (ref[0][..., 3] == 11).all()
    for key, got in out.items():
        for k in range(4):
            assert np.array_equal(np.asarray(ref[k]), np.asarray(got[k])), f"train, bands {key}: buffer {k} differs from the per-frame pipeline's"
    gpu_ctx.set_option(core.OPT_TRAIN_PARTS, 2)
    gpu_ctx.set_option(core.OPT_TRAVERSAL, core.TRAVERSAL_FLAT)


@pytest.mark.parametrize("trav", [core.TRAVERSAL_FLAT, core.TRAVERSAL_FLAT_ORDERED, core.TRAVERSAL_CERTIFIED])
@pytest.mark.parametrize("reseed,parts", [(0, 1), (0, 2), (1, 3), (0, 4)])
def test_frame_train_equals_sequential(gpu_ctx, oracle_lib, monkey, trav, reseed, parts):
    """The frame train (all frames of a call stepped together, one traversal launch and one shade pass per step): the
    buffers are those of the same frames rendered one after the other, for calls shorter than, as long as and longer than the
    ring (numRays + 1 stacks), with the ring growing between calls; per-depth ray counts equal the per-pipe pipeline's;
    single frames, dispatch-site calls and the per-pipe pipeline continue the RNG chain from the train's last frame and
    the train continues theirs."""
    W, H, NR = 160, 88, 4
    osc, fr = _setup(gpu_ctx, monkey, W, H, NR)
    gpu_ctx.set_option(core.OPT_TRAVERSAL, trav)
    gpu_ctx.set_option(core.OPT_RESEED, reseed)
    gpu_ctx.set_option(core.OPT_TRAIN_PARTS, parts)        # bands of rows, each a train of its own
    mode = oracle.MODE_ZERO_NEXT_SLOT | oracle.MODE_KILL_ON_MISS | (oracle.MODE_RESEED_EVERY_FRAME if reseed else 0)
    near_first = trav == core.TRAVERSAL_FLAT_ORDERED
    c = _consts(monkey, W, H, NR, 1, {})
    gpu_ctx.set_constants(c)
    frame = 1

    def oracle_frames(n):
        nonlocal frame
        for _ in range(n):
            c["frameIndex"] = frame
            oracle.render_frame(osc, fr, c, mode=mode)
            frame += 1

    def check():
        _compare(gpu_ctx, fr, strict=not near_first, live_slots_only=True)

    counts = {}
    for train in (1, 0):                                   # the same 11 frames by train and by pipes: equal ray counts per depth
        gpu_ctx.set_option(core.OPT_FRAME_TRAIN, train)
        gpu_ctx.resize(W, H, NR)
        gpu_ctx.set_constants(c)
        counts[train] = gpu_ctx.render_samples(1, 11, want_ray_counts=True)
        if train:
            first = (gpu_ctx.readback_accumulation(), gpu_ctx.readback_current())
        else:
            assert np.array_equal(first[0], gpu_ctx.readback_accumulation()) and np.array_equal(first[1], gpu_ctx.readback_current())
    assert np.array_equal(counts[0], counts[1]) and int(counts[1][0]) == 11 * W * H
    gpu_ctx.set_option(core.OPT_FRAME_TRAIN, 1)
    gpu_ctx.resize(W, H, NR)
    gpu_ctx.set_constants(c)
    for n in (1, 2, 1, NR + 1, 3, 2 * NR + 3):                # the ring grows 1 -> 2 -> 5 stacks, then frames lap it
        rc = gpu_ctx.render_samples(frame, n, want_ray_counts=True)
        before = fr.ray_counts().copy()
        oracle_frames(n)
        check()
        if not near_first:
            after = fr.ray_counts()
            assert int(rc[0::2][:NR].sum()) == int(after[0] - before[0]) and int(rc[1::2][:NR].sum()) == int(after[1] - before[1])
    # dispatch sites one by one continue from the train's last frame ...
    c["frameIndex"] = frame
    gpu_ctx.set_constants(c)
    gpu_ctx.launch()
    for d in range(NR):
        gpu_ctx.trace(d)
    gpu_ctx.resolve_accumulate()
    oracle_frames(1)
    check()
    # ... the train from theirs, the per-pipe pipeline from the train's, and the train from the pipes'
    gpu_ctx.render_samples(frame, 3); oracle_frames(3); check()
    gpu_ctx.set_option(core.OPT_FRAME_TRAIN, 0)
    gpu_ctx.set_option(core.OPT_FRAMES_IN_FLIGHT, 3)
    gpu_ctx.render_samples(frame, 4); oracle_frames(4); check()
    gpu_ctx.set_option(core.OPT_FRAME_TRAIN, 1)
    gpu_ctx.set_option(core.OPT_TRAIN_PARTS, 1 if parts > 1 else 2)      # another number of bands: the ring is rebuilt, its head kept
    gpu_ctx.render_samples(frame, 6); oracle_frames(6); check()
    # a train that may not have the memory it needs (PT_OPT_TRAIN_MAX_MB; the same path serves a device that is full): the
    # frames run one per pipe, from the RNG states the train left, and the train picks up again when it may
    gpu_ctx.set_option(core.OPT_TRAIN_MAX_MB, 1)
    gpu_ctx.render_samples(frame, 5); oracle_frames(5); check()
    gpu_ctx.render_sample(frame); oracle_frames(1); check()
    gpu_ctx.set_option(core.OPT_TRAIN_MAX_MB, 0)
    gpu_ctx.render_samples(frame, 5); oracle_frames(5); check()
    # a cleared accumulation reseeds exactly the first frame of the next call (launch.comp:49-50)
    gpu_ctx.clear_accumulation(); fr.clear_accumulation()
    gpu_ctx.render_samples(frame, 3); oracle_frames(3); check()
    assert (gpu_ctx.readback_accumulation()[..., 3] == 3).all()
    gpu_ctx.set_option(core.OPT_FRAMES_IN_FLIGHT, 2)
    gpu_ctx.set_option(core.OPT_TRAVERSAL, core.TRAVERSAL_FLAT)
    gpu_ctx.set_option(core.OPT_RESEED, 0)
    gpu_ctx.set_option(core.OPT_TRAIN_PARTS, 2)


@pytest.mark.parametrize("trav,parts", [(core.TRAVERSAL_FLAT, 2), (core.TRAVERSAL_CERTIFIED, 3), (core.TRAVERSAL_FLAT, 1)])
def test_kept_train_equals_sequential(gpu_ctx, oracle_lib, monkey, trav, parts):
    """PT_OPT_TRAIN_KEEP: render calls return with frames still in the train.  One frame per call, as the reference's
    render thread asks for them: the accumulation read between calls is the oracle's after the frames that have left the
    train (it lags numRays frames behind), pt_render_flush and every call that must see finished frames (clear, ray-stack
    read, changed constants, options, a call that wants ray counts, the dispatch sites) bring every buffer to what
    frame-by-frame rendering leaves."""
    W, H, NR = 160, 88, 4
    osc, fr = _setup(gpu_ctx, monkey, W, H, NR)
    gpu_ctx.set_option(core.OPT_TRAVERSAL, trav)
    gpu_ctx.set_option(core.OPT_TRAIN_PARTS, parts)
    gpu_ctx.set_option(core.OPT_TRAIN_KEEP, 1)
    mode = oracle.MODE_ZERO_NEXT_SLOT | oracle.MODE_KILL_ON_MISS
    c = _consts(monkey, W, H, NR, 1, {})
    frame = 1                 # next frame index, the same on both sides
    snaps = [fr.accumulation.copy()]          # the oracle's accumulation after 0, 1, 2, ... frames since the last clear

    def both(n, per_call=1):
        nonlocal frame
        for first in range(frame, frame + n, per_call):
            k = min(per_call, frame + n - first)
            c["frameIndex"] = first
            gpu_ctx.set_constants(c)                  # a new frameIndex alone leaves the train as it is
            gpu_ctx.render_samples(first, k)
            for i in range(k):
                c["frameIndex"] = first + i
                oracle.render_frame(osc, fr, c, mode=mode)
                snaps.append(fr.accumulation.copy())
        frame += n

    both(2)                                           # fewer calls than the path is deep: nothing has left the train
    assert (gpu_ctx.readback_accumulation() == 0).all()
    both(5)
    acc = gpu_ctx.readback_accumulation()             # 7 frames entered, 3 left
    assert (acc[..., 3] == 7 - NR).all() and np.array_equal(acc, snaps[7 - NR])
    gpu_ctx.synchronize()
    both(3, per_call=2)                               # calls of two frames and of one
    acc = gpu_ctx.readback_accumulation()
    assert np.array_equal(acc, snaps[10 - NR])
    gpu_ctx.render_flush()
    _compare(gpu_ctx, fr, live_slots_only=True)
    gpu_ctx.render_flush()                            # nothing left to do
    both(6)
    _compare(gpu_ctx, fr, live_slots_only=True)       # the ray-stack read finishes the frames first
    both(3)
    gpu_ctx.clear_accumulation(); fr.clear_accumulation(); snaps[:] = [fr.accumulation.copy()]
    both(NR + 2)
    acc = gpu_ctx.readback_accumulation()
    assert np.array_equal(acc, snaps[2])
    # a call that wants ray counts counts its own frames only, and returns with them finished
    before = fr.ray_counts().copy()
    rc = gpu_ctx.render_samples(frame, 3, want_ray_counts=True)
    for i in range(3):
        c["frameIndex"] = frame + i
        oracle.render_frame(osc, fr, c, mode=mode)
    frame += 3
    after = fr.ray_counts()
    assert int(rc[0::2][:NR].sum()) == int(after[0] - before[0]) and int(rc[1::2][:NR].sum()) == int(after[1] - before[1])
    _compare(gpu_ctx, fr, live_slots_only=True)
    # constants that differ in more than the frame index: the frames in flight are finished under the old ones
    both(3)
    c2 = _consts(monkey, W, H, NR, frame, dict(orbit_angle=0.4))
    gpu_ctx.set_constants(c2)
    gpu_ctx.render_samples(frame, 2)
    for i in range(2):
        c2["frameIndex"] = frame + i
        oracle.render_frame(osc, fr, c2, mode=mode)
    frame += 2
    # the dispatch sites take over from a train with frames in it
    c2["frameIndex"] = frame
    gpu_ctx.set_constants(c2)
    gpu_ctx.launch()
    for d in range(NR):
        gpu_ctx.trace(d)
    gpu_ctx.resolve_accumulate()
    oracle.render_frame(osc, fr, c2, mode=mode)
    frame += 1
    _compare(gpu_ctx, fr, live_slots_only=True)
    # an option change with frames in flight, then a resize
    gpu_ctx.render_samples(frame, 3)
    for i in range(3):
        c2["frameIndex"] = frame + i
        oracle.render_frame(osc, fr, c2, mode=mode)
    frame += 3
    gpu_ctx.set_option(core.OPT_TRAIN_KEEP, 0)
    assert np.array_equal(gpu_ctx.readback_accumulation(), fr.accumulation)
    gpu_ctx.set_option(core.OPT_TRAIN_KEEP, 1)
    gpu_ctx.render_samples(frame, 2)
    gpu_ctx.resize(W, H, NR)                          # frames in flight are finished into buffers that are then replaced
    assert (gpu_ctx.readback_accumulation() == 0).all()
    gpu_ctx.set_option(core.OPT_TRAIN_KEEP, 0)
    gpu_ctx.set_option(core.OPT_TRAIN_PARTS, 2)
    gpu_ctx.set_option(core.OPT_TRAVERSAL, core.TRAVERSAL_FLAT)


def test_kept_train_scene_swap_two_contexts_and_destroy(oracle_lib, monkey, bunny):
    """Frames in a kept train when the scene is replaced are finished on the scene they entered with; two contexts on one
    device keep trains of their own; a context destroyed with frames in flight goes quietly."""
    W, H, NR = 96, 64, 3
    mode = oracle.MODE_ZERO_NEXT_SLOT | oracle.MODE_KILL_ON_MISS
    a, b = pt.PathTracer(0), pt.PathTracer(0)
    try:
        osc_a, fr_a = _setup(a, monkey, W, H, NR)
        osc_b, fr_b = _setup(b, bunny, W, H, NR)
        for ctx in (a, b):
            ctx.set_option(core.OPT_TRAIN_KEEP, 1)
        ca, cb = _consts(monkey, W, H, NR, 1, {}), _consts(bunny, W, H, NR, 1, BUNNY_CAM)
        a.set_constants(ca); b.set_constants(cb)
        for frame in range(1, 6):                       # alternating calls, one frame each
            a.render_sample(frame); b.render_sample(frame)
            ca["frameIndex"] = cb["frameIndex"] = frame
            oracle.render_frame(osc_a, fr_a, ca, mode=mode)
            oracle.render_frame(osc_b, fr_b, cb, mode=mode)
        # a gets b's scene with three frames in flight: they finish on the monkey
        helpers.upload(a, bunny, pt.default_lights(), pt.load_blue_noise())
        assert np.array_equal(a.readback_accumulation(), fr_a.accumulation)
        a.set_constants(_consts(bunny, W, H, NR, 6, BUNNY_CAM))
        a.render_samples(6, 2)
        osc_a2 = helpers.oracle_scene(bunny, pt.default_lights(), pt.load_blue_noise())
        for frame in (6, 7):
            oracle.render_frame(osc_a2, fr_a, _consts(bunny, W, H, NR, frame, BUNNY_CAM), mode=mode)
        a.render_flush()
        assert np.array_equal(a.readback_accumulation(), fr_a.accumulation)
        b.render_flush()
        _compare(b, fr_b, live_slots_only=True)
        b.render_samples(6, 4)                          # left in flight when the context goes
    finally:
        b.close()
        a.close()


def test_cornell_box_512_four_bounces(gpu_ctx, oracle_lib):
    """The plumbing configuration: Cornell box (stand-in geometry, 34 triangles) at 512 x 512 with 4 bounces from the
    reference's default camera; every buffer equals the oracle's after two frames."""
    pos, aux, idx = pt.scenes.cornell_box()
    nodes, ridx, _ = pt.build_bvh(pos, idx, 1)
    model = dict(positions=pos, aux=aux, indices=ridx, nodes=nodes)
    W, H, NR = 512, 512, 4
    osc, fr = _setup(gpu_ctx, model, W, H, NR)
    cam = dict(focus_center=(0.0, 1.0, 0.0), orbit_height=1.0, orbit_radius=3.4, orbit_angle=0.0)
    for frame in (1, 2):
        c = _consts(model, W, H, NR, frame, cam)
        gpu_ctx.set_constants(c)
        gpu_ctx.render_sample(frame)
        oracle.render_frame(osc, fr, c, mode=oracle.MODE_ZERO_NEXT_SLOT | oracle.MODE_KILL_ON_MISS)
        _compare(gpu_ctx, fr)
    acc = gpu_ctx.readback_accumulation()
    assert (acc[..., :3].sum(axis=2) > 0).mean() > 0.3          # the box is in view and lit
    # this small tree also exercises the walk that runs off the end of the node array and arms origBvhId
    # (intersections.glsl:109-113): the shadow rays of such hits skip the leaf they start on
    rays = helpers.random_rays(nodes[0]["a"], nodes[0]["b"], 20000, seed=1)
    _, ran = osc.trace(rays, want_ranoff=True)
    assert (ran != 0).mean() > 0.01


def test_frames_in_flight_survive_resize_clear_and_speculative(gpu_ctx, oracle_lib, monkey):
    """The extra frame pipes are rebuilt after pt_resize, a cleared accumulation reseeds exactly the first frame of the
    next batch (launch.comp:49-50), an empty batch does nothing, and the subgroup-vote build of depth 0 works with
    frames in flight."""
    gpu_ctx.set_option(core.OPT_FRAMES_IN_FLIGHT, 3)
    mode = oracle.MODE_ZERO_NEXT_SLOT | oracle.MODE_KILL_ON_MISS
    for (W, H, NR) in ((96, 64, 3), (128, 72, 2)):
        osc, fr = _setup(gpu_ctx, monkey, W, H, NR)
        gpu_ctx.set_option(core.OPT_FRAMES_IN_FLIGHT, 3)
        c = _consts(monkey, W, H, NR, 1, {})
        gpu_ctx.set_constants(c)
        gpu_ctx.render_samples(1, 0)
        assert (gpu_ctx.readback_accumulation() == 0).all()
        gpu_ctx.render_samples(1, 4, speculative_depth0=True)
        for frame in range(1, 5):
            c["frameIndex"] = frame
            oracle.render_frame(osc, fr, c, subgroup0=64, mode=mode)
        _compare(gpu_ctx, fr, live_slots_only=True)
        gpu_ctx.clear_accumulation()
        fr.clear_accumulation()
        gpu_ctx.render_samples(5, 5)
        for frame in range(5, 10):
            c["frameIndex"] = frame
            oracle.render_frame(osc, fr, c, mode=mode)
        _compare(gpu_ctx, fr, live_slots_only=True)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_soup_frames_match_oracle(gpu_ctx, oracle_lib, seed):
    """Whole frames on random triangle soups with random vertex colours, transparent vertices (alpha 0: the refraction
    branch, trace.comp:136-149) and normals that are neither unit length nor consistent with the winding: every buffer
    equals the oracle's."""
    rng = np.random.default_rng(seed)
    n = 96
    pos = np.ones((3 * n, 4), np.float32)
    pos[:, :3] = rng.uniform(-1, 1, (3 * n, 3)).astype(np.float32)
    pos[:, 1] += 1.0
    idx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    aux = np.zeros(3 * n, dtype=pt.AUX_DT)
    aux["normal"] = rng.normal(0, 1, (3 * n, 3)).astype(np.float32)
    aux["color"] = rng.integers(0, 256, (3 * n, 4), dtype=np.uint8)
    aux["color"][:, 3] = np.where(rng.random(3 * n) < 0.3, 0, 255)
    nodes, ridx, _ = pt.build_bvh(pos, idx, 1)
    model = dict(positions=pos, aux=aux, indices=ridx, nodes=nodes)
    W, H, NR = 96, 64, 5
    osc, fr = _setup(gpu_ctx, model, W, H, NR)
    cam = dict(focus_center=(0.0, 1.0, 0.0), orbit_height=1.2, orbit_radius=3.0, orbit_angle=0.3 * seed)
    for frame in (1, 2, 3):
        c = _consts(model, W, H, NR, frame, cam)
        gpu_ctx.set_constants(c)
        gpu_ctx.render_sample(frame)
        oracle.render_frame(osc, fr, c, mode=oracle.MODE_ZERO_NEXT_SLOT | oracle.MODE_KILL_ON_MISS)
        _compare(gpu_ctx, fr)
    assert (gpu_ctx.readback_accumulation()[..., :3] > 0).mean() > 0.05


def test_two_contexts_and_scene_swap(oracle_lib, monkey, bunny):
    """Two contexts on one device queue frames alternately, and one of them is given another scene half way: each
    context's buffers stay the oracle's for its own scene, nothing leaks from one to the other."""
    W, H, NR = 80, 56, 3
    a, b = pt.PathTracer(0), pt.PathTracer(0)
    try:
        osc_a, fr_a = _setup(a, monkey, W, H, NR)
        osc_b, fr_b = _setup(b, bunny, W, H, NR)
        for frame in (1, 2):
            for ctx, osc, fr, model, cam in ((a, osc_a, fr_a, monkey, {}), (b, osc_b, fr_b, bunny, BUNNY_CAM)):
                c = _consts(model, W, H, NR, frame, cam)
                ctx.set_constants(c)
                ctx.render_samples(frame, 1)               # queued, not waited for
                oracle.render_frame(osc, fr, c, mode=oracle.MODE_ZERO_NEXT_SLOT | oracle.MODE_KILL_ON_MISS)
        _compare(a, fr_a)
        _compare(b, fr_b)
        # context a changes scene (frame resources stay): a fresh accumulation, the new scene's images
        helpers.upload(a, bunny, pt.default_lights(), pt.load_blue_noise())
        a.clear_accumulation()
        fr_a2 = oracle.Frame(W, H, NR)
        fr_a2.stack[:] = fr_a.stack                        # the ray stack keeps what the last frame left (RNG states)
        for frame in (3, 4):
            c = _consts(bunny, W, H, NR, frame, BUNNY_CAM)
            a.set_constants(c)
            a.render_samples(frame, 1)
            oracle.render_frame(osc_b, fr_a2, c, mode=oracle.MODE_ZERO_NEXT_SLOT | oracle.MODE_KILL_ON_MISS)
        _compare(a, fr_a2)
        _compare(b, fr_b)                                  # untouched by all that
    finally:
        a.close(); b.close()


def test_uploaded_ray_stack_and_jobs_are_what_gets_traced(gpu_ctx, oracle_lib, monkey):
    """pt_upload_raystack: the host edits the ray stack and the job list between dispatch sites (a third of the jobs
    killed, some rays turned round), and the passes that follow work on exactly that; a job that names a slot outside
    the ray stack is refused."""
    W, H, NR = 96, 64, 3
    osc, fr = _setup(gpu_ctx, monkey, W, H, NR)
    c = _consts(monkey, W, H, NR, 1, {})
    gpu_ctx.set_constants(c)
    gpu_ctx.launch()
    oracle.launch(osc, fr, c)
    slots, jobs = gpu_ctx.readback_raystack()
    jobs = jobs.copy(); slots = slots.copy()
    jobs[::3] = 0xFFFFFFFF
    turned = np.arange(0, W * H, 7) * NR
    slots["rayDirection"][turned] = -slots["rayDirection"][turned]
    gpu_ctx.upload_raystack(slots, jobs)
    fr.stack[:] = slots; fr.jobs[:] = jobs
    for d in range(NR):
        gpu_ctx.trace(d)
        oracle.trace(osc, fr, c, mode=oracle.MODE_ZERO_NEXT_SLOT | oracle.MODE_KILL_ON_MISS)
    got, gjobs = gpu_ctx.readback_raystack()
    assert np.array_equal(got.view(np.uint8), fr.stack.view(np.uint8))
    assert np.array_equal(np.sort(gjobs[gjobs != 0xFFFFFFFF]), np.sort(fr.jobs[fr.jobs != 0xFFFFFFFF]))
    gpu_ctx.resolve_accumulate()
    oracle.composite(fr, c)
    _compare(gpu_ctx, fr)
    bad = jobs.copy(); bad[5] = W * H * NR
    with pytest.raises(pt.PtError):
        gpu_ctx.upload_raystack(None, bad)
    gpu_ctx.upload_raystack(None, jobs)          # the limit itself minus one is fine
    ok = jobs.copy(); ok[5] = W * H * NR - 1
    gpu_ctx.upload_raystack(None, ok)


def test_contexts_give_their_device_memory_back(bunny):
    """Create, use (scene, frame pipes, ordering scratch, spill areas) and destroy a context several times: the free
    device memory afterwards is what it was before, to within what the runtime keeps for itself."""
    import torch
    torch.cuda.synchronize()

    def cycle():
        ctx = pt.PathTracer(0)
        try:
            helpers.upload(ctx, bunny, pt.default_lights(), pt.load_blue_noise())
            ctx.resize(640, 480, 4)
            ctx.set_constants(pt.fill_constants(640, 480, 4, bunny["indices"].shape[0], bunny["nodes"].shape[0], **BUNNY_CAM))
            ctx.set_option(core.OPT_FRAMES_IN_FLIGHT, 4)
            ctx.render_samples(1, 8)
            ctx.resize(320, 240, 2)                          # frees and rebuilds the per-resolution buffers
            ctx.set_constants(pt.fill_constants(320, 240, 2, bunny["indices"].shape[0], bunny["nodes"].shape[0], **BUNNY_CAM))
            ctx.render_samples(9, 4)
            ctx.set_option(core.OPT_SORT_RAYS, 1)
            rays = helpers.random_rays(bunny["nodes"][0]["a"], bunny["nodes"][0]["b"], 50_000, seed=3)
            ctx.trace_rays(rays)
            helpers.upload(ctx, bunny, pt.default_lights(), pt.load_blue_noise())      # a second scene replaces the first
            ctx.synchronize()
        finally:
            ctx.close()

    cycle()                                                  # first use: the runtime's own one-time allocations
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(4):
        cycle()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 32 << 20, f"{(free0 - free1) / 2**20:.1f} MiB of device memory not returned"


def test_certified_walk_frames(gpu_ctx, oracle_lib, monkey):
    """PT_OPT_TRAVERSAL = 3 in the frame pipeline: closest-hit passes by the certified walk plus the second walk of the
    jobs it lists, shadow passes near child first.  Every buffer equals the oracle's: several frames in flight, dispatch
    site by dispatch site, and on a random soup with glass."""
    W, H, NR = 160, 96, 5
    osc, fr = _setup(gpu_ctx, monkey, W, H, NR)
    gpu_ctx.set_option(core.OPT_TRAVERSAL, core.TRAVERSAL_CERTIFIED)
    gpu_ctx.set_option(core.OPT_FRAMES_IN_FLIGHT, 3)
    try:
        c0 = _consts(monkey, W, H, NR, 1, {})
        gpu_ctx.set_constants(c0)
        gpu_ctx.render_samples(1, 5)
        for frame in range(1, 6):
            oracle.render_frame(osc, fr, _consts(monkey, W, H, NR, frame, {}), mode=oracle.MODE_ZERO_NEXT_SLOT | oracle.MODE_KILL_ON_MISS)
        _compare(gpu_ctx, fr, live_slots_only=True)
        # dispatch sites one by one (the passes then run one after another on one stream and rewrite jobs in place)
        osc, fr = _setup(gpu_ctx, monkey, W, H, NR)
        gpu_ctx.set_option(core.OPT_TRAVERSAL, core.TRAVERSAL_CERTIFIED)
        c = _consts(monkey, W, H, NR, 1, {})
        gpu_ctx.set_constants(c)
        gpu_ctx.launch(); oracle.launch(osc, fr, c)
        for d in range(NR):
            gpu_ctx.trace(d)
            oracle.trace(osc, fr, c, mode=oracle.MODE_ZERO_NEXT_SLOT | oracle.MODE_KILL_ON_MISS)
            slots, _ = gpu_ctx.readback_raystack()
            assert np.array_equal(slots.view(np.uint8), fr.stack.view(np.uint8)), f"depth {d}"
        gpu_ctx.resolve_accumulate(); oracle.composite(fr, c)
        _compare(gpu_ctx, fr)
        # a soup: random colours, glass, normals unrelated to the winding
        rng = np.random.default_rng(9)
        n = 200
        pos = np.ones((3 * n, 4), np.float32)
        pos[:, :3] = rng.uniform(-1, 1, (3 * n, 3)).astype(np.float32); pos[:, 1] += 1.0
        idx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
        aux = np.zeros(3 * n, dtype=pt.AUX_DT)
        aux["normal"] = rng.normal(0, 1, (3 * n, 3)).astype(np.float32)
        aux["color"] = rng.integers(0, 256, (3 * n, 4), dtype=np.uint8)
        aux["color"][:, 3] = np.where(rng.random(3 * n) < 0.3, 0, 255)
        nodes, ridx, _ = pt.build_bvh(pos, idx, 1)
        model = dict(positions=pos, aux=aux, indices=ridx, nodes=nodes)
        osc, fr = _setup(gpu_ctx, model, W, H, NR)
        gpu_ctx.set_option(core.OPT_TRAVERSAL, core.TRAVERSAL_CERTIFIED)
        cam = dict(focus_center=(0.0, 1.0, 0.0), orbit_height=1.2, orbit_radius=3.0, orbit_angle=0.7)
        for frame in (1, 2, 3):
            c = _consts(model, W, H, NR, frame, cam)
            gpu_ctx.set_constants(c)
            gpu_ctx.render_sample(frame)
            oracle.render_frame(osc, fr, c, mode=oracle.MODE_ZERO_NEXT_SLOT | oracle.MODE_KILL_ON_MISS)
            _compare(gpu_ctx, fr)
    finally:
        gpu_ctx.set_option(core.OPT_TRAVERSAL, core.TRAVERSAL_FLAT)
        gpu_ctx.set_option(core.OPT_FRAMES_IN_FLIGHT, 2)


def test_cpp_host_renders_what_the_python_host_renders(gpu_ctx, bunny, tmp_path):
    """examples/pt_host.cpp - a C++ program that links libptcore.so and nothing else, and follows the reference's render
    thread through the C ABI (load, BuildBVH, upload, per frame launch / trace / raysort / composite) - leaves the
    accumulation this Python host leaves for the same frames, byte for byte (compared by hash), through the dispatch sites
    through one pt_render_samples call and through one call per frame with the frame train kept between the calls; its frame
    dump and picture are there."""
    import os, re, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "pt_host")
    if not os.path.exists(exe):
        pytest.fail("examples/pt_host is not built (pathtracer_amd/build.py builds it with the library)")
    W, H, NR, NF = 320, 184, 4, 6
    model = os.path.join(root, "tests", "golden", "models", "CBbunny.ply")
    noise = os.path.join(root, "pathtracer_amd", "data", "blue_noise_u16le.bin")
    hashes = {}
    for mode in ("sites", "batch", "kept"):
        out = tmp_path / mode
        out.mkdir()
        r = subprocess.run([exe, model, noise, str(out), str(W), str(H), str(NR), str(NF), mode], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        m = re.search(r"accumulation fnv1a ([0-9a-f]{16}), samples per pixel (\d+)", r.stdout)
        assert m and int(m.group(2)) == NF, r.stdout
        hashes[mode] = m.group(1)
        assert (out / f"RayStackBuffer_{NF}.bin.z").exists() and (out / "pt_host.ppm").stat().st_size > W * H * 3
    assert hashes["sites"] == hashes["batch"] == hashes["kept"]
    _setup(gpu_ctx, bunny, W, H, NR)
    gpu_ctx.set_option(core.OPT_TRAVERSAL, core.TRAVERSAL_DEFAULT)      # the library's default, which the C++ host runs with
    for frame in range(1, NF + 1):
        gpu_ctx.set_constants(_consts(bunny, W, H, NR, frame, BUNNY_CAM))
        gpu_ctx.render_sample(frame)
    acc = gpu_ctx.readback_accumulation()
    h = 1469598103934665603
    # FNV-1a over the bytes, as the C++ program computes it (vectorised: the hash is sequential, so do it in Python ints over chunks)
    data = acc.tobytes()
    for b in data:
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert f"{h:016x}" == hashes["sites"]
    gpu_ctx.set_option(core.OPT_TRAVERSAL, core.TRAVERSAL_FLAT)


def test_frame_train_random_configurations(bunny, monkey):
    """Randomised differential test (fixed seed; scripts/exp_fuzz_frames.py is the long-running twin): random image sizes,
    path depths, slot counts, traversals, record layouts, reseeding, numbers of bands and sequences of call lengths with a clear
    in between - the frame train leaves the accumulation, the current image, the per-depth ray counts and the live part of the
    ray stack that the per-frame pipeline leaves for the same calls, and so does the train kept between calls
    (PT_OPT_TRAIN_KEEP) with every call cut into random pieces and reads that do not empty it in between."""
    rng = np.random.default_rng(20261111)
    for case in range(24):
        model = (bunny, monkey)[int(rng.integers(2))]
        W, H = int(rng.integers(8, 200)), int(rng.integers(8, 120))
        slots = int(rng.integers(1, 10)); NR = int(rng.integers(1, slots + 1))
        trav = int(rng.choice([1, 2, 3])); layout = int(rng.choice([0, 0, 1])); reseed = int(rng.integers(2)); parts = int(rng.integers(1, 5))
        calls = [int(v) for v in rng.integers(1, 2 * NR + 4, size=int(rng.integers(1, 5)))]
        in_flight = int(rng.integers(1, 5))
        desc = f"case {case}: {W}x{H} slots {slots} numRays {NR} traversal {trav} layout {layout} reseed {reseed} bands {parts} calls {calls}"
        out = {}
        rng2 = np.random.default_rng(case)
        for train in (1, 0, 2):               # 2: the train kept between calls, every call cut into random pieces
            ctx = pt.PathTracer(0)
            try:
                ctx.set_option(core.OPT_RECORD_LAYOUT, layout)
                helpers.upload(ctx, model, pt.default_lights(), pt.load_blue_noise())
                ctx.resize(W, H, slots)
                for o, v in ((core.OPT_TRAVERSAL, trav), (core.OPT_RESEED, reseed), (core.OPT_FRAME_TRAIN, min(train, 1)), (core.OPT_TRAIN_PARTS, parts),
                             (core.OPT_FRAMES_IN_FLIGHT, in_flight), (core.OPT_TRAIN_KEEP, int(train == 2))):
                    ctx.set_option(o, v)
                ctx.set_constants(_consts(model, W, H, NR, 1, BUNNY_CAM))
                frame, rcs = 1, []
                for k, n in enumerate(calls):
                    if train == 2:
                        left = n
                        while left:
                            m = int(rng2.integers(1, left + 1))
                            ctx.set_constants(_consts(model, W, H, NR, frame, BUNNY_CAM))      # a new frameIndex alone leaves the train alone
                            ctx.render_samples(frame, m); frame += m; left -= m
                            (ctx.readback_accumulation, ctx.synchronize, ctx.readback_current, lambda: None)[int(rng2.integers(4))]()
                        rcs.append(out[1][2][2 * NR * k:2 * NR * (k + 1)])        # (asking for ray counts would empty the train)
                    else:
                        rcs.append(ctx.render_samples(frame, n, want_ray_counts=True)[:2 * NR].copy()); frame += n
                    if k == 1:
                        ctx.clear_accumulation()           # the next call's first frame reseeds from its jitter
                stack, _ = ctx.readback_raystack()
                out[train] = (ctx.readback_accumulation(), ctx.readback_current(), np.concatenate(rcs),
                              stack[:W * H * NR].reshape(W * H, NR))          # a pixel's slots lie numRays apart
            finally:
                ctx.close()
        b = out[0]
        prob = b[3]["prob"].view(np.float16).astype(np.float32)
        live = np.cumprod(np.concatenate([np.ones((prob.shape[0], 1), bool), prob[:, :-1] > 0], axis=1), axis=1).astype(bool) & (prob > 0)
        for a, how in ((out[1], "train"), (out[2], "kept train")):
            for k, what in enumerate(("accumulation", "current image", "ray counts")):
                assert np.array_equal(np.asarray(a[k]).view(np.uint8), np.asarray(b[k]).view(np.uint8)), f"{desc}: {what} differs ({how})"
            assert np.array_equal(a[3][live].view(np.uint8), b[3][live].view(np.uint8)), f"{desc}: ray stack differs in a live slot ({how})"


@pytest.mark.parametrize("ntri", [40, 100])
def test_certified_frames_on_a_comb_tree(gpu_ctx, oracle_lib, ntri):
    """Whether the reference's walk ran off the end of its array is settled inside the certified pass from the boxes of the
    path to the array's last node.  A comb - every interior node has a leaf on the left and the rest of the tree on the
    right - makes that path as long as the tree has triangles and puts a large share of the rays on it: with 40 triangles
    the table of the path's boxes is used, with 100 the path is longer than the table and such rays take the second walk
    as they used to.  Either way every buffer is the oracle's, byte for byte."""
    rng = np.random.default_rng(ntri)
    pos = np.ones((3 * ntri, 4), np.float32)
    centre = rng.uniform(-1, 1, (ntri, 1, 3)); centre[:, :, 1] += 1.0
    pos[:, :3] = (centre + rng.uniform(-0.4, 0.4, (ntri, 3, 3))).reshape(-1, 3).astype(np.float32)
    pos[-3:, :3] = np.array([[-2.5, 0.0, -2.5], [2.5, 0.0, -2.5], [0.0, 0.3, 3.0]], np.float32)      # the last leaf: a floor under everything
    idx = np.arange(3 * ntri, dtype=np.uint32).reshape(ntri, 3)
    aux = np.zeros(3 * ntri, dtype=pt.AUX_DT)
    aux["normal"] = rng.normal(0, 1, (3 * ntri, 3)).astype(np.float32)
    aux["color"] = rng.integers(64, 256, (3 * ntri, 4), dtype=np.uint8)
    aux["color"][:, 3] = 255
    # pre-order comb: interior node at 2k (left child = leaf 2k+1 holding triangle k, right child = node 2k+2), last node a leaf
    N = 2 * ntri - 1
    nodes = np.zeros(N, dtype=pt.NODE_DT)
    pad = np.float32(6e-5)
    tri_lo = pos[:, :3].reshape(ntri, 3, 3).min(axis=1) - pad
    tri_hi = pos[:, :3].reshape(ntri, 3, 3).max(axis=1) + pad
    suffix_lo = np.minimum.accumulate(tri_lo[::-1], axis=0)[::-1]
    suffix_hi = np.maximum.accumulate(tri_hi[::-1], axis=0)[::-1]
    for k in range(ntri - 1):
        nodes[2 * k] = (suffix_lo[k], 0, suffix_hi[k], 2 * k + 2)                     # skip link 0: its subtree ends the array
        nodes[2 * k + 1] = (tri_lo[k], 2 * k + 2, tri_hi[k], -3 * k)
    nodes[N - 1] = (tri_lo[ntri - 1], 0, tri_hi[ntri - 1], -3 * (ntri - 1))
    model = dict(positions=pos, aux=aux, indices=idx, nodes=nodes)
    W, H, NR = 96, 64, 4
    osc, fr = _setup(gpu_ctx, model, W, H, NR)
    cam = dict(focus_center=(0.0, 1.0, 0.0), orbit_height=1.2, orbit_radius=3.5, orbit_angle=0.4)
    try:
        for train in (0, 1):
            gpu_ctx.set_option(core.OPT_TRAVERSAL, core.TRAVERSAL_CERTIFIED)
            gpu_ctx.set_option(core.OPT_FRAME_TRAIN, train)
            gpu_ctx.clear_accumulation(); fr.clear_accumulation()
            for frame in (1, 2, 3):
                c = _consts(model, W, H, NR, frame, cam)
                gpu_ctx.set_constants(c)
                gpu_ctx.render_sample(frame)
                oracle.render_frame(osc, fr, c, mode=oracle.MODE_ZERO_NEXT_SLOT | oracle.MODE_KILL_ON_MISS)
                _compare(gpu_ctx, fr, live_slots_only=bool(train))
        assert (gpu_ctx.readback_accumulation()[..., :3] > 0).mean() > 0.2
    finally:
        gpu_ctx.set_option(core.OPT_TRAVERSAL, core.TRAVERSAL_DEFAULT)
        gpu_ctx.set_option(core.OPT_FRAME_TRAIN, 1)


@pytest.mark.parametrize("doubled", [False, True])
def test_train_second_walk_settings_short_and_long_lists(gpu_ctx, monkey, doubled):
    """The frame train under the certified walk with PT_OPT_SECOND_WALK 0, 1 and 2 (the default, under which a short list is
    walked inside the shade launch and a long one by a launch of its own): every buffer equals what the reference-order walk
    leaves frame by frame.  CBmonkey lists a handful of jobs per step; with every triangle in the scene twice every hit is a
    tie, nearly every job is listed, and the lists are far beyond the length at which the device changes the kind of walk."""
    model = monkey
    if doubled:
        idx = np.concatenate([monkey["indices_in"], monkey["indices_in"]])
        nodes, ridx, _ = pt.build_bvh(monkey["positions"], idx)
        model = dict(positions=monkey["positions"], aux=monkey["aux"], indices=ridx, nodes=nodes)
    W, H, NR, NF = 256, 160, 4, 7
    lights, bn = pt.default_lights(), pt.load_blue_noise()
    helpers.upload(gpu_ctx, model, lights, bn)
    gpu_ctx.resize(W, H, NR)
    c = _consts(model, W, H, NR, 1, {})

    def render(trav, train, second):
        for o, v in ((core.OPT_TRAVERSAL, trav), (core.OPT_FRAME_TRAIN, train), (core.OPT_SECOND_WALK, second), (core.OPT_TRAIN_PARTS, 2),
                     (core.OPT_TRAIN_KEEP, 0), (core.OPT_RESEED, 0), (core.OPT_STALE_SLOTS, 0)):
            gpu_ctx.set_option(o, v)
        gpu_ctx.set_constants(c)
        gpu_ctx.clear_accumulation()
        rc = gpu_ctx.render_samples(1, NF, want_ray_counts=True)
        return gpu_ctx.readback_accumulation(), gpu_ctx.readback_current(), rc

    try:
        ref = render(core.TRAVERSAL_FLAT, 0, 2)
        assert (ref[0][..., 3] == NF).all() and (ref[0][..., :3] > 0).mean() > 0.05
        for second in (0, 1, 2):
            got = render(core.TRAVERSAL_CERTIFIED, 1, second)
            for a, b, what in zip(got, ref, ("accumulation", "current image", "ray counts")):
                assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), f"second walk {second}: {what} differs"
    finally:
        for o, v in ((core.OPT_TRAVERSAL, core.TRAVERSAL_DEFAULT), (core.OPT_FRAME_TRAIN, 1), (core.OPT_SECOND_WALK, 2)):
            gpu_ctx.set_option(o, v)


@pytest.mark.parametrize("trav,parts", [(core.TRAVERSAL_FLAT_ORDERED, 2), (core.TRAVERSAL_CERTIFIED, 3)])
def test_traversal_timing_option_counts_launches_and_changes_nothing(gpu_ctx, monkey, trav, parts):
    """PT_OPT_TIME_TRAVERSAL brackets every traversal launch of the frame train with an event pair: a call of N frames has
    (N + numRays) steps of `parts` launches each, every one with a positive duration, the record starts over with each
    reading, nothing is recorded with the option off, and the images are the same bytes either way."""
    W, H, NR, N = 192, 120, 4, 5
    _setup(gpu_ctx, monkey, W, H, NR)
    try:
        gpu_ctx.set_option(core.OPT_TRAVERSAL, trav)
        gpu_ctx.set_option(core.OPT_TRAIN_PARTS, parts)
        gpu_ctx.set_option(core.OPT_RESEED, 1)            # a frame then depends on its index only: the two renders below are comparable
        gpu_ctx.set_constants(_consts(monkey, W, H, NR, 1, {}))
        gpu_ctx.render_samples(1, N)
        plain = gpu_ctx.readback_accumulation()
        assert len(gpu_ctx.traversal_ms()) == 0
        gpu_ctx.clear_accumulation()
        gpu_ctx.set_option(core.OPT_TIME_TRAVERSAL, 1)
        gpu_ctx.render_samples(1, N)
        ms = gpu_ctx.traversal_ms()
        assert len(ms) == (N + NR) * parts and (ms > 0).all() and ms.max() < 1000.0
        assert len(gpu_ctx.traversal_ms()) == 0
        timed = gpu_ctx.readback_accumulation()
        assert (timed[..., 3] == N).all() and np.array_equal(plain.view(np.uint8), timed.view(np.uint8))
    finally:
        gpu_ctx.set_option(core.OPT_TIME_TRAVERSAL, 0)
        gpu_ctx.set_option(core.OPT_RESEED, 0)
        gpu_ctx.set_option(core.OPT_TRAIN_PARTS, 2)
        gpu_ctx.set_option(core.OPT_TRAVERSAL, core.TRAVERSAL_FLAT)
