"""pt_trace_rays_noalloc, the entry point that allocates nothing: its hits equal pt_trace_rays' byte for byte on two
calls in a row, the second with a smaller batch (the static scratch array then still holds the first call's order,
keys and counters), for closest-hit and any-hit queries, with and without the ordering pass, under the default walk
and under the certified walk (whose list of rays for the second pass lives in the array too); the device's free memory
does not move across the calls; and a batch beyond the array's limit is refused with a clear message.

The file runs last (its name sorts after the other test files) so that a failure here hides no other test."""
import numpy as np
import pytest

import helpers
import pathtracer_amd as pt
from pathtracer_amd import core

pytestmark = pytest.mark.gpu

MAX_RAYS = 1 << 21


@pytest.fixture(scope="module")
def scene():
    pos, aux, idx = pt.scenes.courtyard(400_000, seed=3)
    nodes, ridx, depth = pt.build_bvh(pos, idx, 8)
    assert depth > 12                       # walks spill past the LDS part of the stack: the array's spill area is in use
    return dict(positions=pos, aux=aux, indices=ridx, nodes=nodes)


def _both(ctx, d_rays, n, d_a, d_b, any_hit):
    ctx.trace_rays_device(d_rays, n, d_a, any_hit=any_hit)
    ctx.trace_rays_device_noalloc(d_rays, n, d_b, any_hit=any_hit)
    a, b = d_a.to_host(pt.HIT_DT, n), d_b.to_host(pt.HIT_DT, n)
    assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), f"{int((a != b).sum())} of {n} hits differ (any_hit={any_hit})"
    return a


@pytest.mark.parametrize("traversal", [core.TRAVERSAL_FLAT_ORDERED, core.TRAVERSAL_CERTIFIED, core.TRAVERSAL_FLAT])
def test_noalloc_equals_trace_rays_on_two_calls_in_a_row(scene, traversal):
    import torch
    ctx = pt.PathTracer(0)
    try:
        helpers.upload(ctx, scene)
        ctx.set_option(core.OPT_TRAVERSAL, traversal)
        n1, n2 = MAX_RAYS, 300_017
        d_rays, d_a, d_b = ctx.alloc(n1 * 32), ctx.alloc(n1 * 16), ctx.alloc(n1 * 16)
        ctx.generate_rays(d_rays, n1, seed=99)
        for sort in (2, 1, 0):               # default (orders batches of 2^20 rays or more), always, never
            ctx.set_option(core.OPT_SORT_RAYS, sort)
            for any_hit in (False, True):
                first = _both(ctx, d_rays, n1, d_a, d_b, any_hit)                   # the largest batch the array takes
                second = _both(ctx, d_rays, n2, d_a, d_b, any_hit)                  # right after it, a smaller one
                assert np.array_equal(first[:n2].view(np.uint8), second.view(np.uint8))
                assert 0.3 < (first["i1"] >= 0).mean() < 0.99
        # nothing is allocated or freed on the way: the free device memory is the same before and after
        ctx.set_option(core.OPT_SORT_RAYS, 1)
        ctx.synchronize(); torch.cuda.synchronize()
        free0, _ = torch.cuda.mem_get_info()
        for n in (n1, n2, 1, 64, 65):
            ctx.trace_rays_device_noalloc(d_rays, n, d_b)
        ctx.synchronize()
        free1, _ = torch.cuda.mem_get_info()
        assert free0 == free1
        # limits: one ray more than the array holds is refused, and says why; an empty batch is a no-op
        big = ctx.alloc((MAX_RAYS + 1) * 32)
        try:
            with pytest.raises(pt.PtError, match="2\\^21"):
                ctx._chk(ctx._L.pt_trace_rays_noalloc(ctx._h, big.ptr, MAX_RAYS + 1, 0, d_b.ptr))
        finally:
            big.free()
        ctx.trace_rays_device_noalloc(d_rays, 0, d_b)
        d_rays.free(); d_a.free(); d_b.free()
    finally:
        ctx.close()


def test_noalloc_on_a_second_context_and_scene(scene, bunny):
    """The array belongs to the GPU, not to a context: another context with another scene uses it right after."""
    a, b = pt.PathTracer(0), pt.PathTracer(0)
    try:
        helpers.upload(a, scene)
        helpers.upload(b, bunny)
        for ctx, model, n in ((a, scene, 200_000), (b, bunny, 70_001), (a, scene, 5_000)):
            rays = helpers.random_rays(model["nodes"][0]["a"], model["nodes"][0]["b"], n, seed=n)
            d_rays, d_a, d_b = ctx.alloc(n * 32), ctx.alloc(n * 16), ctx.alloc(n * 16)
            d_rays.from_host(rays)
            ctx.set_option(core.OPT_SORT_RAYS, 1)
            _both(ctx, d_rays, n, d_a, d_b, False)
            ctx.synchronize()                  # calls on one GPU share the array: one at a time
            d_rays.free(); d_a.free(); d_b.free()
    finally:
        a.close(); b.close()


def test_noalloc_calls_of_two_contexts_back_to_back_without_a_host_sync(scene, bunny):
    """Two contexts on one GPU run on different non-blocking streams and share the GPU's one array.  Calls queued back to
    back from both, with no host synchronisation in between and batches of very different sizes, take turns on the array
    (each call's stream waits for the event recorded behind the previous call on that GPU): every call's hits equal
    pt_trace_rays', and the certified walk - whose list of rays for the second pass and whose ordering live in the array -
    is among them."""
    a, b = pt.PathTracer(0), pt.PathTracer(0)
    bufs = []
    try:
        helpers.upload(a, scene)
        helpers.upload(b, bunny)
        plan = []
        for ctx, model, n, trav in ((a, scene, MAX_RAYS, core.TRAVERSAL_FLAT_ORDERED), (b, bunny, 3_001, core.TRAVERSAL_CERTIFIED),
                                    (a, scene, 1_500_000, core.TRAVERSAL_CERTIFIED), (b, bunny, 40_000, core.TRAVERSAL_FLAT_ORDERED)):
            ctx.set_option(core.OPT_SORT_RAYS, 1)
            d_rays, d_ref, d_out = ctx.alloc(n * 32), ctx.alloc(n * 16), ctx.alloc(n * 16)
            bufs += [d_rays, d_ref, d_out]
            if model is scene:
                ctx.generate_rays(d_rays, n, seed=7 + n)
            else:
                d_rays.from_host(helpers.random_rays(model["nodes"][0]["a"], model["nodes"][0]["b"], n, seed=n))
            ctx.set_option(core.OPT_TRAVERSAL, trav)
            ctx.trace_rays_device(d_rays, n, d_ref)
            plan.append((ctx, d_rays, d_ref, d_out, n, trav))
        a.synchronize(); b.synchronize()
        for rounds in range(3):
            for ctx, d_rays, d_ref, d_out, n, trav in plan:            # a, b, a, b: no host sync anywhere in the loop
                if ctx.get_option(core.OPT_TRAVERSAL) != trav:
                    ctx.set_option(core.OPT_TRAVERSAL, trav)
                ctx.trace_rays_device_noalloc(d_rays, n, d_out)
        a.synchronize(); b.synchronize()
        for ctx, d_rays, d_ref, d_out, n, trav in plan:
            ref, out = d_ref.to_host(pt.HIT_DT, n), d_out.to_host(pt.HIT_DT, n)
            assert np.array_equal(ref.view(np.uint8), out.view(np.uint8)), f"{int((ref != out).sum())} of {n} hits differ (traversal {trav})"
    finally:
        for x in bufs:
            x.free()
        a.close(); b.close()
