"""Experiment: randomised differential test of the frame pipelines (GPU box).  Random image sizes, path depths, slot counts,
traversals, record layouts, reseeding, numbers of bands and sequences of call lengths: the frame train must leave the
accumulation, the current image, the ray counts and the live part of the ray stack that the per-frame pipeline leaves for the
same calls, and so must the train kept between calls (PT_OPT_TRAIN_KEEP) with each call cut into random pieces."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt
from pathtracer_amd import core
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
models = {}
for name in ("CBbunny", "CBmonkey"):
    pos, aux, idx = pt.load_ply(os.path.join(ROOT, "tests", "golden", "models", name + ".ply"))
    nodes, ridx, _ = pt.build_bvh(pos, idx)
    models[name] = (pos, aux, ridx, nodes)
cam = dict(orbit_angle=-np.pi / 2, orbit_height=0.75, focus_center=(0.0, 0.75, 0.0))
for case in range(cases):
    name = ("CBbunny", "CBmonkey")[int(rng.integers(2))]
    pos, aux, ridx, nodes = models[name]
    W, H = int(rng.integers(8, 200)), int(rng.integers(8, 120))
    slots = int(rng.integers(1, 10)); NR = int(rng.integers(1, slots + 1))
    trav = int(rng.choice([1, 2, 3])); layout = int(rng.choice([0, 0, 1])); reseed = int(rng.integers(2)); parts = int(rng.integers(1, 5))
    calls = [int(v) for v in rng.integers(1, 2 * NR + 4, size=int(rng.integers(1, 5)))]
    desc = f"case {case}: {name} {W}x{H} slots {slots} numRays {NR} traversal {trav} layout {layout} reseed {reseed} bands {parts} calls {calls}"
    out = {}
    for train in (1, 0):
        ctx = pt.PathTracer(0)
        ctx.set_option(core.OPT_RECORD_LAYOUT, layout)
        ctx.upload_scene(pos, aux, ridx, nodes, pt.default_lights(), pt.load_blue_noise())
        ctx.resize(W, H, slots)
        for o, v in ((core.OPT_TRAVERSAL, trav), (core.OPT_RESEED, reseed), (core.OPT_FRAME_TRAIN, train), (core.OPT_TRAIN_PARTS, parts),
                     (core.OPT_FRAMES_IN_FLIGHT, int(rng.integers(1, 5)) if not train else 3)):
            ctx.set_option(o, v)
        ctx.set_constants(pt.fill_constants(W, H, NR, ridx.shape[0], nodes.shape[0], **cam))
        frame = 1; rcs = []
        for k, n in enumerate(calls):
            rcs.append(ctx.render_samples(frame, n, want_ray_counts=True)[:2 * NR].copy()); frame += n
            if k == 1:
                ctx.clear_accumulation()           # the next call's first frame reseeds from its jitter
        slots_, _ = ctx.readback_raystack()
        out[train] = (ctx.readback_accumulation(), ctx.readback_current(), np.concatenate(rcs), slots_[:W * H * NR].reshape(W * H, NR))       # a pixel's slots lie numRays apart
        ctx.close()
    # the same frames once more with the train kept between calls: each call cut into random pieces, with reads that do not
    # empty the train (and must not disturb it) in between; ray counts are not asked for (that would empty it)
    rng2 = np.random.default_rng(1000 * seed + case)
    ctx = pt.PathTracer(0)
    ctx.set_option(core.OPT_RECORD_LAYOUT, layout)
    ctx.upload_scene(pos, aux, ridx, nodes, pt.default_lights(), pt.load_blue_noise())
    ctx.resize(W, H, slots)
    for o, v in ((core.OPT_TRAVERSAL, trav), (core.OPT_RESEED, reseed), (core.OPT_FRAME_TRAIN, 1), (core.OPT_TRAIN_PARTS, parts), (core.OPT_TRAIN_KEEP, 1)):
        ctx.set_option(o, v)
    consts = pt.fill_constants(W, H, NR, ridx.shape[0], nodes.shape[0], **cam)
    ctx.set_constants(consts)
    frame = 1
    for k, n in enumerate(calls):
        left = n
        while left:
            m = int(rng2.integers(1, left + 1))
            consts["frameIndex"] = frame
            ctx.set_constants(consts)
            ctx.render_samples(frame, m); frame += m; left -= m
            what = int(rng2.integers(4))
            if what == 0: ctx.readback_accumulation()
            elif what == 1: ctx.synchronize()
            elif what == 2: ctx.readback_current()
        if k == 1:
            ctx.clear_accumulation()
    slots_, _ = ctx.readback_raystack()
    kept = (ctx.readback_accumulation(), ctx.readback_current(), None, slots_[:W * H * NR].reshape(W * H, NR))
    ctx.close()
    a, b = out[1], out[0]
    same = [np.array_equal(np.asarray(a[k]).view(np.uint8), np.asarray(b[k]).view(np.uint8)) for k in range(3)]
    ok = all(same)
    # the live part of the last frame's stack: slots up to the first whose prob is not positive
    prob = b[3]["prob"].view(np.float16).astype(np.float32)
    live = np.cumprod(np.concatenate([np.ones((prob.shape[0], 1), bool), prob[:, :-1] > 0], axis=1), axis=1).astype(bool) & (prob > 0)
    same.append(np.array_equal(a[3][live].view(np.uint8), b[3][live].view(np.uint8)))
    same.append(np.array_equal(kept[0], b[0]) and np.array_equal(kept[1], b[1]) and np.array_equal(kept[3][live].view(np.uint8), b[3][live].view(np.uint8)))
    ok = all(same)
    print(desc + (": identical" if ok else f": DIFFERS (accumulation, current, ray counts, live stack, kept train: {same})"), flush=True)
    assert ok
print("fuzz ok")
