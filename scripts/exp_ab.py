"""A/B measurement of one library build (PTCORE_LIB selects it) on the bench scene, GPU box: the 2^26-ray batch (near-first,
any hit, certified; kernel ms by HIP events, best of --reps) and 1080p x 8 frames through the frame train (near-first and
certified).  Prints one line.  usage: exp_ab.py [--what batch,anyhit,cert,frames,certframes] [--layout L] [--frames N]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import pathtracer_amd as pt
from pathtracer_amd import core
ap = argparse.ArgumentParser()
ap.add_argument("--what", default="batch,anyhit,cert,frames,certframes")
ap.add_argument("--layout", type=int, default=0)
ap.add_argument("--frames", type=int, default=120)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--rays", type=int, default=1 << 26)
ap.add_argument("--tris", type=int, default=5_800_000)
ap.add_argument("--blocks", type=int, default=0)
ap.add_argument("--refill", type=int, default=0)
ap.add_argument("--stats", type=int, default=0)
ap.add_argument("--check", type=int, default=0, help="1: compare the batch's hits with the reference-order walk's on the device")
args = ap.parse_args()
what = args.what.split(",")
pos, aux, idx = pt.scenes.san_miguel_like(args.tris)
nodes, ridx, _ = pt.build_bvh(pos, idx, 16)
ctx = pt.PathTracer(0)
ctx.set_option(core.OPT_RECORD_LAYOUT, args.layout)
ctx.upload_scene(pos, aux, ridx, nodes, pt.default_lights(), pt.load_blue_noise())
if args.blocks: ctx.set_option(core.OPT_PERSIST_BLOCKS, args.blocks)
if args.refill: ctx.set_option(core.OPT_REFILL_MIN, args.refill)
out = [os.path.basename(os.environ.get("PTCORE_LIB", "product")).replace("libptcore_exp_", "").replace(".so", ""), f"layout {args.layout}"]
n = args.rays
if any(w in what for w in ("batch", "anyhit", "cert")):
    d_rays, d_hits = ctx.alloc(n * 32), ctx.alloc(n * 16)
    ctx.generate_rays(d_rays, n, seed=20261111)
    ctx.synchronize()
    def batch(trav, any_hit=False):
        ctx.set_option(core.OPT_TRAVERSAL, trav)
        ctx.trace_rays_device(d_rays, n, d_hits, any_hit=any_hit); ctx.synchronize()
        ms = []
        t = time.perf_counter()
        for _ in range(args.reps):
            ctx.trace_rays_device(d_rays, n, d_hits, any_hit=any_hit); ms.append(ctx.last_kernel_ms())
        ctx.synchronize()
        wall = (time.perf_counter() - t) / args.reps * 1e3
        return min(ms), wall
    if "batch" in what:
        k, w = batch(2); out.append(f"batch {k:.2f} ms kernel {w:.2f} ms call {n / w / 1e3:.0f} Mrays/s")
        if args.stats:
            st = ctx.trace_rays_device(d_rays, 1 << 20, d_hits, want_stats=True)
            out.append(f"box/ray {st[0] / 2**20:.1f} tri/ray {st[1] / 2**20:.2f} rec64/ray {st[2] / 2**20:.1f}")
        if args.check:
            import torch
            class _A:
                def __init__(s, ptr, nb): s.__cuda_array_interface__ = {"shape": (nb // 4,), "typestr": "<i4", "data": (ptr, False), "version": 3, "strides": None}
            other = ctx.alloc(n * 16)
            ctx.set_option(core.OPT_TRAVERSAL, 2); ctx.trace_rays_device(d_rays, n, d_hits); ctx.synchronize()
            a = torch.as_tensor(_A(d_hits.ptr, n * 16), device="cuda:0").view(-1, 4).clone()
            ctx.set_option(core.OPT_TRAVERSAL, 1); ctx.trace_rays_device(d_rays, n, other); ctx.synchronize()
            b = torch.as_tensor(_A(other.ptr, n * 16), device="cuda:0").view(-1, 4)
            out.append(f"near-first differs from reference order on {int((a != b).any(dim=1).sum())}")
            other.free()
    if "anyhit" in what:
        k, w = batch(2, True); out.append(f"anyhit {k:.2f} ms")
    if "cert" in what:
        k, w = batch(3); out.append(f"cert {k:.2f} ms kernel {w:.2f} ms call {n / w / 1e3:.0f} Mrays/s")
        if args.stats:
            st = ctx.trace_rays_device(d_rays, 1 << 22, d_hits, want_stats=True)
            out.append(f"cert rec64/ray {st[2] / 2**22:.2f} second walks {int(st[3]) - 2**22} of 2^22")
        if args.check:
            import torch
            class _B:
                def __init__(s, ptr, nb): s.__cuda_array_interface__ = {"shape": (nb // 4,), "typestr": "<i4", "data": (ptr, False), "version": 3, "strides": None}
            other = ctx.alloc(n * 16)
            ctx.set_option(core.OPT_TRAVERSAL, 3); ctx.trace_rays_device(d_rays, n, d_hits); ctx.synchronize()
            a = torch.as_tensor(_B(d_hits.ptr, n * 16), device="cuda:0").view(-1, 4).clone()
            ctx.set_option(core.OPT_TRAVERSAL, 1); ctx.trace_rays_device(d_rays, n, other); ctx.synchronize()
            b = torch.as_tensor(_B(other.ptr, n * 16), device="cuda:0").view(-1, 4)
            out.append(f"certified differs from reference order on {int((a != b).any(dim=1).sum())}")
            other.free()
    d_rays.free(); d_hits.free()
W, H, NR = 1920, 1080, 8
for name, trav in (("frames", 2), ("certframes", 3)):
    if name not in what: continue
    ctx.set_option(core.OPT_TRAVERSAL, trav)
    ctx.resize(W, H, NR)
    ctx.set_constants(pt.fill_constants(W, H, NR, ridx.shape[0], nodes.shape[0], **pt.scenes.COURTYARD_CAMERA))
    ctx.render_samples(1, NR + 1); ctx.synchronize()
    best = 0
    for rep in range(2):
        t = time.perf_counter()
        ctx.render_samples(100, args.frames); ctx.synchronize()
        best = max(best, args.frames / (time.perf_counter() - t))
    acc = ctx.readback_accumulation()
    out.append(f"{name} {best:.1f} spp/s (acc sum {float(acc[..., :3].sum()):.6e})")
print(" | ".join(out), flush=True)
ctx.close()
