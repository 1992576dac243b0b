#!/usr/bin/env python3
"""Headline benchmark: incoherent closest-hit Mrays/s on the SanMiguel-class scene (BASELINE.json
config "SanMiguel.ply 64M random incoherent rays, pure traversal microbench vs HBM roofline").

    python bench.py --gpus N --steps K --warmup W

One process per GPU (torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE).  A step is one
pass of pt_trace_rays over 2^26 synthetic rays per GPU that are already resident in HBM: the library
builds an order for the batch by origin cell (a counting sort of indices, three small kernels) and
traces the rays in that order; both are timed.  The walk is the library's default, the certified walk
(PT_OPT_TRAVERSAL = 3), whose hits are the reference-order walk's bit for bit: a first pass near child
first and a second, in reference order, over the rays the first lists; extra.near_first_walk is the
same batch under the faster walk that differs from the reference's on about one ray in 10^5.
The scene is replicated on every GPU and every rank traces its own range of the ray sequence, so
the path has no data exchange: weak scaling, value = rays of all ranks / max-over-ranks time.

SanMiguel.ply is a Git-LFS pointer stub in the reference checkout, so the scene is the procedural
stand-in pathtracer_amd.scenes.san_miguel_like (5.8 M triangles); config.workload says so.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     - bytes that crossed the fabric per launch (rocprofv3 counters of this same command, kept in
                 profiles/traffic.json) / mean kernel time (HIP events on the kernel's stream) against the 8 TB/s
                 HBM peak; the algorithmic bytes the walk asks for are reported beside it (requested_GBps),
  cpu_baseline - the CPU oracle in reference-faithful mode timed on a bounded sample of the same
                 rays on this box's host cores (N = 1 only),
  extra        - measured outside the timed region, for context: the near-first and any-hit walks on the same rays, spp/s
                 of the full wavefront path tracer (1080p and 4K x 8 bounces on the SanMiguel and sponza stand-ins,
                 CBbunny primary + 1 bounce; library defaults, and the SanMiguel frames under the near-first walk as
                 well); with N > 1 the sample-partitioned render and its all-reduce instead.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# The path tracer keeps three frames in flight on six streams; with four hardware queues per process (HIP's default,
# and what the test boxes export) streams share queues and wait on each other: about a tenth fewer spp/s at 1080p than
# with eight.  The setting is read when the HIP runtime starts, hence before torch is imported.  A value the
# environment already exports is what a C++ host linking the library would get too, so it is kept; PT_HW_QUEUES
# overrides it, and the effective value is printed with the result (config.hw_queues_per_process).  The headline
# (one traversal kernel on one stream) does not depend on it.
if "PT_HW_QUEUES" in os.environ:
    os.environ["GPU_MAX_HW_QUEUES"] = os.environ["PT_HW_QUEUES"]
else:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak
SAN_MIGUEL_OID = "db0ad4bbfa207d97d1a8aed92dc798078afaf846fc9ac2538567294ac905c139"      # reference Models/SanMiguel.ply:2


def sha256_of(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 24), b""):
            h.update(chunk)
    return h.hexdigest()


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# what the N > 1 context numbers were doing when the watchdog fired: set before every collective and every library call that
# may block on a peer, printed with the time-out so that the one record names the step that hung
STAGE = ["start"]


def stage(name):
    STAGE[0] = name


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: starts N ranks of this same command line under torch.distributed.run
    (one process per GPU, rendezvous on 127.0.0.1) as a child process and returns its exit status; rank 0's JSON line
    goes to this process's stdout.  The parent never initialises the GPU."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("bench.py: no launcher in the environment, starting the ranks: " + " ".join(cmd))
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--tris", type=int, default=5_800_000)
    ap.add_argument("--rays", type=int, default=1 << 26)
    ap.add_argument("--traversal", type=int, default=-1, help="-1 = library default")
    ap.add_argument("--layout", type=int, default=-1, help="record layout (PT_OPT_RECORD_LAYOUT); -1 = library default")
    ap.add_argument("--scene", type=str, default="", help="a real SanMiguel.ply; accepted only if its sha256 is the oid of the reference's Git-LFS stub")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    if args.gpus < 1:
        log("--gpus must be at least 1")
        sys.exit(2)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started without a launcher: this process becomes the launcher.  Nothing here has touched the GPU or imported
        # torch, and the ranks are children (one per GPU, torch.distributed.run), never an exec of this process.
        sys.exit(spawn_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # a line for N GPUs measured on another number of ranks would be filed under the wrong N
        log(f"error: WORLD_SIZE={world} but --gpus {args.gpus}: start {args.gpus} ranks (or run without a launcher, which starts them)")
        sys.exit(2)

    import torch
    import torch.distributed as dist
    # PT_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (ranks then share devices)
    backend = os.environ.get("PT_BENCH_BACKEND", "nccl")
    if backend == "nccl" and world > torch.cuda.device_count():
        log(f"error: {world} ranks but {torch.cuda.device_count()} GPUs visible; RCCL needs one GPU per rank "
            f"(PT_BENCH_BACKEND=gloo rehearses the rank logic with ranks sharing devices)")
        sys.exit(2)
    device = local_rank % max(1, torch.cuda.device_count()) if backend != "nccl" else local_rank
    torch.cuda.set_device(device)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)

    import pathtracer_amd as pt
    from pathtracer_amd import core

    threads = max(1, min(16, (os.cpu_count() or 16) // max(1, world)))
    t0 = time.time()
    scene_name = None
    if args.scene:
        # the reference's Models/SanMiguel.ply is a Git-LFS pointer whose second line names the real file's sha256
        digest = sha256_of(args.scene)
        if digest == SAN_MIGUEL_OID:
            pos, aux, idx = pt.load_ply(args.scene)
            scene_name = f"SanMiguel.ply (the reference's own file, sha256 {digest[:16]}...)"
        elif rank == 0:
            log(f"--scene {args.scene}: sha256 {digest} is not the oid of the reference's SanMiguel.ply stub ({SAN_MIGUEL_OID}); using the stand-in")
    if scene_name is None:
        pos, aux, idx = pt.scenes.san_miguel_like(args.tris)
    t1 = time.time()
    nodes, ridx, depth = pt.build_bvh(pos, idx, threads)
    t2 = time.time()
    ctx = pt.PathTracer(device)
    if args.layout >= 0:
        ctx.set_option(core.OPT_RECORD_LAYOUT, args.layout)      # read when the scene is flattened
    ctx.upload_scene(pos, aux, ridx, nodes, pt.default_lights(), pt.load_blue_noise())
    if args.traversal >= 0:
        ctx.set_option(core.OPT_TRAVERSAL, args.traversal)
    t3 = time.time()
    if rank == 0:
        log(f"scene: {ridx.shape[0]} triangles, {nodes.shape[0]} nodes, depth {depth}; generate {t1-t0:.1f}s, "
            f"BVH build {t2-t1:.1f}s ({threads} threads), flatten+upload {t3-t2:.1f}s")

    n = args.rays
    d_rays, d_hits = ctx.alloc(n * 32), ctx.alloc(n * 16)
    ctx.generate_rays(d_rays, n, seed=20261111, first=rank * n)
    ctx.synchronize()

    # algorithmic bytes per ray: the counting build of the same kernel on a 2^20-ray prefix
    nprefix = min(n, 1 << 20)
    st = ctx.trace_rays_device(d_rays, nprefix, d_hits, want_stats=True)
    box_per_ray, tri_per_ray, rec_per_ray = (float(st[k]) / nprefix for k in range(3))
    bytes_per_ray = 32.0 + 16.0 + 64.0 * rec_per_ray          # records are counted in 64 B units, whatever the layout
    kernel_name = {0: "k_trace_ref<MODE_RAYS>", 3: "k_trace_cert<MODE_RAYS> + second walk of the listed rays"}.get(
        args.traversal if args.traversal >= 0 else ctx.get_option(core.OPT_TRAVERSAL), "k_trace_flat<MODE_RAYS>")

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        ctx.trace_rays_device(d_rays, n, d_hits)
    barrier()
    kernel_ms = []
    tstart = time.perf_counter()
    for _ in range(args.steps):
        ctx.trace_rays_device(d_rays, n, d_hits)
        kernel_ms.append(None)          # event time is read after the timed region
    ctx.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - tstart
    if world > 1:
        dist.barrier()
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # mean kernel duration from HIP events (one more launch per sample so the timed loop above stays untouched)
    ev = []
    for _ in range(max(3, min(args.steps, 10))):
        ctx.trace_rays_device(d_rays, n, d_hits)
        ev.append(ctx.last_kernel_ms())
    kernel_ms = float(np.mean(ev))

    hits_prefix = d_hits.to_host(pt.HIT_DT, nprefix)
    hit_rate = float((hits_prefix["i1"] >= 0).mean())

    total_rays = float(n) * world * args.steps
    mrays = total_rays / elapsed / 1e6
    requested = bytes_per_ray * n / (kernel_ms * 1e-3) / 1e9

    # HBM-side bytes per launch come from separate rocprofv3 --pmc passes of this same command (bench.py cannot
    # collect counters on itself); profiles/traffic.json holds the latest, keyed by workload and record layout
    traffic, l2_hit, traffic_src, valu_busy = None, None, None, None
    trav = args.traversal if args.traversal >= 0 else ctx.get_option(core.OPT_TRAVERSAL)
    layout = ctx.get_option(core.OPT_RECORD_LAYOUT)
    try:
        for rec in json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["records"]:
            k = rec["key"]
            if (k["triangles"] == int(ridx.shape[0]) and k["rays_per_gpu"] == n and k["traversal"] == trav
                    and k.get("record_layout", 0) == layout):
                traffic, l2_hit, traffic_src = rec["traffic_bytes_per_launch"], rec.get("l2_hit_rate"), rec.get("source")
                valu_busy = rec.get("valu_busy")
    except (OSError, KeyError, ValueError):
        pass
    # the roofline figure is what crosses the fabric per unit of kernel time; the bytes the walk asks for (most of
    # which the L2 serves when the batch is traced in origin-cell order) are reported beside it, not against the peak
    hbm_gbs = traffic / (kernel_ms * 1e-3) / 1e9 if traffic else None

    result = {
        "metric": "incoherent closest-hit Mrays/s",
        "value": round(mrays, 2),
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f16 box test + f32 triangle test",
        "data": "synthetic",
        "config": {
            "workload": "SanMiguel 64M random incoherent closest-hit rays (BASELINE configs[4]); scene = " +
                        (f"{scene_name}, {ridx.shape[0]} triangles" if scene_name else
                         f"procedural stand-in san_miguel_like, {ridx.shape[0]} triangles (SanMiguel.ply is a Git-LFS stub in the reference)"),
            "rays_per_gpu": n,
            "triangles": int(ridx.shape[0]),
            "bvh_nodes": int(nodes.shape[0]),
            "traversal": {0: "reference 32 B nodes, stackless", 1: "flattened records, reference visit order",
                          2: "flattened records, near child first",
                          3: "flattened records, certified walk (hits identical to the reference order's)"}[trav],
            "record_layout": {0: "64 B two-box records", 1: "48 B two-box records", 2: "128 B two-level treelet records"}.get(layout, layout),
            "ray_ordering": "each batch is traced in the order of its origin cells (16^3 grid over the scene box); the order is "
                            "built inside the timed step, results are stored under the rays' own indices",
            "other_kernels_ms_per_step": round(elapsed / args.steps * 1e3 - kernel_ms, 3),
            "hit_rate": round(hit_rate, 4),
            "box_tests_per_ray": round(box_per_ray, 2),
            "triangle_tests_per_ray": round(tri_per_ray, 3),
            "parallelism": f"{world} x replicated scene, disjoint ray ranges, no exchange",
            "hw_queues_per_process": int(os.environ["GPU_MAX_HW_QUEUES"]),
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(hbm_gbs, 1) if hbm_gbs else None,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(hbm_gbs / HBM_PEAK_GBS, 4) if hbm_gbs else None,
            "traffic": traffic,
            "traffic_source": traffic_src,
            "l2_hit_rate": l2_hit,
            "valu_busy": valu_busy,
            "requested_GBps": round(requested, 1),
            "kernel": f"{kernel_name} (closest hit)",
            "kernel_ms": round(kernel_ms, 3),
            "bytes_per_ray": round(bytes_per_ray, 1),
            "records_per_ray": round(rec_per_ray, 2),
            "note": "achieved = bytes that crossed the fabric per launch (traffic: 2 x FETCH_SIZE + WRITE_SIZE of separate "
                    "rocprofv3 --pmc passes of this command, profiles/traffic.json) / kernel time measured here with HIP events; "
                    "null when no counter record matches this workload.  requested_GBps = algorithmic bytes the walk asks for "
                    "(48 B per ray + the records it fetches, counted by the counting build of the same kernel on a 2^20-ray "
                    "prefix) / the same time: the L2 serves most of those, so it is not a fraction of the HBM peak",
        },
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(pos, aux, ridx, nodes, d_rays, args.cpu_seconds)
    extra = {}

    def extra_item(name, fn):
        """One context number; a failure is reported under the item's own name and leaves the headline and the other items alone."""
        try:
            extra[name] = fn()
        except Exception as e:
            extra[name] = {"error": f"{type(e).__name__}: {e}"}

    single = rank == 0 and world == 1 and not args.no_extra
    if single:
        extra_item("near_first_walk" if trav == 3 else "certified_walk", lambda: other_walk(ctx, d_rays, d_hits, n, trav))
        extra_item("any_hit", lambda: any_hit_walk(ctx, d_rays, d_hits, n))
    d_rays.free(); d_hits.free()
    if single:
        extra_item("san_miguel_stand_in", lambda: path_tracer_spp(ctx, ridx.shape[0], nodes.shape[0], per_call=True))
        extra_item("san_miguel_stand_in_4k", lambda: path_tracer_spp(ctx, ridx.shape[0], nodes.shape[0], 3840, 2160, 96))
        extra_item("san_miguel_stand_in_near_first_frames" if ctx.get_option(core.OPT_TRAVERSAL) == 3 else "san_miguel_stand_in_certified_frames",
                   lambda: other_walk_frames(ctx, ridx.shape[0], nodes.shape[0]))

        def sponza():
            spos, saux, sidx = pt.scenes.sponza_like()
            snodes, sridx, _ = pt.build_bvh(spos, sidx, threads)
            ctx.upload_scene(spos, saux, sridx, snodes, pt.default_lights(), pt.load_blue_noise())
            return path_tracer_spp(ctx, sridx.shape[0], snodes.shape[0], camera=pt.scenes.SPONZA_CAMERA)
        extra_item("sponza_stand_in", sponza)

        def bunny():
            # BASELINE configs[1]: the reference's own model, primary + one diffuse bounce (numRays = 2), shadow rays included
            bpos, baux, bidx = pt.load_ply(os.path.join(ROOT, "tests", "golden", "models", "CBbunny.ply"))
            bnodes, bridx, _ = pt.build_bvh(bpos, bidx, threads)
            ctx.upload_scene(bpos, baux, bridx, bnodes, pt.default_lights(), pt.load_blue_noise())
            cam = dict(orbit_angle=-np.pi / 2, orbit_height=0.75, focus_center=(0.0, 0.75, 0.0))
            return path_tracer_spp(ctx, bridx.shape[0], bnodes.shape[0], nf=600, camera=cam, NR=2)
        extra_item("cb_bunny_primary_plus_one_bounce", bunny)
        result["extra"] = extra
    # From here on nothing is timed for the headline, and with N > 1 every step is a collective: a rank that dies or hangs
    # in one would hold the others, and the measured line with them.  A watchdog therefore ends the process once the context
    # numbers have taken longer than PT_BENCH_EXTRA_TIMEOUT seconds (default 300): rank 0 prints the line it has, with the
    # time-out and the stage it struck in named under extra, and every rank leaves with os._exit(3) (no exec, no wait on a peer).
    import threading
    printed = threading.Lock()

    def emit(note=None):
        """Prints the line once; False if another thread has already taken that on."""
        if not printed.acquire(blocking=False):
            return False
        if rank == 0:
            if note:
                result.setdefault("extra", {})["error"] = note
            print(json.dumps(result), flush=True)
        return True

    def give_up():
        if not emit(f"context numbers for N > 1 did not finish within {extra_timeout:.0f} s (stage: {STAGE[0]}); the headline above was measured before them"):
            return      # the main thread is printing its line: it got there after all
        log(f"rank {rank}: watchdog: timed out after {extra_timeout:.0f} s in stage '{STAGE[0]}'")
        sys.stderr.flush()
        os._exit(3)     # a time-out is a failure: the harness must see a non-zero status, whatever the line says

    extra_timeout = float(os.environ.get("PT_BENCH_EXTRA_TIMEOUT", "300"))
    watchdog = None
    if world > 1:
        watchdog = threading.Timer(extra_timeout, give_up)
        watchdog.daemon = True
        watchdog.start()
    if world > 1 and not args.no_extra:
        parts = {}
        for name, shape in (("san_miguel_stand_in_partitioned", (1920, 1080, 96)), ("san_miguel_stand_in_partitioned_4k", (3840, 2160, 32))):
            try:
                parts[name] = path_tracer_spp_partitioned(ctx, ridx.shape[0], nodes.shape[0], rank, world, backend, *shape)
            except Exception as e:      # a rank-local failure outside the guarded parts: report it, skip what would follow
                parts[name] = {"error": f"{type(e).__name__}: {e}"}
                break
        if rank == 0:
            result["extra"] = parts
    try:
        ctx.close()
        if world > 1:
            stage("final torch barrier")
            dist.barrier()
    finally:
        if watchdog:
            watchdog.cancel()
        emit()
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(pos, aux, ridx, nodes, d_rays, budget_s):
    """The CPU oracle (reference-faithful walk: 32 B nodes, fixed order, fp16 box test) on a bounded
    sample of the same rays, one thread per core of this box's CPU share."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    import pathtracer_amd as pt
    cores = max(1, min(16, os.cpu_count() or 1))
    osc = oracle.Scene(pos, aux, ridx, nodes)
    probe = d_rays.to_host(pt.RAY_DT, 1 << 16)
    t = time.perf_counter()
    osc.trace(probe, threads=cores)
    rate = len(probe) / (time.perf_counter() - t)
    nsample = int(min(1 << 24, max(1 << 16, rate * budget_s)))
    nsample = 1 << int(np.floor(np.log2(nsample)))
    rays = d_rays.to_host(pt.RAY_DT, nsample)
    times = []
    for _ in range(3):
        t = time.perf_counter()
        osc.trace(rays, threads=cores)
        times.append(time.perf_counter() - t)
    med = float(np.median(times))
    out = {"value": round(nsample / med / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
           "sample": f"first {nsample} rays of rank 0's ray set, same scene, 3 repeats, median; reference-faithful CPU walk "
                     f"(clang++ -O3, {cores} threads)"}
    # spp/s of the whole frame graph (launch, 8 x {trace + shade + shadow walk, raysort}, composite) on the same scene and
    # camera as the GPU's 1080p context line, at a size the CPU serves in seconds; a sample costs the same per pixel
    # at any size, so the 1080p figure is this one scaled by the pixel count
    W, H, NR = 480, 270, 8
    fr = oracle.Frame(W, H, NR)
    osc2 = oracle.Scene(pos, aux, ridx, nodes, pt.default_lights(), pt.load_blue_noise())
    consts = pt.fill_constants(W, H, NR, ridx.shape[0], nodes.shape[0], **pt.scenes.COURTYARD_CAMERA)
    mode = oracle.MODE_ZERO_NEXT_SLOT          # missed rays stay in the job list and are traced again each bounce, as in the reference
    frames, t = 0, time.perf_counter()
    while frames < 2 or (time.perf_counter() - t < 0.5 * budget_s and frames < 64):
        consts["frameIndex"] = frames + 1
        oracle.render_frame(osc2, fr, consts, mode=mode, threads=cores)
        frames += 1
    dt = time.perf_counter() - t
    out["spp_per_s"] = {"value_at_480x270": round(frames / dt, 3), "scaled_to_1080p": round(frames / dt * (W * H) / (1920 * 1080), 4),
                        "sample": f"{frames} frames of {W}x{H} x {NR} bounces + shadow rays, reference-faithful frame graph, trace passes on {cores} threads"}
    return out


def other_walk(ctx, d_rays, d_hits, n, headline_traversal):
    """Context number: the same batch under the other of the two production walks - the near-first walk (PT_OPT_TRAVERSAL = 2)
    when the headline is the certified walk (= 3, the library default), the certified walk otherwise - timed like the headline
    (whole calls, ordering pass included); and, compared on the device, on how many rays each of the two differs from the
    reference-order walk (= 1)."""
    import torch
    from pathtracer_amd import core

    class _Alias:
        def __init__(self, ptr, nbytes):
            self.__cuda_array_interface__ = {"shape": (nbytes // 4,), "typestr": "<i4", "data": (ptr, False), "version": 3, "strides": None}

    def hits_of(traversal, buf):
        ctx.set_option(core.OPT_TRAVERSAL, traversal)
        ctx.trace_rays_device(d_rays, n, buf)
        ctx.synchronize()
        return torch.as_tensor(_Alias(buf.ptr, n * 16), device=torch.device("cuda", ctx.device_id)).view(-1, 4)

    alt = 2 if headline_traversal == 3 else 3
    other = ctx.alloc(n * 16)
    try:
        ctx.set_option(core.OPT_TRAVERSAL, alt)
        for _ in range(2):
            ctx.trace_rays_device(d_rays, n, d_hits)
        ctx.synchronize()
        steps = 5
        t = time.perf_counter()
        for _ in range(steps):
            ctx.trace_rays_device(d_rays, n, d_hits)
        ctx.synchronize()
        dt = time.perf_counter() - t
        mine = hits_of(alt, d_hits)
        ref = hits_of(1, other)
        differ_alt = int((mine != ref).any(dim=1).sum().item())
        head = hits_of(headline_traversal, d_hits)
        differ_head = int((head != ref).any(dim=1).sum().item())
    finally:
        ctx.set_option(core.OPT_TRAVERSAL, headline_traversal)
        other.free()
    return {"traversal": alt, "Mrays_per_s": round(n * steps / dt / 1e6, 1), "ms_per_step": round(dt / steps * 1e3, 3),
            "rays_whose_hit_differs_from_the_reference_order_walk": differ_alt,
            "rays_whose_hit_differs_from_the_reference_order_walk_in_the_headline_walk": differ_head, "rays": n}


def any_hit_walk(ctx, d_rays, d_hits, n, steps=5):
    """Context number: the same ray set as any-hit rays (the walk of the shadow rays, intersections.glsl:73-77 with stopIfHit):
    a ray ends at the first triangle it hits.  The occluded share must equal the closest-hit walk's hit rate."""
    import pathtracer_amd as pt
    ctx.trace_rays_device(d_rays, n, d_hits, any_hit=True)
    ctx.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        ctx.trace_rays_device(d_rays, n, d_hits, any_hit=True)
    ctx.synchronize()
    dt = time.perf_counter() - t
    m = min(n, 1 << 20)
    occluded = float((d_hits.to_host(pt.HIT_DT, m)["i1"] >= 0).mean())
    return {"Mrays_per_s": round(n * steps / dt / 1e6, 1), "ms_per_step": round(dt / steps * 1e3, 3), "occluded_share_of_first_2^20": round(occluded, 4), "rays": n}


def other_walk_frames(ctx, num_triangles, num_nodes):
    """Context number: the path tracer at 1080p x 8 bounces under the other of the two production walks (near-first when the
    library default is the certified walk, the certified walk otherwise), and on how many pixels the accumulation after six
    frames from a cleared buffer differs, byte for byte, from that of the reference-order walk (= 1) - under that walk and
    under the default."""
    import pathtracer_amd as pt
    from pathtracer_amd import core
    W, H, NR, nf = 1920, 1080, 8, 240
    default = ctx.get_option(core.OPT_TRAVERSAL)
    alt = 2 if default == 3 else 3
    ctx.resize(W, H, NR)
    ctx.set_constants(pt.fill_constants(W, H, NR, num_triangles, num_nodes, **pt.scenes.COURTYARD_CAMERA))
    acc = {}
    try:
        for trav in (alt, 1, default):
            ctx.set_option(core.OPT_TRAVERSAL, trav)
            ctx.clear_accumulation()
            ctx.render_samples(1, 6)
            acc[trav] = ctx.readback_accumulation()
        ctx.set_option(core.OPT_TRAVERSAL, alt)
        ctx.render_samples(7, NR + 1)
        ctx.synchronize()
        t = time.perf_counter()
        ctx.render_samples(20, nf)
        ctx.synchronize()
        dt = time.perf_counter() - t
        roof = frame_roofline(ctx, num_triangles, W, H, NR, nf)
    finally:
        ctx.set_option(core.OPT_TRAVERSAL, default)
    differ = lambda a, b: int((a.view(np.uint32) != b.view(np.uint32)).any(axis=2).sum())
    out = {"traversal": alt, "spp_per_s_1080p": round(nf / dt, 2), "frames_timed": nf,
           "pixels_differing_from_the_reference_order_walk_after_6_frames": differ(acc[alt], acc[1]),
           "pixels_differing_from_the_reference_order_walk_after_6_frames_in_the_default_walk": differ(acc[default], acc[1]), "pixels": W * H}
    if roof:
        out["roofline"] = roof
    return out


def path_tracer_spp(ctx, num_triangles, num_nodes, W=1920, H=1080, nf=240, camera=None, NR=8, per_call=False):
    """Context number: samples per pixel per second of the wavefront path tracer at 1080p (or the size given), 8 bounces
    with next-event estimation, on the scene the context holds (BASELINE configs[2] shape), library defaults: one
    pt_render_samples call of nf frames through the frame train, which takes nf + 8 steps (the eight that fill and drain the
    ring are inside the timed region; paths of fewer than three slots go frame by frame, three frames in flight)."""
    import pathtracer_amd as pt
    ctx.resize(W, H, NR)
    ctx.set_constants(pt.fill_constants(W, H, NR, num_triangles, num_nodes, **(camera or pt.scenes.COURTYARD_CAMERA)))
    ctx.render_samples(1, NR + 1)          # warm-up: brings the ring of ray stacks to its full size
    ctx.synchronize()
    t = time.perf_counter()
    ctx.render_samples(100, nf)
    ctx.synchronize()
    dt = time.perf_counter() - t
    rc = ctx.render_sample(100 + nf, want_ray_counts=True)
    rays = int(rc[:2 * NR].sum())
    spp = nf / dt
    out = {f"spp_per_s_{H}p": round(spp, 2), "frames_timed": nf, "triangles": int(num_triangles), "bounces": NR, "next_event_estimation": True,
           "rays_per_sample": rays, "Mrays_per_s": round(rays * nf / dt / 1e6, 1),
           # rays of one sample by bounce (closest-hit rays traced at that depth, shadow rays its shade pass queued) x samples per second
           "Mrays_per_s_by_bounce": {"closest_hit": [round(float(rc[2 * d]) * spp / 1e6, 1) for d in range(NR)],
                                     "shadow": [round(float(rc[2 * d + 1]) * spp / 1e6, 1) for d in range(NR)]}}
    roof = frame_roofline(ctx, num_triangles, W, H, NR, nf)
    if roof:
        out["roofline"] = roof
    if per_call:
        # the reference's own calling pattern (one f_onRender per frame, PathTracer.cpp:672): one render call per frame, with
        # the frame train flushed at the end of every call (library default) and kept between the calls (PT_OPT_TRAIN_KEEP)
        from pathtracer_amd import core
        for keep in (0, 1):
            ctx.set_option(core.OPT_TRAIN_KEEP, keep)
            ctx.render_samples(1, NR + 1)
            ctx.render_flush()
            ctx.synchronize()
            t = time.perf_counter()
            for f in range(nf // 2):
                ctx.render_sample(200 + f)
            ctx.render_flush()
            ctx.synchronize()
            out["spp_per_s_one_frame_per_call" + ("_train_kept" if keep else "")] = round((nf // 2) / (time.perf_counter() - t), 2)
        ctx.set_option(core.OPT_TRAIN_KEEP, 0)
    return out


def frame_roofline(ctx, num_triangles, W, H, NR, nf):
    """The frame train's traversal launch (one per band of rows and step: the closest-hit rays of every frame in flight and the
    shadow rays of the step before) on the same footing as the batch kernel: its mean duration measured here with HIP event
    pairs on the launches' own streams (PT_OPT_TIME_TRAVERSAL, a separate short render so that the timed frame rate above carries
    no events), against the bytes that crossed the fabric per launch, the L2 hit rate and the vector ALU's busy share from
    rocprofv3 --pmc passes of the same workload (profiles/traffic.json; null without a matching record)."""
    from pathtracer_amd import core
    trav, layout = ctx.get_option(core.OPT_TRAVERSAL), ctx.get_option(core.OPT_RECORD_LAYOUT)
    if NR < 3:
        return None          # short paths go through the per-frame pipeline: no train launch to speak of
    parts = ctx.get_option(core.OPT_TRAIN_PARTS)
    frames = min(nf, 64)
    ctx.set_option(core.OPT_TIME_TRAVERSAL, 1)
    try:
        ctx.traversal_ms()                      # start the record over
        ctx.render_samples(5000, frames)        # frames + NR steps, `parts` launches each, in launch order
        ms = ctx.traversal_ms()
    finally:
        ctx.set_option(core.OPT_TIME_TRAVERSAL, 0)
    # steps NR .. frames - 1 carry a frame at every depth: the steady state a long render runs in
    steady = ms[NR * parts: frames * parts]
    if len(ms) != (frames + NR) * parts or not len(steady):
        return None
    kernel_ms = float(steady.mean())
    launches = int(len(steady))
    rec = None
    try:
        for r in json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["records"]:
            k = r["key"]
            if (k.get("kernel") == "frame_train_traversal" and k["triangles"] == int(num_triangles) and k["width"] == W and k["height"] == H
                    and k["slots"] == NR and k["traversal"] == trav and k.get("record_layout", 0) == layout):
                rec = r
    except (OSError, KeyError, ValueError):
        pass
    out = {"bound": "hbm", "kernel": "k_trace_cert<MODE_MIXED> (first pass; a short list's second walk runs inside the shade launch, a long one's is in this figure)" if trav == 3 else "k_trace_flat<MODE_MIXED>" if not layout == 2 else "k_trace_tree<MODE_MIXED>",
           "kernel_ms_per_band_step": round(kernel_ms, 3), "bands": parts, "launches_timed": launches, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "traffic": None, "achieved": None, "frac": None, "l2_hit_rate": None, "valu_busy": None, "floor_ratio": None}
    if rec:
        traffic = rec["traffic_bytes_per_launch"]
        gbs = traffic / (kernel_ms * 1e-3) / 1e9
        out.update({"traffic": traffic, "achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4), "l2_hit_rate": rec.get("l2_hit_rate"),
                    "valu_busy": rec.get("valu_busy"), "traffic_source": rec.get("source")})
        if rec.get("additive_floor_ms"):
            # what the chip's measured L2-miss and L2-hit rates alone would take for this launch's misses and hits
            out["floor_ratio"] = round(rec["additive_floor_ms"] / kernel_ms, 3)
    return out


def path_tracer_spp_partitioned(ctx, num_triangles, num_nodes, rank, world, backend, W=1920, H=1080, nf=96):
    """Context number for N > 1: sample-partitioned rendering.  Every rank renders its own block of frame indices at
    1080p (or the size given; 3840 x 2160 is BASELINE configs[3]) x 8 bounces into its own accumulation buffer (per-frame reseed), then one all-reduce (RCCL) sums the RGBA32F
    buffers, .w = sample count included.  Every rank takes part in every collective whatever happens to it locally:
    a rank that fails says so in a flag that is reduced first, and then nobody enters the data all-reduce."""
    import torch
    import torch.distributed as dist
    import pathtracer_amd as pt
    from pathtracer_amd import core, distributed
    NR = 8
    dev = "cuda" if backend == "nccl" else "cpu"
    ok, err = 1, ""
    stage(f"{W}x{H}: set-up and warm-up render")
    try:
        ctx.set_option(core.OPT_RESEED, 1)
        ctx.resize(W, H, NR)
        ctx.set_constants(pt.fill_constants(W, H, NR, num_triangles, num_nodes, **pt.scenes.COURTYARD_CAMERA))
        ctx.render_samples(1 + rank * 16, NR + 1)       # warm-up, also brings the ring of ray stacks to its full size
        ctx.clear_accumulation()
        ctx.synchronize()
    except Exception as e:
        ok, err = 0, str(e)
    stage(f"{W}x{H}: torch all_reduce of the set-up flag")
    flag = torch.tensor([ok], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        return {"error": err or "another rank failed during set-up"}
    stage(f"{W}x{H}: torch barrier before the timed render")
    dist.barrier()
    stage(f"{W}x{H}: rendering this rank's frames (pt_render_samples)")
    t = time.perf_counter()
    try:
        ctx.render_samples(1000 + rank * nf, nf)
        ctx.synchronize()
        t_render = time.perf_counter() - t
    except Exception as e:
        ok, err, t_render = 0, str(e), 0.0
    stage(f"{W}x{H}: torch all_reduce of the render flag")
    flag = torch.tensor([ok], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        return {"error": err or "another rank failed while rendering"}
    # the library's own RCCL entry points (what a C++ host without torch calls) on the same buffer first: the sum is taken,
    # read back and the buffer restored, then torch's all-reduce runs on it and the two results are compared
    native = native_allreduce_check(ctx, rank, world, backend, dev)
    stage(f"{W}x{H}: torch all_reduce deciding whether there is a native result")
    # whether there is a native result to compare is decided by all ranks together: the comparison below ends in a collective
    have = torch.tensor([0 if native.get("result") is None else 1], dtype=torch.int32, device=dev)
    dist.all_reduce(have, op=dist.ReduceOp.MIN)
    if int(have.item()) == 0 and native.get("result") is not None:
        native["result"] = None
        native.setdefault("error", "another rank has no native result")
    t_ar = time.perf_counter()
    stage(f"{W}x{H}: torch all_reduce of the accumulation ({16 * W * H} bytes)")
    distributed.allreduce_accumulation_torch(ctx)       # raises on every rank or on none: same library, same buffer size
    dt = time.perf_counter() - t - native["seconds_spent"]
    t_torch = time.perf_counter() - t_ar
    stage(f"{W}x{H}: torch all_reduce (max) of the times")
    tt = torch.tensor([dt, t_render], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt, t_render = float(tt[0].item()), float(tt[1].item())
    out = {f"spp_per_s_{H}p": round(world * nf / dt, 2), "frames_per_rank": nf, "bounces": NR, "triangles": int(num_triangles),
           "render_s": round(t_render, 4), "allreduce_s": round(dt - t_render, 4), "allreduce_bytes": 16 * W * H}
    acc = ctx.readback_accumulation()
    if native.get("result") is not None:
        same = np.array_equal(native["result"][..., 3], acc[..., 3]) and np.allclose(native["result"], acc, rtol=1e-6, atol=1e-7)
        native["equals_torch_allreduce"] = bool(same)
        native["max_abs_difference"] = float(np.abs(native["result"] - acc).max())
        native["torch_allreduce_s"] = round(t_torch, 4)
        stage(f"{W}x{H}: torch all_reduce of the native-equals-torch verdict")
        agree = torch.tensor([1 if same else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(agree, op=dist.ReduceOp.MIN)
        native["equals_torch_allreduce_on_every_rank"] = bool(int(agree.item()))
    native.pop("result", None); native.pop("seconds_spent", None)
    out["native_rccl_entry_points"] = native
    if rank == 0:
        out["samples_per_pixel_after_allreduce"] = [float(acc[..., 3].min()), float(acc[..., 3].max())]
    ctx.set_option(core.OPT_RESEED, 0)
    return out


def native_allreduce_check(ctx, rank, world, backend, dev):
    """pt_comm_unique_id / pt_comm_init / pt_allreduce_accumulation with N > 1: rank 0's id goes to every rank over the
    torch group, every rank joins the communicator and sums its accumulation in place; the sum is read back and the
    buffer put back as it was, so that the torch all-reduce that follows starts from the same data.  Every rank enters
    every collective: a rank that fails locally says so in a flag that is reduced before the next one.  Skipped where
    ranks share a device (the gloo rehearsal), which RCCL refuses."""
    import torch
    import torch.distributed as dist
    from pathtracer_amd import core
    t0 = time.perf_counter()
    if backend != "nccl":
        return {"skipped": "ranks share a device in this rehearsal", "result": None, "seconds_spent": time.perf_counter() - t0}
    out = {"result": None}
    try:
        uid = torch.zeros(128, dtype=torch.uint8, device=dev)
        ok = 1
        if rank == 0:
            try:
                uid = torch.from_numpy(core.comm_unique_id().copy()).to(dev)
            except Exception as e:
                ok, out["error"] = 0, str(e)
        stage("native RCCL: torch broadcast of the id flag")
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.broadcast(flag, 0)
        if int(flag.item()) == 0:
            out.setdefault("error", "rank 0 could not make a communicator id")
            out["seconds_spent"] = time.perf_counter() - t0
            return out
        stage("native RCCL: torch broadcast of the communicator id")
        dist.broadcast(uid, 0)
        before = ctx.readback_accumulation()
        stage("native RCCL: pt_comm_init (ncclCommInitRank)")
        ctx.comm_init(rank, world, uid.cpu().numpy())            # collective: every rank is here
        ctx.synchronize()
        stage("native RCCL: torch barrier after pt_comm_init")
        dist.barrier()
        t = time.perf_counter()
        stage("native RCCL: pt_allreduce_accumulation (ncclAllReduce)")
        ctx.allreduce_accumulation()
        ctx.synchronize()
        out["allreduce_s"] = round(time.perf_counter() - t, 4)
        out["result"] = ctx.readback_accumulation()
        ctx.upload_accumulation(before)
    except Exception as e:
        out["error"] = str(e)
        out["result"] = None
    out["seconds_spent"] = time.perf_counter() - t0
    return out


if __name__ == "__main__":
    main()
