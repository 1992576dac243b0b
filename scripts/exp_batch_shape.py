"""Experiment: launch shape of the ordered batch (GPU box): workgroups per CU x refill threshold, the bench's scene and rays."""
import itertools, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt
from pathtracer_amd import core
trav = int(sys.argv[1]) if len(sys.argv) > 1 else 2
pos, aux, idx = pt.scenes.san_miguel_like()
nodes, ridx, _ = pt.build_bvh(pos, idx, 16)
ctx = pt.PathTracer(0)
ctx.upload_scene(pos, aux, ridx, nodes)
ctx.set_option(core.OPT_TRAVERSAL, trav)
n = 1 << 26
d_rays, d_hits = ctx.alloc(n * 32), ctx.alloc(n * 16)
ctx.generate_rays(d_rays, n, seed=20261111)
any_hit = len(sys.argv) > 4 and sys.argv[4] == "anyhit"
def run(label):
    for _ in range(2): ctx.trace_rays_device(d_rays, n, d_hits, any_hit=any_hit)
    ctx.synchronize(); t = time.perf_counter()
    for _ in range(4): ctx.trace_rays_device(d_rays, n, d_hits, any_hit=any_hit)
    ctx.synchronize(); dt = (time.perf_counter() - t) / 4
    print(f"traversal {trav}{' any-hit' if any_hit else ''} {label}: {dt * 1e3:.2f} ms per step, {n / dt / 1e6:.0f} Mrays/s", flush=True)
run("defaults")
blocks_list = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else (5, 6, 7, 8)
refill_list = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else (16, 24, 32)
for blocks, refill in itertools.product(blocks_list, refill_list):
    ctx.set_option(core.OPT_PERSIST_BLOCKS, blocks); ctx.set_option(core.OPT_REFILL_MIN, refill)
    run(f"blocks {blocks} refill {refill}")
ctx.close()
