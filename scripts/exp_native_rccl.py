"""Diagnostic: bench.py's native RCCL leg with a one-rank communicator (the path N > 1 takes, minus the peers)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29671")
import numpy as np
import torch, torch.distributed as dist
import bench
import pathtracer_amd as pt
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
pos, aux, idx = pt.scenes.sponza_like(60000)
nodes, ridx, _ = pt.build_bvh(pos, idx, 8)
ctx = pt.PathTracer(0)
ctx.upload_scene(pos, aux, ridx, nodes, pt.default_lights(), pt.load_blue_noise())
out = bench.path_tracer_spp_partitioned(ctx, ridx.shape[0], nodes.shape[0], 0, 1, "nccl", 640, 360, 6)
print(out)
assert out["native_rccl_entry_points"]["equals_torch_allreduce_on_every_rank"]
ctx.close(); dist.destroy_process_group()
