"""The N > 1 path on CPU: two gloo ranks render their share of the samples (the oracle stands in for
the GPU), all-reduce the accumulation buffers, and must reproduce the single-process image."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import helpers
    import oracle
    import pathtracer_amd as pt
    from pathtracer_amd import distributed as ptd
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    pos, aux, idx = pt.load_ply(os.path.join(ROOT, "tests", "golden", "models", "CBmonkey.ply"))
    nodes, ridx, _ = pt.build_bvh(pos, idx, 2)
    model = dict(positions=pos, aux=aux, indices=ridx, nodes=nodes)
    osc = helpers.oracle_scene(model, pt.default_lights(), pt.load_blue_noise())
    W, H, NR, NF = 32, 24, 3, 5
    fr = oracle.Frame(W, H, NR)
    mode = oracle.MODE_ZERO_NEXT_SLOT | oracle.MODE_RESEED_EVERY_FRAME

    def render(f):
        c = pt.fill_constants(W, H, NR, ridx.shape[0], nodes.shape[0], frame_index=f)
        oracle.render_frame(osc, fr, c, mode=mode)

    n = ptd.render_partitioned(render, 1, NF, rank, world)
    acc = fr.accumulation.copy()
    ptd.allreduce_accumulation_host(acc)
    np.save(os.path.join(out_dir, f"acc_{rank}.npy"), acc)
    np.save(os.path.join(out_dir, f"n_{rank}.npy"), np.array([n]))
    dist.barrier()
    dist.destroy_process_group()


def test_partition_is_a_partition():
    from pathtracer_amd import distributed as ptd
    for world in (1, 2, 3, 8):
        for nf in (0, 1, 7, 16):
            got = sorted(sum((ptd.partition_samples(5, nf, r, world) for r in range(world)), []))
            assert got == list(range(5, 5 + nf))
            blocks = [ptd.partition_block(5, nf, r, world) for r in range(world)]
            assert sorted(sum((list(range(f, f + c)) for f, c in blocks), [])) == list(range(5, 5 + nf))
            assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1
    with pytest.raises(ValueError):
        ptd.partition_samples(0, 4, 2, 2)
    with pytest.raises(ValueError):
        ptd.partition_block(0, 4, 2, 2)


def test_two_ranks_reproduce_single_process(tmp_path, oracle_lib):
    import torch.multiprocessing as mp
    import helpers
    import oracle
    import pathtracer_amd as pt
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a0, a1 = np.load(tmp_path / "acc_0.npy"), np.load(tmp_path / "acc_1.npy")
    assert np.array_equal(a0, a1)
    assert int(np.load(tmp_path / "n_0.npy")[0]) == 3 and int(np.load(tmp_path / "n_1.npy")[0]) == 2
    # single process, all five frames, same per-frame reseed
    pos, aux, idx = pt.load_ply(os.path.join(ROOT, "tests", "golden", "models", "CBmonkey.ply"))
    nodes, ridx, _ = pt.build_bvh(pos, idx, 2)
    model = dict(positions=pos, aux=aux, indices=ridx, nodes=nodes)
    osc = helpers.oracle_scene(model, pt.default_lights(), pt.load_blue_noise())
    W, H, NR, NF = 32, 24, 3, 5
    fr = oracle.Frame(W, H, NR)
    for f in range(1, 1 + NF):
        c = pt.fill_constants(W, H, NR, ridx.shape[0], nodes.shape[0], frame_index=f)
        oracle.render_frame(osc, fr, c, mode=oracle.MODE_ZERO_NEXT_SLOT | oracle.MODE_RESEED_EVERY_FRAME)
    assert (a0[..., 3] == NF).all()
    # same samples, summed in a different order: equal up to fp32 rounding of the sums
    assert np.allclose(a0, fr.accumulation, rtol=1e-5, atol=1e-6)
