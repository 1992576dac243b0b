"""bench.py as the harness runs it: `python bench.py --gpus N` with no launcher in the command line starts its own N ranks
(one process per GPU, torch.distributed.run on 127.0.0.1) and fails loudly - a rank count that does not match --gpus, or
context numbers that hang, end in a non-zero status.  On the one-GPU test box the ranks share the device over gloo
(PT_BENCH_BACKEND=gloo rehearses the rank logic; RCCL refuses ranks that share a device)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--tris", "200000", "--rays", "1048576", "--steps", "2", "--warmup", "1"]


def run_bench(args, env_extra=None, timeout=420):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p.returncode, [json.loads(l) for l in lines], p.stderr


def test_bench_starts_its_own_ranks():
    rc1, one, err1 = run_bench(["--gpus", "1", "--no-extra", "--no-cpu-baseline"] + SMALL)
    assert rc1 == 0 and len(one) == 1, err1[-2000:]
    rc2, two, err2 = run_bench(["--gpus", "2"] + SMALL, {"PT_BENCH_BACKEND": "gloo"})
    assert rc2 == 0 and len(two) == 1, err2[-2000:]                       # ONE line, from rank 0
    line = two[0]
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["rays_per_gpu"] == 1048576
    # two ranks share the one GPU here, so the whole-job value is about the one-rank value (not twice it), and never far below
    assert 0.5 * one[0]["value"] < line["value"] < 2.5 * one[0]["value"]
    part = line["extra"]["san_miguel_stand_in_partitioned"]
    assert "error" not in part, part
    assert part["samples_per_pixel_after_allreduce"] == [2.0 * part["frames_per_rank"]] * 2      # every pixel got every rank's samples
    assert part["native_rccl_entry_points"].get("skipped")                                       # ranks share a device in this rehearsal
    assert "cpu_baseline" not in line


def test_bench_watchdog_ends_in_a_failure_status_and_a_line():
    rc, lines, err = run_bench(["--gpus", "2"] + SMALL, {"PT_BENCH_BACKEND": "gloo", "PT_BENCH_EXTRA_TIMEOUT": "0.001"})
    assert rc != 0, "a timed-out run must not look clean"
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["value"] > 0          # the headline was measured before the time-out
    assert "stage:" in lines[0]["extra"]["error"]
