"""bench.py's launcher logic that needs no GPU: a rank count that does not match --gpus is refused before anything is measured."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_refuses_a_rank_count_that_is_not_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-extra", "--no-cpu-baseline"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and not p.stdout.strip() and "WORLD_SIZE=1 but --gpus 2" in p.stderr


def test_bench_refuses_zero_gpus():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "0"], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and not p.stdout.strip()
