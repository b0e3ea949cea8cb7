"""`python bench.py --gpus N` must work as typed (no torch.distributed.run around it): the command starts its N ranks as children
before it touches the GPU.  Here, without a GPU, every rank stops with the product's "needs a GPU" message — which shows that
the ranks were started with the right environment and that a failing rank makes the command fail."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_starts_its_ranks_and_propagates_their_failure():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present: tests/test_gpu_scale_parity.py runs the real two-rank bench")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--games", "64", "--steps", "1",
                        "--warmup", "0"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs a GPU") >= 2, r.stderr[-2000:]      # both ranks ran main() under WORLD_SIZE=2
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]     # no result line from a failed run
