"""First contact of the multi-GPU path with real devices (images shard over ranks, ONE RCCL
broadcast of the weights, no collective in the loop).  The build container has no GPU and gpurun leases one, so
the two-GPU test SKIPS there -- it is collected by `pytest -m gpu` and runs the day a lease has two devices;
the one-GPU test proves on real hardware that two ranks mapped onto one device are refused."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env_extra=None, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                       text=True, timeout=timeout)
    return p, [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]


@pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2,
                    reason="needs two visible GPUs (RCCL, backend 'nccl')")
def test_two_ranks_over_rccl():
    p, lines = _bench(["--gpus", "2", "--steps", "5", "--warmup", "2", "--no-secondary", "--no-cpu-baseline",
                       "--no-fp32-leg"])
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1
    r = lines[0]
    assert r["n_gpus"] == 2 and r["rccl_ranks_seen"] == 2 and r["backend"].startswith("nccl")
    assert len({d["uuid"] for d in r["rank_devices"]}) == 2, r["rank_devices"]
    assert r["weight_broadcast_ms"] > 0 and r["weight_broadcast_bytes"] > 40e6
    assert r["cross_rank_agreement"]["ok"]
    assert len(r["per_rank_img_s"]) == 2 and min(r["per_rank_img_s"]) > 0.5 * max(r["per_rank_img_s"])
    assert r["config"]["global_batch"] == 64


@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
def test_two_ranks_on_one_gpu_are_refused():
    """BENCH_DEVICE=0 maps both ranks onto cuda:0 (over gloo: RCCL itself would refuse the duplicate): the
    identity check must stop the run before any number is printed."""
    p, lines = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--res", "128",
                       "--no-secondary", "--no-cpu-baseline", "--no-fp32-leg"],
                      {"BENCH_DEVICE": "0", "BENCH_BACKEND": "gloo"})
    assert p.returncode != 0
    assert not lines
    assert "two ranks share a GPU" in (p.stderr + p.stdout)
