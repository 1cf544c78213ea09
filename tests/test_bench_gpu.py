"""GPU tier: bench.py's multi-rank path, launched the way the driver launches it (`python bench.py --gpus N`, no
external launcher).  On the one-GPU box the RCCL calls are exercised with a single rank (--force-dist) and the rank
logic with two gloo ranks sharing the GPU (--rehearse-gloo); the N > 1 RCCL run itself is the driver's."""
import json
import os
import subprocess
import sys

import pytest

from testlib import REPO

pytestmark = pytest.mark.gpu
SMALL = ["--steps", "2", "--warmup", "1", "--frames", "6", "--height", "256", "--width", "512", "--cpu-frames", "0"]


def _bench(*flags):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *flags, *SMALL], env=env, capture_output=True,
                         text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = os.path.join(REPO, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "bench_dist_" + "_".join(f.strip("-") for f in flags) + ".json"), "w") as fh:
        fh.write(lines[0] + "\n")
    return json.loads(lines[0])


def test_single_rank_rccl_gather():
    """one rank, backend nccl (= RCCL): init_process_group, the asynchronous gather of packed bits overlapped with the
    next step's kernels, barrier, all_reduce - the calls of the N > 1 path"""
    r = _bench("--gpus", "1", "--force-dist")
    assert r["n_gpus"] == 1 and r["gather_ok"] is True and r["payload_bit_errors"] == 0
    assert "rccl" in r["config"]["collective"]


def test_two_rank_gloo_rehearsal_self_launched():
    """`python bench.py --gpus 2 --rehearse-gloo` starts its own two ranks (they share GPU 0), gathers in rank order"""
    r = _bench("--gpus", "2", "--rehearse-gloo")
    assert r["n_gpus"] == 2 and r["gather_ok"] is True and r["payload_bit_errors"] == 0
    assert r["scaling"] == "weak" and "rehearsal" in r["data"] and "gloo" in r["config"]["collective"]
