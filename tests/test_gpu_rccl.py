"""-m gpu: the multi-process path on hardware.  Each test starts FRESH children with torch.distributed.run (one process
per GPU, backend "nccl" = RCCL, rendezvous on 127.0.0.1) -- never an exec from this process, which has touched the GPU --
with as many ranks as GPUs are visible, capped at 2."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(script_and_args, nproc):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(port)] + script_and_args
    return subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)


def _ranks():
    return max(1, min(2, torch.cuda.device_count()))


def test_bench_runs_under_the_launcher_and_reduces_over_rccl():
    n = _ranks()
    r = _launch([os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1", "--no-decode",
                 "--no-wide", "--no-cpu-baseline"], n)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["steps"] == 3 and out["scaling"] == "weak"
    assert "RCCL all_reduce" in out["config"]["api"]
    assert out["value"] > 0 and out["roofline"]["frac"] > 0
    # the shipped module's own leg (ShardedCTCLoss + backward, one synchronous all-reduce per call) beside the C-ABI leg
    assert out["sharded_module_ms_per_step"] > 0 and "ShardedCTCLoss" in out["sharded_module_api"]
    # weak scaling: every rank has its own 256 utterances
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 - n * 256 * 1000) <= 1e-6 * n * 256 * 1000


def test_sharded_loss_on_the_real_kernels_matches_the_unsharded_module():
    n = _ranks()
    r = _launch([os.path.join(ROOT, "tests", "rccl_worker.py")], n)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    assert "RCCL_OK world=%d" % n in r.stdout
