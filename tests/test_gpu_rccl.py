"""The multi-GPU path on hardware at world size 1: torch.distributed "nccl" (= RCCL) is initialised by
``torch.distributed.run`` exactly as the driver launches ``bench.py --gpus N``, and the collective really runs on the
zero-copy alias of the engine's accumulator block (the CPU suite covers world size 2 with gloo and the oracle)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _torchrun(args, port):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(port)] + args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)


def test_bench_under_torchrun_world_size_1():
    r = _torchrun(["bench.py", "--gpus", "1", "--photons", "2e6", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"], 29613)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["unit"] == "packets/s" and d["value"] > 1e7
    assert d["config"]["packets_per_iteration"] == 2000000
    assert "roofline" in d and d["roofline"]["achieved"] > 0


def test_sharded_iterations_with_real_all_reduce():
    r = _torchrun([os.path.join("tests", "rccl_ws1_check.py")], 29614)
    assert r.returncode == 0 and "RCCL_WS1_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
