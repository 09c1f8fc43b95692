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


def _torchrun(args, port, nproc=1, prefix=(), **extra_env):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    cmd = list(prefix) + [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
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


def test_bench_at_world_size_2_on_one_gpu_through_gloo():
    """The N > 1 path of bench.py end to end with two processes and real kernels -- id ranges per rank, the all-reduce of the device
    block (gloo here: one GPU cannot hold two RCCL ranks), the epilogue on every rank, the self-checks of the line: ranks seen, one digest
    of the specific energy on all ranks, per-rank times."""
    r = _torchrun(["bench.py", "--gpus", "2", "--grid", "64", "--photons", "2e6", "--steps", "2", "--warmup", "1"], 29615, nproc=2,
                  HYP_BENCH_BACKEND="gloo")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["packets_per_iteration"] == 4000000 and d["value"] > 1e6
    assert d["rccl"]["ranks_seen"] == 2 and d["rccl"]["world_size"] == 2 and d["rccl"]["backend"] == "gloo"
    assert d["rccl"]["specific_energy_identical_on_all_ranks"] is True
    assert len(d["per_rank_ms_per_step"]["launch_ms"]["ranks"]) == 2
    assert "cpu_baseline" not in d and "extra" not in d          # N = 1 only


def test_bench_at_world_size_8_on_one_gpu_inside_16_cpus():
    """configs[2]'s process layout under the GPU box's real constraint: EIGHT ranks -- eight host-driven generation
    loops, eight engines, the all-reduce and the epilogue on every rank -- pinned to at most 16 CPUs (the cgroup quota of the pool's
    boxes), on the one GPU through gloo.  The line must show 8 ranks seen, one digest of the specific energy on all ranks, id ranges
    that tile [0, N) exactly, and per-rank host times whose spread stays bounded (nobody starved)."""
    import shutil
    cpus = sorted(os.sched_getaffinity(0))[:16]
    prefix = ["taskset", "-c", ",".join(map(str, cpus))] if shutil.which("taskset") else []
    r = _torchrun(["bench.py", "--gpus", "8", "--photons", "2e6", "--steps", "2", "--warmup", "1"], 29616, nproc=8, prefix=prefix,
                  HYP_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    n = 8 * 2000000
    assert d["n_gpus"] == 8 and d["config"]["packets_per_iteration"] == n and d["value"] > 1e6
    assert d["rccl"]["ranks_seen"] == 8 and d["rccl"]["world_size"] == 8
    assert d["rccl"]["specific_energy_identical_on_all_ranks"] is True
    shards = d["rccl"]["shards"]
    assert len(shards) == 8 and shards[0][0] == 0 and sum(s[1] for s in shards) == n
    assert all(shards[i][0] + shards[i][1] == shards[i + 1][0] for i in range(7))
    launch = d["per_rank_ms_per_step"]["launch_ms"]["ranks"]
    assert len(launch) == 8 and max(launch) < 4.0 * min(launch) + 50.0, launch          # eight loops share the CPUs: nobody starves
    assert d["d2h_ms"] > 0
