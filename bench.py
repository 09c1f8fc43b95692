#!/usr/bin/env python
"""Benchmark of the hot path named by BASELINE.json: photon packets/s and Lucy-
iteration wall time on the 128^3 Cartesian grid.

  N = 1: configs[1] -- one central point source, single grey dust species, 1e8
         packets per Lucy iteration.
  N > 1: configs[2] -- the same grid, 1e9 packets per Lucy iteration sharded by
         packet id over the N GPUs (1e9 / N each), one RCCL all-reduce of the
         accumulator block per iteration; total work is fixed: "strong" scaling.

A "step" is one whole Lucy iteration (packet propagation kernel + accumulator
all-reduce when N>1 + update_energy_abs epilogue) over synthetic inputs already
resident in HBM.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


# L2<->fabric bytes per cell crossing from the committed rocprofv3 PMC passes of the 128^3
# uniform benchmark (FETCH_SIZE / WRITE_SIZE in KiB, separate --pmc runs):
#  persistent lucy_kernel<1> (profiles/r01b_summary.md): 8.08522e7 / 1.08239e8 KiB per launch of 3.46321e9 crossings
#  brick-tiled schedule, all tile_* kernels (profiles/r01d_summary.md): 4.13481e8 / 8.0876e8 KiB over 3 x 1.7319e10 crossings
PMC_B_PER_CROSSING = {
    0: (8.08522e7 + 1.08239e8) * 1024 / 3.46321e9,
    1: (4.13481e8 + 8.0876e8) * 1024 / (3 * 1.73190e10),
}


def cpu_baseline(prob, n_sample):
    """Oracle (CPU restatement) timed on this host's cores on a bounded sample
    of the same workload.  Test infrastructure used as the reported baseline,
    never as the product."""
    from oracle_lib import Oracle
    cores = os.cpu_count() or 1
    threads = min(cores, 64)         # per-thread accumulators: 64 x 16 MiB at 128^3
    orc = Oracle(prob)
    orc.lucy_iteration(min(n_sample, 20000), 1, n_threads=threads)   # warm-up (page faults, thread pool)
    t0 = time.time()
    _, st = orc.lucy_iteration(n_sample, 1, n_threads=threads)
    dt = time.time() - t0
    orc.close()
    return {"value": n_sample / dt, "unit": "packets/s", "cores": threads, "kind": "port",
            "sample": "%d packets of the same 128^3 workload, 1 Lucy iteration, %d OpenMP threads (%.1f s)" % (n_sample, threads, dt),
            "crossings_per_s": st["crossings"] / dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--grid", type=int, default=128)
    ap.add_argument("--photons", type=float, default=None,
                    help="packets per Lucy iteration PER GPU (default: 1e8 at --gpus 1 = configs[1]; 1e9 / N at --gpus N = configs[2])")
    ap.add_argument("--density", default="uniform")
    ap.add_argument("--cpu-sample", type=float, default=2e7)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--option", action="append", default=[], help="engine option name=value")
    args = ap.parse_args()

    import torch
    import hyperion_amd
    from hyperion_amd.benchmark import make_benchmark_problem
    from hyperion_amd.distributed import lucy_iteration_sharded

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:      # launched by torch.distributed.run: always exercise RCCL
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    if args.photons is None:
        n_total = 100000000 if world == 1 else 1000000000      # configs[1] / configs[2]
        config_name, scaling = ("configs[1]", "weak") if world == 1 else ("configs[2]", "strong")
    else:
        n_total = int(args.photons) * world
        config_name, scaling = "configs[1] shape", "weak"
    n_per_gpu = n_total // world
    prob = make_benchmark_problem(args.grid, density=args.density, n_photons=n_total, n_iter=args.steps)
    eng = hyperion_amd.Engine(prob, device=local_rank)
    for o in args.option:
        k, v = o.split("=")
        eng.set_option(k, int(v))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    it = 0
    for _ in range(args.warmup):
        it += 1
        lucy_iteration_sharded(eng, n_total, it, rank, world, want_output=False, force_collective=dist is not None)
    barrier()
    t0 = time.perf_counter()
    kernel_ms, crossings, finish_ms = [], 0, []
    for _ in range(args.steps):
        it += 1
        _, st = lucy_iteration_sharded(eng, n_total, it, rank, world, want_output=False, force_collective=dist is not None)
        a, b = eng.last_kernel_ms()
        kernel_ms.append(a)
        finish_ms.append(b)
        crossings = st["crossings"]          # whole-job crossings of the last step
    barrier()
    dt = time.perf_counter() - t0
    tiled = eng.get_option("last_lucy_mode") == 1
    t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    if rank == 0:
        n_dust = prob.n_dust
        k_ms = sum(kernel_ms) / len(kernel_ms)
        alg_bytes = 24.0 * n_dust * crossings / world      # per launch on this GPU
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        out = {
            "metric": "photon packets/sec, Lucy iteration, %d^3 Cartesian grid" % args.grid,
            "value": n_total * args.steps / dt, "unit": "packets/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %d^3 Cartesian, central 6000 K point source, grey isotropic dust (tau=1 centre-to-face, albedo 0.5), %g packets per Lucy iteration in total (%g per GPU), %s density"
                                   % (config_name, args.grid, n_total, n_per_gpu, args.density),
                       "packets_per_iteration": n_total, "parallelism": "packets sharded by id range over %d GPU(s), one f64 all-reduce per iteration" % world,
                       "crossings_per_packet": crossings / n_total},
            "lucy_kernel_ms": k_ms, "finish_ms": sum(finish_ms) / len(finish_ms),
            "lucy_schedule": "brick-tiled (tile_prepare/sort/tile_walk generations)" if tiled else "persistent kernel, global atomics",
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0,
                         # L2<->fabric bytes per launch from the committed PMC passes of this schedule/config
                         # (KiB units x1024, separate --pmc runs; the x2 wide-load correction of the guide does
                         # not apply to 8-byte scattered loads, one 64-B request each = TCC_EA0_RDREQ x 64).
                         "traffic": PMC_B_PER_CROSSING[1 if tiled else 0] * crossings / world / 1e9
                                    if args.grid == 128 and args.density == "uniform" else None,
                         "traffic_unit": "GB per launch (L2<->fabric; Infinity-Cache hits included, not HBM-only)",
                         "note": ("24 B x n_dust per cell crossing (8 B density load + 16 B accumulator RMW), crossings counted in-kernel; "
                                  "launch = the whole propagation of one iteration (all generations of tile_prepare/count/scan/scatter/walk on "
                                  "three streams), timed with HIP events on the engine's stream.  Density and accumulators of a 16^3 brick "
                                  "live in LDS, so the algorithmic bytes no longer go to memory: the dominant kernel tile_walk_kernel is "
                                  "VALU-issue bound (profiles/r01d_summary.md), %.2f x the memory-side atomic rate that bounds the "
                                  "persistent kernel (2.38e10 atomics/s, profiles/r01_atomic_rate_ubench.md)"
                                  % (crossings / world / (k_ms * 1e-3) / 2.38e10)) if tiled else
                                 ("24 B x n_dust per cell crossing (8 B density load + 16 B accumulator RMW), crossings counted in-kernel; "
                                  "the kernel is bound by the memory-side scattered-atomic rate (2.38e10/s measured, profiles/r01_atomic_rate_ubench.md): "
                                  "atomic-rate fraction %.2f" % (crossings / world / (k_ms * 1e-3) / 2.38e10))},
        }
        if world == 1 and not args.no_cpu_baseline:
            sample_prob = make_benchmark_problem(args.grid, density=args.density, n_photons=int(args.cpu_sample), n_iter=1)
            out["cpu_baseline"] = cpu_baseline(sample_prob, int(args.cpu_sample))
        print(json.dumps(out), flush=True)
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
