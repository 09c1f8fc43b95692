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


PEAK_HBM_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak
N_SIMD, CLOCK_HZ = 1024, 2.4e9  # 256 CUs x 4 SIMDs, peak engine clock; one wave64 VALU instruction issues in 4 cycles


def committed_pmc(tiled):
    """Per-crossing PMC totals of the committed rocprofv3 passes of THIS bench command (profiles/r02_pmc.json, written by
    tools/r02_profile.sh + tools/summarize_tiled.py on the GPU box).  Counters cannot be read inside the timed run, so the
    bench line carries them with their source; None when the file is not there or does not apply."""
    path = os.path.join(ROOT, "profiles", "r02_pmc.json")
    if not tiled or not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)


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


def extras(n):
    """The other single-GPU configurations of BASELINE.json, after the timed region, as extra entries of the same JSON line
    (not `value`): configs[3] = adaptive octree + peel-off imaging to a 512 x 512 Stokes image, configs[4] = the real
    100 000-site voro++ tessellation with two anisotropic polarising species and an external source.  One warm-up and one
    timed pass each; unit of work = cell crossing (24 B x n_dust algorithmic)."""
    import hyperion_amd
    from hyperion_amd.benchmark import make_octree_problem
    res = []

    def timed(fn):
        fn()
        t0 = time.perf_counter()
        st = fn()
        return st, time.perf_counter() - t0

    p = make_octree_problem(max_level=7, n_photons=n, n_iter=1)
    e = hyperion_amd.Engine(p)
    state = {"it": 0}

    def lucy():
        state["it"] += 1
        return e.lucy_iteration(n, state["it"], want_output=False)[1]
    st, dt = timed(lucy)
    k_ms = e.last_kernel_ms()[0]
    res.append({"config": "configs[3] Lucy iteration: octree depth 7 (%d cells), central source" % p.n_cells, "packets": n,
                "packets_per_s": n / dt, "crossings_per_s": st["crossings"] / (k_ms * 1e-3), "kernel_ms": k_ms,
                "hbm_frac": 24.0 * st["crossings"] / (k_ms * 1e-3) / 1e9 / PEAK_HBM_GBS})
    st, dt = timed(lambda: e.final_iteration(n)[1])
    rounds = e.get_option("last_defer_rounds")
    res.append({"config": "configs[3] imaging iteration: peel-off to a 512x512 Stokes image, 1 view, forced first interaction", "packets": n,
                "schedule": ("deferred peel-off (hyp_defer.h), %d rounds, %.2f events/packet" % (rounds, e.get_option("last_defer_events") / n))
                            if rounds else "inline peel-off",
                "packets_per_s": n / dt, "kernel_ms": e.last_kernel_ms()[0],
                "crossings_per_s": st["crossings"] / dt, "hbm_frac": 24.0 * st["crossings"] / dt / 1e9 / PEAK_HBM_GBS})
    e.close()
    try:
        from cases import voronoi_big_problem
        p = voronoi_big_problem(n_photons=n)
    except Exception as ex:           # the tessellation fixture travels with tests/golden
        res.append({"config": "configs[4]", "error": str(ex)})
        return res
    e = hyperion_amd.Engine(p)
    state["it"] = 0

    def lucy4():
        state["it"] += 1
        return e.lucy_iteration(n, state["it"], want_output=False)[1]
    st, dt = timed(lucy4)
    k_ms = e.last_kernel_ms()[0]
    res.append({"config": "configs[4] Lucy iteration: voro++ tessellation of 100000 random sites (15.2 neighbours / cell), 2 HG-like polarising "
                          "species, point + external box source", "packets": n, "packets_per_s": n / dt,
                "crossings_per_s": st["crossings"] / (k_ms * 1e-3), "kernel_ms": k_ms,
                "hbm_frac": 24.0 * 2 * st["crossings"] / (k_ms * 1e-3) / 1e9 / PEAK_HBM_GBS})
    e.close()
    return res


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
    ap.add_argument("--no-extras", action="store_true", help="skip the configs[3] / configs[4] lines (N = 1 only)")
    ap.add_argument("--extra-photons", type=float, default=2e7, help="packets per iteration of the extra configurations")
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
            "lucy_schedule": ("brick-tiled: generations of tile_interact / tile_emit / tile_scan / tile_scatter / tile_walk on %d slot pools (streams)"
                              % eng.get_option("tile_pools")) if tiled else "persistent kernel, global atomics",
        }
        std_case = args.grid == 128 and args.density == "uniform"
        pmc = committed_pmc(tiled) if std_case else None
        per = pmc["per_crossing"] if pmc else {}
        xs = crossings / world                      # crossings of one launch on this GPU
        roof = {"bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": achieved / PEAK_HBM_GBS,
                "kernel": "whole propagation of one Lucy iteration (all generations), HIP events on the engine's stream",
                "algorithmic_bytes_per_unit": "24 B x n_dust per cell crossing (8 B density load + 16 B accumulator read-modify-write, "
                                              "src/grid/grid_propagate_3d.f90:131-160); crossings counted in-kernel"}
        if per:
            # FETCH_SIZE / WRITE_SIZE are in KiB; L2 <-> fabric requests, Infinity-Cache hits included.  The guide's x2
            # correction for wide streaming reads applies to the 16 B/lane record loads of tile_walk / tile_interact, so
            # the read half is a lower bound: both are given.
            rd, wr = per["FETCH_SIZE"] * 1024.0, per["WRITE_SIZE"] * 1024.0
            roof["traffic"] = (rd + wr) * xs / 1e9
            roof["traffic_unit"] = "GB per launch, L2<->fabric (FETCH_SIZE + WRITE_SIZE as reported)"
            roof["traffic_reads_x2"] = (2 * rd + wr) * xs / 1e9
            roof["traffic_source"] = "profiles/r02_pmc.json (rocprofv3 --pmc passes of this command, tools/r02_profile.sh); not measured in this run"
        else:
            roof["traffic"] = None
        if tiled:
            walk_ms = eng.get_option("last_walk_us") / 1e3
            n_walk = eng.get_option("last_walk_launches")
            roof["dominant_kernel"] = {
                "name": "tile_walk_kernel", "launches_per_iteration": n_walk, "sum_ms_per_iteration": walk_ms,
                "avg_launch_us": walk_ms * 1e3 / max(n_walk, 1),
                "achieved_GBs_over_its_own_time": alg_bytes / (walk_ms * 1e-3) / 1e9 if walk_ms > 0 else None,
                "note": "HIP events around every tile_walk launch on its pool's stream (last timed step); with several pools the "
                        "launches overlap other kernels and stretch -- profiles/r02_serial_summary.md has the one-pool trace"}
            roof["note"] = ("density and accumulators of a 16^3 brick live in LDS, so the 24 B per crossing never go to memory: the HBM "
                            "fraction says how far the path is from a streaming bound it does not have; what limits it is VALU issue "
                            "(see issue_roofline) and, for tile_interact, random 128-byte record traffic")
        else:
            roof["note"] = ("bound by the memory-side scattered-atomic rate (2.38e10/s, profiles/r01_atomic_rate_ubench.md): fraction %.2f"
                            % (xs / (k_ms * 1e-3) / 2.38e10))
        out["roofline"] = roof
        if per:
            # second ceiling: VALU issue.  wave-instructions x 4 cycles / (SIMDs x clock) is the time the chip needs to issue
            # the vector instructions of one launch if every SIMD issued one every cycle it could.
            t_issue = per["SQ_INSTS_VALU"] * xs * 4.0 / (N_SIMD * CLOCK_HZ)
            out["issue_roofline"] = {"bound": "valu_issue", "valu_wave_instructions_per_crossing": per["SQ_INSTS_VALU"],
                                     "ideal_issue_ms": t_issue * 1e3, "measured_ms": k_ms, "frac": t_issue / (k_ms * 1e-3),
                                     "peak": "%d SIMDs x %.1f GHz / 4 cycles per wave64 instruction" % (N_SIMD, CLOCK_HZ / 1e9),
                                     "source": "profiles/r02_pmc.json (SQ_INSTS_VALU summed over all tile_* kernels)"}
        if world == 1 and not args.no_cpu_baseline:
            sample_prob = make_benchmark_problem(args.grid, density=args.density, n_photons=int(args.cpu_sample), n_iter=1)
            out["cpu_baseline"] = cpu_baseline(sample_prob, int(args.cpu_sample))
        eng.close()
        eng = None
        if world == 1 and std_case and not args.no_extras:
            out["extra"] = extras(int(args.extra_photons))
        print(json.dumps(out), flush=True)
    if eng is not None:
        eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
