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

# The CPU baseline's OpenMP threads: one per physical core, spread over the sockets and pinned (read by libgomp when it is loaded, so
# set before anything imports it).  The GPU path does not use host threads.
def cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max or v1 cfs quota), or None without a limit: a box can show 256
    logical CPUs and allow 16 of them at a time -- threads beyond the quota are throttled, not run."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
            return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


if cpu_quota() is None:      # the whole machine is ours: pin; under a quota on a shared host the scheduler places the few threads better
    os.environ.setdefault("OMP_PROC_BIND", "spread")
    os.environ.setdefault("OMP_PLACES", "cores")

HOST_CPUS = sorted(os.sched_getaffinity(0))      # taken now: with the variables above libgomp pins this (main) thread to its first place when it loads

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


PEAK_HBM_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak
N_SIMD, CLOCK_HZ = 1024, 2.4e9  # 256 CUs x 4 SIMDs, peak engine clock
# SIMD cycles per wave64 instruction at 4 waves per SIMD, measured with tools/ubench/valu_issue.hip (profiles/r03_tiled_log.md):
# FP64 add / mul / fma / compare / ldexp and 64-bit integer 4.2-4.9 -> 4.3; 32-bit 2.2-2.6 -> 2.4; v_rcp_f64 16
CYC_F64, CYC_32, CYC_TRANS_F64 = 4.3, 2.4, 16.0
PMC_FILE = "profiles/r06_pmc.json"


def committed_pmc(workload):
    """Counter sums of the committed rocprofv3 passes of one workload (profiles/r06_pmc.json, written on the GPU box by
    tools/profile_round.sh + tools/summarize_profile_round.py: `car` = THIS bench command, `oct_lucy` / `oct_img` / `vor` = the extra
    configurations at 1e8 packets).  Counters cannot be read inside the timed run, so the bench line carries them with
    their source; None when the file is not there."""
    path = os.path.join(ROOT, PMC_FILE)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f).get(workload)


def host_cpu():
    try:
        with open("/proc/cpuinfo") as f:
            for l in f:
                if l.startswith("model name"):
                    return l.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def traffic_fields(pmc):
    """bytes per crossing (L2 <-> fabric, FETCH_SIZE / WRITE_SIZE in KiB as reported; reads also with the guide's x2 for wide
    streaming loads) and their ratio to the algorithmic 24 B x n_dust, from a workload entry of profiles/r06_pmc.json."""
    if not pmc or "bytes_per_crossing" not in pmc:
        return {"bytes_per_crossing": None}
    b = pmc["bytes_per_crossing"]
    out = {"bytes_per_crossing": {"fetched": b["fetched"], "fetched_reads_x2": b["fetched_x2"], "written": b["written"], "algorithmic": b["algorithmic"]},
           "traffic_over_algorithmic": b["traffic_over_algorithmic"], "traffic_over_algorithmic_reads_x2": b["traffic_over_algorithmic_reads_x2"],
           "pmc_source": PMC_FILE}
    per = pmc.get("per_crossing", {})
    if "TCC_EA0_ATOMIC_sum" in per:
        out["memory_side_atomics_per_crossing"] = per["TCC_EA0_ATOMIC_sum"]
    return out


def physical_cores():
    """Distinct (socket, core) pairs of /proc/cpuinfo that this process may run on; half the logical CPUs if that cannot be read."""
    allowed = set(HOST_CPUS)
    try:
        seen, cpu, phys = set(), None, None
        with open("/proc/cpuinfo") as f:
            for l in f:
                if l.startswith("processor"):
                    cpu, phys = int(l.split(":")[1]), None
                elif l.startswith("physical id"):
                    phys = int(l.split(":")[1])
                elif l.startswith("core id") and cpu in allowed:
                    seen.add((phys, int(l.split(":")[1])))
        if seen:
            return len(seen)
    except (OSError, ValueError):
        pass
    return max(1, len(allowed) // 2)


def cpu_baseline(prob, n_sample):
    """Oracle (CPU restatement) timed on this host on a bounded sample of the same workload: on as many threads as the host gives
    this process -- physical cores, or the container's CPU quota when there is one (the GPU boxes of this pool show 256 logical
    CPUs and allow 16) --, on four times as many (what oversubscription does) and on ONE thread; without a quota the threads are
    pinned one per core and spread over the sockets; per-thread accumulators are first touched by their own thread.  Test
    infrastructure used as the reported baseline, never as the product."""
    from oracle_lib import Oracle
    logical = len(HOST_CPUS)
    cores = physical_cores()
    quota = cpu_quota()
    usable = cores if quota is None else max(1, min(cores, int(quota + 0.5)))
    orc = Oracle(prob)

    def timed(n, threads):
        orc.lucy_iteration(min(n, 2000 * threads), 1, n_threads=threads)   # warm-up (page faults of the accumulators, thread pool)
        t0 = time.time()
        _, st = orc.lucy_iteration(n, 1, n_threads=threads)
        dt = time.time() - t0
        return {"value": n / dt, "unit": "packets/s", "cores": threads, "crossings_per_s": st["crossings"] / dt,
                "sample": "%d packets of the same workload, 1 Lucy iteration, %d OpenMP thread(s) (%.1f s)" % (n, threads, dt)}
    one = timed(max(min(n_sample // 80, 250000), 1000), 1)
    # (the sample scales with the threads so that each leg stays near ten seconds whatever the host)
    per_thread = max(min(n_sample // 16, 400000), 2000)
    legs = {"1": one, str(usable): timed(min(n_sample, per_thread * usable), usable)}
    if usable < cores:      # what oversubscribing the quota gives (round 4 ran 64 threads on a box that allows 16 CPUs)
        more = min(cores, 4 * usable)
        legs[str(more)] = timed(min(n_sample, per_thread * usable), more)
    orc.close()
    best = max(legs.values(), key=lambda l: l["value"])
    out = dict(best)
    out.update({"kind": "port", "host_cpu": host_cpu(), "host_logical_cpus": logical, "host_physical_cores": cores,
                "one_core": one, "threads": legs, "scaling_over_one_core": best["value"] / one["value"],
                "host_cpu_quota": quota, "host_load_average": os.getloadavg()[0],
                "omp": {k: os.environ.get(k) for k in ("OMP_PROC_BIND", "OMP_PLACES")},
                "note": "the CPU restatement (oracle/hyp_oracle.c), not the Fortran: the reference needs its absent fortranlib submodule to build "
                        "(DESIGN.md section 6 has the survey's probe of the reference's own geometry loop for calibration); every thread deposits "
                        "into its own 16 MiB accumulator copy, reduced in parallel at the end; threads = min(physical cores, the container's CPU quota) -- the box shows "
                        "host_logical_cpus and lets the container use host_cpu_quota of them at a time; the value is the best of the thread counts tried"})
    return out


def extras(n, passes=3):
    """The other single-GPU configurations of BASELINE.json, after the timed region, as extra entries of the same JSON line
    (not `value`): configs[3] = adaptive octree + peel-off imaging to a 512 x 512 Stokes image, configs[4] = the real
    100 000-site voro++ tessellation with two anisotropic polarising species and an external source.  One warm-up and
    `passes` timed passes each (mean, and every pass listed).  Unit of work = cell crossing.  Algorithmic bytes (SURVEY
    section 8d): a Lucy iteration moves 24 B x n_dust per crossing; an imaging iteration deposits nothing on its crossings --
    8 B x n_dust per crossing (density read of grid_integrate_noenergy / grid_escape_tau) + 16 B x n_stokes per binned
    peel-off event (event x view)."""
    import hyperion_amd
    from hyperion_amd.benchmark import make_octree_problem
    res = []

    def timed(fn):
        fn()
        dts, st = [], None
        for _ in range(passes):
            t0 = time.perf_counter()
            st = fn()
            dts.append(time.perf_counter() - t0)
        return st, sum(dts) / len(dts), dts

    def common(st, n, dt, dts, k_ms, alg_bytes):
        return {"packets": n, "passes": len(dts), "packets_per_s": n / dt, "pass_ms": [x * 1e3 for x in dts], "kernel_ms": k_ms,
                "crossings_per_s": st["crossings"] / (k_ms * 1e-3), "crossings_per_packet": st["crossings"] / n,
                "killed_geo": int(st["killed_geo"]), "killed_int": int(st["killed_int"]),
                "algorithmic_GB": alg_bytes / 1e9, "hbm_frac": alg_bytes / (k_ms * 1e-3) / 1e9 / PEAK_HBM_GBS}

    def pmc_fields(name, alg_per_crossing):
        f = traffic_fields(committed_pmc(name))
        b = f.get("bytes_per_crossing")
        if b:       # the committed counters are per crossing: restate their ratio against THIS entry's algorithmic bytes
            b["algorithmic"] = alg_per_crossing
            f["traffic_over_algorithmic"] = (b["fetched"] + b["written"]) / alg_per_crossing
            f["traffic_over_algorithmic_reads_x2"] = (b["fetched_reads_x2"] + b["written"]) / alg_per_crossing
        return f

    for label, pos in (("central source on a vertex of the tree (BASELINE's config)", (0.0, 0.0, 0.0)),
                       ("the same tree, source moved off the vertex to (0.013, 0.007, -0.011) pc", (0.013 * 3.08568025e18, 0.007 * 3.08568025e18, -0.011 * 3.08568025e18))):
        p = make_octree_problem(max_level=7, n_photons=n, n_iter=1, source_position=pos)
        e = hyperion_amd.Engine(p)
        state = {"it": 0}

        def lucy():
            state["it"] += 1
            return e.lucy_iteration(n, state["it"], want_output=False)[1]
        st, dt, dts = timed(lucy)
        k_ms = e.last_kernel_ms()[0]
        res.append({"config": "configs[3] Lucy iteration: octree depth 7 (%d cells), %s" % (p.n_cells, label),
                    "schedule": "cluster-tiled (hyp_otile.h), %d clusters" % e.get_option("ot_clusters") if e.get_option("last_lucy_mode") == 1 else "persistent kernel",
                    **common(st, n, dt, dts, k_ms, 24.0 * st["crossings"]), **pmc_fields("oct_lucy", 24.0)})
        st, dt, dts = timed(lambda: e.final_iteration(n)[1])
        rounds = e.get_option("last_defer_rounds")
        k_ms = e.last_kernel_ms()[0]
        events = e.get_option("last_defer_events") if rounds else 0
        n_view, n_stokes = 1, 4
        alg = 8.0 * st["crossings"] + 16.0 * n_stokes * events * n_view
        ent = {"config": "configs[3] imaging iteration: peel-off to a 512x512 Stokes image, 1 view, forced first interaction; %s" % label,
               "schedule": ("deferred peel-off (hyp_defer.h), %d rounds, %.2f events/packet%s" % (rounds, events / n,
                            ", emission + forced first interaction ahead of the rounds (ff_walk_kernel)" if e.get_option("last_ff_prepass") else "")
                            + (", propagation half on the slot-pool schedule (tile_interact / tile_emit <IMG>, LDS walk)" if e.get_option("last_tiled_imaging") else "")
                            + (", direct light of the point source walked once per view (direct_column_kernel)" if e.get_option("last_direct_memo") else ""))
                           if rounds else "inline peel-off",
               **common(st, n, dt, dts, k_ms, alg),
               "algorithmic_bytes": "8 B x n_dust per crossing (no deposit in the imaging iteration) + 16 B x 4 Stokes x %d binned events (events x views)" % (events * n_view)}
        if st["crossings"]:
            ent.update(pmc_fields("oct_img", alg / st["crossings"]))
        res.append(ent)
        e.close()
    try:
        from cases import voronoi_big_problem
        p = voronoi_big_problem(n_photons=n)
    except Exception as ex:           # the tessellation fixture travels with tests/golden
        res.append({"config": "configs[4]", "error": str(ex)})
        return res
    e = hyperion_amd.Engine(p)
    state = {"it": 0}

    def lucy4():
        state["it"] += 1
        return e.lucy_iteration(n, state["it"], want_output=False)[1]
    st, dt, dts = timed(lucy4)
    k_ms = e.last_kernel_ms()[0]
    res.append({"config": "configs[4] Lucy iteration: voro++ tessellation of 100000 random sites (15.2 neighbours / cell), 2 HG-like polarising "
                          "species, point + external box source",
                "schedule": "cluster-tiled (hyp_vtile.h), %d clusters" % e.get_option("vt_clusters") if e.get_option("last_lucy_mode") == 1 else "persistent kernel",
                **common(st, n, dt, dts, k_ms, 24.0 * 2 * st["crossings"]), **pmc_fields("vor", 48.0)})
    e.close()
    res += other_iterations(passes)
    return res


def other_iterations(passes=3):
    """Iterations no BASELINE config names, so that the driver's own run carries them (profiles/r05_other_iterations.md,
    tools/mono_bench.py, tools/yso_probe.py have the same workloads): the monochromatic final iteration on a 64^3 Cartesian grid
    (5 wavelengths x source + thermal packets), and what an AnalyticalYSOModel-like run costs -- a 400 x 200 spherical polar grid lit by
    a star WITH A RADIUS, Lucy iteration and imaging iteration (SEDs for two views).  Kernel time from HIP events, mean of `passes`."""
    import numpy as np
    import hyperion_amd
    from hyperion_amd.benchmark import LSUN, PC, make_benchmark_problem
    from hyperion_amd.problem import PeeledImages, Source
    res = []
    try:
        n = 20_000_000
        p = make_benchmark_problem(64, tau=1.0)
        p.config.monochromatic = True
        p.config.frequencies = 2.99792458e14 / np.array([1.0, 3.0, 10.0, 30.0, 100.0])
        p.peeled = [PeeledImages(theta=[45.0], phi=[45.0], n_x=256, n_y=256, x_min=-1.5 * PC, x_max=1.5 * PC, y_min=-1.5 * PC, y_max=1.5 * PC,
                                 n_ap=1, ap_min=3 * PC, ap_max=3 * PC, n_wav=5, wav_min=1.0, wav_max=100.0, inu_min=1, inu_max=5)]
        e = hyperion_amd.Engine(p)
        e.lucy_iteration(10_000_000, 1, want_output=False)
        e.mono_iteration(n // 100, n // 100)
        ms, st = [], None
        for _ in range(passes):
            _, st = e.mono_iteration(n // 10, n // 10)
            ms.append(e.last_kernel_ms()[0])
        k = sum(ms) / len(ms)
        res.append({"config": "monochromatic final iteration: 64^3 Cartesian grid, 5 wavelengths x (source + thermal packets), one 256^2 image",
                    "schedule": "deferred peel-off, one launch per (part, wavelength)" if e.get_option("last_mono_deferred") else "general kernel, inline peel-off",
                    "packets": n, "kernel_ms": k, "pass_ms": ms, "packets_per_s": n / k * 1e3, "crossings_per_packet": st["crossings"] / n,
                    "crossings_per_s": st["crossings"] / k * 1e3})
        e.close()
    except Exception as ex:
        res.append({"config": "monochromatic final iteration", "error": str(ex)})
    try:
        from test_gpu_polar import config0_problem
        n = 10_000_000
        p = config0_problem(n_r=400, n_t=200, tau=3.0, log_r=True, peeled=True)
        p.sources = [Source(type="sphere", luminosity=LSUN, temperature=6000.0, position=(0.0, 0.0, 0.0), radius=0.002 * PC)]
        e = hyperion_amd.Engine(p)
        e.lucy_iteration(n, 1, want_output=False)       # (full size, like --warmup: the slot pool is allocated and touched here, as in a run's first iteration)
        ms, st = [], None
        for i in range(passes):
            _, st = e.lucy_iteration(n, 2 + i, want_output=False)
            ms.append(e.last_kernel_ms()[0])
        k = sum(ms) / len(ms)
        res.append({"config": "spherical polar grid 400 x 200 (log r, cavity), star with a radius: Lucy iteration",
                    "schedule": "brick-tiled (hyp_ptile.h)" if e.get_option("last_lucy_mode") == 1 else "persistent kernel",
                    "packets": n, "kernel_ms": k, "pass_ms": ms, "packets_per_s": n / k * 1e3, "crossings_per_packet": st["crossings"] / n,
                    "crossings_per_s": st["crossings"] / k * 1e3})
        e.final_iteration(n // 10)
        ms = []
        for _ in range(passes):
            _, st = e.final_iteration(n)
            ms.append(e.last_kernel_ms()[0])
        k = sum(ms) / len(ms)
        res.append({"config": "... imaging iteration: peeled SEDs for two views, forced first interaction",
                    "schedule": (("deferred peel-off with the general-source kernels (hyp_defer.h: GEN), %d rounds" % e.get_option("last_defer_rounds"))
                                 + (", propagation half on the slot-pool schedule (GEN instances of tile_emit / tile_interact <IMG>, brick walk)" if e.get_option("last_tiled_imaging") else ""))
                                if e.get_option("last_defer_rounds") else "general kernel, inline peel-off",
                    "packets": n, "kernel_ms": k, "pass_ms": ms, "packets_per_s": n / k * 1e3, "crossings_per_packet": st["crossings"] / n,
                    "crossings_per_s": st["crossings"] / k * 1e3})
        e.close()
    except Exception as ex:
        res.append({"config": "spherical polar grid, star with a radius", "error": str(ex)})
    try:
        from hyperion_amd.benchmark import make_cyl_disc_problem
        n = 20_000_000
        e = hyperion_amd.Engine(make_cyl_disc_problem(peeled=True))
        e.lucy_iteration(n, 1, want_output=False)
        ms, st = [], None
        for i in range(passes):
            _, st = e.lucy_iteration(n, 2 + i, want_output=False)
            ms.append(e.last_kernel_ms()[0])
        k = sum(ms) / len(ms)
        res.append({"config": "cylindrical polar grid 400 x 200 (flared disc, log w), central point source: Lucy iteration",
                    "schedule": "brick-tiled (hyp_ptile.h), the flight's reciprocals in the wall search" if e.get_option("last_lucy_mode") == 1 else "persistent kernel",
                    "packets": n, "kernel_ms": k, "pass_ms": ms, "packets_per_s": n / k * 1e3, "crossings_per_packet": st["crossings"] / n,
                    "crossings_per_s": st["crossings"] / k * 1e3})
        m = n // 4
        e.final_iteration(m // 10)
        ms = []
        for _ in range(passes):
            _, st = e.final_iteration(m)
            ms.append(e.last_kernel_ms()[0])
        k = sum(ms) / len(ms)
        res.append({"config": "... imaging iteration: peeled SEDs for two views, forced first interaction",
                    "schedule": ("deferred peel-off, %d rounds" % e.get_option("last_defer_rounds")) if e.get_option("last_defer_rounds") else "inline peel-off",
                    "packets": m, "kernel_ms": k, "pass_ms": ms, "packets_per_s": m / k * 1e3, "crossings_per_packet": st["crossings"] / m,
                    "crossings_per_s": st["crossings"] / k * 1e3})
        e.close()
    except Exception as ex:
        res.append({"config": "cylindrical polar grid, flared disc", "error": str(ex)})
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
    ap.add_argument("--extra-photons", type=float, default=1e8, help="packets per iteration of the extra configurations (BASELINE: 1e8)")
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
    # (HYP_BENCH_BACKEND=gloo: the N > 1 path of this script on a box with ONE GPU -- every rank on device local_rank modulo the
    # device count, the all-reduce through gloo on the same device tensors; tests/test_gpu_rccl.py.  The driver's runs use RCCL.)
    backend = os.environ.get("HYP_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    elif local_rank >= torch.cuda.device_count():       # (never wrap RCCL ranks onto shared devices)
        raise SystemExit("rank %d: LOCAL_RANK %d but only %d GPUs are visible -- an N-GPU line needs N devices" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:      # launched by torch.distributed.run: always exercise RCCL
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    rccl = None
    if dist is not None:
        # RCCL writes a version banner to STDOUT when the first communicator is created (the all-reduce below): sent to stderr, so
        # that the JSON line is the only thing this command prints
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        # What RCCL itself saw, so that the line cannot claim N GPUs on the strength of WORLD_SIZE alone: an all-reduce of ones,
        # the process group's size, and every rank's device (index, PCI bus id) gathered through the same backend.
        ones = torch.ones(1, dtype=torch.float64, device="cuda")
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        props = torch.cuda.get_device_properties(local_rank)
        mine = {"rank": rank, "device": torch.cuda.current_device(), "name": props.name,
                "pci": ("%04x:%02x:%02x" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)) if hasattr(props, "pci_bus_id")
                       else "device-%d" % torch.cuda.current_device()}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        torch.cuda.synchronize()
        sys.stdout.flush()
        os.dup2(saved_fd, 1)
        os.close(saved_fd)
        rccl = {"backend": dist.get_backend(), "ranks_seen": int(round(float(ones.item()))), "world_size": dist.get_world_size(), "devices": gathered,
                "distinct_devices": len({g["pci"] for g in gathered})}
        if rccl["ranks_seen"] != world or rccl["world_size"] != world:
            raise SystemExit("RCCL joined %d ranks (process group of %d), the launcher named %d" % (rccl["ranks_seen"], rccl["world_size"], world))
        if rccl["distinct_devices"] != world:      # (reported, not fatal: partitioned GPUs may share a bus address)
            rccl["warning"] = "%d ranks on %d distinct PCI addresses" % (world, rccl["distinct_devices"])

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
    parts = {"t_launch_s": 0.0, "t_kernel_s": 0.0, "t_collective_s": 0.0, "t_finish_s": 0.0}      # this rank's host clock, summed over the steps
    for _ in range(args.steps):
        it += 1
        _, st = lucy_iteration_sharded(eng, n_total, it, rank, world, want_output=False, force_collective=dist is not None)
        a, b = eng.last_kernel_ms()
        kernel_ms.append(a)
        finish_ms.append(b)
        for k in parts:
            parts[k] += st.get(k, 0.0)
        crossings = st["crossings"]          # whole-job crossings of the last step
    barrier()
    dt = time.perf_counter() - t0
    tiled = eng.get_option("last_lucy_mode") == 1
    if tiled:
        # the dominant kernel's launch durations, live: one more step (outside the timed region) with HIP events around every
        # tile_walk launch -- the instrumentation is an option of the engine, off in the product path and in the timed steps
        eng.set_option("tile_time_walk", 1)
        it += 1
        lucy_iteration_sharded(eng, n_total, it, rank, world, want_output=False, force_collective=dist is not None)
        eng.set_option("tile_time_walk", 0)
    # what the C-ABI's specific_energy_out costs on top of a step (the timed steps run with want_output=False, the
    # inputs / outputs of `value` are resident): the device -> host copy of the result array in the reference's layout, measured here
    d2h = []
    for _ in range(5):
        torch.cuda.synchronize()
        t_c = time.perf_counter()
        eng.specific_energy()
        d2h.append((time.perf_counter() - t_c) * 1e3)
    d2h_ms = sorted(d2h)[len(d2h) // 2]
    t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    if dist is not None:
        # Every rank applied update_energy_abs to the same all-reduced block (no broadcast): the specific energy must be the same
        # bits everywhere.  A 64-bit digest of each rank's array, min and max over the ranks in one all-reduce of two numbers.
        import hashlib
        import numpy as np
        se = eng.specific_energy()
        dig = int.from_bytes(hashlib.blake2b(np.ascontiguousarray(se).tobytes(), digest_size=6).digest(), "little")     # 48 bits: exact in a double
        d2 = torch.tensor([float(dig), -float(dig)], dtype=torch.float64, device="cuda")
        dist.all_reduce(d2, op=dist.ReduceOp.MAX)
        rccl["specific_energy_digest_rank0"] = "%012x" % dig
        rccl["specific_energy_identical_on_all_ranks"] = bool(d2[0].item() == -d2[1].item())
        rccl["allreduce_bytes_per_step"] = int(eng.get_option("lucy_block_doubles")) * 8
        if not rccl["specific_energy_identical_on_all_ranks"]:
            raise SystemExit("the ranks hold different specific energies after the all-reduce")
    # where each rank's time went (ms per step): launches (host side of the generations, which includes most of the
    # propagation: the tiled schedule polls the device), waiting for the kernels, the all-reduce, the epilogue; device
    # time of the propagation from HIP events.  Gathered after the timed region.
    mine = torch.tensor([parts["t_launch_s"], parts["t_kernel_s"], parts["t_collective_s"], parts["t_finish_s"],
                         sum(kernel_ms) * 1e-3, sum(finish_ms) * 1e-3], dtype=torch.float64, device="cuda") * (1e3 / args.steps)
    per_rank = [mine]
    if dist is not None and world > 1:
        per_rank = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(per_rank, mine)
    per_rank = torch.stack(per_rank).cpu().numpy()

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
            "d2h_ms": d2h_ms,
            "d2h_note": "median of 5 copies of specific_energy (%d doubles) device -> host in the reference's layout = what hyp_lucy_iteration's specific_energy_out adds to a step; "
                        "not inside the timed steps (want_output=False): value x ms_per_step / (ms_per_step + d2h_ms) is the rate with the copy" % int(prob.density.size),
            "iterations_in_process": it,        # warm-up + timed steps + the extra walk-timing step: what a rocprofv3 pass of this command sees
            "lucy_schedule": ("brick-tiled: generations of tile_interact / tile_emit / tile_sort / tile_walk on %d slot pools (streams)"
                              % eng.get_option("tile_pools")) if tiled else "persistent kernel, global atomics",
        }
        if rccl is not None:
            from hyperion_amd.distributed import shard_range
            rccl["shards"] = [list(shard_range(n_total, r, world)) for r in range(world)]       # [first id, packets] per rank
            out["rccl"] = rccl
        if world > 1:
            out["ms_per_step_note"] = "max over ranks of the barrier-to-barrier time of the timed steps / steps"
        names = ("launch_ms", "kernel_wait_ms", "allreduce_ms", "finish_ms", "device_propagate_ms", "device_finish_ms")
        out["per_rank_ms_per_step"] = {n: {"min": float(per_rank[:, i].min()), "max": float(per_rank[:, i].max()),
                                           "ranks": [float(x) for x in per_rank[:, i]]} for i, n in enumerate(names)}
        out["per_rank_ms_per_step"]["note"] = ("host clock of each rank per step: launch = enqueueing the generations (the tiled schedule polls the device, so "
                                               "most of the propagation is spent here), kernel_wait = waiting for the remaining kernels + replica reduction, allreduce = "
                                               "RCCL all-reduce of the accumulator block (the error flag rides in its tail: one collective) + stream sync, finish = "
                                               "update_energy_abs epilogue; device_* = HIP events on the engine's stream")
        std_case = args.grid == 128 and args.density == "uniform"
        pmc = committed_pmc("car") if (std_case and tiled) else None
        per = pmc.get("per_crossing", {}) if pmc else {}
        xs = crossings / world                      # crossings of one launch on this GPU
        roof = {"bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": achieved / PEAK_HBM_GBS,
                "kernel": "whole propagation of one Lucy iteration (all generations), HIP events on the engine's stream",
                "algorithmic_bytes_per_unit": "24 B x n_dust per cell crossing (8 B density load + 16 B accumulator read-modify-write, "
                                              "src/grid/grid_propagate_3d.f90:131-160); crossings counted in-kernel"}
        if per and "FETCH_SIZE" in per:
            # FETCH_SIZE / WRITE_SIZE are in KiB; L2 <-> fabric requests, Infinity-Cache hits included.  The guide's x2
            # correction for wide streaming reads applies to the 16 B/lane record loads of tile_walk / tile_interact, so
            # the read half is a lower bound: both are given.
            rd, wr = per["FETCH_SIZE"] * 1024.0, per["WRITE_SIZE"] * 1024.0
            roof["traffic"] = (rd + wr) * xs / 1e9
            roof["traffic_unit"] = "GB per launch, L2<->fabric (FETCH_SIZE + WRITE_SIZE as reported)"
            roof["traffic_reads_x2"] = (2 * rd + wr) * xs / 1e9
            roof["traffic_over_algorithmic"] = (rd + wr) / (24.0 * n_dust)
            roof["traffic_source"] = PMC_FILE + " (rocprofv3 --pmc passes of this command, tools/profile_round.sh); not measured in this run"
        else:
            roof["traffic"] = None
        if tiled:
            walk_ms = eng.get_option("last_walk_us") / 1e3
            n_walk = eng.get_option("last_walk_launches")
            if n_walk:
                roof["dominant_kernel"] = {
                    "name": "tile_walk_kernel", "launches_per_iteration": n_walk, "sum_ms_per_iteration": walk_ms,
                    "avg_launch_us": walk_ms * 1e3 / max(n_walk, 1),
                    "achieved_GBs_over_its_own_time": alg_bytes / (walk_ms * 1e-3) / 1e9 if walk_ms > 0 else None,
                    "note": "HIP events around every tile_walk launch on its pool's stream, in one extra step after the timed region (option tile_time_walk, off in the timed steps); with several "
                            "pools the launches overlap other kernels and stretch -- profiles/r06_car1_summary.md has the one-pool trace"}
            roof["limiter"] = "valu_issue + service-phase latency"
            roof["note"] = ("density and accumulators of a 32 x 16 x 16 brick live in LDS, so the 24 B per crossing never go to memory: `bound` names the "
                            "nominal roofline of the path (HBM), the fraction says how far it is from a streaming bound it does not have; the counters "
                            "name the limiter: VALU issue (issue_roofline) and the service phase of tile_walk, 38 % of a wave's clocks "
                            "(-DHYP_TILE_STATS build of round 5, DESIGN.md section 4.1)")
        else:
            roof["limiter"] = "memory-side atomics"
            roof["note"] = ("bound by the memory-side scattered-atomic rate (2.38e10/s, profiles/r01_atomic_rate_ubench.md): fraction %.2f"
                            % (xs / (k_ms * 1e-3) / 2.38e10))
        out["roofline"] = roof
        if per and "SQ_INSTS_VALU" in per:
            # second ceiling: VALU issue, priced per instruction class with the measured issue costs (tools/ubench/valu_issue.hip):
            # the dynamic FP64 / 64-bit-integer / transcendental-FP64 counts of the PMC passes at 4.3 / 4.3 / 16 cycles, the
            # rest of SQ_INSTS_VALU at 2.4; SALU instructions issue from the scalar unit beside them and are reported, not priced.
            f64 = sum(per.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_INT64"))
            trans = per.get("SQ_INSTS_VALU_TRANS_F64", 0.0)
            rest = max(per["SQ_INSTS_VALU"] - f64 - trans, 0.0)
            cyc = f64 * CYC_F64 + trans * CYC_TRANS_F64 + rest * CYC_32
            t_issue = cyc * xs / (N_SIMD * CLOCK_HZ)
            out["issue_roofline"] = {"bound": "valu_issue", "valu_wave_instructions_per_crossing": per["SQ_INSTS_VALU"],
                                     "fp64_and_int64_per_crossing": f64, "trans_f64_per_crossing": trans, "other_valu_per_crossing": rest,
                                     "salu_per_crossing": per.get("SQ_INSTS_SALU"),
                                     "cycles_per_instruction": {"fp64_int64": CYC_F64, "trans_f64": CYC_TRANS_F64, "other": CYC_32,
                                                                "source": "tools/ubench/valu_issue.hip at 4 waves per SIMD, profiles/r03_tiled_log.md"},
                                     "measured_avg_cycles_per_valu_instruction": (4.0 * per["SQ_ACTIVE_INST_VALU"] / per["SQ_INSTS_VALU"]) if "SQ_ACTIVE_INST_VALU" in per else None,
                                     "ideal_issue_ms": t_issue * 1e3, "measured_ms": k_ms, "frac": t_issue / (k_ms * 1e-3),
                                     "peak": "%d SIMDs x %.1f GHz" % (N_SIMD, CLOCK_HZ / 1e9),
                                     "source": PMC_FILE + " (dynamic counts summed over all tile_* kernels of this command)"}
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
