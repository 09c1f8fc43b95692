"""Ad-hoc GPU bring-up script (not part of the test-suite): parity of the HIP
engine against the CPU oracle on a few cases + quick timings."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from hyperion_amd import Engine, Problem
from hyperion_amd.benchmark import make_benchmark_problem
from oracle_lib import Oracle


def compare(name, prob, n, iters=1, **opts):
    eng = Engine(prob)
    for k, v in opts.items():
        eng.set_option(k, v)
    orc = Oracle(prob)
    for it in range(1, iters + 1):
        t0 = time.time(); a, sa = eng.lucy_iteration(n, it); t1 = time.time()
        b, sb = orc.lucy_iteration(n, it); t2 = time.time()
        nz = b != 0
        rel = np.abs(a[nz] - b[nz]) / np.abs(b[nz])
        print("%s it%d n=%d gpu %.3fs (kernel %.1f ms) cpu %.3fs | max rel %.3e  mismatched zeros %d | gpu %s | cpu %s"
              % (name, it, n, t1 - t0, eng.last_kernel_ms()[0], t2 - t1, rel.max() if rel.size else 0.0,
                 int(((a != 0) != nz).sum()),
                 {k: sa[k] for k in ("energy_current", "killed_geo", "killed_int", "crossings", "interactions")},
                 {k: sb[k] for k in ("energy_current", "killed_geo", "killed_int", "crossings", "interactions")}), flush=True)
    eng.close(); orc.close()


if __name__ == "__main__":
    g = os.path.join(ROOT, "tests", "golden")
    compare("kmh1", Problem.from_npz(os.path.join(g, "car_specific_energy.False.False.npz")), 20000, iters=2)
    compare("kmh1-even", Problem.from_npz(os.path.join(g, "car_specific_energy.True.False.npz")), 20000)
    compare("kmh3", Problem.from_npz(os.path.join(g, "car_specific_energy.False.True.npz")), 20000, iters=2)
    compare("bench16", make_benchmark_problem(16), 100000, iters=2)
    compare("bench64", make_benchmark_problem(64), 200000)
    compare("bench64-8copies", make_benchmark_problem(64), 200000, accum_copies=8)
    p = make_benchmark_problem(128)
    eng = Engine(p)
    for opts in ({}, {"accum_copies": 8}, {"interact_threshold": 16}, {"interact_threshold": 32, "emit_threshold": 32},
                 {"interact_threshold": 48, "emit_threshold": 32}, {"blocks_per_cu": 1}, {"blocks_per_cu": 3}):
        for k, v in dict(accum_copies=1, interact_threshold=24, emit_threshold=16, blocks_per_cu=0).items():
            eng.set_option(k, v)
        for k, v in opts.items():
            eng.set_option(k, v)
        for n in (2_000_000, 10_000_000):
            t0 = time.time(); _, st = eng.lucy_iteration(n, 1, want_output=False); dt = time.time() - t0
            ms = eng.last_kernel_ms()[0]
            print("bench128 %s n=%d wall %.3fs kernel %.1f ms -> %.3e packets/s, %.1f crossings/packet, %.3e crossings/s, alg GB/s %.1f"
                  % (opts, n, dt, ms, n / (ms * 1e-3), st["crossings"] / n, st["crossings"] / (ms * 1e-3),
                     24 * st["crossings"] / (ms * 1e-3) / 1e9), flush=True)
