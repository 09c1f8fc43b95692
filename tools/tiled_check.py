"""Bring-up of the brick-tiled Lucy iteration: parity vs the oracle + timing vs the persistent kernel."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from hyperion_amd import Engine, Problem
from hyperion_amd.benchmark import make_benchmark_problem
from oracle_lib import Oracle
from cases import golden_problem, ragged_grid_problem

def compare(name, prob, n, iters=2, **opts):
    eng = Engine(prob); eng.set_option("lucy_mode", 1)
    for k, v in opts.items(): eng.set_option(k, v)
    orc = Oracle(prob)
    for it in range(1, iters + 1):
        a, sa = eng.lucy_iteration(n, it); b, sb = orc.lucy_iteration(n, it)
        nz = b != 0
        rel = np.abs(a[nz] - b[nz]) / np.abs(b).max()
        keys = ("energy_current", "killed_geo", "killed_int", "crossings", "interactions")
        ok = all(sa[k] == sb[k] for k in keys[1:])
        print("%s it%d n=%d tallies_equal=%s max|diff|/max %.2e  gpu %s cpu %s" % (name, it, n, ok, rel.max(), [sa[k] for k in keys], [sb[k] for k in keys]), flush=True)
    eng.close(); orc.close()

if "--bench-only" not in sys.argv:
    compare("bench16", make_benchmark_problem(16), 50000)
    compare("bench16-small-pool", make_benchmark_problem(16), 50000, tile_slots=4096, tile_task=256)
    compare("bench40", make_benchmark_problem(40), 100000, iters=1)
    compare("kmh1", golden_problem("car_specific_energy.False.False.npz")[0], 20000)
    compare("kmh3", golden_problem("car_specific_energy.False.True.npz")[0], 20000)
    compare("ragged", ragged_grid_problem(), 30000)
p = make_benchmark_problem(128)
eng = Engine(p)
n = int(float(os.environ.get("N", "2e7")))
for opts in ({"lucy_mode": 0}, {"lucy_mode": 1}, {"lucy_mode": 1, "tile_slots": 1 << 23}, {"lucy_mode": 1, "tile_slots": 1 << 24},
             {"lucy_mode": 1, "tile_task": 2048}, {"lucy_mode": 1, "tile_task": 8192}):
    for k, v in dict(lucy_mode=0, tile_slots=1 << 22, tile_task=4096, accum_copies=16).items(): eng.set_option(k, v)
    for k, v in opts.items(): eng.set_option(k, v)
    eng.lucy_iteration(n // 10, 1, want_output=False)
    t0 = time.time(); _, st = eng.lucy_iteration(n, 2, want_output=False); dt = time.time() - t0
    ms = eng.last_kernel_ms()[0]
    print("bench128 %s n=%d wall %.3fs device %.1f ms -> %.3e packets/s, %.3e crossings/s" % (opts, n, dt, ms, n / ms * 1e3, st["crossings"] / ms * 1e3), flush=True)
