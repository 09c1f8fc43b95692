import os, sys, time

sys.path.insert(0, "."); sys.path.insert(0, "tests")
from oracle_lib import Oracle
from hyperion_amd.benchmark import make_benchmark_problem
p = make_benchmark_problem(32, n_photons=1000000, n_iter=1)
o = Oracle(p)
o.lucy_iteration(100000, 1, n_threads=128)
base = None
for nt in (1, 2, 4, 8, 16, 32, 64, 128):
    n = 150000 * nt
    t0 = time.time(); o.lucy_iteration(n, 1, n_threads=nt); dt = time.time() - t0
    r = n / dt; base = base or r
    print(nt, "threads: %.3g packets/s, x%.1f" % (r, r / base), flush=True)
