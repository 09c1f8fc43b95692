"""Bring-up of the record ring (hyp_tiled.h: RecRing): parity of the tiled Lucy iteration with the oracle with the ring on and
off, on a tuning variant (HYP_LIB) or the main library; then timings at 1e8 packets."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hyperion_amd
if os.environ.get("HYP_LIB"):
    import hyperion_amd.engine as E
    E._lib = E.load_library(os.environ["HYP_LIB"])
from hyperion_amd.benchmark import make_benchmark_problem
from oracle_lib import Oracle
from cases import ragged_grid_problem

KEYS = ("killed_geo", "killed_int", "crossings", "interactions")


def compare(name, prob, n, **opts):
    orc = Oracle(prob)
    b, sb = orc.lucy_iteration(n, 1)
    orc.close()
    for ring in (0, 1):
        eng = hyperion_amd.Engine(prob); eng.set_option("lucy_mode", 1); eng.set_option("tile_ring", ring)
        for k, v in opts.items():
            eng.set_option(k, v)
        a, sa = eng.lucy_iteration(n, 1)
        ok = all(sa[k] == sb[k] for k in KEYS)
        rel = np.abs(a - b).max() / np.abs(b).max()
        print("%s ring=%d (used %d) n=%d tallies_equal=%s max|diff|/max %.2e" % (name, ring, eng.get_option("last_tile_ring"), n, ok, rel), flush=True)
        eng.close()


if "--bench-only" not in sys.argv:
    compare("bench16", make_benchmark_problem(16), 50000)
    compare("bench16-small-pool", make_benchmark_problem(16), 50000, tile_slots=4096, tile_task=256, tile_drain=0)
    compare("bench40", make_benchmark_problem(40), 200000, tile_slots=16384, tile_task=512, tile_drain=0, tile_pools=2)
    compare("ragged", ragged_grid_problem(), 30000, tile_slots=8192, tile_drain=0)
n = int(float(os.environ.get("N", "1e8")))
p = make_benchmark_problem(128)
for opts in ({"tile_ring": 0}, {"tile_ring": 1}, {"tile_ring": 1, "tile_pools": 1}, {"tile_ring": 0, "tile_pools": 1}):
    eng = hyperion_amd.Engine(p)
    for k, v in opts.items():
        eng.set_option(k, v)
    eng.lucy_iteration(n // 10, 1, want_output=False)
    best = 1e30
    for it in (2, 3):
        _, st = eng.lucy_iteration(n, it, want_output=False)
        best = min(best, eng.last_kernel_ms()[0])
    print("bench128 %s (ring used %d) n=%d device %.1f ms -> %.3e packets/s, walk kernels %.1f ms in %d launches"
          % (opts, eng.get_option("last_tile_ring"), n, best, n / best * 1e3, eng.get_option("last_walk_us") / 1e3, eng.get_option("last_walk_launches")), flush=True)
    eng.close()
