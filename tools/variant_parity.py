"""Parity of a tuning variant (HYP_LIB=build/variants/x.so, built for one geometry and one species by tools/variants.py) against the CPU oracle
on identical streams: 128^3 / 40^3 Cartesian, tiled schedule with small and default pools.  Prints PARITY OK or raises."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hyperion_amd
import hyperion_amd.engine as E
if os.environ.get("HYP_LIB"):
    E._lib = E.load_library(os.environ["HYP_LIB"])
from hyperion_amd.benchmark import make_benchmark_problem
from oracle_lib import Oracle
from cases import assert_parity
for n_grid, n, opts in ((40, 300000, dict(lucy_mode=1, tile_slots=196608, tile_pools=3, tile_drain=0)), (40, 100000, dict(lucy_mode=1, tile_slots=4096 * 3, tile_task=256, tile_drain=0)),
                        (128, 3000000, dict()), (64, 1000000, dict(lucy_mode=1, tile_pools=1, tile_slots=65536))):
    p = make_benchmark_problem(n_grid, density="powerlaw" if n_grid == 64 else "uniform")
    orc = Oracle(p)
    eng = hyperion_amd.Engine(p)
    for k, v in opts.items():
        eng.set_option(k, v)
    for it in (1, 2):
        a, sa = eng.lucy_iteration(n, it)
        b, sb = orc.lucy_iteration(n, it)
        for k in ("crossings", "interactions", "killed_geo", "killed_int"):
            assert sa[k] == sb[k], (n_grid, it, k, sa, sb)
        assert_parity(a, b)
    assert eng.get_option("last_lucy_mode") == 1
    eng.close(); orc.close()
print("PARITY OK", os.path.basename(os.environ.get("HYP_LIB", "default")))
