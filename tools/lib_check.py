"""Parity of an alternative build of the library (tools/variants.py) against the oracle on the
small tiled cases.  usage: python tools/lib_check.py build/variants/<name>.so [option=value ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hyperion_amd.engine as E
E._lib = E.load_library(os.path.join(ROOT, sys.argv[1]))
import numpy as np
import hyperion_amd
from hyperion_amd.benchmark import make_benchmark_problem
from cases import golden_problem, ragged_grid_problem
from oracle_lib import Oracle
opts = dict(a.split("=") for a in sys.argv[2:])
for name, prob, n in (("bench16", make_benchmark_problem(16), 50000), ("kmh1", golden_problem("car_specific_energy.False.False.npz")[0], 20000),
                      ("ragged", ragged_grid_problem(), 30000), ("bench40", make_benchmark_problem(40), 100000)):
    eng = hyperion_amd.Engine(prob); eng.set_option("lucy_mode", 1)
    for k, v in opts.items(): eng.set_option(k, int(v))
    orc = Oracle(prob)
    a, sa = eng.lucy_iteration(n, 1); b, sb = orc.lucy_iteration(n, 1)
    keys = ("crossings", "interactions", "killed_geo", "killed_int")
    print(name, "tallies_equal", all(sa[k] == sb[k] for k in keys), [sa[k] for k in keys], [sb[k] for k in keys],
          "max rel diff %.2e" % (np.abs(a - b).max() / np.abs(b).max()), flush=True)
    eng.close(); orc.close()
