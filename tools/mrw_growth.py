import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hyperion_amd
from hyperion_amd.benchmark import make_benchmark_problem
from oracle_lib import Oracle
import test_gpu_mrw as t
pid = int(sys.argv[1]); mrw = int(sys.argv[2])
for m in [1, 2, 5, 10, 20, 50, 100, 200, 400, 800, 1200, 1300, 1350, 1363]:
    p = t.thicken(make_benchmark_problem(8, n_photons=2000, n_iter=2), n_species=1)
    p.config.n_inter_max = m
    p.config.mrw = bool(mrw)
    eng = hyperion_amd.Engine(p); eng.lucy_launch(pid, 1, 1); a = eng.lucy_accumulators_tensor().cpu().numpy().copy(); _, sa = eng.lucy_finish(); eng.close()
    o = Oracle(p); b, sb = o.lucy_accumulate(pid, 1, 1, n_threads=1); o.close()
    a = np.asarray(a).ravel()[: b.size]; b = b.ravel()
    d = np.abs(a - b)
    print(m, "crossings", sa["crossings"], sb["crossings"], "max abs diff / max", d.max() / b.max(), "cells differing", int((d > 0).sum()), flush=True)
