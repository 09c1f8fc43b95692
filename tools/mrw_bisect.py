"""Find the first packet on which the HIP engine and the oracle disagree (MRW set-up)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hyperion_amd
from hyperion_amd.benchmark import make_benchmark_problem
from oracle_lib import Oracle
import test_gpu_mrw as t

def prob(n_inter_max=None, mrw_max=1000):
    p = t.thicken(make_benchmark_problem(8, n_photons=2000, n_iter=2), n_species=1, n_mrw_max=mrw_max)
    if n_inter_max is not None:
        p.config.n_inter_max = n_inter_max
    return p

def gpu(p, first, n):
    eng = hyperion_amd.Engine(p)
    eng.lucy_launch(first, n, 1); eng.lucy_accumulators()
    _, st = eng.lucy_finish()
    eng.close()
    return st

def cpu(p, first, n):
    o = Oracle(p)
    _, st = o.lucy_accumulate(first, n, 1, n_threads=4)
    o.close()
    return st

p = prob()
bad = None
for first in range(0, 2000, 100):
    a, b = gpu(p, first, 100), cpu(p, first, 100)
    if a["crossings"] != b["crossings"] or a["interactions"] != b["interactions"]:
        print("chunk", first, a["crossings"], b["crossings"], a["interactions"], b["interactions"])
        for i in range(first, first + 100):
            a, b = gpu(p, i, 1), cpu(p, i, 1)
            if a["crossings"] != b["crossings"] or a["interactions"] != b["interactions"]:
                print(" packet", i, a["crossings"], b["crossings"], a["interactions"], b["interactions"])
                if bad is None: bad = i
        break
if bad is not None and len(sys.argv) > 1:
    os.environ["ORC_TRACE"] = "1"
    q = prob(int(sys.argv[1]))
    print(gpu(q, bad, 1)["crossings"], cpu(q, bad, 1)["crossings"])
elif bad is not None:
    # sweep the interaction limit: the first limit at which the crossing counts differ
    lo, hi = 0, 1 << 20
    for nmax in [1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 512, 1024, 2048, 4096]:
        q = prob(nmax)
        a, b = gpu(q, bad, 1), cpu(q, bad, 1)
        print("  n_inter_max", nmax, a["crossings"], b["crossings"], a["interactions"], b["interactions"], a["killed_int"], b["killed_int"])
        if a["crossings"] != b["crossings"]:
            for m in range(prev + 1, nmax + 1):
                q = prob(m)
                a, b = gpu(q, bad, 1), cpu(q, bad, 1)
                print("   n_inter_max", m, a["crossings"], b["crossings"], a["interactions"], b["interactions"])
                if a["crossings"] != b["crossings"]:
                    break
            break
        prev = nmax
