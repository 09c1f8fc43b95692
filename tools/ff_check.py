"""Emission + forced first interaction ahead of the rounds (hyp_defer.h: ff_walk_kernel, EmitRec) against the propagation kernel doing
both itself: same tallies, same images (sums in another order), on configs[3]'s octree.   python tools/ff_check.py [packets]      (HYP_LIB: a tools/variants.py build)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import hyperion_amd
if os.environ.get("HYP_LIB"):
    import hyperion_amd.engine as E
    E._lib = E.load_library(os.environ["HYP_LIB"])
from hyperion_amd.benchmark import make_octree_problem

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 300000
eng = hyperion_amd.Engine(make_octree_problem(max_level=7))
eng.lucy_iteration(n, 1, want_output=False)
res = {}
for ff in (1, 0):
    eng.set_option("ff_prepass", ff)
    r, st = eng.final_iteration(n)
    assert eng.get_option("last_ff_prepass") == ff, (ff, eng.get_option("last_ff_prepass"))
    res[ff] = (r, st)
    print("ff_prepass", ff, {k: st[k] for k in ("crossings", "killed_geo", "killed_int", "interactions", "energy_current")}, "rounds", eng.get_option("last_defer_rounds"))
(ra, sa), (rb, sb) = res[1], res[0]
for k in ("crossings", "killed_geo", "killed_int", "interactions"):
    assert sa[k] == sb[k], (k, sa[k], sb[k])
assert sa["energy_current"] == sb["energy_current"]
worst = 0.0
for ga, gb in zip(ra, rb):
    for k in ga:
        a, b = ga[k], gb[k]
        scale = np.abs(b).max()
        if scale > 0:
            worst = max(worst, np.abs(a - b).max() / scale)
        assert np.allclose(a, b, rtol=1e-9, atol=1e-12 * scale), k
print("images equal; worst |a - b| / max|b| = %.3g" % worst)
