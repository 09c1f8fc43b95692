"""-DHYP_VTILE_VERIFY build (tools/variants.py: GEOM=2 ND=2 PARTS=tile ... verify:"-DHYP_VTILE_VERIFY"): every step of the tiled Voronoi walk runs the
FP32 filter AND the reference's loop; prints the steps walked and the disagreements (must be 0).  Two-species problems only (the variant is built for ND = 2).
   HYP_LIB=build/variants/verify.so python tools/voronoi_verify.py [packets]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hyperion_amd
import hyperion_amd.engine as E
E._lib = E.load_library(os.environ["HYP_LIB"])
from cases import golden_problem, voronoi_big_problem
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
steps = bad = 0
for name, prob, m, opts in (("100 000 sites", voronoi_big_problem(n_photons=n), n, {}),
                            ("config5 small", golden_problem("vor_config5.npz")[0], 2_000_000, {"vt_cells": 16, "lucy_mode": 1})):
    eng = hyperion_amd.Engine(prob)
    for k, v in opts.items():
        eng.set_option(k, v)
    _, st = eng.lucy_iteration(m, 1, want_output=False)
    assert eng.get_option("last_lucy_mode") == 1
    b = eng.get_option("last_vt_mismatch")
    print("%s: %.3e steps, exact-loop steps %.3e, disagreements %d" % (name, st["crossings"], eng.get_option("last_vt_exact_steps"), b), flush=True)
    steps += st["crossings"]; bad += b
    eng.close()
print("VERIFY total: %.3e steps, %d disagreements" % (steps, bad))
