"""Parity of an octree tuning variant (HYP_LIB=build/variants/x.so, GEOM=1 ND=1) against the CPU oracle: Lucy + imaging on the depth-5 and depth-7 trees."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hyperion_amd
import hyperion_amd.engine as E
if os.environ.get("HYP_LIB"):
    E._lib = E.load_library(os.environ["HYP_LIB"])
from hyperion_amd.benchmark import PC, make_octree_problem
from oracle_lib import Oracle
from cases import assert_parity
for lvl, n, m, opts, pos in ((5, 60000, 40000, dict(lucy_mode=1, ot_cells=80, tile_slots=8192, tile_task=512, tile_drain=0, defer_peel=2), (0.0, 0.0, 0.0)),
                             (7, 2000000, 1000000, dict(defer_peel=2), (0.0, 0.0, 0.0)),
                             (7, 1000000, 500000, dict(lucy_mode=1, tile_pools=3, tile_slots=3 * 131072, tile_drain=0, defer_peel=2), (0.0123 * PC, -0.0217 * PC, 0.005 * PC))):
    p = make_octree_problem(max_level=lvl, n_pix=64, source_position=pos)
    orc = Oracle(p); eng = hyperion_amd.Engine(p)
    for k, v in opts.items():
        eng.set_option(k, v)
    for it in (1, 2):
        a, sa = eng.lucy_iteration(n, it); b, sb = orc.lucy_iteration(n, it)
        for k in ("crossings", "interactions", "killed_geo", "killed_int"):
            assert sa[k] == sb[k], (lvl, it, k, sa, sb)
        assert_parity(a, b)
    assert eng.get_option("last_lucy_mode") == 1
    ga, sa = eng.final_iteration(m); gb, sb = orc.final_iteration(m)
    for k in ("crossings", "interactions", "killed_geo", "killed_int"):
        assert sa[k] == sb[k], (lvl, "final", k, sa, sb)
    for xa, xb in zip(ga, gb):
        for name in xb:
            np.testing.assert_allclose(xa[name], xb[name], rtol=1e-9, atol=1e-11 * np.nanmax(np.abs(xb[name])), err_msg=name)
    eng.close(); orc.close()
print("PARITY OK", os.path.basename(os.environ.get("HYP_LIB", "default")))
