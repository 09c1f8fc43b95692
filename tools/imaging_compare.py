"""Deferred vs inline peel-off on grids other than the octree of configs[3]: the default must not lose anywhere."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hyperion_amd
if os.environ.get("HYP_LIB"):        # a tuning variant built by tools/variants.py (one geometry, one species)
    import hyperion_amd.engine as E
    E._lib = E.load_library(os.environ["HYP_LIB"])
from cases import imaging_problem, golden_problem
from hyperion_amd.problem import PeeledImages
from hyperion_amd.benchmark import PC
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_000_000
cases = [("car 64^3 tau=1, 1 view 256^2 Stokes", imaging_problem(64, n_x=256, n_y=256)),
         ("car 64^3 tau=5, 3 views", imaging_problem(64, tau=5.0, n_x=128, n_y=128, theta=[30., 60., 90.], phi=[0., 120., 240.]))]
for g in (() if os.environ.get("HYP_LIB") else ("sph", "cyl", "amr")):
    p, _ = golden_problem("%s_specific_energy.False.False.npz" % g)
    p.peeled = [PeeledImages(theta=[45.0], phi=[45.0], n_wav=4, wav_min=0.1, wav_max=1000.0, n_x=64, n_y=64,
                             x_min=-PC, x_max=PC, y_min=-PC, y_max=PC, n_ap=1, ap_min=PC, ap_max=PC)]
    cases.append(("%s golden model (5 point sources), 1 view 64^2" % g, p))
for name, p in cases:
    eng = hyperion_amd.Engine(p)
    eng.lucy_iteration(n // 5, 1, want_output=False)
    line = name + ": plain=%d" % eng.get_option("plain_imaging")
    for defer in (1, 0):
        eng.set_option("defer_peel", defer)
        eng.final_iteration(n // 10)
        res, st = eng.final_iteration(n)
        ms = eng.last_kernel_ms()[0]
        line += " | %s %.1f ms (%.3e packets/s, %d rounds)" % ("deferred" if defer else "inline", ms, n / ms * 1e3, eng.get_option("last_defer_rounds"))
    print(line, flush=True)
    eng.close()
