"""Cylindrical polar grid 400 x 200 (a flared disc, central point source): tiled Lucy iteration.
   [HYP_LIB=...] python tools/cyl_probe.py [packets]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import hyperion_amd
if os.environ.get("HYP_LIB"):
    import hyperion_amd.engine as E
    E._lib = E.load_library(os.environ["HYP_LIB"])
from hyperion_amd.benchmark import LSUN, PC, load_test_dust
from hyperion_amd.problem import Problem, RunConfig, Source
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
from hyperion_amd.benchmark import make_cyl_disc_problem
p = make_cyl_disc_problem()
e = hyperion_amd.Engine(p)
for a in sys.argv[2:]:
    e.set_option(a.split("=")[0], int(a.split("=")[1]))
e.lucy_iteration(n // 10, 1, want_output=False)
for it in (2, 3):
    _, st = e.lucy_iteration(n, it, want_output=False)
    ms = e.last_kernel_ms()[0]
    print("cyl 400 x 200: %.1f ms, %.3g packets/s, %.0f crossings/packet, %.3g crossings/s, mode %d, killed_geo %d" % (ms, n / ms * 1e3, st["crossings"] / n, st["crossings"] / ms * 1e3, e.get_option("last_lucy_mode"), st["killed_geo"]), flush=True)
# imaging iteration on the same disc: peeled SEDs for two views (walks along fixed directions: cyl_find_wall_inv through find_wall_fixed_dir)
p = make_cyl_disc_problem(peeled=True)
e2 = hyperion_amd.Engine(p)
e2.lucy_iteration(n // 10, 1, want_output=False)
m = n // 4
e2.final_iteration(m // 10)
_, st = e2.final_iteration(m)
ms = e2.last_kernel_ms()[0]
print("cyl 400 x 200 imaging, two views: %.1f ms, %.3g packets/s, %.0f crossings/packet, %.3g crossings/s, deferred rounds %d" % (ms, m / ms * 1e3, st["crossings"] / m, st["crossings"] / ms * 1e3, e2.get_option("last_defer_rounds")), flush=True)
