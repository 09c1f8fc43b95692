"""Throughput of the Lucy iteration on a spherical polar grid of configs[0]'s shape at scale (400 x 200 x 1 cells), persistent
kernel and brick-tiled schedule (hyp_ptile.h):  python tools/polar_bench.py [packets] [opt=value ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hyperion_amd
if os.environ.get("HYP_LIB"):
    import hyperion_amd.engine as E
    E._lib = E.load_library(os.environ["HYP_LIB"])
from test_gpu_polar import config0_problem
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
p = config0_problem(n_r=400, n_t=200, tau=3.0)
nd = int(os.environ.get("SPECIES", "1"))      # the same medium as `nd` identical species
if nd > 1:
    import numpy as np
    p.density = np.repeat(p.density / nd, nd, axis=0)
    p.dust = [p.dust[0]] * nd
for mode in ((0, -1) if "tiled_only" not in sys.argv else (-1,)):
    e = hyperion_amd.Engine(p)
    e.set_option("lucy_mode", mode)
    for a in sys.argv[2:]:
        if "=" in a:
            e.set_option(a.split("=")[0], int(a.split("=")[1]))
    e.lucy_iteration(n // 4, 1, want_output=False)
    for it in (2, 3):
        _, st = e.lucy_iteration(n, it, want_output=False)
        ms = e.last_kernel_ms()[0]
    print("mode %d gens %d: n=%d kernel %.1f ms -> %.3e packets/s, %.1f crossings/packet, %.3e crossings/s, killed_geo %d"
          % (e.get_option("last_lucy_mode"), e.get_option("last_generations"), n, ms, n / ms * 1e3, st["crossings"] / n, st["crossings"] / ms * 1e3, st["killed_geo"]), flush=True)
    e.close()
