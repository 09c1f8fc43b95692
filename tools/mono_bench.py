"""Monochromatic final iteration on a 64^3 Cartesian grid, 5 wavelengths x (sources + dust) -- the row of profiles/r03_other_iterations.md:
   python tools/mono_bench.py [packets in total] [option=value ...]      (mono_defer=0: the general kernel with inline peel-off)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import hyperion_amd
if os.environ.get("HYP_LIB"):
    import hyperion_amd.engine as E
    E._lib = E.load_library(os.environ["HYP_LIB"])
from hyperion_amd.benchmark import PC, make_benchmark_problem
from hyperion_amd.problem import PeeledImages

args = sys.argv[1:]
n = int(float(args.pop(0))) if args and args[0][0].isdigit() else 20_000_000
p = make_benchmark_problem(64, tau=1.0)
p.config.monochromatic = True
p.config.frequencies = 2.99792458e14 / np.array([1.0, 3.0, 10.0, 30.0, 100.0])
p.peeled = [PeeledImages(theta=[45.0], phi=[45.0], n_x=256, n_y=256, x_min=-1.5 * PC, x_max=1.5 * PC, y_min=-1.5 * PC, y_max=1.5 * PC,
                         n_ap=1, ap_min=3 * PC, ap_max=3 * PC, n_wav=5, wav_min=1.0, wav_max=100.0, inu_min=1, inu_max=5)]
e = hyperion_amd.Engine(p)
for a in args:
    e.set_option(a.split("=")[0], int(a.split("=")[1]))
e.lucy_iteration(10_000_000, 1, want_output=False)
e.mono_iteration(n // 100, n // 100)
import time
for rep in range(2):
    t0 = time.perf_counter()
    r, st = e.mono_iteration(n // 10, n // 10)
    wall = (time.perf_counter() - t0) * 1e3
    ms = e.last_kernel_ms()[0]
    print("mono %s: %d packets, kernels %.1f ms (wall %.1f ms) -> %.3g packets/s, %.1f crossings per packet, %.3g crossings/s, deferred %d, rounds %d, image sum %.6e"
          % (" ".join(args), n, ms, wall, n / ms * 1e3, st["crossings"] / n, st["crossings"] / ms * 1e3, e.get_option("last_mono_deferred"),
             e.get_option("last_defer_rounds"), float(np.nansum(r[0]["img"]))), flush=True)
e.close()
