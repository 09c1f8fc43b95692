"""Timings of the iterations that no BASELINE config measures (they had no number in any record): the raytracing iteration, the
monochromatic final iteration, the general imaging kernel (a stellar sphere makes the problem non-plain), a spherical polar grid
at scale.  One warm-up and one timed call each; HIP events (hyp_last_kernel_ms).
   python tools/r03_other.py [packets]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hyperion_amd
if os.environ.get("HYP_LIB"):        # a tuning variant built by tools/variants.py (GEOM=1: only the octree lines run)
    import hyperion_amd.engine as E
    E._lib = E.load_library(os.environ["HYP_LIB"])
from hyperion_amd.benchmark import LSUN, PC, make_benchmark_problem, make_octree_problem
from hyperion_amd.problem import PeeledImages, Source

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000


def line(what, m, ms, st, note=""):
    print("| %s | %.3g | %.1f | %.3g | %.1f | %.3g | %s |" % (what, m, ms, m / ms * 1e3, st["crossings"] / m, st["crossings"] / ms * 1e3, note), flush=True)


print("| iteration | packets | ms | packets/s | crossings per packet | crossings/s | note |\n|---|---|---|---|---|---|---|")
# (a) raytracing on configs[3]'s octree: the final iteration peels scattered light only, do_raytracing adds sources + dust
p = make_octree_problem(max_level=7)
p.config.raytracing = True
e = hyperion_amd.Engine(p)
e.lucy_iteration(n, 1, want_output=False)
e.final_iteration(n // 10)
_, st = e.final_iteration(n); line("configs[3] octree, final iteration with raytracing on (scattered light only)", n, e.last_kernel_ms()[0], st)
e.raytracing_iteration(n // 10, n // 10)
_, st = e.raytracing_iteration(n, n); line("configs[3] octree, raytracing iteration (sources + dust), 512^2 image, 1 frequency bin", 2 * n, e.last_kernel_ms()[0], st)
e.close()
# (b) general imaging kernel: the same tree lit by a stellar sphere (re-absorption, limb: not a PLAIN problem)
p = make_octree_problem(max_level=7)
p.sources = [Source(type="sphere", luminosity=LSUN, temperature=6000.0, position=(0.0, 0.0, 0.0), radius=0.004 * PC)]
e = hyperion_amd.Engine(p)
e.lucy_iteration(n, 1, want_output=False)
mode = e.get_option("last_lucy_mode")
_, st = e.lucy_iteration(n, 2, want_output=False); line("configs[3] octree with a stellar sphere, Lucy iteration (lucy_mode %d)" % mode, n, e.last_kernel_ms()[0], st)
e.final_iteration(n // 10)
_, st = e.final_iteration(n); line("... imaging iteration (plain_imaging %d, deferred rounds %d)" % (e.get_option("plain_imaging"), e.get_option("last_defer_rounds")), n, e.last_kernel_ms()[0], st)
e.close()
if os.environ.get("HYP_LIB"):
    sys.exit(0)
# (c) monochromatic final iteration on a 64^3 Cartesian grid, 5 wavelengths, one 256^2 image
p = make_benchmark_problem(64, tau=1.0)
p.config.monochromatic = True
p.config.frequencies = 2.99792458e14 / np.array([1.0, 3.0, 10.0, 30.0, 100.0])
p.peeled = [PeeledImages(theta=[45.0], phi=[45.0], n_x=256, n_y=256, x_min=-1.5 * PC, x_max=1.5 * PC, y_min=-1.5 * PC, y_max=1.5 * PC,
                         n_ap=1, ap_min=3 * PC, ap_max=3 * PC, n_wav=5, wav_min=1.0, wav_max=100.0, inu_min=1, inu_max=5)]
e = hyperion_amd.Engine(p)
e.lucy_iteration(n, 1, want_output=False)
e.mono_iteration(n // 50, n // 50)
_, st = e.mono_iteration(n // 5, n // 5); line("Cartesian 64^3, monochromatic final iteration, 5 wavelengths x (sources + dust)", 2 * n, e.last_kernel_ms()[0], st)
e.close()
# (d) spherical polar grid: configs[0]'s shape at 400 x 200 x 1 cells
from test_gpu_polar import config0_problem
p = config0_problem(n_r=400, n_t=200, tau=3.0)
for mode in (0, -1):
    e = hyperion_amd.Engine(p)
    e.set_option("lucy_mode", mode)
    e.lucy_iteration(n // 10 if mode == 0 else n, 1, want_output=False)
    _, st = e.lucy_iteration(n, 2, want_output=False)
    line("spherical polar 400 x 200 (configs[0]'s shape), Lucy iteration, %s" % ("brick-tiled (hyp_ptile.h)" if e.get_option("last_lucy_mode") == 1 else "persistent kernel"),
         n, e.last_kernel_ms()[0], st)
    e.close()
