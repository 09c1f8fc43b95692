"""What an AnalyticalYSOModel-like run costs per iteration: spherical polar grid 400 x 200 (log r with a cavity), a star WITH A RADIUS or a
point source, one peeled group (SEDs, two views), Lucy iteration + imaging iteration.
   python tools/yso_probe.py [packets] [option=value ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hyperion_amd
if os.environ.get("HYP_LIB"):
    import hyperion_amd.engine as E
    E._lib = E.load_library(os.environ["HYP_LIB"])
from hyperion_amd.benchmark import LSUN, PC
from hyperion_amd.problem import Source
from test_gpu_polar import config0_problem
args = sys.argv[1:]
n = int(float(args.pop(0))) if args and args[0][0].isdigit() else 10_000_000
for kind in ("point", "sphere"):
    p = config0_problem(n_r=400, n_t=200, tau=3.0, log_r=True, peeled=True)
    if kind == "sphere":
        p.sources = [Source(type="sphere", luminosity=LSUN, temperature=6000.0, position=(0.0, 0.0, 0.0), radius=0.002 * PC)]
    e = hyperion_amd.Engine(p)
    for a in args:
        e.set_option(a.split("=")[0], int(a.split("=")[1]))
    e.lucy_iteration(n // 10, 1, want_output=False)
    _, st = e.lucy_iteration(n, 2, want_output=False)
    ms = e.last_kernel_ms()[0]
    print("%s source: Lucy %.1f ms (%.3g packets/s, %.0f crossings/packet, mode %d)" % (kind, ms, n / ms * 1e3, st["crossings"] / n, e.get_option("last_lucy_mode")), flush=True)
    e.final_iteration(n // 10)
    _, st = e.final_iteration(n)
    ms = e.last_kernel_ms()[0]
    print("%s source: imaging %.1f ms (%.3g packets/s, %.0f crossings/packet, deferred rounds %d, plain %d)"
          % (kind, ms, n / ms * 1e3, st["crossings"] / n, e.get_option("last_defer_rounds"), e.get_option("plain_imaging")), flush=True)
    e.close()
# the same with raytracing on (the usual choice for SEDs): the imaging iteration peels scattered light only, the raytracing iteration adds
# the sources' and the dust's own emission
for kind in ("point", "sphere"):
    p = config0_problem(n_r=400, n_t=200, tau=3.0, log_r=True, peeled=True)
    p.config.raytracing = True
    if kind == "sphere":
        p.sources = [Source(type="sphere", luminosity=LSUN, temperature=6000.0, position=(0.0, 0.0, 0.0), radius=0.002 * PC)]
    e = hyperion_amd.Engine(p)
    for a in args:
        e.set_option(a.split("=")[0], int(a.split("=")[1]))
    e.lucy_iteration(n, 1, want_output=False)
    e.final_iteration(n // 10)
    _, st = e.final_iteration(n)
    ms = e.last_kernel_ms()[0]
    print("%s source, raytracing on: imaging %.1f ms (%.3g packets/s, %.0f crossings/packet, deferred rounds %d)"
          % (kind, ms, n / ms * 1e3, st["crossings"] / n, e.get_option("last_defer_rounds")), flush=True)
    e.raytracing_iteration(n // 10, n // 10)
    _, st = e.raytracing_iteration(n, n)
    ms = e.last_kernel_ms()[0]
    print("%s source, raytracing iteration: %.1f ms (%.3g packets/s, %.0f crossings/packet)" % (kind, ms, 2 * n / ms * 1e3, st["crossings"] / (2 * n)), flush=True)
    e.close()
