"""Lucy iteration on a grid of maximum size (tests/test_gpu_large_grids.py's medium), persistent kernel against the tiled schedule:
   [HYP_LIB=...] python tools/big_grid_probe.py n1 n2 n3 [packets] [option=value ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hyperion_amd
if os.environ.get("HYP_LIB"):
    import hyperion_amd.engine as E
    E._lib = E.load_library(os.environ["HYP_LIB"])
from test_gpu_large_grids import big_problem
n1, n2, n3 = (int(a) for a in sys.argv[1:4])
n = int(float(sys.argv[4])) if len(sys.argv) > 4 else 50_000_000
t0 = time.time()
p = big_problem(n1, n2, n3)
print("problem built in %.1f s" % (time.time() - t0), flush=True)
for mode in (0, 1):
    t0 = time.time()
    e = hyperion_amd.Engine(p)
    e.set_option("lucy_mode", mode)
    for a in sys.argv[5:]:
        e.set_option(a.split("=")[0], int(a.split("=")[1]))
    t1 = time.time()
    e.lucy_iteration(n // 4, 1, want_output=False)
    for it in (2, 3):
        _, st = e.lucy_iteration(n, it, want_output=False)
        ms = e.last_kernel_ms()[0]
    print("%d x %d x %d, asked mode %d ran %d (engine created in %.1f s): n=%d kernel %.1f ms -> %.3e packets/s, %.1f crossings/packet, %.3e crossings/s, killed_geo %d, generations %d"
          % (n1, n2, n3, mode, e.get_option("last_lucy_mode"), t1 - t0, n, ms, n / ms * 1e3, st["crossings"] / n, st["crossings"] / ms * 1e3, st["killed_geo"], e.get_option("last_generations")), flush=True)
    e.close()
