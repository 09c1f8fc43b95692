"""Lucy iteration on a nested AMR hierarchy (hyperion_amd.benchmark.make_amr_problem): time per iteration for a set of options.
   python tools/amr_lucy.py 1e8 [opt=value ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hyperion_amd
if os.environ.get("HYP_LIB"):
    import hyperion_amd.engine as E
    E._lib = E.load_library(os.environ["HYP_LIB"])
from hyperion_amd.benchmark import make_amr_problem
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
p = make_amr_problem(n=int(os.environ.get("N", "64")), levels=int(os.environ.get("LEVELS", "3")))
eng = hyperion_amd.Engine(p)
for a in sys.argv[2:]:
    eng.set_option(a.split("=")[0], int(a.split("=")[1]))
eng.lucy_iteration(n // 10, 1, want_output=False)
best = 1e30
for it in (2, 3):
    _, st = eng.lucy_iteration(n, it, want_output=False)
    best = min(best, eng.last_kernel_ms()[0])
print("%s cells %d mode %d slabs %d (max %d cells) gens %d: n=%d kernel %.1f ms -> %.3e packets/s, %.1f crossings/packet, %.3e crossings/s, killed_geo %d"
      % (" ".join(sys.argv[2:]), p.n_cells, eng.get_option("last_lucy_mode"), eng.get_option("at_slabs"), eng.get_option("at_max_cells"),
         eng.get_option("last_generations"), n, best, n / best * 1e3, st["crossings"] / n, st["crossings"] / best * 1e3, st["killed_geo"]), flush=True)
