"""Throughput of BASELINE config 4 (adaptive octree depth 7 + 512x512 peel-off)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hyperion_amd
if os.environ.get("HYP_LIB"):        # a tuning variant built by tools/variants.py
    import hyperion_amd.engine as E
    E._lib = E.load_library(os.environ["HYP_LIB"])
from hyperion_amd.benchmark import make_octree_problem
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
p = make_octree_problem(max_level=7)
print("cells", p.n_cells, "refined", int(p.refined.sum()))
eng = hyperion_amd.Engine(p)
for a in sys.argv[2:]:
    eng.set_option(a.split("=")[0], int(a.split("=")[1]))
eng.lucy_iteration(n // 10, 1, want_output=False)
for it in (2, 3):
    _, st = eng.lucy_iteration(n, it, want_output=False)
    ms = eng.last_kernel_ms()[0]
    print("lucy  n=%d kernel %.1f ms -> %.3e packets/s, %.1f crossings/packet, %.3e crossings/s, killed_geo %d"
          % (n, ms, n / ms * 1e3, st["crossings"] / n, st["crossings"] / ms * 1e3, st["killed_geo"]))
for rep in range(2):
    res, st = eng.final_iteration(n)
    ms = eng.last_kernel_ms()[0]
    print("final (deferred peel-off, %d rounds, %.2f events/packet) n=%d kernel %.1f ms -> %.3e packets/s, %.1f crossings/packet (incl. peel-off walks), %.3e crossings/s"
          % (eng.get_option("last_defer_rounds"), eng.get_option("last_defer_events") / n, n, ms, n / ms * 1e3, st["crossings"] / n, st["crossings"] / ms * 1e3))
eng.set_option("defer_peel", 0)
res, st = eng.final_iteration(n)
ms = eng.last_kernel_ms()[0]
print("final (inline peel-off) n=%d kernel %.1f ms -> %.3e packets/s, %.1f crossings/packet (incl. peel-off walks), %.3e crossings/s"
      % (n, ms, n / ms * 1e3, st["crossings"] / n, st["crossings"] / ms * 1e3))
if os.environ.get("HYP_LIB"):
    sys.exit(0)
eng.set_option("oct_neighbours", 0)         # geo_advance climbing and descending as the reference does
_, st = eng.lucy_iteration(n, 3, want_output=False)
ms = eng.last_kernel_ms()[0]
print("lucy  (no neighbour table) kernel %.1f ms -> %.3e packets/s" % (ms, n / ms * 1e3))
res, st = eng.final_iteration(n)
ms = eng.last_kernel_ms()[0]
print("final (inline peel-off, no neighbour table) kernel %.1f ms -> %.3e packets/s" % (ms, n / ms * 1e3))
eng.set_option("oct_neighbours", 1)
eng.set_option("plain_imaging", 0)          # the general imaging kernel, for comparison
res, st = eng.final_iteration(n)
ms = eng.last_kernel_ms()[0]
print("final (general kernel) n=%d kernel %.1f ms -> %.3e packets/s" % (n, ms, n / ms * 1e3))
