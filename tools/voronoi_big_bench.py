"""Throughput of BASELINE configs[4] on the real tessellation (tests/golden/vor_big.npz: voro++ cells of 100 000 random sites)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hyperion_amd
if os.environ.get("HYP_LIB"):        # a tuning variant built by tools/variants.py
    import hyperion_amd.engine as E
    E._lib = E.load_library(os.environ["HYP_LIB"])
from cases import voronoi_big_problem
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
p = voronoi_big_problem(n_photons=n, two_species="one" not in sys.argv[2:])      # `one`: single species (tuning variants are built for one)
eng = hyperion_amd.Engine(p)
for a in sys.argv[2:]:
    if "=" in a:
        eng.set_option(a.split("=")[0], int(a.split("=")[1]))
eng.lucy_iteration(n // 10, 1, want_output=False)
for it in (2, 3):
    _, st = eng.lucy_iteration(n, it, want_output=False)
    ms = eng.last_kernel_ms()[0]
    print("mode %d gens %d clusters %d (<= %d cells, %d B of LDS) exact-loop steps %.2e of %.2e " % (
        eng.get_option("last_lucy_mode"), eng.get_option("last_generations"), eng.get_option("vt_clusters"), eng.get_option("vt_max_cells"),
        eng.get_option("vt_max_lds"), eng.get_option("last_vt_exact_steps"), st["crossings"]), end="")
    print("lucy  n=%d kernel %.1f ms -> %.3e packets/s, %.1f crossings/packet, %.3e crossings/s, killed_geo %d, interactions/packet %.2f"
          % (n, ms, n / ms * 1e3, st["crossings"] / n, st["crossings"] / ms * 1e3, st["killed_geo"], st["interactions"] / n))
