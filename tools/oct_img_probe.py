"""configs[3] imaging iteration timed with option overrides:  python tools/oct_img_probe.py [packets] [vertex|off] [option=value ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hyperion_amd
if os.environ.get("HYP_LIB"):
    import hyperion_amd.engine as E
    E._lib = E.load_library(os.environ["HYP_LIB"])
from hyperion_amd.benchmark import PC, make_octree_problem
args = sys.argv[1:]
n = int(float(args.pop(0))) if args and args[0][0].isdigit() else 100_000_000
pos = (0.0, 0.0, 0.0)
if args and args[0] in ("vertex", "off"):
    if args.pop(0) == "off":
        pos = (0.0123 * PC, -0.0217 * PC, 0.005 * PC)
p = make_octree_problem(max_level=7, source_position=pos)
eng = hyperion_amd.Engine(p)
eng.lucy_iteration(n // 10, 1, want_output=False)
for a in args:
    eng.set_option(a.split("=")[0], int(a.split("=")[1]))
eng.final_iteration(n // 10)
ms = []
for rep in range(3):
    _, st = eng.final_iteration(n)
    ms.append(eng.last_kernel_ms()[0])
print("oct_img %s n=%d: %s ms, best %.1f -> %.3e packets/s; %d generations, end-game %d packets, %d rounds, %.2f events/packet, %.1f crossings/packet, killed %d"
      % (" ".join(args), n, ["%.1f" % m for m in ms], min(ms), n / min(ms) * 1e3, eng.get_option("last_generations"), eng.get_option("last_end_game"),
         eng.get_option("last_defer_rounds"), eng.get_option("last_defer_events") / n, st["crossings"] / n, st["killed_geo"]))
