"""configs[1] (128^3 Cartesian) timed with configuration / option overrides:
   [HYP_LIB=build/variants/x.so] python tools/car_probe.py [packets] [cfg:field=value ...] [option=value ...]
cfg:propagation_check_frequency=0 switches the propagation check off (an upper bound on what its code path costs)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hyperion_amd
if os.environ.get("HYP_LIB"):
    import hyperion_amd.engine as E
    E._lib = E.load_library(os.environ["HYP_LIB"])
from hyperion_amd.benchmark import make_benchmark_problem

args = sys.argv[1:]
n = int(float(args.pop(0))) if args and args[0][0].isdigit() else 100_000_000
p = make_benchmark_problem(128)
opts = {}
for a in args:
    k, v = a.split("=")
    if k.startswith("cfg:"):
        setattr(p.config, k[4:], type(getattr(p.config, k[4:]))(float(v)))
    else:
        opts[k] = int(v)
eng = hyperion_amd.Engine(p)
for k, v in opts.items():
    eng.set_option(k, v)
eng.lucy_iteration(n // 4, 1, want_output=False)
ms = []
for it in (2, 3, 4):
    _, st = eng.lucy_iteration(n, it, want_output=False)
    ms.append(eng.last_kernel_ms()[0])
print(json.dumps({"lib": os.path.basename(os.environ.get("HYP_LIB", "default")), "args": args, "ms": [round(m, 1) for m in ms], "best": round(min(ms), 1),
                  "packets_per_s": n / min(ms) * 1e3, "crossings": st["crossings"], "killed_geo": st["killed_geo"]}), flush=True)
