import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hyperion_amd import Engine
from hyperion_amd.benchmark import make_benchmark_problem
p = make_benchmark_problem(128)
eng = Engine(p); eng.set_option("lucy_mode", 1)  # tiled even for short runs
for a in sys.argv[1:]:
    k, v = a.split("="); eng.set_option(k, int(v))
n = int(float(os.environ.get("N", "2e7")))
eng.lucy_iteration(n // 10, 1, want_output=False)
_, st = eng.lucy_iteration(n, 2, want_output=False)
ms = eng.last_kernel_ms()[0]
print("tiled n=%d device %.1f ms -> %.3e packets/s %.3e crossings/s" % (n, ms, n / ms * 1e3, st["crossings"] / ms * 1e3))
