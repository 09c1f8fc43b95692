import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import hyperion_amd
from hyperion_amd.benchmark import make_benchmark_problem, make_octree_problem
from test_gpu_polar import config0_problem
for name, mk in (("sph400x200", lambda: config0_problem(n_r=400, n_t=200, tau=3.0, log_r=True)), ("car128", lambda: make_benchmark_problem(128)), ("oct7", lambda: make_octree_problem(max_level=7, imaging=False))):
    p = mk()
    for n in (100000, 400000, 1000000, 2000000, 4000000):
        row = []
        for mode in (-1, 0, 1):
            e = hyperion_amd.Engine(p)
            e.set_option("lucy_mode", mode)
            e.lucy_iteration(n, 1, want_output=False)
            e.lucy_iteration(n, 2, want_output=False)
            row.append("%s %.2f ms (ran %d)" % ({-1: "auto", 0: "persistent", 1: "tiled"}[mode], e.last_kernel_ms()[0], e.get_option("last_lucy_mode")))
            e.close()
        print(name, n, " | ".join(row), flush=True)
