import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import hyperion_amd
from hyperion_amd.benchmark import make_benchmark_problem, make_octree_problem
from test_gpu_polar import config0_problem
from cases import voronoi_big_problem
hyperion_amd.Engine(make_benchmark_problem(16)).close()
for name, mk in (("car128", lambda: make_benchmark_problem(128)), ("oct7", lambda: make_octree_problem(max_level=7)), ("sph400x200", lambda: config0_problem(n_r=400, n_t=200, tau=3.0)), ("vor1e5", lambda: voronoi_big_problem(n_photons=1000000))):
    t0 = time.perf_counter(); p = mk(); t1 = time.perf_counter(); e = hyperion_amd.Engine(p); t2 = time.perf_counter()
    e.lucy_iteration(4000000, 1, want_output=False); t3 = time.perf_counter()
    e.lucy_iteration(4000000, 2, want_output=False); t4 = time.perf_counter()
    r = e.lucy_iteration(4000000, 3); t5 = time.perf_counter()
    e.close()
    print("%s: problem %.2f s, Engine() %.2f s, first tiled iteration (4e6) %.3f s, second %.3f s, third with output %.3f s" % (name, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4), flush=True)
