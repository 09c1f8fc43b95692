"""Wall time of hyp_create (tables, upload) and of the first tiled iteration's set-up (cluster / brick builders) for the BASELINE configurations."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import hyperion_amd
from hyperion_amd.benchmark import make_benchmark_problem, make_octree_problem
from cases import voronoi_big_problem
for name, make in (("configs[1] 128^3 Cartesian", lambda: make_benchmark_problem(128)),
                   ("configs[3] octree depth 7", lambda: make_octree_problem(max_level=7, n_photons=10000000, n_iter=1)),
                   ("configs[4] Voronoi 100000 sites", lambda: voronoi_big_problem(n_photons=10000000))):
    p = make()
    t0 = time.time(); e = hyperion_amd.Engine(p); t1 = time.time()
    e.lucy_iteration(4000000, 1, want_output=False); t2 = time.time()
    e.lucy_iteration(4000000, 2, want_output=False); t3 = time.time()
    print("%-34s create %.3f s, first tiled iteration (4e6 packets, builds the clusters) %.3f s, second %.3f s" % (name, t1 - t0, t2 - t1, t3 - t2), flush=True)
    e.close()
