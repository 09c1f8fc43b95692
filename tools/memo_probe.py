import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import hyperion_amd
from hyperion_amd.benchmark import make_octree_problem, PC
n = 40_000_000
for pos in ((0.0, 0.0, 0.0), (0.013 * PC, 0.007 * PC, 0.011 * PC)):
    p = make_octree_problem(max_level=7, source_position=pos)
    for memo in (1, 0):
        e = hyperion_amd.Engine(p)
        e.set_option("direct_memo", memo)
        e.lucy_iteration(n // 10, 1, want_output=False)
        e.final_iteration(n // 10)
        _, st = e.final_iteration(n)
        print(pos[0] != 0, "memo", memo, e.get_option("last_direct_memo"), "ms %.1f" % e.last_kernel_ms()[0], st["crossings"], st["killed_geo"], flush=True)
        e.close()
