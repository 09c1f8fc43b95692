"""Maximum sizes (round 6): grids far beyond the bench's 128^3 -- 3.0e7 cells with ragged bricks and unevenly spaced walls, and
512^3 = 1.34e8 cells (1 GiB of density, 16 384 bricks) -- where the CPU oracle would take minutes per iteration.  The check is the
size-independent one the engine offers: its two independent Lucy schedules (the persistent kernel with one memory-side atomic per
crossing, and the brick-tiled slot-pool schedule with LDS accumulators) run the same packets on the same Philox streams, so their
integer tallies must be EQUAL and their specific energies equal to rounding; both are pinned against the oracle at small size
(tests/test_gpu_parity.py) and on the bench's grid (tests/test_gpu_baseline_grids.py).  Plus what physics fixes at any size:
nothing killed, every packet's energy accounted for, absorbed luminosity = the path-length estimator's sum.
Reference: src/grid/grid_propagate_3d.f90:35-234, src/grid/grid_geometry_cartesian_3d.f90:424-521, src/core/type_cell_id_3d.f90:97-102."""
import numpy as np
import pytest

import hyperion_amd
from cases import assert_parity
from hyperion_amd.benchmark import LSUN, PC, load_test_dust
from hyperion_amd.problem import Problem, RunConfig, Source

pytestmark = pytest.mark.gpu
INT_KEYS = ("crossings", "interactions", "killed_geo", "killed_int")


def big_problem(n1, n2, n3, tau=2.0):
    rng = np.random.RandomState(n1 + n2 + n3)

    def walls(n, lo, hi):      # uneven spacing: a smooth stretch plus 20 % jitter
        s = np.cumsum(1.0 + 0.5 * np.sin(np.linspace(0.0, 3.0, n)) + 0.2 * rng.uniform(size=n))
        return lo + (hi - lo) * np.hstack([0.0, s / s[-1]])

    w = [walls(n1, -1.0 * PC, 1.3 * PC), walls(n2, -0.8 * PC, 0.8 * PC), walls(n3, -0.5 * PC, 0.6 * PC)]
    c = [0.5 * (a[1:] + a[:-1]) for a in w]
    # a clumpy medium without a meshgrid of the full size in FP64 temporaries: separable profile x (1 + cheap hash noise)
    fx, fy, fz = [(1.0 + 0.8 * np.cos(3.0 * a / PC)).astype(np.float64) for a in c]
    rho = (tau / PC) * fz[:, None, None] * fy[None, :, None] * fx[None, None, :]
    rho[(rho > 0) & (((np.arange(n3)[:, None, None] * 7 + np.arange(n2)[None, :, None] * 3 + np.arange(n1)[None, None, :]) % 11) == 0)] = 0.0   # empty cells
    cfg = RunConfig()
    src = [Source(type="point", luminosity=LSUN, position=(0.11 * PC, -0.07 * PC, 0.05 * PC), temperature=5000.0),
           Source(type="point", luminosity=0.3 * LSUN, position=(-0.6 * PC, 0.4 * PC, -0.3 * PC), temperature=9000.0)]
    return Problem(walls=w, density=rho[None], dust=[load_test_dust()], sources=src, config=cfg)


@pytest.mark.parametrize("shape", [(500, 300, 200), (512, 512, 512)])
def test_two_schedules_agree_on_a_grid_of_maximum_size(shape):
    prob = big_problem(*shape)
    n = 4_000_000
    res = []
    for opts in (dict(lucy_mode=0), dict(lucy_mode=1, tile_pools=3)):
        eng = hyperion_amd.Engine(prob)
        for k, v in opts.items():
            eng.set_option(k, v)
        eng.lucy_iteration(n, 1, want_output=False)
        se, st = eng.lucy_iteration(n, 2)          # (the second iteration re-emits from the first one's temperatures)
        assert eng.get_option("last_lucy_mode") == opts["lucy_mode"]
        res.append((se, st))
        eng.close()
    # left to itself the engine keeps a grid of thousands of bricks on the persistent kernel until ~1 500 packets per brick are in flight
    # (a brick's load and flush must be shared by enough packets: hyp_lucy.hip, tools/big_grid_probe.py)
    eng = hyperion_amd.Engine(prob)
    eng.lucy_iteration(n, 1, want_output=False)
    assert eng.get_option("last_lucy_mode") == 0
    eng.close()
    (a, sa), (b, sb) = res
    for k in INT_KEYS:
        assert sa[k] == sb[k], (k, sa, sb)
    assert sa["killed_geo"] == 0 and sa["killed_int"] == 0
    assert sa["energy_current"] == pytest.approx(sb["energy_current"], rel=1e-13)
    assert_parity(a, b)
    np.testing.assert_array_equal(a == 0, b == 0)
    # every cell a packet can reach has been crossed at this packet count only on the small grid; on both: empty cells stay at the floor
    assert np.isfinite(a).all() and (a >= 0).all()
    np.testing.assert_allclose(sa["energy_abs_tot"], sb["energy_abs_tot"], rtol=1e-10)


def test_slot_pool_shrinks_when_the_device_memory_is_taken():
    """The tiled schedule's pool defaults to 25e6 slots (7 GB), within a third of the free memory and never below 6.3e6; on a device whose
    memory other handles or processes hold even that may not be there.  The pool is then halved until it fits -- fewer packets in flight,
    the same packets and the same answer -- instead of failing the iteration."""
    import torch
    from hyperion_amd.benchmark import make_benchmark_problem
    prob = make_benchmark_problem(64)
    n = 20_000_000
    ref = hyperion_amd.Engine(prob)
    ref.set_option("lucy_mode", 1)
    a, sa = ref.lucy_iteration(n, 1)
    assert ref.get_option("last_tile_slots") >= n       # (every packet of this iteration has its slot: fewer than the default 25e6)
    ref.close()
    eng = hyperion_amd.Engine(prob)
    eng.set_option("lucy_mode", 1)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()           # (what earlier tests left in torch's cache would be handed out again instead of taken from the device)
    hog = []
    for _ in range(8):                 # leave less than the 1.76 GB of the smallest default pool
        free, _ = torch.cuda.mem_get_info()
        if free < 1.5e9:
            break
        hog.append(torch.empty(int(free - 1.4e9), dtype=torch.uint8, device="cuda"))
    assert torch.cuda.mem_get_info()[0] < 1.6e9
    try:
        b, sb = eng.lucy_iteration(n, 1)
        assert eng.get_option("last_lucy_mode") == 1
        assert 0 < eng.get_option("last_tile_slots") < (3 << 21), "the pool was not halved"
    finally:
        del hog
        torch.cuda.empty_cache()
        eng.close()
    for k in INT_KEYS:
        assert sa[k] == sb[k], (k, sa, sb)
    assert_parity(a, b)
