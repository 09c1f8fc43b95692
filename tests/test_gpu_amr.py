"""Patch-based AMR grid (src/grid/grid_geometry_amr.f90) on the GPU: parity with the CPU
oracle on identical Philox streams for the reference's own AMR regression model (two levels,
refinement 1 x 2 x 10), a three-level nest with side-by-side grids, and the equivalence of a
single-grid AMR hierarchy with the Cartesian grid of the same cells."""
import numpy as np
import pytest

import hyperion_amd
from cases import assert_parity, golden_problem
from hyperion_amd.benchmark import LSUN, PC, load_test_dust, make_benchmark_problem
from hyperion_amd.problem import PeeledImages, Problem, RunConfig, Source
from oracle_lib import Oracle

pytestmark = pytest.mark.gpu
INT_KEYS = ("crossings", "interactions", "killed_geo", "killed_int")


def run_both(prob, n, iters=1, n_img=0):
    eng = hyperion_amd.Engine(prob)
    orc = Oracle(prob)
    for it in range(1, iters + 1):
        a, sa = eng.lucy_iteration(n, it)
        b, sb = orc.lucy_iteration(n, it)
        for k in INT_KEYS:
            assert sa[k] == sb[k], (k, sa, sb)
        assert_parity(a, b)
    res = None
    if n_img:
        ra, sa = eng.final_iteration(n_img)
        rb, sb = orc.final_iteration(n_img)
        for k in INT_KEYS:
            assert sa[k] == sb[k], (k, sa, sb)
        for ga, gb in zip(ra, rb):
            for name in gb:
                np.testing.assert_allclose(ga[name], gb[name], rtol=1e-9, atol=1e-11 * np.nanmax(np.abs(gb[name])), err_msg=name)
        res = ra
    eng.close(); orc.close()
    return a, sa, res


def nested_amr_problem(tau=2.0):
    """Three levels: 8^3 root; two level-2 grids side by side (refinement 2) around the centre;
    one level-3 grid (refinement 2 again) straddling the boundary between them."""
    u = PC
    lev = [1, 2, 2, 3]
    n = [[8, 8, 8], [4, 8, 8], [4, 8, 8], [8, 8, 4]]
    h = u / 4.0          # level-1 cell width
    b = [[-u, u, -u, u, -u, u],
         [-2 * h / 2 * 2, 0.0, -h * 2, h * 2, -h * 2, h * 2],
         [0.0, h * 2, -h * 2, h * 2, -h * 2, h * 2],
         [-h, h, -h, h, -h / 2, h / 2]]
    ncell = int(np.prod(np.array(n), axis=1).sum())
    rng = np.random.default_rng(7)
    rho0 = tau / (1.0 * u)
    dens = rho0 * (0.5 + rng.random(ncell))
    dust = load_test_dust()
    src = [Source(type="point", luminosity=LSUN, temperature=6000.0, position=(0.3 * h, -0.2 * h, 0.1 * h)),
           Source(type="point", luminosity=0.5 * LSUN, temperature=3000.0, position=(-0.7 * u, 0.6 * u, 0.2 * u))]
    cfg = RunConfig()
    cfg.n_initial_iter = 2
    return Problem(walls=[], density=dens[None], dust=[dust], sources=src, config=cfg, grid_type="amr",
                   amr_level=np.array(lev), amr_n=np.array(n), amr_bounds=np.array(b))


@pytest.mark.parametrize("name", ["False.False", "True.False", "False.True", "True.True"])
def test_reference_amr_model(name):
    """test_bit_level.py:137-173 with grid_type='amr' (8x6x4 root + a 4x6x20 level-2 grid)."""
    prob, _ = golden_problem("amr_specific_energy.%s.npz" % name)
    a, st, _ = run_both(prob, 30000, iters=3)
    assert st["killed_geo"] == 0


@pytest.mark.parametrize("evenly", [False, True])
def test_reference_amr_peeloff_model(evenly):
    prob, _ = golden_problem("amr_peeloff.%s.npz" % evenly)
    run_both(prob, 5000, iters=2, n_img=20000)


def test_three_level_nest():
    p = nested_amr_problem()
    a, st, _ = run_both(p, 60000, iters=2)
    assert st["killed_geo"] == 0 and st["killed_int"] == 0
    # covered cells hold no dust and keep the minimum specific energy
    w = p.density * p.volumes
    assert (a * w).sum() > 0


def test_single_grid_amr_equals_cartesian_grid():
    """One level, one grid = a uniform Cartesian grid.  AMR's find_wall has no epsilon merging
    and its walls come from linspace, so the walks agree statistically, not bit by bit."""
    pc = make_benchmark_problem(8)
    pa = Problem(walls=[], density=pc.density.reshape(1, -1), dust=pc.dust, sources=pc.sources, config=pc.config, grid_type="amr",
                 amr_level=np.array([1]), amr_n=np.array([[8, 8, 8]]), amr_bounds=np.array([[-PC, PC, -PC, PC, -PC, PC]]))
    ea, ec = hyperion_amd.Engine(pa), hyperion_amd.Engine(pc)
    a, sa = ea.lucy_iteration(200000, 1)
    b, sb = ec.lucy_iteration(200000, 1)
    np.testing.assert_allclose(a.reshape(b.shape), b, rtol=0.02, atol=5e-3 * b.max())
    assert sa["interactions"] == pytest.approx(sb["interactions"], rel=5e-3)
    ea.close(); ec.close()


def test_golden_amr_statistical():
    """GPU vs the Fortran-produced golden of the AMR model, first iteration, conserved total."""
    prob, z = golden_problem("amr_specific_energy.False.False.npz")
    gold = z["golden/specific_energy"][0]
    prob.config.seed = -77
    eng = hyperion_amd.Engine(prob)
    big, st = eng.lucy_iteration(1000000, 1)
    eng.close()
    w = prob.density * prob.volumes
    assert (gold * w).sum() == pytest.approx((big * w).sum(), rel=0.04)


# --- brick-tiled Lucy iteration (hyp_atile.h, lucy_mode=1): same packets, same answer -----------------------

def run_tiled(prob, n, iters=1, **opts):
    eng = hyperion_amd.Engine(prob)
    eng.set_option("lucy_mode", 1)
    for k, v in opts.items():
        eng.set_option(k, v)
    orc = Oracle(prob)
    for it in range(1, iters + 1):
        a, sa = eng.lucy_iteration(n, it)
        assert eng.get_option("last_lucy_mode") == 1
        b, sb = orc.lucy_iteration(n, it)
        for k in INT_KEYS:
            assert sa[k] == sb[k], (k, sa, sb)
        assert sa["energy_current"] == pytest.approx(sb["energy_current"], rel=1e-12)
        assert_parity(a, b)
    n_sl = eng.get_option("at_slabs")
    eng.close(); orc.close()
    return a, sa, n_sl


@pytest.mark.parametrize("name", ["False.False", "True.True"])
def test_atile_reference_amr_model(name):
    """The reference's AMR regression model (two levels, refinement 1 x 2 x 10) through the brick-tiled schedule: every grid one
    brick, then bricks of a single cell each (a packet changes brick at every crossing)."""
    prob, _ = golden_problem("amr_specific_energy.%s.npz" % name)
    a, st, n_sl = run_tiled(prob, 30000, iters=2)
    assert prob.amr_n.shape[0] <= n_sl <= 2 * prob.amr_n.shape[0]       # bricks of up to 16^3 cells: one or two per grid here
    a, st, n_sl = run_tiled(prob, 30000, at_cells=1, tile_slots=4096, tile_task=256, tile_drain=0, tile_poll=1)
    assert n_sl == int(np.prod(prob.amr_n, axis=1).sum())


def test_atile_nested_grids_side_by_side():
    """Three levels with two level-2 grids side by side and a level-3 grid straddling their boundary: steps between grids of
    one level, down and up the hierarchy (find_position_in_grid through the goto tables), bricks of 4 x 4 x 4 cells, small pools,
    generations all the way down; then the default brick size with the drain launch."""
    p = nested_amr_problem()
    a, st, n_sl = run_tiled(p, 60000, iters=2, at_cells=64, tile_slots=8192, tile_task=512, tile_drain=0, tile_poll=1)
    assert n_sl > 8 and st["killed_geo"] == 0
    run_tiled(p, 60000, tile_slots=16384, tile_pools=2, tile_drain=500)


def test_atile_two_species():
    p = nested_amr_problem()
    p.density = np.vstack([p.density * 0.6, p.density * 0.8])
    p.dust = [p.dust[0], p.dust[0]]
    run_tiled(p, 40000, at_cells=100, tile_slots=8192, tile_drain=100)


def test_atile_matches_persistent_at_scale():
    """Three nested 32^3 grids, 2e6 packets: both GPU schedules walk the same packets; integer tallies equal, sums to rounding."""
    from hyperion_amd.benchmark import make_amr_problem
    prob = make_amr_problem(n=32, levels=3)
    res = []
    for mode in (0, -1):
        eng = hyperion_amd.Engine(prob)
        eng.set_option("lucy_mode", mode)
        res.append(eng.lucy_iteration(2_000_000, 1))
        assert eng.get_option("last_lucy_mode") == (0 if mode == 0 else 1)
        eng.close()
    (a, sa), (b, sb) = res
    for k in INT_KEYS:
        assert sa[k] == sb[k], (k, sa, sb)
    assert_parity(a, b)
