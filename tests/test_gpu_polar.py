"""Spherical and cylindrical polar grids (src/grid/grid_geometry_spherical_3d.f90,
grid_geometry_cylindrical_3d.f90) on the GPU: parity with the CPU oracle on identical Philox
streams for the reference's own regression models (test_bit_level.py:37-63: 5x7x3 spherical,
7x3x5 cylindrical cells around five off-centre sources), for BASELINE configs[0] (2D spherical
polar grid, one central point source, isotropic dust, 1e5 packets, 1 Lucy iteration), for
logarithmic r walls starting at a cavity, and for the modified random walk."""
import numpy as np
import pytest

import hyperion_amd
from cases import assert_parity, golden_problem
from hyperion_amd.benchmark import LSUN, PC, load_test_dust
from hyperion_amd.problem import PeeledImages, Problem, RunConfig, Source
from oracle_lib import Oracle

pytestmark = pytest.mark.gpu
INT_KEYS = ("crossings", "interactions", "killed_geo", "killed_int")


def run_both(prob, n, iters=1, n_img=0, rtol=1e-9, **options):
    eng = hyperion_amd.Engine(prob)
    for k, v in options.items():
        eng.set_option(k, v)
    orc = Oracle(prob)
    for it in range(1, iters + 1):
        a, sa = eng.lucy_iteration(n, it)
        b, sb = orc.lucy_iteration(n, it)
        for k in INT_KEYS:
            assert sa[k] == sb[k], (k, sa, sb)
        assert_parity(a, b, rtol=rtol)
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
    if options.get("lucy_mode") == 1:
        assert eng.get_option("last_lucy_mode") == 1
    eng.close(); orc.close()
    return a, sa, res


@pytest.mark.parametrize("grid", ["sph", "cyl"])
@pytest.mark.parametrize("name", ["False.False", "True.False", "False.True", "True.True"])
def test_reference_models(grid, name):
    """test_bit_level.py:137-173 with grid_type='sph' / 'cyl'."""
    prob, _ = golden_problem("%s_specific_energy.%s.npz" % (grid, name))
    a, st, _ = run_both(prob, 30000, iters=3)
    assert st["killed_geo"] == 0


@pytest.mark.parametrize("grid", ["sph", "cyl"])
@pytest.mark.parametrize("evenly", [False, True])
def test_reference_peeloff_models(grid, evenly):
    prob, _ = golden_problem("%s_peeloff.%s.npz" % (grid, evenly))
    run_both(prob, 5000, iters=2, n_img=20000)


def config0_problem(n_r=64, n_t=48, tau=3.0, log_r=False, n_p=1, peeled=False):
    """BASELINE configs[0]: 2D spherical polar grid, central point source, grey isotropic dust."""
    if log_r:
        r = np.hstack([0.0, np.logspace(np.log10(0.01 * PC), np.log10(PC), n_r)])
    else:
        r = np.linspace(0.0, PC, n_r + 1)
    t = np.linspace(0.0, np.pi, n_t + 1)
    p = np.linspace(0.0, 2 * np.pi, n_p + 1)
    dens = np.full((1, n_p, n_t, r.size - 1), tau / PC)
    if log_r:
        dens[..., 0] = 0.0              # the cavity
    # a flattened envelope: denser towards the mid-plane
    tc = 0.5 * (t[1:] + t[:-1])
    dens = dens * (0.2 + np.sin(tc)[None, None, :, None] ** 2)
    cfg = RunConfig()
    peel = []
    if peeled:
        # (views that do not run along a theta wall of the grid as seen from the central source: a line of sight inside a cone
        # wall is a knife edge between two cells of different density, and device / host atan2 differ in the last bit)
        peel = [PeeledImages(theta=[33.0, 95.0], phi=[10.0, 200.0], n_x=8, n_y=8, x_min=-PC, x_max=PC, y_min=-PC, y_max=PC,
                             n_ap=2, ap_min=0.2 * PC, ap_max=1.5 * PC, n_wav=6, wav_min=0.1, wav_max=1000.0)]
    return Problem(walls=[r, t, p], density=dens, dust=[load_test_dust()],
                   sources=[Source(type="point", luminosity=LSUN, temperature=6000.0, position=(0.0, 0.0, 0.0))],
                   config=cfg, peeled=peel, grid_type="sph_pol")


def test_baseline_config0_spherical_2d():
    """configs[0]: 1e5 packets, one Lucy iteration; the source sits on the origin, where theta and
    phi of a packet come from its direction (find_cell :251-268)."""
    p = config0_problem()
    a, st, _ = run_both(p, 100000)
    assert st["killed_geo"] == 0 and st["killed_int"] == 0
    # energy conservation: everything absorbed is re-emitted, the field integrates to the luminosity-weighted path
    w = p.density * p.volumes
    assert (a * w).sum() > 0


def test_spherical_3d_log_walls_with_cavity_and_images():
    p = config0_problem(n_r=40, n_t=24, n_p=12, log_r=True, peeled=True)
    a, st, _ = run_both(p, 40000, iters=2, n_img=40000)
    assert st["killed_geo"] == 0


def test_cylindrical_disc_with_images():
    w = np.hstack([0.0, np.logspace(np.log10(0.02 * PC), np.log10(PC), 30)])
    z = np.linspace(-0.5 * PC, 0.5 * PC, 21)
    ph = np.linspace(0.0, 2 * np.pi, 9)
    wc = 0.5 * (w[1:] + w[:-1]); zc = 0.5 * (z[1:] + z[:-1])
    dens = (3.0 / PC) * np.exp(-0.5 * (zc[None, :, None] / (0.15 * PC)) ** 2) * np.ones((8, 1, 1)) * (wc[None, None, :] > 0.03 * PC)
    peel = [PeeledImages(theta=[60.0], phi=[45.0], n_x=8, n_y=8, x_min=-PC, x_max=PC, y_min=-PC, y_max=PC,
                         n_ap=2, ap_min=0.2 * PC, ap_max=1.5 * PC, n_wav=6, wav_min=0.1, wav_max=1000.0)]
    src = [Source(type="point", luminosity=LSUN, temperature=6000.0, position=(0.0, 0.0, 0.0)),
           Source(type="sphere", luminosity=0.3 * LSUN, temperature=4000.0, position=(0.3 * PC, 0.1 * PC, 0.05 * PC), radius=0.01 * PC)]
    p = Problem(walls=[w, z, ph], density=dens[None], dust=[load_test_dust()], sources=src, config=RunConfig(), peeled=peel,
                grid_type="cyl_pol")
    a, st, _ = run_both(p, 40000, iters=2, n_img=40000)
    assert st["killed_geo"] == 0


# --- the brick-tiled Lucy schedule on polar grids (hyp_ptile.h): bricks of (r, theta, phi) / (w, z, phi) indices in LDS --------
# small bricks (pt_lds_kb = 1: 64 cells), small slot pools and tasks, so that packets change brick, wait and refill all the time
PT = dict(lucy_mode=1, pt_lds_kb=1, tile_slots=8192, tile_task=256, tile_drain=200)


@pytest.mark.parametrize("grid", ["sph", "cyl"])
def test_ptile_reference_models(grid):
    """the reference's regression model (7 x 5 x 3 cells, five sources, three species in the True.True variant) on the tiled schedule"""
    for name in ("False.False", "True.True"):
        prob, _ = golden_problem("%s_specific_energy.%s.npz" % (grid, name))
        a, st, _ = run_both(prob, 30000, iters=2, **PT)
        assert st["killed_geo"] == 0


def test_ptile_config0_and_periodic_phi():
    """configs[0]'s shape (source on the origin), then twelve phi cells with logarithmic r walls and a cavity: bricks of 16 x 4 x 1
    and 8 x 4 x 2 cells, packets crossing the phi = 0 seam between the first and the last brick"""
    a, st, _ = run_both(config0_problem(), 100000, **PT)
    assert st["killed_geo"] == 0 and st["killed_int"] == 0
    a, st, _ = run_both(config0_problem(n_r=40, n_t=24, n_p=12, log_r=True), 60000, iters=2, **PT)
    assert st["killed_geo"] == 0
    # one brick holds the whole grid (the default budget): visits end only with interactions
    a, st, _ = run_both(config0_problem(n_r=40, n_t=24, n_p=12, log_r=True), 60000, lucy_mode=1, tile_slots=16384, tile_task=512)
    assert st["killed_geo"] == 0


def test_ptile_cylindrical_disc_with_a_reabsorbing_sphere():
    w = np.hstack([0.0, np.logspace(np.log10(0.02 * PC), np.log10(PC), 30)])
    z = np.linspace(-0.5 * PC, 0.5 * PC, 21)
    ph = np.linspace(0.0, 2 * np.pi, 9)
    wc = 0.5 * (w[1:] + w[:-1]); zc = 0.5 * (z[1:] + z[:-1])
    dens = (3.0 / PC) * np.exp(-0.5 * (zc[None, :, None] / (0.15 * PC)) ** 2) * np.ones((8, 1, 1)) * (wc[None, None, :] > 0.03 * PC)
    src = [Source(type="point", luminosity=LSUN, temperature=6000.0, position=(0.0, 0.0, 0.0)),
           Source(type="sphere", luminosity=0.3 * LSUN, temperature=4000.0, position=(0.3 * PC, 0.1 * PC, 0.05 * PC), radius=0.01 * PC)]
    p = Problem(walls=[w, z, ph], density=dens[None], dust=[load_test_dust()], sources=src, config=RunConfig(), grid_type="cyl_pol")
    a, st, _ = run_both(p, 60000, iters=2, **PT)
    assert st["killed_geo"] == 0


def test_ptile_matches_persistent_at_scale():
    """400 x 200 cells (the benchmark row of tools/r03_other.py) at 3e6 packets: auto mode picks the tiled schedule, integer tallies
    equal to the persistent kernel's, sums to rounding"""
    prob = config0_problem(n_r=400, n_t=200, tau=3.0)
    res = []
    for mode in (0, -1):
        eng = hyperion_amd.Engine(prob)
        eng.set_option("lucy_mode", mode)
        res.append(eng.lucy_iteration(3_000_000, 1))
        assert eng.get_option("last_lucy_mode") == (0 if mode == 0 else 1)
        eng.close()
    (a, sa), (b, sb) = res
    for k in INT_KEYS:
        assert sa[k] == sb[k], (k, sa, sb)
    assert_parity(a, b)


@pytest.mark.parametrize("grid", ["sph_pol", "cyl_pol"])
def test_modified_random_walk(grid):
    """grid_mrw_3d.f90 with distance_to_closest_wall of the polar grids; trajectories are cut after
    100 interactions like the other MRW parity tests (the walk is chaotic, tests/test_gpu_mrw.py)."""
    from test_oracle_mrw import realistic_dust
    if grid == "sph_pol":
        walls = [np.linspace(0.0, PC, 9), np.linspace(0.0, np.pi, 7), np.linspace(0.0, 2 * np.pi, 5)]
    else:
        walls = [np.linspace(0.0, PC, 9), np.linspace(-PC, PC, 7), np.linspace(0.0, 2 * np.pi, 5)]
    dust = realistic_dust()
    # chi_inv_planck optical depth ~ 8 across a radial cell: the random walk kicks in away from the walls
    dens = np.full((1, 4, 6, 8), 8.0 / (float(dust.mo_chi_inv_planck[0]) * PC / 8.0))
    cfg = RunConfig()
    cfg.mrw = True; cfg.mrw_gamma = 0.2; cfg.n_inter_mrw_max = 1000; cfg.n_inter_max = 100
    p = Problem(walls=walls, density=dens, dust=[dust],
                sources=[Source(type="point", luminosity=LSUN, temperature=6000.0, position=(0.11 * PC, 0.07 * PC, -0.05 * PC))],
                config=cfg, grid_type=grid)
    eng = hyperion_amd.Engine(p)
    orc = Oracle(p)
    a, sa = eng.lucy_iteration(4000, 1)
    b, sb = orc.lucy_iteration(4000, 1)
    eng.close(); orc.close()
    for k in INT_KEYS:
        assert sa[k] == sb[k], (k, sa, sb)
    assert_parity(a, b, atol_rel=1e-10)
    # the walk was used: more path than interactions alone would give
    assert sa["killed_int"] > 0 or sa["interactions"] > 0


def test_golden_statistical():
    """GPU vs the Fortran-produced goldens of the sph / cyl models, first iteration, conserved total."""
    for grid in ("sph", "cyl"):
        prob, z = golden_problem("%s_specific_energy.False.False.npz" % grid)
        gold = z["golden/specific_energy"][0]
        prob.config.seed = -77
        eng = hyperion_amd.Engine(prob)
        big, st = eng.lucy_iteration(1000000, 1)
        eng.close()
        w = prob.density * prob.volumes
        assert (gold * w).sum() == pytest.approx((big * w).sum(), rel=0.04)
        sel = gold > 0
        assert np.median(np.abs(gold[sel] / big[sel] - 1.0)) < 0.2


def test_wall_errors():
    p = config0_problem(n_r=4, n_t=4)
    p.walls[1] = np.array([0.0, 1.0, 2.0, 3.0, 4.0])
    with pytest.raises(hyperion_amd.EngineError, match="theta walls should be between 0 and pi"):
        hyperion_amd.Engine(p)
