"""Identical-stream parity against the CPU oracle ON THE GRIDS THE BENCH TIMES (round 6; VERDICT r05 "Next" #1): the
128^3 Cartesian grid of BASELINE configs[1] (256 bricks, interior bricks with all 26 neighbours), the depth-7 octree of
configs[3] with its 512 x 512 Stokes detector (source on a vertex of the tree and off it), and the 100 000-site
tessellation of configs[4] with both polarising species.  The schedule that the bench runs (tiled, three pools) is
forced AND chosen by the engine itself; integer tallies must be equal, specific_energy / cubes to cases.assert_parity's
tolerance (rtol 1e-9: FP64 atomic order and 1-ulp libm differences; atol 1e-12 of the peak, 1e-10 where anisotropic
scattering chains amplify a libm ulp -- the same allowance tests/test_gpu_parity.py documents at small size).
Reference: src/grid/grid_propagate_3d.f90:35-234, src/images/images_peeled.f90:95-270, src/main/iter_final.f90:160-273.
Packet counts are what the oracle finishes in seconds on the GPU box's 16 threads."""
import numpy as np
import pytest

import hyperion_amd
from cases import assert_parity, voronoi_big_problem
from hyperion_amd.benchmark import PC, make_benchmark_problem, make_octree_problem
from oracle_lib import Oracle

pytestmark = pytest.mark.gpu
INT_KEYS = ("crossings", "interactions", "killed_geo", "killed_int")


def lucy_against_oracle(prob, n, iters, schedules, atol_rel=1e-12):
    """`schedules`: list of option dicts; every one runs `iters` Lucy iterations from a fresh engine against ONE oracle run."""
    orc = Oracle(prob)
    want = [orc.lucy_iteration(n, it) for it in range(1, iters + 1)]
    orc.close()
    out = []
    for opts in schedules:
        eng = hyperion_amd.Engine(prob)
        for k, v in opts.items():
            eng.set_option(k, v)
        for it in range(1, iters + 1):
            a, sa = eng.lucy_iteration(n, it)
            b, sb = want[it - 1]
            assert eng.get_option("last_lucy_mode") == 1, opts          # the schedule the bench times
            for k in INT_KEYS:
                assert sa[k] == sb[k], (opts, it, k, sa, sb)
            assert sa["energy_current"] == pytest.approx(sb["energy_current"], rel=1e-13)
            np.testing.assert_allclose(sa["energy_abs_tot"], sb["energy_abs_tot"], rtol=1e-9)
            assert_parity(a, b, atol_rel=atol_rel)
            np.testing.assert_array_equal(a == 0, b == 0)
        out.append((eng.get_option("last_generations"), sa))
        eng.close()
    return out


def test_configs1_grid_128_cubed_three_pools():
    """configs[1]'s own grid: 128^3 cells = 256 bricks of 32 x 16 x 16, 3e6 packets, 2 iterations (the second one re-emits
    from the first one's temperatures), three slot pools on three streams; once with the defaults (auto picks the tiled
    schedule from 1.5e6 packets) and once forced with a pool a fifth of the packet count and no drain launch, so that every
    slot is refilled several times and the last packet ends in a generation."""
    prob = make_benchmark_problem(128)
    res = lucy_against_oracle(prob, 3_000_000, 2, [dict(), dict(lucy_mode=1, tile_pools=3, tile_slots=3 * 196608, tile_drain=0)])
    assert res[0][0] > 1 and res[1][0] > res[0][0]
    assert 172.0 < res[0][1]["crossings"] / 3e6 < 175.0           # the bench's crossings per packet


@pytest.mark.parametrize("vertex", [True, False])
def test_configs3_octree_depth7_lucy_and_imaging(vertex):
    """configs[3]'s own tree (depth 7, 30 217 cells, rho ~ r^-1.5) and detector (512 x 512, Stokes): 2e6 Lucy packets on the
    subtree-tiled schedule, then 1e6 imaging packets three ways -- deferred peel-off with the propagation half on the tiled
    schedule (forced: what runs from 4e6 packets, i.e. in the bench), deferred on the persistent kernel (what the engine
    picks at this count) and inline -- all against the oracle's cube.  With the source on a vertex of the tree the kill
    counts (find_wall's negative-t branch, grid_geometry_octree.f90:527-535) must agree exactly too."""
    pos = (0.0, 0.0, 0.0) if vertex else (0.0123 * PC, -0.0217 * PC, 0.005 * PC)
    prob = make_octree_problem(max_level=7, source_position=pos)
    assert prob.n_cells > 30000
    n, m = 2_000_000, 1_000_000
    res = lucy_against_oracle(prob, n, 1, [dict(), dict(lucy_mode=1, tile_pools=3, tile_slots=3 * 131072, tile_drain=0)])
    st = res[0][1]
    if vertex:
        assert 1.0e-3 < st["killed_geo"] / n < 2.0e-3
    else:
        assert st["killed_geo"] == 0
    orc = Oracle(prob)
    orc.lucy_iteration(n, 1)
    want, sw = orc.final_iteration(m)
    orc.close()
    eng = hyperion_amd.Engine(prob)
    eng.lucy_iteration(n, 1)
    for defer in (2, 1, 0):
        eng.set_option("defer_peel", defer)
        got, sg = eng.final_iteration(m)
        assert eng.get_option("last_tiled_imaging") == (1 if defer == 2 else 0), defer
        for k in INT_KEYS:
            assert sg[k] == sw[k], (defer, k, sg, sw)
        assert sg["energy_current"] == pytest.approx(sw["energy_current"], rel=1e-13)
        for ga, gb in zip(got, want):
            assert gb["img"].shape == (4, 1, 1, 512, 512, 1)
            for name in gb:
                np.testing.assert_allclose(ga[name], gb[name], rtol=1e-9, atol=1e-11 * np.nanmax(np.abs(gb[name])),
                                           err_msg="defer_peel=%d %s" % (defer, name))
    if vertex:
        assert sw["killed_geo"] > 0
    eng.close()


def test_configs4_voronoi_100000_sites_two_species():
    """configs[4]'s own tessellation (100 000 voro++ cells, ~15.5 walls each), two anisotropic polarising species, point +
    external box source: 2e6 packets on the cluster-tiled schedule (FP32 wall filter + the reference's FP64 expression for
    the wall it names), auto and forced with small pools.  The absolute term is 1e-10 of the peak: chains of anisotropic
    scatterings amplify 1-ulp libm differences (tests/test_gpu_parity.py::test_tiled_many_bricks_several_species)."""
    prob = voronoi_big_problem(n_photons=2_000_000)
    assert prob.n_dust == 2 and prob.vor_sites.shape[0] == 100000
    res = lucy_against_oracle(prob, 2_000_000, 1, [dict(), dict(lucy_mode=1, tile_pools=3, tile_slots=3 * 131072, tile_drain=0)],
                              atol_rel=1e-10)
    assert res[0][1]["killed_geo"] == 0 and res[0][1]["killed_int"] == 0


@pytest.mark.parametrize("grid", ["sph", "cyl"])
def test_polar_grids_of_the_bench_rows_with_a_stellar_sphere(grid):
    """The polar rows of bench.py at their own size: the 400 x 200 spherical grid (log r with a cavity; SURVEY section 8 f3, "the
    geometries most YSO users run") lit by a star WITH A RADIUS -- emission from the surface with limb darkening, re-absorption by
    the star, re-emission -- and the 400 x 200 cylindrical flared disc.  Lucy iteration on the brick-tiled schedule (auto from 3e6
    packets, and forced with small pools), then the imaging iteration (SEDs, two views, forced first interaction) with its
    propagation half on the tiled schedule -- the GEN instances of the IMG kernels on the spherical grid -- and on the deferred
    rounds, against the oracle on identical streams.  Reference: src/grid/grid_geometry_spherical_3d.f90:741-1073,
    src/grid/grid_geometry_cylindrical_3d.f90:593-771, src/sources/source_type.f90:604-976, src/main/iter_final.f90:160-273."""
    from hyperion_amd.benchmark import LSUN, make_cyl_disc_problem
    from hyperion_amd.problem import Source
    if grid == "sph":
        from test_gpu_polar import config0_problem
        prob = config0_problem(n_r=400, n_t=200, tau=3.0, log_r=True, peeled=True)
        prob.sources = [Source(type="sphere", luminosity=LSUN, temperature=6000.0, position=(0.0, 0.0, 0.0), radius=0.002 * PC, limb_darkening=True)]
    else:
        prob = make_cyl_disc_problem(peeled=True)
    assert prob.n_cells == 80000
    n, m = 3_000_000, 600_000
    lucy_against_oracle(prob, n, 1, [dict(), dict(lucy_mode=1, tile_pools=3, tile_slots=3 * 65536, tile_drain=0)], atol_rel=1e-10)
    orc = Oracle(prob)
    orc.lucy_iteration(n, 1)
    want, sw = orc.final_iteration(m)
    orc.close()
    eng = hyperion_amd.Engine(prob)
    eng.lucy_iteration(n, 1)
    for defer in (2, 1):
        eng.set_option("defer_peel", defer)
        got, sg = eng.final_iteration(m)
        assert eng.get_option("last_tiled_imaging") == (1 if defer == 2 else 0), defer
        for k in INT_KEYS:
            assert sg[k] == sw[k], (defer, k, sg, sw)
        assert sg["energy_current"] == pytest.approx(sw["energy_current"], rel=1e-13)
        for ga, gb in zip(got, want):
            for name in gb:
                np.testing.assert_allclose(ga[name], gb[name], rtol=1e-9, atol=1e-10 * np.nanmax(np.abs(gb[name])),
                                           err_msg="defer_peel=%d %s" % (defer, name))
    eng.close()
