"""Voronoi grid (src/grid/grid_geometry_voronoi.f90), external sources
(source_type.f90:748-933) and multi-species Henyey-Greenstein dust on the GPU:
BASELINE config 5 in small, against the CPU oracle on identical Philox streams."""
import numpy as np
import pytest

import hyperion_amd
from cases import assert_parity, golden_problem
from hyperion_amd.benchmark import LSUN, PC, make_benchmark_problem, make_octree_problem
from hyperion_amd.problem import PeeledImages, Source
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
        assert sa["energy_current"] == pytest.approx(sb["energy_current"], rel=1e-12)
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


def test_config5_small_voronoi_hg_dust_external_source():
    prob, _ = golden_problem("vor_config5.npz")
    assert prob.grid_type == "vor" and prob.n_dust == 2 and [s.type for s in prob.sources] == ["point", "extern_box"]
    a, st, res = run_both(prob, 60000, iters=3, n_img=40000)
    assert st["killed_int"] == 0
    assert res[0]["sed"].shape == (4, 4, 2, 3, 4)
    assert np.abs(res[0]["sed"][1]).max() > 0          # HG dust with p_lin_max > 0 polarises


def test_voronoi_lattice_equals_cartesian():
    """Sites on a (1e-3 jittered) 6^3 lattice tessellate into the cells of a 6^3
    Cartesian grid: same Philox streams, same interactions, cell energies within
    the jitter."""
    pv, _ = golden_problem("vor_lattice.npz")
    a, sa, _ = run_both(pv, 200000)
    pc_ = make_benchmark_problem(6)
    pc_.sources[0].position = pv.sources[0].position
    eng = hyperion_amd.Engine(pc_)
    b, sb = eng.lucy_iteration(200000, 1)
    assert sa["interactions"] == sb["interactions"]
    w = np.linspace(-1, 1, 7) * PC
    ix, iy, iz = (np.searchsorted(w, pv.vor_sites[:, k]) - 1 for k in range(3))
    np.testing.assert_allclose(a[0], b[0][iz, iy, ix], rtol=0.02, atol=0.01 * b.max())
    assert a.sum() == pytest.approx(b.sum(), rel=1e-3)


@pytest.mark.parametrize("grid", ["car", "oct", "vor"])
@pytest.mark.parametrize("kind", ["extern_box", "extern_sph"])
def test_external_sources_on_every_grid(grid, kind):
    if grid == "car":
        p = make_benchmark_problem(8, tau=0.7)
    elif grid == "oct":
        p = make_octree_problem(max_level=4, imaging=False)
    else:
        p, _ = golden_problem("vor_lattice.npz")
    if kind == "extern_box":
        src = Source(type="extern_box", luminosity=LSUN, temperature=4000.0, box=(-PC, PC, -PC, PC, -PC, PC))
    else:
        src = Source(type="extern_sph", luminosity=LSUN, temperature=4000.0, position=(0.0, 0.0, 0.0), radius=0.95 * PC)
    p.sources = [src, Source(type="point", luminosity=0.5 * LSUN, temperature=7000.0, position=(0.11 * PC, 0.07 * PC, -0.13 * PC))]
    p.peeled = [PeeledImages(theta=[35.0, 140.0], phi=[20.0, 200.0], n_wav=2, wav_min=0.1, wav_max=1000.0, n_x=6, n_y=6,
                             x_min=-1.5 * PC, x_max=1.5 * PC, y_min=-1.5 * PC, y_max=1.5 * PC, n_ap=2, ap_min=0.5 * PC,
                             ap_max=2 * PC, track_origin="detailed")]
    run_both(p, 30000, iters=2, n_img=30000)


def test_external_box_gives_uniform_isotropic_field():
    """Lambertian emission from the faces of a closed box fills it with a uniform
    isotropic radiation field: in the optically thin limit every cell absorbs
    4 kappa L / A per unit mass (A = total face area)."""
    p = make_benchmark_problem(10, tau=1e-6)
    p.sources = [Source(type="extern_box", luminosity=LSUN, temperature=5000.0, box=(-PC, PC, -PC, PC, -PC, PC))]
    eng = hyperion_amd.Engine(p)
    se, st = eng.lucy_iteration(2_000_000, 1)
    expect = 4.0 * 0.5 * LSUN / (24.0 * PC * PC)
    assert se.mean() == pytest.approx(expect, rel=5e-3)
    assert se.std() / se.mean() < 0.03


def test_unpeeled_external_source_and_messages():
    p = make_benchmark_problem(6)
    p.sources = [Source(type="extern_sph", luminosity=LSUN, temperature=4000.0, radius=0.9 * PC, peeloff=False)]
    p.peeled = [PeeledImages(theta=[50.0], phi=[10.0], n_wav=1, wav_min=0.01, wav_max=1e5, compute_image=False,
                             n_ap=1, ap_min=3 * PC, ap_max=3 * PC, track_origin="basic")]
    a, st, res = run_both(p, 20000, n_img=20000)
    assert res[0]["sed"][0, 0].sum() == 0.0            # direct source light is not peeled
    assert res[0]["sed"][0, 1:].sum() > 0.0            # dust emission / scattered light is
    p.sources[0].type = "extern_box"
    p.sources[0].box = (-9 * PC, 9 * PC, -PC, PC, -PC, PC)     # box sticks out of the grid
    eng = hyperion_amd.Engine(p)
    with pytest.raises(hyperion_amd.EngineError, match="photon was not emitted inside a cell"):
        eng.lucy_iteration(1000, 1)


def test_voronoi_random_position_cell_map_source_and_raytracing():
    """random_position_cell of the Voronoi grid (grid_geometry_voronoi.f90:285-310: rejection sampling in the cell's
    bounding box) makes luminosity-map sources and the raytracing iteration's dust emission available there."""
    prob, _ = golden_problem("vor_config5.npz")
    assert prob.vor_bb is not None and prob.vor_bb.shape == (400, 6)
    rng = np.random.RandomState(5)
    lum = rng.uniform(0.0, 1.0, 400); lum[rng.uniform(size=400) < 0.5] = 0.0
    prob.sources.append(Source(type="map", luminosity=0.7 * LSUN, temperature=2500.0, map=lum))
    prob.config.raytracing = True
    prob.config.output_n_photons = "last"
    eng = hyperion_amd.Engine(prob); orc = Oracle(prob)
    for it in (1, 2):
        a, sa = eng.lucy_iteration(40000, it); b, sb = orc.lucy_iteration(40000, it)
        assert all(sa[k] == sb[k] for k in INT_KEYS)
        assert_parity(a, b)
    ra, sa = eng.final_iteration(20000); rb, sb = orc.final_iteration(20000)
    assert all(sa[k] == sb[k] for k in INT_KEYS)
    ra, sa = eng.raytracing_iteration(20000, 20000); rb, sb = orc.raytracing_iteration(20000, 20000)
    for ga, gb in zip(ra, rb):
        for name in gb:
            np.testing.assert_allclose(ga[name], gb[name], rtol=1e-9, atol=1e-11 * np.nanmax(np.abs(gb[name])), err_msg=name)
    eng.close(); orc.close()
    # every packet of a map source that lights one cell starts in that cell
    prob, _ = golden_problem("vor_config5.npz")
    one = np.zeros(400); one[123] = 1.0
    prob.sources = [Source(type="map", luminosity=LSUN, temperature=3000.0, map=one)]
    prob.config.output_n_photons = "last"
    prob.density = prob.density * 1e-6          # thin: hardly any packet comes back to the cell it started in
    eng = hyperion_amd.Engine(prob); orc = Oracle(prob)
    eng.lucy_iteration(5000, 1); orc.lucy_iteration(5000, 1)
    n, c = eng.n_photons(), orc.n_photons()
    eng.close(); orc.close()
    assert c[123] == 5000 and c.max() == 5000
    assert n[123] == 5000 and n.max() == 5000
    np.testing.assert_array_equal(n, c)


def test_full_size_voronoi_config5_properties():
    """configs[4] on the real 100 000-site tessellation at 1e7 packets: nothing killed, every packet accounted for, the
    interaction statistics of the same medium on a Cartesian grid (uniform density: the mean number of interactions per
    packet does not depend on how space is cut into cells), absorbed energy = emitted energy to the precision of the
    estimator."""
    from cases import voronoi_big_problem
    n = 10_000_000
    prob = voronoi_big_problem(n_photons=n)
    assert prob.vor_sites.shape[0] == 100000 and 14.0 < (np.diff(prob.vor_idx).mean()) < 17.5
    eng = hyperion_amd.Engine(prob)
    se, st = eng.lucy_iteration(n, 1)
    eng.close()
    assert st["killed_geo"] == 0 and st["killed_int"] == 0
    car = make_benchmark_problem(32, n_photons=n, n_iter=1)
    car.dust = prob.dust
    car.density = np.concatenate([np.full((1, 32, 32, 32), prob.density[d, 0]) for d in range(prob.n_dust)], axis=0)
    car.sources = prob.sources
    eng = hyperion_amd.Engine(car)
    sc, stc = eng.lucy_iteration(n, 1)
    eng.close()
    assert st["interactions"] == pytest.approx(stc["interactions"], rel=5e-3)
    assert st["energy_current"] == pytest.approx(stc["energy_current"], rel=1e-12)
    wv = prob.density * prob.vor_volume[None]
    wc = car.density * car.volumes
    assert (se * wv).sum() == pytest.approx((sc * wc).sum(), rel=5e-3)
    assert st["crossings"] / n > 30


# --- cluster-tiled Lucy iteration (hyp_vtile.h, lucy_mode=1): same packets, same answer ---------------------

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
        np.testing.assert_array_equal(a == 0, b == 0)
    n_cl = eng.get_option("vt_clusters")
    eng.close(); orc.close()
    return a, sa, n_cl


def test_vtile_config5_small_many_clusters():
    """configs[4] in small (400 cells, two polarising species, point + external box source) through the cluster-tiled
    schedule with clusters of 16 cells: packets change cluster every few crossings; slot pools far smaller than the
    packet count, generations all the way down (no drain launch), one pool and three."""
    prob, _ = golden_problem("vor_config5.npz")
    a, st, n_cl = run_tiled(prob, 60000, iters=2, vt_cells=16, tile_slots=8192, tile_task=512, tile_drain=0, tile_poll=1)
    assert n_cl == 25 and st["killed_geo"] == 0
    run_tiled(prob, 60000, vt_cells=40, tile_slots=24576, tile_pools=3, tile_task=256, tile_drain=0)
    run_tiled(prob, 60000, vt_cells=7, tile_slots=16384, tile_pools=2, tile_drain=500)
    run_tiled(prob, 60000)          # default cluster size: the whole grid is two clusters


def test_vtile_lattice_ties():
    """A (jittered) lattice tessellation: packets from the central source run along cell edges and through vertices,
    where several bisector planes are hit at once -- the guard band of the division-free wall search hands those steps
    to the reference's loop."""
    pv, _ = golden_problem("vor_lattice.npz")
    run_tiled(pv, 200000, vt_cells=12, tile_slots=32768, tile_drain=0)


def test_vtile_spherical_source_reabsorption():
    """Packets re-absorbed by a stellar sphere (LS_REABS) on the tiled Voronoi schedule."""
    prob, _ = golden_problem("vor_config5.npz")
    prob.sources = [Source(type="sphere", luminosity=LSUN, temperature=5000.0, position=(0.05 * PC, -0.02 * PC, 0.03 * PC), radius=0.08 * PC)]
    run_tiled(prob, 40000, vt_cells=16, tile_slots=8192, tile_drain=100)


def test_vtile_matches_persistent_at_scale():
    """The real 100 000-site tessellation at 2e6 packets: both GPU schedules walk the same packets; integer tallies
    equal, sums equal to rounding; the tiled one is what lucy_mode -1 picks at this size."""
    from cases import voronoi_big_problem
    prob = voronoi_big_problem(n_photons=2_000_000)
    res = []
    for mode in (0, -1):
        eng = hyperion_amd.Engine(prob)
        eng.set_option("lucy_mode", mode)
        res.append(eng.lucy_iteration(2_000_000, 1))
        assert eng.get_option("last_lucy_mode") == (0 if mode == 0 else 1)
        if mode:
            assert eng.get_option("vt_clusters") > 100 and eng.get_option("vt_max_cells") > 256      # 16-bit cell indices: a whole CU's LDS per cluster
            # the FP32 filter names the winning wall on all but a handful of the ~1e8 steps (the rest run the reference's loop)
            assert eng.get_option("last_vt_exact_steps") < 1e-3 * res[-1][1]["crossings"]
        eng.close()
    (a, sa), (b, sb) = res
    for k in INT_KEYS:
        assert sa[k] == sb[k], (k, sa, sb)
    assert_parity(a, b)


def test_config5_at_baseline_packet_count():
    """BASELINE configs[4] as BASELINE.json states it: the 100 000-site tessellation, two anisotropic polarising species,
    point + external source, 1e8 packets (cluster-tiled schedule, ~280 generations).  Nothing killed, every packet accounted
    for, absorbed energy = what the cells hold, and the interaction count per packet of the same medium at 1e7 packets
    (uniform density: it does not depend on the packet count beyond noise)."""
    from cases import voronoi_big_problem
    n = 100_000_000
    prob = voronoi_big_problem(n_photons=n)
    eng = hyperion_amd.Engine(prob)
    se, st = eng.lucy_iteration(n, 1)
    assert eng.get_option("last_lucy_mode") == 1 and eng.get_option("last_generations") > 50
    assert st["killed_geo"] == 0 and st["killed_int"] == 0 and st["n_packets"] == n
    w = prob.density * prob.vor_volume[None]
    tot = np.array(st["energy_abs_tot"][:prob.n_dust])
    np.testing.assert_allclose((se * w).sum(axis=1), tot, rtol=1e-10)
    _, s7 = eng.lucy_iteration(10_000_000, 2)
    eng.close()
    assert st["interactions"] / n == pytest.approx(s7["interactions"] / 1e7, rel=2e-3)
    assert st["crossings"] / n == pytest.approx(s7["crossings"] / 1e7, rel=2e-3)


def test_a_grid_beyond_the_tile_builder_runs_on_the_persistent_kernel_in_auto_mode():
    """More clusters than the tiled schedule's tables hold (vt_cells = 1: one cluster per cell, 100 000 > HYP_TILE_MAX_BRICKS): auto
    mode falls back to the persistent kernel -- as large AMR hierarchies and deep octrees did before the tiled schedules existed --
    and only a FORCED tiled iteration reports the limit."""
    from cases import voronoi_big_problem
    from hyperion_amd.engine import EngineError
    prob = voronoi_big_problem(n_photons=2_000_000)
    eng = hyperion_amd.Engine(prob)
    eng.set_option("vt_cells", 1)
    a, sa = eng.lucy_iteration(2_000_000, 1)
    assert eng.get_option("last_lucy_mode") == 0 and sa["killed_geo"] == 0
    eng.set_option("lucy_mode", 1)
    with pytest.raises(EngineError, match="too many cells"):
        eng.lucy_iteration(2_000_000, 2)
    eng.set_option("lucy_mode", -1)
    eng.set_option("vt_cells", 0)          # the default clusters fit: the tiled schedule again, same packets as the persistent kernel
    b, sb = eng.lucy_iteration(2_000_000, 1)
    assert eng.get_option("last_lucy_mode") == 1
    eng.close()
    for k in INT_KEYS:
        assert sa[k] == sb[k], (k, sa, sb)
