"""Spherical sources (source_type.f90 'sphere': emit_from_sphere, limb darkening, re-absorption of
packets that run into the star -- iter_lucy.f90:155-185, iter_final.f90:213-243) on the GPU:
parity with the CPU oracle on identical Philox streams."""
import numpy as np
import pytest

import hyperion_amd
from cases import assert_parity, imaging_problem
from hyperion_amd.benchmark import LSUN, PC, make_benchmark_problem
from hyperion_amd.problem import Source
from oracle_lib import Oracle

pytestmark = pytest.mark.gpu
INT_KEYS = ("crossings", "interactions", "killed_geo", "killed_int")


def star_problem(limb, tau=3.0, n=16, radius=0.3):
    p = make_benchmark_problem(n, tau=tau)
    p.sources = [Source(type="sphere", luminosity=LSUN, position=(0.1 * PC, 0.0, -0.05 * PC), radius=radius * PC,
                        temperature=5000.0, limb_darkening=limb),
                 Source(type="point", luminosity=0.3 * LSUN, position=(-0.7 * PC, 0.5 * PC, 0.2 * PC), temperature=3000.0)]
    return p


def run_both(prob, n, iters=1, n_img=0, ray=None, **opts):
    eng, orc = hyperion_amd.Engine(prob), Oracle(prob)
    for k, v in opts.items():
        eng.set_option(k, v)
    for it in range(1, iters + 1):
        a, sa = eng.lucy_iteration(n, it)
        b, sb = orc.lucy_iteration(n, it)
        for k in INT_KEYS:
            assert sa[k] == sb[k], (k, sa, sb)
        assert_parity(a, b, atol_rel=1e-10)
    if n_img:
        ra, sa = eng.final_iteration(n_img)
        rb, sb = orc.final_iteration(n_img)
        for k in INT_KEYS:
            assert sa[k] == sb[k], (k, sa, sb)
        if ray:
            ra, _ = eng.raytracing_iteration(*ray)
            rb, _ = orc.raytracing_iteration(*ray)
        for ga, gb in zip(ra, rb):
            for name in gb:
                np.testing.assert_allclose(ga[name], gb[name], rtol=1e-9, atol=1e-10 * np.nanmax(np.abs(gb[name])), err_msg=name)
    eng.close(); orc.close()
    return a, sa


@pytest.mark.parametrize("limb", [False, True])
def test_star_with_reabsorption(limb):
    """A star of 0.3 pc radius in an optically thick cube: a good fraction of the packets scatter
    back into it and are re-emitted from its surface."""
    a, st = run_both(star_problem(limb), 100000, iters=2)
    assert st["killed_int"] == 0
    # the inside of the star sees no radiation
    p = star_problem(limb)
    c = 0.5 * (p.walls[0][1:] + p.walls[0][:-1])
    z, y, x = np.meshgrid(c, c, c, indexing="ij")
    r = np.sqrt((x - 0.1 * PC) ** 2 + y ** 2 + (z + 0.05 * PC) ** 2)
    assert a[0][r < 0.15 * PC].max() < 1e-3 * a[0][(r > 0.35 * PC) & (r < 0.5 * PC)].mean()


def test_reabsorption_limit_kills_packets():
    """Two stars facing each other in a thin medium: a packet that goes A -> B -> A without an
    interaction in between exceeds n_reabs_max = 1 and is killed (iter_lucy.f90:178-183)."""
    p = star_problem(False, tau=0.2)
    p.sources = [Source(type="sphere", luminosity=LSUN, position=(-0.35 * PC, 0.0, 0.0), radius=0.3 * PC, temperature=5000.0),
                 Source(type="sphere", luminosity=LSUN, position=(0.35 * PC, 0.0, 0.0), radius=0.3 * PC, temperature=4000.0, limb_darkening=True)]
    p.config.n_reabs_max = 1
    a, st = run_both(p, 60000)
    assert st["killed_int"] > 100
    p.config.n_reabs_max = 100
    a, st = run_both(p, 60000)
    assert st["killed_int"] == 0


def test_star_imaging_and_raytracing():
    p = imaging_problem(12, tau=1.5)
    p.sources = star_problem(True).sources
    run_both(p, 30000, n_img=60000)
    p.config.raytracing = True
    run_both(p, 30000, n_img=40000, ray=(20000, 20000))


@pytest.mark.parametrize("limb", [False, True])
def test_star_in_the_brick_tiled_schedule(limb):
    """The slot records carry the path length of the current integration and the nearest source,
    so re-absorption works across brick visits; re-emission happens in tile_prepare."""
    p = star_problem(limb, n=40, tau=3.0)
    run_both(p, 150000, iters=2, lucy_mode=1, tile_slots=32768, tile_pools=2, tile_drain=0)
    run_both(p, 150000, lucy_mode=1, tile_slots=32768, tile_drain=50000)


def test_two_stars_tiled_with_reabsorption_limit():
    p = star_problem(False, n=40, tau=0.2)
    p.sources = [Source(type="sphere", luminosity=LSUN, position=(-0.35 * PC, 0.0, 0.0), radius=0.3 * PC, temperature=5000.0),
                 Source(type="sphere", luminosity=LSUN, position=(0.35 * PC, 0.0, 0.0), radius=0.3 * PC, temperature=4000.0, limb_darkening=True)]
    p.config.n_reabs_max = 1
    a, st = run_both(p, 100000, lucy_mode=1, tile_slots=32768, tile_drain=0)
    assert st["killed_int"] > 100
    a, st = run_both(p, 100000, lucy_mode=1, tile_slots=32768, tile_drain=10 ** 9)      # everything through the drain kernel
    assert st["killed_int"] > 100


def test_auto_schedule_with_a_star_is_tiled():
    p = star_problem(False, n=64, tau=1.0, radius=0.05)
    eng = hyperion_amd.Engine(p)
    eng.lucy_iteration(4000000, 1, want_output=False)
    assert eng.get_option("last_lucy_mode") == 1
    eng.close()


def test_plane_parallel_and_point_collection_sources():
    """emit_from_plane_parallel (source_type.f90:935-975: a disc of parallel rays, never peeled) and
    emit_from_point_collection (:570-598: one member picked by luminosity)."""
    p = imaging_problem(12, tau=1.0)
    rng = np.random.default_rng(11)
    pts = rng.uniform(-0.8 * PC, 0.8 * PC, (7, 3))
    lum = rng.uniform(0.1, 1.0, 7) * LSUN
    p.sources = [Source(type="plane_parallel", luminosity=0.7 * LSUN, position=(0.0, 0.0, 0.7 * PC), radius=0.5 * PC,
                        direction=(160.0, 30.0), temperature=4500.0, peeloff=False),
                 Source(type="point_collection", luminosity=float(lum.sum()), points=pts, point_luminosity=lum, temperature=7000.0)]
    a, st = run_both(p, 60000, iters=2, n_img=60000)
    assert st["killed_geo"] == 0
    # the beam travels towards -z (theta = 160 deg): its entry side is the hotter one
    assert a[0][-3:].mean() > a[0][:3].mean()
    p.sources[0].peeloff = True
    with pytest.raises(hyperion_amd.EngineError, match="plane parallel sources cannot be peeled"):
        hyperion_amd.Engine(p)


def test_spotted_star_parity_and_spectrum():
    """Spots on a spherical source (source_type.f90:150-188, 421-427, 632-636): the reference's own regression
    model (hyperion/model/tests/test_spot_source.py: sphere and spot with disjoint emission bands) through the
    C ABI, parity with the oracle on identical streams, and a dusty model with a blackbody spot in the Lucy,
    imaging and monochromatic iterations."""
    from test_oracle_units import spotted_star_problem
    from hyperion_amd.problem import PeeledImages, Spot
    p = spotted_star_problem()
    eng, orc = hyperion_amd.Engine(p), Oracle(p)
    ra, sa = eng.final_iteration(100000)
    rb, sb = orc.final_iteration(100000)
    eng.close(); orc.close()
    np.testing.assert_allclose(ra[0]["sed"], rb[0]["sed"], rtol=1e-9, atol=1e-12 * rb[0]["sed"].max())
    sed = ra[0]["sed"][0, 0]
    assert sed[1].sum() < sed[0].sum()              # the far side does not see the spot

    q = make_benchmark_problem(10, tau=1.0)
    q.sources = [Source(type="sphere", luminosity=LSUN, temperature=5000.0, position=(0.05 * PC, 0.0, -0.02 * PC), radius=0.03 * PC,
                        limb_darkening=True,
                        spots=[Spot(longitude=40.0, latitude=200.0, radius=30.0, luminosity=0.7 * LSUN, temperature=9000.0),
                               Spot(longitude=130.0, latitude=20.0, radius=10.0, luminosity=0.2 * LSUN, temperature=3000.0)])]
    q.peeled = [PeeledImages(theta=[40.0, 130.0], phi=[200.0, 30.0], n_wav=5, wav_min=0.1, wav_max=100.0, n_x=4, n_y=4,
                             x_min=-1.5 * PC, x_max=1.5 * PC, y_min=-1.5 * PC, y_max=1.5 * PC, n_ap=1, ap_min=2 * PC, ap_max=2 * PC)]
    eng, orc = hyperion_amd.Engine(q), Oracle(q)
    for it in (1, 2):
        a, sa = eng.lucy_iteration(30000, it)
        b, sb = orc.lucy_iteration(30000, it)
        for k in ("crossings", "interactions", "killed_geo", "killed_int"):
            assert sa[k] == sb[k], (k, sa, sb)
        assert_parity(a, b)
    ra, sa = eng.final_iteration(30000)
    rb, sb = orc.final_iteration(30000)
    eng.close(); orc.close()
    assert sa["crossings"] == sb["crossings"] and sa["interactions"] == sb["interactions"]
    for name in rb[0]:
        np.testing.assert_allclose(ra[0][name], rb[0][name], rtol=1e-9, atol=1e-11 * np.nanmax(np.abs(rb[0][name])), err_msg=name)


def test_star_images_on_the_deferred_schedule_like_on_the_general_kernel():
    """A star with a radius (limb darkening, re-absorption): the imaging iteration on the deferred schedule's GEN kernels (default since
    round 4; the lean specialisation of the general kernel that round 3 used here is gone) against the general kernel: tallies
    identical, cubes equal to summation order."""
    from hyperion_amd.problem import PeeledImages
    prob = star_problem(True, tau=2.0, n=12)
    prob.peeled = [PeeledImages(theta=[30.0, 100.0], phi=[20.0, 250.0], n_wav=3, wav_min=0.1, wav_max=1000.0, n_x=8, n_y=8,
                                x_min=-1.5 * PC, x_max=1.5 * PC, y_min=-1.5 * PC, y_max=1.5 * PC, n_ap=2, ap_min=0.3 * PC, ap_max=2.0 * PC,
                                compute_stokes=True, uncertainties=True, track_origin="detailed")]
    eng = hyperion_amd.Engine(prob)
    eng.lucy_iteration(20000, 1, want_output=False)
    assert eng.get_option("plain_imaging") == 0 and eng.get_option("gen_defer") == 1
    ra, sa = eng.final_iteration(30000)
    assert eng.get_option("last_defer_rounds") >= 1
    eng.set_option("gen_defer", 0)
    rb, sb = eng.final_iteration(30000)
    assert eng.get_option("last_defer_rounds") == 0
    eng.close()
    for k in INT_KEYS:
        assert sa[k] == sb[k], (k, sa, sb)
    assert sa["energy_current"] == pytest.approx(sb["energy_current"], rel=1e-12)
    for ga, gb in zip(ra, rb):
        for name in gb:
            np.testing.assert_allclose(ga[name], gb[name], rtol=1e-9, atol=1e-11 * np.nanmax(np.abs(gb[name])), err_msg=name)   # sums in another order
