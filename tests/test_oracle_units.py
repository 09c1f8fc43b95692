"""Known-answer checks that pin the CPU oracle (no GPU needed)."""
import numpy as np
import pytest

import oracle_lib
from oracle_lib import Oracle, lib, philox
from cases import golden_problem, ragged_grid_problem
from hyperion_amd.benchmark import PC, make_benchmark_problem
from hyperion_amd.problem import Problem, RunConfig, Source


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    assert philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_uniform_stream_is_uniform():
    u = np.array([lib().orc_probe_uniform(-124902, 1, i, 0) for i in range(20000)])
    assert 0.0 <= u.min() and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1.0 / 12.0) < 0.005
    # different draw index / iteration / packet id give different numbers
    assert lib().orc_probe_uniform(-124902, 1, 7, 0) != lib().orc_probe_uniform(-124902, 1, 7, 1)
    assert lib().orc_probe_uniform(-124902, 1, 7, 0) != lib().orc_probe_uniform(-124902, 2, 7, 0)


def unit_grid(n=128):
    p = make_benchmark_problem(4)
    w = np.linspace(-1.0, 1.0, n + 1)
    return Problem(walls=[w, w, w], density=np.ones((1, n, n, n)), dust=p.dust,
                   sources=[Source(luminosity=1.0, temperature=5000.0)], config=RunConfig())


def test_walk_known_answer_from_reference_geometry():
    """the reference's own find_wall/next_cell (flang build
    of src/grid/grid_geometry_cartesian_3d.f90) walks v=(0.6,0.48,0.64) from the
    centre of a 128^3 unit grid in 149 crossings, path 1.5625000000000002."""
    o = Oracle(unit_grid(128))
    n, path = o.walk_ray([0.0, 0.0, 0.0], [0.6, 0.48, 0.64])
    assert n == 149
    assert path == 1.5625000000000002


def test_walk_mean_crossings_isotropic():
    """Same probe: 2e5 isotropic rays from the centre -> 117.4 crossings, mean
    path 1.2206 half-widths (here 2e4 rays)."""
    o = Oracle(unit_grid(128))
    rng = np.random.RandomState(1)
    mu = rng.uniform(-1, 1, 20000)
    ph = rng.uniform(0, 2 * np.pi, 20000)
    s = np.sqrt(1 - mu * mu)
    v = np.stack([s * np.cos(ph), s * np.sin(ph), mu], axis=1)
    res = np.array([o.walk_ray([0.0, 0.0, 0.0], vi) for vi in v])
    assert abs(res[:, 0].mean() - 117.4) < 0.5
    assert abs(res[:, 1].mean() - 1.2206) < 0.005


def test_walk_axis_aligned_and_scaled_grids():
    for scale in (1e-20, 1.0, 1e20):
        p = ragged_grid_problem(scale=scale)
        o = Oracle(p)
        n, path = o.walk_ray([0.0, 0.0, 0.0], [1.0, 0.0, 0.0])
        assert n >= 1 and path == pytest.approx(1.0 * scale, rel=1e-14)
        # from a vertex along a diagonal of the box
        x0 = [p.walls[0][0], p.walls[1][0], p.walls[2][0]]
        n, path = o.walk_ray(x0, list(np.ones(3) / np.sqrt(3.0)))
        assert n >= 1 and path == pytest.approx(2.0 * np.sqrt(3.0) * scale, rel=1e-12)


def test_planck_sampling_mean():
    """<h nu / kT> of the Planck energy distribution is 360 zeta(5)/pi^4 = 3.8322."""
    T = 6000.0
    h, k = 6.6260755e-27, 1.380658e-16
    nu = np.array([lib().orc_probe_planck(T, -1, i) for i in range(40000)])
    x = h * nu / (k * T)
    assert abs(x.mean() - 3.8322) < 0.03
    assert abs(np.median(x) - 3.503) < 0.04     # median of the same distribution


def test_emissivity_sampling_follows_the_pdf():
    prob, _ = golden_problem("car_specific_energy.False.False.npz")
    o = Oracle(prob)
    d = prob.dust[0]
    j = 40
    xi = (np.arange(20000) + 0.5) / 20000
    nu = np.array([lib().orc_probe_sample_jnu(o.h, 0, j, 0.0, x) for x in xi])
    assert np.all(np.diff(nu) >= 0)              # inverse CDF is monotonic
    # the energy-weighted mean frequency of the table
    x, y = d.emiss_nu, d.emiss_jnu[:, j]
    lx = np.log(x)
    mean_tab = np.trapezoid(y * x * x, lx) / np.trapezoid(y * x, lx)
    assert nu.mean() == pytest.approx(mean_tab, rel=0.01)
    # interpolation between neighbouring emissivities is log-linear
    a = lib().orc_probe_sample_jnu(o.h, 0, j, 0.0, 0.3)
    b = lib().orc_probe_sample_jnu(o.h, 0, j, 1.0, 0.3)
    m = lib().orc_probe_sample_jnu(o.h, 0, j, 0.5, 0.3)
    assert m == pytest.approx(np.sqrt(a * b), rel=1e-12)
    assert b == pytest.approx(lib().orc_probe_sample_jnu(o.h, 0, j + 1, 0.0, 0.3), rel=1e-12)


def test_optical_constants_interpolation():
    prob, _ = golden_problem("car_specific_energy.False.False.npz")
    o = Oracle(prob)
    d = prob.dust[0]
    out = (oracle_lib.C.c_double * 3)()
    for i in (0, 10, 50, d.nu.size - 1):
        lib().orc_probe_optconsts(o.h, 0, float(d.nu[i]), out)
        assert out[0] == pytest.approx(d.chi[i], rel=1e-12)
        assert out[1] == pytest.approx(d.albedo[i], rel=1e-12)
    nu = np.sqrt(d.nu[10] * d.nu[11])
    lib().orc_probe_optconsts(o.h, 0, float(nu), out)
    assert out[0] == pytest.approx(np.sqrt(d.chi[10] * d.chi[11]), rel=1e-12)
    assert out[2] == pytest.approx(out[0] * (1 - out[1]), rel=1e-15)


def _scatter(o, nu, a, s, pid):
    a_in = (oracle_lib.C.c_double * 4)(*a)
    s_in = (oracle_lib.C.c_double * 4)(*s)
    a_out = (oracle_lib.C.c_double * 4)()
    s_out = (oracle_lib.C.c_double * 4)()
    lib().orc_probe_scatter(o.h, 0, nu, a_in, s_in, -5, pid, a_out, s_out)
    return np.array(a_out), np.array(s_out)


def test_scattering_angles_and_stokes():
    prob, _ = golden_problem("car_specific_energy.False.False.npz")
    o = Oracle(prob)
    d = prob.dust[0]
    nu = float(d.nu[60])
    th, ph = 1.1, 0.7
    a = [np.cos(th), np.sin(th), np.cos(ph), np.sin(ph)]
    v0 = np.array([a[1] * a[2], a[1] * a[3], a[0]])
    cos_t, q = [], []
    for pid in range(4000):
        ao, so = _scatter(o, nu, a, [1.0, 0.0, 0.0, 0.0], pid)
        assert ao[0] ** 2 + ao[1] ** 2 == pytest.approx(1.0, abs=1e-12)
        assert ao[2] ** 2 + ao[3] ** 2 == pytest.approx(1.0, abs=1e-12)
        assert so[0] == 1.0 and so[1] ** 2 + so[2] ** 2 + so[3] ** 2 <= 1.0 + 1e-9
        v1 = np.array([ao[1] * ao[2], ao[1] * ao[3], ao[0]])
        cos_t.append(v0 @ v1)
        q.append(so[1])
    cos_t = np.array(cos_t)
    # mean scattering cosine g of the tabulated phase function at this frequency
    P1 = d.P1[60]
    g = np.trapezoid(P1 * d.mu, d.mu) / np.trapezoid(P1, d.mu)
    assert cos_t.mean() == pytest.approx(g, abs=0.03)
    assert abs(np.mean(q)) < 0.2


def _table_P(d, inu, mu):
    """P1..P3 of the scattering table at a tabulated frequency, linear in mu (interp2d, dust_type_4elem.f90:543-546)."""
    return [np.interp(mu, d.mu, P[inu]) for P in (d.P1, d.P2, d.P3)]


def test_polarisation_degree_of_one_scattering_known_answer():
    """The AMPLITUDE of scatter_stokes (src/dust/dust_type_4elem.f90:603-690) against closed forms that follow from its
    Mueller matrix R = [[P1 P2 0 0] [P2 P1 0 0] [0 0 P3 -P4] [0 0 P4 P3]] between two rotations (which preserve Q^2 + U^2),
    on the reference's kmh dust (P2 != 0), packet by packet, no statistics:

      * unpolarised light: degree of polarisation = |P2 / P1| at the sampled angle, whatever the directions;
      * ... travelling along +z: the scattering plane is the new meridian plane, so Q = P2 / P1 with its sign and U = 0;
      * light polarised (1, q, 0, 0): Q'^2 + U'^2 = ((P2 + P1 q c)^2 + (P3 q s)^2) / (P1 + P2 q c)^2 with c, s = cos, sin of
        twice the azimuth of the scattering plane about the old direction, measured from its meridian plane -- which is
        read off the two directions, not from the code under test;
      * the sample mean of Q / I over unpolarised scatterings = int P2 dmu / int P1 dmu (mu is drawn from P1)."""
    prob, _ = golden_problem("car_specific_energy.False.False.npz")
    o = Oracle(prob)
    d = prob.dust[0]
    inu = 60
    nu = float(d.nu[inu])
    assert np.abs(d.P2[inu]).max() > 0.1 * d.P1[inu].mean()       # a polarising table

    def vec(a):
        return np.array([a[1] * a[2], a[1] * a[3], a[0]])

    # (1) unpolarised, generic direction; (2) along +z
    th, ph = 2.1, 4.0
    a_gen = [np.cos(th), np.sin(th), np.cos(ph), np.sin(ph)]
    a_z = [1.0, 0.0, 1.0, 0.0]
    qs = []
    for pid in range(3000):
        ao, so = _scatter(o, nu, a_gen, [1.0, 0.0, 0.0, 0.0], pid)
        mu = float(vec(a_gen) @ vec(ao))
        P1, P2, P3 = _table_P(d, inu, mu)
        assert np.hypot(so[1], so[2]) == pytest.approx(abs(P2 / P1), rel=1e-6, abs=1e-9)
        assert so[3] == 0.0
        ao, so = _scatter(o, nu, a_z, [1.0, 0.0, 0.0, 0.0], pid)
        P1, P2, P3 = _table_P(d, inu, float(ao[0]))
        assert so[1] == pytest.approx(P2 / P1, rel=1e-6, abs=1e-9) and abs(so[2]) < 1e-9
        qs.append(so[1])
    expected = np.trapezoid(d.P2[inu], d.mu) / np.trapezoid(d.P1[inu], d.mu)
    assert abs(expected) > 0.05
    assert np.mean(qs) == pytest.approx(expected, abs=4.0 * np.std(qs) / np.sqrt(len(qs)))
    assert np.mean(qs) == pytest.approx(expected, rel=0.1)
    # (3) polarised input
    q_in = 0.6
    v0 = vec(a_gen)
    e_theta = np.array([np.cos(th) * np.cos(ph), np.cos(th) * np.sin(ph), -np.sin(th)])
    for pid in range(3000):
        ao, so = _scatter(o, nu, a_gen, [1.0, q_in, 0.0, 0.0], pid)
        v1 = vec(ao)
        mu = float(v0 @ v1)
        t = v1 - mu * v0
        if t @ t < 1e-12:
            continue
        t /= np.sqrt(t @ t)
        cos_c = -(t @ e_theta)                  # angle at the old direction between the arcs to the pole and to the new direction
        c2, s2sq = 2.0 * cos_c ** 2 - 1.0, 1.0 - (2.0 * cos_c ** 2 - 1.0) ** 2
        P1, P2, P3 = _table_P(d, inu, mu)
        want = ((P2 + P1 * q_in * c2) ** 2 + P3 ** 2 * q_in ** 2 * s2sq) / (P1 + P2 * q_in * c2) ** 2
        assert so[0] == 1.0
        assert so[1] ** 2 + so[2] ** 2 == pytest.approx(want, rel=1e-6, abs=1e-12)
    o.close()


def test_isotropic_dust_scatters_isotropically():
    p = make_benchmark_problem(4)
    o = Oracle(p)
    a = [0.3, np.sqrt(1 - 0.09), 1.0, 0.0]
    v0 = np.array([a[1], 0.0, a[0]])
    cs = []
    for pid in range(6000):
        ao, so = _scatter(o, 1e14, a, [1.0, 0.0, 0.0, 0.0], pid)
        cs.append(v0 @ np.array([ao[1] * ao[2], ao[1] * ao[3], ao[0]]))
        assert abs(so[1]) < 1e-12 and abs(so[2]) < 1e-12
    cs = np.array(cs)
    assert abs(cs.mean()) < 0.03 and abs((cs ** 2).mean() - 1.0 / 3.0) < 0.02


def test_error_messages_match_the_reference():
    """hyperion/model/tests/test_fortran.py:12-84 greps the log for these."""
    from oracle_lib import OracleError
    p = make_benchmark_problem(4)
    p.sources[0].position = (5 * PC, 0.0, 0.0)
    o = Oracle(p)
    with pytest.raises(OracleError, match="photon was not emitted inside a cell"):
        o.lucy_iteration(10, 1)
    p = make_benchmark_problem(4)
    p.sources[0].temperature = 1e9            # blackbody far above the dust table
    o = Oracle(p)
    with pytest.raises(OracleError, match=r"photon frequency .* is outside the range defined for the dust optical properties"):
        o.lucy_iteration(100, 1)
    p = make_benchmark_problem(4)
    p.sources[0].temperature = None
    p.sources[0].spectrum_nu = np.array([1e12, 1e14, 1e13])
    p.sources[0].spectrum_fnu = np.ones(3)
    with pytest.raises(OracleError, match="spectrum frequency should be monotonically increasing"):
        Oracle(p)


def test_energy_conservation_and_optically_thin_limit():
    """sum(E rho V) = absorbed luminosity, and for
    optically thin grey dust around a point source E(r) -> kappa L / (4 pi r^2)
    (grid_propagate_3d.f90:153-154 + grid_physics_3d.f90:515)."""
    from hyperion_amd.benchmark import LSUN
    p = make_benchmark_problem(9, tau=1e-4)
    o = Oracle(p)
    se, st = o.lucy_iteration(400000, 1)
    tot = (se * p.density * p.volumes).sum()
    assert tot == pytest.approx(st["energy_abs_tot"][0], rel=1e-12)
    c = 0.5 * (p.walls[0][1:] + p.walls[0][:-1])
    z, y, x = np.meshgrid(c, c, c, indexing="ij")
    r = np.sqrt(x * x + y * y + z * z)
    kappa = 0.5
    expect = kappa * LSUN / (4 * np.pi * r * r)
    sel = (r > 0.45 * PC) & (r < 0.9 * PC)          # away from the source cell
    ratio = se[0][sel] / expect[sel]
    assert ratio.mean() == pytest.approx(1.0, abs=0.02)


@pytest.mark.parametrize("limb", [False, True])
def test_spherical_source_far_field_and_shadow(limb):
    """emit_from_sphere (source_type.f90:604-690): a Lambertian (or limb-darkened) sphere radiates
    isotropically in the far field, E(r) -> kappa L / (4 pi r^2) in thin dust, and nothing reaches
    its inside; the few packets that scatter back into it are re-emitted (iter_lucy.f90:155-185)."""
    from hyperion_amd.benchmark import LSUN
    from hyperion_amd.problem import Source
    p = make_benchmark_problem(11, tau=1e-4)
    p.sources = [Source(type="sphere", luminosity=LSUN, position=(0.0, 0.0, 0.0), radius=0.2 * PC, temperature=6000.0,
                        limb_darkening=limb)]
    o = Oracle(p)
    se, st = o.lucy_iteration(400000, 1)
    assert st["killed_geo"] == 0 and st["killed_int"] == 0 and st["energy_current"] == 400000
    c = 0.5 * (p.walls[0][1:] + p.walls[0][:-1])
    z, y, x = np.meshgrid(c, c, c, indexing="ij")
    r = np.sqrt(x * x + y * y + z * z)
    expect = 0.5 * LSUN / (4 * np.pi * r * r)
    sel = (r > 0.5 * PC) & (r < 0.9 * PC)
    assert (se[0][sel] / expect[sel]).mean() == pytest.approx(1.0, abs=0.03)
    # (the floor is the lowest specific energy of the dust's emissivity table: check_energy_abs)
    assert se[0][r < 0.08 * PC].max() == se.min() < 1e-3 * se[0][sel].min()


def test_limb_darkening_sampler_matches_its_pdf():
    """ran_mu_limb(1.5, 1) (source_type.f90:982-1086): <mu> of the pdf 1.5 mu^2 + mu on [0, 1] is
    (1.5/4 + 1/3) / (1.5/3 + 1/2) = 0.70833; a Lambertian surface has <mu> = 2/3.  Read off the
    far-field-normalised surface brightness: flux ratios do not depend on it, so compare means of
    the emission angle through the energy deposited just above the surface along the normal --
    here simply through the analytic CDF inversion of the cubic."""
    xi = np.linspace(0.001, 0.999, 500)
    s, t = (1.5 / 3) / (1.5 / 3 + 0.5), 0.5 / (1.5 / 3 + 0.5)
    # numpy roots of s mu^3 + t mu^2 - xi = 0 in (0, 1)
    mu = np.array([[r.real for r in np.roots([s, t, 0.0, -x]) if abs(r.imag) < 1e-12 and 0 <= r.real <= 1][0] for x in xi])
    assert np.trapz(mu, xi) / (xi[-1] - xi[0]) == pytest.approx(0.70833, abs=2e-3)


def test_rotate_and_difference_are_inverse_and_geometric():
    """rotate_angle3d / difference_angle3d (fortranlib, restated): the deflection
    angle between old and new direction is the local polar angle, and
    difference(rotate(local)) == local."""
    rng = np.random.RandomState(5)
    D4 = oracle_lib.C.c_double * 4
    for _ in range(2000):
        tl, pl, tc, pc = rng.uniform(0.05, np.pi - 0.05), rng.uniform(0, 2 * np.pi), rng.uniform(0.05, np.pi - 0.05), rng.uniform(0, 2 * np.pi)
        loc = D4(np.cos(tl), np.sin(tl), np.cos(pl), np.sin(pl))
        co = D4(np.cos(tc), np.sin(tc), np.cos(pc), np.sin(pc))
        fin, back = D4(), D4()
        lib().orc_probe_rotate(loc, co, fin, back)
        v0 = np.array([co[1] * co[2], co[1] * co[3], co[0]])
        v1 = np.array([fin[1] * fin[2], fin[1] * fin[3], fin[0]])
        assert v0 @ v1 == pytest.approx(np.cos(tl), abs=1e-12)
        np.testing.assert_allclose(list(back), list(loc), atol=1e-7)
        # orientation pinned by the reference's golden Stokes U: a local azimuth in
        # (0, pi) turns the direction towards increasing phi
        dphi = np.arctan2(fin[3], fin[2]) - pc
        s = np.sin(dphi)
        if abs(s) > 1e-9:
            assert (s > 0) == (np.sin(pl) > 0)


def test_external_box_source_uniform_field_and_voronoi_lattice():
    """Oracle-level checks of the pieces that have no reference golden: (a) a
    Lambertian box source fills the box with a uniform isotropic field,
    E = 4 kappa L / A in the thin limit (source_type.f90:822-907); (b) a Voronoi
    tessellation of a (jittered) cubic lattice reproduces the Cartesian result on
    the same Philox streams (grid_geometry_voronoi.f90:322-402)."""
    from hyperion_amd.benchmark import LSUN
    p = make_benchmark_problem(8, tau=1e-6)
    p.sources = [Source(type="extern_box", luminosity=LSUN, temperature=5000.0, box=(-PC, PC, -PC, PC, -PC, PC))]
    se, st = Oracle(p).lucy_iteration(400000, 1)
    assert st["killed_geo"] == 0
    assert se.mean() == pytest.approx(4.0 * 0.5 * LSUN / (24.0 * PC * PC), rel=1e-2)
    assert se.std() / se.mean() < 0.05
    pv, _ = golden_problem("vor_lattice.npz")
    a, sa = Oracle(pv).lucy_iteration(100000, 1)
    pc_ = make_benchmark_problem(6)
    pc_.sources[0].position = pv.sources[0].position
    b, sb = Oracle(pc_).lucy_iteration(100000, 1)
    assert sa["interactions"] == sb["interactions"] and sa["killed_geo"] == 0
    w = np.linspace(-1, 1, 7) * PC
    ix, iy, iz = (np.searchsorted(w, pv.vor_sites[:, k]) - 1 for k in range(3))
    np.testing.assert_allclose(a[0], b[0][iz, iy, ix], rtol=0.03, atol=0.01 * b.max())


def _polar_problem(grid_type, walls, tau):
    from hyperion_amd.benchmark import LSUN, load_test_dust
    shape = tuple(w.size - 1 for w in walls)[::-1]
    return Problem(walls=walls, density=np.full((1,) + shape, tau / PC), dust=[load_test_dust()],
                   sources=[Source(type="point", luminosity=LSUN, temperature=6000.0, position=(0.0, 0.0, 0.0))],
                   config=RunConfig(), grid_type=grid_type)


def test_spherical_polar_grid_optically_thin_shells():
    """BASELINE configs[0] shape (2D spherical polar grid, central point source): a packet from the
    origin crosses every shell radially, so E = kappa L dr / (4 pi dr3 / 3) in every cell of a shell
    -- exact per packet, whatever its direction (grid_geometry_spherical_3d.f90:146-155 volumes,
    find_wall :741-1073 spheres only: radial packets never reach a cone)."""
    from hyperion_amd.benchmark import LSUN
    r = np.hstack([0.0, np.logspace(-2, 0, 12) * PC])
    p = _polar_problem("sph_pol", [r, np.linspace(0, np.pi, 9), np.array([0.0, 2 * np.pi])], 1e-6)
    o = Oracle(p)
    se, st = o.lucy_iteration(200000, 1)
    o.close()
    assert st["killed_geo"] == 0
    assert st["crossings"] >= 200000 * 12
    expect = 0.5 * LSUN * np.diff(r) / (4 * np.pi * np.diff(r ** 3) / 3.0)
    # the solid angle of a theta band is sampled by ~N dcos/2 packets
    band = se[0, 0].mean(axis=0)
    np.testing.assert_allclose(band, expect, rtol=0.02)
    assert (se * p.density * p.volumes).sum() == pytest.approx(st["energy_abs_tot"][0], rel=1e-12)


def test_cylindrical_polar_grid_optically_thin_mean_chord():
    """Cylindrical polar grid (grid_geometry_cylindrical_3d.f90): the absorbed energy inside the cylinder
    w < R, |z| < H of an optically thin uniform medium around a central source is kappa rho L <path>, with
    the mean path to the surface computed by quadrature; the off-centre walk (cylinders, z planes, phi
    half-planes) conserves it cell by cell: sum over phi and z bands equals the same integral."""
    from hyperion_amd.benchmark import LSUN
    R, H = PC, 0.5 * PC
    w = np.linspace(0.0, R, 9); z = np.linspace(-H, H, 7); ph = np.linspace(0.0, 2 * np.pi, 6)
    p = _polar_problem("cyl_pol", [w, z, ph], 1e-6)
    p.sources[0].position = (0.21 * PC, -0.13 * PC, 0.08 * PC)
    o = Oracle(p)
    se, st = o.lucy_iteration(300000, 1)
    o.close()
    assert st["killed_geo"] == 0
    absorbed = (se * p.density * p.volumes).sum()
    assert absorbed == pytest.approx(st["energy_abs_tot"][0], rel=1e-12)
    # mean chord from the source to the cylinder surface over isotropic directions (numerical quadrature)
    mu = np.linspace(-1, 1, 801); phi = np.linspace(0, 2 * np.pi, 721)[:-1]
    M, PH = np.meshgrid(mu, phi, indexing="ij")
    s = np.sqrt(1 - M * M)
    vx, vy, vz = s * np.cos(PH), s * np.sin(PH), M
    x0, y0, z0 = p.sources[0].position
    a = vx * vx + vy * vy; b = 2 * (x0 * vx + y0 * vy); c = x0 * x0 + y0 * y0 - R * R
    with np.errstate(divide="ignore", invalid="ignore"):
        t_cyl = np.where(a > 0, (-b + np.sqrt(b * b - 4 * a * c)) / (2 * a), np.inf)
        t_z = np.where(vz > 0, (H - z0) / vz, np.where(vz < 0, (-H - z0) / vz, np.inf))
    chord = np.trapezoid(np.minimum(t_cyl, t_z).mean(axis=1), mu) / 2.0
    kappa, rho = 0.5, 1e-6 / PC
    assert absorbed == pytest.approx(kappa * rho * LSUN * chord, rel=0.01)


def binned_problem(n_theta=4, n_phi=3):
    """Spherically symmetric set-up (uniform spherical polar grid, central source): every direction bin
    of the binned images must show the same SED as a peeled image from any viewing angle."""
    from hyperion_amd.problem import PeeledImages
    p = _polar_problem("sph_pol", [np.linspace(0, PC, 9), np.linspace(0, np.pi, 5), np.linspace(0, 2 * np.pi, 4)], 1.0)
    p.config.forced_first_interaction = False       # setup_rt.f90:329: binned images exclude it
    # (odd pixel counts: unscattered packets of the central source project onto (0, 0) +- round-off, which must not be a pixel edge)
    kw = dict(n_wav=5, wav_min=0.1, wav_max=1000.0, n_x=5, n_y=5, x_min=-1.5 * PC, x_max=1.5 * PC, y_min=-1.5 * PC, y_max=1.5 * PC,
              n_ap=1, ap_min=2 * PC, ap_max=2 * PC, track_origin="basic")
    p.peeled = [PeeledImages(theta=[60.0], phi=[20.0], **kw)]
    p.binned = PeeledImages(theta=[0.0], phi=[0.0], **kw)
    p.n_binned_theta, p.n_binned_phi = n_theta, n_phi
    return p


def test_binned_images_agree_with_peeled_images_in_a_symmetric_model():
    """images_binned.f90: packets leaving the grid are binned by direction (cos(theta) x phi bins) and scaled
    by n_theta n_phi L / E (binned_images_adjust_scale :34-38), i.e. to the 4 pi normalisation of peel-off."""
    p = binned_problem()
    o = Oracle(p)
    for it in (1, 2):
        o.lucy_iteration(40000, it)
    res, st = o.final_iteration(300000)
    o.close()
    assert len(res) == 2
    peel, binned = res[0]["sed"], res[1]["sed"]
    assert binned.shape == (4, 4, 12, 1, 5) and peel.shape == (4, 4, 1, 1, 5)
    I_peel = peel[0, :, 0, 0, :]                      # (origin, wavelength)
    I_bin = binned[0, :, :, 0, :]                     # (origin, bin, wavelength)
    sel = I_peel > 0.02 * I_peel.max()
    mean_bin = I_bin.mean(axis=1)
    assert np.abs(mean_bin[sel] / I_peel[sel] - 1.0).max() < 0.03
    for k in range(12):
        assert np.abs(I_bin[:, k, :][sel] / I_peel[sel] - 1.0).max() < 0.15
    # images: the summed image of a bin carries the same flux as its SED (aperture larger than the image)
    np.testing.assert_allclose(res[1]["img"][0].sum(axis=(2, 3)), binned[0, :, :, 0, :], rtol=1e-9)
    # Stokes Q, U of a symmetric model average out over the bins
    assert abs(binned[1].sum()) < 0.02 * binned[0].sum()


def test_binned_images_refuse_forced_first_interaction():
    p = binned_problem()
    p.config.forced_first_interaction = True
    with pytest.raises(oracle_lib.OracleError, match="can't use binned images with forced first interaction"):
        Oracle(p)


def map_source_problem(lte=False, n=4, tau=1e-3, seed=5):
    from hyperion_amd.benchmark import LSUN, load_test_dust
    rng = np.random.default_rng(seed)
    x = np.linspace(-PC, PC, n + 1)
    lum_map = rng.random((n, n, n)) ** 3           # strongly non-uniform
    dens = np.full((1, n, n, n), tau / PC)
    src = Source(type="map", luminosity=LSUN, map=lum_map, temperature=None if lte else 5000.0, lte=lte)
    p = Problem(walls=[x, x, x], density=dens, dust=[load_test_dust()], sources=[src], config=RunConfig())
    if lte:
        p.specific_energy = np.full(dens.shape, 1e3) * (0.5 + rng.random(dens.shape))
    return p, lum_map


def test_map_source_equals_a_point_collection_with_the_same_luminosity_distribution():
    """emit_from_map (source_type.f90:713-741): cell from the luminosity map, uniform position inside it,
    isotropic direction -- statistically the same radiation field as a point_collection source whose points
    are drawn uniformly inside the cells with the cells' luminosities."""
    pm, lum_map = map_source_problem()
    n = lum_map.shape[0]
    rng = np.random.default_rng(9)
    k = 40                                           # points per cell
    edges = np.linspace(-PC, PC, n + 1)
    iz, iy, ix = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    lo = np.stack([edges[ix], edges[iy], edges[iz]], axis=-1).reshape(-1, 1, 3)
    pts = (lo + rng.random((n ** 3, k, 3)) * (2 * PC / n)).reshape(-1, 3)
    lum = np.repeat(lum_map.ravel() / k, k)
    pc_ = Problem(walls=pm.walls, density=pm.density, dust=pm.dust, config=RunConfig(),
                  sources=[Source(type="point_collection", luminosity=pm.sources[0].luminosity, temperature=5000.0, points=pts,
                                  point_luminosity=lum * pm.sources[0].luminosity / lum.sum())])
    res = []
    for p in (pm, pc_):
        o = Oracle(p)
        se, st = o.lucy_iteration(400000, 1)
        o.close()
        assert st["killed_geo"] == 0
        res.append(se[0])
    a, b = res
    assert a.sum() == pytest.approx(b.sum(), rel=0.01)
    # (a cell's own emitters dominate its energy: 40 fixed points per cell against a fresh position per packet)
    np.testing.assert_allclose(a, b, rtol=0.12)
    assert np.median(np.abs(a / b - 1.0)) < 0.02
    # the field follows the map: the brightest cell of the map is the hottest one
    assert np.unravel_index(a.argmax(), a.shape) == np.unravel_index(lum_map.argmax(), lum_map.shape)


def test_lte_map_source_emits_the_dust_emissivity_of_its_cell():
    """spectrum 'lte' (source_type.f90:455-459, 486-491): in monochromatic mode and an optically thin cell the SED
    of an lte map source is L x (emission probability of the cell's dust at nu) -- the same spectral shape as the
    thermal packets of the same cell (emit_from_monochromatic_grid_pdf)."""
    from hyperion_amd.problem import PeeledImages
    C_CGS = 29979245800.0
    x = np.array([-1.0, 1.0])
    wav = np.logspace(0.5, 3.0, 8)
    cfg = RunConfig()
    cfg.monochromatic = True; cfg.frequencies = C_CGS / (wav * 1e-4); cfg.n_initial_iter = 0
    from hyperion_amd.benchmark import LSUN, load_test_dust
    peel = [PeeledImages(theta=[45.0], phi=[45.0], n_x=3, n_y=3, x_min=-2.0, x_max=2.0, y_min=-2.0, y_max=2.0,
                         n_ap=1, ap_min=10.0, ap_max=10.0, track_origin="basic", n_wav=len(wav))]
    p = Problem(walls=[x, x, x], density=np.full((1, 1, 1, 1), 1e-10), dust=[load_test_dust()], config=cfg, peeled=peel,
                sources=[Source(type="map", luminosity=LSUN, map=np.ones((1, 1, 1)), lte=True)],
                specific_energy=np.full((1, 1, 1, 1), 3e4))
    o = Oracle(p)
    res, st = o.mono_iteration(20000, 20000)
    o.close()
    sed = res[0]["sed"][0, :, 0, 0, :]              # (origin: source, dust, source scattered, dust scattered; frequency)
    src, dust = sed[0], sed[1]
    ok = (src > 0) & (dust > 0)
    assert ok.sum() >= 6
    ratio = src[ok] / dust[ok]
    assert ratio.std() / ratio.mean() < 1e-6          # same spectral shape, frequency by frequency
    assert src.max() > 0 and np.all(sed[2:] < 1e-6 * src.max())      # optically thin: nothing scattered


def spotted_star_problem(lum_spot=1.0):
    """hyperion/model/tests/test_spot_source.py: a sphere and a spot with disjoint emission bands, no dust."""
    from hyperion_amd.benchmark import load_test_dust
    from hyperion_amd.problem import PeeledImages, Spot
    x = np.array([-1e12, 1e12])
    nu = np.logspace(np.log10(3e12), np.log10(1e15), 300)
    fnu_sphere = np.where((nu > 1e13) & (nu < 2e13), 1.0, 0.0)      # ~15-30 micron
    fnu_spot = np.where((nu > 3e14) & (nu < 6e14), 1.0, 0.0)        # ~0.5-1 micron
    # the spot centre as the reference builds it: angle3d_deg(longitude, latitude) -> theta = longitude, phi = latitude
    spot = Spot(longitude=60.0, latitude=30.0, radius=25.0, luminosity=lum_spot, spectrum_nu=nu, spectrum_fnu=fnu_spot)
    src = Source(type="sphere", luminosity=1.0, position=(0.0, 0.0, 0.0), radius=1e11, spectrum_nu=nu, spectrum_fnu=fnu_sphere, spots=[spot])
    cfg = RunConfig()
    cfg.n_initial_iter = 0
    # views: onto the spot, and from the opposite side
    peel = [PeeledImages(theta=[60.0, 120.0], phi=[30.0, 210.0], n_wav=60, wav_min=0.1, wav_max=100.0, compute_image=False,
                         n_ap=1, ap_min=1e12, ap_max=1e12)]
    return Problem(walls=[x, x, x], density=np.zeros((1, 1, 1, 1)), dust=[load_test_dust()], sources=[src], config=cfg, peeled=peel)


def test_spot_uses_its_own_spectrum_and_is_seen_from_its_side_only():
    """The reference's regression test test_spot_source.py (spot photons carry the spot's spectrum, source_type.f90:
    447-461, 480-492) plus the geometry of emit_from_sphere(spot) :632-636: the spot is visible from the hemisphere
    it faces, with the flux of a Lambertian patch, and invisible from the opposite side."""
    p = spotted_star_problem()
    o = Oracle(p)
    res, st = o.final_iteration(200000)
    o.close()
    sed = res[0]["sed"][0, 0, :, 0, :]              # (view, wavelength bin)
    wav = np.logspace(np.log10(0.1), np.log10(100.0), 61)
    wav_c = np.sqrt(wav[1:] * wav[:-1])[::-1]       # bins are in frequency order: long wavelengths first
    sphere_band = (wav_c > 10.0) & (wav_c < 40.0)
    spot_band = (wav_c > 0.4) & (wav_c < 1.2)
    other = ~(sphere_band | spot_band)
    assert np.all(sed[:, other] == 0)
    f_sphere, f_spot = sed[:, sphere_band].sum(axis=1), sed[:, spot_band].sum(axis=1)
    # half of the packets come from the spot (equal luminosities); energies are scaled to L_tot = L_sphere = 1
    assert f_sphere[0] == pytest.approx(0.5, rel=0.03) and f_sphere[1] == pytest.approx(0.5, rel=0.03)
    # a Lambertian cap of half-angle a seen along its axis: 4 <mu> x (1/2) with <mu> = (1 + cos a) / 2
    ca = np.cos(np.radians(25.0))
    assert f_spot[0] == pytest.approx(0.5 * 4.0 * 0.5 * (1.0 + ca), rel=0.03)
    assert f_spot[1] == 0.0


def inside_observer_problem(tau=0.2):
    from hyperion_amd.problem import PeeledImages
    p = make_benchmark_problem(8, tau=tau)
    d = 0.6 * PC
    # observer on the -y axis; the viewing angle is the direction the photons at the map centre TRAVEL in (as for external
    # observers): (theta, phi) = (90, 270) = -y puts the central source at the map centre
    # (slightly off the x = 0 and z = 0 wall planes of the grid: a line of sight that ends exactly on a cell wall is a knife edge)
    p.peeled = [PeeledImages(theta=[90.0], phi=[270.0], inside_observer=True, peeloff_origin=(0.013 * PC, -d, 0.021 * PC), compute_sed=False,
                             n_x=9, n_y=5, x_min=180.0, x_max=-180.0, y_min=-90.0, y_max=90.0, n_wav=1, wav_min=0.01, wav_max=1e5)]
    return p, d


def test_inside_observer_sees_the_source_diluted_by_distance():
    """peeloff_photon for inside observers (images_peeled.f90:158-205, 236): peel-off towards the observer's position,
    optical depth integrated up to the observer only, 1 / (4 pi d^2) dilution, sky position in degrees of
    longitude / latitude around the viewing direction."""
    from hyperion_amd.benchmark import LSUN
    p, d = inside_observer_problem()
    p.config.n_initial_iter = 0
    o = Oracle(p)
    res, st = o.final_iteration(200000)
    o.close()
    img = res[0]["img"][0, 0, 0, :, :, 0]           # (y = latitude, x = longitude)
    iy, ix = np.unravel_index(img.argmax(), img.shape)
    assert (iy, ix) == (2, 4)                        # the source sits at the map centre
    # direct light: L / (4 pi d^2) exp(-tau_d) with tau_d = chi rho d; everything else is scattered / re-emitted light
    d = np.sqrt(d * d + (0.013 * PC) ** 2 + (0.021 * PC) ** 2)
    tau_d = 0.2 * d / PC
    direct = LSUN / (4 * np.pi * d * d) * np.exp(-tau_d)
    assert img[iy, ix] > direct and img[iy, ix] < 1.25 * direct
    rest = img.sum() - img[iy, ix]
    assert 0.0 < rest < 0.5 * direct
    # the observer looks the other way: the source is at longitude 180 = the edge pixels
    p.peeled[0].phi = np.array([90.0])
    o = Oracle(p)
    res2, _ = o.final_iteration(100000)
    o.close()
    img2 = res2[0]["img"][0, 0, 0, :, :, 0]
    assert img2[2, 0] + img2[2, 8] > 0.8 * direct and img2[2, 4] < 0.05 * direct
