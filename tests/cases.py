"""Shared problem builders for the tests (small cases of BASELINE.json's configs
and the edge cases the reference's own tests exercise)."""
import os

import numpy as np

from hyperion_amd.benchmark import LSUN, PC, load_test_dust, make_benchmark_problem
from hyperion_amd.problem import Problem, RunConfig, Source

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_problem(name):
    path = os.path.join(GOLDEN, name)
    return Problem.from_npz(path), np.load(path, allow_pickle=False)


def ragged_grid_problem(scale=1.0, seed=3, n_sources=2, vertex_source=True):
    """Irregular (ragged) wall spacing, random density with some empty cells,
    sources on cell vertices / edges (hyperion/model/tests/test_propagation.py:24-135)
    and an arbitrary overall length scale (:138-149 uses 1e-20 .. 1e20)."""
    rng = np.random.RandomState(seed)
    def walls(n):
        w = np.sort(rng.uniform(-1.0, 1.0, n - 1))
        return np.concatenate([[-1.0], w, [1.0]]) * scale
    w = [walls(6), walls(5), walls(7)]
    n1, n2, n3 = (a.size - 1 for a in w)
    rho = rng.uniform(0.2, 3.0, (1, n3, n2, n1)) / scale
    rho[0][rng.uniform(size=(n3, n2, n1)) < 0.15] = 0.0      # empty cells
    srcs = []
    for i in range(n_sources):
        if vertex_source and i == 0:
            pos = (w[0][2], w[1][2], w[2][3])                 # exactly on a vertex
        elif vertex_source and i == 1:
            pos = (w[0][3], 0.123 * scale, w[2][1])           # on an edge
        else:
            pos = tuple(rng.uniform(-0.9, 0.9, 3) * scale)
        srcs.append(Source(type="point", luminosity=LSUN * (i + 1), position=pos, temperature=3000.0 + 2000.0 * i))
    return Problem(walls=w, density=rho, dust=[load_test_dust()], sources=srcs, config=RunConfig())


def spectrum_source_problem():
    p = make_benchmark_problem(12, tau=2.0)
    nu = np.logspace(12.0, 15.5, 40)
    fnu = nu ** -0.7 * np.exp(-nu / 2e15)
    p.sources = [Source(type="point", luminosity=LSUN, position=(0.1 * PC, -0.2 * PC, 0.3 * PC),
                        spectrum_nu=nu, spectrum_fnu=fnu)]
    return p


def assert_parity(a, b, rtol=1e-9, atol_rel=1e-12):
    """GPU vs oracle on identical Philox streams.  rtol covers FP64 atomic
    summation order and 1-ulp libm differences; the absolute term (atol_rel x the
    largest cell value) covers cells whose whole content is one cancellation-
    dominated partial step."""
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol_rel * np.abs(b).max())


def imaging_problem(n=12, tau=1.0, n_x=16, n_y=16, **peeled_kw):
    """Small version of BASELINE config 4's imaging set-up on a Cartesian grid:
    central source, one peeled group, one view (45, 45), Stokes on."""
    from hyperion_amd.problem import PeeledImages
    p = make_benchmark_problem(n, tau=tau)
    kw = dict(theta=[45.0], phi=[45.0], n_wav=3, wav_min=0.1, wav_max=1000.0,
              n_x=n_x, n_y=n_y, x_min=-1.5 * PC, x_max=1.5 * PC, y_min=-1.5 * PC, y_max=1.5 * PC,
              n_ap=3, ap_min=0.2 * PC, ap_max=2.0 * PC, compute_stokes=True)
    kw.update(peeled_kw)
    p.peeled = [PeeledImages(**kw)]
    return p


def pda_block_problem(n=14, tau_cell=100.0, n_photons=30000, pda=True, grid="car"):
    """A thin medium with an opaque block in it, lit from the side: packets are absorbed and re-emitted in the skin of
    the block and hardly ever reach its inner cells -- the situation the partial diffusion approximation is for
    (src/grid/grid_pda_3d.f90: cells with fewer than max(30, 0.5 % of the mean) packets)."""
    prob = make_benchmark_problem(n, n_photons=n_photons, n_iter=1)
    w = prob.walls[0]
    dx = w[1] - w[0]
    rho = np.full(prob.density.shape, 0.05 / (w[-1] - w[0]))
    lo, hi = n // 2 - 3, n // 2 + 3
    rho[0, lo:hi, lo:hi, lo:hi] = tau_cell / dx
    prob.density = rho
    prob.sources[0].position = (0.8 * w[0] + 0.0, 0.13 * dx, -0.21 * dx)
    prob.config.pda = pda
    prob.config.output_n_photons = "last"
    return prob


def voronoi_big_problem(n_photons=10_000_000, n_iter=1, two_species=True):
    """BASELINE configs[4] on a REAL tessellation: the voro++ cells of 100 000 random sites in a 2 pc box
    (tests/golden/vor_big.npz, computed by the reference front-end; ~15.5 neighbours per cell), two anisotropically
    scattering polarising dust species, a point source and an external box source."""
    from hyperion_amd.benchmark import make_voronoi_lattice_problem
    z = np.load(os.path.join(GOLDEN, "vor_big.npz"), allow_pickle=False)
    tpl = make_voronoi_lattice_problem(n=2, n_photons=n_photons, n_iter=n_iter, two_species=two_species)     # dust, sources, config
    n = z["sites"].shape[0]
    tau = 1.0
    rho = np.full((1, n), tau / PC)
    if two_species:
        rho = np.vstack([rho * 0.6, rho * 0.8])
    return Problem(walls=[], density=rho, dust=tpl.dust, sources=tpl.sources, config=tpl.config, grid_type="vor",
                   vor_sites=z["sites"], vor_volume=z["volume"], vor_idx=z["idx"], vor_neighs=z["neighs"],
                   vor_box=tuple(z["box"]), vor_bb=np.concatenate([z["bb_lo"].astype(np.float64), z["bb_hi"].astype(np.float64)], axis=1))
