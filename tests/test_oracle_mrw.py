"""Modified random walk (src/grid/grid_mrw_3d.f90): the oracle against the reference's own
known-answer table (hyperion/model/tests/test_mrw.py:10-31 -- equilibrium temperature of a
single-cell grid for 18 densities, 10 % tolerance; same set-up: L=1, T=6000 K point source,
realistic dust, 1000 packets, <=30 iterations with convergence 99 % / 2 / 1.02, gamma=2,
n_inter_max=1e9)."""
import os

import numpy as np
import pytest

from cases import GOLDEN
from hyperion_amd.problem import Dust, PeeledImages, Problem, RunConfig, Source
from hyperion_amd.run import ConvergenceCheck
from oracle_lib import Oracle, OracleError

D_REF = np.logspace(-5.0, 12.0, 18)
T_REF = [24.75280, 24.66414, 24.52175, 21.97109, 15.53059, 10.76363, 7.810127, 6.672520, 6.902798,
         16.64318, 53.08394, 162.0158, 438.9026, 1013.141, 2156.520, 4642.825, 9948.065, 21211.08]

_KEYS = ("nu", "albedo", "chi", "mu", "P1", "P2", "P3", "P4", "emiss_nu", "emiss_jnu", "emiss_var",
         "mo_specific_energy", "mo_chi_rosseland", "mo_kappa_planck", "mo_chi_inv_planck", "mo_temperature")


def realistic_dust(**drop):
    """get_realistic_test_dust() of the reference's test helpers, tabulated by
    tests/golden/make_fixtures.py mrw."""
    z = np.load(os.path.join(GOLDEN, "realistic_dust.npz"), allow_pickle=False)
    return Dust(version=int(z["version"]), **{k: z[k] for k in _KEYS if k not in drop})


def temperature(dust, e):
    """specific_energy2temperature: hyperion/dust/dust_type.py:479-510 (log-log interpolation)."""
    return 10.0 ** np.interp(np.log10(e), np.log10(dust.mo_specific_energy), np.log10(dust.mo_temperature))


def single_cell_problem(densities, **cfg):
    w = np.array([-1.0, 1.0])
    kw = dict(n_inter_max=1000000000, mrw=True, mrw_gamma=2.0)
    kw.update(cfg)
    rho = np.asarray(densities, dtype=np.float64).reshape(-1, 1, 1, 1)
    return Problem(walls=[w, w, w], density=rho, dust=[realistic_dust() for _ in range(rho.shape[0])],
                   sources=[Source(type="point", luminosity=1.0, position=(0.0, 0.0, 0.0), temperature=6000.0)],
                   config=RunConfig(**kw))


def converge(engine, n_packets=1000, n_iter=30):
    chk = ConvergenceCheck(2.0, 1.02, 99.0)
    for it in range(1, n_iter + 1):
        e, st = engine.lucy_iteration(n_packets, it)
        if chk(engine):
            break
    return e, st


@pytest.mark.parametrize("i", range(18))
def test_single_temperature(i):
    p = single_cell_problem([D_REF[i]])
    o = Oracle(p)
    e, _ = converge(o)
    o.close()
    t = temperature(p.dust[0], e[0, 0, 0, 0])
    assert T_REF[i] / t < 1.1 and t / T_REF[i] < 1.1


@pytest.mark.parametrize("i", range(0, 18, 3))
def test_multi_temperature(i):
    """test_mrw.py:66-102: the same mass split over four identical dust populations."""
    p = single_cell_problem(D_REF[i] * np.array([0.1, 0.2, 0.3, 0.4]))
    o = Oracle(p)
    e, _ = converge(o)
    o.close()
    for d in range(4):
        t = temperature(p.dust[d], e[d, 0, 0, 0])
        assert T_REF[i] / t < 1.1 and t / T_REF[i] < 1.1


def test_mrw_step_limit_kills():
    """iter_lucy.f90:146-151: a packet still in the diffusion regime after n_inter_mrw_max
    steps is killed."""
    p = single_cell_problem([1e8], n_inter_mrw_max=1)
    o = Oracle(p)
    _, st = o.lucy_iteration(200, 1)
    o.close()
    assert st["killed_int"] > 0


def test_mrw_needs_mean_opacities():
    p = single_cell_problem([1.0])
    p.dust = [realistic_dust(mo_kappa_planck=1, mo_chi_inv_planck=1)]
    o = Oracle(p)
    with pytest.raises(OracleError, match="kappa_planck"):
        o.lucy_iteration(10, 1)
    o.close()


def test_mrw_imaging_runs_and_conserves_energy():
    """iter_final.f90:165-183: MRW steps of the imaging iteration deposit no energy and peel
    off as isotropic emission; the SED of a spherically symmetric set-up summed over
    frequency stays close to L/(4 pi d^2) x 4 pi d^2 = 1 per unit luminosity."""
    p = single_cell_problem([1e4])
    p.peeled = [PeeledImages(theta=[30.0], phi=[60.0], n_wav=40, wav_min=0.01, wav_max=5000.0, n_x=1, n_y=1,
                             x_min=-2.0, x_max=2.0, y_min=-2.0, y_max=2.0, n_ap=1, ap_min=3.0, ap_max=3.0,
                             compute_image=False, compute_stokes=False)]
    o = Oracle(p)
    converge(o)
    cubes, st = o.final_iteration(4000)
    o.close()
    total = cubes[0]["sed"].sum()
    assert st["killed_int"] == 0
    assert 0.8 < total < 1.2
