"""The HIP path against the reference's own golden outputs (statistical, the
RNG streams differ): hyperion/model/tests/data/test_specific_energy.grid_type=car.*.rtout."""
import numpy as np
import pytest

import hyperion_amd
from cases import golden_problem

pytestmark = pytest.mark.gpu
K = 16


@pytest.mark.parametrize("name", ["False.False", "True.False", "False.True", "True.True"])
def test_all_five_iterations_match_reference_golden(name):
    prob, z = golden_problem("car_specific_energy.%s.npz" % name)
    gold = z["golden/specific_energy"]                     # (5, n_dust, 3, 5, 7)
    chains = []
    for s in range(K):
        prob.config.seed = -(7000 + s)
        eng = hyperion_amd.Engine(prob)
        chains.append([eng.lucy_iteration(10000, it)[0] for it in range(1, 6)])
        eng.close()
    chains = np.array(chains)                              # (K, 5, ...)
    prob.config.seed = -424242
    eng = hyperion_amd.Engine(prob)
    big = np.array([eng.lucy_iteration(1000000, it)[0] for it in range(1, 6)])
    eng.close()
    sigma = chains.std(axis=0, ddof=1)
    zs = (gold - big) / sigma
    assert np.abs(zs).max() < 6.5
    assert abs(zs.mean()) < 0.25
    assert 0.7 < (zs ** 2).mean() < 1.6
    w = prob.density * prob.volumes
    for it in range(5):
        assert (gold[it] * w).sum() == pytest.approx((big[it] * w).sum(), rel=0.04)
    assert (gold * w).sum() == pytest.approx((big * w).sum(), rel=0.02)
