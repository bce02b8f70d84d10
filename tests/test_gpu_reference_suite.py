"""GPU: the reference's OWN tests (tests/test_conditional_flow_matcher.py, tests/test_time_t.py,
tests/test_optimal_transport.py), transliterated against cfm_amd.  Inputs are CPU tensors exactly
as in the reference; the package runs them through the HIP path and hands CPU tensors back.
`ot.emd/emd2` of the originals is played by the oracle (SciPy LSAP)."""
import math

import numpy as np
import pytest
import torch

import cfm_oracle as oracle
from cfm_amd.conditional_flow_matching import (
    ConditionalFlowMatcher,
    ExactOptimalTransportConditionalFlowMatcher,
    SchrodingerBridgeConditionalFlowMatcher,
    TargetConditionalFlowMatcher,
    VariancePreservingConditionalFlowMatcher,
    pad_t_like_x,
)
from cfm_amd.optimal_transport import OTPlanSampler, wasserstein

pytestmark = pytest.mark.gpu

TEST_SEED = 1994
TEST_BATCH_SIZE = 128
SIGMA_CONDITION = {"sb_cfm": lambda x: x <= 0}


def random_samples(shape, batch_size=TEST_BATCH_SIZE):
    dims = [shape] if isinstance(shape, int) else list(shape)
    return [torch.randn(batch_size, *dims) for _ in range(2)]


def _interp(x0, x1, t):
    return t * x1 + (1 - t) * x0


# The closed forms the reference's test asserts against (tests/test_conditional_flow_matcher.py:35-68),
# one entry per probability path: method -> (mu_t, sigma_t, u_t).  The operation order inside every
# expression is the reference's (the comparison below is bitwise).
_HALF_PI = math.pi / 2
_PATHS = {
    "vp_cfm": (
        lambda x0, x1, t, s: torch.cos(_HALF_PI * t) * x0 + torch.sin(_HALF_PI * t) * x1,
        lambda t, s: s,
        lambda x0, x1, t, s, xt, sig_t: _HALF_PI * (torch.cos(_HALF_PI * t) * x1 - torch.sin(_HALF_PI * t) * x0),
    ),
    "t_cfm": (
        lambda x0, x1, t, s: t * x1,
        lambda t, s: 1 - (1 - s) * t,
        lambda x0, x1, t, s, xt, sig_t: (x1 - (1 - s) * xt) / sig_t,
    ),
    "sb_cfm": (
        lambda x0, x1, t, s: _interp(x0, x1, t),
        lambda t, s: s * torch.sqrt(t * (1 - t)),
        lambda x0, x1, t, s, xt, sig_t: (1 - 2 * t) / (2 * t * (1 - t) + 1e-8) * (xt - _interp(x0, x1, t)) + x1 - x0,
    ),
    "i_cfm": (
        lambda x0, x1, t, s: _interp(x0, x1, t),
        lambda t, s: s,
        lambda x0, x1, t, s, xt, sig_t: x1 - x0,
    ),
}
_PATHS["exact_ot_cfm"] = _PATHS["i_cfm"]

_MATCHERS = {
    "vp_cfm": lambda sigma: VariancePreservingConditionalFlowMatcher(sigma=sigma),
    "t_cfm": lambda sigma: TargetConditionalFlowMatcher(sigma=sigma),
    "sb_cfm": lambda sigma: SchrodingerBridgeConditionalFlowMatcher(sigma=sigma, ot_method="sinkhorn"),
    "exact_ot_cfm": lambda sigma: ExactOptimalTransportConditionalFlowMatcher(sigma=sigma),
    "i_cfm": lambda sigma: ConditionalFlowMatcher(sigma=sigma),
}


def compute_xt_ut(method, x0, x1, t_given, sigma, epsilon):
    mu, sig, flow = _PATHS[method]
    sigma_t = sig(t_given, sigma)
    computed_xt = mu(x0, x1, t_given, sigma) + sigma_t * epsilon
    return computed_xt, flow(x0, x1, t_given, sigma, computed_xt, sigma_t)


def get_flow_matcher(method, sigma):
    return _MATCHERS[method](sigma)


def sample_plan(method, x0, x1, sigma):
    if method == "sb_cfm":
        x0, x1 = OTPlanSampler(method="sinkhorn", reg=2 * (sigma**2)).sample_plan(x0, x1)
    elif method == "exact_ot_cfm":
        x0, x1 = OTPlanSampler(method="exact").sample_plan(x0, x1)
    return x0, x1


@pytest.mark.parametrize("method", ["vp_cfm", "t_cfm", "sb_cfm", "exact_ot_cfm", "i_cfm"])
@pytest.mark.parametrize("sigma", [0.0, 5e-4, 0.5, 1.5, 0, 1])
@pytest.mark.parametrize("shape", [[1], [2], [1, 2], [3, 4, 5]])
def test_fm(method, sigma, shape):
    batch_size = TEST_BATCH_SIZE
    if method in SIGMA_CONDITION.keys() and SIGMA_CONDITION[method](sigma):
        with pytest.raises(ValueError):
            get_flow_matcher(method, sigma)
        return
    FM = get_flow_matcher(method, sigma)
    x0, x1 = random_samples(shape, batch_size=batch_size)
    torch.manual_seed(TEST_SEED)
    np.random.seed(TEST_SEED)
    t, xt, ut, eps = FM.sample_location_and_conditional_flow(x0, x1, return_noise=True)
    _ = FM.compute_lambda(t)
    if method in ["sb_cfm", "exact_ot_cfm"]:
        torch.manual_seed(TEST_SEED)
        np.random.seed(TEST_SEED)
        x0, x1 = sample_plan(method, x0, x1, sigma)
    torch.manual_seed(TEST_SEED)
    t_given_init = torch.rand(batch_size)
    t_given = t_given_init.reshape(-1, *([1] * (x0.dim() - 1)))
    sigma_pad = pad_t_like_x(sigma, x0)
    epsilon = torch.randn_like(x0)
    computed_xt, computed_ut = compute_xt_ut(method, x0, x1, t_given, sigma_pad, epsilon)
    assert torch.all(ut.eq(computed_ut))
    assert torch.all(xt.eq(computed_xt))
    assert torch.all(eps.eq(epsilon))
    assert any(t_given_init == t)


@pytest.mark.parametrize("FM", [
    ConditionalFlowMatcher(sigma=0.0),
    ExactOptimalTransportConditionalFlowMatcher(sigma=0.0),
    TargetConditionalFlowMatcher(sigma=0.0),
    SchrodingerBridgeConditionalFlowMatcher(sigma=0.1),
    VariancePreservingConditionalFlowMatcher(sigma=0.0),
])
def test_random_Tensor_t(FM):
    x0 = torch.randn(128, 2)
    x1 = torch.randn(128, 2)
    torch.manual_seed(1994)
    t_given = torch.rand(128)
    t_given, xt, ut = FM.sample_location_and_conditional_flow(x0, x1, t=t_given)
    torch.manual_seed(1994)
    t_random, xt, ut = FM.sample_location_and_conditional_flow(x0, x1, t=None)
    assert any(t_given == t_random)


@pytest.mark.parametrize("FM", [
    ExactOptimalTransportConditionalFlowMatcher(sigma=0.0),
    SchrodingerBridgeConditionalFlowMatcher(sigma=0.1),
])
@pytest.mark.parametrize("return_noise", [True, False])
def test_guided_random_Tensor_t(FM, return_noise):
    x0 = torch.randn(128, 2)
    y0 = torch.randint(high=10, size=(128, 1))
    x1 = torch.randn(128, 2)
    y1 = torch.randint(high=10, size=(128, 1))
    torch.manual_seed(1994)
    t_given = torch.rand(128)
    out = FM.guided_sample_location_and_conditional_flow(x0, x1, y0=y0, y1=y1, t=t_given, return_noise=return_noise)
    assert len(out) == (6 if return_noise else 5) and out[3].shape == y0.shape
    t_given = out[0]
    torch.manual_seed(1994)
    t_random = FM.guided_sample_location_and_conditional_flow(x0, x1, y0=y0, y1=y1, t=None, return_noise=return_noise)[0]
    assert any(t_given == t_random)


ot_sampler = OTPlanSampler(method="exact")


def test_sample_map(batch_size=128):
    map = np.eye(batch_size)
    rng = np.random.default_rng()
    permuted_map = rng.permutation(map, axis=1)
    indices = ot_sampler.sample_map(permuted_map, batch_size=batch_size, replace=False)
    reconstructed_map = np.zeros((batch_size, batch_size))
    for i in range(batch_size):
        reconstructed_map[indices[0][i], indices[1][i]] = 1
    assert np.array_equal(reconstructed_map, permuted_map)


def test_get_map(batch_size=128):
    x0 = torch.randn(batch_size, 2, 2, 2)
    x1 = torch.randn(batch_size, 2, 2, 2)
    M = torch.cdist(x0.reshape(x0.shape[0], -1), x1.reshape(x1.shape[0], -1)) ** 2
    pot_pi = oracle.perm_plan(oracle.exact_perm(M.numpy()))     # = ot.emd(unif, unif, M)
    pi = ot_sampler.get_map(x0, x1)
    assert np.array_equal(pi, pot_pi)


def test_sample_plan(batch_size=128, seed=1980):
    torch.manual_seed(seed)
    np.random.seed(seed)
    x0 = torch.randn(batch_size, 2, 2, 2)
    x1 = torch.randn(batch_size, 2, 2, 2)
    pi = ot_sampler.get_map(x0, x1)
    indices_i, indices_j = ot_sampler.sample_map(pi, batch_size=batch_size, replace=True)
    new_x0, new_x1 = x0[indices_i], x1[indices_j]
    torch.manual_seed(seed)
    np.random.seed(seed)
    sampled_x0, sampled_x1 = ot_sampler.sample_plan(x0, x1, replace=True)
    assert torch.equal(new_x0, sampled_x0)
    assert torch.equal(new_x1, sampled_x1)


def test_wasserstein(batch_size=128, seed=1980):
    torch.manual_seed(seed)
    np.random.seed(seed)
    x0 = torch.randn(batch_size, 2, 2, 2)
    x1 = torch.randn(batch_size, 2, 2, 2)
    M = torch.cdist(x0.reshape(x0.shape[0], -1), x1.reshape(x1.shape[0], -1))
    M2 = (M**2).numpy()
    pot_W2 = np.sqrt(oracle.assignment_cost(M2, oracle.exact_perm(M2)) / batch_size)
    W2 = wasserstein(x0, x1, "exact")
    M1 = M.numpy()
    pot_W1 = oracle.assignment_cost(M1, oracle.exact_perm(M1)) / batch_size
    W1 = wasserstein(x0, x1, "exact", power=1)
    # the reference asserts equality to POT on POT's own fp32 matrix; our cost kernel is the
    # (more accurate) direct form, so the values agree to fp32 cost rounding, not bitwise
    assert W2 == pytest.approx(pot_W2, rel=2e-6)
    assert W1 == pytest.approx(pot_W1, rel=2e-6)
    uo, vo, _, _ = oracle.sinkhorn_log(M1, 0.1, numItermax=20000)
    pot_eot = float((oracle.sinkhorn_plan(M1, 0.1, uo, vo) * M1.astype(np.float64)).sum())
    eot = wasserstein(x0, x1, "sinkhorn", reg=0.1, power=1)
    assert eot == pytest.approx(pot_eot, rel=1e-4)
    with pytest.raises(ValueError):
        wasserstein(x0, x1, "noname", reg=0.01, power=1)
