"""GPU: OTPlanSampler(method="unbalanced" | "partial") — the kernel-space fp64 solvers behind
cfm_unbalanced_sinkhorn_f64 / cfm_partial_entropic_f64 against the committed golden plans
(recorded from the reference wrapper over the restated POT loops) and against the oracle.

Tolerance: plans are float64 sums of ~B terms in a different summation order than NumPy's
dot -> 1e-9 relative to the largest entry (the north-star bound for potentials is 1e-5)."""
import os
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RTOL = 1e-9        # same fp32 cost matrix on both sides: summation order only
RTOL_E2E = 2e-5    # golden plans were recorded from torch.cdist(...)**2 on the CPU: the two fp32 cost
                   # matrices differ in the last bit and K = exp(-M/reg) amplifies that by M/reg


def _close(p, ref, rtol=RTOL):
    scale = np.abs(ref).max()
    return np.abs(p - ref).max() <= rtol * scale


@pytest.fixture(scope="module")
def dev():
    from cfm_amd import _lib
    _lib.load()
    return _lib.require_gpu()


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "ub_cases.npz"))


@pytest.mark.parametrize("reg,reg_m", [(0.5, 1.0), (1.0, 0.2), (0.3, 5.0)])
def test_unbalanced_get_map_matches_golden(gold, reg, reg_m):
    from cfm_amd.optimal_transport import OTPlanSampler
    x0, x1 = torch.from_numpy(gold["x0"]), torch.from_numpy(gold["x1"])
    s = OTPlanSampler(method="unbalanced", reg=reg, reg_m=reg_m)
    pi = s.get_map(x0, x1)
    ref = gold[f"unb_pi_{reg}_{reg_m}"]
    assert pi.dtype == np.float64 and pi.shape == ref.shape
    assert _close(pi, ref, RTOL_E2E), np.abs(pi - ref).max() / np.abs(ref).max()
    # the device cdf search on the recorded plan reproduces the recorded draws bit for bit
    np.random.seed(3)
    i0, j0 = s.sample_map(ref, x0.shape[0])
    assert np.array_equal(i0, gold[f"unb_i_{reg}_{reg_m}"]) and np.array_equal(j0, gold[f"unb_j_{reg}_{reg_m}"])


@pytest.mark.parametrize("reg", [0.5, 2.0])
def test_partial_get_map_matches_golden(gold, reg):
    from cfm_amd.optimal_transport import OTPlanSampler
    x0, x1 = torch.from_numpy(gold["x0"]), torch.from_numpy(gold["x1"])
    s = OTPlanSampler(method="partial", reg=reg)
    pi = s.get_map(x0, x1)
    ref = gold[f"par_pi_{reg}"]
    assert _close(pi, ref, RTOL_E2E), np.abs(pi - ref).max() / np.abs(ref).max()
    assert abs(pi.sum() - 1.0) < 1e-12
    np.random.seed(4)
    i0, j0 = s.sample_map(ref, x0.shape[0])
    assert np.array_equal(i0, gold[f"par_i_{reg}"]) and np.array_equal(j0, gold[f"par_j_{reg}"])


def test_rectangular_plans(gold):
    from cfm_amd.optimal_transport import OTPlanSampler
    x0, x2 = torch.from_numpy(gold["x0"]), torch.from_numpy(gold["x2"])
    pu = OTPlanSampler(method="unbalanced", reg=0.7, reg_m=1.0).get_map(x0, x2)
    pp = OTPlanSampler(method="partial", reg=0.7).get_map(x0, x2)
    assert pu.shape == (96, 64) and _close(pu, gold["unb_rect"], RTOL_E2E)
    assert pp.shape == (96, 64) and _close(pp, gold["par_rect"], RTOL_E2E)


def test_partial_mass_below_one(gold, dev):
    import cfm_amd.optimal_transport as ot
    M = ot.cost_matrix(torch.from_numpy(gold["x0"]).to(dev), torch.from_numpy(gold["x1"]).to(dev))
    import cfm_oracle as oracle
    plan, info = ot.partial_plan(M, 0.5, m=0.6)
    p = plan.cpu().numpy()
    assert _close(p, gold["par_m06"], RTOL_E2E)
    assert _close(p, oracle.entropic_partial_wasserstein(M.cpu().numpy(), 0.5, m=0.6))     # same M: tight
    assert abs(p.sum() - 0.6) < 1e-12
    assert (p.sum(1) <= 1.0 / 96 + 1e-12).all() and (p.sum(0) <= 1.0 / 96 + 1e-12).all()
    it, status = info.tolist()[:2]
    # POT's stopThr = 1e-100 only fires on an exactly stationary iterate; the vector form reaches
    # that (err == 0) where the matrix form keeps a 1e-18 jitter: any count up to numItermax is fine
    assert status == 0 and 0 < it <= 1000


@pytest.mark.parametrize("reg,reg_m", [(0.5, 1.0), (1.0, 0.2), (0.3, 5.0)])
def test_unbalanced_same_cost_matrix_tight(gold, dev, reg, reg_m):
    import cfm_oracle as oracle
    import cfm_amd.optimal_transport as ot
    M = ot.cost_matrix(torch.from_numpy(gold["x0"]).to(dev), torch.from_numpy(gold["x1"]).to(dev))
    plan, info = ot.unbalanced_plan(M, reg, reg_m)
    ref, log = oracle.sinkhorn_knopp_unbalanced(M.cpu().numpy(), reg, reg_m, log=True)
    assert _close(plan.cpu().numpy(), ref)
    assert info.tolist()[:2] == [log["iters"], log["status"]]


def test_unbalanced_docstring_kat(dev):
    """runner/src/models/components/sinkhorn_knopp_unbalanced.py:88-94 with uniform marginals:
    a = b = [.5, .5] is exactly what get_map builds for B = 2; the cost matrix goes in directly."""
    import cfm_amd.optimal_transport as ot
    M = torch.tensor([[0.0, 1.0], [1.0, 0.0]], device=dev)
    plan, info = ot.unbalanced_plan(M, 1.0, 1.0)
    p = plan.cpu().numpy()
    assert np.abs(p - np.array([[0.51122814, 0.18807032], [0.18807032, 0.51122814]])).max() < 5e-7


@pytest.mark.parametrize("method", ["unbalanced", "partial"])
def test_early_iterations_match_oracle_fullsize(dev, method):
    """B = 4096 (config C2 shape), a fixed number of iterations of the exact loop."""
    import cfm_oracle as oracle
    import cfm_amd.optimal_transport as ot
    x0, x1 = oracle.config_inputs("C2")
    M = ot.cost_matrix(x0.to(dev), x1.to(dev))
    Mnp = M.cpu().numpy()
    if method == "unbalanced":
        plan, info = ot.unbalanced_plan(M, 40.0, 1.0, max_iter=12)
        ref, log = oracle.sinkhorn_knopp_unbalanced(Mnp, 40.0, 1.0, numItermax=12, log=True)
    else:
        plan, info = ot.partial_plan(M, 40.0, 1.0, max_iter=12)
        ref, log = oracle.entropic_partial_wasserstein(Mnp, 40.0, numItermax=12, log=True)
    assert _close(plan.cpu().numpy(), ref)
    assert info.tolist()[0] == log["iters"]


def test_unbalanced_underflow_reverts_and_falls_back(dev):
    """K = exp(-M/reg) underflows row-wise -> K^T u == 0 / inf at iteration 0 -> POT keeps the
    initial (u, v) = 1/n: plan = K / n^2 ~ 0 -> get_map's uniform fallback (ref:93-96)."""
    from cfm_amd.optimal_transport import OTPlanSampler
    g = torch.Generator().manual_seed(0)
    x0 = torch.randn(32, 2, generator=g) * 40
    x1 = torch.randn(32, 2, generator=g) * 40 + 300
    s = OTPlanSampler(method="unbalanced", reg=0.05)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        pi = s.get_map(x0, x1)
    assert np.allclose(pi, 1.0 / pi.size)
    assert any("reverting to uniform plan" in str(x.message) for x in w)
    np.random.seed(0)
    a, b = s.sample_plan(x0, x1)
    assert a.shape == x0.shape and b.shape == x1.shape


def test_partial_zeros_in_kernel_poison_like_pot(dev, capsys):
    from cfm_amd.optimal_transport import OTPlanSampler
    g = torch.Generator().manual_seed(0)
    x0 = torch.randn(32, 2, generator=g) * 10
    x1 = torch.randn(32, 2, generator=g) * 10 + 30
    s = OTPlanSampler(method="partial", reg=0.05)
    pi = s.get_map(x0, x1)
    assert np.isnan(pi).all()
    assert "ERROR: p is not finite" in capsys.readouterr().out
    with pytest.raises(ValueError):
        s.sample_plan(x0, x1)


@pytest.mark.parametrize("method", ["unbalanced", "partial"])
def test_sample_plan_end_to_end(dev, method):
    """sample_plan == get_map + sample_map + gather with the same RNG stream (ref:123-145)."""
    from cfm_amd.optimal_transport import OTPlanSampler
    torch.manual_seed(5)
    x0 = torch.randn(128, 2, 2, 2)
    x1 = torch.randn(128, 2, 2, 2)
    s = OTPlanSampler(method=method, reg=1.0)
    pi = s.get_map(x0, x1)
    np.random.seed(9)
    i, j = s.sample_map(pi, 128)
    np.random.seed(9)
    a, b = s.sample_plan(x0, x1)
    assert torch.equal(a, x0[i]) and torch.equal(b, x1[j])
