"""GPU: cfm_transport_exact_f32 — exact OT between uniform marginals of different sizes without the lcm expansion.
Checked against an independent LP solver (scipy.optimize.linprog / HiGHS) on the same fp32 matrix: equal optimal cost,
exact marginals, a basic plan (fewer than B0 + B1 entries)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lp_cost(M):
    from scipy.optimize import linprog
    import scipy.sparse as sp
    B0, B1 = M.shape
    A = sp.vstack([sp.kron(sp.eye(B0), np.ones((1, B1))), sp.kron(np.ones((1, B0)), sp.eye(B1))]).tocsr()
    b = np.concatenate([np.full(B0, 1.0 / B0), np.full(B1, 1.0 / B1)])
    r = linprog(M.astype(np.float64).ravel(), A_eq=A, b_eq=b, bounds=(0, None), method="highs")
    assert r.status == 0
    return float(r.fun)


def _cloud_cost(B0, B1, d, seed, dev):
    import cfm_amd.optimal_transport as ot
    g = torch.Generator().manual_seed(seed)
    x0 = torch.randn(B0, d, generator=g); x1 = torch.randn(B1, d, generator=g) * 0.7 + 0.5
    return ot.cost_matrix(x0.to(dev), x1.to(dev))


@pytest.mark.parametrize("B0,B1,d", [(127, 128, 2), (128, 127, 2), (100, 60, 16), (37, 50, 3), (255, 256, 8), (300, 257, 4),
                                     (3, 2, 1), (1, 5, 2), (97, 211, 5)])
def test_transport_cost_equals_the_lp_optimum(B0, B1, d):
    import cfm_amd.optimal_transport as ot
    from cfm_amd import _lib
    dev = _lib.require_gpu()
    M = _cloud_cost(B0, B1, d, 100 * B0 + B1, dev)
    plan, cost = ot.transport_exact(M)
    P = plan.cpu().numpy(); Mh = M.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(P.sum(1), 1.0 / B0, rtol=0, atol=1e-15)
    np.testing.assert_allclose(P.sum(0), 1.0 / B1, rtol=0, atol=1e-15)
    assert (P >= 0).all() and (P > 0).sum() <= B0 + B1 - 1
    assert cost == pytest.approx(float((P * Mh).sum()), rel=1e-12)
    assert cost == pytest.approx(_lp_cost(M.cpu().numpy()), rel=1e-9, abs=1e-12)


def test_rectangular_exact_plan_beyond_the_lcm_bound_goes_through_the_transport_solver():
    """exact_plan_rect / OTPlanSampler('exact') / wasserstein for 127 vs 128 (lcm 16256 > 8192): no NotImplementedError
    any more; the plan is a transport plan with the LP's cost."""
    import cfm_amd.optimal_transport as ot
    from cfm_amd import _lib
    dev = _lib.require_gpu()
    g = torch.Generator().manual_seed(3)
    x0 = torch.randn(127, 2, generator=g); x1 = torch.randn(128, 2, generator=g) + 1.0
    M = ot.cost_matrix(x0.to(dev), x1.to(dev))
    plan, cost = ot.exact_plan_rect(M)
    ref = _lp_cost(M.cpu().numpy())
    assert cost == pytest.approx(ref, rel=1e-9)
    s = ot.OTPlanSampler(method="exact")
    pi = s.get_map(x0, x1)
    assert pi.shape == (127, 128) and np.allclose(pi.sum(1), 1 / 127) and np.allclose(pi.sum(0), 1 / 128)
    w2 = ot.wasserstein(x0, x1, method="exact", power=2)
    assert w2 == pytest.approx(np.sqrt(ref), rel=1e-6)


def test_transport_with_ties_and_sizes_it_refuses():
    import cfm_amd.optimal_transport as ot
    from cfm_amd import _lib
    dev = _lib.require_gpu()
    r = np.random.RandomState(0)
    M = torch.from_numpy(r.randint(0, 3, (45, 64)).astype(np.float32)).to(dev)       # massively tied
    plan, cost = ot.transport_exact(M)
    assert cost == pytest.approx(_lp_cost(M.cpu().numpy()), rel=1e-9, abs=1e-12)
    big = torch.rand(1500, 1501, device=dev)
    with pytest.raises(NotImplementedError):
        ot.exact_plan_rect(big)
