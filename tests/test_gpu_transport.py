"""GPU: cfm_transport_exact_f32 — exact OT between uniform marginals of different sizes without the lcm expansion
(round 6: primal-dual phases with a tree push, optional assignment warm start).  Checked against an independent LP
solver (scipy.optimize.linprog / HiGHS) on the same fp32 matrix: equal optimal cost, exact marginals; cold start and
warm start must agree; sizes up to B0 + B1 = 2048."""
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


@pytest.mark.parametrize("warm", [False, True])
@pytest.mark.parametrize("B0,B1,d", [(127, 128, 2), (128, 127, 2), (100, 60, 16), (37, 50, 3), (255, 256, 8), (300, 257, 4),
                                     (3, 2, 1), (1, 5, 2), (97, 211, 5), (125, 128, 2), (2, 2, 1), (64, 64, 3)])
def test_transport_cost_equals_the_lp_optimum(B0, B1, d, warm):
    import cfm_amd.optimal_transport as ot
    from cfm_amd import _lib
    dev = _lib.require_gpu()
    M = _cloud_cost(B0, B1, d, 100 * B0 + B1, dev)
    plan, cost, info = ot.transport_exact(M, warm_start=warm, return_info=True)
    P = plan.cpu().numpy(); Mh = M.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(P.sum(1), 1.0 / B0, rtol=0, atol=1e-15)
    np.testing.assert_allclose(P.sum(0), 1.0 / B1, rtol=0, atol=1e-15)
    assert (P >= 0).all()
    assert cost == pytest.approx(float((P * Mh).sum()), rel=1e-12)
    assert cost == pytest.approx(_lp_cost(M.cpu().numpy()), rel=1e-9, abs=1e-12)
    if warm and B0 != B1:
        assert info["warm_start_used"]
        if abs(B0 - B1) == 1:
            assert info["phases"] == 1            # one forest into the one open column carries every row's last unit


def test_transport_is_deterministic_and_staged_equals_global():
    """Same input, same plan (the pushes are integer sums, the returns go in row order); and the LDS-staged form
    (127 x 128 fits) and the global-memory form of the same kernel (140 x 163 does not fit), each against the LP."""
    import cfm_amd.optimal_transport as ot
    from cfm_amd import _lib
    dev = _lib.require_gpu()
    M = _cloud_cost(127, 128, 2, 5, dev)
    p1, c1, i1 = ot.transport_exact(M, warm_start=False, return_info=True)
    p2, c2, i2 = ot.transport_exact(M, warm_start=False, return_info=True)
    assert i1["staged"] and torch.equal(p1, p2) and c1 == c2 and i1["phases"] == i2["phases"]
    M2 = _cloud_cost(140, 163, 2, 6, dev)
    p3, c3, i3 = ot.transport_exact(M2, warm_start=False, return_info=True)
    p4, c4, i4 = ot.transport_exact(M2, warm_start=False, return_info=True)
    assert not i3["staged"] and torch.equal(p3, p4) and c3 == c4
    assert c3 == pytest.approx(_lp_cost(M2.cpu().numpy()), rel=1e-9)


@pytest.mark.parametrize("B0,B1,d", [(511, 512, 2), (1000, 1001, 8), (1023, 1025, 3), (700, 1300, 4)])
def test_transport_at_the_sizes_round_5_refused(B0, B1, d):
    """B0 + B1 up to 2048 (round 5: 512).  Certified in fp64 by the solver's own pass; checked here against the
    dual bound it certifies with: cost == sum_i u_i / B0 + sum_j v_j / B1 cannot be recomputed from outside, so the
    independent check is the LP for 511 x 512 and, beyond, agreement of cold and warm start on the unique optimal cost."""
    import cfm_amd.optimal_transport as ot
    from cfm_amd import _lib
    dev = _lib.require_gpu()
    M = _cloud_cost(B0, B1, d, B0 + 7 * B1, dev)
    plan, cost, info = ot.transport_exact(M, return_info=True)
    P = plan.cpu().numpy()
    np.testing.assert_allclose(P.sum(1), 1.0 / B0, rtol=0, atol=1e-15)
    np.testing.assert_allclose(P.sum(0), 1.0 / B1, rtol=0, atol=1e-15)
    if B0 * B1 <= 300000:
        assert cost == pytest.approx(_lp_cost(M.cpu().numpy()), rel=1e-9)
    else:
        _, cost2, info2 = ot.transport_exact(M, warm_start=not info["warm_start_used"], return_info=True)
        assert cost == pytest.approx(cost2, rel=1e-12)
    print(f"{B0}x{B1}: phases {info['phases']} sweeps {info['sweeps']} support {info['support']} warm {info['warm_start_used']}")


def test_an_invalid_or_suboptimal_warm_start_is_ignored():
    """sigma is validated on the device (distinct, in range) and its duals must exist (label correcting must converge):
    a permutation that is NOT optimal has a negative cycle and is dropped; the result is the optimum either way."""
    import ctypes
    import cfm_amd.optimal_transport as ot
    from cfm_amd import _lib
    from cfm_amd._lib import check, ptr, stream_ptr
    dev = _lib.require_gpu()
    lib = _lib.load()
    B0, B1 = 63, 64
    M = _cloud_cost(B0, B1, 2, 9, dev)
    ref = _lp_cost(M.cpu().numpy())
    for sigma in (torch.arange(B0, dtype=torch.int32),                       # valid, not optimal
                  torch.zeros(B0, dtype=torch.int32),                        # not distinct
                  torch.full((B0,), 99, dtype=torch.int32)):                 # out of range
        plan = torch.empty((B0, B1), dtype=torch.float64, device=dev)
        tot = torch.empty(1, dtype=torch.float64, device=dev); info = torch.empty(8, dtype=torch.int32, device=dev)
        ws = _lib.workspace(_lib.OP_TRANSPORT, B0, B1, 0, dev)
        check(lib.cfm_transport_exact_f32(ptr(M), B0, B1, ptr(sigma.to(dev)), ptr(plan), ptr(tot), ptr(info), ptr(ws), stream_ptr()), "tp")
        st = info.cpu()
        assert int(st[0]) == 1 and not (int(st[7]) & 2)
        assert float(tot.cpu()[0]) == pytest.approx(ref, rel=1e-9)


def test_wasserstein_between_eval_sets_of_1000_and_1500_points():
    """The reference's evaluation call wasserstein(x0, x1) on sets of different sizes (optimal_transport.py:286-292): lcm 3000
    — the expanded assignment route; checked against SciPy's LSAP on the expanded matrix (an independent exact solver
    on the same fp32 costs) and, at 100 vs 150, against HiGHS."""
    import cfm_amd.optimal_transport as ot
    import cfm_oracle as oracle
    from cfm_amd import _lib
    dev = _lib.require_gpu()
    g = torch.Generator().manual_seed(12)
    x = torch.randn(1000, 2, generator=g); y = torch.randn(1500, 2, generator=g) * 0.8 + 0.4
    M = ot.cost_matrix(x.to(dev), y.to(dev)).cpu().numpy().astype(np.float64)
    _, ref_cost = oracle.exact_plan_rect(M)
    assert ot.wasserstein(x, y, method="exact", power=2) == pytest.approx(np.sqrt(ref_cost), rel=1e-9)
    Ms = ot.cost_matrix(x[:100].to(dev), y[:150].to(dev)).cpu().numpy()
    assert ot.wasserstein(x[:100], y[:150], method="exact", power=2) == pytest.approx(np.sqrt(_lp_cost(Ms)), rel=1e-6)


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
