"""GPU parity tests: every HIP kernel, called through the C ABI, against the CPU oracle on the
same seeded inputs.  Integer / index results must be bit-exact; floating-point tolerances are
written next to each assertion."""
import ctypes
import os

import numpy as np
import pytest
import torch

import cfm_oracle as oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import cfm_amd  # noqa: F401
    from cfm_amd import _lib
    _lib.load()
    return _lib.require_gpu()


def _ot():
    import cfm_amd.optimal_transport as ot
    return ot


def _rand(B, d, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, d, generator=g) * scale


# ------------------------------------------------------------------ K1 cost
@pytest.mark.parametrize("B0,B1,d", [(256, 256, 2), (300, 200, 3), (128, 128, 8), (1, 1, 4), (5, 1030, 1),
                                      (257, 129, 50), (64, 64, 17), (512, 512, 784), (1024, 1024, 784),
                                      (1100, 1030, 36), (130, 1500, 12)])
def test_cost_matches_f64_oracle(dev, B0, B1, d):
    ot = _ot()
    x0, x1 = _rand(B0, d, 1), _rand(B1, d, 2) + 0.3
    M = ot.cost_matrix(x0.to(dev), x1.to(dev)).cpu().numpy()
    ref = oracle.sqeuclid_cost_f64(x0.numpy(), x1.numpy())
    # fp32 accumulation of d squared differences: error <= ~sqrt(d) ulp of the sum
    tol = 4e-7 * max(4.0, np.sqrt(d))
    err = np.abs(M - ref) / np.maximum(ref, 1e-12)
    assert M.shape == (B0, B1)
    assert err.max() < tol, (err.max(), np.unravel_index(err.argmax(), err.shape))
    assert (M >= 0).all()


def _direct_cost(x0, x1, dev):
    """The scratch-free entry point: always the direct-difference kernels."""
    from cfm_amd import _lib
    lib = _lib.load()
    a, b = x0.to(dev), x1.to(dev)
    M = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=dev)
    _lib.check(lib.cfm_sqeuclid_cost_f32(_lib.ptr(a), _lib.ptr(b), a.shape[0], b.shape[0], a.shape[1], _lib.ptr(M),
                                         None, _lib.stream_ptr()), "cfm_sqeuclid_cost_f32")
    return M.cpu().numpy()


def _cloud(kind, B0, B1, d):
    g = torch.Generator().manual_seed(5)
    rn = lambda *s: torch.randn(*s, generator=g)
    if kind == "offset":            # a cloud far from the origin: the uncentred Gram form would lose 4 digits
        return rn(B0, d) + 100.0, rn(B1, d) + 100.3
    if kind == "near_duplicates":   # every diagonal pair 1e-3 apart in a cloud of width 1
        x = rn(B0, d)
        return x, x[torch.arange(B1) % B0] + 1e-3 * rn(B1, d)
    if kind == "blobs":             # tight clusters: a whole block of pairs cancels
        c = rn(8, d) * 5
        return c[torch.arange(B0) % 8] + 0.01 * rn(B0, d), c[torch.arange(B1) % 8] + 0.01 * rn(B1, d)
    if kind == "pixels":            # the C3 shape: Gaussian noise against [0, 1] pixel-like rows
        return rn(B0, d), torch.rand(B1, d, generator=g).pow(4)
    if kind == "tiny":
        return rn(B0, d) * 1e-3, rn(B1, d) * 1e-3
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["offset", "near_duplicates", "blobs", "pixels", "tiny"])
@pytest.mark.parametrize("B0,B1,d", [(256, 256, 64), (300, 260, 100), (257, 515, 65), (640, 512, 784)])
def test_cost_matrix_core_form_vs_f64_oracle(dev, kind, B0, B1, d):
    """d >= 64, B >= 256: the centred Gram form on the MFMA units, cancelling entries recomputed
    directly.  Same bound as the direct kernels, entry by entry, on clouds built to break a Gram form."""
    ot = _ot()
    x0, x1 = _cloud(kind, B0, B1, d)
    M = ot.cost_matrix(x0.to(dev), x1.to(dev)).cpu().numpy()
    ref = oracle.sqeuclid_cost_f64(x0.numpy(), x1.numpy())
    tol = 4e-7 * max(4.0, np.sqrt(d))
    err = np.abs(M - ref) / np.maximum(ref, 1e-30)
    assert err.max() < tol, (kind, err.max(), np.unravel_index(err.argmax(), err.shape))
    assert (M >= 0).all()
    # and it agrees with the direct kernels to the sum of the two bounds
    Md = _direct_cost(x0, x1, dev)
    assert (np.abs(M - Md) <= 2 * tol * np.maximum(ref, 1e-30)).all()


def test_cost_matrix_core_form_self_distance_is_zero(dev):
    """x against itself: the diagonal cancels completely in a Gram form; it is recomputed -> exact 0."""
    ot = _ot()
    x = _rand(512, 128, 3) * 3 + 7
    M = ot.cost_matrix(x.to(dev), x.clone().to(dev)).cpu().numpy()
    assert (np.diag(M) == 0).all()
    assert np.array_equal(M, M.T) or np.abs(M - M.T).max() <= 1e-6 * M.max()
    ref = oracle.sqeuclid_cost_f64(x.numpy(), x.numpy())
    off = ~np.eye(512, dtype=bool)
    assert (np.abs(M - ref)[off] / ref[off]).max() < 4e-7 * np.sqrt(128)


def test_cost_8gaussians_moons_and_normalize(dev):
    ot = _ot()
    x0, x1 = oracle.config_inputs("C1")
    M = ot.cost_matrix(x0.to(dev), x1.to(dev)).cpu().numpy()
    ref = oracle.sqeuclid_cost_f64(x0, x1)
    assert np.abs(M - ref).max() <= 4e-7 * ref.max()
    Mn = ot.cost_matrix(x0.to(dev), x1.to(dev), normalize=True).cpu().numpy()
    assert Mn.max() == 1.0
    np.testing.assert_array_equal(Mn, M / M.max())          # IEEE division, same as M / M.max()
    M1 = ot.cost_matrix(x0.to(dev), x1.to(dev), squared=False).cpu().numpy()
    np.testing.assert_allclose(M1, np.sqrt(ref), rtol=1e-6, atol=1e-6)


# ------------------------------------------------------------------ K5 Sinkhorn
def _potentials(ot, r, B0, B1, dev):
    from cfm_amd import _lib
    lib = _lib.load()
    u = torch.empty(B0, dtype=torch.float64, device=dev)
    v = torch.empty(B1, dtype=torch.float64, device=dev)
    _lib.check(lib.cfm_sinkhorn_potentials_f64(_lib.ptr(r.ws), B0, B1, _lib.ptr(u), _lib.ptr(v),
                                               _lib.stream_ptr()), "potentials")
    return u.cpu().numpy(), v.cpu().numpy()


@pytest.mark.parametrize("reg,iters", [(0.05, 25), (0.5, 40)])
def test_sinkhorn_fixed_iterations_vs_golden(dev, golden_dir, reg, iters):
    """Potentials after a fixed number of iterations vs the float64 log-domain oracle
    (north star: within 1e-5 relative in fp32)."""
    ot = _ot()
    d = np.load(os.path.join(golden_dir, "sinkhorn_cases.npz"))
    M = torch.from_numpy(d["M"]).to(dev)
    r = ot.sinkhorn_log(M, reg, max_iter=iters, stop_thr=0.0)
    u, v = _potentials(ot, r, *M.shape, dev)
    uo, vo = d[f"u_{reg}"], d[f"v_{reg}"]
    assert int(r.iters.cpu()) == iters
    # fp64 log-scalings kept in the workspace
    assert np.abs(u - uo).max() < 2e-5 and np.abs(v - vo).max() < 2e-5, (np.abs(u - uo).max(), np.abs(v - vo).max())
    # fp32 potentials f = reg*u, g = reg*v : 1e-5 relative (to the potential scale)
    f, g = r.f.cpu().numpy(), r.g.cpu().numpy()
    sc = max(np.abs(reg * uo).max(), np.abs(reg * vo).max())
    assert np.abs(f - reg * uo).max() <= 1e-5 * sc
    assert np.abs(g - reg * vo).max() <= 1e-5 * sc
    assert float(r.err.cpu()) == pytest.approx(float(d[f"err_{reg}"]), rel=1e-3, abs=1e-12)


def test_sinkhorn_converges_with_pot_iteration_count(dev, golden_dir):
    ot = _ot()
    d = np.load(os.path.join(golden_dir, "sinkhorn_cases.npz"))
    M = torch.from_numpy(d["M"]).to(dev)
    r = ot.sinkhorn_log(M, 2.0)
    assert int(r.iters.cpu()) == int(d["it_conv"])
    u, v = _potentials(ot, r, *M.shape, dev)
    assert np.abs(u - d["u_conv"]).max() < 1e-6 and np.abs(v - d["v_conv"]).max() < 1e-6
    P = ot.sinkhorn_plan(r).cpu().numpy()
    np.testing.assert_allclose(P, d["knopp_conv"], rtol=2e-5, atol=1e-12)   # == POT's default plan
    np.testing.assert_allclose(P.sum(1), 1.0 / M.shape[0], rtol=1e-6)
    assert float(r.err.cpu()) < 1e-9


@pytest.mark.parametrize("B0,B1,reg", [(512, 384, 0.1), (1000, 1000, 0.05), (257, 1031, 1.0), (7, 5, 0.3)])
def test_sinkhorn_shapes_vs_oracle(dev, B0, B1, reg):
    ot = _ot()
    x0, x1 = _rand(B0, 2, 5, 2.0), _rand(B1, 2, 6, 2.0) + 1.0
    M = ot.cost_matrix(x0.to(dev), x1.to(dev))
    iters = 30
    r = ot.sinkhorn_log(M, reg, max_iter=iters, stop_thr=0.0)
    u, v = _potentials(ot, r, B0, B1, dev)
    uo, vo, _, _ = oracle.sinkhorn_log(M.cpu().numpy(), reg, numItermax=iters, stopThr=0.0)
    sc = max(np.abs(uo).max(), np.abs(vo).max(), 1.0)
    assert np.abs(u - uo).max() <= 1e-5 * sc and np.abs(v - vo).max() <= 1e-5 * sc, \
        (np.abs(u - uo).max(), np.abs(v - vo).max(), sc)


def test_sinkhorn_zero_iterations_and_cost(dev):
    ot = _ot()
    M = torch.rand(16, 16, device=dev)
    r = ot.sinkhorn_log(M, 0.5, max_iter=0)
    assert int(r.iters.cpu()) == 0 and float(r.f.abs().max().cpu()) == 0.0
    from cfm_amd import _lib
    lib = _lib.load()
    r = ot.sinkhorn_log(M, 0.5)
    out = torch.zeros(1, dtype=torch.float64, device=dev)
    _lib.check(lib.cfm_sinkhorn_cost_f64(_lib.ptr(M), 16, 16, 0.5, _lib.ptr(r.ws), _lib.ptr(out),
                                         _lib.stream_ptr()), "cost")
    uo, vo, _, _ = oracle.sinkhorn_log(M.cpu().numpy(), 0.5)
    P = oracle.sinkhorn_plan(M.cpu().numpy(), 0.5, uo, vo)
    assert float(out.cpu()) == pytest.approx(float((P * M.cpu().numpy().astype(np.float64)).sum()), rel=1e-6)


# ------------------------------------------------------------------ K4 exact assignment
def _check_perm(ot, Mnp, dev, expect_unique=True):
    M = torch.from_numpy(np.ascontiguousarray(Mnp, dtype=np.float32)).to(dev)
    perm, info = ot.assign_exact(M, return_info=True)
    p = perm.cpu().numpy().astype(np.int64)
    n = len(p)
    assert sorted(p.tolist()) == list(range(n)), "not a permutation"
    ref = oracle.exact_perm(Mnp)
    cost, cref = oracle.assignment_cost(Mnp, p), oracle.assignment_cost(Mnp, ref)
    assert info["certified"]
    assert info["total_cost"] == pytest.approx(cost, rel=1e-12, abs=1e-9)
    assert cost <= cref + 1e-9 * max(1.0, abs(cref)), (cost, cref, info)
    if expect_unique:
        assert np.array_equal(p, ref), (int((p != ref).sum()), info)   # plan indices bit-exact
    return info


@pytest.mark.parametrize("n", [2, 3, 7, 64, 128, 257, 512, 1000, 1024])
def test_assign_random_cost_matches_scipy(dev, n):
    rng = np.random.RandomState(n)
    _check_perm(_ot(), rng.rand(n, n).astype(np.float32) * 10, dev)


@pytest.mark.parametrize("cfg,B", [("C1", 256), ("C2", 1024), ("C3", 512), ("C5", 768)])
def test_assign_on_config_data_matches_scipy(dev, cfg, B):
    ot = _ot()
    x0, x1 = oracle.config_inputs(cfg, B=B)
    M = ot.cost_matrix(x0.to(dev), x1.to(dev)).cpu().numpy()
    info = _check_perm(ot, M, dev)
    print(cfg, B, info)


def test_assign_golden_c1_reference_cost_matrix(dev, golden_dir):
    """Solver-level parity on the reference's own fp32 matrix (torch.cdist ** 2)."""
    d = np.load(os.path.join(golden_dir, "ot_cases.npz"))
    M = oracle.ref_cost_f32(torch.from_numpy(d["c1_x0"]), torch.from_numpy(d["c1_x1"]))
    M_t = torch.from_numpy(M).to(dev)
    perm = _ot().assign_exact(M_t).cpu().numpy()
    assert np.array_equal(perm, d["c1_perm"])


def test_assign_degenerate_costs(dev):
    ot = _ot()
    rng = np.random.RandomState(0)
    _check_perm(ot, rng.randint(0, 5, size=(200, 200)).astype(np.float32), dev, expect_unique=False)  # many ties
    _check_perm(ot, np.zeros((65, 65), dtype=np.float32), dev, expect_unique=False)                   # all equal
    M = rng.rand(130, 130).astype(np.float32)
    M[np.arange(130), (np.arange(130) * 7) % 130] -= 5.0                                               # planted optimum
    _check_perm(ot, M, dev)
    _check_perm(ot, -rng.rand(90, 90).astype(np.float32) * 1e4, dev)                                   # negative, large scale
    _check_perm(ot, np.array([[3.0]], dtype=np.float32), dev)                                          # 1 x 1
    with pytest.raises(NotImplementedError):
        ot.assign_exact(torch.zeros(4, 5, device=dev))


@pytest.mark.parametrize("B0,B1", [(128, 64), (96, 64), (60, 100), (7, 3)])
def test_exact_rectangular_plan(dev, B0, B1):
    """Unequal batch sizes (pot.emd on unif(B0), unif(B1), ref:49,79): the plan is a transportation
    plan, not a permutation.  The optimal cost is unique (checked against SciPy on the lcm-expanded
    matrix built from the SAME fp32 costs); the plan itself must be feasible."""
    from cfm_amd.optimal_transport import OTPlanSampler, wasserstein
    ot = _ot()
    x0, x1 = _rand(B0, 3, 11), _rand(B1, 3, 12, scale=1.3) + 0.4
    M = ot.cost_matrix(x0.to(dev), x1.to(dev))
    s = OTPlanSampler(method="exact")
    pi = s.get_map(x0, x1)
    assert pi.shape == (B0, B1) and pi.dtype == np.float64
    assert np.abs(pi.sum(1) - 1.0 / B0).max() < 1e-12 and np.abs(pi.sum(0) - 1.0 / B1).max() < 1e-12
    Mnp = M.cpu().numpy().astype(np.float64)
    ref_pi, ref_cost = oracle.exact_plan_rect(Mnp)
    assert abs((pi * Mnp).sum() - ref_cost) <= 1e-12 * max(1.0, abs(ref_cost))
    assert abs(wasserstein(x0, x1, "exact") - np.sqrt(ref_cost)) <= 1e-9 * np.sqrt(ref_cost)
    np.random.seed(2)
    a, b = s.sample_plan(x0, x1)
    assert a.shape == x0.shape and b.shape == (B0,) + tuple(x1.shape[1:])
    np.random.seed(2)
    i, j = s.sample_map(pi, B0)
    assert torch.equal(a, x0[i]) and torch.equal(b, x1[j])


def test_exact_rectangular_large_lcm_refused(dev):
    """Unequal sizes whose lcm expansion is too large AND that are beyond the transportation solver (B0 + B1 <= 2048 since
    round 6; 1000 vs 1001 is solved now: tests/test_gpu_transport.py) are refused, not approximated."""
    from cfm_amd.optimal_transport import OTPlanSampler
    with pytest.raises(NotImplementedError):
        OTPlanSampler(method="exact").get_map(_rand(1500, 2, 1), _rand(1501, 2, 2))


# ------------------------------------------------------------------ K6 sampling
@pytest.mark.parametrize("B", [100, 128, 1000, 1023, 4096])
def test_sample_perm_bit_exact(dev, B):
    ot = _ot()
    rng = np.random.RandomState(B)
    perm = rng.permutation(B)
    u = rng.random_sample(B)
    i, j = ot.sample_perm(torch.from_numpy(perm.astype(np.int32)).to(dev), torch.from_numpy(u).to(dev), B)
    io, jo = oracle.sample_perm_given_u(perm, u)
    assert np.array_equal(i.cpu().numpy(), io) and np.array_equal(j.cpu().numpy(), jo)


def test_sample_pi_matches_numpy_choice(dev):
    ot = _ot()
    rng = np.random.RandomState(1)
    for (B0, B1, n) in ((64, 64, 64), (100, 37, 250), (257, 300, 1000)):
        pi = rng.rand(B0, B1)
        pi[rng.rand(B0, B1) < 0.7] = 0
        pi[3] = 0                                       # an empty row
        np.random.seed(B0)
        ref = np.divmod(np.random.choice(B0 * B1, p=(pi / pi.sum()).flatten(), size=n), B1)
        np.random.seed(B0)
        u = np.random.random_sample(n)
        i, j = ot.sample_pi(torch.from_numpy(pi).to(dev), torch.from_numpy(u).to(dev))
        assert np.array_equal(i.cpu().numpy(), ref[0]) and np.array_equal(j.cpu().numpy(), ref[1])


def test_sample_dense_from_potentials_matches_oracle(dev):
    """Indices drawn from the never-materialised Sinkhorn plan == np.random.choice on the dense
    fp64 plan built from the same device potentials."""
    ot = _ot()
    x0, x1 = oracle.config_inputs("C1")
    M = ot.cost_matrix(x0.to(dev), x1.to(dev))
    for reg in (0.05, 1.0):
        r = ot.sinkhorn_log(M, reg)
        u, v = _potentials(ot, r, 256, 256, dev)
        P = oracle.sinkhorn_plan(M.cpu().numpy(), reg, u, v)
        np.random.seed(3)
        uu = np.random.random_sample(256)
        i, j = ot.sample_dense(r, torch.from_numpy(uu).to(dev))
        io, jo = oracle.sample_map_given_u(P, uu)
        i, j = i.cpu().numpy(), j.cpu().numpy()
        mism = int((i != io).sum() + (j != jo).sum())
        assert mism == 0, mism
        Pd = ot.sinkhorn_plan(r).cpu().numpy()
        np.testing.assert_allclose(Pd, P, rtol=1e-12, atol=1e-300)


# ------------------------------------------------------------------ K7/K8 fused xt/ut
VAR = {"icfm": 0, "sb": 1, "target": 2, "vp": 3}


@pytest.mark.parametrize("method", ["icfm", "sb", "target", "vp"])
@pytest.mark.parametrize("sigma", [0.0, 5e-4, 0.5, 1.5, 0, 1])
@pytest.mark.parametrize("shape", [(2,), (1, 2), (3, 4, 5), (784,), (7,)])
def test_xt_ut_bit_exact(dev, method, sigma, shape):
    from cfm_amd.conditional_flow_matching import _fused_xt_ut
    if method == "sb" and sigma <= 0:
        pytest.skip("SB needs sigma > 0")
    B = 96
    g = torch.Generator().manual_seed(len(shape) * 100 + int(sigma * 10))
    x0, x1 = torch.randn(B, *shape, generator=g), torch.randn(B, *shape, generator=g)
    t, eps = torch.rand(B, generator=g), torch.randn(B, *shape, generator=g)
    xt, ut = _fused_xt_ut(VAR[method], sigma, x0, x1, t, eps)
    xo, uo = oracle.xt_ut(method, x0, x1, t, eps, sigma)
    assert torch.all(xt.eq(xo)), (xt - xo).abs().max()
    assert torch.all(ut.eq(uo)), (ut - uo).abs().max()
    # gather-on-load == index then compute
    i = torch.randint(0, B, (B,), generator=g)
    j = torch.randint(0, B, (B,), generator=g)
    xt2, ut2 = _fused_xt_ut(VAR[method], sigma, x0, x1, t, eps, idx=(i.to(dev), j.to(dev)))
    xo2, uo2 = oracle.xt_ut(method, x0[i], x1[j], t, eps, sigma)
    assert torch.all(xt2.eq(xo2)) and torch.all(ut2.eq(uo2))
    # compute_conditional_flow with a caller-supplied xt
    _, ut3 = _fused_xt_ut(VAR[method], sigma, x0, x1, t, None, xt_in=xo, want_xt=False)
    assert torch.all(ut3.eq(uo))


def test_gather_rows(dev):
    ot = _ot()
    g = torch.Generator().manual_seed(0)
    for shape, dt in (((50, 7), torch.float32), ((33, 4, 4), torch.float32), ((40, 1), torch.int64), ((40,), torch.int64)):
        src = (torch.randn(*shape, generator=g) * 100).to(dt)
        idx = torch.randint(0, shape[0], (77,), generator=g)
        out = ot.gather_rows(src.to(dev), idx.to(dev)).cpu()
        assert torch.equal(out, src[idx])


# ------------------------------------------------------------------ K10 MLP, K11 ODE
def _mlp(d, w, seed=0):
    import cfm_amd
    torch.manual_seed(seed)
    return cfm_amd.MLP(dim=d, time_varying=True, w=w)


@pytest.mark.parametrize("B,d,w", [(64, 2, 64), (300, 2, 64), (256, 50, 64), (512, 784, 512), (130, 5, 96)])
def test_mlp_forward_vs_f64_oracle(dev, B, d, w):
    m = _mlp(d, w)
    Ws = [l.weight.detach().numpy() for l in m._linears()]
    bs = [l.bias.detach().numpy() for l in m._linears()]
    x = _rand(B, d, 3)
    for t in (0.3, torch.rand(B)):
        y = m.forward_hip(x, t).numpy()
        ref = oracle.mlp_forward_f64(Ws, bs, x.numpy(), t if isinstance(t, float) else t.numpy())
        # north star: 1e-5 relative in fp32 (relative to the output scale)
        assert np.abs(y - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), np.abs(y - ref).max()
    # reference layout: time already concatenated, no-grad forward() -> HIP path
    xin = torch.cat([x, torch.full((B, 1), 0.3)], 1)
    with torch.no_grad():
        y2 = m(xin.to(dev)).cpu().numpy()
    ref = oracle.mlp_forward_f64(Ws, bs, x.numpy(), 0.3)
    assert np.abs(y2 - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


def test_ode_euler_and_dopri5_vs_golden(dev, golden_dir):
    import cfm_amd
    from cfm_amd.ode import NeuralODE
    from cfm_amd.utils import torch_wrapper
    d = np.load(os.path.join(golden_dir, "ode_cases.npz"))
    m = cfm_amd.MLP(dim=2, time_varying=True, w=64)
    for k, l in enumerate(m._linears()):
        l.weight.data = torch.from_numpy(d[f"W{k}"]); l.bias.data = torch.from_numpy(d[f"b{k}"])
    x, ts = torch.from_numpy(d["x"]), torch.from_numpy(d["t_span"])
    node = NeuralODE(torch_wrapper(m), solver="euler")
    tr = node.trajectory(x, ts).numpy()
    assert tr.shape == d["euler"].shape and node.nfe == len(ts) - 1
    assert np.abs(tr - d["euler"]).max() <= 1e-5 * np.abs(d["euler"]).max()
    node = NeuralODE(torch_wrapper(m), solver="dopri5", sensitivity="adjoint", atol=1e-4, rtol=1e-4)
    tr = node.trajectory(x, ts).numpy()
    # same step sequence as the restated controller, trajectory within 1e-5 relative
    assert node.n_steps == int(d["dopri5_steps"]) and node.nfe == int(d["dopri5_nfe"]), (node.n_steps, node.nfe)
    assert np.abs(tr - d["dopri5"]).max() <= 1e-5 * np.abs(d["dopri5"]).max(), np.abs(tr - d["dopri5"]).max()
    np.testing.assert_array_equal(tr[0], d["x"])


@pytest.mark.parametrize("B,d,w,n_t", [(300, 2, 64, 25), (257, 50, 64, 12), (64, 63, 33, 4), (8192, 50, 64, 6),
                                        (8231, 3, 48, 5), (20000, 2, 64, 4)])
def test_ode_fused_small_field_equals_layer_path(dev, B, d, w, n_t):
    """Small vector fields (every width <= 64) take the fused drivers (the whole adaptive dopri5 solve
    in one persistent launch with the controller on the device -- x / k1 resident in registers up to
    B = 8192, streamed through the parity buffers above (the last two cases); the whole t_span in one
    launch for euler).  Same ascending-k fp32 fma chain and epilogues as the layer-per-kernel path on the
    register-staged core (cfm_mlp_set_glds(0)): the trajectories are bit-equal.  The layers' default engine since
    round 6 (direct-to-LDS operands, gemm_glds64.h) sums k in a different fixed order: within 1e-5 of it."""
    import cfm_amd
    from cfm_amd import _lib
    from cfm_amd.ode import NeuralODE
    from cfm_amd.utils import torch_wrapper
    lib = _lib.load()
    torch.manual_seed(3)
    model = cfm_amd.MLP(dim=d, time_varying=True, w=w).to(dev)
    x = _rand(B, d, 9).to(dev)
    ts = torch.linspace(0, 1, n_t, device=dev)
    out = {}
    glds0 = lib.cfm_mlp_get_glds()
    try:
        # (the fused dopri5 driver takes its first two field evaluations — Hairer's initial step — from the layer kernels:
        #  both legs of the bit comparison run them on the same engine)
        for fused, glds in ((1, 0), (0, 0), (1, glds0), (0, glds0)):
            lib.cfm_ode_set_fused(fused); lib.cfm_mlp_set_glds(glds)
            for solver in ("dopri5", "euler"):
                node = NeuralODE(torch_wrapper(model), solver=solver, atol=1e-4, rtol=1e-4)
                with torch.no_grad():
                    out[(fused, glds, solver)] = (node.trajectory(x, ts).cpu(), node.nfe, node.n_steps)
    finally:
        lib.cfm_ode_set_fused(1); lib.cfm_mlp_set_glds(glds0)
    for solver in ("dopri5", "euler"):
        a, b = out[(1, 0, solver)], out[(0, 0, solver)]
        assert torch.equal(a[0], b[0]) and a[1] == b[1] and a[2] == b[2], solver
        for c in (out[(1, glds0, solver)], out[(0, glds0, solver)]):      # the default engine: another fixed k order
            if solver == "euler" or c[2] == a[2]:       # (an adaptive solve may take another step sequence on other bits)
                assert float((c[0] - a[0]).abs().max()) <= 1e-5 * float(a[0].abs().max()), solver
