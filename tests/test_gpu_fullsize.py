"""GPU: BASELINE.json's full sizes.  Where the oracle is too slow at full size, parity goes
through size-independent properties (permutation + certificate + cost vs SciPy on a matching
matrix, marginals of the entropic plan, round trips of the sampler)."""
import time

import numpy as np
import pytest
import torch

import cfm_oracle as oracle

pytestmark = pytest.mark.gpu


def test_c3_exact_ot_b4096_d784_plan_indices_bit_exact():
    """North-star headline: B=4096, d=784 exact-OT plan indices identical to SciPy's LSAP on the
    same fp32 cost matrix, end to end (own cost kernel)."""
    import cfm_amd.optimal_transport as ot
    from cfm_amd import _lib
    dev = _lib.require_gpu()
    x0, x1 = oracle.config_inputs("C3")
    M = ot.cost_matrix(x0.to(dev), x1.to(dev))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    perm, info = ot.assign_exact(M, return_info=True)
    torch.cuda.synchronize()
    t_gpu = time.perf_counter() - t0
    Mh = M.cpu().numpy()
    t0 = time.perf_counter()
    ref = oracle.exact_perm(Mh)
    t_cpu = time.perf_counter() - t0
    p = perm.cpu().numpy()
    print(f"assign B=4096 d=784: gpu {t_gpu*1e3:.1f} ms, scipy {t_cpu:.1f} s, stats {info['stats']}")
    assert sorted(p.tolist()) == list(range(4096))
    assert info["certified"]
    assert np.array_equal(p, ref), int((p != ref).sum())
    # end-to-end vs the reference's own cost matrix (torch.cdist**2 on CPU): SURVEY §0.5 found
    # 0 differing indices at this size
    # END TO END vs the reference path on the same inputs (optimal_transport.py:84-87): LSAP on the reference's OWN
    # fp32 matrix torch.cdist(x0, x1) ** 2 (CPU) against the product path (Gram-form MFMA cost kernel + HIP solver).
    # SURVEY 0.5 measured 0 differing indices at this size; where the two fp32 matrices ever decide an index
    # differently, the two optima must agree to 1e-9 relative when both are priced on the reference's matrix.
    Mref = oracle.ref_cost_f32(x0, x1)
    ref2 = oracle.exact_perm(Mref)
    ndiff = int((p != ref2).sum())
    ar = np.arange(4096)
    c_ref = float(Mref.astype(np.float64)[ar, ref2].sum()); c_gpu = float(Mref.astype(np.float64)[ar, p].sum())
    gap = (c_gpu - c_ref) / abs(c_ref)
    print(f"C3 end to end: {ndiff} of 4096 indices differ from LSAP on the reference's cdist**2 matrix; "
          f"cost gap on that matrix {gap:.3e} relative")
    assert gap >= -1e-12                      # ref2 is optimal on Mref
    assert ndiff == 0 or gap <= 1e-9, (ndiff, gap)
    assert ndiff == 0, ndiff                  # the north star's "plan indices bit-exact" at the headline config


def test_c2_dense_d2_b4096_certificate_and_cost():
    """B=4096, d=2 (8gaussians->moons): SciPy needs ~45 s here, so check the device certificate
    and compare the total cost with SciPy on the B=2048 sub-problem."""
    import cfm_amd.optimal_transport as ot
    from cfm_amd import _lib
    dev = _lib.require_gpu()
    x0, x1 = oracle.config_inputs("C2")
    M = ot.cost_matrix(x0.to(dev), x1.to(dev))
    perm, info = ot.assign_exact(M, return_info=True)
    p = perm.cpu().numpy()
    assert sorted(p.tolist()) == list(range(4096)) and info["certified"]
    print("C2 assign stats", info)
    Ms = M[:2048, :2048].contiguous()
    perm2, info2 = ot.assign_exact(Ms, return_info=True)
    ref = oracle.exact_perm(Ms.cpu().numpy())
    assert np.array_equal(perm2.cpu().numpy(), ref)


def test_c2_sinkhorn_b4096_eps005_properties():
    """Sinkhorn eps=0.05 at B=4096,d=2: after the row update the row marginals are exact; the
    potentials match the float64 oracle after 20 iterations; sampler round trip."""
    import cfm_amd.optimal_transport as ot
    from cfm_amd import _lib
    dev = _lib.require_gpu()
    lib = _lib.load()
    x0, x1 = oracle.config_inputs("C2")
    M = ot.cost_matrix(x0.to(dev), x1.to(dev))
    r = ot.sinkhorn_log(M, 0.05, max_iter=20, stop_thr=0.0)
    u = torch.empty(4096, dtype=torch.float64, device=dev); v = torch.empty_like(u)
    _lib.check(lib.cfm_sinkhorn_potentials_f64(_lib.ptr(r.ws), 4096, 4096, _lib.ptr(u), _lib.ptr(v), _lib.stream_ptr()), "pot")
    uo, vo, _, _ = oracle.sinkhorn_log(M.cpu().numpy(), 0.05, numItermax=20, stopThr=0.0)
    sc = max(np.abs(uo).max(), np.abs(vo).max())
    du, dv = np.abs(u.cpu().numpy() - uo).max(), np.abs(v.cpu().numpy() - vo).max()
    print("C2 sinkhorn potentials abs err", du, dv, "scale", sc)
    assert du <= 1e-5 * sc and dv <= 1e-5 * sc
    P = ot.sinkhorn_plan(r)
    rs = P.sum(1).cpu().numpy()
    np.testing.assert_allclose(rs, 1.0 / 4096, rtol=1e-5)
    np.random.seed(1)
    uu = np.random.random_sample(4096)
    i, j = ot.sample_dense(r, torch.from_numpy(uu).to(dev))
    io, jo = oracle.sample_map_given_u(P.cpu().numpy(), uu)
    assert np.array_equal(i.cpu().numpy(), io) and np.array_equal(j.cpu().numpy(), jo)


def test_c3_full_step_matches_oracle_step():
    """Whole OT-CFM step at B=4096,d=784 vs the CPU restatement on the same cost matrix."""
    import cfm_amd.optimal_transport as ot
    from cfm_amd import _lib
    from cfm_amd.conditional_flow_matching import ExactOptimalTransportConditionalFlowMatcher
    dev = _lib.require_gpu()
    x0, x1 = oracle.config_inputs("C3")
    fm = ExactOptimalTransportConditionalFlowMatcher(sigma=0.1)
    torch.manual_seed(0); np.random.seed(0)
    t, xt, ut, eps = fm.sample_location_and_conditional_flow(x0, x1, return_noise=True)
    M = ot.cost_matrix(x0.to(dev), x1.to(dev)).cpu().numpy()
    torch.manual_seed(0); np.random.seed(0)
    t2, xt2, ut2, eps2, _ = oracle.ot_cfm_step(x0, x1, sigma=0.1, M=M)
    assert torch.equal(t, t2) and torch.equal(eps, eps2)
    assert torch.all(xt.eq(xt2)) and torch.all(ut.eq(ut2))


def test_c5_shapes_run():
    """C5 (B=8192, d=50): Sinkhorn eps=0.1 + dopri5 sampling run and satisfy their invariants."""
    import cfm_amd
    import cfm_amd.optimal_transport as ot
    from cfm_amd import _lib
    from cfm_amd.ode import NeuralODE
    from cfm_amd.utils import torch_wrapper
    dev = _lib.require_gpu()
    x0, x1 = oracle.config_inputs("C5")
    M = ot.cost_matrix(x0.to(dev), x1.to(dev))
    r = ot.sinkhorn_log(M, 0.1, max_iter=50, stop_thr=0.0)
    P = ot.sinkhorn_plan(r)
    np.testing.assert_allclose(P.sum(1).cpu().numpy(), 1.0 / 8192, rtol=1e-5)
    torch.manual_seed(0)
    m = cfm_amd.MLP(dim=50, time_varying=True, w=64)
    node = NeuralODE(torch_wrapper(m), solver="dopri5", sensitivity="adjoint", atol=1e-4, rtol=1e-4)
    traj = node.trajectory(x0, torch.linspace(0, 1, 100))
    assert traj.shape == (100, 8192, 50) and torch.isfinite(traj).all()
    print("C5 dopri5 steps", node.n_steps, "nfe", node.nfe)


def test_c2_same_matrix_b4096_all_indices():
    """Solver-level bit-exactness at the FULL C2 size (VERDICT r4 Next #7a): ONE fp32 matrix — the reference's own
    torch.cdist(x0, x1) ** 2 on CPU (optimal_transport.py:84) — goes to both cfm_assign_exact_f32 and SciPy's LSAP
    (the reference's solver at :179; float64 on the same fp32 values): all 4096 indices must agree.  (End to end the
    d = 2 optimum is decided below fp32 cost rounding — SURVEY 0.5 — which is why the matrix is shared here.)"""
    import cfm_amd.optimal_transport as ot
    from cfm_amd import _lib
    dev = _lib.require_gpu()
    x0, x1 = oracle.config_inputs("C2")
    Mref = oracle.ref_cost_f32(x0, x1)
    assert Mref.shape == (4096, 4096) and Mref.dtype == np.float32
    perm, info = ot.assign_exact(torch.from_numpy(Mref).to(dev), return_info=True)
    p = perm.cpu().numpy()
    assert info["certified"] and sorted(p.tolist()) == list(range(4096))
    t0 = time.perf_counter()
    ref = oracle.exact_perm(Mref)
    print(f"C2 same-matrix: SciPy LSAP {time.perf_counter() - t0:.1f} s")
    ndiff = int((p != ref).sum())
    if ndiff:       # a tie between two optima is the only legitimate difference: say so precisely
        ar = np.arange(4096); M64 = Mref.astype(np.float64)
        print("cost gap", float(M64[ar, p].sum() - M64[ar, ref].sum()))
    assert ndiff == 0, ndiff


def test_c5_sample_dense_b8192_vs_oracle():
    """C5 full size (B = 8192, d = 50, eps = 0.1): the plan SAMPLER at full size (VERDICT r4 Next #7b).
    (i) oracle.sample_map_given_u on the device solver's own float64 plan == the device draws, all 8192;
    (ii) against the plan built from the C ORACLE's potentials (oracle/sinkhorn_oracle.c, 50 iterations of POT's
    loop on the same fp32 matrix): the potentials agree to 1e-5, so a draw can only differ where u falls within
    that of a cdf step — at most a handful, and then by a neighbouring support entry."""
    import cfm_amd.optimal_transport as ot
    import sinkhorn_c
    from cfm_amd import _lib
    dev = _lib.require_gpu()
    x0, x1 = oracle.config_inputs("C5")
    B = 8192
    M = ot.cost_matrix(x0.to(dev), x1.to(dev))
    r = ot.sinkhorn_log(M, 0.1, max_iter=50, stop_thr=0.0)
    np.random.seed(5)
    uu = np.random.random_sample(B)
    i, j = ot.sample_dense(r, torch.from_numpy(uu).to(dev))
    i, j = i.cpu().numpy(), j.cpu().numpy()
    P = ot.sinkhorn_plan(r).cpu().numpy()
    io, jo = oracle.sample_map_given_u(P, uu)
    assert np.array_equal(i, io) and np.array_equal(j, jo)
    del P
    Mh = M.cpu().numpy()
    uo, vo, _, _ = sinkhorn_c.sinkhorn_log(Mh, 0.1, numItermax=50, stopThr=0.0)
    Po = np.exp(uo[:, None] + vo[None, :] - Mh.astype(np.float64) / 0.1)
    ic, jc = oracle.sample_map_given_u(Po, uu)
    same = (i == ic) & (j == jc)
    print(f"C5 sampler vs C-oracle plan: {int(same.sum())} of {B} draws identical")
    assert same.mean() >= 0.995
    bad = np.where(~same)[0]
    # a differing draw sits on a cdf boundary: both candidates carry mass in BOTH plans' neighbourhoods
    for q in bad:
        assert Po[i[q], j[q]] > 0 and abs(int(i[q]) - int(ic[q])) <= 1
