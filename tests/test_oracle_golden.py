"""CPU: the oracle restatement against the golden vectors recorded from the reference's own
code (tests/golden/make_golden.py) — this is what pins the oracle."""
import os

import numpy as np
import pytest
import torch

import cfm_oracle as oracle

METHOD = {"i_cfm": "icfm", "exact_ot_cfm": "icfm", "t_cfm": "target", "vp_cfm": "vp",
          "sb_cfm_exact": "sb", "sb_cfm_sinkhorn": "sb"}


def _fm():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "fm_cases.npz"))


def _case(d, key):
    return {k: torch.from_numpy(d[f"{key}|{k}"]) for k in ("x0", "x1", "t", "xt", "ut", "eps")}


@pytest.mark.parametrize("key", list(_fm()["names"]))
def test_oracle_reproduces_reference_fm(key):
    """xt/ut/eps/t bit-equal to the reference classes under the reference seeds
    (the contract of reference tests/test_conditional_flow_matcher.py:97-127)."""
    d = _fm()
    c = _case(d, key)
    mname, sig, _ = key.split("|")
    sigma = eval(sig)
    x0, x1 = c["x0"], c["x1"]
    torch.manual_seed(1994)
    np.random.seed(1994)
    B = x0.shape[0]
    if mname in ("exact_ot_cfm", "sb_cfm_exact"):
        M = oracle.ref_cost_f32(x0, x1)
        i, j = oracle.sample_perm_given_u(oracle.exact_perm(M), np.random.random_sample(B))
        x0, x1 = x0[i], x1[j]
    elif mname == "sb_cfm_sinkhorn":
        M = oracle.ref_cost_f32(x0, x1)
        pi = oracle.sinkhorn_knopp(M, 2 * sigma**2)
        i, j = oracle.sample_map_given_u(pi, np.random.random_sample(B))
        x0, x1 = x0[i], x1[j]
    t = torch.rand(B).type_as(x0)
    eps = torch.randn_like(x0)
    xt, ut = oracle.xt_ut(METHOD[mname], x0, x1, t, eps, sigma)
    assert torch.equal(t, c["t"])
    assert torch.equal(eps, c["eps"])
    assert torch.all(xt.eq(c["xt"]))
    assert torch.all(ut.eq(c["ut"]))


def test_oracle_reproduces_reference_ot(golden_dir):
    d = np.load(os.path.join(golden_dir, "ot_cases.npz"))
    x0, x1 = torch.from_numpy(d["x0"]), torch.from_numpy(d["x1"])
    M = oracle.ref_cost_f32(x0, x1)
    perm = oracle.exact_perm(M)
    assert np.array_equal(perm, d["perm"])
    assert oracle.perm_plan(perm).max() == d["pi_nnz_value"]
    # sample_map (replace=True) restatement: flattened cdf == O(B) permutation restatement
    np.random.seed(7)
    u = np.random.random_sample(len(perm))
    i, j = oracle.sample_map_given_u(oracle.perm_plan(perm), u)
    assert np.array_equal(i, d["map_i"]) and np.array_equal(j, d["map_j"])
    i2, j2 = oracle.sample_perm_given_u(perm, u)
    assert np.array_equal(i2, d["map_i"]) and np.array_equal(j2, d["map_j"])
    # sample_plan consumes np.random first
    np.random.seed(1980)
    i, j = oracle.sample_perm_given_u(perm, np.random.random_sample(len(perm)))
    assert np.array_equal(x0[i].numpy(), d["sx0"]) and np.array_equal(x1[j].numpy(), d["sx1"])
    # scipy pairing keeps x0 order
    assert np.array_equal(x1.reshape(len(perm), -1)[perm].numpy(), d["scipy_x1"])
    # wasserstein
    W2 = np.sqrt(oracle.assignment_cost(M, perm) / len(perm))
    assert W2 == pytest.approx(float(d["W2_exact"]), rel=1e-12)
    M1 = torch.cdist(x0.reshape(len(perm), -1), x1.reshape(len(perm), -1)).numpy()
    p1 = oracle.exact_perm(M1)
    assert oracle.assignment_cost(M1, p1) / len(perm) == pytest.approx(float(d["W1_exact"]), rel=1e-12)
    # C1 (8gaussians -> moons, B=256): seeded generators are stable
    a, b = oracle.config_inputs("C1")
    assert np.array_equal(a.numpy(), d["c1_x0"]) and np.array_equal(b.numpy(), d["c1_x1"])
    assert np.array_equal(oracle.exact_perm(oracle.ref_cost_f32(a, b)), d["c1_perm"])


def test_oracle_sinkhorn_log_vs_knopp_and_golden(golden_dir):
    d = np.load(os.path.join(golden_dir, "sinkhorn_cases.npz"))
    M = d["M"]
    u, v, it, err = oracle.sinkhorn_log(M, 2.0)
    assert it == int(d["it_conv"])
    np.testing.assert_allclose(u, d["u_conv"], rtol=0, atol=1e-12)
    # at convergence the log-domain plan equals POT's default (Knopp) plan
    np.testing.assert_allclose(oracle.sinkhorn_plan(M, 2.0, u, v), d["knopp_conv"], rtol=1e-6, atol=1e-12)
    # marginals
    P = oracle.sinkhorn_plan(M, 2.0, u, v)
    np.testing.assert_allclose(P.sum(1), 1.0 / M.shape[0], rtol=1e-8)
    np.testing.assert_allclose(P.sum(0), 1.0 / M.shape[1], rtol=1e-6)
    for reg in (0.05, 0.5):
        uu, vv, _, _ = oracle.sinkhorn_log(M, reg, numItermax=int(d[f"it_{reg}"]), stopThr=0.0)
        np.testing.assert_allclose(uu, d[f"u_{reg}"], atol=1e-10)
        np.testing.assert_allclose(vv, d[f"v_{reg}"], atol=1e-10)


def test_choice_restatement_matches_numpy():
    """A.3: the cdf/searchsorted restatement is np.random.choice bit for bit."""
    rng = np.random.RandomState(0)
    for B in (16, 100, 128, 1000):
        p = rng.rand(B * B)
        p[rng.rand(B * B) < 0.9] = 0
        p /= p.sum()
        np.random.seed(B)
        ref = np.random.choice(B * B, p=p, size=B)
        np.random.seed(B)
        mine = oracle.choice_flat(p, np.random.random_sample(B))
        assert np.array_equal(ref, mine)
    for B in (100, 128, 1023, 4096):
        perm = rng.permutation(B)
        np.random.seed(B)
        ref = np.divmod(np.random.choice(B * B, p=(oracle.perm_plan(perm).flatten()), size=B), B) \
            if B <= 1023 else None
        np.random.seed(B)
        i, j = oracle.sample_perm_given_u(perm, np.random.random_sample(B))
        if ref is not None:
            assert np.array_equal(ref[0], i) and np.array_equal(ref[1], j)
        assert np.array_equal(j, perm[i])


def test_oracle_ode(golden_dir):
    d = np.load(os.path.join(golden_dir, "ode_cases.npz"))
    Ws = [d[f"W{k}"] for k in range(4)]
    bs = [d[f"b{k}"] for k in range(4)]
    f = lambda t, y: oracle.mlp_forward_f64(Ws, bs, y, t)
    # MLP restatement == torch eager (float64) on the reference module layout
    lin = [torch.nn.Linear(W.shape[1], W.shape[0]).double() for W in Ws]
    for l, W, b in zip(lin, Ws, bs):
        l.weight.data = torch.from_numpy(W).double(); l.bias.data = torch.from_numpy(b).double()
    net = torch.nn.Sequential(lin[0], torch.nn.SELU(), lin[1], torch.nn.SELU(), lin[2], torch.nn.SELU(), lin[3])
    x = torch.from_numpy(d["x"]).double()
    inp = torch.cat([x, torch.full((len(x), 1), 0.3, dtype=torch.float64)], 1)
    np.testing.assert_allclose(net(inp).detach().numpy(), d["mlp_out"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(oracle.mlp_forward_f64(Ws, bs, d["x"], 0.3), d["mlp_out"], rtol=1e-12)
    traj, info = oracle.dopri5_trajectory(f, d["x"], d["t_span"], 1e-4, 1e-4, return_log=True)
    assert info["steps"] == int(d["dopri5_steps"]) and info["nfe"] == int(d["dopri5_nfe"])
    np.testing.assert_allclose(traj, d["dopri5"], rtol=1e-12)
    # accuracy of the restated integrator against an independent high-order solver
    from scipy.integrate import solve_ivp
    x0 = d["x"][:8].astype(np.float64)
    sol = solve_ivp(lambda t, y: f(t, y.reshape(8, 2)).ravel(), (0.0, 1.0), x0.ravel(), method="DOP853",
                    rtol=1e-10, atol=1e-12)
    mine = oracle.dopri5_trajectory(f, x0, d["t_span"], 1e-6, 1e-6)
    assert np.abs(mine[-1].ravel() - sol.y[:, -1]).max() < 1e-4
    eul = oracle.euler_trajectory(f, d["x"], d["t_span"])
    np.testing.assert_allclose(eul, d["euler"], rtol=1e-12)


# ----------------------------------------------------------------------------- unbalanced / partial
def test_unbalanced_docstring_kat():
    """KAT of runner/src/models/components/sinkhorn_knopp_unbalanced.py:88-94 (printed to 8
    digits by a POT build whose stopping point differs in the 7th: 5e-7)."""
    G = oracle.sinkhorn_knopp_unbalanced([[0.0, 1.0], [1.0, 0.0]], 1.0, 1.0, a=[0.5, 0.5], b=[0.5, 0.5])
    assert np.abs(G - np.array([[0.51122814, 0.18807032], [0.18807032, 0.51122814]])).max() < 5e-7


def test_unbalanced_partial_fixtures_reproduce(golden_dir):
    d = np.load(os.path.join(golden_dir, "ub_cases.npz"))
    M = oracle.ref_cost_f32(d["x0"], d["x1"])
    assert np.array_equal(oracle.sinkhorn_knopp_unbalanced(M, 0.5, 1.0), d["unb_pi_0.5_1.0"])
    assert np.array_equal(oracle.entropic_partial_wasserstein(M, 2.0), d["par_pi_2.0"])
    P = d["par_m06"]
    assert abs(P.sum() - 0.6) < 1e-12 and (P.sum(1) <= 1 / 96 + 1e-12).all() and (P.sum(0) <= 1 / 96 + 1e-12).all()


def test_partial_vector_form_equals_matrix_form():
    """The device solver carries POT's three full Dykstra correction matrices as two vectors and a
    scalar; the algebra (q1 row-constant, q2 column-constant, q3 constant) is checked here."""
    rng = np.random.default_rng(1)
    M = rng.uniform(0, 3, size=(40, 28))
    ref = oracle.entropic_partial_wasserstein(M, 0.7, m=0.8, numItermax=60)
    n0, n1 = M.shape
    a, b, m = 1.0 / n0, 1.0 / n1, 0.8
    K0 = np.exp(M / -0.7); K0 *= m / K0.sum()
    al, be, rho, ka, sig = np.ones(n0), np.ones(n1), np.ones(n0), np.ones(n1), 1.0
    for _ in range(60):
        a1 = al * rho
        r = np.minimum(a / (a1 * (K0 @ be)), 1.0)
        a2 = r * a1
        rho = rho * al / a2
        b1 = be * ka
        kta = K0.T @ a2
        c = np.minimum(b / (b1 * kta), 1.0)
        b2 = c * b1
        ka = ka * be / b2
        S = sig * np.sum(b2 * kta)
        s = m / S
        al, be, sig = a2 * (sig * s), b2, 1.0 / s
    P = al[:, None] * K0 * be[None, :]
    assert np.abs(P - ref).max() <= 1e-12 * np.abs(ref).max()


def test_rectangular_exact_oracle_is_a_transport_plan():
    rng = np.random.default_rng(3)
    M = rng.uniform(0, 2, size=(6, 4))
    pi, cost = oracle.exact_plan_rect(M)
    assert np.allclose(pi.sum(1), 1 / 6) and np.allclose(pi.sum(0), 1 / 4)
    # brute-force check against a tiny LP (scipy.optimize.linprog, HiGHS)
    from scipy.optimize import linprog
    A = np.zeros((10, 24))
    for i in range(6):
        A[i, i * 4:(i + 1) * 4] = 1
    for j in range(4):
        A[6 + j, j::4] = 1
    res = linprog(M.ravel(), A_eq=A, b_eq=np.r_[np.full(6, 1 / 6), np.full(4, 1 / 4)], bounds=(0, None))
    assert abs(res.fun - cost) < 1e-10


@pytest.mark.parametrize("kind", ["offset", "duplicates", "self", "small_d"])
def test_cost_oracle_is_cancellation_safe(kind):
    """The float64 cost oracle (centred Gram form + plain differences where that cancels) against
    the plain-difference sum evaluated with math.fsum-grade care (float128 accumulation)."""
    rng = np.random.default_rng(11)
    d = 8 if kind == "small_d" else 96
    a = rng.standard_normal((40, d)).astype(np.float32)
    b = rng.standard_normal((37, d)).astype(np.float32)
    if kind == "offset":
        a, b = a + np.float32(1e4), b + np.float32(1e4)
    elif kind == "duplicates":
        b = (a[np.arange(37) % 40] + np.float32(1e-4) * rng.standard_normal((37, d))).astype(np.float32)
    elif kind == "self":
        b = a[:37].copy()
    M = oracle.sqeuclid_cost_f64(a, b)
    diff = a.astype(np.longdouble)[:, None, :] - b.astype(np.longdouble)[None, :, :]
    ref = (diff * diff).sum(-1).astype(np.float64)
    assert M.shape == ref.shape and (M >= 0).all()
    assert np.all(np.abs(M - ref) <= 1e-9 * np.maximum(ref, 1e-300))
    if kind == "self":
        assert (np.diag(M[:37, :37]) == 0).all()
