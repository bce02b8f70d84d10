"""CPU: the round-2 oracle pieces are pinned to each other and to the committed fixtures —
the threaded C Sinkhorn oracle against the NumPy one, the two independent dopri5 restatements
against each other (accept / reject sequences), the reference's O(B^2) sample_map against the
O(B) restatement the device kernel follows."""
import os

import numpy as np
import pytest
import torch

import cfm_oracle as oracle
import sinkhorn_c


def test_c_sinkhorn_oracle_equals_numpy_oracle():
    rng = np.random.RandomState(0)
    x, y = rng.randn(300, 2), rng.randn(200, 2) + 1
    M = ((x[:, None] - y[None]) ** 2).sum(-1).astype(np.float32)
    for reg, kw in ((0.05, dict(numItermax=40, stopThr=0.0)), (2.0, {}), (0.3, dict(numItermax=25, stopThr=0.0, check_every=5))):
        a = oracle.sinkhorn_log(M, reg, **kw)
        b = sinkhorn_c.sinkhorn_log(M, reg, **kw)
        assert a[2] == b[2]
        assert np.abs(a[0] - b[0]).max() < 1e-11 and np.abs(a[1] - b[1]).max() < 1e-11
        assert b[3] == pytest.approx(a[3], rel=1e-6, abs=1e-15)


def test_c_sinkhorn_oracle_matches_golden_fixture(golden_dir):
    d = np.load(os.path.join(golden_dir, "sinkhorn_cases.npz"))
    u, v, it, err = sinkhorn_c.sinkhorn_log(d["M"], 2.0)
    assert it == int(d["it_conv"])
    assert np.abs(u - d["u_conv"]).max() < 1e-11 and np.abs(v - d["v_conv"]).max() < 1e-11


def test_reference_sample_map_equals_given_u_restatements():
    rng = np.random.RandomState(3)
    perm = rng.permutation(64)
    pi = oracle.perm_plan(perm)
    np.random.seed(11); i_ref, j_ref = oracle.sample_map_reference(pi, 64)
    np.random.seed(11); u = np.random.random_sample(64)
    i1, j1 = oracle.sample_map_given_u(pi, u)
    i2, j2 = oracle.sample_perm_given_u(perm, u)
    assert np.array_equal(i_ref, i1) and np.array_equal(j_ref, j1)
    assert np.array_equal(i_ref, i2) and np.array_equal(j_ref, j2)


def _ctrl_field(d, name):
    Wc = [d[f"c_W{k}"].copy() for k in range(4)]
    Wc[0][:, 2] *= float(d[f"c_{name}_tw"]); Wc[3] *= float(d[f"c_{name}_ow"])
    return Wc, [d[f"c_b{k}"] for k in range(4)]


@pytest.mark.parametrize("name", ["a", "b", "c", "d"])
def test_dopri5_restatements_agree(golden_dir, name):
    """cfm_oracle.dopri5_trajectory (float64 state, float32 scalar controller) against the independent
    eager-torch restatement run the way torchdyn runs (everything in the dtype of x).  float32: the
    accept / reject sequences are identical.  float64: identical except case b, where the float64
    controller needs one more step to land on T (t + dt vs T compares differently in the last ulp) —
    the documented place where float32 control diverges from float64 control; the states at the
    t_span points agree to 1e-6 either way."""
    d = np.load(os.path.join(golden_dir, "ode2_cases.npz"))
    Wc, bc = _ctrl_field(d, name)
    x, ts, tol = d["c_x"], d[f"c_{name}_t_span"], float(d[f"c_{name}_tol"])
    f = lambda t, y: oracle.mlp_forward_f64(Wc, bc, y, t)
    tr, info = oracle.dopri5_trajectory(f, x, ts, tol, tol, return_log=True)
    acc = [l[3] for l in info["log"]]
    assert acc == d[f"c_{name}_accept"].tolist() and info["steps"] == int(d[f"c_{name}_steps"])
    np.testing.assert_allclose(tr, d[f"c_{name}_traj"], rtol=0, atol=1e-12)
    if name == "c":
        assert acc.count(False) >= 2            # the case with rejected steps
    for dt in (torch.float32, torch.float64):
        tr2, info2 = oracle.dopri5_trajectory_torch(oracle.mlp_field_torch(Wc, bc, dt), x, ts, tol, tol, dtype=dt,
                                                    return_log=True)
        acc2 = [l[3] for l in info2["log"]]
        if dt == torch.float32 or name != "b":
            assert acc2 == acc, (name, dt)
        else:
            assert len(acc2) == len(acc) + 1 and acc2.count(False) == acc.count(False)
        scale = np.abs(tr).max()
        assert np.abs(tr2.double().numpy() - tr).max() <= (5e-6 if dt == torch.float32 else 1e-6) * scale


def test_ode2_fixture_is_what_the_oracle_computes(golden_dir):
    d = np.load(os.path.join(golden_dir, "ode2_cases.npz"))
    Ws, bs = [d[f"s_W{k}"] for k in range(4)], [d[f"s_b{k}"] for k in range(4)]
    f = lambda t, y: oracle.mlp_forward_f64(Ws, bs, y, t)
    tr, info = oracle.dopri5_trajectory(f, d["s_x"], d["s_t_span"], 1e-4, 1e-4, return_log=True)
    assert info["steps"] == int(d["s_steps"]) and info["nfe"] == int(d["s_nfe"])
    np.testing.assert_allclose(tr, d["s_dopri5"], rtol=0, atol=1e-12)
    # accuracy of the restated integrator itself: against SciPy's DOP853 at tight tolerances
    from scipy.integrate import solve_ivp
    n, dd = d["s_x"].shape
    sol = solve_ivp(lambda t, y: f(t, y.reshape(n, dd)).ravel(), (0.0, 1.0), d["s_x"].astype(np.float64).ravel(),
                    method="DOP853", rtol=1e-10, atol=1e-12)
    end = sol.y[:, -1].reshape(n, dd)
    assert np.abs(tr[-1] - end).max() <= 1e-3 * max(1.0, np.abs(end).max())       # O(atol = rtol = 1e-4)


def test_mlp_backward_oracle_equals_torch_autograd_f64():
    torch.manual_seed(2)
    lins = [torch.nn.Linear(6, 16), torch.nn.Linear(16, 16), torch.nn.Linear(16, 16), torch.nn.Linear(16, 5)]
    net = torch.nn.Sequential(lins[0], torch.nn.SELU(), lins[1], torch.nn.SELU(), lins[2], torch.nn.SELU(), lins[3]).double()
    x = torch.randn(20, 6, dtype=torch.float64, requires_grad=True)
    dout = torch.randn(20, 5, dtype=torch.float64)
    out = net(x)
    (out * dout).sum().backward()
    Ws = [l.weight.detach().numpy() for l in lins]; bs = [l.bias.detach().numpy() for l in lins]
    o, dW, db, dx = oracle.mlp_backward_f64(Ws, bs, x.detach().numpy(), dout.numpy())
    np.testing.assert_allclose(o, out.detach().numpy(), rtol=1e-12, atol=1e-12)
    for l, lin in enumerate(lins):
        np.testing.assert_allclose(dW[l], lin.weight.grad.numpy(), rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(db[l], lin.bias.grad.numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(dx, x.grad.numpy(), rtol=1e-10, atol=1e-12)


def test_adam_oracle_equals_torch_adam_f64():
    torch.manual_seed(4)
    p = torch.nn.Parameter(torch.randn(7, 3, dtype=torch.float64))
    p0 = p.detach().numpy().copy()
    opt = torch.optim.Adam([p], lr=3e-3, weight_decay=0.02)
    m = np.zeros_like(p0); v = np.zeros_like(p0); q = p0
    for step in range(1, 4):
        g = torch.randn(7, 3, dtype=torch.float64)
        p.grad = g.clone(); opt.step()
        q, m, v = oracle.adam_step_f64(q, g.numpy(), m, v, step, lr=3e-3, weight_decay=0.02)
        np.testing.assert_allclose(q, p.detach().numpy(), rtol=1e-12, atol=1e-14)
