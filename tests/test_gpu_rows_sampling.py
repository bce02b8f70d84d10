"""GPU: per-row plan sampling on the device (cfm_plan_sample_rows_dense / _pi_f64) — the inner loop of
OTPlanSampler.sample_trajectory (torchcfm/optimal_transport.py:237-246) — against np.random.choice on the plan rows."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from cfm_amd import _lib
    _lib.load()
    return _lib.require_gpu()


def _host_rows(pi, rows, seed):
    np.random.seed(seed)
    return np.array([np.random.choice(pi.shape[1], p=pi[i] / pi[i].sum()) for i in rows])


@pytest.mark.parametrize("B0,B1,d,reg", [(512, 512, 5, 2.0), (300, 417, 3, 1.0), (1024, 1024, 50, 8.0)])
def test_rows_dense_equals_numpy_choice_on_the_plan_rows(dev, B0, B1, d, reg):
    import cfm_amd.optimal_transport as ot
    g = torch.Generator().manual_seed(B0 + d)
    x0 = torch.randn(B0, d, generator=g).to(dev); x1 = (torch.randn(B1, d, generator=g) * 0.8 + 0.3).to(dev)
    M = ot.cost_matrix(x0, x1)
    r = ot.sinkhorn_log(M, reg)
    pi = ot.sinkhorn_plan(r).cpu().numpy()
    rows = np.random.RandomState(3).randint(0, B0, size=700)            # repeated and out-of-order rows
    want = _host_rows(pi, rows, 11)
    np.random.seed(11)
    u = torch.from_numpy(np.random.random_sample(len(rows))).to(dev)
    got = ot.sample_rows_dense(r, torch.from_numpy(rows).to(dev), u).cpu().numpy()
    assert (got != want).sum() <= 1          # a draw within rounding of a cdf step may land on the neighbour
    assert got.min() >= 0 and got.max() < B1


def test_rows_pi_equals_numpy_choice_and_handles_sparse_rows(dev):
    import cfm_amd.optimal_transport as ot
    rs = np.random.RandomState(5)
    pi = rs.rand(200, 333)
    pi[pi < 0.7] = 0.0                       # sparse rows: long runs of zero mass
    pi[17] = 0.0; pi[17, 332] = 2.5          # all the mass in the last column
    pi[18] = 0.0; pi[18, 0] = 1e-300         # ... in the first, tiny
    rows = rs.randint(0, 200, size=500); rows[:2] = (17, 18)
    want = _host_rows(pi, rows, 2)
    np.random.seed(2)
    u = torch.from_numpy(np.random.random_sample(len(rows))).to(dev)
    got = ot.sample_rows_pi(torch.from_numpy(pi).to(dev), torch.from_numpy(rows).to(dev), u).cpu().numpy()
    assert np.array_equal(got, want)


def test_trajectory_of_unbalanced_plans_matches_the_host_chain(dev):
    """plan-kind slices (fp64 plans on the device) through sample_trajectory: the reference's loop on get_map()'s plans"""
    from cfm_amd.optimal_transport import OTPlanSampler
    g = torch.Generator().manual_seed(21)
    X = torch.randn(96, 4, 2, generator=g)
    s = OTPlanSampler(method="unbalanced", reg=1.0, reg_m=2.0)
    np.random.seed(4)
    out = s.sample_trajectory(X)
    np.random.seed(4)
    idx = [np.arange(96)]
    for t in range(3):
        pi = s.get_map(X[:, t], X[:, t + 1])
        idx.append(np.array([np.random.choice(96, p=pi[i] / pi[i].sum()) for i in idx[-1]]))
    ref = np.stack([X[:, t].numpy()[idx[t]] for t in range(4)], axis=1)
    assert np.array_equal(out, ref)


def test_trajectory_diagnostics_of_a_plan_without_mass_or_with_nans(dev, monkeypatch):
    """sample_trajectory goes through get_map's diagnostics per slice like the reference (ref:88-96 via :233): a plan
    without mass reverts to the uniform plan (with the warning), a non-finite one is reported and raises what
    np.random.choice raises, a visited row without mass raises too — never silent last-column indices."""
    import warnings
    from cfm_amd.optimal_transport import OTPlanSampler
    g = torch.Generator().manual_seed(8)
    X = torch.randn(64, 3, 2, generator=g)
    s = OTPlanSampler(method="unbalanced", reg=1.0, reg_m=2.0)
    real = s._solve_many

    def patched(kind):
        def solve_many(pairs, workers=3):
            sols = real(pairs, workers)
            k, plan, M = sols[1]
            plan = plan.clone()
            if kind == "zero":
                plan.zero_()
            elif kind == "nan":
                plan[3, 5] = float("nan")
            else:                                    # half of the rows without mass, the plan as a whole fine
                plan[:] = 1.0 / plan.numel(); plan[:32] = 0.0
            sols[1] = (k, plan, M)
            return sols
        return solve_many

    monkeypatch.setattr(s, "_solve_many", patched("zero"))
    np.random.seed(0)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = s.sample_trajectory(X)
    assert out.shape == (64, 3, 2) and any("reverting to uniform plan" in str(x.message) for x in w)
    # the uniform slice draws searchsorted(cumsum(1 / B), u): column floor(u * B) up to rounding — spread over the columns
    monkeypatch.setattr(s, "_solve_many", patched("nan"))
    with pytest.raises(ValueError, match="probabilities contain NaN"):
        s.sample_trajectory(X)
    monkeypatch.setattr(s, "_solve_many", patched("row"))
    np.random.seed(1)
    try:
        s.sample_trajectory(X)           # raises iff a massless row of slice 1 is visited by the chain (64 draws over 64 columns, half of them massless rows: certain)
        visited = False
    except ValueError as e:
        visited = "probabilities contain NaN" in str(e)
    assert visited
