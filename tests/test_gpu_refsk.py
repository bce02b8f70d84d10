"""GPU: the HIP Sinkhorn solvers at convergence against plans computed by the REFERENCE-HELD code
(runner/src/models/components/sinkhorn_knopp_unbalanced.py run unmodified -> tests/golden/refsk_cases.npz):
  * cfm_sinkhorn_log_f32 (matrix streaming, log domain) on the fixture's own fp32 matrix vs the balanced limit
    (reg_m = 1e12) of the reference loop: plan <= 1e-6 relative;
  * cfm_sinkhorn_log_points_f32 (variant B, cost recomputed from the coordinates) vs the same plans, end to end
    (its fp32 cost chain rounds differently from torch.cdist ** 2: <= 2e-5);
  * cfm_unbalanced_sinkhorn_f64 at its fixed point vs the reference loop's unbalanced plans: <= 1e-8."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BAL = (("bal", "M", "x", 1.0), ("bal", "M", "x", 2.0), ("bal", "M", "x", 5.0), ("bal8g", "M8", "y", 2.0), ("bal8g", "M8", "y", 5.0))


@pytest.fixture(scope="module")
def dev():
    from cfm_amd import _lib
    _lib.load()
    return _lib.require_gpu()


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "refsk_cases.npz"))


def plan_close(P, ref, rtol):
    big = ref.max()
    assert np.abs(P - ref).max() <= rtol * big, (np.abs(P - ref).max(), big)
    m = ref > 1e-9 * big
    assert np.abs(P[m] / ref[m] - 1.0).max() <= 10 * rtol, np.abs(P[m] / ref[m] - 1.0).max()


@pytest.mark.parametrize("name,mkey,xkey,reg", BAL)
def test_log_domain_solver_converged_plan_equals_reference_balanced_limit(dev, gold, name, mkey, xkey, reg):
    import cfm_amd.optimal_transport as ot
    M = torch.from_numpy(np.ascontiguousarray(gold[mkey])).to(dev)
    ref = gold[f"{name}_{reg}"]
    r = ot.sinkhorn_log(M, reg)                        # POT's loop: stopThr = 1e-9, numItermax = 1000
    assert int(r.iters.cpu()) < 1000 and float(r.err.cpu()) < 1e-9
    plan_close(ot.sinkhorn_plan(r).cpu().numpy(), ref, 1e-6)
    r = ot.sinkhorn_log(M, reg, max_iter=20000, stop_thr=1e-12)
    plan_close(ot.sinkhorn_plan(r).cpu().numpy(), ref, 1e-6)
    # potentials: f_i + g_j reproduces reg * log(plan) + M on the support (the 1e-5 north-star bound)
    f, g = r.f.cpu().numpy().astype(np.float64), r.g.cpu().numpy().astype(np.float64)
    lhs = f[:, None] + g[None, :]
    rhs = reg * np.log(ref) + gold[mkey].astype(np.float64)
    assert np.abs(lhs - rhs).max() <= 1e-5 * np.abs(rhs).max()


@pytest.mark.parametrize("name,mkey,xkey,reg", BAL)
def test_points_variant_converged_plan_equals_reference_balanced_limit(dev, gold, name, mkey, xkey, reg):
    import cfm_amd.optimal_transport as ot
    a = torch.from_numpy(gold[xkey + "0"]).to(dev); b = torch.from_numpy(gold[xkey + "1"]).to(dev)
    M = ot.cost_matrix(a, b)
    r = ot.sinkhorn_log_points(a, b, M, reg, max_iter=20000, stop_thr=1e-12)
    plan_close(ot.sinkhorn_plan(r).cpu().numpy(), gold[f"{name}_{reg}"], 2e-5)
    # and through the sampler's own entry point (OTPlanSampler picks variant B for d <= 8)
    pi = ot.OTPlanSampler(method="sinkhorn", reg=reg).get_map(a, b)
    plan_close(pi, gold[f"{name}_{reg}"], 2e-5)


@pytest.mark.parametrize("reg,reg_m", [(0.5, 1.0), (1.0, 0.2), (0.3, 5.0)])
def test_unbalanced_solver_fixed_point_equals_reference_held_plan(dev, gold, reg, reg_m):
    import cfm_amd.optimal_transport as ot
    M = torch.from_numpy(np.ascontiguousarray(gold["M"])).to(dev)
    P, info = ot.unbalanced_plan(M, reg, reg_m, max_iter=200000, stop_thr=1e-15)
    plan_close(P.cpu().numpy(), gold[f"ub_{reg}_{reg_m}"], 1e-8)
