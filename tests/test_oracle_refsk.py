"""CPU: the Sinkhorn restatements of oracle/ pinned to plans computed by the REFERENCE-HELD code itself
(runner/src/models/components/sinkhorn_knopp_unbalanced.py, pure NumPy, imported unmodified by
tests/golden/make_golden.py -> tests/golden/refsk_cases.npz; VERDICT r2 Next #4):
  * unbalanced: the oracle's restatement of POT's loop, at its fixed point, against the reference's plans;
  * balanced: reg_m_1 = reg_m_2 = 1e12 makes the reference loop's fixed point the balanced entropic plan — the
    float64 log-domain oracle (NumPy and C) at convergence must give the same plan (<= 1e-6 relative).
When /root/reference is present the fixture itself is re-derived and compared."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import cfm_oracle as oracle  # noqa: E402
import ref_import  # noqa: E402

UNB = ((0.5, 1.0), (1.0, 0.2), (0.3, 5.0))
BAL = (("bal", "M", 1.0), ("bal", "M", 2.0), ("bal", "M", 5.0), ("bal8g", "M8", 2.0), ("bal8g", "M8", 5.0))


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "refsk_cases.npz"))


def plan_close(P, ref, rtol):
    """max-norm relative AND element-wise relative on every entry that carries mass above 1e-9 of the largest."""
    P, ref = np.asarray(P, np.float64), np.asarray(ref, np.float64)
    big = ref.max()
    assert np.abs(P - ref).max() <= rtol * big, (np.abs(P - ref).max(), big)
    m = ref > 1e-9 * big
    assert np.abs(P[m] / ref[m] - 1.0).max() <= 10 * rtol


def test_docstring_kat_through_the_real_function(gold):
    np.testing.assert_allclose(gold["kat"], [[0.51122814, 0.18807032], [0.18807032, 0.51122814]], atol=5e-7)   # default stopThr = 1e-6


@pytest.mark.parametrize("reg,reg_m", UNB)
def test_oracle_unbalanced_fixed_point_equals_reference_held_plan(gold, reg, reg_m):
    P = oracle.sinkhorn_knopp_unbalanced(gold["M"], reg, reg_m, numItermax=200000, stopThr=1e-15)
    plan_close(P, gold[f"ub_{reg}_{reg_m}"], 1e-9)


@pytest.mark.parametrize("name,mkey,reg", BAL)
def test_oracle_log_domain_converged_plan_equals_reference_balanced_limit(gold, name, mkey, reg):
    M = gold[mkey]
    ref = gold[f"{name}_{reg}"]
    # the reference plan is balanced to ~reg / reg_m = 1e-12
    n = M.shape[0]
    assert np.abs(ref.sum(1) - 1.0 / n).max() <= 1e-9 / n and np.abs(ref.sum(0) - 1.0 / n).max() <= 1e-9 / n
    u, v, it, err = oracle.sinkhorn_log(M, reg, numItermax=100000, stopThr=1e-13)
    assert err <= 1e-12
    P = np.exp(-np.asarray(M, np.float64) / reg + u[:, None] + v[None, :])
    plan_close(P, ref, 1e-6)
    # POT's own stopping rule (stopThr = 1e-9, check every 10) already lands within 1e-6
    u, v, it, err = oracle.sinkhorn_log(M, reg)
    assert it < 1000
    plan_close(np.exp(-np.asarray(M, np.float64) / reg + u[:, None] + v[None, :]), ref, 1e-6)


@pytest.mark.parametrize("name,mkey,reg", BAL[:2] + BAL[3:4])
def test_c_oracle_converged_plan_equals_reference_balanced_limit(gold, name, mkey, reg):
    import sinkhorn_c
    M = np.ascontiguousarray(gold[mkey], dtype=np.float32)
    u, v, it, err = sinkhorn_c.sinkhorn_log(M, reg, numItermax=100000, stopThr=1e-13)
    P = np.exp(-M.astype(np.float64) / reg + u[:, None] + v[None, :])
    plan_close(P, gold[f"{name}_{reg}"], 1e-6)


def test_knopp_restatement_equals_reference_balanced_limit(gold):
    """POT's default `ot.sinkhorn` (kernel-space Knopp, what optimal_transport.py:51 binds) where it is numerically alive."""
    P = oracle.sinkhorn_knopp(gold["M"], 2.0, numItermax=100000, stopThr=1e-14)
    plan_close(P, gold["bal_2.0"], 1e-6)


@pytest.mark.skipif(not ref_import.available(), reason="reference tree not present (GPU box)")
def test_fixture_is_what_the_reference_code_returns_here(gold):
    import make_golden
    sk = ref_import.import_runner_sinkhorn()
    x0, x1, y0, y1 = make_golden.refsk_inputs()
    np.testing.assert_array_equal(oracle.ref_cost_f32(x0, x1), gold["M"])
    P = sk.sinkhorn_knopp_unbalanced([], [], gold["M"], 0.5, 1.0, 1.0, numItermax=200000, stopThr=1e-15)
    np.testing.assert_allclose(P, gold["ub_0.5_1.0"], rtol=1e-12, atol=0)
    P = sk.sinkhorn_knopp_unbalanced([], [], gold["M8"], 5.0, 1e12, 1e12, numItermax=200000, stopThr=1e-15)
    np.testing.assert_allclose(P, gold["bal8g_5.0"], rtol=1e-12, atol=0)
