"""CPU, build container only: the oracle against the LIVE reference imported from
/root/reference (skipped where the reference tree is absent, e.g. on the GPU box)."""
import numpy as np
import pytest
import torch

import cfm_oracle as oracle
import ref_import

pytestmark = pytest.mark.skipif(not ref_import.available(), reason="reference tree not present")


@pytest.fixture(scope="module")
def ref():
    return ref_import.import_reference()


@pytest.mark.parametrize("B,shape", [(64, (2,)), (128, (2, 2, 2)), (100, (5,))])
def test_exact_ot_step_matches_reference(ref, B, shape):
    cfm, ot = ref
    g = torch.Generator().manual_seed(B)
    x0, x1 = torch.randn(B, *shape, generator=g), torch.randn(B, *shape, generator=g)
    fm = cfm.ExactOptimalTransportConditionalFlowMatcher(sigma=0.25)
    torch.manual_seed(3); np.random.seed(3)
    t, xt, ut, eps = fm.sample_location_and_conditional_flow(x0, x1, return_noise=True)
    torch.manual_seed(3); np.random.seed(3)
    t2, xt2, ut2, eps2, _ = oracle.ot_cfm_step(x0, x1, sigma=0.25)
    assert torch.equal(t, t2) and torch.equal(eps, eps2)
    assert torch.all(xt.eq(xt2)) and torch.all(ut.eq(ut2))


@pytest.mark.parametrize("method,cls,kw", [
    ("icfm", "ConditionalFlowMatcher", {}),
    ("target", "TargetConditionalFlowMatcher", {}),
    ("vp", "VariancePreservingConditionalFlowMatcher", {}),
])
@pytest.mark.parametrize("sigma", [0.0, 5e-4, 0.5, 1.5, 0, 1])
def test_closed_forms_match_reference(ref, method, cls, kw, sigma):
    cfm, _ = ref
    x0, x1 = torch.randn(32, 3, 4), torch.randn(32, 3, 4)
    fm = getattr(cfm, cls)(sigma=sigma, **kw)
    torch.manual_seed(9)
    t, xt, ut, eps = fm.sample_location_and_conditional_flow(x0, x1, return_noise=True)
    xt2, ut2 = oracle.xt_ut(method, x0, x1, t, eps, sigma)
    assert torch.all(xt.eq(xt2)) and torch.all(ut.eq(ut2))


def test_reference_own_tests_pass_with_standin(ref):
    """The stand-in is faithful enough for the reference's own assertions."""
    cfm, ot = ref
    s = ot.OTPlanSampler(method="exact")
    rng = np.random.default_rng(0)
    pm = rng.permutation(np.eye(64), axis=1)
    ii, jj = s.sample_map(pm, batch_size=64, replace=False)
    rec = np.zeros((64, 64)); rec[ii, jj] = 1
    assert np.array_equal(rec, pm)
    with pytest.raises(ValueError):
        ot.OTPlanSampler(method="nope")
    with pytest.raises(ValueError):
        cfm.SchrodingerBridgeConditionalFlowMatcher(sigma=0.0)
