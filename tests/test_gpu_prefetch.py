"""GPU: cfm_amd.prefetch.CouplingPrefetcher hands back exactly what the synchronous call computes
(same RNG stream, same pairs, same xt / ut), with one worker and with several in flight."""
import collections

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from cfm_amd import _lib
    _lib.load()
    return _lib.require_gpu()


def _batches(n, B, d, dev):
    g = torch.Generator().manual_seed(123)
    return [(torch.randn(B, d, generator=g).to(dev), (torch.randn(B, d, generator=g) * 0.7 + 0.5).to(dev))
            for _ in range(n)]


def test_one_worker_equals_synchronous(dev):
    from cfm_amd.conditional_flow_matching import ExactOptimalTransportConditionalFlowMatcher
    from cfm_amd.prefetch import CouplingPrefetcher
    fm = ExactOptimalTransportConditionalFlowMatcher(sigma=0.1)
    data = _batches(5, 256, 8, dev)
    torch.manual_seed(7); np.random.seed(7)
    ref = [fm.sample_location_and_conditional_flow(a, b) for a, b in data]
    torch.manual_seed(7); np.random.seed(7)
    pre = CouplingPrefetcher(fm, dev, workers=1)
    handles = [pre.submit(a, b) for a, b in data]          # one worker: executed in submission order
    got = [h.result() for h in handles]
    pre.close()
    torch.cuda.synchronize()
    for r, g in zip(ref, got):
        for x, y in zip(r, g):
            assert torch.equal(x, y)


def test_several_in_flight_with_draws_on_the_caller(dev):
    import cfm_amd.optimal_transport as ot
    from cfm_amd.conditional_flow_matching import ExactOptimalTransportConditionalFlowMatcher
    from cfm_amd.prefetch import CouplingPrefetcher
    fm = ExactOptimalTransportConditionalFlowMatcher(sigma=0.0)
    B = 512
    data = _batches(9, B, 16, dev)

    def draw():
        return np.random.random_sample(B), torch.rand(B)

    def couple(x0, x1, drawn):
        u, t = drawn
        M = ot.cost_matrix(x0, x1)
        perm = ot.assign_exact(M)
        i, j = ot.sample_perm(perm, torch.from_numpy(u).to(dev), B)
        return fm._sample(x0, x1, t.type_as(x0), False, idx=(i, j))

    torch.manual_seed(3); np.random.seed(3)
    ref = [couple(a, b, draw()) for a, b in data]
    torch.manual_seed(3); np.random.seed(3)
    pre = CouplingPrefetcher(fm, dev, workers=3)
    inflight = collections.deque(pre.submit(a, b, hook=couple, draw=draw) for a, b in data[:3])
    got = []
    for k in range(len(data)):
        got.append(inflight.popleft().result())
        if k + 3 < len(data):
            inflight.append(pre.submit(*data[k + 3], hook=couple, draw=draw))
    pre.close()
    torch.cuda.synchronize()
    for r, g in zip(ref, got):
        for x, y in zip(r, g):
            assert torch.equal(x, y)


def test_grouped_couplings_equal_the_sequential_ones(dev):
    """submit_group: G minibatches coupled together (assign_exact_batch: one chain of launches for the G assignment
    problems) give, minibatch by minibatch, exactly what coupling them one after the other gives."""
    import cfm_amd.optimal_transport as ot
    from cfm_amd.conditional_flow_matching import ExactOptimalTransportConditionalFlowMatcher
    from cfm_amd.prefetch import CouplingPrefetcher
    fm = ExactOptimalTransportConditionalFlowMatcher(sigma=0.0)
    B = 640
    data = _batches(10, B, 24, dev)

    def draw():
        return np.random.random_sample(B), torch.rand(B)

    def couple(x0, x1, drawn):
        u, t = drawn
        perm = ot.assign_exact(ot.cost_matrix(x0, x1))
        i, j = ot.sample_perm(perm, torch.from_numpy(u).to(dev), B)
        return fm._sample(x0, x1, t.type_as(x0), False, idx=(i, j))

    def couple_group(batches, drawn):
        perms = ot.assign_exact_batch([ot.cost_matrix(a, b) for a, b in batches])
        out = []
        for (a, b), (u, t), perm in zip(batches, drawn, perms):
            i, j = ot.sample_perm(perm, torch.from_numpy(u).to(dev), B)
            out.append(fm._sample(a, b, t.type_as(a), False, idx=(i, j)))
        return out

    torch.manual_seed(5); np.random.seed(5)
    ref = [couple(a, b, draw()) for a, b in data]
    torch.manual_seed(5); np.random.seed(5)
    pre = CouplingPrefetcher(fm, dev, workers=2)
    handles = [pre.submit_group(data[0:4], couple_group, draw), pre.submit_group(data[4:8], couple_group, draw),
               pre.submit_group(data[8:10], couple_group, draw)]
    got = [r for h in handles for r in h.result()]
    pre.close()
    torch.cuda.synchronize()
    assert len(got) == len(ref)
    for r, g in zip(ref, got):
        for x, y in zip(r, g):
            assert torch.equal(x, y)


def test_group_method_of_the_flow_matcher_equals_one_call_per_pair(dev):
    """FM.sample_location_and_conditional_flow_group: same tensors, same host-RNG consumption as one
    sample_location_and_conditional_flow call per pair (exact and entropic couplings)."""
    from cfm_amd.conditional_flow_matching import (ExactOptimalTransportConditionalFlowMatcher,
                                                   SchrodingerBridgeConditionalFlowMatcher)
    data = _batches(5, 384, 8, dev)
    for fm in (ExactOptimalTransportConditionalFlowMatcher(sigma=0.1),
               SchrodingerBridgeConditionalFlowMatcher(sigma=1.0, ot_method="sinkhorn")):
        torch.manual_seed(11); np.random.seed(11); torch.cuda.manual_seed(11)
        ref = [fm.sample_location_and_conditional_flow(a, b) for a, b in data]
        torch.manual_seed(11); np.random.seed(11); torch.cuda.manual_seed(11)
        got = fm.sample_location_and_conditional_flow_group(data)
        assert len(got) == len(ref)
        for r, g in zip(ref, got):
            for x, y in zip(r, g):
                assert torch.equal(x, y)
