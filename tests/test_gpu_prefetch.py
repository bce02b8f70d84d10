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
