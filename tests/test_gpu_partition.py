"""GPU: chip partition (cfm_amd.streams) — CU-masked streams give the same results as ordinary streams, the exact
solver is redirected onto the thread's solver stream, and a partitioned prefetcher hands back what the synchronous
call computes.  (The partition changes WHERE kernels run, never what they compute.)"""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from cfm_amd import _lib
    _lib.load()
    return _lib.require_gpu()


def test_mask_validation(dev):
    from cfm_amd import _lib
    lib = _lib.load()
    out = ctypes.c_void_p(0)
    words = (ctypes.c_uint32 * 8)(*([0] * 8))
    assert lib.cfm_stream_create_cu_mask(words, 8, ctypes.byref(out)) == -1          # no CU at all
    words[0] = 0x01010101                                                            # CUs of XCD 0 only
    assert lib.cfm_stream_create_cu_mask(words, 8, ctypes.byref(out)) == -1          # seven XCDs without a CU
    assert lib.cfm_stream_create_cu_mask(None, 8, ctypes.byref(out)) == -1
    with pytest.raises(ValueError):
        from cfm_amd.streams import ChipPartition
        ChipPartition(dev, solver_cus_per_xcd=0)


def test_solver_on_masked_stream_equals_default(dev):
    import cfm_amd.optimal_transport as ot
    from cfm_amd.streams import ChipPartition, solver_stream
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(1024, 32, generator=g).to(dev); x1 = (torch.randn(1024, 32, generator=g) * 0.7 + 0.3).to(dev)
    M = ot.cost_matrix(x0, x1)
    ref = ot.assign_exact(M).cpu()
    part = ChipPartition(dev, solver_cus_per_xcd=4)
    assert part.solver_cus == 32 and part.dense_cus == part.ncu - 32
    s_solver, s_dense = part.solver_stream(), part.dense_stream()
    with torch.cuda.stream(s_dense):
        M2 = ot.cost_matrix(x0, x1)                      # dense product on the dense subset
        with solver_stream(s_solver):
            perm = ot.assign_exact(M2)                   # redirected: runs on s_solver, ordered after M2
            perms = ot.assign_exact_batch([M2, M])
        i, j = ot.sample_perm(perm, torch.from_numpy(np.linspace(0, 1, 1024, endpoint=False)).to(dev), 1024)
    torch.cuda.synchronize()
    assert torch.equal(M, M2)
    assert torch.equal(perm.cpu(), ref) and torch.equal(perms[0].cpu(), ref) and torch.equal(perms[1].cpu(), ref)
    assert torch.equal(j.cpu(), ref.long()[i.cpu()])
    part.close()


def test_partitioned_prefetcher_equals_synchronous(dev):
    from cfm_amd.conditional_flow_matching import ExactOptimalTransportConditionalFlowMatcher
    from cfm_amd.prefetch import CouplingPrefetcher
    from cfm_amd.streams import ChipPartition
    fm = ExactOptimalTransportConditionalFlowMatcher(sigma=0.1)
    g = torch.Generator().manual_seed(11)
    data = [(torch.randn(512, 16, generator=g).to(dev), (torch.randn(512, 16, generator=g) * 0.5 + 1).to(dev))
            for _ in range(4)]
    torch.manual_seed(3); np.random.seed(3)
    ref = [fm.sample_location_and_conditional_flow(a, b) for a, b in data]
    torch.manual_seed(3); np.random.seed(3)
    part = ChipPartition(dev, solver_cus_per_xcd=6)
    pre = CouplingPrefetcher(fm, dev, workers=1, partition=part)
    main = part.dense_stream()
    with torch.cuda.stream(main):
        got = [h.result() for h in [pre.submit(a, b) for a, b in data]]
        sums = [float(x.sum()) for x in got[-1]]         # consumed on the dense main stream
    pre.close(); torch.cuda.synchronize()
    for r, q in zip(ref, got):
        for x, y in zip(r, q):
            assert torch.equal(x, y)
    assert np.allclose(sums, [float(x.sum()) for x in ref[-1]])
    part.close()


@pytest.mark.parametrize("cus_per_xcd", [4, 2])
def test_c3_sized_solves_on_a_masked_stream_keep_their_speed(dev, cus_per_xcd):
    """asg_auction needs its whole grid resident at once; on a CU-masked solver stream (32 / 16 CUs) a lone
    n = 4096 solve used to launch 64 workgroups of which half could not start before the others left — counted as "all
    rows unmatched" for the grace (~40 ms), then left out (> 64 free rows: the dense fallback).  The grid is now capped at
    the CUs the stream may use (hipExtStreamGetCUMask): same permutation, no fallback, and nowhere near the 40 ms."""
    import time
    import cfm_amd.optimal_transport as ot
    from cfm_amd import _lib
    from cfm_amd.streams import ChipPartition, solver_stream
    lib = _lib.load()
    g = torch.Generator().manual_seed(41)
    Ms = []
    for _ in range(2):
        x0 = torch.randn(4096, 64, generator=g).to(dev); x1 = (torch.randn(4096, 64, generator=g) * 0.7 + 0.3).to(dev)
        Ms.append(ot.cost_matrix(x0, x1))
    ref = [ot.assign_exact(M).cpu() for M in Ms]
    fb0 = (ctypes.c_int * 2)(); lib.cfm_assign_debug_fallback(fb0)
    part = ChipPartition(dev, solver_cus_per_xcd=cus_per_xcd)
    s_solver, s_dense = part.solver_stream(), part.dense_stream()
    with torch.cuda.stream(s_dense):
        with solver_stream(s_solver):
            ot.assign_exact(Ms[0]); ot.assign_exact_batch(Ms)          # one-time costs (workspaces, launch programs)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            lone = [ot.assign_exact(M).cpu() for M in Ms]
            t_lone = (time.perf_counter() - t0) / len(Ms)
            t0 = time.perf_counter()
            batch = [p.cpu() for p in ot.assign_exact_batch(Ms)]
            t_batch = time.perf_counter() - t0
    torch.cuda.synchronize()
    fb1 = (ctypes.c_int * 2)(); lib.cfm_assign_debug_fallback(fb1)
    for r, a, b in zip(ref, lone, batch):
        assert torch.equal(r, a) and torch.equal(r, b)
    assert fb1[0] == fb0[0], "a masked-stream solve fell back to the dense machine"
    assert t_lone < 0.03 and t_batch < 0.03, (t_lone, t_batch)      # ~2-8 ms on 16-32 CUs; the stall was >= 40 ms
    part.close()
