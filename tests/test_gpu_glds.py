"""GPU: the cost matrix on the direct-to-LDS tile engine (csrc/gemm_glds.h, the only matrix-core cost path since round
5) at the shapes that stress its PADDED operand layout — ragged tile edges in both directions, a K tail (d not a
multiple of the 32-float stage), d not a multiple of 4 (unaligned rows), duplicated points (the cancellation path) —
against the float64 oracle, entry by entry, and against the direct-difference kernels (the scratch-free entry point)."""
import numpy as np
import pytest
import torch

import cfm_oracle as oracle

pytestmark = pytest.mark.gpu

SHAPES = ((1024, 1024, 784), (1000, 777, 100), (512, 300, 64), (256, 256, 788), (640, 512, 192), (300, 1025, 77),
          (257, 259, 65), (4096, 4096, 784))


@pytest.mark.parametrize("B0,B1,d", SHAPES)
def test_glds_cost_matrix_vs_f64_oracle(B0, B1, d):
    import cfm_amd.optimal_transport as ot
    from cfm_amd import _lib
    dev = _lib.require_gpu()
    lib = _lib.load()
    g = torch.Generator().manual_seed(B0 + 7 * B1 + 13 * d)
    a = torch.randn(B0, d, generator=g); b = torch.randn(B1, d, generator=g) * 0.5 + 0.2
    b[:5] = a[:5]
    ad, bd = a.to(dev), b.to(dev)
    M = ot.cost_matrix(ad, bd)
    Mh = M.cpu().numpy()
    assert Mh.shape == (B0, B1) and (Mh >= 0).all()
    assert float(M[:5, :5].diagonal().abs().max()) == 0.0          # duplicates: the recomputed entries are exact zeros
    if B0 * B1 <= 2 ** 21:
        ref = oracle.sqeuclid_cost_f64(a.numpy(), b.numpy())
        rows = slice(None)
    else:                                                          # full C3 size: a band of rows at each end + the middle
        rows = np.r_[0:64, B0 // 2 - 32:B0 // 2 + 32, B0 - 64:B0]
        ref = oracle.sqeuclid_cost_f64(a.numpy()[rows], b.numpy())
    tol = 4e-7 * max(4.0, np.sqrt(d))
    err = np.abs(Mh[rows] - ref) / np.maximum(ref, 1e-30)
    assert err.max() < tol, (err.max(), np.unravel_index(err.argmax(), err.shape))
    # the scratch-free entry point (direct-difference kernels) agrees to the sum of the two bounds
    Md = torch.empty_like(M)
    _lib.check(lib.cfm_sqeuclid_cost_f32(_lib.ptr(ad), _lib.ptr(bd), B0, B1, d, _lib.ptr(Md), None, _lib.stream_ptr()),
               "cfm_sqeuclid_cost_f32")
    assert (np.abs(Mh[rows] - Md.cpu().numpy()[rows]) <= 2 * tol * np.maximum(ref, 1e-30)).all()


def test_glds_cost_matrix_is_deterministic_and_workspace_is_private():
    """Two calls give the same bits; so does a cloud whose base pointer is not 16-byte aligned (the padded copies take any alignment)."""
    import cfm_amd.optimal_transport as ot
    from cfm_amd import _lib
    dev = _lib.require_gpu()
    g = torch.Generator().manual_seed(3)
    a = torch.randn(700, 130, generator=g).to(dev); b = torch.randn(513, 130, generator=g).to(dev)
    M1 = ot.cost_matrix(a, b).clone()
    M2 = ot.cost_matrix(a, b)
    assert torch.equal(M1, M2)
    flat = torch.zeros(700 * 130 + 1, device=dev)             # a base pointer off the 16-byte grid (4-byte aligned only)
    a2 = flat[1:].view(700, 130)
    a2.copy_(a)
    assert a2.data_ptr() % 16 != 0 and a2.is_contiguous()
    assert torch.equal(ot.cost_matrix(a2, b), M1)
