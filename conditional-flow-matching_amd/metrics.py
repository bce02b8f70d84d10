"""Distribution distances of the evaluation loop on the device — counterpart of
runner/src/models/components/distribution_distances.py:11-74 (+ mmd.py): per time point
1-/2-Wasserstein (exact OT: the HIP assignment solver), linear / polynomial / mixture-RBF MMD, and
mean / median errors; same names, same order, same return convention.  The O(B^2) pieces never
leave the GPU: W1 / W2 reuse the cost + assignment kernels, the RBF MMD sums its three kernel
matrices on the fly from squared-distance matrices (``cfm_rbf_mix_sum_f32``)."""
import math
from typing import Union

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr
from .optimal_transport import cost_matrix, wasserstein


def compute_distances(pred, true):
    """mse, sqrt(mse), mae between two vectors (distribution_distances.py:11-16)."""
    mse = torch.nn.functional.mse_loss(pred, true).item()
    return mse, math.sqrt(mse), torch.mean(torch.abs(pred - true)).item()


def linear_mmd2(f_of_X, f_of_Y):
    """Linear-time MMD with a linear kernel (mmd.py:17-21)."""
    delta = f_of_X - f_of_Y
    return torch.mean((delta[:-1] * delta[1:]).sum(1))


def poly_mmd2(f_of_X, f_of_Y, d=2, alpha=1.0, c=2.0):
    """Linear-time MMD with the kernel (alpha <x, y> + c)^d (mmd.py:28-41)."""
    def k(a, b):
        return torch.mean((alpha * (a[:-1] * b[1:]).sum(1) + c).pow(d))
    return k(f_of_X, f_of_X) + k(f_of_Y, f_of_Y) - k(f_of_X, f_of_Y) - k(f_of_Y, f_of_X)


def mix_rbf_mmd2(X, Y, sigma_list, biased=True):
    """Biased (V-statistic) MMD^2 with a mixture of RBF kernels (mmd.py:43-63,80-110 with
    const_diagonal=False): (sum K_XX + sum K_YY - 2 sum K_XY) / m^2, kernel sums on the device."""
    if not biased:
        raise NotImplementedError("mix_rbf_mmd2: only the biased estimator the reference calls is built")
    assert X.shape[0] == Y.shape[0]
    lib = _lib.load()
    dev = _lib.require_gpu()
    a, b = _lib.to_dev_f32(X.reshape(X.shape[0], -1), dev), _lib.to_dev_f32(Y.reshape(Y.shape[0], -1), dev)
    m = a.shape[0]
    gam = torch.tensor([1.0 / (2.0 * s * s) for s in sigma_list], dtype=torch.float32, device=dev)
    sums = torch.zeros(3, dtype=torch.float64, device=dev)
    for q, (p0, p1) in enumerate(((a, a), (b, b), (a, b))):
        D = cost_matrix(p0, p1, squared=True, matrix_cores=False)
        check(lib.cfm_rbf_mix_sum_f32(ptr(D), D.numel(), ptr(gam), len(sigma_list),
                                      ctypes_ptr(sums, q), stream_ptr()), "cfm_rbf_mix_sum_f32")
    s = sums.cpu()
    return torch.tensor(float((s[0] + s[1] - 2.0 * s[2]) / (m * m)))


def ctypes_ptr(t, index):
    import ctypes
    return ctypes.c_void_p(t.data_ptr() + index * t.element_size())


def compute_distribution_distances(pred: Union[torch.Tensor, list], true: Union[torch.Tensor, list]):
    """names, values — distribution_distances.py:19-74 (jagged ``true`` / ``pred`` lists drop the MMDs)."""
    NAMES = ["1-Wasserstein", "2-Wasserstein", "Linear_MMD", "Poly_MMD", "RBF_MMD", "Mean_MSE", "Mean_L2",
             "Mean_L1", "Median_MSE", "Median_L2", "Median_L1"]
    is_jagged, pred_is_jagged = isinstance(true, list), isinstance(pred, list)
    dists, to_return, names = [], [], []
    filtered = [n for n in NAMES if not is_jagged or not n.endswith("MMD")]
    ts = len(pred) if pred_is_jagged else pred.shape[1]
    for t in np.arange(ts):
        a = pred[t] if pred_is_jagged else pred[:, t, :]
        b = true[t] if is_jagged else true[:, t, :]
        w1 = wasserstein(a, b, power=1)
        w2 = wasserstein(a, b, power=2)
        mean_d = compute_distances(torch.mean(a, dim=0), torch.mean(b, dim=0))
        med_d = compute_distances(torch.median(a, dim=0)[0], torch.median(b, dim=0)[0])
        if pred_is_jagged or is_jagged:
            dists.append((w1, w2, *mean_d, *med_d))
        else:
            mmds = (linear_mmd2(a, b).item(), poly_mmd2(a, b, d=2, alpha=1.0, c=2.0).item(),
                    mix_rbf_mmd2(a, b, sigma_list=[0.01, 0.1, 1, 10, 100]).item())
            dists.append((w1, w2, *mmds, *mean_d, *med_d))
        if ts > 1:
            names.extend([f"t{t+1}/{n}" for n in filtered])
            to_return.extend(dists[-1])
    to_return.extend(np.array(dists).mean(axis=0))
    names.extend(filtered)
    return names, to_return
