"""Stand-in for POT (``import ot``), which is not installed and not installable in this
image.  It exists ONLY so the unmodified reference (/root/reference/torchcfm) can be imported
to generate golden vectors and to time the reference's own wrapper code; it is test
infrastructure like the rest of oracle/.

emd/emd2: uniform equal-size marginals only -> permutation plan from SciPy's LSAP (the solver
the reference itself uses at torchcfm/optimal_transport.py:170,179).  sinkhorn/sinkhorn2:
POT's default Sinkhorn-Knopp restated.  unbalanced / partial: the loops of
cfm_oracle.sinkhorn_knopp_unbalanced / entropic_partial_wasserstein.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cfm_oracle as _o  # noqa: E402


def unif(n, type_as=None):
    return np.ones((n,)) / n


def emd(a, b, M, numItermax=100000, log=False, center_dual=True, numThreads=1, **kw):
    a, b, M = np.asarray(a, np.float64), np.asarray(b, np.float64), np.asarray(M, np.float64)
    if not (np.allclose(a, a[0]) and np.allclose(b, b[0])):
        raise NotImplementedError("ot stand-in: emd supports uniform marginals only")
    if len(a) != len(b):
        return _o.exact_plan_rect(M)[0] * a.sum()
    return _o.perm_plan(_o.exact_perm(M)) * a.sum()


def emd2(a, b, M, numItermax=100000, **kw):
    G = emd(a, b, M)
    return float(np.sum(G * np.asarray(M, np.float64)))


def sinkhorn(a, b, M, reg, method="sinkhorn", numItermax=1000, stopThr=1e-9, **kw):
    if method == "sinkhorn_log":
        u, v, _, _ = _o.sinkhorn_log(M, reg, numItermax, stopThr)
        return _o.sinkhorn_plan(M, reg, u, v)
    return _o.sinkhorn_knopp(M, reg, numItermax, stopThr)


def sinkhorn2(a, b, M, reg, numItermax=1000, stopThr=1e-9, **kw):
    G = sinkhorn(a, b, M, reg, numItermax=numItermax, stopThr=stopThr)
    return float(np.sum(G * np.asarray(M, np.float64)))


class _NS:
    def __init__(self, name):
        self._n = name

    def __getattr__(self, k):
        def _f(*a, **kw):
            raise NotImplementedError(f"ot stand-in: {self._n}.{k} is not restated")
        return _f


class _Unbalanced(_NS):
    @staticmethod
    def sinkhorn_knopp_unbalanced(a, b, M, reg, reg_m, numItermax=1000, stopThr=1e-6, **kw):
        return _o.sinkhorn_knopp_unbalanced(M, reg, reg_m, numItermax, stopThr, a=a, b=b)


class _Partial(_NS):
    @staticmethod
    def entropic_partial_wasserstein(a, b, M, reg, m=None, numItermax=1000, stopThr=1e-100, **kw):
        return _o.entropic_partial_wasserstein(M, reg, m, numItermax, stopThr, a=a, b=b)


unbalanced = _Unbalanced("unbalanced")
partial = _Partial("partial")
