"""ctypes front of oracle/sinkhorn_oracle.c (TEST INFRASTRUCTURE: the threaded float64 Sinkhorn
oracle for the full BASELINE sizes).  build() compiles it with gcc into oracle/_build/ (kept out of
git, shipped to the GPU box with the snapshot)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "sinkhorn_oracle.c")
LIB = os.path.join(_HERE, "_build", "libsk_oracle.so")
_lib = None


def build():
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.run(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", SRC, "-o", LIB, "-lm"], check=True)
    return LIB


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(LIB)
        _lib.cfm_oracle_sinkhorn_log.restype = ctypes.c_int
        _lib.cfm_oracle_sinkhorn_log.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                                 ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_void_p,
                                                 ctypes.c_void_p, ctypes.c_void_p]
    return _lib


def sinkhorn_log(M, reg, numItermax=1000, stopThr=1e-9, check_every=10):
    """Same contract as cfm_oracle.sinkhorn_log: (u, v, n_iter, err) for the fp32 cost matrix M."""
    lib = _load()
    M = np.ascontiguousarray(M, dtype=np.float32)
    n, m = M.shape
    u = np.zeros(n); v = np.zeros(m); err = ctypes.c_double(0.0)
    it = lib.cfm_oracle_sinkhorn_log(M.ctypes.data, n, m, float(reg), int(numItermax), float(stopThr),
                                     int(check_every), u.ctypes.data, v.ctypes.data, ctypes.byref(err))
    return u, v, int(it), float(err.value)
