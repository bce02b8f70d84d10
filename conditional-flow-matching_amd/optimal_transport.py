"""MI355X-native minibatch OT coupling — drop-in for ``torchcfm.optimal_transport``.

Same class / method names, arguments, return types and error behaviour as the
reference (``/root/reference/torchcfm/optimal_transport.py``, cited per method as
``ref:LINE``); every numerical step runs in the gfx950 HIP kernels behind the C
ABI (``include/cfm_gfx950.h``).  What changes underneath:

* the cost matrix never leaves HBM (ref:87 copies it to the host every step);
* ``method="exact"`` is solved by the device auction + shortest-augmenting-path
  solver (``cfm_assign_exact_f32``) instead of POT's network simplex (ref:49);
* ``method="sinkhorn"`` runs the log-domain iteration (``cfm_sinkhorn_log_f32``)
  instead of POT's kernel-space Sinkhorn-Knopp (ref:51) — same loop semantics,
  no underflow, so the uniform-plan fallback (ref:93-96) only triggers on NaNs;
* ``sample_plan`` never materialises the B x B plan: the host draws the same
  ``np.random`` uniforms ``np.random.choice`` would consume (ref:118-120) and the
  device does the inverse-cdf lookup (``cfm_plan_sample_perm`` / ``_dense``).

CPU tensors are accepted (they are moved to the GPU and the results moved back)
so the reference's own tests run unchanged; without a GPU every call raises.
"""
import math
import warnings
from typing import Optional, Union

import numpy as np
import torch

from . import _lib
from . import streams as _streams
from ._lib import CfmBackendError, check, ptr, stream_ptr

_SINKHORN_MAX_ITER = 1000      # POT ot.sinkhorn default numItermax
_SINKHORN_STOP_THR = 1e-9      # POT default stopThr
_SINKHORN_CHECK_EVERY = 10     # POT checks the marginal every 10 iterations


def _flatten2(x):
    return x.reshape(x.shape[0], -1) if x.dim() > 2 else x


# --------------------------------------------------------------------------- device steps
def cost_matrix(x0, x1, squared=True, normalize=False, matrix_cores=True):
    """[B0,B1] fp32 cost on the GPU (ref:80-86).  x0/x1: device fp32 [B,d].

    matrix_cores: d >= 64 and B >= 256 take the centred Gram form on the MFMA units (cancelling
    entries recomputed directly; measured closer to fp64 than the direct kernels and 1.6x faster at
    d = 784).  Every sampler path uses it.  (Round 1 kept the exact-OT path on the direct kernels because its
    solver's tail ran 0.21 ms longer on the Gram-form matrix; with the round-2 solver the solve time is the
    same on both — 4.13 vs 4.17 ms over 40 C3 instances — and the cost build is 0.37 instead of 0.61 ms.)"""
    lib = _lib.load()
    B0, B1, d = x0.shape[0], x1.shape[0], x0.shape[1]
    if x1.shape[1] != d:
        raise ValueError("x0 and x1 must have the same feature size")
    M = torch.empty((B0, B1), dtype=torch.float32, device=x0.device)
    mx = torch.empty(1, dtype=torch.float32, device=x0.device) if normalize else None
    if B0 == 0 or B1 == 0:
        return M
    if matrix_cores:
        ws = _lib.workspace(_lib.OP_COST, B0, B1, d, x0.device)
        check(lib.cfm_sqeuclid_cost_ws_f32(ptr(x0), ptr(x1), B0, B1, d, ptr(M), ptr(mx), ptr(ws), stream_ptr()),
              "cfm_sqeuclid_cost_ws_f32")
    else:
        check(lib.cfm_sqeuclid_cost_f32(ptr(x0), ptr(x1), B0, B1, d, ptr(M), ptr(mx), stream_ptr()),
              "cfm_sqeuclid_cost_f32")
    if not squared:
        check(lib.cfm_sqrt_inplace_f32(ptr(M), M.numel(), stream_ptr()), "cfm_sqrt_inplace_f32")
        if normalize:  # max of the un-squared cost
            mx = torch.sqrt(mx)
    if normalize:
        check(lib.cfm_scale_inv_f32(ptr(M), M.numel(), ptr(mx), stream_ptr()), "cfm_scale_inv_f32")
    return M


def assign_exact(M, return_info=False):
    """Optimal permutation of a square fp32 cost matrix on the GPU (int32 [B]).  Runs on the calling thread's solver
    stream when one is set (``cfm_amd.streams.solver_stream``: the CU subset a ``ChipPartition`` reserves for it)."""
    return _streams.run_on_solver_stream(_assign_exact, M, return_info)


def _assign_exact(M, return_info=False):
    lib = _lib.load()
    B = M.shape[0]
    if M.shape[1] != B:
        raise NotImplementedError(
            "exact OT on the gfx950 backend needs equal batch sizes (uniform marginals -> "
            f"assignment problem); got {tuple(M.shape)}")
    dev = M.device
    perm = torch.empty(B, dtype=torch.int32, device=dev)
    # (no zero fills: the library writes all three on success and returns an error code otherwise — three eager fill
    #  launches per coupling less)
    cert = torch.empty(1, dtype=torch.int32, device=dev)
    tot = torch.empty(1, dtype=torch.float64, device=dev)
    stats = torch.empty(8, dtype=torch.int32, device=dev)
    ws = _lib.workspace(_lib.OP_ASSIGN, B, B, 0, dev)
    check(lib.cfm_assign_exact_f32(ptr(M), B, ptr(perm), ptr(cert), ptr(tot), ptr(stats), ptr(ws),
                                   stream_ptr()), "cfm_assign_exact_f32")
    # The call returns only once the result is resident, and it returns CFM_ENOCONV rather than an
    # uncertified permutation, so nothing has to be read back unless the caller wants the statistics.
    info = None
    if return_info and B > 0:
        c, s, t = cert.cpu(), stats.cpu(), tot.cpu()
        if int(c[0]) != 1:
            raise CfmBackendError("exact assignment finished without an optimality certificate")
        info = {"certified": True, "total_cost": float(t[0]), "stats": s.tolist()}
    return (perm, info) if return_info else perm


def assign_exact_batch(Ms, return_info=False):
    """Optimal permutations of several square fp32 cost matrices of the SAME size, solved together: every launch of the
    latency-bound solve carries all of them (cfm_assign_exact_batch_f32), so nb couplings cost little more than one.
    `Ms`: a list of [B,B] tensors or one [nb,B,B] tensor.  Returns an int32 [nb,B] tensor (row b = assign_exact(Ms[b])).
    Runs on the calling thread's solver stream when one is set (``cfm_amd.streams.solver_stream``)."""
    return _streams.run_on_solver_stream(_assign_exact_batch, [m for m in Ms], return_info)


def _assign_exact_batch(Ms, return_info=False):
    import ctypes
    lib = _lib.load()
    Ms = [m for m in Ms]
    nb = len(Ms)
    if nb == 0:
        raise ValueError("assign_exact_batch needs at least one cost matrix")
    B = Ms[0].shape[0]
    dev = Ms[0].device
    for m in Ms:
        if m.dim() != 2 or m.shape[0] != B or m.shape[1] != B:
            raise NotImplementedError("assign_exact_batch needs square cost matrices of one size; got "
                                      f"{[tuple(x.shape) for x in Ms]}")
        if m.device != dev or m.dtype != torch.float32 or not m.is_contiguous():
            raise ValueError("assign_exact_batch: contiguous fp32 matrices on one device")
    # (slices of a stacked [nb,B,B] tensor with odd B start off the 16-byte grid the kernels load on: those are copied)
    Ms = [m if m.data_ptr() % 16 == 0 else m.clone() for m in Ms]
    perm = torch.empty((nb, B), dtype=torch.int32, device=dev)
    cert = torch.empty(nb, dtype=torch.int32, device=dev)
    tot = torch.empty(nb, dtype=torch.float64, device=dev)
    stats = torch.empty((nb, 8), dtype=torch.int32, device=dev)
    ws = _lib.workspace(_lib.OP_ASSIGN, B, B, nb, dev)
    m_ptrs = (ctypes.c_void_p * nb)(*[m.data_ptr() for m in Ms])
    p_ptrs = (ctypes.c_void_p * nb)(*[perm[b].data_ptr() for b in range(nb)])
    check(lib.cfm_assign_exact_batch_f32(m_ptrs, nb, B, p_ptrs, ptr(cert), ptr(tot), ptr(stats), ptr(ws),
                                         stream_ptr()), "cfm_assign_exact_batch_f32")
    if not return_info:
        return perm
    c, st, t = cert.cpu(), stats.cpu(), tot.cpu()
    if B > 0 and not bool((c == 1).all()):
        raise CfmBackendError("exact assignment finished without an optimality certificate")
    return perm, [{"certified": True, "total_cost": float(t[b]), "stats": st[b].tolist()} for b in range(nb)]


# largest lcm(B0, B1) the rectangular exact path expands to.  The expanded problem repeats every row L/B0 and every
# column L/B1 times: it is massively tied, which is the slow regime of every assignment solver (127 x 128 -> L = 16256
# was measured at ~200 s on MI355X), so the bound is a usability bound, not a memory bound.
_RECT_EXACT_MAX = 8192
_RECT_SCIPY_MAX = 1024      # padded LSAP between batches of very different sizes (sample_plan_with_scipy)


# cfm_transport_exact_f32 (round 6: primal-dual phases with a tree push, one workgroup) takes B0 + B1 <= 2048 — the node
# state of the solver lives in the LDS of one CU.  With the assignment warm start sizes that differ by one take ONE phase.
_RECT_TRANSPORT_MAX = 2048
# the warm start solves a max(B0, B1)-sized SQUARE assignment problem on the matrix padded with zero rows: worth it while
# the padding is small (identical zero rows are the tied regime of the square solver)
_RECT_WARM_PAD_FRAC = 0.25


def transport_exact(M, warm_start=None, return_info=False):
    """Exact OT plan between uniform marginals of different sizes on the B0 x B1 matrix itself (primal-dual phases,
    cfm_transport_exact_f32): device fp64 [B0,B1] plan and its cost.  Raises unless the fp64 certificate holds.
    warm_start: None — decide from the sizes; True / False — force.  The warm start is an optimal assignment of the
    smaller side's rows to distinct indices of the larger side (``assign_exact`` on the matrix padded with zero rows);
    the solver validates it and never depends on it."""
    lib = _lib.load()
    B0, B1 = M.shape
    dev = M.device
    M = M.contiguous()
    R, C = min(B0, B1), max(B0, B1)
    if warm_start is None:
        warm_start = (C - R) <= _RECT_WARM_PAD_FRAC * C and C >= 8
    sigma = None
    if warm_start and C > R:
        Ms = torch.zeros((C, C), dtype=torch.float32, device=dev)
        Ms[:R] = M if B0 <= B1 else M.t()
        sigma = assign_exact(Ms)[:R].contiguous()
    plan = torch.empty((B0, B1), dtype=torch.float64, device=dev)
    tot = torch.empty(1, dtype=torch.float64, device=dev)
    info = torch.empty(8, dtype=torch.int32, device=dev)
    ws = _lib.workspace(_lib.OP_TRANSPORT, B0, B1, 0, dev)
    check(lib.cfm_transport_exact_f32(ptr(M), B0, B1, ptr(sigma), ptr(plan), ptr(tot), ptr(info), ptr(ws), stream_ptr()),
          "cfm_transport_exact_f32")
    st = info.cpu()
    if int(st[0]) != 1:
        raise CfmBackendError(f"transportation solver stopped with status {int(st[0])} (phases {int(st[1])}, "
                              f"sweeps {int(st[2])}, violations {int(st[4])})")
    if return_info:
        return plan, float(tot.cpu()[0]), {"phases": int(st[1]), "sweeps": int(st[2]), "support": int(st[3]),
                                           "staged": bool(int(st[7]) & 1), "warm_start_used": bool(int(st[7]) & 2)}
    return plan, float(tot.cpu()[0])


def exact_plan_rect(M):
    """Exact OT plan between uniform marginals of DIFFERENT sizes (what pot.emd returns for
    x0.shape[0] != x1.shape[0], ref:49,79,87) as a device fp64 [B0,B1] tensor, with its cost.

    With L = lcm(B0, B1) every source carries L/B0 units and every target L/B1 units of mass 1/L,
    so the transportation problem is the L x L assignment problem on the cost matrix with rows /
    columns repeated; the device assignment solver does the work, the expansion / folding are
    index plumbing.  The optimal COST is unique; the plan need not be (neither is POT's)."""
    B0, B1 = M.shape
    L = B0 * B1 // math.gcd(B0, B1)
    if L > _RECT_EXACT_MAX:
        if B0 + B1 <= _RECT_TRANSPORT_MAX:
            return transport_exact(M)
        raise NotImplementedError(
            f"exact OT between batches of {B0} and {B1} samples: the lcm expansion ({L} > {_RECT_EXACT_MAX}) is too "
            f"large and the transportation solver takes B0 + B1 <= {_RECT_TRANSPORT_MAX}; use equal batch sizes")
    dev = M.device
    ri = torch.arange(B0, device=dev).repeat_interleave(L // B0)
    ci = torch.arange(B1, device=dev).repeat_interleave(L // B1)
    Mx = M[ri][:, ci].contiguous()
    perm, info = assign_exact(Mx, return_info=True)
    pi = torch.zeros((B0, B1), dtype=torch.float64, device=dev)
    pi.index_put_((ri, ci[perm.long()]), torch.full((L,), 1.0 / L, dtype=torch.float64, device=dev),
                  accumulate=True)
    return pi, info["total_cost"] / L


class SinkhornResult:
    __slots__ = ("f", "g", "iters", "err", "ws", "reg", "M")


def sinkhorn_log(M, reg, max_iter=_SINKHORN_MAX_ITER, stop_thr=_SINKHORN_STOP_THR,
                 check_every=_SINKHORN_CHECK_EVERY):
    """Log-domain Sinkhorn on the GPU; potentials stay resident (fp64) in the workspace."""
    lib = _lib.load()
    B0, B1 = M.shape
    dev = M.device
    r = SinkhornResult()
    r.f = torch.empty(B0, dtype=torch.float32, device=dev)
    r.g = torch.empty(B1, dtype=torch.float32, device=dev)
    r.iters = torch.zeros(1, dtype=torch.int32, device=dev)
    r.err = torch.zeros(1, dtype=torch.float32, device=dev)
    # The fp64 potentials stay in this buffer and the result object refers to them later (plan,
    # sampling, cost): it owns the buffer — a cached, shared workspace would be overwritten by the
    # next same-shape solve on this stream while earlier results are still alive.
    r.ws = torch.empty(lib.cfm_workspace_bytes(_lib.OP_SINKHORN, B0, B1, 0), dtype=torch.uint8, device=dev)
    r.reg, r.M = float(reg), M
    check(lib.cfm_sinkhorn_log_f32(ptr(M), B0, B1, float(reg), int(max_iter), float(stop_thr),
                                   int(check_every), ptr(r.f), ptr(r.g), ptr(r.iters), ptr(r.err),
                                   ptr(r.ws), stream_ptr()), "cfm_sinkhorn_log_f32")
    return r


_POINTS_MAX_DIM = 8     # cfm_sinkhorn_log_points_f32: cost entries recomputed on the fly up to this dimension
# ... and the dimension up to which OTPlanSampler TAKES that solver.  Measured in round 4 (fp32-exp regime, 200
# iterations, variant B / matrix streaming): B = 4096: d = 1..3 1.55 / 1.34 / 1.19, d = 4..8 0.95 ... 0.54;
# B = 8192: 1.47 / 1.28 / 1.15, then 0.96 ... 0.56; B = 1024-2048: 1.1 ... 0.9 at d <= 3, 0.86 ... 0.37 beyond.
# The per-pair cost chain grows with d while a matrix entry stays 4 bytes (round 3 sent every d <= 8 to variant B).
_POINTS_TAKE_DIM = 3


def sinkhorn_log_points(x0, x1, M, reg, max_iter=_SINKHORN_MAX_ITER, stop_thr=_SINKHORN_STOP_THR,
                        check_every=_SINKHORN_CHECK_EVERY):
    """The same solve for d <= 8 without streaming the matrix (cfm_sinkhorn_log_points_f32).  M — the fp32
    matrix the direct cost kernel built from the same x0 / x1 — is only attached to the result for the
    consumers that read it (dense sampling, plan, <pi, M>)."""
    lib = _lib.load()
    B0, d = x0.shape
    B1 = x1.shape[0]
    dev = x0.device
    r = SinkhornResult()
    r.f = torch.empty(B0, dtype=torch.float32, device=dev)
    r.g = torch.empty(B1, dtype=torch.float32, device=dev)
    r.iters = torch.zeros(1, dtype=torch.int32, device=dev)
    r.err = torch.zeros(1, dtype=torch.float32, device=dev)
    r.ws = torch.empty(lib.cfm_workspace_bytes(_lib.OP_SINKHORN, B0, B1, 0), dtype=torch.uint8, device=dev)
    r.reg, r.M = float(reg), M
    check(lib.cfm_sinkhorn_log_points_f32(ptr(x0), ptr(x1), B0, B1, d, float(reg), int(max_iter), float(stop_thr),
                                          int(check_every), ptr(r.f), ptr(r.g), ptr(r.iters), ptr(r.err),
                                          ptr(r.ws), stream_ptr()), "cfm_sinkhorn_log_points_f32")
    return r


def sinkhorn_plan(r):
    lib = _lib.load()
    B0, B1 = r.M.shape
    pi = torch.empty((B0, B1), dtype=torch.float64, device=r.M.device)
    check(lib.cfm_sinkhorn_plan_f64(ptr(r.M), B0, B1, r.reg, ptr(r.ws), ptr(pi), stream_ptr()),
          "cfm_sinkhorn_plan_f64")
    return pi


_UNBALANCED_MAX_ITER, _UNBALANCED_STOP_THR = 1000, 1e-6   # POT sinkhorn_knopp_unbalanced defaults
_PARTIAL_MAX_ITER, _PARTIAL_STOP_THR = 1000, 1e-100         # POT entropic_partial_wasserstein defaults


def _kernel_space_plan(fn_name, M, reg, second, max_iter, stop_thr):
    lib = _lib.load()
    B0, B1 = M.shape
    dev = M.device
    plan = torch.empty((B0, B1), dtype=torch.float64, device=dev)
    info = torch.zeros(4, dtype=torch.int32, device=dev)
    ws = _lib.workspace(_lib.OP_UNBALANCED, B0, B1, 0, dev)
    check(getattr(lib, fn_name)(ptr(M), B0, B1, float(reg), float(second), int(max_iter), float(stop_thr),
                                ptr(plan), ptr(info), ptr(ws), stream_ptr()), fn_name)
    return plan, info


def unbalanced_plan(M, reg, reg_m, max_iter=_UNBALANCED_MAX_ITER, stop_thr=_UNBALANCED_STOP_THR):
    """fp64 plan of pot.unbalanced.sinkhorn_knopp_unbalanced (ref:52-53,87) on the GPU.
    Returns (plan [B0,B1] device fp64, info int32[4] = iterations, status, zeros / non-finite in K)."""
    return _kernel_space_plan("cfm_unbalanced_sinkhorn_f64", M, reg, reg_m, max_iter, stop_thr)


def partial_plan(M, reg, m=1.0, max_iter=_PARTIAL_MAX_ITER, stop_thr=_PARTIAL_STOP_THR):
    """fp64 plan of pot.partial.entropic_partial_wasserstein (ref:54-55,87) on the GPU
    (m = min(|a|, |b|) = 1 for the uniform marginals of get_map)."""
    return _kernel_space_plan("cfm_partial_entropic_f64", M, reg, m, max_iter, stop_thr)


def _u01_to_device(u, dev):
    return torch.from_numpy(np.ascontiguousarray(u, dtype=np.float64)).to(dev)


def sample_perm(perm, u01, B):
    lib = _lib.load()
    n = u01.numel()
    i = torch.empty(n, dtype=torch.int64, device=perm.device)
    j = torch.empty(n, dtype=torch.int64, device=perm.device)
    check(lib.cfm_plan_sample_perm(ptr(perm), ptr(u01), B, n, ptr(i), ptr(j), stream_ptr()),
          "cfm_plan_sample_perm")
    return i, j


def sample_dense(r, u01):
    lib = _lib.load()
    B0, B1 = r.M.shape
    n = u01.numel()
    dev = r.M.device
    i = torch.empty(n, dtype=torch.int64, device=dev)
    j = torch.empty(n, dtype=torch.int64, device=dev)
    ws = _lib.workspace(_lib.OP_SAMPLE_DENSE, B0, B1, 0, dev)
    check(lib.cfm_plan_sample_dense(ptr(r.M), B0, B1, r.reg, ptr(r.ws), ptr(u01), n, ptr(i), ptr(j),
                                    ptr(ws), stream_ptr()), "cfm_plan_sample_dense")
    return i, j


def sample_pi(pi_dev, u01):
    lib = _lib.load()
    B0, B1 = pi_dev.shape
    n = u01.numel()
    dev = pi_dev.device
    i = torch.empty(n, dtype=torch.int64, device=dev)
    j = torch.empty(n, dtype=torch.int64, device=dev)
    ws = _lib.workspace(_lib.OP_SAMPLE_DENSE, B0, B1, 0, dev)
    check(lib.cfm_plan_sample_pi_f64(ptr(pi_dev), B0, B1, ptr(u01), n, ptr(i), ptr(j), ptr(ws),
                                     stream_ptr()), "cfm_plan_sample_pi_f64")
    return i, j


def sample_rows_dense(r, rows, u01):
    """One column per entry of `rows` drawn from that row of the entropic plan (ref:244: np.random.choice(B, p=pi[i] /
    pi[i].sum())): device int64 [n].  The plan is never materialised (potentials + cost row)."""
    lib = _lib.load()
    B0, B1 = r.M.shape
    n = rows.numel()
    dev = r.M.device
    j = torch.empty(n, dtype=torch.int64, device=dev)
    ws = _lib.workspace(_lib.OP_SAMPLE_DENSE, B0, B1, 0, dev)
    check(lib.cfm_plan_sample_rows_dense(ptr(r.M), B0, B1, r.reg, ptr(r.ws), ptr(rows), ptr(u01), n, ptr(j), ptr(ws),
                                         stream_ptr()), "cfm_plan_sample_rows_dense")
    return j


def sample_rows_pi(pi_dev, rows, u01):
    """The same draw from an explicit device fp64 plan (unbalanced / partial / rectangular exact plans)."""
    lib = _lib.load()
    B0, B1 = pi_dev.shape
    n = rows.numel()
    j = torch.empty(n, dtype=torch.int64, device=pi_dev.device)
    check(lib.cfm_plan_sample_rows_pi_f64(ptr(pi_dev), B0, B1, ptr(rows), ptr(u01), n, ptr(j), stream_ptr()),
          "cfm_plan_sample_rows_pi_f64")
    return j


def take_rows(x, idx):
    """x[idx] for device index vectors: the byte-copy kernel, unless x carries an autograd graph
    (a learned encoder in front of the coupling, ref:145 keeps x0[i] differentiable) — then a plain
    differentiable index gather."""
    if x.requires_grad and torch.is_grad_enabled():
        return x[idx.to(x.device)]
    return gather_rows(x.detach().to(idx.device), idx).to(x.device)


def gather_rows(src, idx):
    """src[idx] along dim 0 on the GPU (src: contiguous device tensor, idx: device int64)."""
    lib = _lib.load()
    src = src.contiguous()
    n = idx.numel()
    out = torch.empty((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    row_bytes = src.element_size() * int(np.prod(src.shape[1:], dtype=np.int64)) if src.dim() > 1 \
        else src.element_size()
    check(lib.cfm_gather_rows(ptr(src), ptr(idx), n, row_bytes, ptr(out), stream_ptr()),
          "cfm_gather_rows")
    return out


# --------------------------------------------------------------------------- public API
class OTPlanSampler:
    """OTPlanSampler implements sampling coordinates according to an OT plan (wrt squared
    Euclidean cost) with different implementations of the plan calculation (ref:11-13)."""

    def __init__(
        self,
        method: str,
        reg: float = 0.05,
        reg_m: float = 1.0,
        normalize_cost: bool = False,
        num_threads: Union[int, str] = 1,
        warn: bool = True,
    ) -> None:
        # ref:15-61.  num_threads is accepted for signature compatibility (the device
        # solver has no host thread pool).
        if method not in ("exact", "sinkhorn", "unbalanced", "partial"):
            raise ValueError(f"Unknown method: {method}")
        self.method = method
        self.reg = reg
        self.reg_m = reg_m
        self.normalize_cost = normalize_cost
        self.num_threads = num_threads
        self.warn = warn
        self._last = None  # diagnostics of the most recent solve

    # ---- device-resident solve (no host plan) ----
    def _prepare(self, x0, x1):
        # The reference computes torch.cdist in the INPUT dtype (ref:84) and hands POT that matrix; the device
        # solvers take an fp32 cost matrix (fp64 duals / potentials on top of it).  float64 clouds are therefore
        # coupled on the fp32 rounding of their coordinates — said once per sampler, never silently (the matchers'
        # x_t / u_t arithmetic does honour float64: it takes the composed eager path).
        if self.warn and not getattr(self, "_warned_f64", False) and \
                (getattr(x0, "dtype", None) == torch.float64 or getattr(x1, "dtype", None) == torch.float64):
            self._warned_f64 = True
            warnings.warn("OTPlanSampler: float64 inputs are coupled on a float32 cost matrix (the device solvers' input "
                          "precision); the reference would solve on the float64 matrix.", UserWarning, stacklevel=3)
        dev = _lib.require_gpu()
        a = _lib.to_dev_f32(_flatten2(x0), dev)
        b = _lib.to_dev_f32(_flatten2(x1), dev)
        M = cost_matrix(a, b, squared=True, normalize=self.normalize_cost)
        return dev, M, a, b

    def _solve(self, x0, x1):
        """-> ("perm", perm), ("dense", SinkhornResult) or ("plan", device fp64 plan)."""
        dev, M, a, b = self._prepare(x0, x1)
        if self.method == "unbalanced":
            plan, info = unbalanced_plan(M, self.reg, self.reg_m)
            self._last = info
            return "plan", plan, M
        if self.method == "partial":
            plan, info = partial_plan(M, self.reg)
            self._last = info
            return "plan", plan, M
        if self.method == "exact":
            if M.shape[0] != M.shape[1]:
                plan, _ = exact_plan_rect(M)
                self._last = torch.zeros(4, dtype=torch.int32)
                return "plan", plan, M
            perm = assign_exact(M)        # raises unless the fp64 certificate holds; nothing is read back
            self._last = {"certified": True}
            return "perm", perm, M
        if a.shape[1] <= _POINTS_TAKE_DIM and not self.normalize_cost:
            r = sinkhorn_log_points(a, b, M, self.reg)      # low-dimensional clouds: no pass over the matrix
        else:
            r = sinkhorn_log(M, self.reg)
        self._last = r
        return "dense", r, M

    def get_map(self, x0, x1):
        """Compute the OT plan between a source and a target minibatch (ref:63-97).

        Returns a NumPy float64 array of shape (bs, bs), like the reference.
        """
        kind, sol, M = self._solve(x0, x1)
        if kind == "perm":
            B = M.shape[0]
            p = np.zeros((B, B), dtype=np.float64)
            p[np.arange(B), sol.cpu().numpy().astype(np.int64)] = 1.0 / B
        elif kind == "plan":
            p = sol.cpu().numpy()
            if int(self._last[1].item()) == 1:     # POT warns and keeps the last usable iterate
                warnings.warn("Numerical errors at iteration %d" % int(self._last[0].item()))
        else:
            p = sinkhorn_plan(sol).cpu().numpy()
        if not np.all(np.isfinite(p)):   # ref:88-92
            print("ERROR: p is not finite")
            print(p)
            print("Cost mean, max", M.mean(), M.max())
            print(x0, x1)
        if np.abs(p.sum()) < 1e-8:       # ref:93-96
            if self.warn:
                warnings.warn("Numerical errors in OT plan, reverting to uniform plan.")
            p = np.ones_like(p) / p.size
        return p

    def sample_map(self, pi, batch_size, replace=True):
        r"""Draw source and target samples from pi $(x,z) \sim \pi$ (ref:99-121).

        ``pi`` is a NumPy plan; the uniforms come from the global ``np.random`` state exactly
        as ``np.random.choice(..., p=p, size=batch_size, replace=replace)`` consumes them; the
        cdf search runs on the GPU.
        """
        dev = _lib.require_gpu()
        pi = np.asarray(pi)
        B0, B1 = pi.shape
        pi_dev = torch.from_numpy(np.ascontiguousarray(pi, dtype=np.float64)).to(dev)
        if replace:
            u = _u01_to_device(np.random.random_sample(batch_size), dev)
            i, j = sample_pi(pi_dev, u)
            return i.cpu().numpy(), j.cpu().numpy()
        # replace=False: numpy's rejection loop (legacy RandomState.choice), cdf on the device
        if batch_size > int(np.count_nonzero(pi > 0)):
            raise ValueError("Fewer non-zero entries in p than size")
        lib = _lib.load()
        pi_dev = pi_dev.clone()
        found = np.zeros(batch_size, dtype=np.int64)
        n_uniq = 0
        while n_uniq < batch_size:
            x = np.random.random_sample(batch_size - n_uniq)
            if n_uniq > 0:
                fl = torch.from_numpy(found[:n_uniq].copy()).to(dev)
                check(lib.cfm_plan_zero_entries_f64(ptr(pi_dev), ptr(fl), n_uniq, stream_ptr()),
                      "cfm_plan_zero_entries_f64")
            i, j = sample_pi(pi_dev, _u01_to_device(x, dev))
            new = (i * B1 + j).cpu().numpy()
            _, unique_indices = np.unique(new, return_index=True)
            unique_indices.sort()
            new = new.take(unique_indices)
            found[n_uniq:n_uniq + new.size] = new
            n_uniq += new.size
        return np.divmod(found, B1)

    def _sample_indices(self, x0, x1, replace=True):
        """Device int64 index pairs drawn from the plan of (x0, x1), plus the device copies."""
        n = x0.shape[0]
        if not replace:
            pi = self.get_map(x0, x1)
            i, j = self.sample_map(pi, n, replace=False)
            dev = _lib.require_gpu()
            return torch.from_numpy(i).to(dev), torch.from_numpy(j).to(dev)
        return self._indices_from_solution(x0, x1, self._solve(x0, x1))

    def _indices_from_solution(self, x0, x1, solution):
        """The draw of ``_sample_indices`` for an already solved coupling (``_solve`` / ``_solve_many`` result)."""
        n = x0.shape[0]
        kind, sol, M = solution
        dev = M.device
        if kind == "plan":
            # get_map's diagnostics (ref:88-96) on the device plan, then sample_map (ref:116-121)
            finite = bool(torch.isfinite(sol).all())
            total = float(sol.sum()) if finite else float("nan")
            if not finite:
                print("ERROR: p is not finite")
                print(sol)
                print("Cost mean, max", M.mean(), M.max())
                print(x0, x1)
                raise ValueError("probabilities contain NaN")   # what np.random.choice raises, ref:118
            u = _u01_to_device(np.random.random_sample(n), dev)
            if abs(total) < 1e-8:
                if self.warn:
                    warnings.warn("Numerical errors in OT plan, reverting to uniform plan.")
                flat = torch.clamp((u * (M.shape[0] * M.shape[1])).floor().long(), max=M.numel() - 1)
                return flat // M.shape[1], flat % M.shape[1]
            return sample_pi(sol, u)
        if kind == "dense":
            fin = bool(torch.isfinite(sol.f).all() and torch.isfinite(sol.g).all())
            if not fin:
                # what the reference does with a non-finite plan (ref:88-96, 118): the diagnostics, no uniform
                # revert (|NaN| < 1e-8 is False), and np.random.choice raises
                print("ERROR: p is not finite")
                print("Cost mean, max", M.mean(), M.max())
                print(x0, x1)
                raise ValueError("probabilities contain NaN")   # (before the draw: the NumPy stream is not consumed)
            u = _u01_to_device(np.random.random_sample(n), dev)
            return sample_dense(sol, u)
        u = _u01_to_device(np.random.random_sample(n), dev)
        return sample_perm(sol, u, M.shape[0])

    def sample_plan(self, x0, x1, replace=True):
        r"""Compute the OT plan and draw source and target samples from it (ref:123-145)."""
        i, j = self._sample_indices(x0, x1, replace=replace)
        dev = i.device
        g0 = take_rows(x0, i)
        g1 = take_rows(x1, j)
        return g0, g1

    def sample_plan_with_scipy(self, x0, x1):
        r"""Deterministic OT pairing: keeps x0's order and returns x1[perm] (ref:147-182).

        The reference calls scipy.optimize.linear_sum_assignment on the host; here the same
        permutation comes from the device solver.
        """
        x0f, x1f = _flatten2(x0), _flatten2(x1)
        dev = _lib.require_gpu()
        a, b = _lib.to_dev_f32(x0f, dev), _lib.to_dev_f32(x1f, dev)
        M = cost_matrix(a, b, squared=True, normalize=self.normalize_cost)
        B0, B1 = M.shape
        if B0 == B1:
            perm = assign_exact(M).long()
        else:
            # scipy.optimize.linear_sum_assignment on a rectangular matrix pairs min(B0, B1) rows / columns at
            # minimum total cost and returns the column indices in row order (ref:179: `_, j = ...`).  Zero-cost
            # dummy rows / columns make it a square problem with the same optimum on the real entries.
            n = max(B0, B1)
            if n > _RECT_SCIPY_MAX and n > 4 * min(B0, B1):
                # |B0 - B1| identical all-zero dummy lines are the massively tied regime of the assignment solver
                # (every dummy bids for the same cheapest column: one award per auction round)
                raise NotImplementedError(
                    f"sample_plan_with_scipy between {B0} and {B1} samples pads to a {n} x {n} problem with "
                    f"{abs(B0 - B1)} tied dummy lines; supported up to n = {_RECT_SCIPY_MAX} or a 4 : 1 size ratio")
            Mx = torch.zeros((n, n), dtype=torch.float32, device=dev)
            Mx[:B0, :B1] = M
            full = assign_exact(Mx).long()[:B0]
            perm = full[full < B1]
        return x0f, gather_rows(x1f.detach().to(dev), perm).to(x1.device)

    def sample_plan_with_labels(self, x0, x1, y0=None, y1=None, replace=True):
        r"""As sample_plan, also gathering the labels (ref:184-219)."""
        i, j = self._sample_indices(x0, x1, replace=replace)
        dev = i.device
        return (
            take_rows(x0, i),
            take_rows(x1, j),
            take_rows(y0, i) if y0 is not None else None,
            take_rows(y1, j) if y1 is not None else None,
        )

    def _solve_many(self, pairs, workers=3, throughput=False):
        """Solve independent couplings together: exact couplings of one square size as ONE batch of the solver
        (``assign_exact_batch``), everything else concurrently with one host thread + one HIP stream per worker.
        Returns the ``_solve`` results in order, usable on the caller's stream.
        throughput: the caller couples ahead of a training loop (``sample_location_and_conditional_flow_group``): also a
        SINGLE exact coupling then goes through the batch entry, i.e. on the throughput grid (<= 64 workgroups) instead of
        the chip-wide grid of a latency-critical lone solve — a chip-sized persistent auction grid beside the grids of two
        other prefetch jobs was the 27 - 30 ms stall of round 5's public-API loop (profiles/r6_tail_public.txt)."""
        import concurrent.futures as cf
        import threading
        dev = _lib.require_gpu()
        n0 = pairs[0][0].shape[0] if pairs else 0
        one_batch = (self.method == "exact" and n0 > 256 and
                     all(a.shape[0] == n0 and b.shape[0] == n0 for a, b in pairs))
        if (len(pairs) <= 1 or workers <= 1) and not (throughput and one_batch and len(pairs) == 1):
            return [self._solve(a, b) for a, b in pairs]
        if one_batch:
            # equal, square sizes beyond the one-workgroup solver: the assignment problems share ONE chain of launches
            # (decided from the shapes: the other cases build their cost matrices inside the workers, once)
            Ms = [self._prepare(a, b)[1] for a, b in pairs]
            perms = assign_exact_batch(Ms)
            self._last = {"certified": True}
            return [("perm", perms[k], Ms[k]) for k in range(len(Ms))]
        tls = threading.local()
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(dev))
        last = self._last

        def work(a, b):
            torch.cuda.set_device(dev)
            if not hasattr(tls, "stream"):
                tls.stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(tls.stream):
                tls.stream.wait_event(ready)
                out = self._solve(a, b)
                done = torch.cuda.Event()
                done.record(tls.stream)
            return out, done

        with cf.ThreadPoolExecutor(max_workers=min(workers, len(pairs))) as pool:
            res = list(pool.map(lambda ab: work(*ab), pairs))
        cur = torch.cuda.current_stream(dev)
        outs = []
        for out, done in res:
            cur.wait_event(done)
            outs.append(out)
        self._last = last
        return outs

    def sample_trajectory(self, X):
        """OT trajectories across consecutive time slices (ref:221-251).

        The times-1 couplings are independent of each other, so they are solved concurrently on
        side streams (SURVEY §8f "multi-marginal batching") and stay on the device; only the
        per-row draws are chained in time order, consuming the global ``np.random`` stream exactly
        like the reference's per-row ``np.random.choice(p=pi[i] / pi[i].sum())`` (one uniform per
        row, rows in order, slices in order).
        """
        times = X.shape[1]
        dev = _lib.require_gpu()
        n = X.shape[0]
        sols = self._solve_many([(X[:, t], X[:, t + 1]) for t in range(times - 1)])
        # The chain of per-row draws stays on the device: one kernel per slice reads the plan row where it lies
        # (cfm_plan_sample_rows_*), the indices of a slice feed the next one directly, and the host sees them once,
        # at the end.  The host RNG is consumed exactly as the reference's loop consumes it: one uniform per row and
        # slice, in row order (np.random.choice draws one double per call, ref:244).
        rows = torch.arange(n, dtype=torch.int64, device=dev)
        chain = [rows]
        # get_map's diagnostics (ref:88-96; the reference calls get_map for every slice before it draws, ref:233): a
        # non-finite plan is reported and np.random.choice raises on it (ref:244); a plan without mass reverts to the
        # uniform plan.  ONE small device reduction per entropic slice, ONE read-back for all slices, before the chain of
        # draws starts (exact slices have neither case).  A potential-form ("dense") slice cannot lose its mass while its
        # potentials are finite: the row update of the log-domain loop normalises every row to 1 / n by construction —
        # its check is the finiteness of f and g.
        diag = []
        for kind, sol, M in sols:
            if kind == "plan":
                diag.append(torch.stack([torch.isfinite(sol).all().double(), torch.nan_to_num(sol, nan=0.0, posinf=0.0, neginf=0.0).sum()]))
            elif kind == "dense":
                diag.append(torch.stack([(torch.isfinite(sol.f).all() & torch.isfinite(sol.g).all()).double(),
                                         torch.ones((), dtype=torch.float64, device=dev)]))
        diag = torch.stack(diag).cpu().numpy() if diag else np.zeros((0, 2))
        sols = list(sols)
        q = 0
        for t, (kind, sol, M) in enumerate(sols):
            if kind == "perm":
                continue
            finite, total = bool(diag[q, 0]), float(diag[q, 1]); q += 1
            if not finite:
                print("ERROR: p is not finite")
                if kind == "plan":
                    print(sol)
                print("Cost mean, max", M.mean(), M.max())
                print(X[:, t], X[:, t + 1])
                raise ValueError("probabilities contain NaN")
            if abs(total) < 1e-8:
                if self.warn:
                    warnings.warn("Numerical errors in OT plan, reverting to uniform plan.")
                sols[t] = ("plan", torch.full(tuple(M.shape), 1.0 / M.numel(), dtype=torch.float64, device=dev), M)   # ref:96
        massless = torch.zeros((), dtype=torch.bool, device=dev)
        for t, (kind, sol, M) in enumerate(sols):
            if kind == "plan":
                # a visited row without mass: pi[i] / pi[i].sum() is NaN and np.random.choice raises (ref:244) — the flag
                # stays on the device and is read once, behind the last slice (the draws of a failing call are discarded)
                massless = massless | (sol.sum(1)[rows] <= 0).any()
            u = _u01_to_device(np.random.random_sample(n), dev)
            if kind == "perm":
                # a permutation plan: row i has the single nonzero pi[i, perm[i]] (the draw is consumed)
                j = sol.long()[rows]
            elif kind == "dense":
                j = sample_rows_dense(sol, rows, u)
            else:
                j = sample_rows_pi(sol.contiguous(), rows, u)
            chain.append(j)
            rows = j
        if bool(massless):
            raise ValueError("probabilities contain NaN")
        idx = torch.stack(chain).cpu().numpy()              # [times, n]: the ONE device-to-host copy of the indices
        Xh = np.asarray(X.detach().cpu() if isinstance(X, torch.Tensor) else X)
        return np.stack([Xh[:, t][idx[t]] for t in range(times)], axis=1)


def wasserstein(
    x0: torch.Tensor,
    x1: torch.Tensor,
    method: Optional[str] = None,
    reg: float = 0.05,
    power: int = 2,
    **kwargs,
) -> float:
    """Wasserstein-1/2 distance between two minibatches (ref:254-303)."""
    assert power == 1 or power == 2
    if method == "exact" or method is None:
        exact = True
    elif method == "sinkhorn":
        exact = False
    else:
        raise ValueError(f"Unknown method: {method}")
    dev = _lib.require_gpu()
    a = _lib.to_dev_f32(_flatten2(x0), dev)
    b = _lib.to_dev_f32(_flatten2(x1), dev)
    M = cost_matrix(a, b, squared=(power == 2))
    if exact and M.shape[0] != M.shape[1]:
        _, ret = exact_plan_rect(M)
    elif exact:
        perm, info = assign_exact(M, return_info=True)
        ret = info["total_cost"] / M.shape[0]
    else:
        lib = _lib.load()
        r = sinkhorn_log(M, reg, max_iter=int(kwargs.get("numItermax", int(1e7))))  # ref:300
        out = torch.zeros(1, dtype=torch.float64, device=dev)
        check(lib.cfm_sinkhorn_cost_f64(ptr(M), M.shape[0], M.shape[1], float(reg), ptr(r.ws), ptr(out),
                                        stream_ptr()), "cfm_sinkhorn_cost_f64")
        ret = float(out.cpu()[0])
    if power == 2:
        ret = math.sqrt(ret)
    return ret
