"""ctypes binding of libcfm_gfx950.so (the C ABI declared in include/cfm_gfx950.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; every
numerical step of the hot path runs in the hand-written HIP kernels behind the
C ABI.  There is deliberately NO CPU fallback: if the library or an MI355X is
missing the calls raise (``CfmBackendError``).
"""
import ctypes
import os
import subprocess
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcfm_gfx950.so")
if os.environ.get("CFM_LIB_PATH"):      # a variant build for A/B measurements (tools/probe/build_variant.sh); never a fallback
    LIB_PATH = os.path.abspath(os.environ["CFM_LIB_PATH"])
_lock = threading.Lock()
_lib = None

ABI_VERSION = 3      # CFM_ABI_VERSION of include/cfm_gfx950.h

# ops (include/cfm_gfx950.h)
OP_SINKHORN, OP_ASSIGN, OP_SAMPLE_DENSE, OP_MLP, OP_ODE, OP_UNBALANCED, OP_COST, OP_MLP_TRAIN, OP_TRANSPORT = 1, 2, 3, 4, 5, 6, 7, 8, 9
VARIANT_ICFM, VARIANT_SB, VARIANT_TARGET, VARIANT_VP = 0, 1, 2, 3

ERRORS = {
    -1: "CFM_EINVAL (bad shape / null pointer / unsupported size)",
    -2: "CFM_EALIGN (pointer not aligned)",
    -3: "CFM_ENOCONV (solver stopped without a certificate)",
    -4: "CFM_ETIMEOUT (device state machine made no progress)",
}


class CfmBackendError(RuntimeError):
    """The gfx950 HIP backend is unavailable or a C-ABI call failed."""


_vp = ctypes.c_void_p
_i = ctypes.c_int
_sz = ctypes.c_size_t
_f = ctypes.c_float
_d = ctypes.c_double

# symbol -> (restype, argtypes); must list every symbol of include/cfm_gfx950.h
SIGNATURES = {
    "cfm_abi_version": (_i, []),
    "cfm_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "cfm_stream_create_cu_mask": (_i, [_vp, _i, _vp]),
    "cfm_stream_destroy": (_i, [_vp]),
    "cfm_sqeuclid_cost_f32": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "cfm_sqeuclid_cost_ws_f32": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "cfm_scale_inv_f32": (_i, [_vp, _sz, _vp, _vp]),
    "cfm_sqrt_inplace_f32": (_i, [_vp, _sz, _vp]),
    "cfm_sinkhorn_log_f32": (_i, [_vp, _i, _i, _d, _i, _d, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cfm_sinkhorn_log_points_f32": (_i, [_vp, _vp, _i, _i, _i, _d, _i, _d, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cfm_sinkhorn_potentials_f64": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "cfm_sinkhorn_plan_f64": (_i, [_vp, _i, _i, _d, _vp, _vp, _vp]),
    "cfm_sinkhorn_cost_f64": (_i, [_vp, _i, _i, _d, _vp, _vp, _vp]),
    "cfm_unbalanced_sinkhorn_f64": (_i, [_vp, _i, _i, _d, _d, _i, _d, _vp, _vp, _vp, _vp]),
    "cfm_partial_entropic_f64": (_i, [_vp, _i, _i, _d, _d, _i, _d, _vp, _vp, _vp, _vp]),
    "cfm_assign_exact_f32": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cfm_assign_exact_batch_f32": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cfm_transport_exact_f32": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cfm_plan_sample_perm": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    "cfm_plan_sample_dense": (_i, [_vp, _i, _i, _d, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "cfm_plan_sample_pi_f64": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "cfm_plan_sample_rows_dense": (_i, [_vp, _i, _i, _d, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "cfm_plan_sample_rows_pi_f64": (_i, [_vp, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "cfm_sample_xt_ut_f32": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _d, _vp, _vp, _vp, _i, _i,
                                  _vp, _vp, _vp, _vp, _vp]),
    "cfm_gather_rows": (_i, [_vp, _vp, _i, _sz, _vp, _vp]),
    "cfm_mlp_forward_f32": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "cfm_mlp_forward_train_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "cfm_mlp_backward_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cfm_sde_em_mlp_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _i, _vp, ctypes.c_ulonglong, _vp, _vp, _vp]),
    "cfm_mlp_regression_step_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cfm_adam_step_f32": (_i, [_vp, _i, _d, _d, _d, _d, _d, _i, _d, _vp]),
    "cfm_sde_em_step_f32": (_i, [_vp, _vp, _vp, _vp, _d, _d, _d, _sz, _vp]),
    "cfm_rbf_mix_sum_f32": (_i, [_vp, _sz, _vp, _i, _vp, _vp]),
    "cfm_ode_euler_mlp_f32": (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "cfm_ode_dopri5_mlp_f32": (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _i, _f, _f, _vp, _vp, _vp,
                                   _vp, _vp]),
}
# helpers that are not part of the documented ABI (tuning / replace=False bookkeeping)
EXTRA_SIGNATURES = {
    "cfm_assign_set_params": (None, [_d, _d, _d, _d, _i, _i, _i]),
    "cfm_assign_set_mode": (None, [_i]),
    "cfm_assign_set_handoff": (None, [_i]),
    "cfm_assign_debug_fallback": (None, [_vp]),
    "cfm_assign_debug_times": (_i, [_vp, _vp]),
    "cfm_assign_set_wide_blocks": (None, [_i]),
    "cfm_assign_set_stop_early": (None, [_d]),
    "cfm_assign_set_bulk": (None, [_i, _i]),
    "cfm_assign_set_small": (None, [_i]),
    "cfm_assign_set_async": (None, [_i, _i, _i]),
    "cfm_assign_get_async": (None, [_vp]),
    "cfm_assign_set_async_min_n": (None, [_i]),
    "cfm_assign_debug_small": (None, [_vp]),
    "cfm_assign_debug_solver": (_i, [_vp, _i, _vp]),
    "cfm_plan_zero_entries_f64": (_i, [_vp, _vp, _i, _vp]),
    "cfm_ode_set_fused": (None, [_i]),
    "cfm_mlp_set_glds": (None, [_i]),
    "cfm_mlp_get_glds": (_i, []),
    "cfm_set_blocking_sync": (None, [_i]),
}


def build(verbose=False):
    """Compile every HIP source for gfx950 into libcfm_gfx950.so (in-tree)."""
    script = os.path.join(_HERE, "csrc", "build.sh")
    res = subprocess.run(["bash", script], capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
        print(res.stderr)
    if res.returncode != 0:
        raise CfmBackendError("hipcc build of libcfm_gfx950.so failed:\n" + res.stderr[-4000:])
    return LIB_PATH


def load():
    """dlopen the C-ABI library (after torch, so both share one HIP runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise CfmBackendError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback."
            )
        lib = ctypes.CDLL(LIB_PATH)
        # version first: a stale build (or a CFM_LIB_PATH variant of another round) must say so, not die on getattr
        try:
            lib.cfm_abi_version.restype = _i
            have = lib.cfm_abi_version()
        except AttributeError:
            have = None
        if have != ABI_VERSION:
            raise CfmBackendError(f"{LIB_PATH}: ABI version {have}, this binding needs {ABI_VERSION} — rebuild "
                                  "(python -c 'import __graft_entry__ as g; g.build()')")
        for name, (res, args) in list(SIGNATURES.items()) + list(EXTRA_SIGNATURES.items()):
            try:
                fn = getattr(lib, name)
            except AttributeError:
                raise CfmBackendError(f"{LIB_PATH} does not export {name}: stale build, rebuild it") from None
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def exported_symbols():
    lib = load()
    return [n for n in SIGNATURES if hasattr(lib, n)]


def require_gpu():
    if not torch.cuda.is_available():
        raise CfmBackendError(
            "cfm_amd needs an AMD MI355X (gfx950) GPU: the HIP kernels are the only compute path "
            "(no CPU fallback)."
        )
    return torch.device("cuda", torch.cuda.current_device())


def check(rc, what):
    if rc == 0:
        return
    if rc < 0:
        raise CfmBackendError(f"{what}: {ERRORS.get(rc, rc)}")
    raise CfmBackendError(f"{what}: hipError_t {rc}")


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


_ws_tls = threading.local()
_WS_CACHE_MAX = 64


def workspace(op, B0, B1, d=0, device=None, tag=0):
    """Cached scratch buffer for (op, shape, stream) of the CALLING thread; a uint8 tensor.

    The cache is per host thread (coupling workers each own a stream) and is never cleared under
    another thread's feet; when it is full the least recently used entry of this thread is dropped
    (the buffer goes back to torch's allocator, which orders its reuse behind the stream it was
    allocated on)."""
    lib = load()
    device = device or require_gpu()
    cache = getattr(_ws_tls, "cache", None)
    if cache is None:
        cache = _ws_tls.cache = {}
    key = (op, B0, B1, d, str(device), tag, torch.cuda.current_stream().cuda_stream)
    buf = cache.pop(key, None)
    if buf is None:
        n = lib.cfm_workspace_bytes(op, B0, B1, d)
        if n == 0:
            raise CfmBackendError(f"cfm_workspace_bytes({op},{B0},{B1},{d}) = 0")
        buf = torch.empty(n, dtype=torch.uint8, device=device)
        while len(cache) >= _WS_CACHE_MAX:
            cache.pop(next(iter(cache)))
    cache[key] = buf          # (re-)inserted last: dicts keep insertion order = recency
    return buf


def to_dev_f32(x, device=None):
    """Contiguous fp32 copy/view of `x` on the GPU."""
    device = device or require_gpu()
    return x.detach().to(device=device, dtype=torch.float32).contiguous()
