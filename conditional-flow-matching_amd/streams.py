"""Chip partition: HIP streams restricted to CU subsets, and the thread-local "solver stream".

The reference runs its coupling on the host between two model steps (torchcfm/conditional_flow_matching.py:271-272,
examples/images/cifar10/train_cifar10.py:141-151): nothing overlaps.  Here the couplings of the next minibatches run
beside the model step (``cfm_amd.prefetch``).  Two kinds of kernels then meet on the chip: the exact solver's
latency-bound rounds (1024-thread workgroups, 48 KiB of LDS, ~8 us each, ~100 in a dependent chain) and the dense
fp32-MFMA products (cost matrix, model step).  Sharing every CU, each waits for workgroup slots the other holds
(round 3: a round took 14 us instead of 8.9 inside the loop, the cost product 535 us instead of 320).  A
``ChipPartition`` gives the solver a CU subset of its own — ``solver_cus_per_xcd`` CUs of every XCD — and the dense
products the complement, through ``hipExtStreamCreateWithCUMask`` (C ABI: ``cfm_stream_create_cu_mask``).

CU mask bit i = XCD i % 8, CU i / 8 of that XCD (gfx950, SPX mode); every XCD keeps CUs on both sides, because the
dispatcher deals the workgroups of a grid round-robin over the XCDs whatever the mask.
"""
import contextlib
import ctypes
import threading

import torch

from . import _lib
from ._lib import check

_tls = threading.local()
N_XCD = 8
_pool, _pool_lock = {}, threading.Lock()      # (device, mask words) -> parked CU-masked streams (see ChipPartition.close)


def _mask_words(bits, ncu):
    words = [0] * ((ncu + 31) // 32)
    for i in bits:
        words[i >> 5] |= 1 << (i & 31)
    return words


class ChipPartition:
    """``solver_cus_per_xcd`` CUs of each XCD for the exact solver's streams, the rest for the dense products.

    ``solver_stream()`` / ``dense_stream()`` create a NEW stream on that subset each call (a worker thread keeps its
    own); ``close()`` synchronises and parks them for reuse.  The streams are ``torch.cuda.ExternalStream`` objects: usable with
    ``torch.cuda.stream(...)``, events and ``record_stream`` like any other."""

    def __init__(self, device=None, solver_cus_per_xcd=4, dense_all=False, solver_all=False):
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        ncu = torch.cuda.get_device_properties(self.device).multi_processor_count
        # The mask layout (bit i = XCD i % 8, CU i // 8 of it) is the SPX mode of an 8-XCD part (MI355X: 256 CUs).  A
        # CPX / NPS-partitioned device or another part exposes a different CU count per agent: refuse instead of building
        # masks whose two halves are no longer complementary per XCD.
        if ncu % N_XCD != 0 or ncu < 2 * N_XCD:
            raise RuntimeError(f"ChipPartition assumes {N_XCD} XCDs in SPX mode; this device reports {ncu} CUs "
                               "(compute-partitioned mode?): run without a partition")
        per = ncu // N_XCD
        k = int(solver_cus_per_xcd)
        if not 1 <= k < per:
            raise ValueError(f"solver_cus_per_xcd must be in 1..{per - 1} (the device has {per} CUs per XCD)")
        self.ncu, self.k = ncu, k
        self.solver_bits = list(range(0, N_XCD * (per if solver_all else k)))
        self.dense_bits = list(range(0 if dense_all else N_XCD * k, N_XCD * per))
        self._streams = []
        self._lock = threading.Lock()

    @property
    def solver_cus(self):
        return len(self.solver_bits)

    @property
    def dense_cus(self):
        return len(self.dense_bits)

    def _create(self, bits):
        words = tuple(_mask_words(bits, self.ncu))
        key = (str(self.device), words)
        with _pool_lock:
            free = _pool.setdefault(key, [])
            s = free.pop() if free else None
        if s is None:
            lib = _lib.load()
            arr = (ctypes.c_uint32 * len(words))(*words)
            out = ctypes.c_void_p(0)
            with torch.cuda.device(self.device):
                check(lib.cfm_stream_create_cu_mask(arr, len(words), ctypes.byref(out)), "cfm_stream_create_cu_mask")
            s = torch.cuda.ExternalStream(out.value, device=self.device)
        with self._lock:
            self._streams.append((key, s))
        return s

    def solver_stream(self):
        return self._create(self.solver_bits)

    def dense_stream(self):
        return self._create(self.dense_bits)

    def close(self):
        """Synchronise this partition's streams and park them for the next partition with the same masks.  They are NOT
        destroyed: torch's caching allocator keeps referring to every stream a tensor was ever recorded on
        (``record_stream`` -> an event recorded on that stream when the block is freed), so a destroyed handle would be
        used after free; ``cfm_stream_destroy`` is for callers that own the whole lifetime of what ran on the stream."""
        with self._lock:
            streams, self._streams = self._streams, []
        for key, s in streams:
            s.synchronize()
            with _pool_lock:
                _pool.setdefault(key, []).append(s)


@contextlib.contextmanager
def solver_stream(stream):
    """Within the block (this host thread only) the exact solver — ``assign_exact`` / ``assign_exact_batch`` and
    everything built on them — runs on ``stream`` instead of the current one, ordered after the current stream's work
    and before whatever the current stream does next.  ``None`` switches the redirection off."""
    prev = getattr(_tls, "solver", None)
    _tls.solver = stream
    try:
        yield stream
    finally:
        _tls.solver = prev


def current_solver_stream():
    return getattr(_tls, "solver", None)


def _mark(o, stream):
    if isinstance(o, torch.Tensor):
        if o.is_cuda:
            o.record_stream(stream)
    elif isinstance(o, (list, tuple)):
        for x in o:
            _mark(x, stream)
    elif isinstance(o, dict):
        for x in o.values():
            _mark(x, stream)


def run_on_solver_stream(fn, *args, **kw):
    """``fn(*args, **kw)`` on this thread's solver stream (if one is set and differs from the current stream):
    solver waits for the current stream, the current stream waits for the solver, tensors crossing either way are
    recorded on the stream that uses them next."""
    s = current_solver_stream()
    if s is None:
        return fn(*args, **kw)
    cur = torch.cuda.current_stream(s.device)
    if s.cuda_stream == cur.cuda_stream:
        return fn(*args, **kw)
    s.wait_stream(cur)
    _mark(args, s)
    with torch.cuda.stream(s):
        out = fn(*args, **kw)
    cur.wait_stream(s)
    _mark(out, cur)
    return out
