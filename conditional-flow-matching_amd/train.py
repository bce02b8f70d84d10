"""The regression step of the reference's training loops as ONE call into the HIP library.

Every loop of the reference ends the same way (examples/images/cifar10/train_cifar10.py:141-151, the 2-D
tutorials, single-cell_example.ipynb cell 6):

    optimizer.zero_grad()
    t, xt, ut = FM.sample_location_and_conditional_flow(x0, x1)
    vt = model(torch.cat([xt, t[:, None]], dim=-1))
    loss = torch.mean((vt - ut) ** 2)
    loss.backward()
    optimizer.step()

``RegressionStep(model, optimizer)(t, xt, ut)`` is the last four lines for a ``cfm_amd.MLP`` field and a
``cfm_amd.FusedAdam``: ``cfm_mlp_regression_step_f32`` (forward with the time column fused into the first layer's
epilogue, MSE + its gradient seed, dgrad / wgrad, one fixed-order reduction) followed by ``cfm_adam_step_f32`` — 14
launches of this library's kernels and no eager PyTorch op in between (the eager form spends ~10 extra
``at::native`` launches on cat / sub / pow / mean / fill per step).  Same arithmetic as the autograd path
(same kernels); gradients land in persistent ``.grad`` buffers, so the optimizer's pointer table is built once.
With ``torch.distributed`` initialised (one process per GPU, rank-local coupling) the gradients are averaged the way
the reference's DDP wrapper does it (train_cifar10_ddp.py:92,167-180): one bucket per layer, all-reduced on a
communication stream as soon as that layer's gradient is final (``layer_done`` events recorded by the C call between
its launches) while the remaining layers' products still run; the compute stream waits for the buckets, and the
``1 / world`` of the mean is a factor inside the Adam launch (``grad_scale``) — no eager op in the step for N > 1 either.
"""
import ctypes

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr
from .models import MLP


class RegressionStep:
    def __init__(self, model, optimizer, data_parallel=None):
        """data_parallel: None — the bucketed gradient all-reduce runs iff torch.distributed is initialised with more than
        one rank; True — whenever a process group exists, also with ONE rank (the N > 1 composition — communication
        stream, per-layer events, grad_scale — on the real backend of a one-GPU box); False — never."""
        self.data_parallel = data_parallel
        net = model.module if hasattr(model, "module") and isinstance(model.module, MLP) else model
        if not isinstance(net, MLP):
            raise TypeError("RegressionStep drives a cfm_amd.MLP vector field")
        self.net, self.opt = net, optimizer
        from .optim import FusedAdam
        self._fused_opt = isinstance(optimizer, FusedAdam)           # its step() takes the 1 / world of the mean (grad_scale)
        self.lins = net._linears()
        self.n = len(self.lins)
        dev = self.lins[0].weight.device
        if dev.type != "cuda" or any(l.weight.dtype != torch.float32 or l.bias is None for l in self.lins):
            raise TypeError("RegressionStep needs an fp32 MLP with biases on the GPU")
        self.dev = dev
        self.dims = [self.lins[0].in_features] + [l.out_features for l in self.lins]
        # ONE flat gradient buffer; every parameter's .grad is a view of it (one all-reduce for data parallel runs)
        sizes = []
        for l in self.lins:
            sizes += [l.weight.numel(), l.bias.numel()]
        self.flat_grad = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        views, off = [], 0
        for l in self.lins:
            for p in (l.weight, l.bias):
                v = self.flat_grad[off:off + p.numel()].view_as(p); off += p.numel()
                p.grad = v
                views.append(v)
        self._gviews = views
        # per-layer buckets of the flat buffer ([W_l, b_l] are adjacent) + one event per layer for the data-parallel form
        self._buckets, off = [], 0
        for l in self.lins:
            nl = l.weight.numel() + l.bias.numel()
            self._buckets.append(self.flat_grad[off:off + nl]); off += nl
        self._events = self._comm = self._evp = None
        self._B = None
        self.loss = torch.zeros((), dtype=torch.float32, device=dev)

    def _buffers(self, B):
        if self._B != B:
            d, n, dev = self.dims, self.n, self.dev
            self.hidden = [torch.empty((B, d[l + 1]), dtype=torch.float32, device=dev) for l in range(n - 1)]
            self.preact = [torch.empty((B, d[l + 1]), dtype=torch.float32, device=dev) for l in range(n - 1)]
            self.g = torch.empty((B, d[n]), dtype=torch.float32, device=dev)
            self.ws = torch.empty(_lib.load().cfm_workspace_bytes(_lib.OP_MLP_TRAIN, B, max(d), max(d[l] * d[l + 1] for l in range(n))),
                                  dtype=torch.uint8, device=dev)
            self.hp = (ctypes.c_void_p * max(1, n - 1))(*([h.data_ptr() for h in self.hidden] or [0]))
            self.zp = (ctypes.c_void_p * max(1, n - 1))(*([z.data_ptr() for z in self.preact] or [0]))
            self.cd = (ctypes.c_int * (n + 1))(*d)
            self._B = B

    def _dp_setup(self):
        """events (created by a first record: torch makes the HIP event lazily) and the communication stream"""
        if self._events is None:
            cur = torch.cuda.current_stream(self.dev)
            self._events = [torch.cuda.Event() for _ in range(self.n)]
            for e in self._events:
                e.record(cur)
            self._evp = (ctypes.c_void_p * self.n)(*[e.cuda_event for e in self._events])
            self._comm = torch.cuda.Stream(device=self.dev)
        return self._evp

    def backward_only(self, t, xt, ut, layer_events=None):
        """forward + loss + backward; returns the loss (0-dim device tensor, overwritten by the next call).
        layer_events: ctypes array of n hipEvent_t (see ``cfm_mlp_regression_step_f32``: ``layer_done``) or None."""
        lib = _lib.load()
        xt = xt.detach().reshape(xt.shape[0], -1)
        ut = ut.detach().reshape(ut.shape[0], -1)
        B = xt.shape[0]
        tv = self.net.time_varying
        if xt.shape[1] != self.dims[0] - int(bool(tv)) or ut.shape[1] != self.dims[self.n]:
            raise RuntimeError(f"RegressionStep: xt has {xt.shape[1]} / ut has {ut.shape[1]} columns, the net maps "
                               f"{self.dims[0] - int(bool(tv))} (+ time) -> {self.dims[self.n]}")
        xt = _lib.to_dev_f32(xt, self.dev); ut = _lib.to_dev_f32(ut, self.dev)
        tt = _lib.to_dev_f32(t.detach().reshape(-1), self.dev) if tv else None
        if tv and tt.numel() != B:
            raise RuntimeError("RegressionStep: one time per row is required")
        self._buffers(B)
        n = self.n
        # the parameters' storage may have moved (load_state_dict keeps it, .to() does not): pointers are taken per call
        for v, p in zip(self._gviews, (q for l in self.lins for q in (l.weight, l.bias))):
            if p.grad is not v:
                p.grad = v
        Wp = (ctypes.c_void_p * n)(*[l.weight.data_ptr() for l in self.lins])
        bp = (ctypes.c_void_p * n)(*[l.bias.data_ptr() for l in self.lins])
        dWp = (ctypes.c_void_p * n)(*[self._gviews[2 * l].data_ptr() for l in range(n)])
        dbp = (ctypes.c_void_p * n)(*[self._gviews[2 * l + 1].data_ptr() for l in range(n)])
        check(lib.cfm_mlp_regression_step_f32(ptr(xt), ptr(tt), ptr(ut), Wp, bp, self.cd, n, B, self.hp, self.zp,
                                              ptr(self.g), dWp, dbp, ptr(self.loss), layer_events, ptr(self.ws),
                                              stream_ptr()),
              "cfm_mlp_regression_step_f32")
        return self.loss

    def __call__(self, t, xt, ut):
        import torch.distributed as dist
        have_pg = dist.is_available() and dist.is_initialized()
        world = dist.get_world_size() if have_pg else 1
        if self.data_parallel is False or not have_pg or (world <= 1 and not self.data_parallel):
            loss = self.backward_only(t, xt, ut)
            self.opt.step()
            return loss
        evp = self._dp_setup()
        loss = self.backward_only(t, xt, ut, layer_events=evp)       # asynchronous: the backward is still running
        works = []
        with torch.cuda.stream(self._comm):
            for l in range(self.n - 1, -1, -1):                      # the order the gradients become final in
                self._comm.wait_event(self._events[l])
                works.append(dist.all_reduce(self._buckets[l], async_op=True))
        for w in works:
            w.wait()                                                 # the compute stream waits for the buckets
        if self._fused_opt:
            self.opt.step(grad_scale=1.0 / world)                    # sum -> mean inside the Adam launch
        else:                                                        # any other optimizer: scale the buckets, then its own step
            self.flat_grad.mul_(1.0 / world)
            self.opt.step()
        return loss
