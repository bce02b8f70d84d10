"""One-launch Adam — counterpart of ``torch.optim.Adam`` as the reference's training loops use it
(``torch.optim.Adam(model.parameters())``: examples/2D_tutorials/*.ipynb, ``lr=2e-4`` in
examples/images/cifar10/train_cifar10.py:93).  Same state (``step``, ``exp_avg``, ``exp_avg_sq``),
same arithmetic, one ``cfm_adam_step_f32`` launch for all fp32 CUDA parameters of a group instead of
~10 foreach kernels.  amsgrad / maximize / capturable / sparse gradients are not built: they raise."""
import ctypes
import struct

import torch

from . import _lib
from ._lib import check, stream_ptr


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False,
                 maximize=False):
        if amsgrad or maximize:
            raise NotImplementedError("FusedAdam: amsgrad / maximize are not built")
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._tables = {}

    def _table(self, gi, ps, states):
        """Device pointer table of a group; rebuilt only when a pointer changed (set_to_none gradients
        come back at new addresses, so the grads are part of the key)."""
        key = tuple((p.data_ptr(), p.grad.data_ptr(), s["exp_avg"].data_ptr(), s["exp_avg_sq"].data_ptr(), p.numel())
                    for p, s in zip(ps, states))
        cached = self._tables.get(gi)
        if cached is None or cached[0] != key:
            raw = b"".join(struct.pack("<QQQQQ", *k) for k in key)
            host = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
            cached = (key, host.to(ps[0].device, non_blocking=False))
            self._tables[gi] = cached
        return cached[1]

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        """grad_scale: every gradient is multiplied by it first, inside the same launch, and left scaled in ``.grad``
        (data parallel: the all-reduced sum -> the mean; ``RegressionStep`` passes 1 / world size)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            for p in ps:
                if p.grad.is_sparse or not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise NotImplementedError("FusedAdam handles dense, contiguous fp32 CUDA parameters")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if not p.grad.is_contiguous():
                    p.grad = p.grad.contiguous()
            states = [self.state[p] for p in ps]
            steps = {int(s["step"]) for s in states}
            if len(steps) != 1:
                raise NotImplementedError("FusedAdam: the parameters of a group must share their step count")
            step = steps.pop() + 1
            table = self._table(gi, ps, states)
            b1, b2 = group["betas"]
            check(lib.cfm_adam_step_f32(ctypes.c_void_p(table.data_ptr()), len(ps), float(group["lr"]), float(b1),
                                        float(b2), float(group["eps"]), float(group["weight_decay"]), step,
                                        float(grad_scale), stream_ptr()), "cfm_adam_step_f32")
            for s in states:
                s["step"] = step
        return loss
