"""MLP vector field — drop-in for ``torchcfm.models.MLP`` (ref: torchcfm/models/models.py:4-21).

Same module tree (``self.net = Sequential(Linear, SELU, Linear, SELU, Linear, SELU,
Linear)``) so reference ``state_dict``s load unchanged.  Both directions run on the
fp32-MFMA HIP kernels: inference (what the ODE solve evaluates hundreds of times)
through ``cfm_mlp_forward_f32``, training through an ``autograd.Function`` over
``cfm_mlp_forward_train_f32`` / ``cfm_mlp_backward_f32`` (dgrad with fused SELU',
split-K wgrad, bias grads); ``cfm_amd.optim.FusedAdam`` is the one-launch optimizer step.
"""
import ctypes

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr


class _MLPTrainFunction(torch.autograd.Function):
    """forward / backward of Linear-SELU ... -Linear on the fp32-MFMA HIP kernels
    (cfm_mlp_forward_train_f32 / cfm_mlp_backward_f32).  Arguments: x [B, in], then W_0, b_0, W_1, ..."""

    @staticmethod
    def forward(ctx, x, *params):
        lib = _lib.load()
        dev = x.device
        n = len(params) // 2
        Ws = [params[2 * l].detach().contiguous() for l in range(n)]
        bs = [params[2 * l + 1].detach().contiguous() for l in range(n)]
        xd = x.detach().contiguous()
        B = xd.shape[0]
        dims = [Ws[0].shape[1]] + [w.shape[0] for w in Ws]
        hidden = [torch.empty((B, dims[l + 1]), dtype=torch.float32, device=dev) for l in range(n - 1)]
        preact = [torch.empty((B, dims[l + 1]), dtype=torch.float32, device=dev) for l in range(n - 1)]
        out = torch.empty((B, dims[n]), dtype=torch.float32, device=dev)
        Wp = (ctypes.c_void_p * n)(*[w.data_ptr() for w in Ws])
        bp = (ctypes.c_void_p * n)(*[b.data_ptr() for b in bs])
        hp = (ctypes.c_void_p * max(1, n - 1))(*([h.data_ptr() for h in hidden] or [0]))
        zp = (ctypes.c_void_p * max(1, n - 1))(*([z.data_ptr() for z in preact] or [0]))
        cd = (ctypes.c_int * (n + 1))(*dims)
        check(lib.cfm_mlp_forward_train_f32(ptr(xd), Wp, bp, cd, n, B, hp, zp, ptr(out), stream_ptr()),
              "cfm_mlp_forward_train_f32")
        # x and the parameters are saved as the INPUTS they are (graph attached): a double backward
        # (autograd.grad(..., create_graph=True): GradModel, divergences, Jacobians) re-derives the gradient
        # from them with differentiable torch ops; the HIP backward reads the detached views
        ctx.save_for_backward(x, *params, *hidden, *preact)
        ctx.dims, ctx.n = dims, n
        return out

    @staticmethod
    def backward(ctx, dout):
        n, dims = ctx.n, ctx.dims
        saved = ctx.saved_tensors
        x, params = saved[0], saved[1:1 + 2 * n]
        if torch.is_grad_enabled():
            # create_graph=True: the gradient itself must be differentiable (w.r.t. dout, x and the weights),
            # which the fused kernels are not — recompute the layer chain with torch ops and let autograd build
            # the second-order graph (exactly what the plain nn.Sequential of the reference would do)
            with torch.enable_grad():
                h = x
                for l in range(n):
                    h = torch.nn.functional.linear(h, params[2 * l], params[2 * l + 1])
                    if l < n - 1:
                        h = torch.nn.functional.selu(h)
                wanted = [k for k in range(1 + 2 * n) if ctx.needs_input_grad[k]]
                srcs = [x if k == 0 else params[k - 1] for k in wanted]
                got = torch.autograd.grad(h, srcs, dout, create_graph=True, allow_unused=True) if srcs else ()
            grads = [None] * (1 + 2 * n)
            for k, g in zip(wanted, got):
                grads[k] = g
            return tuple(grads)
        lib = _lib.load()
        hidden, preact = saved[1 + 2 * n:2 * n + n], saved[2 * n + n:]
        acts = (x.detach().contiguous(),) + tuple(hidden)
        Ws = [params[2 * l].detach().contiguous() for l in range(n)]
        dev = dout.device
        B = dout.shape[0]
        if tuple(dout.shape) != (acts[0].shape[0], dims[n]):
            raise RuntimeError(f"MLP backward: upstream gradient has shape {tuple(dout.shape)}, expected {(acts[0].shape[0], dims[n])}")
        dout = dout.contiguous().float()
        dW = [torch.empty_like(w) for w in Ws]
        db = [torch.empty(w.shape[0], dtype=torch.float32, device=dev) for w in Ws]
        dx = torch.empty_like(acts[0]) if ctx.needs_input_grad[0] else None
        maxw = max(dims)
        maxp = max(dims[l] * dims[l + 1] for l in range(n))
        ws = _lib.workspace(_lib.OP_MLP_TRAIN, B, maxw, maxp, dev)
        ap = (ctypes.c_void_p * n)(*[a.data_ptr() for a in acts])
        zp = (ctypes.c_void_p * n)(*([0] + [z.data_ptr() for z in preact]))
        Wp = (ctypes.c_void_p * n)(*[w.data_ptr() for w in Ws])
        dWp = (ctypes.c_void_p * n)(*[g.data_ptr() for g in dW])
        dbp = (ctypes.c_void_p * n)(*[g.data_ptr() for g in db])
        cd = (ctypes.c_int * (n + 1))(*dims)
        check(lib.cfm_mlp_backward_f32(ap, zp, Wp, cd, n, B, ptr(dout), dWp, dbp, ptr(dx), ptr(ws), stream_ptr()),
              "cfm_mlp_backward_f32")
        grads = [dx]
        for l in range(n):
            grads += [dW[l], db[l]]
        return tuple(grads)


class MLP(torch.nn.Module):
    def __init__(self, dim, out_dim=None, w=64, time_varying=False):
        super().__init__()
        self.time_varying = time_varying
        # in -> w -> w -> w -> out with SELU between the linear layers; the layers are created in
        # order (same parameter-init RNG consumption) and sit at net[0], net[2], net[4], net[6], so the
        # reference's state_dicts load unchanged
        widths = [dim + int(bool(time_varying)), w, w, w, dim if out_dim is None else out_dim]
        layers = []
        for k, (fan_in, fan_out) in enumerate(zip(widths[:-1], widths[1:])):
            if k:
                layers.append(torch.nn.SELU())
            layers.append(torch.nn.Linear(fan_in, fan_out))
        self.net = torch.nn.Sequential(*layers)

    # ---- HIP path -----------------------------------------------------------
    def _linears(self):
        return [m for m in self.net if isinstance(m, torch.nn.Linear)]

    def hip_params(self, device=None):
        """(W_ptrs, b_ptrs, dims, keepalive) as ctypes arrays for the C ABI."""
        dev = device or _lib.require_gpu()
        lins = self._linears()
        Ws = [_lib.to_dev_f32(l.weight, dev) for l in lins]
        bs = [_lib.to_dev_f32(l.bias, dev) for l in lins]
        n = len(lins)
        Wp = (ctypes.c_void_p * n)(*[w.data_ptr() for w in Ws])
        bp = (ctypes.c_void_p * n)(*[b.data_ptr() for b in bs])
        dims = (ctypes.c_int * (n + 1))(*([lins[0].in_features] + [l.out_features for l in lins]))
        return Wp, bp, dims, (Ws, bs)

    @torch.no_grad()
    def forward_hip(self, x, t=None):
        """Inference forward on the HIP kernels.

        x: [B, dim] (the time column NOT concatenated) with ``t`` a 0-dim / [B] tensor or
        float when the net is time varying; or x: [B, dim(+1)] with t=None (reference layout,
        time already in the last column — then the net is evaluated as a plain MLP).
        """
        lib = _lib.load()
        dev = _lib.require_gpu()
        Wp, bp, dims, keep = self.hip_params(dev)
        n = len(dims) - 1
        xd = _lib.to_dev_f32(x, dev)
        B = xd.shape[0]
        td, per_row = None, 0
        if t is not None:
            tt = torch.as_tensor(t, dtype=torch.float32)
            per_row = 1 if tt.numel() == B and tt.dim() > 0 and B > 1 else 0
            td = _lib.to_dev_f32(tt.reshape(-1), dev)
            if xd.shape[1] != dims[0] - 1:
                raise ValueError("x must not contain the time column when t is given")
        else:
            if xd.shape[1] != dims[0]:
                raise ValueError(f"expected {dims[0]} input features, got {xd.shape[1]}")
        if t is None:
            # plain MLP over the full input: describe it to the kernel as "no time column"
            use_dims = dims
        else:
            use_dims = dims
        maxw = max(dims[1:n]) if n > 1 else 1
        ws = _lib.workspace(_lib.OP_MLP, B, maxw, 0, dev)
        out = torch.empty((B, dims[n]), dtype=torch.float32, device=dev)
        check(lib.cfm_mlp_forward_f32(ptr(xd), ptr(td), per_row, Wp, bp, use_dims, n, B, ptr(out),
                                      ptr(ws), stream_ptr()), "cfm_mlp_forward_f32")
        return out.to(x.device)

    hip_training = True     # class switch: False sends the autograd path through PyTorch-ROCm (hipBLASLt)

    def forward(self, x):
        # ref:20-21.  fp32 CUDA tensors: forward AND backward on the fp32-MFMA HIP kernels (an
        # autograd.Function); no-grad inference: the inference kernels.  Anything else (CPU modules and
        # tensors in the CPU tests, other dtypes, double backward) is the plain module graph.
        if torch.is_grad_enabled() or not torch.cuda.is_available():
            if not x.is_cuda and not torch.is_grad_enabled():
                _lib.require_gpu()  # raises: no silent CPU inference path
            lins = self._linears()
            if (self.hip_training and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32
                    and all(l.weight.is_cuda and l.weight.dtype == torch.float32 and l.bias is not None for l in lins)):
                if x.shape[1] != lins[0].in_features:        # what nn.Linear would say (the kernel reads lda = in_features)
                    raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({x.shape[0]}x{x.shape[1]} and "
                                       f"{lins[0].in_features}x{lins[0].out_features})")
                same_dev = all(l.weight.device == x.device and l.bias.device == x.device for l in lins)
                chained = all(a.out_features == b.in_features for a, b in zip(lins[:-1], lins[1:]))
                if same_dev and chained:
                    params = []
                    for l in lins:
                        params += [l.weight, l.bias]
                    return _MLPTrainFunction.apply(x, *params)
            return self.net(x)
        # no-grad inference: any leading shape (the reference's nn.Sequential accepts [..., dim]) and the
        # caller's dtype back
        if x.dim() != 2:
            lead = x.shape[:-1]
            return self.forward_hip(x.reshape(-1, x.shape[-1])).reshape(*lead, -1).to(x.dtype)
        return self.forward_hip(x).to(x.dtype)


class GradModel(torch.nn.Module):
    """Vector field as the gradient of a scalar potential: ``forward(x)`` differentiates
    ``sum(action(x))`` w.r.t. ``x`` (graph kept, so the result can be trained through) and drops
    the last (time) column.  Counterpart of the action-matching helper at
    torchcfm/models/models.py:24-32; pure autograd, nothing to accelerate."""

    def __init__(self, action):
        super().__init__()
        self.action = action

    def forward(self, x):
        inp = x.requires_grad_(True)
        potential = self.action(inp).sum()
        (dpot,) = torch.autograd.grad(potential, inp, create_graph=True)
        return dpot[..., :-1] if dpot.dim() != 2 else dpot[:, :-1]
