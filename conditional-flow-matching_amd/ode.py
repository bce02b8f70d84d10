"""ODE sampling — counterpart of ``torchdyn.core.NeuralODE`` as the reference uses it
(``NeuralODE(torch_wrapper(model), solver=..., sensitivity="adjoint", atol, rtol)
.trajectory(x, t_span)``: examples/2D_tutorials/Flow_matching_tutorial.ipynb cells 11/16,
examples/images/cifar10/utils_cifar.py:63-68).

When the vector field is ``torch_wrapper(MLP(time_varying=True))`` the whole solve runs in
the HIP drivers (``cfm_ode_euler_mlp_f32`` / ``cfm_ode_dopri5_mlp_f32``).  Any other
vector field (e.g. a UNet) is stepped by the same algorithm at the tensor level — host
control flow only, the field itself runs wherever the user's module runs.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr
from .models import MLP
from .utils import torch_wrapper

_DP_C = [1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0]
_DP_A = [
    [1 / 5],
    [3 / 40, 9 / 40],
    [44 / 45, -56 / 15, 32 / 9],
    [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
    [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656],
    [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84],
]
_DP_BSOL = [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0]
_DP_BALT = [1951 / 21600, 0, 22642 / 50085, 451 / 720, -12231 / 42400, 649 / 6300, 1 / 60]


def _hairer_norm(x):
    return x.abs().pow(2).mean().sqrt()


class NeuralODE(torch.nn.Module):
    def __init__(self, vector_field, solver="dopri5", order=1, atol=1e-3, rtol=1e-3,
                 sensitivity="autograd", return_t_eval=True, **kwargs):
        super().__init__()
        if solver not in ("euler", "dopri5"):
            raise NotImplementedError(f"solver {solver!r}: only 'euler' and 'dopri5' are built")
        self.vf = vector_field
        self.solver, self.atol, self.rtol = solver, float(atol), float(rtol)
        self.sensitivity = sensitivity          # inert without autograd through the solve
        self.return_t_eval = return_t_eval
        self.nfe = 0
        self.n_steps = 0

    # ---- dispatch ----
    def _hip_mlp(self):
        vf = self.vf
        if isinstance(vf, torch_wrapper) and isinstance(vf.model, MLP) and vf.model.time_varying:
            m = vf.model
            lins = m._linears()
            if lins[-1].out_features + 1 == lins[0].in_features:
                return m
        return None

    @torch.no_grad()
    def trajectory(self, x, t_span):
        m = self._hip_mlp()
        if m is not None and x.dim() == 2:
            return self._trajectory_hip(m, x, t_span)
        return self._trajectory_generic(x, t_span)

    def forward(self, x, t_span):
        sol = self.trajectory(x, t_span)
        return (t_span, sol) if self.return_t_eval else sol

    # ---- HIP drivers ----
    def _trajectory_hip(self, m, x, t_span):
        lib = _lib.load()
        dev = _lib.require_gpu()
        Wp, bp, dims, keep = m.hip_params(dev)
        n = len(dims) - 1
        xd = _lib.to_dev_f32(x, dev)
        B, d = xd.shape
        ts = np.ascontiguousarray(torch.as_tensor(t_span, dtype=torch.float32).cpu().numpy())
        n_t = ts.shape[0]
        traj = torch.empty((n_t, B, d), dtype=torch.float32, device=dev)
        maxw = max(dims[1:n]) if n > 1 else 1
        ws = _lib.workspace(_lib.OP_ODE, B, maxw, d, dev)
        nfe = ctypes.c_int(0)
        steps = ctypes.c_int(0)
        tsp = ts.ctypes.data_as(ctypes.c_void_p)
        if self.solver == "euler":
            check(lib.cfm_ode_euler_mlp_f32(Wp, bp, dims, n, ptr(xd), B, tsp, n_t, ptr(traj),
                                            ctypes.byref(nfe), ptr(ws), stream_ptr()),
                  "cfm_ode_euler_mlp_f32")
            steps.value = n_t - 1
        else:
            check(lib.cfm_ode_dopri5_mlp_f32(Wp, bp, dims, n, ptr(xd), B, tsp, n_t, self.atol,
                                             self.rtol, ptr(traj), ctypes.byref(steps),
                                             ctypes.byref(nfe), ptr(ws), stream_ptr()),
                  "cfm_ode_dopri5_mlp_f32")
        self.nfe, self.n_steps = nfe.value, steps.value
        return traj.to(x.device)

    # ---- generic vector fields: same algorithm at tensor level ----
    def _trajectory_generic(self, x, t_span):
        f = self.vf
        ts = torch.as_tensor(t_span, dtype=torch.float32)
        sol = [x]
        self.nfe = 0

        def ev(t, y):
            self.nfe += 1
            return f(torch.as_tensor(t, dtype=torch.float32, device=y.device), y)

        if self.solver == "euler":
            for k in range(len(ts) - 1):
                dt = float(ts[k + 1] - ts[k])
                x = x + dt * ev(float(ts[k]), x)
                sol.append(x)
            self.n_steps = len(ts) - 1
            return torch.stack(sol)
        atol, rtol = self.atol, self.rtol
        f32 = np.float32
        t, T = f32(ts[0]), f32(ts[-1])
        k1 = ev(t, x)
        scale = atol + x.abs() * rtol
        d0, d1 = f32(_hairer_norm(x / scale)), f32(_hairer_norm(k1 / scale))
        h0 = f32(1e-6) if (d0 < 1e-5 or d1 < 1e-5) else f32(0.01) * d0 / d1
        f1 = ev(t + h0, x + h0 * k1)
        d2 = f32(_hairer_norm((f1 - k1) / scale)) / h0
        if d1 <= 1e-15 and d2 <= 1e-15:
            h1 = max(f32(1e-6), h0 * f32(1e-3))
        else:
            h1 = f32(f32(0.01) / max(d1, d2)) ** f32(1.0 / 6.0)
        dt = f32(min(f32(100) * h0, h1))
        ckpt, steps = 1, 0
        while t < T:
            if t + dt > T:
                dt = f32(T - t)
            dt_old, flag = dt, False
            if ckpt < len(ts) and t + dt > f32(ts[ckpt]):
                dt_old, flag, dt = dt, True, f32(f32(ts[ckpt]) - t)
            lands = ckpt < len(ts) and (flag or t + dt == f32(ts[ckpt]))
            ks = [k1]
            for s in range(6):
                y = x + float(dt) * sum(float(a) * k for a, k in zip(_DP_A[s], ks))
                ks.append(ev(t + f32(_DP_C[s]) * dt, y))
            x_new = y
            err = float(dt) * sum(float(bs - ba) * k for bs, ba, k in zip(_DP_BSOL, _DP_BALT, ks))
            ratio = f32(_hairer_norm(err / (atol + rtol * torch.max(x.abs(), x_new.abs()))))
            steps += 1
            if ratio <= 1:
                if lands:
                    t = f32(ts[ckpt]); sol.append(x_new); ckpt += 1
                else:
                    t = f32(t + dt)
                x, k1 = x_new, ks[6]
            if flag:
                dt = f32(dt_old - dt)
            if ratio == 0:
                factor = f32(10)
            else:
                minf = f32(1.0) if ratio < 1 else f32(0.2)
                factor = min(f32(10), max(f32(0.9) / ratio ** f32(0.2), minf))
            dt = f32(dt * factor)
            if not dt > 1e-12:
                dt = f32(1e-12)
        self.n_steps = steps
        return torch.stack(sol)
