"""SDE sampling for SF2M — counterpart of how the reference draws stochastic trajectories:
``torchsde.sdeint(SDE(model, score_model, sigma), x0, ts, method="euler")``
(examples/2D_tutorials/SF2M_tutorial.ipynb cell 5) and ``FlowSolver.sdeint`` / ``forward_sde_drift`` /
``backward_sde_drift`` (runner/src/models/components/solver.py:129-139,157-182).

``sdeint`` is fixed-step Euler-Maruyama on the ``ts`` grid refined to steps of at most ``dt``
(torchsde's Euler scheme).  When the drift and the score are ``cfm_amd.MLP`` fields the network
evaluations run on the HIP inference kernels and the state update is one fused HIP kernel per step
(``cfm_sde_em_step_f32``); any other callable pair is stepped by the same scheme in eager torch.
The Brownian increments come from ``torch.randn`` on the state's device (torchsde's BrownianInterval
stream is not reproducible without torchsde: the noise stream is NOT bit-compatible, the scheme is).
"""
import math

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr
from .models import MLP


class FlowScoreSDE(torch.nn.Module):
    """dy = (drift([y, t]) + score([y, t])) dt + sigma dW — the ``SDE`` class of the SF2M tutorial
    (``reverse=True``: backward drift -drift + score evaluated at 1 - t, solver.py:129-139,32-35)."""
    noise_type = "diagonal"
    sde_type = "ito"

    def __init__(self, drift, score, sigma=1.0, reverse=False):
        super().__init__()
        self.drift, self.score, self.sigma, self.reverse = drift, score, sigma, reverse

    def _cat(self, t, y):
        tt = torch.as_tensor(t, dtype=y.dtype, device=y.device)
        tt = tt.repeat(y.shape[0])[:, None] if tt.dim() == 0 else tt.reshape(-1, 1)
        return torch.cat([y, tt], 1)

    def f(self, t, y):
        if self.reverse:
            x = self._cat(1 - t, y)
            return -self.drift(x) + self.score(x)
        x = self._cat(t, y)
        return self.drift(x) + self.score(x)

    def g(self, t, y):
        s = self.sigma(t) if callable(self.sigma) else self.sigma
        return torch.ones_like(y) * s


def _grid(ts, dt):
    """The ts grid refined so that no step exceeds dt; returns [(t, h, is_output)]."""
    pts = [float(x) for x in ts]
    steps = []
    for a, b in zip(pts[:-1], pts[1:]):
        n = max(1, int(math.ceil(abs(b - a) / dt * (1.0 - 1e-6))))      # (ts usually arrives as float32: 0.05 is 0.0500000007)
        for k in range(n):
            steps.append((a + (b - a) * k / n, (b - a) / n, k == n - 1))
    return steps


def _fused_fields(sde):
    """(drift, score) as two 4-layer time-varying MLPs of widths <= 64 with identical layer sizes, or None."""
    f, s = sde.drift, sde.score
    if not (isinstance(f, MLP) and isinstance(s, MLP) and f.time_varying and s.time_varying):
        return None
    lf, ls = f._linears(), s._linears()
    df = [lf[0].in_features] + [l.out_features for l in lf]
    ds = [ls[0].in_features] + [l.out_features for l in ls]
    if len(lf) != 4 or df != ds or max(df[1:]) > 64 or df[0] != df[4] + 1:
        return None
    return f, s, df


@torch.no_grad()
def sdeint(sde, y0, ts, method="euler", dt=1e-3, generator=None, noise="philox", fused=None, **unused):
    """Euler-Maruyama trajectory [len(ts), B, d] of ``sde`` (an object with f(t, y), g(t, y)) from y0.

    Two small ``cfm_amd.MLP`` fields (4 layers, widths <= 64: the SF2M tutorial's and the single-cell models) run the
    WHOLE trajectory in one launch (``cfm_sde_em_mlp_f32``: weights of both fields in LDS, state in registers).
    ``noise="philox"`` (default): N(0, 1) from Philox4x32-10 inside the kernel, seeded from ``generator`` (or torch's
    default CUDA generator), so ``torch.manual_seed`` makes runs repeatable; ``noise="torch"``: the increments are
    drawn with ``torch.randn`` step by step exactly as the launch-per-step scheme draws them — same trajectory, bit for
    bit, as that scheme.  ``fused=False`` forces the launch-per-step scheme (measurement / test switch)."""
    if method != "euler":
        raise NotImplementedError(f"sdeint method {method!r}: only 'euler' (Euler-Maruyama) is built")
    if noise not in ("philox", "torch"):
        raise ValueError("noise must be 'philox' or 'torch'")
    steps = _grid(ts, float(dt))
    fast = (isinstance(sde, FlowScoreSDE) and isinstance(sde.drift, MLP) and isinstance(sde.score, MLP)
            and not callable(sde.sigma) and y0.dim() == 2 and torch.cuda.is_available())
    out = [y0]
    if fast:
        lib = _lib.load()
        dev = _lib.require_gpu()
        y = _lib.to_dev_f32(y0, dev).clone()
        fields = _fused_fields(sde) if (y.shape[1] <= 64 and fused is not False) else None
        if fields is not None and y.shape[1] != fields[2][4]:
            # the one-launch kernel reads y0 with the fields' output width as its pitch: a state of another width
            # must not reach it (the stepping path below raises the layer's shape error instead)
            if fused is True:
                raise ValueError(f"sdeint(fused=True): y0 has {y.shape[1]} columns, the fields map {fields[2][4]}")
            fields = None
        if fused is True and fields is None:
            raise ValueError("sdeint(fused=True): needs two 4-layer time-varying cfm_amd.MLP fields of widths <= 64")
        if fields is not None:
            import ctypes
            import struct
            f, s, dims = fields
            B, d = y.shape
            recs, n_out = [], 0
            for t, h, is_out in steps:
                te = (1.0 - t) if sde.reverse else t
                # the casts of the launch-per-step path: float32 time, float32 step, float32 (g sqrt|h|)
                recs.append(struct.pack("<fffi", te, h, float(sde.sigma) * math.sqrt(abs(h)), 1 if is_out else 0))
                n_out += 1 if is_out else 0
            host = (ctypes.c_char * (16 * len(steps))).from_buffer_copy(b"".join(recs))
            xi = None
            seed = 0
            if noise == "torch":
                xi = torch.stack([torch.randn((B, d), device=dev, dtype=torch.float32, generator=generator) for _ in steps])
            else:
                seed = int(torch.randint(0, 2 ** 62, (1,), device=dev, generator=generator).item())
            Wf, bf, _, keep_f = f.hip_params(dev)
            Ws, bs, _, keep_s = s.hip_params(dev)
            cd = (ctypes.c_int * 5)(*dims)
            traj = torch.empty((n_out, B, d), dtype=torch.float32, device=dev)
            ws = torch.empty(16 * len(steps) + 256, dtype=torch.uint8, device=dev)
            check(lib.cfm_sde_em_mlp_f32(Wf, bf, Ws, bs, cd, 4, ptr(y), B, host, len(steps), 1 if sde.reverse else 0,
                                         ptr(xi), seed, ptr(traj), ptr(ws), stream_ptr()), "cfm_sde_em_mlp_f32")
            torch.cuda.current_stream().synchronize()          # `host` (pageable) must outlive the copy
            return torch.cat([y0.to(torch.float32)[None].to(y0.device), traj.to(y0.device)])
        sign = 1.0
        for t, h, is_out in steps:
            te = (1.0 - t) if sde.reverse else t
            v = sde.drift.forward_hip(y, te)
            s = sde.score.forward_hip(y, te)
            if sde.reverse:
                v = -v
            xi = torch.randn(y.shape, device=dev, dtype=torch.float32, generator=generator)
            check(lib.cfm_sde_em_step_f32(ptr(y), ptr(v), ptr(s), ptr(xi), float(h), float(sde.sigma), sign,
                                          y.numel(), stream_ptr()), "cfm_sde_em_step_f32")
            if is_out:
                out.append(y.clone().to(y0.device))
        return torch.stack([o.to(torch.float32) for o in out])
    y = y0
    for t, h, is_out in steps:
        tt = torch.as_tensor(t, dtype=y.dtype, device=y.device)
        xi = torch.randn(y.shape, device=y.device, dtype=y.dtype, generator=generator)
        y = y + h * sde.f(tt, y) + sde.g(tt, y) * math.sqrt(abs(h)) * xi
        if is_out:
            out.append(y)
    return torch.stack(out)


class FlowSolver(torch.nn.Module):
    """Subset of runner/src/models/components/solver.py:44-230 on this backend: a flow field and an
    optional score field (separate networks, or one network whose output is [flow, score]) behind
    ``odeint`` (NeuralODE) and ``sdeint`` (Euler-Maruyama).  vector_field / score_field are called as
    f(t, x) like torchdyn vector fields."""

    def __init__(self, vector_field, dim, score_field=None, sigma=None, ode_solver="euler", sde_solver="euler",
                 dt=0.01, atol=1e-5, rtol=1e-5, **kwargs):
        super().__init__()
        self.net, self.dim, self.score_net = vector_field, dim, score_field
        self.separate_score = score_field is not None
        self.sigma, self.ode_solver, self.sde_solver = sigma, ode_solver, sde_solver
        self.dt, self.atol, self.rtol, self.nfe = dt, atol, rtol, 0

    def forward_flow_and_score(self, t, x, only_flow=False):       # solver.py:103-121
        if self.separate_score:
            vt, st = self.net(t, x), self.score_net(t, x)
        else:
            vtst = self.net(t, x)
            if vtst.shape[1] == x.shape[1]:
                return vtst
            k = vtst.shape[1] // 2
            vt, st = vtst[:, :k], vtst[:, k:]
        return vt if only_flow else (vt, st)

    def forward_sde_drift(self, t, x):                              # solver.py:123-127
        self.nfe += 1
        vt, st = self.forward_flow_and_score(t, x)
        return vt + st

    def backward_sde_drift(self, t, x):                             # solver.py:129-133
        self.nfe += 1
        vt, st = self.forward_flow_and_score(t, x)
        return -vt + st

    def forward_ode_drift(self, t, x):                              # solver.py:135-138
        self.nfe += 1
        return self.forward_flow_and_score(t, x, only_flow=True)

    def odeint(self, x0, t_span):
        from .ode import NeuralODE
        self.nfe = 0
        node = NeuralODE(self.forward_ode_drift, solver=self.ode_solver, atol=self.atol, rtol=self.rtol,
                         return_t_eval=False)
        return node(x0, t_span)

    def sdeint(self, x0, t_span, reverse=False, generator=None):
        self.nfe = 0
        outer = self

        class _S:
            def f(self, t, y):
                return outer.backward_sde_drift(1 - t, y) if reverse else outer.forward_sde_drift(t, y)

            def g(self, t, y):
                return outer.sigma(t) * torch.ones_like(y)
        return sdeint(_S(), x0, t_span, method=self.sde_solver, dt=self.dt, generator=generator)
