"""MI355X-native conditional flow matchers — drop-in for
``torchcfm.conditional_flow_matching`` (reference lines cited as ``ref:LINE`` into
``/root/reference/torchcfm/conditional_flow_matching.py``).

Same five classes, methods, argument meaning, return tuples, RNG consumption
order (``t`` from the CPU generator, then ``eps`` from x0's generator; the OT
samplers consume ``np.random`` first) and error behaviour.  The elementwise chain
(mu_t, sigma_t, xt, ut) and the OT-index gathers run in ONE fused HIP kernel
(``cfm_sample_xt_ut_f32``) that reproduces eager fp32 bit for bit.
"""
import math
import warnings
from typing import Union

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr
from .optimal_transport import OTPlanSampler, gather_rows


def pad_t_like_x(t, x):
    """Reshape the time vector t by the number of dimensions of x (ref:17-38)."""
    if isinstance(t, (float, int)):
        return t
    return t.reshape(-1, *([1] * (x.dim() - 1)))


def _fused_xt_ut(variant, sigma, x0, x1, t, eps, idx=None, xt_in=None, want_xt=True):
    """Run cfm_sample_xt_ut_f32.  x0/x1: [B,*dim] (any device), t: [B], eps like x0 or None.
    idx = (i, j) device int64 gathers or None.  Returns (xt, ut) on x0's device/dtype."""
    lib = _lib.load()
    dev = _lib.require_gpu()
    out_dev, out_dtype, shape = x0.device, x0.dtype, x0.shape
    B = shape[0]
    a0 = _lib.to_dev_f32(x0.reshape(x0.shape[0], -1), dev)
    a1 = _lib.to_dev_f32(x1.reshape(x1.shape[0], -1), dev)
    d = a0.shape[1]
    td = _lib.to_dev_f32(t.reshape(-1), dev)
    if td.numel() != B:
        raise AssertionError("t has to have batch size dimension")
    ed = _lib.to_dev_f32(eps.reshape(B, -1), dev) if eps is not None else None
    xin = _lib.to_dev_f32(xt_in.reshape(B, -1), dev) if xt_in is not None else None
    c0 = c1 = None
    if variant == _lib.VARIANT_VP:
        # cos/sin come from the tensor library on t's own device so they match eager bit for bit
        ang = math.pi / 2 * t.reshape(-1)
        c0 = _lib.to_dev_f32(torch.cos(ang), dev)
        c1 = _lib.to_dev_f32(torch.sin(ang), dev)
    elif variant == _lib.VARIANT_SB:
        # sigma_t exactly as the reference's compute_sigma_t (ref:446) on t's own device: eager
        # torch.sqrt is not correctly rounded on every backend, so the [B] scalars come from it
        tt = t.reshape(-1)
        c0 = _lib.to_dev_f32(sigma * torch.sqrt(tt * (1 - tt)), dev)
    xt = torch.empty((B, d), dtype=torch.float32, device=dev) if want_xt else None
    ut = torch.empty((B, d), dtype=torch.float32, device=dev)
    gi, gj = (idx if idx is not None else (None, None))
    check(lib.cfm_sample_xt_ut_f32(variant, ptr(a0), ptr(a1), ptr(gi), ptr(gj), ptr(td), ptr(ed),
                                   float(sigma), ptr(c0), ptr(c1), ptr(xin), B, d, ptr(xt), ptr(ut),
                                   ptr(None), ptr(None), stream_ptr()), "cfm_sample_xt_ut_f32")
    fin = lambda z: None if z is None else z.reshape(shape).to(device=out_dev, dtype=out_dtype)
    return fin(xt), fin(ut)


def _native(fn):
    """Marks the built-in (kernel-backed) implementation of an overridable method: a subclass that
    replaces any of them is detected by the absence of the mark."""
    fn._cfm_native = True
    return fn


_OVERRIDABLE = ("compute_mu_t", "compute_sigma_t", "sample_xt", "compute_conditional_flow")


class ConditionalFlowMatcher:
    """Independent conditional flow matching (ref:41-217)."""

    _variant = _lib.VARIANT_ICFM

    def __init__(self, sigma: Union[float, int] = 0.0):
        self.sigma = sigma

    # -- which path -----------------------------------------------------------------------------
    # The fused kernel evaluates the CLASS's closed forms in fp32 and does not build an autograd
    # graph.  The reference composes sample_location_and_conditional_flow out of the overridable
    # methods (ref:189-199) and keeps everything differentiable in the input dtype; so whenever a
    # subclass overrides one of them, an input requires grad, or the inputs are not fp32, the same
    # composition runs in eager torch instead (through the overrides).
    def _customised(self):
        cls = type(self)
        return any(not getattr(getattr(cls, m), "_cfm_native", False) for m in _OVERRIDABLE)

    @staticmethod
    def _needs_eager(*tensors):
        for z in tensors:
            if isinstance(z, torch.Tensor) and (z.dtype != torch.float32 or
                                                (z.requires_grad and torch.is_grad_enabled())):
                return True
        return False

    def _impl(self, name):
        """The user's override of `name`, or the eager restatement of the built-in one."""
        f = getattr(type(self), name)
        return getattr(self, name) if not getattr(f, "_cfm_native", False) else getattr(self, "_eager_" + name)

    # -- eager restatements of the built-in closed forms (all variants; reference operation order) --
    def _eager_compute_mu_t(self, x0, x1, t):
        t = pad_t_like_x(t, x0)
        v = self._variant
        if v == _lib.VARIANT_TARGET:
            return t * x1                                                    # ref:349-350
        if v == _lib.VARIANT_VP:
            return torch.cos(math.pi / 2 * t) * x0 + torch.sin(math.pi / 2 * t) * x1   # ref:588-589
        return t * x1 + (1 - t) * x0                                         # ref:82-83

    def _eager_compute_sigma_t(self, t):
        v = self._variant
        if v == _lib.VARIANT_TARGET:
            return 1 - (1 - self.sigma) * t                                  # ref:368
        if v == _lib.VARIANT_SB:
            return self.sigma * torch.sqrt(t * (1 - t))                      # ref:446
        return self.sigma                                                    # ref:101-102

    def _eager_sample_xt(self, x0, x1, t, epsilon):
        mu_t = self._impl("compute_mu_t")(x0, x1, t)                         # ref:126-129
        sigma_t = pad_t_like_x(self._impl("compute_sigma_t")(t), x0)
        return mu_t + sigma_t * epsilon

    def _eager_compute_conditional_flow(self, x0, x1, t, xt):
        v = self._variant
        if v == _lib.VARIANT_TARGET:
            tp = pad_t_like_x(t, x1)
            return (x1 - (1 - self.sigma) * xt) / (1 - (1 - self.sigma) * tp)    # ref:393-394
        if v == _lib.VARIANT_SB:
            tp = pad_t_like_x(t, x0)
            mu_t = self._impl("compute_mu_t")(x0, x1, tp)                    # ref:474-478
            return (1 - 2 * tp) / (2 * tp * (1 - tp) + 1e-8) * (xt - mu_t) + x1 - x0
        if v == _lib.VARIANT_VP:
            tp = pad_t_like_x(t, x0)
            return math.pi / 2 * (torch.cos(math.pi / 2 * tp) * x1 - torch.sin(math.pi / 2 * tp) * x0)   # ref:617-618
        return x1 - x0                                                       # ref:153-154

    # -- closed forms, each evaluated by the fused kernel --
    @_native
    def compute_mu_t(self, x0, x1, t):
        """t * x1 + (1 - t) * x0 (ref:62-83): the kernel's xt with eps = 0 (mu + 0 = mu)."""
        if self._needs_eager(x0, x1, t):
            return self._eager_compute_mu_t(x0, x1, t)
        xt, _ = _fused_xt_ut(self._variant, self.sigma, x0, x1, self._t_vec(t, x0),
                             torch.zeros_like(x0))
        return xt

    @_native
    def compute_sigma_t(self, t):
        """sigma (ref:85-102)."""
        del t
        return self.sigma

    @_native
    def sample_xt(self, x0, x1, t, epsilon):
        """mu_t + sigma_t * epsilon (ref:104-129)."""
        if self._customised() or self._needs_eager(x0, x1, t, epsilon):
            return self._eager_sample_xt(x0, x1, t, epsilon)
        xt, _ = _fused_xt_ut(self._variant, self.sigma, x0, x1, self._t_vec(t, x0), epsilon)
        return xt

    @_native
    def compute_conditional_flow(self, x0, x1, t, xt):
        """ut(x1|x0) (ref:131-154; overrides :370-394, :448-478, :591-618)."""
        if self._customised() or self._needs_eager(x0, x1, t, xt):
            return self._eager_compute_conditional_flow(x0, x1, t, xt)
        _, ut = _fused_xt_ut(self._variant, self.sigma, x0, x1, self._t_vec(t, x0), None, xt_in=xt,
                             want_xt=False)
        return ut

    def sample_noise_like(self, x):
        return torch.randn_like(x)

    @staticmethod
    def _t_vec(t, x):
        if isinstance(t, (float, int)):
            return torch.full((x.shape[0],), float(t), dtype=torch.float32)
        return t.reshape(-1)

    def _sample(self, x0, x1, t, return_noise, idx=None):
        # ref:189-199 — t from the CPU generator first, then eps from x0's generator
        if t is None:
            t = torch.rand(x0.shape[0]).type_as(x0)
        assert len(t) == x0.shape[0], "t has to have batch size dimension"
        if self._customised() or self._needs_eager(x0, x1, t):
            # the reference's composition, through the overridable methods; the OT pairing is a plain
            # (differentiable) index gather
            if idx is not None:
                i, j = idx
                x0, x1 = x0[i.to(x0.device)], x1[j.to(x1.device)]
            eps = self.sample_noise_like(x0)
            xt = self._impl("sample_xt")(x0, x1, t, eps)
            ut = self._impl("compute_conditional_flow")(x0, x1, t, xt)
        else:
            eps = self.sample_noise_like(x0)
            xt, ut = _fused_xt_ut(self._variant, self.sigma, x0, x1, t, eps, idx=idx)
        if return_noise:
            return t, xt, ut, eps
        return t, xt, ut

    def sample_location_and_conditional_flow(self, x0, x1, t=None, return_noise=False):
        """(t, xt, ut[, eps]) for the independent coupling (ref:159-199)."""
        return self._sample(x0, x1, t, return_noise)

    def compute_lambda(self, t):
        """2 * sigma_t / (sigma**2 + 1e-8) (ref:201-217)."""
        sigma_t = self.compute_sigma_t(t)
        return 2 * sigma_t / (self.sigma**2 + 1e-8)


class _OTMixin:
    """Shared body of the OT-coupled matchers (ref:241-316, :480-556)."""

    def sample_location_and_conditional_flow(self, x0, x1, t=None, return_noise=False):
        # ref:271-272 / :511-512 — couple, then sample; here the gather is fused into the
        # xt/ut kernel (the index pairs never leave the device).
        i, j = self.ot_sampler._sample_indices(x0, x1)
        return self._sample(x0, x1, t, return_noise, idx=(i, j))

    def sample_location_and_conditional_flow_group(self, batches, return_noise=False):
        """Several minibatches at once (not in the reference: its loop couples one minibatch per step, ref:271-272).

        ``batches``: ``[(x0, x1), ...]``.  Returns what one ``sample_location_and_conditional_flow`` call per pair
        returns, pair by pair, with the host RNG consumed in the same order (per pair: the plan-sampling uniforms, then
        ``t``, then the noise) — but the couplings are solved TOGETHER first: exact couplings of one square size share
        one chain of launches (``assign_exact_batch``), which costs little more than one solve.  The couplings of the
        next steps of a training loop depend on the data only, so a loop can ask for a few steps ahead
        (``cfm_amd.prefetch.CouplingPrefetcher.submit_group`` does it on a side stream)."""
        batches = list(batches)
        sols = self.ot_sampler._solve_many(batches, throughput=True)
        out = []
        for (x0, x1), sol in zip(batches, sols):
            i, j = self.ot_sampler._indices_from_solution(x0, x1, sol)
            out.append(self._sample(x0, x1, None, return_noise, idx=(i, j)))
        return out

    def guided_sample_location_and_conditional_flow(
        self, x0, x1, y0=None, y1=None, t=None, return_noise=False
    ):
        # ref:310-316 / :550-556
        i, j = self.ot_sampler._sample_indices(x0, x1)
        dev = i.device
        y0 = gather_rows(y0.detach().to(dev), i).to(y0.device) if y0 is not None else None
        y1 = gather_rows(y1.detach().to(dev), j).to(y1.device) if y1 is not None else None
        if return_noise:
            t, xt, ut, eps = self._sample(x0, x1, t, True, idx=(i, j))
            return t, xt, ut, y0, y1, eps
        t, xt, ut = self._sample(x0, x1, t, False, idx=(i, j))
        return t, xt, ut, y0, y1


class ExactOptimalTransportConditionalFlowMatcher(_OTMixin, ConditionalFlowMatcher):
    """OT-CFM: exact minibatch OT coupling (ref:220-316)."""

    def __init__(self, sigma: Union[float, int] = 0.0):
        super().__init__(sigma)
        self.ot_sampler = OTPlanSampler(method="exact")


class TargetConditionalFlowMatcher(ConditionalFlowMatcher):
    """Lipman et al. 2023 target conditional flow matching (ref:319-394)."""

    _variant = _lib.VARIANT_TARGET

    @_native
    def compute_sigma_t(self, t):
        """1 - (1 - sigma) t (ref:352-368)."""
        return 1 - (1 - self.sigma) * t


class SchrodingerBridgeConditionalFlowMatcher(_OTMixin, ConditionalFlowMatcher):
    """SB-CFM: entropic (or exact) OT coupling + Brownian-bridge path (ref:397-556)."""

    _variant = _lib.VARIANT_SB

    def __init__(self, sigma: Union[float, int] = 1.0, ot_method="exact"):
        if sigma <= 0:
            raise ValueError(f"Sigma must be strictly positive, got {sigma}.")
        elif sigma < 1e-3:
            warnings.warn("Small sigma values may lead to numerical instability.")
        super().__init__(sigma)
        self.ot_method = ot_method
        self.ot_sampler = OTPlanSampler(method=ot_method, reg=2 * self.sigma**2)

    @_native
    def compute_sigma_t(self, t):
        """sigma * sqrt(t (1 - t)) (ref:429-446)."""
        return self.sigma * torch.sqrt(t * (1 - t))


class VariancePreservingConditionalFlowMatcher(ConditionalFlowMatcher):
    """Albergo et al. 2023 trigonometric interpolants (ref:559-618)."""

    _variant = _lib.VARIANT_VP
