"""ORACLE — CPU restatement of the reference algorithm for the OT-coupling + CFM-sampling
hot path.  TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg as the checker; never by the product path
(conditional-flow-matching_amd/ must not import this file).

Every function cites the reference lines it follows (paths relative to the reference repo
root, /root/reference in the build container).  Third-party pieces absent from the
reference tree are restated from their published algorithm:
  * POT (``ot``; unpinned in setup.py:14) — emd -> exact LSAP (equal uniform marginals =>
    permutation plan; SciPy's linear_sum_assignment is the solver the reference itself
    calls at optimal_transport.py:170,179), sinkhorn_knopp / sinkhorn_log loops
    (SURVEY.md Appendix A.2);
  * NumPy legacy RandomState.choice (Appendix A.3);
  * torchdyn >= 1.0.6 NeuralODE / dopri5 (Appendix A.4) — "torchdyn-style dopri5".

Pinning status: the closed forms, RNG order and wrapper semantics are
pinned against the reference's own code, imported from /root/reference with the ``ot``
stand-in, through tests/golden/*.npz (tests/golden/make_golden.py).  The numeric values
of POT's solvers and torchdyn's integrator cannot be pinned here (neither library is in
the image): exact-OT parity is pinned to SciPy's LSAP optimum instead; Sinkhorn and ODE
parity are "parity unpinned" against the third-party code and pinned only to this
restatement.
"""
import math
import warnings

import numpy as np
import torch
from scipy.optimize import linear_sum_assignment


# ----------------------------------------------------------------------------- data
def eight_gaussians(n, seed):
    """8 Gaussians, scale 5, var 0.1 (ref: torchcfm/utils.py:11-32,40-41) — seeded generator."""
    g = torch.Generator().manual_seed(seed)
    s = 1.0 / math.sqrt(2.0)
    centers = torch.tensor([(1, 0), (-1, 0), (0, 1), (0, -1), (s, s), (s, -s), (-s, s), (-s, -s)],
                           dtype=torch.float32) * 5
    noise = torch.randn(n, 2, generator=g) * math.sqrt(math.sqrt(0.1))
    idx = torch.randint(0, 8, (n,), generator=g)
    return (centers[idx] + noise).float()


def two_moons(n, seed):
    """torchdyn generate_moons(n, noise=0.2) * 3 - 1 (ref: torchcfm/utils.py:35-37; A.4)."""
    rng = np.random.RandomState(seed)
    n_out = n // 2
    n_in = n - n_out
    to, ti = np.linspace(0, np.pi, n_out), np.linspace(0, np.pi, n_in)
    X = np.vstack([np.c_[np.cos(to), np.sin(to)], np.c_[1 - np.cos(ti), 0.5 - np.sin(ti)]])
    X += rng.rand(n, 1) * 0.2
    return torch.from_numpy(X.astype(np.float32)) * 3 - 1


def mnist_like(n, seed):
    """Synthetic MNIST-shaped target (no dataset in the image): clip(0.35*randn + mu_k, -1, 1)
    with 10 fixed class means in [-1,1]^784."""
    g = torch.Generator().manual_seed(seed)
    mu = torch.rand(10, 784, generator=g) * 2 - 1
    k = torch.randint(0, 10, (n,), generator=g)
    return torch.clamp(0.35 * torch.randn(n, 784, generator=g) + mu[k], -1, 1)


def gaussian_source(n, d, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, d, generator=g)


def pca_like_pair(n, d, seed):
    """C5: randn*s0 vs rotated randn*s1 + m with an EB-PCA-like spectrum s_k ~ k^-1/2."""
    g = torch.Generator().manual_seed(seed)
    s = torch.arange(1, d + 1, dtype=torch.float32).rsqrt() * 3
    x0 = torch.randn(n, d, generator=g) * s
    Q, _ = torch.linalg.qr(torch.randn(d, d, generator=g))
    x1 = (torch.randn(n, d, generator=g) * s * 1.2) @ Q + 0.5 * torch.randn(1, d, generator=g)
    return x0.contiguous(), x1.contiguous()


def config_inputs(name, B=None, rank=0):
    """Seeded synthetic inputs of the BASELINE configs."""
    if name == "C1":
        B = B or 256
        return eight_gaussians(B, 0 + rank), two_moons(B, 0 + rank)
    if name == "C2":
        B = B or 4096
        return eight_gaussians(B, 0 + rank), two_moons(B, 0 + rank)
    if name in ("C3", "C4"):
        B = B or 4096
        return gaussian_source(B, 784, 1000 + rank), mnist_like(B, 2000 + rank)
    if name == "C5":
        B = B or 8192
        return pca_like_pair(B, 50, 2024 + rank)
    raise ValueError(name)


# ----------------------------------------------------------------------------- cost (K1)
def sqeuclid_cost_f64(x0, x1):
    """sum_k (x0_ik - x1_jk)^2 in float64 (ref: torch.cdist(x0,x1)**2, optimal_transport.py:84)."""
    a = np.asarray(x0, dtype=np.float64).reshape(len(x0), -1)
    b = np.asarray(x1, dtype=np.float64).reshape(len(x1), -1)
    if a.shape[1] <= 16:
        return ((a[:, None, :] - b[None, :, :]) ** 2).sum(-1)
    # float64 Gram form of the centred clouds (the translation changes no distance and keeps the
    # norms at the scale of the cloud); the few entries where even that cancels more than six
    # digits (duplicates, an x-vs-x diagonal) are redone as plain differences
    mu = (a.sum(0) + b.sum(0)) / (len(a) + len(b))
    a, b = a - mu, b - mu
    na, nb = (a * a).sum(1), (b * b).sum(1)
    M = np.maximum(na[:, None] + nb[None, :] - 2.0 * (a @ b.T), 0.0)
    ii, jj = np.nonzero(M < 1e-6 * (na[:, None] + nb[None, :]))
    for s0 in range(0, len(ii), 65536):
        i, j = ii[s0:s0 + 65536], jj[s0:s0 + 65536]
        M[i, j] = ((a[i] - b[j]) ** 2).sum(1)
    return M


def ref_cost_f32(x0, x1):
    """The reference's own fp32 matrix: torch.cdist(x0, x1) ** 2 on CPU (optimal_transport.py:84)."""
    a = torch.as_tensor(x0).reshape(len(x0), -1)
    b = torch.as_tensor(x1).reshape(len(x1), -1)
    return (torch.cdist(a, b) ** 2).numpy()


# ----------------------------------------------------------------------------- exact OT (K4)
def exact_perm(M):
    """Optimal permutation of a square cost matrix given as fp32 values (solved in float64),
    i.e. the support of pot.emd(unif, unif, M) (optimal_transport.py:49,87) and exactly
    scipy.optimize.linear_sum_assignment(M) of :179."""
    r, c = linear_sum_assignment(np.asarray(M, dtype=np.float64))
    assert np.array_equal(r, np.arange(len(r)))
    return c.astype(np.int64)


def exact_plan_rect(M):
    """pot.emd(unif(B0), unif(B1), M) for B0 != B1 (optimal_transport.py:49,79,87): the
    transportation problem with masses 1/B0, 1/B1 is the assignment problem on the cost matrix
    expanded to lcm(B0, B1) unit masses; solved with SciPy's LSAP in float64.  Returns (plan, cost);
    the cost is unique, the plan need not be."""
    M = np.asarray(M, dtype=np.float64)
    B0, B1 = M.shape
    L = B0 * B1 // math.gcd(B0, B1)
    ri = np.repeat(np.arange(B0), L // B0)
    ci = np.repeat(np.arange(B1), L // B1)
    r, c = linear_sum_assignment(M[ri][:, ci])
    pi = np.zeros((B0, B1))
    np.add.at(pi, (ri[r], ci[c]), 1.0 / L)
    return pi, float((pi * M).sum())


def perm_plan(perm):
    """pot.emd's plan for uniform equal marginals: 1/B on the optimal permutation."""
    B = len(perm)
    G = np.zeros((B, B), dtype=np.float64)
    G[np.arange(B), perm] = 1.0 / B
    return G


def assignment_cost(M, perm):
    M = np.asarray(M, dtype=np.float64)
    return float(M[np.arange(len(perm)), perm].sum())


# ----------------------------------------------------------------------------- sampling (K6)
def choice_flat(p_flat, u):
    """np.random.choice(len(p), p=p, size=len(u)) given its uniforms (A.3):
    cdf = p.cumsum(); cdf /= cdf[-1]; idx = cdf.searchsorted(u, side='right')."""
    cdf = np.cumsum(p_flat)
    cdf /= cdf[-1]
    return cdf.searchsorted(u, side="right")


def sample_map_given_u(pi, u):
    """OTPlanSampler.sample_map with replace=True (optimal_transport.py:116-121) for given
    uniforms u (what np.random.choice would draw with np.random.random_sample(len(u)))."""
    p = pi.flatten()
    p = p / p.sum()
    return np.divmod(choice_flat(p, u), pi.shape[1])


def sample_map_reference(pi, batch_size):
    """OTPlanSampler.sample_map as the reference runs it (optimal_transport.py:116-121): flatten the
    dense plan, renormalise, np.random.choice over all B0 * B1 entries (consumes batch_size uniforms
    of the global np.random stream), divmod.  O(B^2); used by bench.py's cpu_baseline leg and to pin
    sample_map_given_u / sample_perm_given_u."""
    p = pi.flatten()
    p = p / p.sum()
    choices = np.random.choice(pi.shape[0] * pi.shape[1], p=p, size=batch_size, replace=True)
    return np.divmod(choices, pi.shape[1])


def sample_perm_given_u(perm, u):
    """O(B) restatement for a permutation plan: only the B non-zeros matter."""
    B = len(perm)
    p = np.full(B, 1.0 / B)
    p = p / p.sum()
    i = choice_flat(p, u)
    return i.astype(np.int64), np.asarray(perm)[i].astype(np.int64)


# ----------------------------------------------------------------------------- Sinkhorn (K5)
def _lse(X, axis):
    m = X.max(axis=axis, keepdims=True)
    return (m + np.log(np.exp(X - m).sum(axis=axis, keepdims=True))).squeeze(axis)


def sinkhorn_log(M, reg, numItermax=1000, stopThr=1e-9, check_every=10):
    """POT ot.bregman.sinkhorn_log loop (A.2) in float64, uniform marginals.
    Returns (u, v, n_iter, err): log-scalings with plan = exp(-M/reg + u_i + v_j)."""
    M = np.asarray(M, dtype=np.float64)
    n, m = M.shape
    loga, logb = math.log(1.0 / n), math.log(1.0 / m)
    Mr = -M / reg
    u = np.zeros(n)
    v = np.zeros(m)
    err, it = 1.0, numItermax
    for ii in range(numItermax):
        v = logb - _lse(Mr + u[:, None], 0)
        u = loga - _lse(Mr + v[None, :], 1)
        if ii % check_every == 0:
            tmp2 = np.exp(Mr + u[:, None] + v[None, :]).sum(0)
            err = float(np.linalg.norm(tmp2 - 1.0 / m))
            if err < stopThr:
                it = ii + 1
                break
    return u, v, it, err


def sinkhorn_plan(M, reg, u, v):
    M = np.asarray(M, dtype=np.float64)
    return np.exp(-M / reg + u[:, None] + v[None, :])


def sinkhorn_knopp(M, reg, numItermax=1000, stopThr=1e-9):
    """POT ot.sinkhorn default (method='sinkhorn' -> sinkhorn_knopp), A.2, uniform marginals.
    M arrives as float32 values (optimal_transport.py:87)."""
    M = np.asarray(M)
    n, m = M.shape
    a = np.full(n, 1.0 / n)
    b = np.full(m, 1.0 / m)
    u = np.ones(n, dtype=M.dtype) / n
    v = np.ones(m, dtype=M.dtype) / m
    K = np.exp(M / (-reg))
    Kp = (1.0 / a).reshape(-1, 1) * K
    for ii in range(numItermax):
        uprev, vprev = u, v
        KtU = K.T @ u
        with np.errstate(divide="ignore", invalid="ignore"):
            v = b / KtU
            u = 1.0 / (Kp @ v)
        if (np.any(KtU == 0) or np.any(np.isnan(u)) or np.any(np.isnan(v))
                or np.any(np.isinf(u)) or np.any(np.isinf(v))):
            warnings.warn("Warning: numerical errors at iteration %d" % ii)
            u, v = uprev, vprev
            break
        if ii % 10 == 0:
            tmp2 = np.einsum("i,ij,j->j", u, K, v)
            if np.linalg.norm(tmp2 - b) < stopThr:
                break
    return u.reshape(-1, 1) * K * v.reshape(1, -1)


def sinkhorn_knopp_unbalanced(M, reg, reg_m, numItermax=1000, stopThr=1e-6, a=None, b=None, log=False):
    """pot.unbalanced.sinkhorn_knopp_unbalanced(a, b, M, reg, reg_m) (optimal_transport.py:52-53)
    as the reference authors restate it in runner/src/models/components/
    sinkhorn_knopp_unbalanced.py:113-201 (reg_m_1 = reg_m_2 = reg_m); float64, uniform marginals
    unless given.  KAT of its docstring (:88-94): a=b=[.5,.5], M=[[0,1],[1,0]], reg=reg_m=1 ->
    [[0.51122814, 0.18807032], [0.18807032, 0.51122814]]."""
    M = np.asarray(M, dtype=np.float64)
    dim_a, dim_b = M.shape
    a = np.full(dim_a, 1.0 / dim_a) if a is None else np.asarray(a, dtype=np.float64)
    b = np.full(dim_b, 1.0 / dim_b) if b is None else np.asarray(b, dtype=np.float64)
    u = np.ones(dim_a) / dim_a
    v = np.ones(dim_b) / dim_b
    K = np.empty(M.shape, dtype=M.dtype)
    np.divide(M, -reg, out=K)
    np.exp(K, out=K)
    fi = reg_m / (reg_m + reg)
    cpt, err, status = 0, 1.0, 0
    while err > stopThr and cpt < numItermax:
        uprev, vprev = u, v
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            Kv = K.dot(v)
            u = (a / Kv) ** fi
            Ktu = K.T.dot(u)
            v = (b / Ktu) ** fi
        if (np.any(Ktu == 0.0) or np.any(np.isnan(u)) or np.any(np.isnan(v))
                or np.any(np.isinf(u)) or np.any(np.isinf(v))):
            warnings.warn("Numerical errors at iteration %s" % cpt)
            u, v, status = uprev, vprev, 1
            break
        if cpt % 10 == 0:
            err_u = abs(u - uprev).max() / max(abs(u).max(), abs(uprev).max(), 1.0)
            err_v = abs(v - vprev).max() / max(abs(v).max(), abs(vprev).max(), 1.0)
            err = 0.5 * (err_u + err_v)
        cpt += 1
    G = u[:, None] * K * v[None, :]
    return (G, {"iters": cpt, "status": status, "err": err}) if log else G


def entropic_partial_wasserstein(M, reg, m=None, numItermax=1000, stopThr=1e-100, a=None, b=None,
                                 log=False):
    """pot.partial.entropic_partial_wasserstein(a, b, M, reg) (optimal_transport.py:54-55):
    POT's Dykstra loop written out with its three full correction matrices q1, q2, q3
    ([RECALLED] from POT 0.8/0.9 ot/partial.py — POT is absent from the reference tree and not
    installable here: parity unpinned); float64, uniform marginals unless given."""
    M = np.asarray(M, dtype=np.float64)
    dim_a, dim_b = M.shape
    a = np.full(dim_a, 1.0 / dim_a) if a is None else np.asarray(a, dtype=np.float64)
    b = np.full(dim_b, 1.0 / dim_b) if b is None else np.asarray(b, dtype=np.float64)
    dx, dy = np.ones(dim_a), np.ones(dim_b)
    if m is None:
        m = min(np.sum(a), np.sum(b)) * 1.0
    if m < 0:
        raise ValueError("Problem infeasible. Parameter m should be greater than 0.")
    if m > min(np.sum(a), np.sum(b)) * (1 + 1e-15):
        raise ValueError("Problem infeasible. Parameter m should lower or equal than min(|a|_1, |b|_1).")
    K = np.empty(M.shape, dtype=M.dtype)
    np.divide(M, -reg, out=K)
    np.exp(K, out=K)
    np.multiply(K, m / np.sum(K), out=K)
    err, cpt, status = 1, 0, 0
    q1, q2, q3 = np.ones(K.shape), np.ones(K.shape), np.ones(K.shape)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        while err > stopThr and cpt < numItermax:
            Kprev = K
            K = K * q1
            K1 = np.dot(np.diag(np.minimum(a / np.sum(K, axis=1), dx)), K)
            q1 = q1 * Kprev / K1
            K1prev = K1
            K1 = K1 * q2
            K2 = np.dot(K1, np.diag(np.minimum(b / np.sum(K1, axis=0), dy)))
            q2 = q2 * K1prev / K2
            K2prev = K2
            K2 = K2 * q3
            K = K2 * (m / np.sum(K2))
            q3 = q3 * K2prev / K
            if np.any(np.isnan(K)) or np.any(np.isinf(K)):
                print("Warning: numerical errors at iteration", cpt)
                status = 1
                break
            if cpt % 10 == 0:
                err = np.linalg.norm(Kprev - K)
            cpt = cpt + 1
    return (K, {"iters": cpt, "status": status, "err": err}) if log else K


# ----------------------------------------------------------------------------- xt / ut (K8)
def pad_t_like_x(t, x):
    if isinstance(t, (float, int)):
        return t
    return t.reshape(-1, *([1] * (x.dim() - 1)))


def xt_ut(method, x0, x1, t, eps, sigma):
    """Closed forms in eager fp32, reference operation order
    (conditional_flow_matching.py:82-83,126-129,153-154 | :446,474-478 | :349-350,368,393-394
    | :588-589,617-618).  method in {icfm, sb, target, vp}."""
    tp = pad_t_like_x(t, x0)
    if method == "icfm":
        mu = tp * x1 + (1 - tp) * x0
        xt = mu + sigma * eps
        ut = x1 - x0
    elif method == "sb":
        sig_t = pad_t_like_x(sigma * torch.sqrt(t * (1 - t)), x0)
        mu = tp * x1 + (1 - tp) * x0
        xt = mu + sig_t * eps
        ratio = (1 - 2 * tp) / (2 * tp * (1 - tp) + 1e-8)
        ut = ratio * (xt - mu) + x1 - x0
    elif method == "target":
        mu = tp * x1
        sig_t = pad_t_like_x(1 - (1 - sigma) * t, x0)
        xt = mu + sig_t * eps
        ut = (x1 - (1 - sigma) * xt) / (1 - (1 - sigma) * tp)
    elif method == "vp":
        mu = torch.cos(math.pi / 2 * tp) * x0 + torch.sin(math.pi / 2 * tp) * x1
        xt = mu + sigma * eps
        ut = math.pi / 2 * (torch.cos(math.pi / 2 * tp) * x1 - torch.sin(math.pi / 2 * tp) * x0)
    else:
        raise ValueError(method)
    return xt, ut


# ----------------------------------------------------------------------------- whole step
def ot_cfm_step(x0, x1, sigma=0.0, method="exact", reg=0.05, M=None):
    """OTPlanSampler(method).sample_plan + ConditionalFlowMatcher.sample_location_and_conditional_flow
    (optimal_transport.py:123-145; conditional_flow_matching.py:159-199,241-272) on CPU tensors,
    consuming np.random / torch RNG in the reference order.  `M` overrides the cost matrix."""
    B = x0.shape[0]
    if M is None:
        M = ref_cost_f32(x0, x1)
    if method == "exact":
        perm = exact_perm(M)
        u = np.random.random_sample(B)
        i, j = sample_perm_given_u(perm, u)
    else:
        uu, vv, _, _ = sinkhorn_log(M, reg)
        pi = sinkhorn_plan(M, reg, uu, vv)
        u = np.random.random_sample(B)
        i, j = sample_map_given_u(pi, u)
    a0, a1 = x0[i], x1[j]
    t = torch.rand(B).type_as(x0)
    eps = torch.randn_like(a0)
    xt, ut = xt_ut("icfm" if method == "exact" else "sb", a0, a1, t, eps, sigma)
    return t, xt, ut, eps, (i, j)


# ----------------------------------------------------------------------------- MLP (K10)
SELU_SCALE = 1.0507009873554804934193349852946
SELU_ALPHA = 1.6732632423543772848170429916717


def mlp_forward_f64(weights, biases, x, t=None):
    """MLP.forward through torch_wrapper (models.py:10-21, utils.py:51-52) in float64.
    weights[l]: [out,in] arrays.  t: None, scalar or [B]."""
    h = np.asarray(x, dtype=np.float64)
    if t is not None:
        tt = np.broadcast_to(np.asarray(t, dtype=np.float64).reshape(-1, 1), (h.shape[0], 1))
        h = np.concatenate([h, tt], 1)
    n = len(weights)
    for l, (W, b) in enumerate(zip(weights, biases)):
        h = h @ np.asarray(W, dtype=np.float64).T + np.asarray(b, dtype=np.float64)
        if l != n - 1:
            h = SELU_SCALE * np.where(h > 0, h, SELU_ALPHA * np.expm1(h))
    return h


def mlp_backward_f64(weights, biases, x, dout, preact=None):
    """Gradients of sum(out * dout) for the Linear-SELU x3 + Linear network (models.py:10-21) in
    float64: what autograd computes for the reference's loss.backward() (train_cifar10.py:149).
    x: [B, in] (time column already concatenated).  Returns (out, dW list, db list, dx).

    SELU' jumps at 0 (scale vs scale * alpha), so a pre-activation within rounding of 0 lands on
    different sides in float32 and float64 and the gradients then differ by O(1 / B) — in ANY
    float32 implementation, PyTorch's included.  `preact` (the float32 forward's own hidden
    pre-activations, one array per hidden layer) pins the branch: SELU' is then evaluated at those
    values, which is what a float32 backward is entitled to be compared with."""
    h = [np.asarray(x, dtype=np.float64)]
    zs = []
    n = len(weights)
    for l, (W, b) in enumerate(zip(weights, biases)):
        z = h[-1] @ np.asarray(W, dtype=np.float64).T + np.asarray(b, dtype=np.float64)
        zs.append(z)
        h.append(SELU_SCALE * np.where(z > 0, z, SELU_ALPHA * np.expm1(z)) if l != n - 1 else z)
    g = np.asarray(dout, dtype=np.float64)
    dW, db = [None] * n, [None] * n
    for l in range(n - 1, -1, -1):
        dW[l] = g.T @ h[l]
        db[l] = g.sum(0)
        g = g @ np.asarray(weights[l], dtype=np.float64)
        if l > 0:
            zz = zs[l - 1] if preact is None else np.asarray(preact[l - 1], dtype=np.float64)
            g = g * (SELU_SCALE * np.where(zz > 0, 1.0, SELU_ALPHA * np.exp(np.minimum(zz, 0.0))))
    return h[-1], dW, db, g


def adam_step_f64(p, g, m, v, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """torch.optim.Adam's update (amsgrad=False) in float64: the reference's optimizer
    (train_cifar10.py:93,150).  Returns the new (p, m, v)."""
    p, g, m, v = (np.asarray(z, dtype=np.float64) for z in (p, g, m, v))
    if weight_decay:
        g = g + weight_decay * p
    m = m + (1 - beta1) * (g - m)
    v = beta2 * v + (1 - beta2) * g * g
    bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
    denom = np.sqrt(v) / math.sqrt(bc2) + eps
    return p - (lr / bc1) * m / denom, m, v


# ----------------------------------------------------------------------------- ODE (K11)
DP_C = [1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0]
DP_A = [
    [1 / 5],
    [3 / 40, 9 / 40],
    [44 / 45, -56 / 15, 32 / 9],
    [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
    [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656],
    [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84],
]
DP_BSOL = [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0]
DP_BALT = [1951 / 21600, 0, 22642 / 50085, 451 / 720, -12231 / 42400, 649 / 6300, 1 / 60]


def euler_trajectory(f, x, t_span):
    """torchdyn fixed-step Euler on t_span (A.4): x <- x + dt f(t, x).  f works in float64."""
    ts = np.asarray(t_span, dtype=np.float32)
    x = np.asarray(x, dtype=np.float64)
    sol = [x]
    for k in range(len(ts) - 1):
        dt = float(np.float32(ts[k + 1] - ts[k]))
        x = x + dt * f(float(ts[k]), x)
        sol.append(x)
    return np.stack(sol)


def _hn(x):
    return math.sqrt(float(np.mean(np.square(x))))


def dopri5_trajectory(f, x, t_span, atol, rtol, return_log=False):
    """torchdyn-style adaptive Dormand-Prince 5(4): Hairer init_step, FSAL,
    global RMS error norm over the batch, every t_span point is a step end, adapt_step with
    safety 0.9 / min 0.2 / max 10 / order 5.  State in float64, the scalar controller (t, dt,
    error ratio) in float32 exactly like conditional-flow-matching_amd/csrc/ode.hip."""
    f32 = np.float32
    ts = np.asarray(t_span, dtype=np.float32)
    x = np.asarray(x, dtype=np.float64)
    atol, rtol = float(f32(atol)), float(f32(rtol))
    sol = [x]
    nfe = 0

    def ev(t, y):
        nonlocal nfe
        nfe += 1
        return f(float(t), y)

    t, T = f32(ts[0]), f32(ts[-1])
    k1 = ev(t, x)
    scale = atol + np.abs(x) * rtol
    d0, d1 = f32(_hn(x / scale)), f32(_hn(k1 / scale))
    h0 = f32(1e-6) if (d0 < f32(1e-5) or d1 < f32(1e-5)) else f32(f32(0.01) * d0 / d1)
    f1 = ev(f32(t + h0), x + float(h0) * k1)
    d2 = f32(f32(_hn((f1 - k1) / scale)) / h0)
    if d1 <= f32(1e-15) and d2 <= f32(1e-15):
        h1 = max(f32(1e-6), f32(h0 * f32(1e-3)))
    else:
        h1 = f32(np.power(f32(f32(0.01) / max(d1, d2)), f32(1.0) / f32(6.0)))
    dt = f32(min(f32(f32(100) * h0), h1))
    ckpt, steps, log = 1, 0, []
    while t < T:
        if f32(t + dt) > T:
            dt = f32(T - t)
        dt_old, flag = dt, False
        if ckpt < len(ts) and f32(t + dt) > ts[ckpt]:
            dt_old, flag, dt = dt, True, f32(ts[ckpt] - t)
        lands = ckpt < len(ts) and (flag or f32(t + dt) == ts[ckpt])
        ks = [k1]
        y = x
        for s in range(6):
            y = x + float(dt) * sum(float(f32(a)) * k for a, k in zip(DP_A[s], ks))
            ks.append(ev(f32(t + f32(DP_C[s]) * dt), y))
        x_new = y
        err = float(dt) * sum(float(f32(bs - ba)) * k for bs, ba, k in zip(DP_BSOL, DP_BALT, ks))
        ratio = f32(_hn(err / (atol + rtol * np.maximum(np.abs(x), np.abs(x_new)))))
        steps += 1
        accept = ratio <= f32(1)
        log.append((float(t), float(dt), float(ratio), bool(accept)))
        if accept:
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
            minf = f32(1.0) if ratio < f32(1) else f32(0.2)
            factor = min(f32(10), max(f32(f32(0.9) / np.power(ratio, f32(0.2))), minf))
        dt = f32(dt * factor)
        if not dt > f32(1e-12):
            dt = f32(1e-12)
    out = np.stack(sol)
    return (out, {"steps": steps, "nfe": nfe, "log": log}) if return_log else out


# ----------------------------------------------------------------------------- ODE, second restatement
def dopri5_trajectory_torch(f, x, t_span, atol, rtol, dtype=torch.float64, return_log=False):
    """A SECOND, independent restatement of torchdyn's adaptive odeint, written as
    torchdyn itself is: eager torch tensors, with the state, t, dt and the error ratio all carried
    in ONE dtype (torchdyn inherits the dtype of x; the reference feeds it float32, and float64
    gives the reference-arithmetic answer).  It shares no code with dopri5_trajectory above (whose
    scalar controller deliberately runs in float32 like the device driver) — tests compare the
    accept / reject sequences of the two and of the HIP driver.  f(t, y) -> dy, torch tensors."""
    x = torch.as_tensor(x).to(dtype)
    t_span = torch.as_tensor(t_span).to(dtype)
    atol_, rtol_ = torch.tensor(float(np.float32(atol)), dtype=dtype), torch.tensor(float(np.float32(rtol)), dtype=dtype)
    c = torch.tensor(DP_C, dtype=dtype)
    a = [torch.tensor(r, dtype=dtype) for r in DP_A]
    bsol = torch.tensor(DP_BSOL, dtype=dtype)
    berr = torch.tensor([s - e for s, e in zip(DP_BSOL, DP_BALT)], dtype=dtype)

    def hairer_norm(z):
        return z.abs().pow(2).mean().sqrt()

    nfe = 0

    def ev(t, y):
        nonlocal nfe
        nfe += 1
        return f(t, y).to(dtype)

    t, T = t_span[0], t_span[-1]
    k1 = ev(t, x)
    # init_step (Hairer II.4), order 5
    scale = atol_ + x.abs() * rtol_
    d0, d1 = hairer_norm(x / scale), hairer_norm(k1 / scale)
    h0 = torch.tensor(1e-6, dtype=dtype) if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    f1 = ev(t + h0, x + h0 * k1)
    d2 = hairer_norm((f1 - k1) / scale) / h0
    if d1 <= 1e-15 and d2 <= 1e-15:
        h1 = torch.max(torch.tensor(1e-6, dtype=dtype), h0 * 1e-3)
    else:
        h1 = (0.01 / torch.max(d1, d2)) ** (1.0 / 6.0)
    dt = torch.min(100 * h0, h1)

    t_eval, ckpt = t_span[1:], 0
    sol, log, steps = [x], [], 0
    while t < T:
        if t + dt > T:
            dt = T - t
        dt_old, flag = dt, False
        if ckpt < len(t_eval) and t + dt > t_eval[ckpt]:
            dt_old, flag = dt, True
            dt = t_eval[ckpt] - t
        # one Dormand-Prince step (FSAL: k1 carried over)
        ks = [k1]
        y = x
        for s in range(6):
            y = x + dt * sum(a[s][q] * ks[q] for q in range(s + 1))
            ks.append(ev(t + c[s] * dt, y))
        x_new = y
        x_err = dt * sum(berr[q] * ks[q] for q in range(7))
        error_scaled = x_err / (atol_ + rtol_ * torch.max(x.abs(), x_new.abs()))
        error_ratio = hairer_norm(error_scaled)
        accept = bool(error_ratio <= 1)
        steps += 1
        log.append((float(t), float(dt), float(error_ratio), accept))
        if accept:
            lands = ckpt < len(t_eval) and (flag or bool(t + dt == t_eval[ckpt]))
            if lands:
                t = t_eval[ckpt]; sol.append(x_new); ckpt += 1
            else:
                t = t + dt
            x, k1 = x_new, ks[6]
        if flag:
            dt = dt_old - dt
        # adapt_step(safety 0.9, min 0.2, max 10, order 5)
        if error_ratio == 0:
            dt = dt * 10.0
        else:
            min_factor = 1.0 if error_ratio < 1 else 0.2
            factor = min(10.0, max(float(0.9 / error_ratio ** 0.2), min_factor))
            dt = dt * factor
        if not dt > 1e-12:
            dt = torch.tensor(1e-12, dtype=dtype)
    out = torch.stack(sol)
    return (out, {"steps": steps, "nfe": nfe, "log": log}) if return_log else out


def mlp_field_torch(weights, biases, dtype=torch.float64):
    """torch_wrapper(MLP) as a torch callable f(t, y) in `dtype` (models.py:10-21, utils.py:51-52)."""
    Ws = [torch.as_tensor(np.asarray(W)).to(dtype) for W in weights]
    bs = [torch.as_tensor(np.asarray(b)).to(dtype) for b in biases]

    def f(t, y):
        h = torch.cat([y.to(dtype), torch.as_tensor(t, dtype=dtype).reshape(1, 1).expand(y.shape[0], 1)], 1)
        for l, (W, b) in enumerate(zip(Ws, bs)):
            h = h @ W.T + b
            if l != len(Ws) - 1:
                h = torch.nn.functional.selu(h)
        return h
    return f
