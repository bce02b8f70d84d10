"""Generate the golden fixtures in this directory.

Runs in the BUILD container only: it imports the UNMODIFIED reference from /root/reference
(oracle/ref_import.py puts the POT stand-in on the path) and records its outputs on seeded
inputs.  The fixtures travel to the GPU box; /root/reference does not.

    python tests/golden/make_golden.py

Fixture provenance
  fm_cases.npz        reference classes (torchcfm/conditional_flow_matching.py) — real reference
                      arithmetic for t/xt/ut/eps; OT pairs through the stand-in (SciPy LSAP /
                      restated Sinkhorn-Knopp).
  ot_cases.npz        reference OTPlanSampler.get_map / sample_plan / wasserstein (wrapper code
                      real, solver = stand-in).
  sinkhorn_cases.npz  oracle float64 log-domain Sinkhorn (POT loop semantics) + restated Knopp.
  ode_cases.npz       oracle torchdyn-style euler / dopri5 on a seeded MLP field.
  metrics_cases.npz   (round 2) reference runner metrics (compute_distribution_distances, mix_rbf_mmd2)
                      imported unmodified from runner/src/models/components/.
  ode2_cases.npz      (round 2) d = 50 / d = 784 fields and controller cases with rejected steps,
                      with the accept / reject logs of two independent restatements.
  ub_cases.npz        reference OTPlanSampler("unbalanced" / "partial") wrapper over the restated
                      POT loops (in-repo unbalanced statement; recalled partial Dykstra loop).
  refsk_cases.npz     (round 3) plans computed by the REFERENCE-HELD Sinkhorn code itself —
                      runner/src/models/components/sinkhorn_knopp_unbalanced.py imported unmodified
                      (pure NumPy) and run to its fixed point: unbalanced plans for three (reg, reg_m),
                      and balanced entropic plans through reg_m_1 = reg_m_2 = 1e12.
"""
import os
import sys
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cfm_oracle as oracle  # noqa: E402
import ref_import  # noqa: E402

SEED = 1994


def fm_cases(cfm):
    out = {}
    names = []
    B = 64
    specs = [
        ("i_cfm", lambda s: cfm.ConditionalFlowMatcher(sigma=s)),
        ("exact_ot_cfm", lambda s: cfm.ExactOptimalTransportConditionalFlowMatcher(sigma=s)),
        ("t_cfm", lambda s: cfm.TargetConditionalFlowMatcher(sigma=s)),
        ("vp_cfm", lambda s: cfm.VariancePreservingConditionalFlowMatcher(sigma=s)),
        ("sb_cfm_exact", lambda s: cfm.SchrodingerBridgeConditionalFlowMatcher(sigma=s, ot_method="exact")),
        ("sb_cfm_sinkhorn", lambda s: cfm.SchrodingerBridgeConditionalFlowMatcher(sigma=s, ot_method="sinkhorn")),
    ]
    for mname, ctor in specs:
        for sigma in ([0.0, 5e-4, 0.5, 1.5, 0, 1] if "sb" not in mname else [0.5, 1.5, 1]):
            if mname == "sb_cfm_sinkhorn" and sigma < 1.0:
                continue  # Knopp only alive for reg = 2 sigma^2 >~ 1 on this data
            for shape in ([2], [1, 2], [3, 4, 5]):
                if shape == [3, 4, 5] and sigma != 0.5 and not (mname.startswith('sb') and sigma == 1.5):
                    continue  # keep the fixture small: the 3-D shape once per method
                g = torch.Generator().manual_seed(zlib.crc32(f"{mname}|{sigma!r}|{shape}".encode()) % (2**31))
                x0 = torch.randn(B, *shape, generator=g)
                x1 = torch.randn(B, *shape, generator=g)
                fm = ctor(sigma)
                torch.manual_seed(SEED)
                np.random.seed(SEED)
                t, xt, ut, eps = fm.sample_location_and_conditional_flow(x0, x1, return_noise=True)
                key = f"{mname}|{sigma!r}|{'x'.join(map(str, shape))}"
                names.append(key)
                for k, v in (("x0", x0), ("x1", x1), ("t", t), ("xt", xt), ("ut", ut), ("eps", eps)):
                    out[f"{key}|{k}"] = v.numpy()
    out["names"] = np.array(names)
    return out


def ot_cases(ot):
    out = {}
    B = 128
    torch.manual_seed(1980)
    np.random.seed(1980)
    x0 = torch.randn(B, 2, 2, 2)
    x1 = torch.randn(B, 2, 2, 2)
    s = ot.OTPlanSampler(method="exact")
    pi = s.get_map(x0, x1)
    torch.manual_seed(1980)
    np.random.seed(1980)
    sx0, sx1 = s.sample_plan(x0, x1, replace=True)
    np.random.seed(7)
    i, j = s.sample_map(pi, B, replace=True)
    np.random.seed(7)
    i2, j2 = s.sample_map(pi, B, replace=False)
    y0 = torch.arange(B).reshape(B, 1)
    y1 = torch.arange(B).reshape(B, 1) + 1000
    np.random.seed(11)
    lx0, lx1, ly0, ly1 = s.sample_plan_with_labels(x0, x1, y0, y1)
    px0, px1 = s.sample_plan_with_scipy(x0, x1)
    out.update(x0=x0.numpy(), x1=x1.numpy(), perm=np.argmax(pi, 1), pi_nnz_value=pi.max(),
               sx0=sx0.numpy(), sx1=sx1.numpy(), map_i=i, map_j=j, map_i_norep=i2, map_j_norep=j2,
               ly0=ly0.numpy(), ly1=ly1.numpy(), lx0=lx0.numpy(), lx1=lx1.numpy(),
               scipy_x1=px1.numpy())
    out["W2_exact"] = ot.wasserstein(x0, x1, "exact")
    out["W1_exact"] = ot.wasserstein(x0, x1, "exact", power=1)
    out["W2_sinkhorn_reg1"] = ot.wasserstein(x0, x1, "sinkhorn", reg=1.0)
    # Sinkhorn sampler in the regime where Knopp is alive
    s2 = ot.OTPlanSampler(method="sinkhorn", reg=2.0)
    pik = s2.get_map(x0, x1)
    np.random.seed(5)
    ki, kj = s2.sample_map(pik, B)
    out.update(sk_i=ki, sk_j=kj, sk_rowsum=pik.sum(1), sk_colsum=pik.sum(0))
    # 8gaussians -> moons B=256 (config C1) permutation + cost
    a, b = oracle.config_inputs("C1")
    M = oracle.ref_cost_f32(a, b)
    perm = oracle.exact_perm(M)
    out.update(c1_x0=a.numpy(), c1_x1=b.numpy(), c1_perm=perm, c1_cost=oracle.assignment_cost(M, perm))
    return out


def sinkhorn_cases():
    out = {}
    g = torch.Generator().manual_seed(3)
    x0 = torch.randn(96, 3, generator=g)
    x1 = torch.randn(80, 3, generator=g) + 0.5
    M = oracle.ref_cost_f32(x0, x1)
    out["M"] = M
    for reg, iters in ((0.05, 25), (0.5, 40)):
        u, v, it, err = oracle.sinkhorn_log(M, reg, numItermax=iters, stopThr=0.0)
        out[f"u_{reg}"] = u; out[f"v_{reg}"] = v; out[f"it_{reg}"] = it; out[f"err_{reg}"] = err
    u, v, it, err = oracle.sinkhorn_log(M, 2.0)          # converges -> iteration count pinned
    out["u_conv"] = u; out["v_conv"] = v; out["it_conv"] = it; out["err_conv"] = err
    out["knopp_conv"] = oracle.sinkhorn_knopp(M, 2.0)
    return out


def ode_cases():
    out = {}
    torch.manual_seed(0)
    d, w, B = 2, 64, 64
    lins = [torch.nn.Linear(d + 1, w), torch.nn.Linear(w, w), torch.nn.Linear(w, w), torch.nn.Linear(w, d)]
    Ws = [l.weight.detach().numpy().copy() for l in lins]
    bs = [l.bias.detach().numpy().copy() for l in lins]
    for k, (W, b) in enumerate(zip(Ws, bs)):
        out[f"W{k}"] = W; out[f"b{k}"] = b
    x = oracle.eight_gaussians(B, 5).numpy()
    f = lambda t, y: oracle.mlp_forward_f64(Ws, bs, y, t)
    ts = np.linspace(0, 1, 11).astype(np.float32)
    out["x"] = x; out["t_span"] = ts
    out["euler"] = oracle.euler_trajectory(f, x, ts)
    traj, info = oracle.dopri5_trajectory(f, x, ts, 1e-4, 1e-4, return_log=True)
    out["dopri5"] = traj; out["dopri5_steps"] = info["steps"]; out["dopri5_nfe"] = info["nfe"]
    out["mlp_out"] = oracle.mlp_forward_f64(Ws, bs, x, 0.3)
    return out


def _seeded_mlp(d, w, seed):
    torch.manual_seed(seed)
    lins = [torch.nn.Linear(d + 1, w), torch.nn.Linear(w, w), torch.nn.Linear(w, w), torch.nn.Linear(w, d)]
    return [l.weight.detach().numpy().copy() for l in lins], [l.bias.detach().numpy().copy() for l in lins]


def ode2_cases():
    """Round 2: (i) a single-cell-shaped field (d = 50, w = 64: the fused small-field drivers);
    (ii) controller cases whose step sequence is NOT dictated by t_span — a field that varies fast in
    t, t_span = [0, 1], including one with rejected steps — each recorded with the accept / reject log
    of BOTH restatements (cfm_oracle.dopri5_trajectory: float64 state, float32 scalar controller;
    cfm_oracle.dopri5_trajectory_torch in float32 = what torchdyn does with float32 inputs, and in
    float64); (iii) a C3-shaped field (d = 784, w = 512: the layer-per-kernel path)."""
    out = {}
    # (i)
    Ws, bs = _seeded_mlp(50, 64, 11)
    for k, (W, b) in enumerate(zip(Ws, bs)):
        out[f"s_W{k}"] = W; out[f"s_b{k}"] = b
    x0, _ = oracle.config_inputs("C5", B=48)
    x = x0.numpy(); ts = np.linspace(0, 1, 7).astype(np.float32)
    f = lambda t, y: oracle.mlp_forward_f64(Ws, bs, y, t)
    out["s_x"] = x; out["s_t_span"] = ts
    out["s_euler"] = oracle.euler_trajectory(f, x, ts)
    tr, info = oracle.dopri5_trajectory(f, x, ts, 1e-4, 1e-4, return_log=True)
    out["s_dopri5"] = tr; out["s_steps"] = info["steps"]; out["s_nfe"] = info["nfe"]
    # (ii)
    W0, b0 = _seeded_mlp(2, 64, 0)
    xg = oracle.eight_gaussians(64, 5).numpy()
    out["c_x"] = xg
    for name, tw, ow, ts, tol in (("a", 30.0, 4.0, [0.0, 1.0], 1e-5), ("b", 80.0, 8.0, [0.0, 1.0], 1e-5),
                                  ("c", 200.0, 10.0, [0.0, 1.0], 1e-5), ("d", 200.0, 10.0, [0.0, 0.25, 0.5, 0.75, 1.0], 1e-4)):
        Wc = [w.copy() for w in W0]; Wc[0][:, 2] *= tw; Wc[3] *= ow
        ts = np.asarray(ts, dtype=np.float32)
        fc = lambda t, y: oracle.mlp_forward_f64(Wc, b0, y, t)
        tr, info = oracle.dopri5_trajectory(fc, xg, ts, tol, tol, return_log=True)
        out[f"c_{name}_tw"] = tw; out[f"c_{name}_ow"] = ow; out[f"c_{name}_t_span"] = ts; out[f"c_{name}_tol"] = tol
        out[f"c_{name}_traj"] = tr; out[f"c_{name}_steps"] = info["steps"]; out[f"c_{name}_nfe"] = info["nfe"]
        out[f"c_{name}_accept"] = np.array([l[3] for l in info["log"]])
        for dt, tag in ((torch.float32, "t32"), (torch.float64, "t64")):
            tr2, info2 = oracle.dopri5_trajectory_torch(oracle.mlp_field_torch(Wc, b0, dt), xg, ts, tol, tol, dtype=dt, return_log=True)
            out[f"c_{name}_{tag}_accept"] = np.array([l[3] for l in info2["log"]])
            out[f"c_{name}_{tag}_traj"] = tr2.double().numpy()
    for k, (W, b) in enumerate(zip(W0, b0)):
        out[f"c_W{k}"] = W; out[f"c_b{k}"] = b
    # (iii): weights are regenerated from the seed by the test (3 MB otherwise); only x and the results are stored
    Wl, bl = _seeded_mlp(784, 512, 5)
    g = torch.Generator().manual_seed(17)
    xl = torch.randn(24, 784, generator=g).numpy()
    tl = np.linspace(0, 1, 4).astype(np.float32)
    fl = lambda t, y: oracle.mlp_forward_f64(Wl, bl, y, t)
    out["l_x"] = xl; out["l_t_span"] = tl
    out["l_euler"] = oracle.euler_trajectory(fl, xl, tl).astype(np.float32)      # (float32 storage: 6e-8 << the 1e-5 parity bar)
    tr, info = oracle.dopri5_trajectory(fl, xl, tl, 1e-4, 1e-4, return_log=True)
    out["l_dopri5"] = tr.astype(np.float32); out["l_steps"] = info["steps"]; out["l_nfe"] = info["nfe"]
    out["l_W0_checksum"] = float(np.abs(Wl[0]).sum())         # guards the seed -> weights reproduction
    return out


def metrics_cases():
    """Round 2: the reference's own evaluation metrics (runner/src/models/components/
    distribution_distances.py, imported unmodified) on seeded point clouds: regular [B, times, d]
    tensors and a jagged list."""
    dd, mmd = ref_import.import_runner_metrics()
    out = {}
    g = torch.Generator().manual_seed(123)
    pred = torch.randn(96, 3, 5, generator=g)
    true = torch.randn(96, 3, 5, generator=g) * 1.3 + 0.4
    names, vals = dd.compute_distribution_distances(pred, true)
    out["pred"], out["true"] = pred.numpy(), true.numpy()
    out["names"] = np.array(names); out["values"] = np.array(vals, dtype=np.float64)
    jag = [torch.randn(n, 5, generator=g) + 0.2 * k for k, n in enumerate((96, 96, 96))]
    names_j, vals_j = dd.compute_distribution_distances(pred, jag)
    for k, t in enumerate(jag):
        out[f"jag{k}"] = t.numpy()
    out["names_jagged"] = np.array(names_j); out["values_jagged"] = np.array(vals_j, dtype=np.float64)
    one_p, one_t = pred[:, :1], true[:, :1]
    names_1, vals_1 = dd.compute_distribution_distances(one_p, one_t)
    out["names_single"] = np.array(names_1); out["values_single"] = np.array(vals_1, dtype=np.float64)
    out["rbf_only"] = float(mmd.mix_rbf_mmd2(pred[:, 0], true[:, 0], sigma_list=[0.5, 2.0]))
    return out


def ub_cases(ot):
    """Reference OTPlanSampler(method="unbalanced" | "partial") wrapper (real code) over the
    stand-in's restated POT loops, plus the docstring KAT of the in-repo unbalanced statement."""
    out = {}
    B = 96
    g = torch.Generator().manual_seed(77)
    x0 = torch.randn(B, 3, generator=g)
    x1 = torch.randn(B, 3, generator=g) * 0.8 + 0.7
    out["x0"], out["x1"] = x0.numpy(), x1.numpy()
    for reg, reg_m in ((0.5, 1.0), (1.0, 0.2), (0.3, 5.0)):
        s = ot.OTPlanSampler(method="unbalanced", reg=reg, reg_m=reg_m)
        out[f"unb_pi_{reg}_{reg_m}"] = s.get_map(x0, x1)
        np.random.seed(3)
        i, j = s.sample_map(out[f"unb_pi_{reg}_{reg_m}"], B)
        out[f"unb_i_{reg}_{reg_m}"], out[f"unb_j_{reg}_{reg_m}"] = i, j
    for reg in (0.5, 2.0):
        s = ot.OTPlanSampler(method="partial", reg=reg)
        out[f"par_pi_{reg}"] = s.get_map(x0, x1)
        np.random.seed(4)
        i, j = s.sample_map(out[f"par_pi_{reg}"], B)
        out[f"par_i_{reg}"], out[f"par_j_{reg}"] = i, j
    # rectangular
    x2 = torch.randn(64, 3, generator=g) + 0.3
    out["x2"] = x2.numpy()
    out["unb_rect"] = ot.OTPlanSampler(method="unbalanced", reg=0.7, reg_m=1.0).get_map(x0, x2)
    out["par_rect"] = ot.OTPlanSampler(method="partial", reg=0.7).get_map(x0, x2)
    # partial with m < 1 (POT API below the wrapper)
    M = oracle.ref_cost_f32(x0, x1)
    out["par_m06"] = oracle.entropic_partial_wasserstein(M, 0.5, m=0.6)
    out["kat_unbalanced"] = oracle.sinkhorn_knopp_unbalanced([[0.0, 1.0], [1.0, 0.0]], 1.0, 1.0, a=[0.5, 0.5], b=[0.5, 0.5])
    return out


REFSK_UNBALANCED = ((0.5, 1.0), (1.0, 0.2), (0.3, 5.0))
REFSK_BALANCED = (1.0, 2.0, 5.0)
REFSK_BALANCED_8G = (2.0, 5.0)
REFSK_REG_M_INF = 1e12


def refsk_inputs():
    g = torch.Generator().manual_seed(77)
    x0 = torch.randn(96, 3, generator=g)
    x1 = torch.randn(96, 3, generator=g) * 0.8 + 0.7
    y0, y1 = oracle.config_inputs("C1", B=96)          # 8 gaussians -> moons (d = 2), the tutorial clouds
    return x0, x1, y0, y1


def refsk_cases():
    """Fixed points of the reference-held loop (stopThr far below the comparison tolerance, so the loop's own
    stopping rule — relative change of u / v, not POT's marginal error — plays no role)."""
    sk = ref_import.import_runner_sinkhorn()
    x0, x1, y0, y1 = refsk_inputs()
    out = {"x0": x0.numpy(), "x1": x1.numpy(), "y0": y0.numpy(), "y1": y1.numpy()}
    M = oracle.ref_cost_f32(x0, x1); M8 = oracle.ref_cost_f32(y0, y1)
    out["M"], out["M8"] = M, M8
    kw = dict(numItermax=200000, stopThr=1e-15)
    for reg, reg_m in REFSK_UNBALANCED:
        out[f"ub_{reg}_{reg_m}"] = sk.sinkhorn_knopp_unbalanced([], [], M, reg, reg_m, reg_m, **kw)
    for reg in REFSK_BALANCED:
        out[f"bal_{reg}"] = sk.sinkhorn_knopp_unbalanced([], [], M, reg, REFSK_REG_M_INF, REFSK_REG_M_INF, **kw)
    for reg in REFSK_BALANCED_8G:
        out[f"bal8g_{reg}"] = sk.sinkhorn_knopp_unbalanced([], [], M8, reg, REFSK_REG_M_INF, REFSK_REG_M_INF, **kw)
    # the docstring KAT through the real function
    out["kat"] = sk.sinkhorn_knopp_unbalanced([0.5, 0.5], [0.5, 0.5], [[0.0, 1.0], [1.0, 0.0]], 1.0, 1.0, 1.0)
    return out


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "refsk":      # reference-held NumPy Sinkhorn, imported unmodified
        np.savez_compressed(os.path.join(HERE, "refsk_cases.npz"), **refsk_cases())
        print("refsk_cases.npz", os.path.getsize(os.path.join(HERE, "refsk_cases.npz")))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "metrics":    # the reference's runner metrics, imported unmodified
        np.savez_compressed(os.path.join(HERE, "metrics_cases.npz"), **metrics_cases())
        print("metrics_cases.npz", os.path.getsize(os.path.join(HERE, "metrics_cases.npz")))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "ode2":       # oracle-only fixtures: no reference import needed
        np.savez_compressed(os.path.join(HERE, "ode2_cases.npz"), **ode2_cases())
        print("ode2_cases.npz", os.path.getsize(os.path.join(HERE, "ode2_cases.npz")))
        return
    cfm, ot = ref_import.import_reference()
    np.savez_compressed(os.path.join(HERE, "ub_cases.npz"), **ub_cases(ot))
    np.savez_compressed(os.path.join(HERE, "refsk_cases.npz"), **refsk_cases())
    np.savez_compressed(os.path.join(HERE, "fm_cases.npz"), **fm_cases(cfm))
    np.savez_compressed(os.path.join(HERE, "ot_cases.npz"), **ot_cases(ot))
    np.savez_compressed(os.path.join(HERE, "sinkhorn_cases.npz"), **sinkhorn_cases())
    np.savez_compressed(os.path.join(HERE, "ode_cases.npz"), **ode_cases())
    np.savez_compressed(os.path.join(HERE, "ode2_cases.npz"), **ode2_cases())
    np.savez_compressed(os.path.join(HERE, "metrics_cases.npz"), **metrics_cases())
    np.savez_compressed(os.path.join(HERE, "metrics_cases.npz"), **metrics_cases())
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
