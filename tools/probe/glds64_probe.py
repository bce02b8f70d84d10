"""Round 6: the MLP layers on the 64 x 64 direct-to-LDS engine (gemm_glds64.h) against the register-staged core and an
fp64 reference, at the C3 shapes and at edge shapes; then the timing of both (tools/gemm_quick.py's figures).
Measurement infrastructure."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import cfm_amd
from cfm_amd import _lib
lib = _lib.load(); dev = _lib.require_gpu(); torch.manual_seed(0)


def ref64(net, x, t):
    with torch.no_grad():
        h = torch.cat([x.double().cpu(), t.double().cpu()[:, None]], 1) if t is not None else x.double().cpu()
        mods = [m for m in net.net]
        for m in mods:
            if isinstance(m, torch.nn.Linear):
                h = h @ m.weight.double().cpu().T + m.bias.double().cpu()
            else:
                h = torch.nn.functional.selu(h)
    return h


worst = 0.0
for (B, d, w, tv) in [(4096, 784, 512, True), (4096, 784, 512, False), (1000, 48, 64, True), (130, 16, 80, True), (257, 32, 512, False), (64, 784, 128, True)]:
    net = cfm_amd.MLP(dim=d, time_varying=tv, w=w).to(dev)
    x = torch.randn(B, d, device=dev); t = torch.rand(B, device=dev) if tv else None
    xin = x if tv else x
    outs = {}
    for mode in (0, 1, 2):
        lib.cfm_mlp_set_glds(mode)
        with torch.no_grad():
            outs[mode] = net.forward_hip(xin, t).double().cpu()
    r = ref64(net, x, t)
    sc = float(r.abs().max())
    e = {m: float((outs[m] - r).abs().max()) / sc for m in outs}
    worst = max(worst, e[1], e[2])
    print(f"B={B} d={d} w={w} tv={tv}: rel err vs fp64  core {e[0]:.2e}  glds {e[1]:.2e}  glds+unaligned {e[2]:.2e}   glds vs core max diff {float((outs[1]-outs[0]).abs().max()):.2e}", flush=True)
print("worst", worst)
assert worst < 1e-5

# training step parity: loss and gradients of the fused regression step, mode 0 vs 1 vs 2
B, d = 4096, 784
a = torch.randn(B, d, device=dev); b = torch.clamp(0.35 * torch.randn(B, d, device=dev) + 0.2, -1, 1); t = torch.rand(B, device=dev)
res = {}
for mode in (0, 1, 2):
    lib.cfm_mlp_set_glds(mode)
    torch.manual_seed(1)
    net = cfm_amd.MLP(dim=d, time_varying=True, w=512).to(dev)
    opt = cfm_amd.FusedAdam(net.parameters(), lr=1e-4); reg = cfm_amd.RegressionStep(net, opt)
    loss = reg.backward_only(t, a, b)
    torch.cuda.synchronize()
    res[mode] = (float(loss), [p.grad.double().cpu().clone() for p in net.parameters()])
for mode in (1, 2):
    dl = abs(res[mode][0] - res[0][0]) / abs(res[0][0])
    dg = max(float((g1 - g0).abs().max() / (g0.abs().max() + 1e-30)) for g1, g0 in zip(res[mode][1], res[0][1]))
    print(f"mode {mode}: loss rel diff {dl:.2e}, worst grad rel diff {dg:.2e}")
    # (a pre-activation that changes sign between the two k orders moves selu' by 67 %: the comparison that counts is the fp64 one of tests/test_gpu_train.py)


def med(fn, reps=30, rounds=5):
    fn(); fn(); torch.cuda.synchronize(); out = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize(); out.append(e0.elapsed_time(e1) / reps * 1e3)
    return float(np.median(out))


net = cfm_amd.MLP(dim=d, time_varying=True, w=512).to(dev)
opt = cfm_amd.FusedAdam(net.parameters(), lr=1e-4); reg = cfm_amd.RegressionStep(net, opt)
for rep in range(2):
    for mode in (0, 1, 2):
        lib.cfm_mlp_set_glds(mode)
        with torch.no_grad():
            us_fwd = med(lambda: net.forward_hip(a, t))
        us_fb = med(lambda: reg.backward_only(t, a, b))
        us_step = med(lambda: reg(t, a, b))
        print(f"glds mode {mode}: fwd {us_fwd:6.1f} us ({10.88e3/us_fwd:5.1f} TF)  fwd+mse+bwd {us_fb:6.1f} us ({32.6e3/us_fb:5.1f} TF)  step {us_step:6.1f} us", flush=True)
