"""GPU (-m gpu): the training step of the MLP vector field on the HIP kernels — forward / backward
through the autograd.Function against the float64 oracle (<= 1e-5 relative) and against PyTorch's own
autograd on the same weights; the fused Adam against torch.optim.Adam (bit-equal state after one and
after several steps on identical gradients) and against the float64 update."""
import numpy as np
import pytest
import torch

import cfm_oracle as oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from cfm_amd import _lib
    _lib.load()
    return _lib.require_gpu()


def _rel(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - b).max() / max(np.abs(b).max(), 1e-30))


def _abi_step(m, x, ut, dev):
    """cfm_mlp_forward_train_f32 + cfm_mlp_backward_f32 through ctypes (what the autograd.Function calls),
    returning everything including the float32 pre-activations."""
    import ctypes
    from cfm_amd import _lib
    lib = _lib.load()
    lins = m._linears()
    n = len(lins)
    Ws = [l.weight.detach().contiguous() for l in lins]; bs = [l.bias.detach().contiguous() for l in lins]
    dims = [Ws[0].shape[1]] + [w.shape[0] for w in Ws]
    B = x.shape[0]
    xd = x.to(dev).contiguous()
    hidden = [torch.empty((B, dims[l + 1]), device=dev) for l in range(n - 1)]
    preact = [torch.empty((B, dims[l + 1]), device=dev) for l in range(n - 1)]
    out = torch.empty((B, dims[n]), device=dev)
    arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() if t is not None else 0 for t in ts])
    cd = (ctypes.c_int * (n + 1))(*dims)
    _lib.check(lib.cfm_mlp_forward_train_f32(_lib.ptr(xd), arr(Ws), arr(bs), cd, n, B, arr(hidden), arr(preact),
                                             _lib.ptr(out), _lib.stream_ptr()), "fwd")
    dout = (2.0 / (B * dims[n])) * (out - ut.to(dev))
    dW = [torch.empty_like(w) for w in Ws]; db = [torch.empty(w.shape[0], device=dev) for w in Ws]
    dx = torch.empty_like(xd)
    ws = torch.empty(lib.cfm_workspace_bytes(_lib.OP_MLP_TRAIN, B, max(dims), max(dims[l] * dims[l + 1] for l in range(n))),
                     dtype=torch.uint8, device=dev)
    _lib.check(lib.cfm_mlp_backward_f32(arr([xd] + hidden), arr([None] + preact), arr(Ws), cd, n, B, _lib.ptr(dout),
                                        arr(dW), arr(db), _lib.ptr(dx), _lib.ptr(ws), _lib.stream_ptr()), "bwd")
    return out, dout, [z.cpu().numpy() for z in preact], dW, db, dx


@pytest.mark.parametrize("B,d,w", [(96, 2, 64), (300, 50, 64), (512, 784, 512), (4096, 784, 512), (130, 7, 33)])
def test_mlp_forward_backward_vs_f64_oracle(dev, B, d, w):
    """The backward kernels against the float64 oracle, SELU' branch pinned by the float32 forward's own
    pre-activations (see cfm_oracle.mlp_backward_f64); then the autograd.Function returns exactly these
    gradients, and agrees with PyTorch's autograd over the module graph."""
    import cfm_amd
    torch.manual_seed(B + d)
    m = cfm_amd.MLP(dim=d, time_varying=True, w=w).to(dev)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, d + 1, generator=g)
    ut = torch.randn(B, d, generator=g)
    out, dout, preact, dW, db, dx = _abi_step(m, x, ut, dev)
    Ws = [l.weight.detach().cpu().numpy() for l in m._linears()]
    bs = [l.bias.detach().cpu().numpy() for l in m._linears()]
    out_o, dW_o, db_o, dx_o = oracle.mlp_backward_f64(Ws, bs, x.numpy(), dout.cpu().numpy(), preact=preact)
    assert _rel(out.cpu().numpy(), out_o) <= 1e-5
    for l in range(len(Ws)):
        assert _rel(dW[l].cpu().numpy(), dW_o[l]) <= 1e-5, ("dW", l, _rel(dW[l].cpu().numpy(), dW_o[l]))
        assert _rel(db[l].cpu().numpy(), db_o[l]) <= 1e-5, ("db", l, _rel(db[l].cpu().numpy(), db_o[l]))
    assert _rel(dx.cpu().numpy(), dx_o) <= 1e-5, _rel(dx.cpu().numpy(), dx_o)
    # the autograd.Function: same kernels, same bits
    xin = x.to(dev).requires_grad_(True)
    vt = m(xin)
    assert vt.grad_fn is not None and "MLPTrain" in type(vt.grad_fn).__name__      # the HIP path, not the module graph
    loss = torch.mean((vt - ut.to(dev)) ** 2)
    loss.backward()
    assert torch.equal(vt.detach(), out)
    for l, lin in enumerate(m._linears()):
        torch.testing.assert_close(lin.weight.grad, dW[l], rtol=1e-6, atol=1e-9)     # (dout formed by eager ops here)
        torch.testing.assert_close(lin.bias.grad, db[l], rtol=1e-5, atol=1e-9)
    # PyTorch's own autograd (module graph, hipBLASLt): equal up to float32 round-off — and up to SELU' kink
    # flips (one flipped unit-sample moves a whole gradient by O(1 / B)), hence the bound on the loss and on the
    # direction of the gradients
    m2 = cfm_amd.MLP(dim=d, time_varying=True, w=w).to(dev)
    m2.load_state_dict(m.state_dict())
    m2.hip_training = False
    x2 = x.to(dev).requires_grad_(True)
    loss2 = torch.mean((m2(x2) - ut.to(dev)) ** 2)
    loss2.backward()
    assert float(loss.detach()) == pytest.approx(float(loss2.detach()), rel=1e-5)
    for a, b in zip(m._linears(), m2._linears()):
        ga, gb = a.weight.grad.cpu().double().numpy().ravel(), b.weight.grad.cpu().double().numpy().ravel()
        assert ga @ gb / np.sqrt((ga @ ga) * (gb @ gb)) >= 1.0 - 1e-4


def test_backward_is_deterministic(dev):
    import cfm_amd
    torch.manual_seed(0)
    m = cfm_amd.MLP(dim=784, time_varying=True, w=512).to(dev)
    x = torch.randn(4096, 785, device=dev)
    ut = torch.randn(4096, 784, device=dev)
    grads = []
    for _ in range(2):
        m.zero_grad(set_to_none=True)
        torch.mean((m(x) - ut) ** 2).backward()
        grads.append([p.grad.clone() for p in m.parameters()])
    for a, b in zip(*grads):
        assert torch.equal(a, b)


@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_fused_adam_bit_equal_to_torch_adam(dev, wd):
    import cfm_amd
    from cfm_amd.optim import FusedAdam
    torch.manual_seed(3)
    a = cfm_amd.MLP(dim=50, time_varying=True, w=64).to(dev)
    b = cfm_amd.MLP(dim=50, time_varying=True, w=64).to(dev)
    b.load_state_dict(a.state_dict())
    oa = FusedAdam(a.parameters(), lr=2e-4, weight_decay=wd)
    ob = torch.optim.Adam(b.parameters(), lr=2e-4, weight_decay=wd)
    g = torch.Generator(device="cpu").manual_seed(5)
    p0 = [p.detach().cpu().double().numpy() for p in a.parameters()]
    first_grads = None
    for step in range(1, 6):
        grads = [torch.randn(p.shape, generator=g) * (10.0 ** float(torch.randint(-4, 2, (1,), generator=g))) for p in a.parameters()]
        if first_grads is None:
            first_grads = [x.double().numpy() for x in grads]
        for pa, pb, gr in zip(a.parameters(), b.parameters(), grads):
            pa.grad = gr.to(dev).clone(); pb.grad = gr.to(dev).clone()
        oa.step(); ob.step()
        for (pa, pb) in zip(a.parameters(), b.parameters()):
            sa, sb = oa.state[pa], ob.state[pb]
            assert torch.equal(sa["exp_avg"], sb["exp_avg"]), ("exp_avg", step)
            assert torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"]), ("exp_avg_sq", step)
            assert torch.equal(pa, pb), ("param", step, float((pa - pb).abs().max()))
            assert int(sa["step"]) == int(sb["step"]) == step
        if step == 1:
            for pa, pz, gz in zip(a.parameters(), p0, first_grads):
                ref, _, _ = oracle.adam_step_f64(pz, gz, np.zeros_like(pz), np.zeros_like(pz), 1, lr=2e-4, weight_decay=wd)
                assert np.abs(pa.detach().cpu().double().numpy() - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max())


def test_training_loop_matches_pytorch_loop(dev):
    """Five OT-CFM steps with (HIP forward/backward + FusedAdam) and with (module graph + torch.optim.Adam)
    from the same initial weights and the same batches: the losses agree to fp32 round-off."""
    import cfm_amd
    from cfm_amd.optim import FusedAdam
    torch.manual_seed(0)
    a = cfm_amd.MLP(dim=2, time_varying=True, w=64).to(dev)
    b = cfm_amd.MLP(dim=2, time_varying=True, w=64).to(dev)
    b.load_state_dict(a.state_dict()); b.hip_training = False
    oa, ob = FusedAdam(a.parameters()), torch.optim.Adam(b.parameters())
    fm = cfm_amd.ExactOptimalTransportConditionalFlowMatcher(sigma=0.1)
    la, lb = [], []
    for k in range(5):
        x0, x1 = oracle.config_inputs("C1", rank=k)
        np.random.seed(k); torch.manual_seed(k)
        t, xt, ut = fm.sample_location_and_conditional_flow(x0.to(dev), x1.to(dev))
        for m, o, log in ((a, oa, la), (b, ob, lb)):
            o.zero_grad(set_to_none=True)
            loss = torch.mean((m(torch.cat([xt, t[:, None]], dim=-1)) - ut) ** 2)
            loss.backward(); o.step(); log.append(float(loss))
    np.testing.assert_allclose(la, lb, rtol=2e-4)


def test_gradmodel_double_backward_through_the_hip_mlp():
    """GradModel(MLP) differentiates the MLP's input gradient again (create_graph=True,
    torchcfm/models/models.py:24-32).  The HIP autograd.Function answers a grad-mode backward with differentiable
    torch ops on its saved inputs, so the second-order graph reaches the weights: one training step gives the same
    parameter gradients as the plain nn.Sequential graph."""
    import cfm_amd
    from cfm_amd.models import GradModel
    from cfm_amd import _lib
    dev = _lib.require_gpu()
    torch.manual_seed(0)
    act = cfm_amd.MLP(dim=2, out_dim=1, w=64, time_varying=True).to(dev)
    x = torch.randn(128, 3, device=dev)
    out = GradModel(act)(x.clone())
    assert out.shape == (128, 2) and out.requires_grad
    (out ** 2).mean().backward()
    # (the last bias does not enter d potential / dx: its gradient is None on both graphs)
    g_hip = [None if p.grad is None else p.grad.clone() for p in act.parameters()]
    assert all(torch.isfinite(g).all() for g in g_hip if g is not None)
    assert sum(g is not None and float(g.abs().max()) > 0 for g in g_hip) >= 6
    act.zero_grad()
    inp = x.clone().requires_grad_(True)
    (d,) = torch.autograd.grad(act.net(inp).sum(), inp, create_graph=True)
    (d[:, :-1] ** 2).mean().backward()
    for g, p in zip(g_hip, act.parameters()):
        assert (g is None) == (p.grad is None)
        if g is not None:
            assert (g - p.grad).abs().max() <= 1e-5 * max(1e-6, float(p.grad.abs().max()))
    # first-order use is still the fused path and still correct after a double-backward use
    act.zero_grad()
    y = act(x); y.sum().backward()
    assert all(p.grad is not None for p in act.parameters())


def test_mlp_training_path_validates_the_feature_count():
    """a forgotten time column must raise like nn.Linear, not misread memory."""
    import cfm_amd
    from cfm_amd import _lib
    dev = _lib.require_gpu()
    net = cfm_amd.MLP(dim=2, w=64, time_varying=True).to(dev)
    with pytest.raises(RuntimeError, match="cannot be multiplied"):
        net(torch.randn(16, 2, device=dev))


def test_regression_step_matches_the_autograd_path_and_float64(dev):
    """cfm_mlp_regression_step_f32 (time column fused, MSE + seed on the device, one reduction) against the
    reference's four lines on the autograd path and against float64 autograd on the host: loss and every gradient
    <= 1e-5 relative (the two HIP paths round the time column differently: epilogue fmaf vs a 785th k step)."""
    import cfm_amd
    torch.manual_seed(3)
    B, d, w = 512, 20, 64
    net = cfm_amd.MLP(dim=d, time_varying=True, w=w).to(dev)
    t = torch.rand(B, device=dev); xt = torch.randn(B, d, device=dev); ut = torch.randn(B, d, device=dev)
    opt = cfm_amd.FusedAdam(net.parameters(), lr=1e-3)
    reg = cfm_amd.RegressionStep(net, opt)
    loss = float(reg.backward_only(t, xt, ut))
    g_fused = [p.grad.detach().double().cpu().clone() for p in net.parameters()]
    # float64 on the host
    n64 = cfm_amd.MLP(dim=d, time_varying=True, w=w).double()
    n64.load_state_dict({k: v.double().cpu() for k, v in net.state_dict().items()})
    l64 = ((n64.net(torch.cat([xt, t[:, None]], -1).double().cpu()) - ut.double().cpu()) ** 2).mean()
    l64.backward()
    assert abs(loss - float(l64)) <= 1e-5 * float(l64)
    for g, p in zip(g_fused, n64.parameters()):
        assert (g - p.grad).abs().max() <= 1e-5 * p.grad.abs().max(), (g - p.grad).abs().max() / p.grad.abs().max()
    # the eager four lines on the autograd.Function path
    for p in net.parameters():
        p.grad = None
    le = torch.mean((net(torch.cat([xt, t[:, None]], -1)) - ut) ** 2); le.backward()
    assert abs(float(le) - loss) <= 1e-5 * abs(loss)
    for g, p in zip(g_fused, net.parameters()):
        assert (g - p.grad.double().cpu()).abs().max() <= 1e-5 * g.abs().max()
    # determinism: the same call twice gives the same bits
    reg2 = cfm_amd.RegressionStep(net, opt)
    a = float(reg2.backward_only(t, xt, ut)); ga = [p.grad.clone() for p in net.parameters()]
    b = float(reg2.backward_only(t, xt, ut))
    assert a == b and all(torch.equal(x, p.grad) for x, p in zip(ga, net.parameters()))


def test_regression_step_c3_shape_and_training_loop(dev):
    """C3 field (785-512-512-512-784, B = 4096): gradients vs float64; then 5 steps of RegressionStep + FusedAdam
    track the eager loop (autograd path + torch.optim.Adam) started from the same weights."""
    import copy
    import cfm_amd
    torch.manual_seed(5)
    B, d = 4096, 784
    net = cfm_amd.MLP(dim=d, time_varying=True, w=512).to(dev)
    ref = copy.deepcopy(net)
    t = torch.rand(B, device=dev); xt = torch.randn(B, d, device=dev); ut = torch.randn(B, d, device=dev) * 0.5
    opt = cfm_amd.FusedAdam(net.parameters(), lr=1e-3)
    reg = cfm_amd.RegressionStep(net, opt)
    loss = float(reg.backward_only(t, xt, ut))
    n64 = cfm_amd.MLP(dim=d, time_varying=True, w=512).double()
    n64.load_state_dict({k: v.double().cpu() for k, v in net.state_dict().items()})
    l64 = ((n64.net(torch.cat([xt, t[:, None]], -1).double().cpu()) - ut.double().cpu()) ** 2).mean(); l64.backward()
    assert abs(loss - float(l64)) <= 1e-5 * float(l64)
    for p, q in zip(net.parameters(), n64.parameters()):
        assert (p.grad.double().cpu() - q.grad).abs().max() <= 1e-5 * q.grad.abs().max()
    o2 = torch.optim.Adam(ref.parameters(), lr=1e-3)
    la, lb = [], []
    for _ in range(5):
        la.append(float(reg(t, xt, ut)))
        o2.zero_grad(set_to_none=True)
        l2 = torch.mean((ref(torch.cat([xt, t[:, None]], -1)) - ut) ** 2); l2.backward(); o2.step(); lb.append(float(l2))
    assert all(abs(a - b) <= 1e-4 * abs(b) for a, b in zip(la, lb)), (la, lb)
    assert la[-1] < la[0]


def test_regression_step_without_time_column(dev):
    import cfm_amd
    torch.manual_seed(7)
    net = cfm_amd.MLP(dim=6, time_varying=False, w=32).to(dev)
    x = torch.randn(200, 6, device=dev); u = torch.randn(200, 6, device=dev)
    reg = cfm_amd.RegressionStep(net, cfm_amd.FusedAdam(net.parameters()))
    loss = float(reg.backward_only(None, x, u))
    g = [p.grad.clone() for p in net.parameters()]
    for p in net.parameters():
        p.grad = None
    le = torch.mean((net(x) - u) ** 2); le.backward()
    assert abs(float(le) - loss) <= 1e-6 * abs(loss)
    for a, p in zip(g, net.parameters()):
        assert (a - p.grad).abs().max() <= 1e-5 * p.grad.abs().max()


def test_bucketed_backward_and_scaled_adam_equal_the_single_gpu_forms(dev):
    """The data-parallel form of the step: per-layer reductions + layer_done events give bit-equal
    gradients and loss; Adam with grad_scale = 1/4 equals `grad.mul_(0.25)` followed by the unscaled launch, bit for
    bit, and leaves the scaled gradient in .grad (the mean, as DDP leaves it: train_cifar10_ddp.py:92)."""
    import copy
    import cfm_amd
    torch.manual_seed(9)
    B, d, w = 1024, 50, 64
    net = cfm_amd.MLP(dim=d, time_varying=True, w=w).to(dev)
    ref = copy.deepcopy(net)
    t = torch.rand(B, device=dev); xt = torch.randn(B, d, device=dev); ut = torch.randn(B, d, device=dev)
    ra = cfm_amd.RegressionStep(net, cfm_amd.FusedAdam(net.parameters(), lr=1e-3))
    rb = cfm_amd.RegressionStep(ref, cfm_amd.FusedAdam(ref.parameters(), lr=1e-3))
    la = float(ra.backward_only(t, xt, ut))
    evp = rb._dp_setup()
    lb = float(rb.backward_only(t, xt, ut, layer_events=evp))
    for e in rb._events:
        e.synchronize()
    assert la == lb
    assert torch.equal(ra.flat_grad, rb.flat_grad)
    # the events order a second stream behind each layer's gradient
    with torch.cuda.stream(rb._comm):
        for l in range(rb.n - 1, -1, -1):
            rb._comm.wait_event(rb._events[l])
        s = rb.flat_grad.clone()
    torch.cuda.synchronize()
    assert torch.equal(s, ra.flat_grad)
    # Adam: scale folded into the launch vs an eager mul_ in front of it
    ra.flat_grad.mul_(0.25); ra.opt.step()
    rb.opt.step(grad_scale=0.25)
    torch.cuda.synchronize()
    assert torch.equal(ra.flat_grad, rb.flat_grad)
    for p, q in zip(net.parameters(), ref.parameters()):
        assert torch.equal(p, q)
    with pytest.raises(Exception):
        rb.opt.step(grad_scale=0.0)
