"""Time + check the fp32-MFMA dense products of the path (gemm_core.h) at the C3 shapes:
  forward layers (mlp_layer), backward (gemm_f32_mfma dgrad / wgrad), cost_gemm, and the whole model step.
    python tools/gemm_bench.py
Checks every result against float64 torch on the host (max relative error vs sum |a||b|).  Measurement
infrastructure; not part of the product path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import cfm_amd
from cfm_amd import _lib
import cfm_amd.optimal_transport as ot
import cfm_oracle as oracle

_lib.load(); dev = _lib.require_gpu()
torch.manual_seed(0)
B = 4096


def timeit(fn, reps=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3     # us


import ctypes
lib = _lib.load()
from cfm_amd._lib import ptr, stream_ptr, check


def c_args(ws_list):
    n = len(ws_list)
    return (ctypes.c_void_p * n)(*[w.data_ptr() for w in ws_list])


# ---- single products through the C ABI in a tight loop (pre-built arguments: the host is not the bottleneck) ----
def bench_forward(K, N, has_t):
    lin = torch.nn.Linear(K + int(has_t), N).to(dev)
    x = torch.randn(B, K, device=dev); t = torch.rand(B, device=dev)
    W = lin.weight.detach().contiguous(); bb = lin.bias.detach().contiguous()
    Wp, bp = c_args([W]), c_args([bb]); dims = (ctypes.c_int * 2)(K + int(has_t), N)
    out = torch.empty(B, N, device=dev)
    ws = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
    sp = stream_ptr()
    tp = ptr(t) if has_t else None

    def run():
        lib.cfm_mlp_forward_f32(ptr(x), tp, 1 if has_t else 0, Wp, bp, dims, 1, B, ptr(out), ptr(ws), sp)
    run(); torch.cuda.synchronize()
    xin = torch.cat([x, t[:, None]], -1) if has_t else x
    ref = xin.double().cpu() @ W.double().cpu().T + bb.double().cpu()
    scale = (xin.abs().double().cpu() @ W.abs().double().cpu().T).max()
    err = float((out.double().cpu() - ref).abs().max() / scale)
    us = timeit(run, reps=50)
    fl = 2.0 * B * K * N
    print(f"forward  [{B}x{K}] . [{N}x{K}]^T       {us:8.1f} us  {fl/us/1e6:6.1f} TFLOP/s   err/scale {err:.2e}")


for (K, N, has_t) in ((784, 512, True), (512, 512, False), (512, 784, False), (64, 64, False)):
    bench_forward(K, N, has_t)


def bench_backward(dims_list):
    """cfm_mlp_backward_f32 on a net with the given layer widths (dz from random data)."""
    n = len(dims_list) - 1
    Ws = [torch.randn(dims_list[l + 1], dims_list[l], device=dev) * 0.05 for l in range(n)]
    acts = [torch.randn(B, dims_list[l], device=dev) for l in range(n)]
    pre = [torch.randn(B, dims_list[l], device=dev) for l in range(n)]
    dout = torch.randn(B, dims_list[n], device=dev)
    dW = [torch.empty_like(w) for w in Ws]; db = [torch.empty(w.shape[0], device=dev) for w in Ws]
    ws = _lib.workspace(_lib.OP_MLP_TRAIN, B, max(dims_list), max(dims_list[l] * dims_list[l + 1] for l in range(n)), dev)
    ap = c_args(acts); zp = (ctypes.c_void_p * n)(*([0] + [z.data_ptr() for z in pre[1:]]))
    Wp, dWp, dbp = c_args(Ws), c_args(dW), c_args(db)
    cd = (ctypes.c_int * (n + 1))(*dims_list)
    sp = stream_ptr()

    def run():
        lib.cfm_mlp_backward_f32(ap, zp, Wp, cd, n, B, ptr(dout), dWp, dbp, None, ptr(ws), sp)
    us = timeit(run, reps=30)
    fl = sum(2.0 * B * dims_list[l] * dims_list[l + 1] * (2 if l > 0 else 1) for l in range(n))
    # check dW of the last layer against float64
    run(); torch.cuda.synchronize()
    ref = dout.double().cpu().T @ acts[n - 1].double().cpu()
    err = float((dW[n - 1].double().cpu() - ref).abs().max() / (dout.abs().double().cpu().T @ acts[n - 1].abs().double().cpu()).max())
    print(f"backward {dims_list}  {us:8.1f} us  {fl/us/1e6:6.1f} TFLOP/s   dW err/scale {err:.2e}")


bench_backward([785, 512, 512, 512, 784])
bench_backward([512, 512])

# ---- whole C3 model: inference forward, training forward + backward, step ----
mt = cfm_amd.MLP(dim=784, time_varying=True, w=512).to(dev)
x0, x1 = oracle.config_inputs("C3")
a, b = x0.to(dev), x1.to(dev)
t = torch.rand(B, device=dev)
xt = torch.cat([a, t[:, None]], -1)
with torch.no_grad():
    us = timeit(lambda: mt.forward_hip(a, t))
    y = mt.forward_hip(a, t); yr = mt.net(xt)
print(f"C3 inference forward (4 layers)      {us:8.1f} us  {10.88e9/us/1e6:6.1f} TFLOP/s   max|hip - torch| {float((y-yr).abs().max()):.2e}")
opt = cfm_amd.FusedAdam(mt.parameters(), lr=1e-4)


def fwd_bwd():
    opt.zero_grad(set_to_none=True)
    loss = ((mt(xt) - b) ** 2).mean()
    loss.backward()


def step():
    fwd_bwd(); opt.step()


us_fb = timeit(fwd_bwd); us_st = timeit(step)
print(f"C3 train fwd + bwd                    {us_fb:8.1f} us  {32.6e9/us_fb/1e6:6.1f} TFLOP/s")
print(f"C3 train fwd + bwd + Adam             {us_st:8.1f} us")
# gradient check against float64 autograd on the host
opt.zero_grad(set_to_none=True)
loss = ((mt(xt) - b) ** 2).mean(); loss.backward()
g_hip = [p.grad.detach().double().cpu() for p in mt.parameters()]
m64 = cfm_amd.MLP(dim=784, time_varying=True, w=512).double()
m64.load_state_dict({k: v.double().cpu() for k, v in mt.state_dict().items()})
l64 = ((m64.net(xt.double().cpu()) - b.double().cpu()) ** 2).mean(); l64.backward()
worst = max(float((g - p.grad).abs().max() / p.grad.abs().max()) for g, p in zip(g_hip, m64.parameters()))
print(f"gradients vs float64 autograd: worst max-norm relative error {worst:.2e}")

# ---- cost matrix ----
us = timeit(lambda: ot.cost_matrix(a, b, matrix_cores=True))
M = ot.cost_matrix(a, b, matrix_cores=True)
Mr = torch.cdist(x0.double(), x1.double()) ** 2
print(f"cost (center + norms + cost_gemm)     {us:8.1f} us  {2.0*B*B*784/us/1e6:6.1f} TFLOP/s   max rel err {float(((M.double().cpu()-Mr)/Mr).abs().max()):.2e}")
