"""A few launches of each dense product (for rocprofv3 --pmc): one 4096 x 512 x 512 forward layer, the C3 backward,
cost_gemm.  Measurement infrastructure; not part of the product path."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import cfm_amd
from cfm_amd import _lib
from cfm_amd._lib import ptr, stream_ptr
import cfm_amd.optimal_transport as ot
lib = _lib.load(); dev = _lib.require_gpu()
torch.manual_seed(0)
B, K, N = 4096, 512, 512
lin = torch.nn.Linear(K, N).to(dev)
x = torch.randn(B, K, device=dev)
W = lin.weight.detach().contiguous(); bb = lin.bias.detach().contiguous()
Wp = (ctypes.c_void_p * 1)(W.data_ptr()); bp = (ctypes.c_void_p * 1)(bb.data_ptr()); dims = (ctypes.c_int * 2)(K, N)
out = torch.empty(B, N, device=dev); ws = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
for _ in range(5):
    lib.cfm_mlp_forward_f32(ptr(x), None, 0, Wp, bp, dims, 1, B, ptr(out), ptr(ws), stream_ptr())
a = torch.randn(B, 784, device=dev); b = torch.randn(B, 784, device=dev)
for _ in range(3):
    ot.cost_matrix(a, b, matrix_cores=True)
torch.cuda.synchronize()
