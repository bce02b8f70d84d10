import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'oracle'))
import torch, numpy as np
from cfm_amd import _lib
import cfm_amd.optimal_transport as ot
import bench
lib=_lib.load(); dev=_lib.require_gpu()
x0,x1=bench.synth_batches(4096,784,1,1000,dev)[0]
for _ in range(3): M=ot.cost_matrix(x0,x1)
torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): M=ot.cost_matrix(x0,x1)
e1.record(); torch.cuda.synchronize()
ms=e0.elapsed_time(e1)/20
print(f"cost_tiled C3: {ms*1e3:.1f} us  = {3*4096*4096*784/ms/1e9:.1f} TFLOP/s (3 flop / element-k), checksum {float(M.double().sum()):.6f}")
