"""Time cfm_assign_exact_batch_f32 at C3 size for several batch sizes: ms per batch and per problem.
    python tools/asg_batch_bench.py [nb ...]
Measurement infrastructure."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import cfm_amd.optimal_transport as ot
import cfm_oracle as oracle

from cfm_amd import _lib
dev = torch.device("cuda", 0)
if os.environ.get("CFM_ASG_BLOCKS"):
    _lib.load().cfm_assign_set_wide_blocks(int(os.environ["CFM_ASG_BLOCKS"]))
sizes = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16]
Ms = []
for k in range(max(sizes)):
    x0, x1 = oracle.config_inputs("C3", rank=k)
    Ms.append(ot.cost_matrix(x0.to(dev), x1.to(dev)))
with torch.cuda.stream(torch.cuda.Stream()):
    for nb in sizes:
        ot.assign_exact_batch(Ms[:nb]); torch.cuda.synchronize()
        ts = []
        for _ in range(8):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ot.assign_exact_batch(Ms[:nb]); torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        print(f"nb={nb:2d}: median {1e3*ts[len(ts)//2]:7.3f} ms per batch = {1e3*ts[len(ts)//2]/nb:6.3f} ms per problem (min {1e3*ts[0]:.3f})")
