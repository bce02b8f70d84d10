"""GPU: the N > 1 code on RCCL, on the one GPU there is.

Nothing of the multi-GPU composition had ever run on the real backend: all_gather_samples returns early at world 1 and
RegressionStep's bucketed gradient all-reduce (communication stream, per-layer `layer_done` events recorded by the C call
between its launches, `grad_scale` inside the Adam launch) only ran over gloo.  A world-size-1 `nccl` (= RCCL) process
group puts exactly those calls through the real backend: it measures nothing, it proves the composition neither
deadlocks nor mis-orders there.  Runs in a child process under a timeout — a hang is a failed test, not a hung box.
Ref: examples/images/cifar10/train_cifar10_ddp.py:124,167-169, utils_cifar.py:35-39."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    import cfm_amd
    from cfm_amd import distributed as D
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=%r)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    assert dist.get_backend() == "nccl"
    # ---- the final-sample all-gather: early return without force, the collective and the P2P form with it
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4096, 784, generator=g).to(dev)
    assert D.all_gather_samples(x) is x
    a = D.all_gather_samples(x, force=True)
    b = D.all_gather_samples(x, direct=True, force=True)
    torch.cuda.synchronize()
    assert a is not x and torch.equal(a, x) and torch.equal(b, x)
    assert D.max_over_ranks(1.25) == 1.25
    D.barrier()
    # ---- the regression step: single-GPU form against the data-parallel form, bit for bit, three steps
    def run(dp):
        torch.manual_seed(0)
        model = cfm_amd.MLP(dim=784, time_varying=True, w=512).to(dev)
        opt = cfm_amd.FusedAdam(model.parameters(), lr=1e-3)
        reg = cfm_amd.RegressionStep(model, opt, data_parallel=dp)
        g = torch.Generator().manual_seed(1)
        losses = []
        for _ in range(3):
            t = torch.rand(4096, generator=g).to(dev); xt = torch.randn(4096, 784, generator=g).to(dev)
            ut = torch.randn(4096, 784, generator=g).to(dev)
            losses.append(reg(t, xt, ut).clone())
        torch.cuda.synchronize()
        return [p.detach().clone() for p in model.parameters()], [l.item() for l in losses], reg.flat_grad.clone()
    p0, l0, g0 = run(False)
    p1, l1, g1 = run(True)           # world 1: sum over one rank, grad_scale = 1.0 — the same numbers, through RCCL
    assert l0 == l1, (l0, l1)
    assert torch.equal(g0, g1)
    for a_, b_ in zip(p0, p1):
        assert torch.equal(a_, b_)
    # ---- any optimizer (the non-fused branch: scale the buckets, then its own step)
    torch.manual_seed(0)
    model = cfm_amd.MLP(dim=16, time_varying=True, w=64).to(dev)
    reg = cfm_amd.RegressionStep(model, torch.optim.SGD(model.parameters(), lr=0.1), data_parallel=True)
    reg(torch.rand(256).to(dev), torch.randn(256, 16).to(dev), torch.randn(256, 16).to(dev))
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print("RCCL_WORLD1_OK")
""")


def test_n_gt_1_composition_runs_on_rccl_with_one_rank():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    res = subprocess.run([sys.executable, "-c", CHILD % (ROOT, str(port))], capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0 and "RCCL_WORLD1_OK" in res.stdout, (res.stdout[-2000:], res.stderr[-4000:])
