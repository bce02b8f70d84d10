import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, cfm_amd
from cfm_amd import _lib
from cfm_amd.optim import FusedAdam
lib=_lib.load(); dev=_lib.require_gpu()
for wd in (0.0, 0.01):
  for variant in range(16):
    if wd == 0.0 and variant >= 8: continue
    lib.cfm_adam_set_variant(variant)
    torch.manual_seed(3)
    a = torch.nn.Parameter(torch.randn(40000, device=dev)); b = torch.nn.Parameter(a.detach().clone())
    oa = FusedAdam([a], lr=2e-4, weight_decay=wd); ob = torch.optim.Adam([b], lr=2e-4, weight_decay=wd)
    res=[]
    g = torch.Generator().manual_seed(5)
    for step in range(1,5):
        gr = torch.randn(40000, generator=g) * 10.0 ** float(torch.randint(-4, 2, (1,), generator=g))
        a.grad = gr.to(dev).clone(); b.grad = gr.to(dev).clone()
        oa.step(); ob.step()
        res.append((bool(torch.equal(oa.state[a]["exp_avg"], ob.state[b]["exp_avg"])), bool(torch.equal(oa.state[a]["exp_avg_sq"], ob.state[b]["exp_avg_sq"])), bool(torch.equal(a, b))))
    print("wd", wd, "variant", variant, res, flush=True)
