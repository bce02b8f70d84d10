"""GPU: the direct-to-LDS tile engine (csrc/gemm_glds.h, CFM_COST_GLDS=1) builds the same cost matrix, bit for bit, as the
register-staged engine — interior tiles, ragged edges in both directions, a K tail, duplicated points (the cancellation
path).  The switch is read once per process, so each engine runs in its own child process."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch, cfm_amd, cfm_amd.optimal_transport as ot
from cfm_amd import _lib
dev = _lib.require_gpu(); torch.manual_seed(0)
out = {}
for (B0, B1, d) in ((1024, 1024, 784), (1000, 777, 100), (512, 300, 64), (256, 256, 788), (640, 512, 192)):
    a = torch.randn(B0, d, device=dev); b = torch.randn(B1, d, device=dev) * 0.5 + 0.2
    b[:5] = a[:5]
    out[(B0, B1, d)] = ot.cost_matrix(a, b).cpu()
torch.save(out, sys.argv[1])
'''


def test_glds_cost_matrix_is_bit_equal_to_the_register_staged_engine(tmp_path):
    res = {}
    for flag in ("0", "1"):
        f = str(tmp_path / f"cost_{flag}.pt")
        env = dict(os.environ, CFM_COST_GLDS=flag)
        p = subprocess.run([sys.executable, "-c", CHILD % (ROOT, os.path.join(ROOT, "oracle")), f], env=env,
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        res[flag] = torch.load(f)
    for key, M in res["0"].items():
        assert torch.equal(M, res["1"][key]), key
        assert float(M[:5, :5].diagonal().abs().max()) == 0.0          # duplicates: the recomputed entries are exact zeros
