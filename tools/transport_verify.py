"""Debugging aid for cfm_transport_exact_f32: run with CFM_TP_VERIFY=1 — the solver then re-derives every label densely
after each phase's label-correcting sweeps and reports mismatches (info[7] bits 8+) — on a few cold-start cases, three times each
(the results must not change from run to run).  Round 6 found a non-uniform branch around a barrier with it.
    CFM_TP_VERIFY=1 python tools/transport_verify.py
Measurement infrastructure."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch, numpy as np
import cfm_amd.optimal_transport as ot
from cfm_amd import _lib
from cfm_amd._lib import ptr, stream_ptr
lib = _lib.load(); dev = torch.device("cuda", 0)
for (B0, B1, d, seed) in [(128, 127, 2, 100 * 128 + 127), (127, 128, 2, 100 * 127 + 128), (255, 256, 2, 255 * 7 + 256), (63, 64, 2, 9), (100, 60, 16, 10060), (300, 257, 4, 30257)]:
    g = torch.Generator().manual_seed(seed)
    x0 = torch.randn(B0, d, generator=g); x1 = torch.randn(B1, d, generator=g) * 0.7 + 0.5
    M = ot.cost_matrix(x0.to(dev), x1.to(dev))
    for rep in range(3):
        plan = torch.empty((B0, B1), dtype=torch.float64, device=dev); tot = torch.empty(1, dtype=torch.float64, device=dev); info = torch.empty(8, dtype=torch.int32, device=dev)
        ws = _lib.workspace(_lib.OP_TRANSPORT, B0, B1, 0, dev)
        rc = lib.cfm_transport_exact_f32(ptr(M), B0, B1, None, ptr(plan), ptr(tot), ptr(info), ptr(ws), stream_ptr())
        st = info.cpu().tolist()
        print(B0, B1, "rc", rc, "status", st[0], "phases", st[1], "sweeps", st[2], "viol", st[4], "flags", st[7] & 3, "label mismatches", (st[7] >> 8) & 0xfff, "first bad phase", st[7] >> 20, flush=True)
