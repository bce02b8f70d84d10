"""The MLP layers on the 64 x 64 direct-to-LDS engine (csrc/gemm_glds64.h, round 6) — the forward of
torchcfm/models/models.py:10-21 as `cfm_mlp_forward_f32` runs it — against an fp64 evaluation of the same network and
against the register-staged core (cfm_mlp_set_glds(0)), at the C3 layer shapes and at every edge the engine has: a K of
16 (the tail step alone), 48 (one step + tail), 32 / 64 (no tail), rows and columns that are not multiples of the tile,
weight rows that are only 4-byte aligned (the 785-wide first layer of a time-varying field), per-row and scalar time."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import cfm_amd  # noqa: F401
    from cfm_amd import _lib
    _lib.load()
    return _lib.require_gpu()


def _ref64(net, x, t):
    with torch.no_grad():
        h = x.double().cpu()
        if t is not None:
            tt = t.double().cpu() if torch.is_tensor(t) else torch.full((x.shape[0],), float(t), dtype=torch.float64)
            h = torch.cat([h, tt.reshape(-1, 1).expand(x.shape[0], 1)], 1)
        for m in net.net:
            if isinstance(m, torch.nn.Linear):
                h = h @ m.weight.double().cpu().T + m.bias.double().cpu()
            else:
                h = torch.nn.functional.selu(h)
    return h


@pytest.mark.parametrize("B,d,w,tv", [(4096, 784, 512, True), (4096, 784, 512, False), (1000, 48, 64, True),
                                      (130, 16, 80, True), (257, 32, 512, False), (64, 784, 128, True),
                                      (100, 16, 16, False), (333, 64, 48, True), (65, 48, 144, False)])
def test_layers_on_the_dma_engine_match_fp64_and_the_register_staged_core(dev, B, d, w, tv):
    import cfm_amd
    from cfm_amd import _lib
    lib = _lib.load()
    torch.manual_seed(B + d + w)
    net = cfm_amd.MLP(dim=d, time_varying=tv, w=w).to(dev)
    x = torch.randn(B, d, device=dev)
    times = (None,) if not tv else (torch.rand(B, device=dev), 0.37)
    mode0 = lib.cfm_mlp_get_glds()
    try:
        for t in times:
            ref = _ref64(net, x, t)
            sc = float(ref.abs().max())
            outs = {}
            for mode in (0, 1, 2):
                lib.cfm_mlp_set_glds(mode)
                with torch.no_grad():
                    outs[mode] = net.forward_hip(x, t).cpu()
                    again = net.forward_hip(x, t).cpu()
                assert torch.equal(outs[mode], again)                                       # a fixed summation order
                err = float((outs[mode].double() - ref).abs().max()) / sc
                assert err <= 1e-5, (mode, err)                                             # north star: 1e-5 rel fp32
            assert float((outs[2] - outs[0]).abs().max()) <= 4e-6 * sc
    finally:
        lib.cfm_mlp_set_glds(mode0)


def test_the_engine_is_the_default_and_the_knob_round_trips(dev):
    from cfm_amd import _lib
    lib = _lib.load()
    m = lib.cfm_mlp_get_glds()
    assert m == 2
    lib.cfm_mlp_set_glds(0); assert lib.cfm_mlp_get_glds() == 0
    lib.cfm_mlp_set_glds(m); assert lib.cfm_mlp_get_glds() == m
