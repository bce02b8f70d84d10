"""GPU: cfm_assign_exact_batch_f32 — several assignment problems of one size in one chain of launches.  Every row of the
result must be what the single solve returns (same kernels, same per-problem state machine) and SciPy's optimum."""
import numpy as np
import ctypes

import pytest
import torch

import cfm_oracle as oracle

pytestmark = pytest.mark.gpu


def _matrices(nb, B, d, seed, dev):
    import cfm_amd.optimal_transport as ot
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(nb):
        x0 = torch.randn(B, d, generator=g)
        x1 = torch.randn(B, d, generator=g) * 0.6 + 0.3
        out.append(ot.cost_matrix(x0.to(dev), x1.to(dev)))
    return out


@pytest.mark.parametrize("nb,B,d", [(2, 300, 2), (3, 512, 16), (4, 1024, 64), (5, 777, 3), (8, 2048, 32)])
def test_batch_equals_single_solves_and_scipy(nb, B, d):
    import cfm_amd.optimal_transport as ot
    from cfm_amd import _lib
    dev = _lib.require_gpu()
    Ms = _matrices(nb, B, d, 10 * nb + B, dev)
    perms, infos = ot.assign_exact_batch(Ms, return_info=True)
    assert perms.shape == (nb, B) and perms.dtype == torch.int32
    for b in range(nb):
        single = ot.assign_exact(Ms[b])
        assert torch.equal(perms[b], single), b
        p = perms[b].cpu().numpy()
        ref = oracle.exact_perm(Ms[b].cpu().numpy())
        assert sorted(p.tolist()) == list(range(B))
        assert infos[b]["certified"]
        assert np.isclose(infos[b]["total_cost"], oracle.assignment_cost(Ms[b].cpu().numpy(), ref), rtol=1e-12, atol=0)
        if not np.array_equal(p, ref):          # ties (d = 2 duplicates do not occur here, but keep the cost statement)
            assert oracle.assignment_cost(Ms[b].cpu().numpy(), p) == pytest.approx(
                oracle.assignment_cost(Ms[b].cpu().numpy(), ref), rel=1e-12)


def test_batch_of_stacked_tensor_and_more_than_one_group():
    """A [nb,B,B] tensor is accepted; nb > 16 runs in groups of 16 on the same workspace."""
    import cfm_amd.optimal_transport as ot
    from cfm_amd import _lib
    dev = _lib.require_gpu()
    Ms = torch.stack(_matrices(19, 320, 5, 3, dev))
    perms = ot.assign_exact_batch(Ms)
    for b in range(19):
        assert np.array_equal(perms[b].cpu().numpy(), oracle.exact_perm(Ms[b].cpu().numpy())), b
    odd = torch.stack(_matrices(3, 301, 4, 9, dev))          # odd size: slices 1, 2 are not 16-byte aligned
    perms = ot.assign_exact_batch(odd)
    for b in range(3):
        assert np.array_equal(perms[b].cpu().numpy(), oracle.exact_perm(odd[b].cpu().numpy())), b


def test_batch_small_sizes_and_single_problem():
    """B <= 256 (the one-workgroup solver) and nb = 1 go one after the other through the single solve."""
    import cfm_amd.optimal_transport as ot
    from cfm_amd import _lib
    dev = _lib.require_gpu()
    for nb, B in ((3, 64), (2, 256), (1, 900)):
        Ms = _matrices(nb, B, 2, B, dev)
        perms = ot.assign_exact_batch(Ms)
        for b in range(nb):
            assert np.array_equal(perms[b].cpu().numpy(), oracle.exact_perm(Ms[b].cpu().numpy())), (nb, B, b)


def test_batch_with_tied_and_degenerate_members():
    """Integer-valued (massively tied) and constant matrices next to a generic one: every member certified, optimal cost;
    a member that needs the dense state machine does not disturb the others."""
    import cfm_amd.optimal_transport as ot
    from cfm_amd import _lib
    dev = _lib.require_gpu()
    B = 384
    r = np.random.RandomState(5)
    mats = [r.randint(0, 4, (B, B)).astype(np.float32), np.full((B, B), 2.5, np.float32),
            r.rand(B, B).astype(np.float32), (r.randint(0, 50, (B, 1)) + r.randint(0, 50, (1, B))).astype(np.float32)]
    Ms = [torch.from_numpy(m).to(dev) for m in mats]
    perms, infos = ot.assign_exact_batch(Ms, return_info=True)
    for b, m in enumerate(mats):
        p = perms[b].cpu().numpy()
        assert sorted(p.tolist()) == list(range(B))
        assert oracle.assignment_cost(m, p) == pytest.approx(oracle.assignment_cost(m, oracle.exact_perm(m)), rel=1e-12, abs=1e-9)


def test_batch_c3_full_size_four_problems():
    """Four C3-sized couplings (B = 4096, d = 784) at once: indices bit-equal to the single solves."""
    import cfm_amd.optimal_transport as ot
    from cfm_amd import _lib
    dev = _lib.require_gpu()
    Ms = []
    for k in range(4):
        x0, x1 = oracle.config_inputs("C3", rank=k)
        Ms.append(ot.cost_matrix(x0.to(dev), x1.to(dev)))
    perms = ot.assign_exact_batch(Ms)
    for b in range(4):
        assert torch.equal(perms[b], ot.assign_exact(Ms[b])), b
    assert not torch.equal(perms[0], perms[1])


def test_batch_argument_checks():
    import cfm_amd.optimal_transport as ot
    from cfm_amd import _lib
    dev = _lib.require_gpu()
    a = torch.rand(300, 300, device=dev); b = torch.rand(301, 301, device=dev)
    with pytest.raises(NotImplementedError):
        ot.assign_exact_batch([a, b])
    with pytest.raises(ValueError):
        ot.assign_exact_batch([])
    with pytest.raises(ValueError):
        ot.assign_exact_batch([a, a.double()])


@pytest.mark.parametrize("B,check_scipy", [(8448, True), (8197, False)])
def test_sizes_beyond_the_lds_price_snapshot(B, check_scipy):
    """n > 8192: the bid rounds have no price snapshot in LDS (keys read with the costs; 8448: the 16-byte form,
    8197: the scalar form).  Certified permutation; for 8448 the cost equals SciPy's optimum."""
    import cfm_amd.optimal_transport as ot
    from cfm_amd import _lib
    dev = _lib.require_gpu()
    g = torch.Generator().manual_seed(B)
    x0 = torch.randn(B, 3, generator=g); x1 = torch.randn(B, 3, generator=g) * 0.8 + 0.2
    M = ot.cost_matrix(x0.to(dev), x1.to(dev))
    perm, info = ot.assign_exact(M, return_info=True)
    p = perm.cpu().numpy()
    assert sorted(p.tolist()) == list(range(B)) and info["certified"]
    Mh = M.cpu().numpy()
    assert info["total_cost"] == pytest.approx(oracle.assignment_cost(Mh, p), rel=1e-12)
    if check_scipy:
        ref = oracle.exact_perm(Mh)
        assert oracle.assignment_cost(Mh, p) == pytest.approx(oracle.assignment_cost(Mh, ref), rel=1e-12)


@pytest.mark.parametrize("n,kind", [(512, "geo"), (640, "ties"), (777, "geo"), (1024, "geo"), (1500, "geo"), (2048, "uniform"), (4096, "ties"), (4096, "geo784"), (8192, "geo")])
def test_async_auction_gives_the_synchronous_rounds_permutation(n, kind):
    """The one-launch asynchronous phase A (asg_auction, default for 512 <= n <= 8192 since round 6) against the synchronous bid
    rounds (cfm_assign_set_async(0, ...)): nothing downstream trusts phase A, so both must end in the SAME certified
    optimum — identical permutations on generic costs, identical optimal cost on heavily tied ones — for single solves
    and for the batch entry (which runs the auction on its own, smaller grid)."""
    import cfm_amd.optimal_transport as ot
    from cfm_amd import _lib
    dev = _lib.require_gpu()
    lib = _lib.load()
    g = torch.Generator().manual_seed(n + len(kind))
    if kind == "uniform":
        Ms = [torch.rand(n, n, generator=g).to(dev) for _ in range(3)]
    elif kind == "ties":
        Ms = [torch.randint(0, 8, (n, n), generator=g).float().to(dev) for _ in range(3)]
    else:
        d = 784 if kind == "geo784" else 3
        Ms = []
        for _ in range(3):
            a = torch.randn(n, d, generator=g).to(dev); b = (torch.randn(n, d, generator=g) * 0.7 + 0.3).to(dev)
            Ms.append(ot.cost_matrix(a, b))
    saved = (ctypes.c_int * 3)(); lib.cfm_assign_get_async(saved)
    assert saved[0] == 2, "the shipped default is mode 2 (every bid in the one auction launch)"
    try:
        res = {}
        for on in (0, 1, 2):            # 2 = the shipped default: epsilon = 0 rounds inside asg_auction, whole solve = the unpolled head
            lib.cfm_assign_set_async(on, -1, -1)
            singles = [ot.assign_exact(M, return_info=True) for M in Ms]
            batch = ot.assign_exact_batch(Ms)
            for (p, info), pb in zip(singles, batch):
                assert info["certified"] and sorted(p.cpu().tolist()) == list(range(n))
                if kind != "ties":
                    assert torch.equal(p, pb)
            res[on] = singles
        for on in (1, 2):
            for (p0, i0), (p1, i1), M in zip(res[0], res[on], Ms):
                if kind == "ties":
                    assert i0["total_cost"] == i1["total_cost"]
                else:
                    assert torch.equal(p0, p1), on
        # the asynchronous path really ran: ~10-20 launches per solve instead of ~100 (n <= 4096: the list solver closes
        # the solve in one more launch; beyond, the dense forest's launch count depends on the free rows left)
        if n <= 4096:
            assert res[1][0][1]["stats"][6] < res[0][0][1]["stats"][6]
            assert res[2][0][1]["stats"][6] <= res[1][0][1]["stats"][6]
    finally:
        lib.cfm_assign_set_async(saved[0], saved[1], saved[2])      # what it found (the default), not a hard-coded mode
        now = (ctypes.c_int * 3)(); lib.cfm_assign_get_async(now)
        assert list(now) == list(saved)


def test_concurrent_lone_solves_on_many_streams_do_not_stall_each_other():
    """Six host threads, six streams, lone solves at the same time: every solve launches an auction grid as large as the
    chip, so their workgroups can only be resident in part — none may wait for workgroups that cannot start before another
    grid has finished for long (asg_auction: a slot never written stops counting after ASG_ASYNC_GRACE looks of the controller, ~40 ms).  Same
    permutations as one after the other, and no solve anywhere near the loop caps (0.3 s)."""
    import concurrent.futures as cf
    import time
    import cfm_amd.optimal_transport as ot
    from cfm_amd import _lib
    dev = _lib.require_gpu()
    n = 2048
    g = torch.Generator().manual_seed(77)
    Ms = []
    for _ in range(6):
        a = torch.randn(n, 16, generator=g).to(dev); b = (torch.randn(n, 16, generator=g) * 0.8 + 0.2).to(dev)
        Ms.append(ot.cost_matrix(a, b))
    want = [ot.assign_exact(M).cpu() for M in Ms]
    torch.cuda.synchronize()

    def work(q):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(torch.cuda.Stream(device=dev)):
            ot.assign_exact(Ms[q])                                   # one-time costs of this thread (workspace, launch programs)
            torch.cuda.current_stream().synchronize()
            t0 = time.perf_counter()
            outs = [ot.assign_exact(Ms[q]).cpu() for _ in range(5)]
            return outs, (time.perf_counter() - t0) / 5

    with cf.ThreadPoolExecutor(max_workers=6) as pool:
        res = list(pool.map(work, range(6)))
    for q, (outs, dt) in enumerate(res):
        assert all(torch.equal(o, want[q]) for o in outs), q
        # six chip-sized grids at once: a workgroup of one grid that cannot start has its rows adopted by the running ones
        # (round 6) — the mutual wait of round 5 cost the grace, ~30 ms per solve; a solve takes ~1.5 ms alone
        assert dt < 0.025, (q, dt)
    print("concurrent lone solves, s per solve and thread:", [round(dt, 4) for _, dt in res])
