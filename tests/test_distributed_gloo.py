"""CPU: the N>1 sharding path with world_size=2 on the gloo backend (one process per rank,
independent minibatches, a single all-gather of the final samples, max-over-ranks timing)."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from cfm_amd import distributed as D
    import cfm_oracle as oracle
    r, l, w = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    # each rank draws ITS OWN minibatch (seed base + rank) and couples it locally
    x0, x1 = oracle.config_inputs("C1", B=32, rank=rank)
    perm = oracle.exact_perm(oracle.ref_cost_f32(x0, x1))
    local = x1[perm]
    gathered = D.all_gather_samples(local)
    assert torch.equal(gathered, D.all_gather_samples(local, direct=True))      # the point-to-point form: same bytes
    assert gathered.shape == (world * 32, 2)
    assert torch.equal(gathered[rank * 32:(rank + 1) * 32], local)
    other = 1 - rank
    xo0, xo1 = oracle.config_inputs("C1", B=32, rank=other)
    assert torch.equal(gathered[other * 32:(other + 1) * 32],
                       xo1[oracle.exact_perm(oracle.ref_cost_f32(xo0, xo1))])
    assert D.max_over_ranks(float(rank + 1)) == float(world)
    D.barrier()
    out[rank] = float(gathered.sum())
    dist.destroy_process_group()


def test_world_size_2_gloo():
    mgr = mp.Manager()
    out = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert len(out) == 2 and out[0] == out[1]


def test_single_process_identity():
    sys.path.insert(0, ROOT)
    from cfm_amd import distributed as D
    x = torch.randn(4, 3)
    assert D.all_gather_samples(x) is x
    assert D.max_over_ranks(2.5) == 2.5
    assert D.shard_seed(1000, 3) == 1003


# ----------------------------------------------------------------------------------------------
# bench.py's own multi-rank loop (init_from_env -> per-rank minibatches -> per-rank coupling
# prefetcher -> DDP model step -> ONE all-gather of the final samples -> barrier / max-over-ranks
# timing) on CPU stand-ins: the oracle couples, a small torch MLP under DDP(gloo) steps.
def _bench_worker(rank, world, port, out):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CFM_DIST_BACKEND="gloo")
    import numpy as np
    import bench
    from cfm_amd import distributed as D
    from cfm_amd.prefetch import CouplingPrefetcher
    import cfm_oracle as oracle
    r, l, w = D.init_from_env(backend="gloo")
    B, d, steps, warm = 32, 2, 5, 2
    pool = [oracle.config_inputs("C1", B=B, rank=D.shard_seed(10 * k, rank)) for k in range(4)]   # this rank's own data
    torch.manual_seed(0)
    model = torch.nn.parallel.DistributedDataParallel(
        torch.nn.Sequential(torch.nn.Linear(d + 1, 16), torch.nn.SELU(), torch.nn.Linear(16, d)))
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    np.random.seed(D.shard_seed(1, rank)); torch.manual_seed(D.shard_seed(1, rank))
    coupled, drawn_log = [], []

    def draw():
        dr = (np.random.random_sample(B), torch.rand(B)); drawn_log.append(dr[0][0]); return dr

    def couple(x0, x1, drawn):
        u, t = drawn
        perm = oracle.exact_perm(oracle.ref_cost_f32(x0, x1))
        i, j = oracle.sample_perm_given_u(perm, u)
        xt, ut = oracle.xt_ut("icfm", x0[i], x1[j], t, torch.zeros(B, d), 0.0)
        coupled.append(float(u[0]))
        return t, xt, ut

    nstep = [0]

    def model_step(t, xt, ut):
        opt.zero_grad(set_to_none=True)
        loss = torch.mean((model(torch.cat([xt, t[:, None]], dim=-1)) - ut) ** 2)
        loss.backward(); opt.step(); nstep[0] += 1

    pre = CouplingPrefetcher(None, torch.device("cpu"), workers=2)
    elapsed, gathered = bench.timed_region(D, lambda: None, pool, warm, steps, couple, model_step, draw, pre, 2,
                                           torch.device("cpu"))
    pre.close()
    # K timed steps = K couplings + K model updates (plus the warm-up's), draws in submission order
    assert nstep[0] == warm + steps and len(coupled) == warm + steps and len(drawn_log) == warm + steps
    assert sorted(coupled) == sorted(drawn_log)
    assert gathered.shape == (world * B, d)
    # the same loop strictly sequential gives the same final samples on this rank (same RNG stream)
    np.random.seed(D.shard_seed(1, rank)); torch.manual_seed(D.shard_seed(1, rank))
    last = None
    for k in range(warm + steps):
        x0, x1 = pool[k % len(pool)] if k < warm else pool[(warm + (k - warm)) % len(pool)]
        last = couple(x0, x1, draw())
    assert torch.equal(gathered[rank * B:(rank + 1) * B], last[1])
    # DDP kept the replicas identical
    flat = torch.cat([p.detach().flatten() for p in model.parameters()])
    both = D.all_gather_samples(flat[None, :])
    assert torch.equal(both[0], both[1])
    out[rank] = (elapsed, float(gathered.sum()))
    dist.destroy_process_group()


def test_bench_loop_world_size_2_gloo():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_bench_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert len(out) == 2
    assert out[0][0] == out[1][0] > 0.0          # max-over-ranks time is the same number on both ranks
    assert out[0][1] == out[1][1]                # both hold the same gathered samples


def _direct_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    from cfm_amd import distributed as D
    x = torch.full((5, 3), float(rank)) + torch.arange(15.0).reshape(5, 3) / 100
    a = D.all_gather_samples(x)
    b = D.all_gather_samples(x, direct=True)
    assert torch.equal(a, b)
    for r in range(world):
        assert torch.equal(b[5 * r:5 * r + 5], torch.full((5, 3), float(r)) + torch.arange(15.0).reshape(5, 3) / 100)
    out[rank] = float(b.sum())
    dist.destroy_process_group()


def test_direct_all_gather_world_size_3_gloo():
    """The fully connected point-to-point all-gather with an odd world size: every rank ends up with every
    block in rank order, identical to the collective."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_direct_worker, args=(3, _free_port(), out), nprocs=3, join=True)
    assert len(out) == 3 and out[0] == out[1] == out[2]
