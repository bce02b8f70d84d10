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
