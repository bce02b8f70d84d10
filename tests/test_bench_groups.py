"""CPU: bench.run_steps with grouped prefetch jobs (--group G) — every step still is one coupling + one model update, in
the pool's order, with the host draws taken in submission order; the result equals the sequential loop's."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _setup():
    import cfm_oracle as oracle
    B, d = 24, 2
    pool = [oracle.config_inputs("C1", B=B, rank=k) for k in range(5)]
    log = {"coupled": [], "stepped": [], "groups": []}

    def draw():
        return np.random.random_sample(B), torch.rand(B)

    def couple(x0, x1, drawn):
        u, t = drawn
        perm = oracle.exact_perm(oracle.ref_cost_f32(x0, x1))
        i, j = oracle.sample_perm_given_u(perm, u)
        xt, ut = oracle.xt_ut("icfm", x0[i], x1[j], t, torch.zeros(B, d), 0.0)
        log["coupled"].append(float(u[0]))
        return t, xt, ut

    def couple_group(batches, drawn):
        log["groups"].append(len(batches))
        return [couple(a, b, dr) for (a, b), dr in zip(batches, drawn)]

    def model_step(t, xt, ut):
        log["stepped"].append(float(xt.sum()))

    return pool, draw, couple, couple_group, model_step, log


def test_grouped_run_steps_equals_sequential():
    import bench
    from cfm_amd.prefetch import CouplingPrefetcher
    pool, draw, couple, couple_group, model_step, log = _setup()
    np.random.seed(1); torch.manual_seed(1)
    last_seq = bench.run_steps(pool, 2, 11, couple, model_step, draw)
    seq = list(log["stepped"]); log["stepped"].clear(); log["coupled"].clear()
    np.random.seed(1); torch.manual_seed(1)
    pre = CouplingPrefetcher(None, torch.device("cpu"), workers=2)
    last = bench.run_steps(pool, 2, 11, couple, model_step, draw, pre, 2, 4, couple_group)
    pre.close()
    assert log["stepped"] == seq                       # same batches, same draws, same order of model updates
    assert log["groups"] == [4, 4, 3] and len(log["coupled"]) == 11
    for a, b in zip(last, last_seq):
        assert torch.equal(a, b)


def test_group_of_one_and_no_prefetcher_fall_back_to_the_plain_loops():
    import bench
    from cfm_amd.prefetch import CouplingPrefetcher
    pool, draw, couple, couple_group, model_step, log = _setup()
    np.random.seed(2); torch.manual_seed(2)
    bench.run_steps(pool, 0, 5, couple, model_step, draw, None, 0, 4, couple_group)
    assert log["groups"] == [] and len(log["stepped"]) == 5
    pre = CouplingPrefetcher(None, torch.device("cpu"), workers=2)
    bench.run_steps(pool, 0, 5, couple, model_step, draw, pre, 2, 1, couple_group)
    pre.close()
    assert log["groups"] == [] and len(log["stepped"]) == 10


def test_prime_runs_once_on_every_worker():
    import threading
    from cfm_amd.prefetch import CouplingPrefetcher
    pre = CouplingPrefetcher(None, torch.device("cpu"), workers=3)
    seen = []
    pre.prime(lambda: seen.append(threading.get_ident()))
    pre.close()
    assert len(seen) == 3 and len(set(seen)) == 3


def test_ramped_first_jobs_keep_order_and_results():
    """--ramp: the first prefetch jobs of a run are smaller (the pipeline starts empty inside the timed call); the steps,
    their order and their results are those of the sequential loop."""
    import bench
    from cfm_amd.prefetch import CouplingPrefetcher
    pool, draw, couple, couple_group, model_step, log = _setup()
    np.random.seed(3); torch.manual_seed(3)
    bench.run_steps(pool, 1, 12, couple, model_step, draw)
    seq = list(log["stepped"]); log["stepped"].clear(); log["coupled"].clear()
    np.random.seed(3); torch.manual_seed(3)
    pre = CouplingPrefetcher(None, torch.device("cpu"), workers=3)
    bench.run_steps(pool, 1, 12, couple, model_step, draw, pre, 3, 4, couple_group, ramp=(1, 2))
    pre.close()
    assert log["stepped"] == seq
    assert log["groups"] == [1, 2, 4, 4, 1]


def test_job_sizes_cover_the_steps_with_ramp_and_tail():
    """bench.job_sizes: ramp first, tail last (dropped from its front when the steps do not suffice), `group` between."""
    import bench
    assert bench.job_sizes(20, 4, (1,), ()) == [1, 4, 4, 4, 4, 3]
    assert bench.job_sizes(20, 4, (1,), (2, 1)) == [1, 4, 4, 4, 4, 2, 1]
    assert bench.job_sizes(20, 4, (1, 2), (2, 1)) == [1, 2, 4, 4, 4, 2, 2, 1]
    assert bench.job_sizes(20, 4) == [4] * 5
    for count in range(1, 30):
        for ramp in ((), (1,), (1, 3), (4,)):
            for tail in ((), (1,), (2, 1), (3, 3)):
                s = bench.job_sizes(count, 4, ramp, tail)
                assert sum(s) == count and all(1 <= k <= 4 for k in s), (count, ramp, tail, s)
