"""Coupling prefetch: overlap the minibatch-OT couplings of the NEXT batches with the model step.

The coupling of a minibatch (cost matrix -> OT plan -> sampled pairs -> x_t, u_t) depends only on
the data, never on the model, so a training loop can compute it ahead of time exactly like a
data-loader worker — the reference does this work synchronously on the host inside
``FM.sample_location_and_conditional_flow`` (torchcfm/conditional_flow_matching.py:271-272).
Here every worker thread owns a HIP stream.  The exact-assignment solver is a chain of small
latency-bound kernels (one-workgroup control steps, a one-workgroup tail solver) that leaves most
CUs idle most of the time: a second coupling in flight fills those holes, and so do the model's
GEMMs.  Random draws stay on the submitting thread (``draw=`` callback) so the global
``np.random`` / ``torch`` CPU generators are consumed in submission order whatever the workers do.
"""
import concurrent.futures as _cf
import threading

import torch

from . import streams as _streams


class CouplingPrefetcher:
    """``submit(x0, x1)`` -> handle; ``handle.result()`` -> the tensors, ready for the caller's
    current stream.  ``workers`` couplings can be in flight at once."""

    def __init__(self, flow_matcher, device=None, workers=1, partition=None, priority=0, blocking_sync=True):
        """partition: a ``cfm_amd.streams.ChipPartition`` — every worker then owns TWO streams: one on the partition's
        dense CU subset (cost matrix, sampling, x_t / u_t) and one on its solver subset, onto which the exact
        solver's launches are redirected (``streams.solver_stream``); run the model step on
        ``partition.dense_stream()`` too and the solver's rounds no longer wait for slots the dense products hold.
        priority: HIP stream priority of the workers' streams when no partition is given (-1 = high).
        blocking_sync: the workers' host waits for the exact solver (one per job) sleep in the driver
        (``cfm_set_blocking_sync``) instead of spinning on a host core each — 8 ranks x 3 workers would otherwise keep
        24 cores busy doing nothing."""
        self.fm = flow_matcher
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.partition, self.priority = partition, int(priority)
        self.blocking_sync = bool(blocking_sync)
        self.workers = max(1, int(workers))
        self._tls = threading.local()
        self._pool = _cf.ThreadPoolExecutor(max_workers=self.workers, thread_name_prefix="cfm-coupling")

    def _stream(self):
        s = getattr(self._tls, "stream", None)
        if s is None:
            torch.cuda.set_device(self.device)
            from . import _lib
            _lib.load().cfm_set_blocking_sync(1 if self.blocking_sync else 0)      # thread-local: this worker's waits
            if self.partition is not None:
                s = self._tls.stream = self.partition.dense_stream()
                self._tls.solver = self.partition.solver_stream()
            else:
                s = self._tls.stream = torch.cuda.Stream(device=self.device, priority=self.priority)
                self._tls.solver = None
        return s

    def _solver_ctx(self):
        return _streams.solver_stream(getattr(self._tls, "solver", None))

    def _work(self, x0, x1, ready, hook, drawn):
        if self.device.type != "cuda":           # host tensors (the CPU multi-process tests): plain worker thread
            return (hook(x0, x1, drawn) if hook is not None
                    else self.fm.sample_location_and_conditional_flow(x0, x1)), None
        stream = self._stream()
        with torch.cuda.stream(stream), self._solver_ctx():
            stream.wait_event(ready)                  # x0 / x1 were produced on the caller's stream
            if hook is not None:
                out = hook(x0, x1, drawn)
            else:
                out = self.fm.sample_location_and_conditional_flow(x0, x1)
            done = torch.cuda.Event()
            done.record(stream)
        return out, done

    def submit(self, x0, x1, hook=None, draw=None):
        """``draw()`` (optional) runs NOW, on the calling thread, and its result is handed to
        ``hook(x0, x1, drawn)`` — keep every host RNG call in there when ``workers`` > 1."""
        ready = None
        if self.device.type == "cuda":
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.device))
        drawn = draw() if draw is not None else None
        return _Handle(self._pool.submit(self._work, x0, x1, ready, hook, drawn), self.device)

    def submit_group(self, batches, hook, draw=None):
        """Several minibatches as ONE job: ``hook(batches, drawn_list)`` couples them together (the exact solver takes a
        batch of problems in one chain of launches, ``optimal_transport.assign_exact_batch``) and returns one result
        tuple per minibatch, in order.  ``draw()`` runs once per minibatch, now, on the calling thread."""
        ready = None
        if self.device.type == "cuda":
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.device))
        drawn = [draw() if draw is not None else None for _ in batches]
        return _Handle(self._pool.submit(self._work_group, list(batches), ready, hook, drawn), self.device)

    def _work_group(self, batches, ready, hook, drawn):
        if self.device.type != "cuda":
            return hook(batches, drawn), None
        stream = self._stream()
        with torch.cuda.stream(stream), self._solver_ctx():
            stream.wait_event(ready)
            out = hook(batches, drawn)
            done = torch.cuda.Event()
            done.record(stream)
        return out, done

    def prime(self, fn):
        """Run ``fn()`` once on EVERY worker thread, on that worker's stream (all workers at the same time), and wait.
        One-time costs are per worker — its stream, its workspaces, the solver's launch programs (hipGraphs are
        captured per host thread) — so a loop that wants none of them inside its first steps primes the workers with a
        throw-away job of the shape it is going to submit."""
        n = self.workers
        gate = threading.Barrier(n)

        def job():
            try:
                gate.wait(timeout=120.0)    # n jobs, n threads: nobody takes two
            except threading.BrokenBarrierError:
                raise RuntimeError(f"CouplingPrefetcher.prime: {n} worker threads did not all start within 120 s "
                                   "(is the pool busy with unfinished jobs?)") from None
            if self.device.type != "cuda":
                fn(); return
            stream = self._stream()
            with torch.cuda.stream(stream), self._solver_ctx():
                fn()
            stream.synchronize()
            if getattr(self._tls, "solver", None) is not None:
                self._tls.solver.synchronize()
        for f in [self._pool.submit(job) for _ in range(n)]:
            f.result()

    def close(self):
        self._pool.shutdown(wait=True)


class _Handle:
    def __init__(self, fut, device):
        self._fut, self._device = fut, device

    def result(self):
        out, done = self._fut.result()
        if done is None:
            return out
        cur = torch.cuda.current_stream(self._device)
        cur.wait_event(done)
        def mark(o):
            if isinstance(o, torch.Tensor):
                if o.is_cuda:
                    o.record_stream(cur)
            elif isinstance(o, (list, tuple)):
                for x in o:
                    mark(x)
        mark(out)
        return out
