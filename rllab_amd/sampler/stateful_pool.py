"""Worker pool with per-process state -- the API of rllab/sampler/stateful_pool.py:40-198 for envs that live on
the CPU (arbitrary Python envs; HIP-native envs are sampled by the lock-step ``VectorizedSampler`` and never
come here).

Conventions kept from the reference: ``singleton_pool``; every runner receives the worker's private ``G``
(a ``SharedGlobal``) as its first argument; ``run_each`` executes exactly once on every worker; ``run_collect``
keeps calling ``collect_once(G, *args) -> (item, increment)`` on all workers until the increments reach a
threshold and returns the concatenated items (the total may overshoot: workers finish what they started);
exceptions raised in a worker reach the caller as ``Exception(<formatted traceback>)``; workers are processes
(fork), never threads, and are kept away from the GPU.  With ``n_parallel == 1`` everything runs inline on the
pool's own ``G``.

Built differently from the reference (a joblib pool driven by ``map_async`` plus two hand-shake queues so that
no worker takes two ``run_each`` tasks, and a master that polls a managed counter every 0.1 s): each worker is a
dedicated process behind its own pipe, so "once per worker" holds by construction, the sample counter is a
lock-protected shared triple (samples so far, collects in flight, collects done) created before the fork, and the master simply blocks on the replies.
"""
import multiprocessing as mp
import os
import sys
import time
import traceback


class SharedGlobal(object):
    """Per-process attribute bag handed to every runner."""


def _collect_until(fn, G, args, threshold, shared):
    """One worker's share of ``run_collect``.  ``shared`` = [sum of increments, collects in flight, collects done].
    A worker starts another collect only while the total would still fall short of the threshold if every collect
    in flight brought the average increment seen so far -- so the pool does not overshoot by a path per worker
    the way a plain check-then-collect loop does (unit increments stop exactly at the threshold)."""
    out = []
    while True:
        with shared.get_lock():
            total, inflight, done = shared[0], shared[1], shared[2]
            if total >= threshold:
                return out
            expect = (total / done) if done else 1.0
            go = total + inflight * max(expect, 1.0) < threshold
            if go:
                shared[1] = inflight + 1
        if not go:
            time.sleep(2e-4)          # enough is in flight: wait for it to land (or to fall short)
            continue
        try:
            item, inc = fn(G, *args)
        except Exception:
            with shared.get_lock():
                shared[1] -= 1
            raise
        out.append(item)
        with shared.get_lock():
            shared[0] += inc
            shared[1] -= 1
            shared[2] += 1


def _worker_main(conn, counter, G):
    # workers never touch the GPU: the parent owns the HIP context (reference: parallel_sampler.py:10-15)
    os.environ["CUDA_VISIBLE_DEVICES"] = ""
    os.environ["HIP_VISIBLE_DEVICES"] = ""
    os.environ["ROCR_VISIBLE_DEVICES"] = ""
    while True:
        try:
            msg = conn.recv()
        except (EOFError, KeyboardInterrupt):
            return
        if msg is None:
            return
        kind, fn, payload = msg
        try:
            if kind == "call":
                out = fn(G, *payload)
            elif kind == "map":
                out = [fn(G, *args) for args in payload]
            else:  # "collect": payload = (threshold, args)
                threshold, args = payload
                out = _collect_until(fn, G, args, threshold, counter)
            conn.send(("ok", out))
        except Exception:
            conn.send(("error", "".join(traceback.format_exception(*sys.exc_info()))))


class StatefulPool(object):
    def __init__(self):
        self.n_parallel = 1
        self.G = SharedGlobal()
        self._procs, self._conns, self._counter = [], [], None

    # -- lifecycle ------------------------------------------------------------------------------------------
    def initialize(self, n_parallel):
        if self._procs:
            print("Warning: terminating existing pool")
            self.terminate()
            self.G = SharedGlobal()
        self.n_parallel = int(n_parallel)
        if self.n_parallel > 1:
            ctx = mp.get_context("fork")
            self._counter = ctx.Array("d", 3)      # [sum of increments, collects in flight, collects done]
            for _ in range(self.n_parallel):
                parent, child = ctx.Pipe()
                p = ctx.Process(target=_worker_main, args=(child, self._counter, SharedGlobal()), daemon=True)
                p.start()
                child.close()
                self._procs.append(p)
                self._conns.append(parent)

    def terminate(self):
        for c in self._conns:
            try:
                c.send(None)
                c.close()
            except (OSError, BrokenPipeError):
                pass
        for p in self._procs:
            p.join(timeout=1.0)
            if p.is_alive():
                p.terminate()
        self._procs, self._conns, self._counter = [], [], None
        self.n_parallel = 1

    @property
    def pool(self):
        """Truthy while worker processes exist (the reference exposes its joblib pool here)."""
        return self._procs or None

    def _gather(self):
        replies = [c.recv() for c in self._conns]
        for status, value in replies:
            if status == "error":
                raise Exception(value)
        return [value for _, value in replies]

    # -- the reference's entry points -------------------------------------------------------------------------
    def run_each(self, runner, args_list=None):
        """Run ``runner(G, *args_list[i])`` once on worker i; returns the list of results."""
        if args_list is None:
            args_list = [tuple()] * self.n_parallel
        assert len(args_list) == self.n_parallel
        if self.n_parallel > 1:
            for c, args in zip(self._conns, args_list):
                c.send(("call", runner, tuple(args)))
            return self._gather()
        return [runner(self.G, *args_list[0])]

    def run_map(self, runner, args_list):
        """``[runner(G, *args) for args in args_list]`` spread over the workers, results in input order."""
        args_list = [tuple(a) for a in args_list]
        if self.n_parallel > 1:
            n = self.n_parallel
            for w, c in enumerate(self._conns):
                c.send(("map", runner, args_list[w::n]))
            chunks = self._gather()
            out = [None] * len(args_list)
            for w, chunk in enumerate(chunks):
                out[w::n] = chunk
            return out
        return [runner(self.G, *args) for args in args_list]

    def run_imap_unordered(self, runner, args_list):
        for x in self.run_map(runner, args_list):
            yield x

    def run_collect(self, collect_once, threshold, args=None, show_prog_bar=True):
        """Collect items from all workers until the increments they report add up to ``threshold``."""
        args = tuple() if args is None else tuple(args)
        if self.n_parallel > 1:
            with self._counter.get_lock():
                self._counter[0] = self._counter[1] = self._counter[2] = 0.0
            for c in self._conns:
                c.send(("collect", collect_once, (threshold, args)))
            return sum(self._gather(), [])
        count, results = 0, []
        while count < threshold:
            item, inc = collect_once(self.G, *args)
            results.append(item)
            count += inc
        return results


singleton_pool = StatefulPool()
