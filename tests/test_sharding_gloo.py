"""CPU (-m "not gpu"): the N>1 path over torch.distributed (gloo, world_size 2): ShardedRansac driving a
deterministic stand-in matcher must end in exactly the state the sequential loop reaches."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FakeOpt:
    terminate_threshold = 1.0


class FakeInfo:
    def __init__(self, n, best):
        self.n_sampled_q, self.best_count = n, best


class FakeMatcher:
    """Same interface as capi.Matcher for the sharding driver; trial outcomes come from a seeded table so every
    rank 'selects' the same bases, and only the owner 'computes' a result."""

    def __init__(self, seed, n_trials, n_q=100, thr=1.0):
        from super4pcs_amd import capi
        self.capi = capi
        rng = np.random.default_rng(seed)
        self.table = [(bool(rng.random() < 0.9), int(rng.integers(0, n_q)), bool(rng.random() < 0.85)) for _ in range(n_trials)]
        self.t = 0
        self.n_q = n_q
        self.best = 3
        self.best_trial = -1
        self.opt = FakeOpt()
        self.opt.terminate_threshold = thr
        self.pending = []
        self.log = []

    def info(self):
        return FakeInfo(self.n_q, self.best)

    def _result(self, t):
        found, count, usable = self.table[t]
        r = self.capi.BaseResult()
        if found and usable:
            r.n_pairs1 = r.n_pairs2 = r.n_quads = 10
            r.n_verified = 5
            r.best_count = count
            r.has_best = 1
            r.best_quad[0] = t
            r.best_transform[0] = float(t)
        return r

    def next_base(self, run_device=True):
        t = self.t
        self.t += 1
        found = self.table[t][0]
        base = np.array([t, 0, 0, 0], np.int32)
        return found, base, (self._result(t) if (run_device and found) else self.capi.BaseResult())

    def next_base_async(self, run_device=True):
        found, base, r = self.next_base(run_device)
        if found:
            self.pending.append(r)
        return found, base

    def wait_base(self):
        return self.pending.pop(0)

    def commit(self, found, base, r):
        if found and r.n_pairs1 and r.n_pairs2 and r.n_quads and r.has_best and r.best_count > self.best:
            self.best = int(r.best_count)
            self.best_trial = int(base[0])
            assert int(r.best_quad[0]) == self.best_trial and int(r.best_transform[0]) == self.best_trial
        return self.best / self.n_q > self.opt.terminate_threshold


def sequential(seed, n_trials, thr):
    m = FakeMatcher(seed, n_trials, thr=thr)
    for _ in range(n_trials):
        found, base, r = m.next_base(True)
        if m.commit(found, base, r):
            break
    return m.best, m.best_trial


def _worker(rank, world, port, seed, n_windows, thr, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from super4pcs_amd import sharding
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    m = FakeMatcher(seed, n_windows * world, thr=thr)
    sh = sharding.ShardedRansac(m, rank, world, dist, None)
    sh.run_windows(n_windows)
    q.put((rank, m.best, m.best_trial, sh.local_candidates))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_two_ranks_reach_the_sequential_result(seed):
    import torch.multiprocessing as mp
    world, n_windows = 2, 12
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, seed, n_windows, 1.0, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = sequential(seed, n_windows * world, 1.0)
    assert (res[0][1], res[0][2]) == want and (res[1][1], res[1][2]) == want      # both ranks hold the same winner
    # every verified candidate is counted exactly once across ranks
    m = FakeMatcher(seed, n_windows * world)
    total = sum(5 for (f, c, u) in m.table if f and u)
    assert res[0][3] + res[1][3] == total


@pytest.mark.parametrize("seed", [4, 6, 8, 21])
def test_termination_inside_a_window_matches_the_sequential_break(seed):
    """The sequential loop stops at the first trial whose LCP crosses the terminate threshold (match4pcsBase.hpp:255);
    the sharded driver has later windows in flight by then and must drain them without committing."""
    import torch.multiprocessing as mp
    world, n_windows, thr = 2, 16, 0.9
    want = sequential(seed, n_windows * world, thr)
    probe = FakeMatcher(seed, n_windows * world, thr=thr)
    assert want[0] / probe.n_q > thr and want[1] < n_windows * world - 2 * world, "seed does not terminate early; pick another"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, seed, n_windows, thr, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert (res[0][1], res[0][2]) == want and (res[1][1], res[1][2]) == want
