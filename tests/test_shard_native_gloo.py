"""CPU (-m "not gpu"): the C++ sharded trial loop (super4pcs_amd/csrc/s4p_shard.cpp) over torch.distributed (gloo) with
world sizes 2 and 4.  No GPU: every rank replays recorded outcomes of its own trials through s4p_shard_replay, i.e.
through the SAME window loop (key packing, all-reduce(MAX), winner broadcast, commit, termination, pipelining) that
s4p_shard_run_windows runs over a real matcher, with the collective supplied through the s4p_collective callbacks.
The commits every rank performs must be exactly those of the sequential reference loop
(match4pcsBase.hpp:236-256 "stop at the first crossing trial", :467-484 "first strictly greater LCP wins")."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_Q = 100


def outcome_table(seed, n_trials):
    rng = np.random.default_rng(seed)
    # (found, inlier count, usable = pairs1/pairs2/quads all non-empty)
    return [(bool(rng.random() < 0.9), int(rng.integers(0, N_Q)), bool(rng.random() < 0.85)) for _ in range(n_trials)]


def sequential_commits(table, start_best, threshold_count):
    """What Perform_N_steps does with these outcomes: commit on every strictly greater count, stop after a crossing one."""
    best, commits = start_best, []
    for t, (found, count, usable) in enumerate(table):
        if found and usable and count > best:
            best = count
            commits.append((t, count))
        if best > threshold_count:
            break
    return commits


def _worker(rank, world, port, seed, n_windows, threshold_count, depth, q, fail_at=None, fail_commit=None):
    sys.path.insert(0, ROOT)
    if fail_commit is not None:
        os.environ["S4P_TEST_FAIL_COMMIT"] = "%d:%d" % fail_commit      # (rank, k): that rank's k-th commit fails (s4p_shard_replay)
    import torch.distributed as dist
    from super4pcs_amd import capi
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    table = outcome_table(seed, n_windows * world)
    found, results = [], []
    for w in range(n_windows):
        f, count, usable = table[w * world + rank]
        r = capi.BaseResult()
        if f and usable:
            r.n_pairs1 = r.n_pairs2 = r.n_quads = 10
            r.n_verified = 5
            r.best_count = count
            r.has_best = 1
            r.best_quad[0] = w * world + rank
        if fail_at == (rank, w):
            f = True
            r.n_quads = 2 ** 64 - 1                     # this rank's device pass of window w "fails"
        found.append(f)
        results.append(r)
    coll = capi.torch_collective(dist)
    try:
        commits, terminated, trials_done = capi.shard_replay(rank, world, coll, found, results, depth, threshold_count, 3)
        q.put((rank, commits, terminated, trials_done))
    except capi.S4PError as e:
        q.put((rank, "error", e.code, 0))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, seed, n_windows, threshold_count, depth, fail_at=None, fail_commit=None):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, seed, n_windows, threshold_count, depth, q, fail_at, fail_commit)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("world,seed,depth", [(2, 1, 3), (2, 2, 1), (4, 3, 3), (4, 5, 2)])
def test_native_loop_commits_equal_the_sequential_loop(world, seed, depth, s4p_lib_built):
    n_windows = 12
    res = _run(world, seed, n_windows, N_Q, depth)                      # threshold never crossed
    want = sequential_commits(outcome_table(seed, n_windows * world), 3, N_Q)
    assert want, "seed produces no improvement; pick another"
    for rank, commits, terminated, trials_done in res:
        # one commit per improving WINDOW: the window's winner is the earliest trial with the window's greatest count, which
        # is what the sequential loop ends the window with; intermediate improvements inside a window are not replayed
        seq_by_window = {}
        for t, c in want:
            seq_by_window[t // world] = (t, c)
        assert commits == [seq_by_window[w] for w in sorted(seq_by_window)]
        assert not terminated and trials_done == n_windows * world


@pytest.mark.parametrize("world,seed", [(2, 4), (2, 8), (4, 6), (4, 21)])
def test_native_loop_stops_committing_at_the_first_crossing_trial(world, seed, s4p_lib_built):
    n_windows, thr = 16, 89
    table = outcome_table(seed, n_windows * world)
    want = sequential_commits(table, 3, thr)
    assert want and want[-1][1] > thr and want[-1][0] < (n_windows - 3) * world, "seed does not terminate early; pick another"
    res = _run(world, seed, n_windows, thr, 3)
    for rank, commits, terminated, trials_done in res:
        assert terminated
        assert commits[-1] == want[-1]                                   # the crossing trial, not a later / greater one
        seq_by_window = {}
        for t, c in want:
            seq_by_window[t // world] = (t, c)
        assert commits == [seq_by_window[w] for w in sorted(seq_by_window)]


@pytest.mark.parametrize("world,fail_rank,fail_window,depth", [(2, 1, 5, 3), (2, 0, 0, 1), (4, 2, 11, 3), (4, 3, 7, 2)])
def test_a_failing_rank_takes_every_rank_out_of_the_loop(world, fail_rank, fail_window, depth, s4p_lib_built):
    """One rank's device pass fails (a refused buffer growth, a HIP error): it still owes the others its all-reduce of that
    window.  It posts the error key, so every rank returns an error from the same window instead of blocking in a
    collective for ever (the run would hit this test's 180 s queue timeout)."""
    n_windows = 12
    res = _run(world, 1, n_windows, N_Q, depth, fail_at=(fail_rank, fail_window))
    assert len(res) == world
    for rank, tag, code, _ in res:
        assert tag == "error"
        assert code == (-5 if rank == fail_rank else -7)        # the failing rank keeps its own error; the others: S4P_ERR_STATE


@pytest.mark.parametrize("world,depth,fail_offset", [(2, 1, 1), (2, 3, 1), (4, 2, 1), (4, 3, 1), (2, 2, 2)])
def test_a_rank_failing_right_after_an_improving_window(world, depth, fail_offset, s4p_lib_built):
    """The failing rank has already posted the window before the one it fails in; if that window improved the best LCP the
    healthy ranks BROADCAST its winner between their next two all-reduces.  The failing rank has to take part in that
    broadcast on its way out (it used to answer it with an all-reduce, i.e. a hang or garbage)."""
    n_windows, seed = 12, 1
    table = outcome_table(seed, n_windows * world)
    improving = sorted({t // world for t, _ in sequential_commits(table, 3, N_Q)})
    w = next(x for x in improving if 0 < x + fail_offset < n_windows - 1)
    owner = next(t for t, _ in sequential_commits(table, 3, N_Q) if t // world == w) % world
    fail_rank = (owner + 1) % world                                   # not the broadcast root
    res = _run(world, seed, n_windows, N_Q, depth, fail_at=(fail_rank, w + fail_offset))
    assert len(res) == world
    for rank, tag, code, _ in res:
        assert tag == "error"
        assert code == (-5 if rank == fail_rank else -7)
    # ... and once with the failing rank BEING the root of that broadcast
    res = _run(world, seed, n_windows, N_Q, depth, fail_at=(owner, w + fail_offset))
    for rank, tag, code, _ in res:
        assert tag == "error" and code == (-5 if rank == owner else -7)


@pytest.mark.parametrize("world,depth,which", [(2, 1, "first"), (2, 3, "middle"), (4, 2, "middle"), (4, 3, "last"), (2, 1, "last")])
def test_a_commit_that_fails_on_one_rank_takes_every_rank_out(world, depth, which, s4p_lib_built):
    """a commit that fails on ONE rank is a local failure like a failed pass -- the other ranks committed the
    same window successfully and go on.  In the middle of a call the failing rank completes the window it has already posted
    and answers the next two reductions with the error key; when the failing commit belongs to the LAST improving window --
    possibly the last window of the call, where no reduction is left to carry it -- the status reduction that closes every
    call tells the others.  Either way every rank returns an error and nobody blocks (180 s queue timeout)."""
    n_windows, seed = 12, 1
    table = outcome_table(seed, n_windows * world)
    improving = sorted({t // world for t, _ in sequential_commits(table, 3, N_Q)})
    assert len(improving) >= 2
    k = {"first": 0, "middle": len(improving) // 2, "last": len(improving) - 1}[which]
    fail_rank = world - 1
    res = _run(world, seed, n_windows, N_Q, depth, fail_commit=(fail_rank, k))
    assert len(res) == world
    for rank, tag, code, _ in res:
        assert tag == "error"
        assert code == (-5 if rank == fail_rank else -7)


def test_a_commit_that_fails_in_the_last_window_of_the_call(s4p_lib_built):
    """The same with the improving window being the very last one of the call: only the closing status reduction is left."""
    world, seed = 2, 1
    table = outcome_table(seed, 12 * world)
    last_improving = max(t // world for t, _ in sequential_commits(table, 3, N_Q))
    n_windows = last_improving + 1                                     # the call ends with that window
    k = len({t // world for t, _ in sequential_commits(table[:n_windows * world], 3, N_Q)}) - 1
    for depth in (1, 3):
        res = _run(world, seed, n_windows, N_Q, depth, fail_commit=(0, k))
        for rank, tag, code, _ in res:
            assert tag == "error" and code == (-5 if rank == 0 else -7)


# ---- SURVEY 8e level 2: every base split over all ranks (SplitLoop in s4p_shard.cpp) -------------------------------------
def split_table(seed, n_trials, world):
    """Per trial: found (same on every rank: all ranks select the same base) and, per rank, its share's (usable, count, tag)."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_trials):
        found = bool(rng.random() < 0.9)
        shares = []
        for _r in range(world):
            usable = bool(rng.random() < 0.8)
            count = int(rng.integers(0, 12 if rng.random() < 0.5 else N_Q))       # many ties on small counts: the tag decides
            tag = (int(rng.integers(0, 1 << 12)) << 32) | int(rng.integers(0, 1 << 32))   # few distinct high words: the low word decides too
            shares.append((usable, count, tag))
        out.append((found, shares))
    return out


def split_sequential(table, start_best, threshold_count):
    best, commits = start_best, []
    for t, (found, shares) in enumerate(table):
        cand = [(c, -tag) for (u, c, tag) in shares if found and u]
        if cand:
            c, ntag = max(cand)                                     # greatest count, then smallest tag: the base's first maximum
            if c > best:
                best = c
                commits.append((t, c, -ntag))
        if best > threshold_count:
            break
    return commits


def _split_worker(rank, world, port, seed, n_trials, threshold_count, depth, q, fail_at=None, fail_commit=None):
    sys.path.insert(0, ROOT)
    if fail_commit is not None:
        os.environ["S4P_TEST_FAIL_COMMIT"] = "%d:%d" % fail_commit
    import torch.distributed as dist
    from super4pcs_amd import capi
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    table = split_table(seed, n_trials, world)
    found, results = [], []
    for t, (f, shares) in enumerate(table):
        usable, count, tag = shares[rank]
        r = capi.BaseResult()
        r.n_pairs1 = r.n_pairs2 = 10
        if usable:
            r.n_quads = 10; r.n_verified = 5; r.best_count = count; r.has_best = 1; r.best_rank = tag
            r.best_quad[0] = t; r.best_quad[1] = rank
        if fail_at == (rank, t):
            f = True
            r.n_quads = 2 ** 64 - 1
        found.append(f)
        results.append(r)
    coll = capi.torch_collective(dist)
    try:
        commits, terminated, trials_done = capi.shard_replay_split(rank, world, coll, found, results, depth, threshold_count, 3)
        q.put((rank, commits, terminated, trials_done))
    except capi.S4PError as e:
        q.put((rank, "error", e.code, 0))
    dist.barrier()
    dist.destroy_process_group()


def _run_split(world, seed, n_trials, threshold_count, depth, fail_at=None, fail_commit=None):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_split_worker, args=(r, world, port, seed, n_trials, threshold_count, depth, q, fail_at, fail_commit)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("world,seed,depth,thr", [(2, 1, 3, N_Q), (4, 2, 1, N_Q), (4, 3, 3, 95), (2, 9, 2, 93)])
def test_split_base_loop_commits_equal_the_sequential_loop(world, seed, depth, thr, s4p_lib_built):
    """Every base split over all ranks: the two all-reduces must pick, per base, the greatest count and among equal counts the
    smallest order tag over ALL shares (the reference's first maximum), commit it only when it improves the best, and stop at
    the first base that crosses the terminate threshold -- on every rank alike."""
    n_trials = 40
    want = split_sequential(split_table(seed, n_trials, world), 3, thr)
    assert len(want) >= 2, "seed produces no improvements; pick another"
    for rank, commits, terminated, trials_done in _run_split(world, seed, n_trials, thr, depth):
        assert commits == want
        assert terminated == (want[-1][1] > thr)


def test_split_base_loop_a_failing_rank_takes_every_rank_out(s4p_lib_built):
    res = _run_split(3, 5, 20, N_Q, 2, fail_at=(1, 6))
    for rank, tag, code, _ in res:
        assert tag == "error" and code == (-5 if rank == 1 else -7)


@pytest.mark.parametrize("which,depth", [("first", 2), ("last", 2), ("first", 1), ("last", 1), ("first", 3)])
def test_split_base_loop_a_failing_commit_takes_every_rank_out(which, depth, s4p_lib_built):
    """A commit that fails on one rank of the split loop: mid-call the next reduction carries the error key, after the last
    improving base of the call the closing status reduction does.  At depth 1 nothing is in flight when the commit
    fails: the failing rank answers the others' next per-trial reduction before it closes the call (the round-5
    library hangs the other ranks here)."""
    world, seed, n_trials = 3, 1, 40
    want = split_sequential(split_table(seed, n_trials, world), 3, N_Q)
    assert len(want) >= 2
    k = 0 if which == "first" else len(want) - 1
    res = _run_split(world, seed, n_trials, N_Q, depth, fail_commit=(2, k))
    for rank, tag, code, _ in res:
        assert tag == "error" and code == (-5 if rank == 2 else -7)
