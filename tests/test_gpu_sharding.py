"""-m gpu: the multi-rank driver with the real engine.  Two processes share the one GPU of the test box and talk
over gloo (the collective is 8 bytes, its transport is irrelevant to correctness); each owns every second trial.
Both must end in exactly the state the sequential CPU oracle reaches after the same number of trials."""
import os
import socket
import sys

import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DELTA, OVERLAP, N_S, N_WINDOWS = 0.01, 0.6, 250, 12


def _worker(rank, world, port, producer, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from super4pcs_amd import capi, sharding
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    P, Q, _ = H.small_pair(30000, delta=DELTA, seed=23)
    m = capi.Matcher(capi.make_options(DELTA, OVERLAP, N_S), device=0, max_pairs=1 << 20, max_quads=4 << 20)
    m.init_full(P, Q)
    sh = sharding.ShardedRansac(m, rank, world, dist, None, producer_threads=producer)
    if not producer:
        m.set_sharding(rank, world, False)
    got = sh.run_windows(N_WINDOWS)
    i = m.info()
    q.put((rank, float(i.best_lcp), list(i.base), list(i.congruent), list(i.transform), int(got)))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("producer", [False, True])
def test_two_ranks_match_the_sequential_oracle(oracle_mod, s4p_lib_built, producer):
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, producer, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    O = oracle_mod
    P, Q, _ = H.small_pair(30000, delta=DELTA, seed=23)
    om = O.Matcher(O.make_options(DELTA, OVERLAP, N_S))
    om.init(P, Q)
    for _ in range(N_WINDOWS * world):
        om.try_one_base()
    T, lcp, base, cong, _, _ = om.best()
    for r in res:
        assert r[1] == lcp and r[2] == base.tolist() and r[3] == cong.tolist()
        assert np.array_equal(np.array(r[4], np.float32).reshape(4, 4), T)
    assert res[0][5] + res[1][5] == om.stats().n_verified        # every candidate verified exactly once, on one rank
    assert res[0][5] > 0 and res[1][5] > 0


def _rccl_worker(port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from super4pcs_amd import capi, sharding
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
    P, Q, _ = H.small_pair(30000, delta=DELTA, seed=23)
    m = capi.Matcher(capi.make_options(DELTA, OVERLAP, N_S), device=0, max_pairs=1 << 20, max_quads=4 << 20)
    m.init_full(P, Q)
    sh = sharding.ShardedRansac(m, 0, 1, dist, dev, producer_threads=True, force_windows=True)
    got = sh.run_windows(2 * N_WINDOWS)
    i = m.info()
    q.put((float(i.best_lcp), list(i.base), list(i.congruent), list(i.transform), int(got)))
    dist.barrier()
    dist.destroy_process_group()


def test_windowed_driver_over_rccl(oracle_mod, s4p_lib_built):
    """The collective path as the multi-GPU job runs it -- pinned key, non-blocking copies, all_reduce(MAX) and the
    winner broadcast on device tensors over RCCL -- with the one rank a single-GPU box allows."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    r = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    O = oracle_mod
    P, Q, _ = H.small_pair(30000, delta=DELTA, seed=23)
    om = O.Matcher(O.make_options(DELTA, OVERLAP, N_S))
    om.init(P, Q)
    for _ in range(2 * N_WINDOWS):
        om.try_one_base()
    T, lcp, base, cong, _, _ = om.best()
    assert r[0] == lcp and r[1] == base.tolist() and r[2] == cong.tolist()
    assert np.array_equal(np.array(r[3], np.float32).reshape(4, 4), T)
    assert r[4] == om.stats().n_verified


# ---- the C++ sharded loop behind the C ABI (s4p_shard_*, super4pcs_amd/csrc/s4p_shard.cpp) ------------------------------
def _native_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from super4pcs_amd import capi
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    P, Q, _ = H.small_pair(30000, delta=DELTA, seed=23)
    m = capi.Matcher(capi.make_options(DELTA, OVERLAP, N_S), device=0, max_pairs=1 << 20, max_quads=4 << 20)
    m.init_full(P, Q)
    sh = capi.Shard(m, rank, world, producer_threads=True)
    sh.use_collective(capi.torch_collective(dist))          # the s4p_collective callback provider, over gloo
    got = sh.run_windows(N_WINDOWS)
    i = m.info()
    q.put((rank, float(i.best_lcp), list(i.base), list(i.congruent), list(i.transform), int(got)))
    dist.barrier()
    dist.destroy_process_group()


def test_native_loop_two_ranks_match_the_sequential_oracle(oracle_mod, s4p_lib_built):
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_native_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    O = oracle_mod
    P, Q, _ = H.small_pair(30000, delta=DELTA, seed=23)
    om = O.Matcher(O.make_options(DELTA, OVERLAP, N_S))
    om.init(P, Q)
    for _ in range(N_WINDOWS * world):
        om.try_one_base()
    T, lcp, base, cong, _, _ = om.best()
    for r in res:
        assert r[1] == lcp and r[2] == base.tolist() and r[3] == cong.tolist()
        assert np.array_equal(np.array(r[4], np.float32).reshape(4, 4), T)
    assert res[0][5] + res[1][5] == om.stats().n_verified and res[0][5] > 0 and res[1][5] > 0


def _native_rccl_worker(q):
    sys.path.insert(0, ROOT)
    import torch                                             # noqa: F401  (loads the process's RCCL first, as bench.py does)
    from super4pcs_amd import capi
    P, Q, _ = H.small_pair(30000, delta=DELTA, seed=23)
    m = capi.Matcher(capi.make_options(DELTA, OVERLAP, N_S), device=0, max_pairs=1 << 20, max_quads=4 << 20)
    sh = capi.Shard(m, 0, 1, producer_threads=True)
    sh.use_rccl(0, capi.rccl_unique_id())                   # ncclCommInitRank + ncclAllReduce/ncclBroadcast from C++ (rccl.h)
    lcp, M, Qt = sh.compute_transformation(P, Q)            # whole ComputeTransformation through the sharded entry point
    i = m.info()
    q.put((float(lcp), M.tolist(), int(i.candidates_verified), int(i.bases_tried)))


def test_native_loop_over_rccl_whole_registration(oracle_mod, s4p_lib_built):
    """s4p_shard_compute_transformation with the built-in RCCL collective (world 1 is what a single-GPU box allows):
    same result as the sequential oracle's ComputeTransformation."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_native_rccl_worker, args=(q,))
    p.start()
    r = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    O = oracle_mod
    P, Q, _ = H.small_pair(30000, delta=DELTA, seed=23)
    om = O.Matcher(O.make_options(DELTA, OVERLAP, N_S), full_counts=False, use_kdtree=True)
    o_lcp, o_M, _ = om.compute_transformation(P, Q)
    assert r[0] == o_lcp and np.max(np.abs(np.array(r[1], np.float32) - o_M)) <= 1e-4
    assert r[2] == om.stats().n_verified


# ---- SURVEY 8e level 2: every base split over the ranks (s4p_shard_set_mode(1), s4p_set_quad_slice) ---------------------
def _split_worker(rank, world, port, n_trials, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from super4pcs_amd import capi
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    P, Q, _ = H.small_pair(30000, delta=DELTA, seed=23)
    m = capi.Matcher(capi.make_options(DELTA, OVERLAP, N_S), device=0, max_pairs=1 << 20, max_quads=4 << 20)
    sh = capi.Shard(m, rank, world, producer_threads=True)
    sh.set_mode(True)                                         # this rank: its share of every base's second pair set
    m.init_full(P, Q)
    sh.use_collective(capi.torch_collective(dist))
    got = sh.run_windows(n_trials)                            # split mode: one trial per window
    i = m.info()
    q.put((rank, float(i.best_lcp), list(i.base), list(i.congruent), list(i.transform), int(got), int(i.quads_total)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_split_bases_over_ranks_match_the_sequential_oracle(oracle_mod, s4p_lib_built, world):
    """Every base over all ranks: each rank enumerates, gates and scores its share of the base's second pair set, two
    all-reduces pick the first maximum among the shares.  Every rank must end in the sequential oracle's state, every
    candidate (and every quad) must have been handled on exactly one rank."""
    import torch.multiprocessing as mp
    n_trials = 16
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_split_worker, args=(r, world, port, n_trials, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    O = oracle_mod
    P, Q, _ = H.small_pair(30000, delta=DELTA, seed=23)
    om = O.Matcher(O.make_options(DELTA, OVERLAP, N_S))
    om.init(P, Q)
    for _ in range(n_trials):
        om.try_one_base()
    T, lcp, base, cong, _, _ = om.best()
    for r in res:
        assert r[1] == lcp and r[2] == base.tolist() and r[3] == cong.tolist()
        assert np.array_equal(np.array(r[4], np.float32).reshape(4, 4), T)
    assert sum(r[5] for r in res) == om.stats().n_verified and all(r[5] > 0 for r in res)
    assert sum(r[6] for r in res) == om.stats().n_quads
