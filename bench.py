#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X-native Super4PCS hot path.

Contract (driver): python bench.py --gpus N --steps K --warmup W ; for N>1 launched by
torch.distributed.run, one rank per GPU.  Rank 0 prints ONE JSON line.

Metric (BASELINE.json): candidate transforms verified / s on the synthetic 1 M-point pair
(configs[2]: 50 % overlap, Gaussian noise sigma = delta = 0.004, sample size 2000).
One "step" = one RANSAC base through the whole hot path on this rank's GPU:
   SelectQuadrilateral (host) -> ExtractPairs x2 -> FindCongruentQuadrilaterals ->
   ComputeRigidTransformation + Verify of every congruent candidate -> best selection,
i.e. Match4PCSBase::TryOneBase (match4pcsBase.hpp:281-360).  A candidate counts when it passed the
rms gate and was LCP-scored over all sampled Q points (reference counter nbCongruentAto, :441).
Inputs (sampled clouds, LCP grid) are resident in HBM before the timed region.
With N GPUs each rank owns every N-th base of the same sequence (weak scaling: K device steps per
rank) and one 8-byte all-reduce(MAX) per window over RCCL picks the winner.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_POINTS = 1_000_000
DELTA = 0.004
OVERLAP = 0.5
SAMPLE = 2000
SEED = 20140814
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec


def bytes_per_candidate(n_q, kbar, cells=27):
    """B_cand = 16 (quad read) + 8 (count write) + n_Q * (12 + c*8 + kbar*12)."""
    return 16 + 8 + n_q * (12 + cells * 8 + kbar * 12)


def cpu_baseline(P, Q, budget_s):
    """CPU path on the same workload, 1 thread (what MatchSuper4PCS does, super4pcs.cc:68-73), bounded sample.

    kind "reference": the reference's own sources (oracle/_ref/libs4p_ref.so, built from /root/reference against
    oracle/eigen_shim) run ComputeTransformation and are cut by a visitor exception after budget_s of RANSAC time.
    kind "port": the oracle restatement, if the prebuilt reference library is not in the tree.
    """
    from oracle import oracle as O
    from oracle import reflib
    if reflib.available():
        rm = reflib.RefMatcher(O.make_options(DELTA, OVERLAP, SAMPLE))
        cut, n, sec = rm.bench(P, Q, budget_s)
        return {"value": n / max(sec, 1e-9), "unit": "candidates/s", "cores": 1, "kind": "reference",
                "sample": "reference ComputeTransformation (kd-tree Verify with early exit) on the same clouds/seed, "
                          "stopped after %.1f s of RANSAC time: %d candidates verified%s" % (sec, n, "" if cut else " (ran to completion)"),
                "seconds": sec}
    O.build()
    om = O.Matcher(O.make_options(DELTA, OVERLAP, SAMPLE), full_counts=False, use_kdtree=True, keep_trace=False)
    om.init(P, Q)
    om.set_budget(budget_s)
    t0 = time.perf_counter()
    bases = 0
    while time.perf_counter() - t0 < budget_s:
        om.try_one_base()
        bases += 1
    dt = time.perf_counter() - t0
    s = om.stats()
    return {
        "value": s.n_verified / dt, "unit": "candidates/s", "cores": 1, "kind": "port",
        "sample": "first %d base(s) of the same seeded sequence, TryCongruentSet cut after %.0f s wall "
                  "(%d candidates verified, kd-tree Verify with the reference's early exit)" % (bases, budget_s, s.n_verified),
        "seconds": dt, "stage_seconds": {"select": s.t_select, "pairs": s.t_pairs, "quads": s.t_quads, "verify": s.t_verify},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline sample (0 = skip)")
    ap.add_argument("--no-time-to-register", dest="time_to_register", action="store_false", default=True)
    ap.add_argument("--points", type=int, default=N_POINTS)
    ap.add_argument("--sample", type=int, default=SAMPLE)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback of the product path)")
    # S4P_BENCH_ONE_GPU=1 (testing only): all ranks share GPU 0 and the 8-byte collective goes over gloo, to exercise
    # the N>1 code path on a single-GPU box.  The driver's multi-GPU runs use one GPU per rank and RCCL.
    one_gpu = os.environ.get("S4P_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if one_gpu:
            dist.init_process_group(backend="gloo")
            dev = torch.device("cpu")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)     # "nccl" is RCCL on ROCm

    from super4pcs_amd import build as B
    if rank == 0 and B.needs_build():
        B.build()
    if dist is not None:
        dist.barrier()
    from super4pcs_amd import capi, datasets, sharding

    P, Q, _ = datasets.bumpy_pair(args.points, overlap=OVERLAP, delta=DELTA, seed=SEED)
    opt = capi.make_options(DELTA, OVERLAP, args.sample)
    m = capi.Matcher(opt, device=local_rank, max_pairs=8 << 20, max_quads=64 << 20)
    m.init_full(P, Q)                       # sampling, grid build, upload: outside the timed region
    info = m.info()
    n_q, n_p = info.n_sampled_q, info.n_sampled_p
    sh = sharding.ShardedRansac(m, rank, world, dist, dev)

    sh.run_windows(args.warmup)
    m.profile_enable(True, False)
    m.profile_get(reset=True)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    sync()
    t0 = time.perf_counter()
    cand = sh.run_windows(args.steps)          # pipelined: host base selection of step t+1 overlaps the GPU pass of step t
    sync()
    dt = time.perf_counter() - t0
    prof = m.profile_get(reset=True)

    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    tc = torch.tensor([cand], dtype=torch.int64, device=dev)
    if dist is not None:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(tc, op=dist.ReduceOp.SUM)
    dt_max, cand_all = float(tt.item()), int(tc.item())

    # k-bar (mean P points distance-tested per query) from two extra, untimed, instrumented bases
    m.profile_enable(False, True)
    m.profile_get(reset=True)
    q_before = m.info().candidates_verified
    sh.run_windows(2)
    pk = m.profile_get(reset=True)
    q_after = m.info().candidates_verified
    queries = max((q_after - q_before) * n_q, 1)
    kbar = pk.verify_point_tests / queries
    m.profile_enable(False, False)

    # time-to-register (the metric's second half): one whole ComputeTransformation on the same pair, wall time from
    # call to return with inputs in host memory (sampling of both 1 M-point clouds, grid build, upload, all trials,
    # final apply).  Reported, never part of `value`.
    ttr = None
    if world == 1 and args.time_to_register:
        m2 = capi.Matcher(opt, device=local_rank, max_pairs=8 << 20, max_quads=64 << 20)
        m2.set_sharding(0, 1, True)
        t_reg = time.perf_counter()
        lcp2, M2, _ = m2.compute_transformation(P, Q)
        t_reg = time.perf_counter() - t_reg
        i2 = m2.info()
        ttr = {"seconds": t_reg, "lcp": float(lcp2), "trials_run": int(i2.bases_tried), "candidates_verified": int(i2.candidates_verified)}
        del m2

    traffic, traffic_note, limiter = None, "no PMC summary found", None
    pmc_file = os.path.join(ROOT, "profiles", "r01_pmc_k_verify.json")
    if os.path.exists(pmc_file):
        try:
            pj = json.load(open(pmc_file))
            traffic = pj["hbm_bytes_per_launch"]
            traffic_note = pj["note"]
            ta = pj.get("ta") or {}
            if ta.get("TA_BUSY_avr") and ta.get("GRBM_GUI_ACTIVE"):
                # what actually binds k_verify: the gather-address path, not HBM
                limiter = {"unit": "TA (texture addresser: divergent 8/16-byte gathers)",
                           "busy_frac": ta["TA_BUSY_avr"] / (ta["GRBM_GUI_ACTIVE"] / 8.0),
                           "wavefront_gathers_per_launch": ta.get("TA_FLAT_READ_WAVEFRONTS_sum"),
                           "source": "profiles/r01_pmc_k_verify.json (rocprofv3 --pmc, S4P_LANES=1; GRBM_GUI_ACTIVE is summed over the 8 XCDs)"}
        except Exception:
            pass

    if rank == 0:
        bc = bytes_per_candidate(n_q, kbar)
        launches = max(prof.verify_launches, 1)
        avg_ms = prof.verify_ms_total / launches
        cand_per_launch = prof.verify_candidates / launches
        achieved = (cand_per_launch * bc) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        out = {
            "metric": "candidate transforms verified/sec", "value": cand_all / dt_max, "unit": "candidates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[2]: synthetic %d-point pair, 50%% overlap, Gaussian noise sigma=delta=%g, "
                                   "sample_size=%d (n_P=%d sampled P points, n_Q=%d); one step = one RANSAC base per GPU"
                                   % (args.points, DELTA, args.sample, n_p, n_q),
                       "n_P": n_p, "n_Q": n_q, "delta": DELTA, "overlap": OVERLAP, "seed": SEED,
                       "candidates_timed": cand_all, "point_queries_per_s": cand_all * n_q / dt_max,
                       "parallelism": "bases sharded over %d GPU(s), one allreduce(max) per window" % world,
                       "time_to_register": ttr},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note, "limiter": limiter,
                         "kernel": "k_verify", "avg_launch_ms": avg_ms, "launches": int(prof.verify_launches),
                         "candidates_per_launch": cand_per_launch, "algorithmic_bytes_per_candidate": bc, "kbar": kbar,
                         "filter_pass_fraction": {"coarse_bitmap": pk.verify_l0_pass / queries, "reach_bit": pk.verify_l1_pass / queries,
                                                  "subcell_mask": pk.verify_l2_pass / queries},
                         "note": "algorithmic bytes (SURVEY.md 8d, no cache credit, c=27 cells) / HIP-event launch time; "
                                 "the bitmap early-out means most of these bytes are never fetched"},
            "stage_ms_per_step": {"pairs": prof.pairs_ms_total / max(prof.quads_launches, 1),
                                  "quads": prof.quads_ms_total / max(prof.quads_launches, 1), "verify": avg_ms},
        }
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(P, Q, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
