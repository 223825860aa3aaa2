#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X-native Super4PCS hot path.

Contract (driver): python bench.py --gpus N --steps K --warmup W ; for N>1 launched by
torch.distributed.run, one rank per GPU.  Rank 0 prints ONE JSON line.

Metric (BASELINE.json): candidate transforms verified / s on the synthetic 1 M-point pair
(configs[2]: 50 % overlap, Gaussian noise sigma = delta = 0.004, sample size 2000; --sample 20000 gives the
"GPU-scale" sample of SURVEY.md 8d as a second line).
One "step" = one RANSAC base through the whole hot path on this rank's GPU:
   SelectQuadrilateral (host) -> ExtractPairs x2 -> FindCongruentQuadrilaterals ->
   ComputeRigidTransformation + Verify of every congruent candidate -> best selection,
i.e. Match4PCSBase::TryOneBase (match4pcsBase.hpp:281-360).  A candidate counts when it passed the
rms gate and entered Verify (reference counter nbCongruentAto, :441); like the reference's Verify (match4pcsBase.cc:558-560)
the device abandons a candidate once it can no longer exceed the best LCP of the registration so far -- the rate with
every candidate scored over all sampled Q points is reported next to it (config.full_count_mode).
Inputs (sampled clouds, LCP grid) are resident in HBM before the timed region.
With N GPUs each rank owns every N-th base of the same sequence (weak scaling: K device steps per
rank) and one 8-byte all-reduce(MAX) per window over RCCL picks the winner.

The timed region (W warm-up steps, then exactly K steps between barrier + device synchronisation) is repeated
`--repeats` times on a fresh matcher with the same seed, i.e. over the SAME bases, so that the spread is timing noise and
not workload variation; `value` / `ms_per_step` are the median repeat, `spread` holds min / median / max.

Parity gate, non-zero exit on any mismatch.  The oracle (oracle/, CPU, OpenMP over candidates) runs the SAME
W + K bases in the reference's mode (kd-tree Verify with early exit) and
  * every repeat of the timed region must end in the oracle's state: best LCP, winning base + quad, 4x4, and must have
    verified exactly the oracle's number of candidates over the K timed bases (parity.bases == steps);
  * a replay of the same bases one by one on a fresh GPU matcher is compared base by base: pair / quad / candidate counts,
    TryOneBase's return value, the running best;
  * for the first `--parity-full-bases` bases the ordered quad list and the inlier count of EVERY candidate are compared with
    the oracle in full-count mode.
The oracle is only ever the checker and the cpu_baseline, never the thing timed.

Roofline: the byte model's inputs (fractions of the queries that pass the three levels of the LCP
structure, exact point tests per query) are measured by an instrumented replay of the TIMED bases; `frac` is the per-step
figure (algorithmic bytes of the K timed bases / timed seconds / HBM peak), which does not depend on how many bases are in
flight; rocprofv3 --pmc passes over an inner run of the same W + K bases give the HBM-side traffic, the L2 hit rate and the
VALU issue utilisation of k_verify, and `binding` names the resource that is closest to its roof.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# (round 6: the 1200-line file is split by subject -- benchlib/: workload + byte models, parity gates, CPU baselines, rocprofv3
# passes, the GPU-scale sample; this file keeps the command line, the timed region and the JSON line.  The names other code uses
# from here -- bench.DELTA, bench.seg_len32, ... -- are re-exported.)
from benchlib.workload import (DELTA, GOLDEN_SCALE, HBM_PEAK_GBS, L2_PEAK_GBS, MAX_PAIRS, MAX_QUADS, N_POINTS, N_SIMDS, OVERLAP, SAMPLE, SEED,      # noqa: E402,F401
                               seg_len32, structure_bytes_per_candidate, survey_bytes_per_candidate)
from benchlib.parity import parity_gate, parity_gate_scale                                                  # noqa: E402
from benchlib.baselines import cpu_baseline                                                                # noqa: E402
from benchlib.profiling import hbm_bound_point, hbm_point_inner, pmc_passes                                # noqa: E402
from benchlib.scale import extra_sample_finish, extra_sample_start, scale_golden_inner                     # noqa: E402


_PHASES = []


def _lap(name, _t=[None]):
    """Wall time of the phases of a run (config.phase_seconds): where the minutes of the default command go."""
    now = time.perf_counter()
    if _t[0] is not None:
        _PHASES.append((name, round(now - _t[0], 2)))
    _t[0] = now


def main():
    _lap("start")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=5, help="timed regions (same bases, fresh matcher): value = median")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the 1-core cpu_baseline sample (0 = skip)")
    ap.add_argument("--no-time-to-register", dest="time_to_register", action="store_false", default=True)
    ap.add_argument("--no-parity", dest="parity", action="store_false", default=True)
    ap.add_argument("--parity-bases", type=int, default=-1, help="timed bases replayed on the oracle (default: all of them up to 40)")
    ap.add_argument("--parity-full-bases", type=int, default=2, help="bases whose EVERY candidate count is checked")
    ap.add_argument("--no-pmc", dest="pmc", action="store_false", default=True, help="skip the rocprofv3 --pmc passes (traffic / valu / l2 = null)")
    ap.add_argument("--no-hbm-point", dest="hbm_point", action="store_false", default=True)
    ap.add_argument("--no-full-count-mode", dest="full_count_mode", action="store_false", default=True,
                    help="skip the extra timed region with the early exit off (config.full_count_mode)")
    ap.add_argument("--no-instrumented", dest="instrumented", action="store_false", default=True,
                    help="skip the instrumented replays of the timed bases (byte model = null): for a clean rocprofv3 kernel trace of the command")
    ap.add_argument("--no-stage-pass", dest="stage_pass", action="store_false", default=True,
                    help="skip the extra pass with events around every stage (stage_ms_per_step = null): for a clean rocprofv3 kernel trace")
    ap.add_argument("--no-exclusive", dest="exclusive", action="store_false", default=True,
                    help="skip the one-base-in-flight re-run (roofline.per_launch.exclusive)")
    ap.add_argument("--profile-dir", default=None, help="keep the k_verify rows of the rocprofv3 outputs here (e.g. profiles/r03_bench)")
    ap.add_argument("--shard-mode", choices=["auto", "base", "split"], default="auto",
                    help="N > 1: trials sharded by base (one all-reduce per window), or every base split over all GPUs (SURVEY 8e "
                         "level 2; auto = split at the GPU-scale sample, where a base takes seconds)")
    ap.add_argument("--inner", action="store_true", help="(used by the --pmc passes) timed region only, no JSON")
    ap.add_argument("--hbm-point-inner", action="store_true", help="(used by the HBM-bound point) one cold scoring launch")
    ap.add_argument("--scale-golden-inner", action="store_true", help="(used by the `extra` object) the golden-checked bases at --sample, one JSON object")
    ap.add_argument("--hbm-transforms", type=int, default=4096)
    ap.add_argument("--points", type=int, default=N_POINTS)
    ap.add_argument("--sample", type=int, default=SAMPLE)
    ap.add_argument("--ttr-configs", default="", help="e.g. 1,3,4: also run tools/init_timing.py for these BASELINE configs (init + whole-registration "
                    "time at SURVEY 8d's sample sizes; minutes) and put the lines into extra.time_to_register_configs")
    ap.add_argument("--no-extra", dest="extra", action="store_false", default=True,
                    help="skip the `extra` object: the same clouds at SURVEY 8d's GPU-scale sample (n = 20 000), two bases, in a process of its own")
    args = ap.parse_args()

    scale_mode = args.sample > 5000          # the "GPU-scale" sample: bases of ~10^9 quads, seconds per base
    if scale_mode:
        # a whole registration, the CPU baselines and the per-launch profiling passes are out of reach at this size (the
        # reference itself cannot finish ONE base); what remains: value, parity through counts + checksums, the byte model
        args.time_to_register = False; args.pmc = False; args.hbm_point = False; args.exclusive = False; args.cpu_seconds = 0.0
        args.repeats = min(args.repeats, 2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # the library's lanes are HIP streams; the runtime reads this when it initialises (the library itself never
    # touches the environment; opt out with S4P_KEEP_HW_QUEUES=1): s4p_capi.hip, DESIGN.md 5.4
    if os.environ.get("S4P_KEEP_HW_QUEUES") != "1":
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
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

    if args.hbm_point_inner:
        hbm_point_inner(local_rank, args.hbm_transforms)
        return
    if args.scale_golden_inner:
        raise SystemExit(scale_golden_inner(args, local_rank))

    P, Q, T_gt = datasets.bumpy_pair(args.points, overlap=OVERLAP, delta=DELTA, seed=SEED)
    opt = capi.make_options(DELTA, OVERLAP, args.sample)
    collective = {"kind": "none (one GPU: the engine's pipelined Perform_N_steps)"}

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def make_driver(m):
        if world > 1:
            # the C++ sharded loop behind the C ABI (s4p_shard_*): RCCL all-reduce(max) of one 8-byte key per window
            sh = capi.Shard(m, rank, world, 2)                  # helper threads where they pay (s4p_matcher_set_sharding)
            split = args.shard_mode == "split" or (args.shard_mode == "auto" and scale_mode)
            if split:
                sh.set_mode(True)
            collective["mode"] = "every base split over the GPUs (2 all-reduces per base)" if split else "trials sharded by base (1 all-reduce per window)"
            collective["split"] = split
            if one_gpu:
                sh.use_collective(capi.torch_collective(dist))          # single-GPU dry run: gloo through the callback provider
                collective["kind"] = "gloo through the callback provider (S4P_BENCH_ONE_GPU dry run)"
            else:
                # the library's own communicator (ncclCommInitRank from the id rank 0 made); should RCCL not bind or not
                # initialise on some rank, every rank falls back to the process group torch already has (same 8-byte
                # all-reduce per window, through the callback provider) -- and the JSON line says which one ran
                ok = 1
                try:
                    idt = torch.zeros(128, dtype=torch.uint8, device=dev)
                    if rank == 0:
                        idt.copy_(torch.frombuffer(bytearray(capi.rccl_unique_id()), dtype=torch.uint8))
                except Exception as e:                          # noqa: BLE001
                    print("rank %d: ncclGetUniqueId through the library failed (%s)" % (rank, e), file=sys.stderr)
                    ok = 0
                flag = torch.tensor([ok], dtype=torch.int32, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()):
                    dist.broadcast(idt, src=0)
                    try:
                        sh.use_rccl(local_rank, idt.cpu().numpy().tobytes())
                    except Exception as e:                      # noqa: BLE001
                        print("rank %d: s4p_shard_use_rccl failed (%s)" % (rank, e), file=sys.stderr)
                        ok = 0
                    flag = torch.tensor([ok], dtype=torch.int32, device=dev)
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()):
                    collective["kind"] = "rccl: ncclAllReduce(uint64, max) per window from the C++ loop (s4p_shard_run_windows, own communicator)"
                    collective["rccl_reported"] = sh.comm_info()          # (ncclCommCount, ncclCommUserRank) of that communicator
                else:
                    sh.use_collective(capi.torch_collective(dist, dev))
                    collective["kind"] = "torch-fallback: torch.distributed all_reduce(max) on the nccl (= RCCL) process group through the callback provider"
            if not collective.get("warmed") and not collective["kind"].startswith("rccl"):
                # a process group sets its connections up on first use of each collective (the library's own communicator
                # does that inside s4p_shard_use_rccl): once, before any timed region
                wt = torch.zeros(1, dtype=torch.int64, device=None if one_gpu else dev)
                dist.all_reduce(wt, op=dist.ReduceOp.MAX)
                wb = torch.zeros(256, dtype=torch.uint8, device=None if one_gpu else dev)
                for root in range(world):
                    dist.broadcast(wb, src=root)
                collective["warmed"] = True
            return sh
        return sharding.ShardedRansac(m, rank, world, dist, dev, producer_threads={"0": False, "1": True}.get(os.environ.get("S4P_BENCH_HELPERS", ""), "auto"))      # world 1: the engine's own pipelined Perform_N_steps

    per_rank = {}

    def timed_region(steps, warmup, early_exit=True, stage_events=False):
        """Fresh matcher, same seed: W untimed steps, then exactly K timed steps between barrier + synchronize.  HIP events
        bracket every k_verify launch of the timed steps (roofline.per_launch); the other stages' events (three more records per
        base, each a barrier in the lane's stream) only in the separate, unreported pass that fills stage_ms_per_step."""
        m = capi.Matcher(opt, device=local_rank, max_pairs=(32 << 20) if scale_mode else MAX_PAIRS, max_quads=(32 << 20) if scale_mode else MAX_QUADS)
        m.early_exit(early_exit)
        m.init_full(P, Q)                       # sampling, grid build, upload: outside the timed region
        sh = make_driver(m)
        sh.run_windows(warmup)
        m.profile_enable(1 if (stage_events or scale_mode) else 2, False)      # (a base of the 20 000-point sample takes seconds: events are free)
        m.profile_get(reset=True)
        sync()
        t0 = time.perf_counter()
        cand = sh.run_windows(steps)            # pipelined: host base selection of step t+1 overlaps the GPU pass of step t
        sync()
        dt = time.perf_counter() - t0
        prof = m.profile_get(reset=True)
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        tc = torch.tensor([cand], dtype=torch.int64, device=dev)
        per_rank["ms_per_step"], per_rank["candidates"] = [dt / max(steps, 1) * 1e3], [int(cand)]
        if dist is not None:
            # what every rank measured on its own clock / counted on its own GPU (the line's ms_per_step is the MAX, value the SUM)
            gt = [torch.zeros_like(tt) for _ in range(world)]
            gc = [torch.zeros_like(tc) for _ in range(world)]
            dist.all_gather(gt, tt)
            dist.all_gather(gc, tc)
            per_rank["ms_per_step"] = [float(x.item()) / max(steps, 1) * 1e3 for x in gt]
            per_rank["candidates"] = [int(x.item()) for x in gc]
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dist.all_reduce(tc, op=dist.ReduceOp.SUM)
        return m, sh, float(tt.item()), int(tc.item()), prof

    if args.inner:                              # profiled inner run of the --pmc passes: the timed region and nothing else
        timed_region(args.steps, args.warmup)
        return

    _lap("import torch + library + synthetic clouds")
    runs, finals = [], []
    m = sh = None
    for _ in range(max(args.repeats, 1)):
        if m is not None:
            if hasattr(sh, "close"):
                sh.close()
            m.close()
        m, sh, dt_max, cand_all, prof = timed_region(args.steps, args.warmup)
        runs.append((cand_all / dt_max, dt_max, cand_all, prof, dict(per_rank)))
        finals.append(m.info())
    order = sorted(range(len(runs)), key=lambda k: runs[k][0])
    med = order[len(order) // 2]
    value, dt_max, cand_all, prof, per_rank_med = runs[med]
    info = finals[-1]
    n_q, n_p = info.n_sampled_q, info.n_sampled_p
    final_info = info                           # state after the last repeat's windows (N > 1: compared across ranks below)
    chunk_stats = m.chunk_stats()
    lane_growths = m.capacity_growths()
    # which code produced this line: commit + source digest stamped at build time, the k_verify instantiation the loop launches
    import hashlib
    from super4pcs_amd import build as _build
    provenance = dict(_build.build_info(), k_verify=m.verify_kernel_info(),
                      bench_py_sha16=hashlib.sha256(open(os.path.abspath(__file__), "rb").read()).hexdigest()[:16],
                      library=os.environ.get("S4P_LIB", "super4pcs_amd/lib/libsuper4pcs_amd.so"),
                      env={k: v for k, v in os.environ.items() if k.startswith("S4P_")})
    if hasattr(sh, "close"):
        sh.close()
    m.close()
    _lap("timed repeats")
    # per-stage HIP-event times: one more pass over the same bases with events around every stage (not part of `value`)
    stage_prof = prof if scale_mode else None
    if not scale_mode and args.stage_pass:
        ms_, shs_, _dt, _c, stage_prof = timed_region(args.steps, args.warmup, stage_events=True)
        if hasattr(shs_, "close"):
            shs_.close()
        ms_.close()
    # the same timed region with every candidate counted in full (no early exit): what rounds 1 and 2 reported as `value`
    full_mode = None
    if world == 1 and args.full_count_mode:
        mf, shf, dt_f, cand_f, prof_f = timed_region(args.steps, args.warmup, early_exit=False)
        fi = mf.info()
        full_mode = {"value": cand_f / dt_f, "ms_per_step": dt_f / args.steps * 1e3, "candidates": cand_f,
                     "same_result_as_default": bool(fi.best_lcp == info.best_lcp and list(fi.transform) == list(info.transform)
                                                    and list(fi.base) == list(info.base) and list(fi.congruent) == list(info.congruent) and cand_f == cand_all),
                     "note": "S4P early exit off: every candidate scored over all n_Q points; one repeat"}
        mf.close()

    # the byte model's inputs, measured on the TIMED bases: a fresh matcher runs the W warm-up bases, then the K timed bases
    # again through the instrumented kernel (slower; untimed) -- in the default mode (early exit: the structure walk the timed
    # kernel really does) and with every candidate counted in full
    # (N > 1: rank 0 replays the trials of ALL ranks' timed windows one after the other -- the job's timed bases)
    trials_per_window = 1 if (world == 1 or collective.get("split")) else world

    def instrumented(early_exit):
        mi = capi.Matcher(opt, device=local_rank, max_pairs=(32 << 20) if scale_mode else MAX_PAIRS, max_quads=(32 << 20) if scale_mode else MAX_QUADS)
        mi.early_exit(early_exit)
        mi.init_full(P, Q)
        shi = sharding.ShardedRansac(mi, 0, 1, None, torch.device("cuda", local_rank), producer_threads="auto")
        shi.run_windows(args.warmup * trials_per_window)
        mi.profile_enable(False, True)
        mi.profile_get(reset=True)
        q_before = mi.info().candidates_verified
        shi.run_windows(args.steps * trials_per_window)
        pk = mi.profile_get(reset=True)
        queries = max((mi.info().candidates_verified - q_before) * n_q, 1)
        mi.close()
        kb = pk.verify_point_tests / queries
        fr = (pk.verify_l0_pass / queries, pk.verify_l1_pass / queries, pk.verify_l2_pass / queries)
        return kb, fr, kb / 4.0 + 0.375 * fr[2]     # listed points / 4, plus the part-filled last group of a list (mean 3/8 of a group)

    f_l0 = f_l1 = f_l2 = kbar = 0.0
    groups_per_query = 0.0
    full_walk = None
    if world > 1 and rank == 0 and not scale_mode and args.instrumented:
        kbar, (f_l0, f_l1, f_l2), groups_per_query = instrumented(True)
    if world == 1 and not scale_mode and args.instrumented:
        kbar, (f_l0, f_l1, f_l2), groups_per_query = instrumented(True)
        kb_f, fr_f, gq_f = instrumented(False)
        full_walk = {"kbar": kb_f, "pass_fractions": {"coarse_bitmap_L0": fr_f[0], "reach_bit_L1": fr_f[1], "subcell_mask_L2": fr_f[2]},
                     "groups_per_query": gq_f, "bytes_per_candidate": structure_bytes_per_candidate(n_q, fr_f[0], fr_f[1], fr_f[2], gq_f)[1]}

    _lap("stage pass + full-count region + instrumented replays")
    # time-to-register (the metric's second half): one whole ComputeTransformation on the same pair, wall time from
    # call to return with inputs in host memory (sampling of both 1 M-point clouds, grid build, upload, all trials,
    # final apply).  Reported, never part of `value`.
    ttr = None
    T2c = None
    if world == 1 and args.time_to_register:
        m2 = capi.Matcher(opt, device=local_rank, max_pairs=MAX_PAIRS, max_quads=MAX_QUADS)
        t_reg = time.perf_counter()
        lcp2, M2, _ = m2.compute_transformation(P, Q)
        t_reg = time.perf_counter() - t_reg
        i2 = m2.info()
        ttr = {"seconds": t_reg, "lcp": float(lcp2), "best_count": int(i2.best_count), "trials_run": int(i2.bases_tried),
               "candidates_verified": int(i2.candidates_verified),
               "rotation_error_vs_ground_truth": float(np.max(np.abs(M2[:3, :3] - T_gt[:3, :3]))),
               "translation_error_vs_ground_truth": float(np.max(np.abs(M2[:3, 3] - T_gt[:3, 3])))}
        T2c = np.array(i2.transform, np.float32).reshape(4, 4)
        m2.close()

    _lap("time-to-register")
    pmc, pmc_note, pmc_kernels = {}, ["skipped"], {}
    if rank == 0 and world == 1 and args.pmc:
        pmc, pmc_note, pmc_kernels = pmc_passes(args, int(prof.verify_launches))
        _lap("rocprofv3 kernel trace + counter passes")
    hbm_point = None
    if rank == 0 and world == 1 and args.hbm_point:
        hbm_point = hbm_bound_point(args, local_rank)

    apply_row = None
    if rank == 0 and world == 1 and args.hbm_point:
        # final apply (match4pcsBase.hpp:265-267) on device-resident points: the product's VALU kernel against its MFMA
        # formulation, GB/s = 24 B per point (12 in, 12 out) / HIP-event time; DESIGN.md section 5.2
        try:
            actx = capi.Context(opt, device=local_rank, max_pairs=1 << 16, max_quads=1 << 16)
            apply_row = {}
            for n in (1_000_000, 10_000_000):
                ms_valu, ms_mfma, mism, maxabs = actx.apply_bench(n, 20)
                apply_row["n=%d" % n] = {"valu_ms": ms_valu, "valu_GBps": 24.0 * n / (ms_valu * 1e-3) / 1e9,
                                         "mfma_ms": ms_mfma, "mfma_GBps": 24.0 * n / (ms_mfma * 1e-3) / 1e9,
                                         "coordinates_differing_from_valu": mism, "of": 3 * n, "max_abs_difference": maxabs}
            actx.close()
        except Exception as e:                                  # noqa: BLE001
            apply_row = {"error": "%s: %s" % (type(e).__name__, e)}

    # k_verify with the chip to itself: the same bases with ONE base in flight (S4P_LANES is read at context creation).
    # With the default number of lanes every launch shares the CUs with the launches of the other lanes, so its HIP-event
    # duration is not the kernel's own time.
    _lap("HBM-bound point + final-apply rows")
    exclusive = None
    if world == 1 and rank == 0 and args.exclusive:
        saved = os.environ.get("S4P_LANES")
        os.environ["S4P_LANES"] = "1"
        try:
            m1 = capi.Matcher(opt, device=local_rank, max_pairs=MAX_PAIRS, max_quads=MAX_QUADS)
            m1.init_full(P, Q)
            m1.perform_n_steps(args.warmup)
            m1.profile_enable(True, False)
            m1.profile_get(reset=True)
            m1.perform_n_steps(args.steps)
            p1 = m1.profile_get(reset=True)
            m1.close()
            if p1.verify_launches:
                exclusive = {"avg_launch_ms": p1.verify_ms_total / p1.verify_launches, "launches": int(p1.verify_launches),
                             "candidates_per_launch": p1.verify_candidates / p1.verify_launches}
        finally:
            if saved is None:
                os.environ.pop("S4P_LANES", None)
            else:
                os.environ["S4P_LANES"] = saved

    _lap("one-base-in-flight rerun")
    # The `extra` figure (the 20 000-point sample, two bases: ~50 s of GPU time in a process of its own) starts HERE, once every
    # measurement that wants the GPU to itself is over, and runs beside the host-bound legs that follow (the oracle's replay of the
    # timed bases in the parity gate, the CPU baselines); it is collected when the line is assembled.
    extra_proc = extra_sample_start(args) if (rank == 0 and world == 1 and args.extra and not scale_mode and not args.inner) else None
    parity = None
    if rank == 0 and world == 1 and args.parity:
        n_par = args.parity_bases if args.parity_bases >= 0 else min(args.steps, 40)
        n_par = min(n_par, args.steps)
        if scale_mode:
            parity, ostate, om_full = parity_gate_scale(P, Q, opt, args.warmup, n_par, local_rank, args.sample)
        else:
            parity, ostate, om_full = parity_gate(P, Q, opt, args.warmup, n_par, args.parity_full_bases, local_rank, args.sample)
        failed = parity.setdefault("failed", [])
        # every repeat of the timed region ended in the same state and verified the same number of candidates ...
        for k, fi in enumerate(finals):
            same = (fi.best_lcp == finals[0].best_lcp and list(fi.transform) == list(finals[0].transform) and list(fi.base) == list(finals[0].base)
                    and list(fi.congruent) == list(finals[0].congruent) and runs[k][2] == runs[0][2])
            if not same:
                failed.append("timed repeat %d ends in a different state than repeat 0" % k)
        # ... which is the oracle's after the same W + K bases (when all K were replayed)
        parity["timed_repeats_checked_against_oracle"] = 0
        if n_par == args.steps and scale_mode:
            for k in range(len(finals)):
                if runs[k][2] != ostate["candidates_timed"]:
                    failed.append("timed repeat %d: %d candidates vs the oracle's %d over the same bases" % (k, runs[k][2], ostate["candidates_timed"]))
                parity["timed_repeats_checked_against_oracle"] += 1
        elif n_par == args.steps:
            for k, fi in enumerate(finals):
                ok = (fi.best_lcp == ostate["best_lcp"] and list(fi.base) == ostate["base"] and list(fi.congruent) == ostate["congruent"]
                      and np.array_equal(np.array(fi.transform, np.float32).reshape(4, 4), ostate["transform"]) and runs[k][2] == ostate["candidates_timed"])
                if not ok:
                    failed.append("timed repeat %d: final state / candidate total differs from the oracle's after the same bases "
                                  "(LCP %r vs %r, candidates %d vs %d)" % (k, fi.best_lcp, ostate["best_lcp"], runs[k][2], ostate["candidates_timed"]))
                parity["timed_repeats_checked_against_oracle"] += 1
        if scale_mode and ostate.get("byte_model", {}).get("queries"):
            bmq = ostate["byte_model"]                # (instrumenting 10^8 candidates per base is out of reach: a subsample of them instead)
            kbar = bmq["tests"] / bmq["queries"]
            f_l0, f_l1, f_l2 = bmq["l0"] / bmq["queries"], bmq["l1"] / bmq["queries"], bmq["l2"] / bmq["queries"]
            groups_per_query = kbar / 4.0 + 0.375 * f_l2
        if ttr is not None:
            # the registration's result, recounted by the oracle's kd-tree Verify on its own sampled clouds
            recount = int(om_full.verify_batch(T2c.reshape(1, 16))[0])
            ttr["oracle_recount_of_final_transform"] = recount
            if recount != ttr["best_count"]:
                failed.append("time-to-register: final LCP %d != oracle recount %d" % (ttr["best_count"], recount))
        parity["mismatches"] = len(failed) if not parity.get("mismatches") else max(parity["mismatches"], len(failed))
        if not failed:
            parity.pop("failed", None)

    # N > 1: the sharded loop must leave every rank with the state the sequential loop reaches after the same trials.
    # All ranks' states are compared with each other and with a sequential single-GPU replay on rank 0 (which is the
    # path the N = 1 parity gate checks against the oracle).
    if world > 1 and args.parity:
        gi = final_info
        mine = np.concatenate([[gi.current_trial, gi.best_count], np.frombuffer(np.float32(gi.best_lcp).tobytes(), np.uint32),
                               np.frombuffer(np.array(gi.transform, np.float32).tobytes(), np.uint32),
                               np.array(gi.base, np.int64), np.array(gi.congruent, np.int64)]).astype(np.int64)
        mt = torch.tensor(mine, dtype=torch.int64, device=dev)
        allt = [torch.zeros_like(mt) for _ in range(world)]
        dist.all_gather(allt, mt)
        if rank == 0:
            states = [t.cpu().numpy() for t in allt]
            failed = ["rank %d ends in a different state than rank 0" % r for r in range(1, world) if not np.array_equal(states[r], states[0])]
            trials = (args.warmup + args.steps) * world           # the last repeat's windows
            seq = capi.Matcher(opt, device=local_rank, max_pairs=MAX_PAIRS, max_quads=MAX_QUADS)
            seq.init_full(P, Q)
            for _ in range(trials):                               # TryOneBase, one after the other: no stop rule, like run_windows
                seq.try_one_base()
            si = seq.info()
            same = (si.best_count == gi.best_count and si.best_lcp == gi.best_lcp
                    and list(si.transform) == list(gi.transform) and list(si.base) == list(gi.base) and list(si.congruent) == list(gi.congruent))
            if not same:
                failed.append("sharded state != sequential single-GPU replay of %d trials (LCP %r vs %r)" % (trials, gi.best_lcp, si.best_lcp))
            seq.close()
            parity = {"what": "N > 1: every rank's final state (trial count, best LCP, 4x4, winning base and quad) equal across ranks and "
                              "equal to a sequential single-GPU replay of the same %d trials on rank 0; the sequential path is the one the "
                              "N = 1 gate checks against the oracle" % trials,
                      "trials": trials, "ranks": world, "mismatches": len(failed)}
            if failed:
                parity["failed"] = failed

    _lap("parity gate (oracle replay of the warm-up + timed bases; the extra process runs beside it)")
    if rank == 0:
        launches = max(prof.verify_launches, 1)
        avg_ms = prof.verify_ms_total / launches
        cand_per_launch = prof.verify_candidates / launches
        sweep_b, gather_b = structure_bytes_per_candidate(n_q, f_l0, f_l1, f_l2, groups_per_query)
        if not (f_l0 > 0.0):                                    # byte model not measured (--no-instrumented, a lab flag): no roofline figure
            gather_b = float("nan")
        achieved = value * gather_b / 1e9                       # per-step figure: bytes of the K timed bases / timed seconds
        survey_b = survey_bytes_per_candidate(n_q, kbar)
        vals = sorted(r[0] for r in runs)

        def mean(name):
            return pmc[name][0] if name in pmc else None

        traffic = None
        if mean("FETCH_SIZE") is not None and mean("WRITE_SIZE") is not None:
            # FETCH_SIZE / WRITE_SIZE are reported in KB; gfx950 tallies 128-B read requests at 64 B, hence the factor 2 on
            # FETCH_SIZE (MI355X_MICROARCH.md, HBM); counts L2 -> fabric requests including Infinity-Cache hits
            traffic = mean("FETCH_SIZE") * 1024.0 * 2.0 + mean("WRITE_SIZE") * 1024.0
        valu = None
        if mean("SQ_ACTIVE_INST_VALU") is not None and mean("GRBM_GUI_ACTIVE"):
            avail = N_SIMDS * mean("GRBM_GUI_ACTIVE") / 8.0 / 4.0     # SIMD quad-cycles of one launch (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
            valu = {"frac": mean("SQ_ACTIVE_INST_VALU") / avail, "active_quad_cycles": mean("SQ_ACTIVE_INST_VALU"), "available_quad_cycles": avail,
                    "valu_instructions_per_candidate": (mean("SQ_INSTS_VALU") or 0) / max(cand_per_launch, 1),
                    "wait_any_frac_of_wave_cycles": (mean("SQ_WAIT_ANY") or 0) / max(mean("SQ_WAVE_CYCLES") or 1, 1),
                    "note": "SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 / 4) of k_verify, launches serialised by the counter collection "
                            "(the kernel alone), mean over the timed bases"}
        l2 = None
        if mean("TCC_HIT_sum") is not None and mean("TCC_MISS_sum") is not None:
            req = mean("TCC_HIT_sum") + mean("TCC_MISS_sum")
            # the launches of the counter passes cover a group of bases like the timed ones and are serialised by the collection: their
            # OWN duration (kernel trace of the same pass) is the time these requests were made in
            kv_us = (pmc_kernels.get("k_verify") or {}).get("avg_us")
            ex_ms = kv_us * 1e-3 if kv_us else (exclusive["avg_launch_ms"] if exclusive else avg_ms)
            l2 = {"hit_rate": mean("TCC_HIT_sum") / max(req, 1), "requests_per_launch": req,
                  "GBps_at_128B_per_request": req * 128.0 / (ex_ms * 1e-3) / 1e9, "peak_GBps": L2_PEAK_GBS,
                  "frac": req * 128.0 / (ex_ms * 1e-3) / 1e9 / L2_PEAK_GBS,
                  "launch_ms": ex_ms,
                  "note": "TCC_HIT_sum + TCC_MISS_sum per launch x 128 B (an upper bound: a 16-B gather moves at most one line) over the kernel's "
                          "own launch time in the counter pass (launches serialised, a group of bases per launch)"}
        # all four kernels of a device pass: own duration under the counter passes (launches serialised), issue /
        # wait shares and L2 hit rate from the same passes; a launch covers a GROUP of bases
        kernels_tbl = {}
        for kn, cs in (pmc_kernels or {}).items():
            if not cs:
                continue
            row = {"launches": cs.get("launches"), "avg_us_per_launch": cs.get("avg_us")}
            if cs.get("GRBM_GUI_ACTIVE") and cs.get("SQ_ACTIVE_INST_VALU") is not None:
                row["valu_frac"] = cs["SQ_ACTIVE_INST_VALU"] / (N_SIMDS * cs["GRBM_GUI_ACTIVE"] / 8.0 / 4.0)
            if cs.get("SQ_WAVE_CYCLES") and cs.get("SQ_WAIT_ANY") is not None:
                row["wait_any_frac_of_wave_cycles"] = cs["SQ_WAIT_ANY"] / cs["SQ_WAVE_CYCLES"]
            if cs.get("TCC_HIT_sum") is not None and cs.get("TCC_MISS_sum") is not None:
                rq = cs["TCC_HIT_sum"] + cs["TCC_MISS_sum"]
                row["l2_hit_rate"] = cs["TCC_HIT_sum"] / max(rq, 1.0)
                if cs.get("avg_us"):
                    row["l2_frac_at_128B_per_request"] = rq * 128.0 / (cs["avg_us"] * 1e-6) / 1e9 / L2_PEAK_GBS
            if cs.get("FETCH_SIZE") is not None and cs.get("avg_us"):
                row["fetch_GBps_uncorrected"] = cs["FETCH_SIZE"] * 1024.0 / (cs["avg_us"] * 1e-6) / 1e9      # FETCH_SIZE in KB; NOT doubled (gathers: the gfx950 factor is calibrated for wide streams only)
            fr = {k: row[k2] for k, k2 in (("valu", "valu_frac"), ("l2", "l2_frac_at_128B_per_request")) if row.get(k2) is not None}
            row["binding"] = ("latency (nothing above 0.5: waiting %.2f of the wave-cycles)" % row.get("wait_any_frac_of_wave_cycles", float("nan"))) if (not fr or max(fr.values()) < 0.5) else max(fr, key=lambda k: fr[k])
            kernels_tbl[kn] = row
        fracs = {"hbm": achieved / HBM_PEAK_GBS} if achieved == achieved else {}
        if valu:
            fracs["valu"] = valu["frac"]
        if l2:
            fracs["l2"] = l2["frac"]
        binding = max(fracs, key=lambda k: fracs[k]) if fracs else None
        out = {
            "metric": "candidate transforms verified/sec", "value": value, "unit": "candidates/s",
            # the same timed region with every candidate counted in full (no early-exit bound): quoted beside `value` everywhere
            "value_full_count": None if full_mode is None else full_mode["value"],
            "steps_note": "the timed region of %d steps is %.2f ms: pipeline fill is in it (the lanes start empty after the warm-up bases have been "
                          "waited for); the default run (200 steps) is the steadier figure" % (args.steps, dt_max * 1e3),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if collective.get("split") else "weak",     # split mode: the K bases are shared by all GPUs; by base: K bases per GPU
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "spread": {"repeats": len(runs), "min": vals[0], "median": vals[len(vals) // 2], "max": vals[-1],
                       "note": "each repeat: fresh matcher, same seed, same %d timed bases" % args.steps},
            "config": {"workload": "configs[2]: synthetic %d-point pair, 50%% overlap, Gaussian noise sigma=delta=%g, "
                                   "sample_size=%d (n_P=%d sampled P points, n_Q=%d); one step = one RANSAC base per GPU"
                                   % (args.points, DELTA, args.sample, n_p, n_q),
                       "n_P": n_p, "n_Q": n_q, "delta": DELTA, "overlap": OVERLAP, "seed": SEED,
                       "candidates_timed": cand_all, "point_queries_per_s": cand_all * n_q / dt_max,
                       "parallelism": "bases sharded over %d GPU(s), one 8-byte all-reduce(max) per window" % world,
                       "collective": collective["kind"], "shard_mode": collective.get("mode"),
                       "ranks": {"n_ranks": world, "torch_distributed_world": (dist.get_world_size() if dist is not None else 1),
                                 "rccl_comm_count": (collective.get("rccl_reported") or (None, None))[0],      # what the library's own communicator says (None: no RCCL communicator in this run)
                                 "rccl_comm_user_rank": (collective.get("rccl_reported") or (None, None))[1],
                                 "per_rank_ms_per_step": per_rank_med.get("ms_per_step"), "per_rank_candidates": per_rank_med.get("candidates"),
                                 "note": "each rank's own wall clock over the K timed steps and its own candidate count in the median repeat; "
                                         "`ms_per_step` is the max over ranks, `value` the sum of the candidates / that time"},
                       "early_exit": {"on": True, "candidates_abandoned": int(prof.verify_pruned), "fraction": prof.verify_pruned / max(cand_all, 1),
                                      "exact_point_tests_per_query": kbar, "exact_point_tests_per_query_full_counts": None if full_walk is None else full_walk["kbar"],
                                      "note": "candidates that can no longer EXCEED the registration's best inlier count are abandoned (every candidate that "
                                              "does not become the new best ends that way, most of them before their exact stage), as the reference's Verify "
                                              "does (match4pcsBase.cc:520,558-560); they count as verified there and here; results identical"},
                       "full_count_mode": full_mode,
                       "chunked_bases": chunk_stats, "lane_growths": lane_growths,
                       "time_to_register": ttr},
            "parity": parity,
            "roofline": {
                "bound": "hbm", "kernel": "k_verify", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                "achieved": achieved, "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_note": "; ".join(pmc_note) if pmc_note else
                           "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over an inner run of the same %d + %d bases, mean per k_verify "
                           "launch of the timed bases; FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B), uncalibrated for 16-B gathers; "
                           "counts L2 -> fabric requests including Infinity-Cache hits" % (args.warmup, args.steps),
                "algorithmic_bytes_per_candidate": gather_b,
                "definition": "achieved = gather bytes the three-level structure requires per candidate (8 B reach word per L0 survivor whose candidate "
                              "outlives its LDS-only sweep, 32 B header per reach survivor that enters an exact batch, 48 B per group of four points a "
                              "mask survivor walks, 64 B candidate record; counted by an instrumented replay of the timed bases IN THE TIMED MODE: what "
                              "is FETCHED -- a candidate the bound dismisses after its sweep has fetched only its record) x candidates of the timed "
                              "bases / timed seconds: the per-step figure, "
                              "independent of how many bases are in flight.  The working set (point lines ~19 MB) is Infinity-Cache resident, so this is "
                              "priced against a roof the kernel is NOT bound by -- `binding` names the resource closest to its roof.  DESIGN.md section 7.",
                "binding": {"resource": binding, "fracs": fracs},
                "kernels": kernels_tbl or None,
                "kernels_note": "all four kernels of a device pass, each launch covering a group of up to 3 bases: own duration with the launches "
                                "serialised by the counter collection, VALU issue share of the SIMD quad-cycles, share of the wave-cycles spent "
                                "waiting, L2 hit rate and request rate against the L2 peak, L2->fabric read rate (FETCH_SIZE as counted, not doubled)",
                "valu": valu, "l2": l2,
                "pass_fractions": {"coarse_bitmap_L0": f_l0, "reach_bit_L1": f_l1, "subcell_mask_L2": f_l2}, "kbar": kbar, "groups_per_query": groups_per_query,
                "full_count_mode": None if (full_walk is None or full_mode is None) else dict(
                    full_walk, achieved=full_mode["value"] * full_walk["bytes_per_candidate"] / 1e9,
                    frac=full_mode["value"] * full_walk["bytes_per_candidate"] / 1e9 / HBM_PEAK_GBS,
                    note="the same figure with the early exit off: every candidate walks the structure for all n_Q queries (what rounds 1-2 reported)"),
                "per_launch": {"avg_launch_ms": avg_ms, "launches": int(prof.verify_launches), "candidates_per_launch": cand_per_launch,
                               "achieved": cand_per_launch * gather_b / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0,
                               "note": "HIP-event duration of a launch in the default configuration: a launch covers a group of up to three bases and the "
                                       "launches of the other groups in flight stretch it, so this is not a per-step cost",
                               "exclusive": None if exclusive is None else dict(
                                   exclusive, achieved=exclusive["candidates_per_launch"] * gather_b / (exclusive["avg_launch_ms"] * 1e-3) / 1e9,
                                   frac=exclusive["candidates_per_launch"] * gather_b / (exclusive["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   note="same bases with one base in flight (S4P_LANES=1): k_verify's own launch time")},
                "lds_sweep": {"bytes_per_candidate": sweep_b * 1.5, "note": "12 B per query out of the workgroup's LDS copy of the float queries "
                              "(lean sweep) + one 4-byte word of the LDS-resident coarse bitmap; never leaves the CU"},
                "survey_8d_model": {"bytes_per_candidate": survey_b, "GBps": value * survey_b / 1e9,
                                    "note": "SURVEY.md 8d figure (27 cells x 8 B per query, no cache credit): a cell-probing kernel this one "
                                            "replaced; kept for reference, not a roofline fraction"},
                "hbm_bound_point": hbm_point,
            },
            "k_apply": apply_row,
            "stage_ms_per_step": None if stage_prof is None else {
                "pairs_and_prep": stage_prof.pairs_ms_total / max(stage_prof.quads_launches, 1),
                "quads_and_gate": stage_prof.quads_ms_total / max(stage_prof.quads_launches, 1),
                "verify_and_select": stage_prof.verify_ms_total / max(stage_prof.verify_launches, 1),
                "note": "HIP-event time per launch (a group of up to three bases) with the other groups in flight, from a separate pass over the same "
                        "bases with events around every stage (the timed passes record events around k_verify only)"},
        }
        _lap("assembling the line")
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(P, Q, args.cpu_seconds, args.sample, ttr["candidates_verified"] if ttr else 0)
        else:
            out["cpu_baseline"] = None
        _lap("cpu_baseline (reference on one core + oracle on all cores)")
        out["provenance"] = provenance
        out["extra"] = extra_sample_finish(extra_proc) if extra_proc is not None else None
        _lap("waiting for the extra process (the 20 000-point sample, two bases)")
        out["config"]["phase_seconds"] = dict(_PHASES)
        if world == 1 and args.ttr_configs and out["extra"] is not None:
            # the metric's second half on the other BASELINE configs: a process of its own per run, reported only
            try:
                pr = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "init_timing.py")] + args.ttr_configs.split(","),
                                    stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=900)
                out["extra"]["time_to_register_configs"] = [json.loads(ln) for ln in pr.stdout.decode().splitlines() if ln.startswith("{")]
            except Exception as e:                              # noqa: BLE001
                out["extra"]["time_to_register_configs"] = {"error": type(e).__name__}
        def clean(o):                                           # (no NaN in the JSON line: a figure that was not measured is null)
            if isinstance(o, dict):
                return {k: clean(v) for k, v in o.items()}
            if isinstance(o, (list, tuple)):
                return [clean(v) for v in o]
            if isinstance(o, float) and o != o:
                return None
            return o
        print(json.dumps(clean(out)))
        sys.stdout.flush()
        if parity is not None and parity["mismatches"]:
            print("PARITY GATE FAILED: %s" % parity.get("failed"), file=sys.stderr)
            if dist is not None:
                dist.destroy_process_group()
            sys.exit(3)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
