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

The timed region (W warm-up steps, then exactly K steps between barrier + device synchronisation) is repeated
`--repeats` times on a fresh matcher with the same seed, i.e. over the SAME bases, so that the spread is timing noise and
not workload variation; `value` / `ms_per_step` are the median repeat, `spread` holds min / median / max.

Every measurement carries a parity gate: the first timed bases are replayed through the oracle
(oracle/, CPU) and compared -- quads in reference order, per-candidate inlier counts, winner, best LCP, transform -- and the
process exits non-zero on any mismatch.  The oracle is only ever the checker and the cpu_baseline, never the thing timed.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
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
L2_PEAK_GBS = 34500.0          # MI355X_MICROARCH.md: aggregate L2 bandwidth, ~34.5 TB/s
MAX_PAIRS, MAX_QUADS = 8 << 20, 64 << 20


def survey_bytes_per_candidate(n_q, kbar, cells=27):
    """SURVEY.md 8d, no cache credit: B_cand = 16 (quad read) + 8 (count write) + n_Q * (12 + c*8 + kbar*12)."""
    return 16 + 8 + n_q * (12 + cells * 8 + kbar * 12)


def structure_bytes_per_candidate(n_q, f_l0, f_l1, kbar):
    """Bytes the three-level LCP structure REQUIRES per verified candidate:
    query sweep 16 B/query (q4v, a 32 KB array every workgroup re-reads: served by L1/L2), reach word 8 B per L0 survivor,
    list header 16 B + query re-read 16 B per L1 survivor, 16 B per exact point test, 48 B transform + 8 B tag + 4 B
    index in, 4 B count out.  The first class never leaves the CU's L1/L2; the rest are dependent gathers into
    structures (reach words 0.5 MB, headers 2.3 MB, point lists 19 MB on the bench workload) that live beyond the L2
    of any single XCD."""
    sweep = 16.0 * n_q
    gathers = n_q * (8.0 * f_l0 + 32.0 * f_l1 + 16.0 * kbar) + 64.0
    return sweep, gathers


def seg_len32(a, b):
    """float32 |a - b| in the reference's evaluation order x + (y + z) (match4pcsBase.hpp:318-321, Eigen 3-vector norm)."""
    d = (np.asarray(a, np.float32) - np.asarray(b, np.float32)).astype(np.float32)
    s = np.float32(d[0] * d[0]) + (np.float32(d[1] * d[1]) + np.float32(d[2] * d[2]))
    return float(np.sqrt(np.float32(s)))


def parity_gate(P, Q, T_gt, opt, warmup, n_bases, per_base_sample, device):
    """Replays the first `n_bases` TIMED bases (the ones after `warmup`) of the seeded sequence on a fresh GPU matcher and
    on the oracle and compares everything the reference's TryOneBase produces.  Returns the `parity` object."""
    from oracle import oracle as O
    from super4pcs_amd import capi
    O.build()
    oopt = O.make_options(DELTA, OVERLAP, int(opt.sample_size))
    om_ref = O.Matcher(oopt, full_counts=False, use_kdtree=True, keep_trace=True)    # reference semantics (early exit)
    om_full = O.Matcher(oopt, full_counts=True, use_kdtree=True, keep_trace=False)   # stage-wise, every inlier counted
    om_ref.init(P, Q)
    om_full.init(P, Q)
    gm = capi.Matcher(opt, device=device, max_pairs=MAX_PAIRS, max_quads=MAX_QUADS)
    gm.init_full(P, Q)
    mism = []
    out = {"bases": 0, "quads": 0, "candidates": 0, "candidates_count_checked": 0}

    def check(ok, what):
        if not ok:
            mism.append(what)

    check(np.array_equal(gm.sampled(0), om_ref.cloud(0)) and np.array_equal(gm.sampled(1), om_ref.cloud(1)), "sampled clouds")
    gi, os_ = gm.info(), om_ref.stats()
    check((gi.n_sampled_p, gi.n_sampled_q, gi.number_of_trials) == (os_.n_P, os_.n_Q, os_.number_of_trials), "sizes / trial count")
    check(gi.best_lcp == os_.best_lcp, "initial LCP (Verify(identity))")
    eps = 2.0 * DELTA
    # the warm-up bases: advance RNG + pair-octree permutation everywhere, score nothing
    for _ in range(warmup):
        for om in (om_ref, om_full):
            ok, _i1, _i2, _b, bx = om.select_quadrilateral()
            if ok:
                om.extract_pairs(seg_len32(bx[0], bx[1]), 0.0, eps, 0, 1)
                om.extract_pairs(seg_len32(bx[2], bx[3]), 0.0, eps, 2, 3)
        gm.next_base(run_device=False)
    for b in range(n_bases):
        g_ok, r = gm.try_one_base()                                   # the fused device pass, as timed
        g_quads, g_counts = gm.last_candidates(r.n_quads)
        o_ok = om_ref.try_one_base()
        rec = om_ref.trace()[0][-1]
        check(g_ok == o_ok, "base %d: TryOneBase return value" % b)
        if rec[0]:
            check((r.n_pairs1, r.n_pairs2) == (rec[5], rec[6]), "base %d: pair counts" % b)
            if rec[5] and rec[6]:
                check((r.n_quads, r.n_verified) == (rec[7], rec[8]), "base %d: quad / candidate counts" % b)
        T, lcp, base, cong, _c1, _c2 = om_ref.best()
        gi = gm.info()
        check(gi.best_lcp == lcp, "base %d: best LCP" % b)
        check(list(gi.base) == base.tolist() and list(gi.congruent) == cong.tolist(), "base %d: winning base / quad" % b)
        check(np.array_equal(np.array(gi.transform, np.float32).reshape(4, 4), T), "base %d: transform" % b)
        # stage-wise replay with full counts: ordered quads, per-candidate inlier counts
        ok, i1, i2, obase, bx = om_full.select_quadrilateral()
        if ok:
            p1 = om_full.extract_pairs(seg_len32(bx[0], bx[1]), 0.0, eps, 0, 1)
            p2 = om_full.extract_pairs(seg_len32(bx[2], bx[3]), 0.0, eps, 2, 3)
            o_quads = om_full.find_congruent(i1, i2, eps, p1, p2, cap=max(int(r.n_quads) + 16, 1 << 16)) if (len(p1) and len(p2)) else np.zeros((0, 4), np.int32)
            same = o_quads.shape == g_quads.shape and np.array_equal(o_quads, g_quads)
            check(same, "base %d: congruent quads (std::set order)" % b)
            if same and len(o_quads):
                K = len(o_quads)
                stride = max(K // max(per_base_sample, 1), 1)
                idx = np.unique(np.concatenate([np.arange(0, K, stride), np.arange(min(K, 256)),
                                                np.flatnonzero(g_counts == g_counts.max())[:4]]))
                _nb, per, _bc, _bi = om_full.try_congruent_set(obase, o_quads[idx])
                check(np.array_equal(per, g_counts[idx]), "base %d: per-candidate inlier counts" % b)
                out["candidates_count_checked"] += int((per >= 0).sum())
            out["quads"] += int(len(o_quads))
        out["candidates"] += int(r.n_verified)
        out["bases"] += 1
    out["mismatches"] = len(mism)
    out["what"] = ("first %d timed bases (after %d warm-up bases) of the seeded sequence: pair/quad/candidate counts, ordered quad list, "
                   "TryOneBase return value, best LCP, winning base+quad and 4x4 against the oracle in reference mode (early exit); "
                   "per-candidate inlier counts of a deterministic subsample against the oracle in full-count mode" % (n_bases, warmup))
    if mism:
        out["failed"] = mism
    del gm
    return out, om_full


def cpu_baseline(P, Q, budget_s):
    """CPU path on the same workload, bounded samples.
    A (reference-faithful): 1 thread -- what MatchSuper4PCS does (super4pcs.cc:68-73), kd-tree Verify with early exit.
       kind "reference": the reference's own sources (oracle/_ref/libs4p_ref.so) run ComputeTransformation and are cut by a
       visitor exception after budget_s of RANSAC time; kind "port" (the oracle) if the prebuilt library is absent.
    B (best-effort CPU): the oracle with its candidate loop under `omp parallel for` on all host
       cores, as the legacy Match4PCS does by default (match4pcsBase.h:190-192); also gives the per-stage split."""
    from oracle import oracle as O
    from oracle import reflib
    O.build()
    nproc = os.cpu_count() or 1

    def port_run(threads, seconds):
        om = O.Matcher(O.make_options(DELTA, OVERLAP, SAMPLE), full_counts=False, use_kdtree=True, keep_trace=False)
        om.set_threads(threads)
        om.init(P, Q)
        om.set_budget(seconds)
        t0 = time.perf_counter()
        bases = 0
        while time.perf_counter() - t0 < seconds:
            om.try_one_base()
            bases += 1
        dt = time.perf_counter() - t0
        s = om.stats()
        return {"value": s.n_verified / dt, "unit": "candidates/s", "cores": threads, "kind": "port",
                "sample": "oracle restatement, first %d base(s) of the same seeded sequence, TryCongruentSet cut after %.0f s wall "
                          "(%d candidates verified, kd-tree Verify with the reference's early exit)" % (bases, seconds, s.n_verified),
                "seconds": dt,
                "stage_seconds": {"select": s.t_select, "pairs": s.t_pairs, "quads": s.t_quads, "verify": s.t_verify}}

    if reflib.available():
        rm = reflib.RefMatcher(O.make_options(DELTA, OVERLAP, SAMPLE))
        cut, n, sec = rm.bench(P, Q, budget_s)
        a = {"value": n / max(sec, 1e-9), "unit": "candidates/s", "cores": 1, "kind": "reference",
             "sample": "reference ComputeTransformation (kd-tree Verify with early exit) on the same clouds/seed, "
                       "stopped after %.1f s of RANSAC time: %d candidates verified%s" % (sec, n, "" if cut else " (ran to completion)"),
             "seconds": sec}
    else:
        a = port_run(1, budget_s)
    a["host_cores"] = nproc
    a["note"] = ("the GPU scores every candidate over all n_Q points (no early exit); the CPU loops stop a candidate as soon as it "
                 "cannot beat the running best (match4pcsBase.cc:558-560), so work per candidate differs: a reported baseline")
    a["openmp_all_cores"] = port_run(nproc, max(budget_s * 0.6, 3.0))
    return a


def pmc_traffic(args, timeout_s=150):
    """HBM-side traffic of the dominant kernel (k_verify) in THIS configuration: two `rocprofv3 --pmc` passes
    (FETCH_SIZE, WRITE_SIZE: they do not fit one pass, MI355X_MICROARCH.md) over a short inner run of this script with
    the default lanes.  Returns (bytes per launch or None, note)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    got = {}
    note = []
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="s4p_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", ctr, "--kernel-trace", "-d", d, "-o", "p", "--output-format", "csv", "--",
               sys.executable, os.path.abspath(__file__), "--inner", "--steps", "30", "--warmup", "3",
               "--points", str(args.points), "--sample", str(args.sample)]
        env = dict(os.environ, TMPDIR="/tmp")
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=True)
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "k_verify<" in row.get("Kernel_Name", "") and row.get("Counter_Name") == ctr:
                        vals.append(float(row["Counter_Value"]))
            if vals:
                got[ctr] = (float(np.mean(vals)), len(vals))
        except Exception as e:                                  # noqa: BLE001 -- the bench line must still be printed
            note.append("%s pass failed: %s" % (ctr, type(e).__name__))
        finally:
            shutil.rmtree(d, ignore_errors=True)
    if "FETCH_SIZE" not in got or "WRITE_SIZE" not in got:
        return None, "; ".join(note) or "no k_verify rows in the counter output"
    # FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1 KB? rocprofv3 reports them in kilobytes (derived from 64 B / 32 B
    # requests); gfx950 tallies 128-B read requests at 64 B, hence the factor 2 on FETCH_SIZE (MI355X_MICROARCH.md, HBM).
    fetch_b = got["FETCH_SIZE"][0] * 1024.0 * 2.0
    write_b = got["WRITE_SIZE"][0] * 1024.0
    return fetch_b + write_b, ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, this run's binary and default lanes, "
                              "30 timed bases; mean per k_verify launch over %d / %d launches; FETCH_SIZE doubled (gfx950 tallies 128-B "
                              "requests at 64 B), uncalibrated for 16-B gathers; counts L2->fabric requests including Infinity-Cache hits"
                              % (got["FETCH_SIZE"][1], got["WRITE_SIZE"][1]))


def hbm_bound_point(device, budget_transforms=4096):
    """One HBM-bound operating point of the same scoring code: BASELINE configs[4] (100 k-point query in a 10 M-point scene),
    n_P ~ 4.2 M sampled scene points -> ~1.4 GB of point lists (>> 256 MB Infinity Cache).  A batch of transforms near the
    ground truth (so that most queries reach the exact stage) is scored with s4p_verify_transforms; the bytes the
    structure requires come from the instrumented kernel's own counters."""
    from super4pcs_amd import capi, datasets
    delta = 0.05
    P, Q, T_gt = datasets.part_in_whole_pair(10_000_000, 100_000, delta=delta)
    opt = capi.make_options(delta, 0.2, 5000)
    m = capi.Matcher(opt, device=device, max_pairs=1 << 20, max_quads=1 << 20)
    m.init_full(P, Q)
    i = m.info()
    Ps, Qs = m.sampled(0), m.sampled(1)
    ctx = capi.Context(opt, device=device, max_pairs=1 << 20, max_quads=1 << 20)
    ctx.set_clouds(Ps, Qs)
    cP, cQ = np.array(i.centroid_p, np.float64), np.array(i.centroid_q, np.float64)
    Tg = np.asarray(T_gt, np.float64)
    Tc = np.eye(4)
    Tc[:3, :3] = Tg[:3, :3]
    Tc[:3, 3] = Tg[:3, :3] @ cQ + Tg[:3, 3] - cP
    rng = np.random.default_rng(11)
    Ts = []
    for _ in range(budget_transforms):
        Tp = np.eye(4)
        Tp[:3, 3] = rng.uniform(-6.0, 6.0, 3) * np.array([1.0, 1.0, 0.05])       # slide the query over the scene's ground
        a = rng.uniform(-np.pi, np.pi)
        Tp[:3, :3] = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
        Ts.append((Tp @ Tc).astype(np.float32))
    Ts = np.stack(Ts)
    ctx.verify_transforms(Ts[:64])
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        counts = ctx.verify_transforms(Ts)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    stats = ctx.verify_stats(Ts)                     # instrumented pass: survivors per level, exact point tests
    n_q = Qs.shape[0]
    queries = float(len(Ts)) * n_q
    sweep, gathers = structure_bytes_per_candidate(n_q, stats["l0"] / queries, stats["l1"] / queries, stats["tests"] / queries)
    gb = len(Ts) * gathers / 1e9
    return {"workload": "configs[4] structure: n_P=%d sampled scene points, n_Q=%d, %d transforms around the ground truth"
                        % (Ps.shape[0], n_q, len(Ts)),
            "seconds": best, "transforms_per_s": len(Ts) / best, "mean_inliers": float(np.mean(counts)),
            "gather_bytes_per_transform": gathers, "achieved_GBps": gb / best, "peak_GBps": HBM_PEAK_GBS,
            "frac": gb / best / HBM_PEAK_GBS,
            "note": "wall time of s4p_verify_transforms incl. the upload of the 4x4s and the read-back of the counts (both < 1 MB); "
                    "bytes = dependent gathers the structure requires (reach words, headers, query re-reads, 16-B point records)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=5, help="timed regions (same bases, fresh matcher): value = median")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the 1-core cpu_baseline sample (0 = skip)")
    ap.add_argument("--no-time-to-register", dest="time_to_register", action="store_false", default=True)
    ap.add_argument("--no-parity", dest="parity", action="store_false", default=True)
    ap.add_argument("--parity-bases", type=int, default=2)
    ap.add_argument("--no-pmc", dest="pmc", action="store_false", default=True, help="skip the rocprofv3 --pmc passes (roofline.traffic = null)")
    ap.add_argument("--no-hbm-point", dest="hbm_point", action="store_false", default=True)
    ap.add_argument("--no-exclusive", dest="exclusive", action="store_false", default=True,
                    help="skip the one-base-in-flight re-run (roofline.exclusive); used for the rocprofv3 kernel-stats run, so "
                         "that its k_verify rows are the default configuration's launches only")
    ap.add_argument("--inner", action="store_true", help="(used by the --pmc passes) timed region only, no JSON")
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

    P, Q, T_gt = datasets.bumpy_pair(args.points, overlap=OVERLAP, delta=DELTA, seed=SEED)
    opt = capi.make_options(DELTA, OVERLAP, args.sample)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_region(steps, warmup):
        """Fresh matcher, same seed: W untimed steps, then exactly K timed steps between barrier + synchronize."""
        m = capi.Matcher(opt, device=local_rank, max_pairs=MAX_PAIRS, max_quads=MAX_QUADS)
        m.init_full(P, Q)                       # sampling, grid build, upload: outside the timed region
        if world > 1:
            # the C++ sharded loop behind the C ABI (s4p_shard_*): RCCL all-reduce(max) of one 8-byte key per window
            sh = capi.Shard(m, rank, world, True)
            if one_gpu:
                sh.use_collective(capi.torch_collective(dist))          # single-GPU dry run: gloo through the callback provider
            else:
                # the library's own communicator (ncclCommInitRank from the id rank 0 made); should RCCL not bind or not
                # initialise on some rank, every rank falls back to the process group torch already has (same 8-byte
                # all-reduce per window, through the callback provider)
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
                if not int(flag.item()):
                    sh.use_collective(capi.torch_collective(dist, dev))
        else:
            sh = sharding.ShardedRansac(m, rank, world, dist, dev)      # world 1: the engine's own pipelined Perform_N_steps
        sh.run_windows(warmup)
        m.profile_enable(True, False)
        m.profile_get(reset=True)
        sync()
        t0 = time.perf_counter()
        cand = sh.run_windows(steps)            # pipelined: host base selection of step t+1 overlaps the GPU pass of step t
        sync()
        dt = time.perf_counter() - t0
        prof = m.profile_get(reset=True)
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        tc = torch.tensor([cand], dtype=torch.int64, device=dev)
        if dist is not None:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dist.all_reduce(tc, op=dist.ReduceOp.SUM)
        return m, sh, float(tt.item()), int(tc.item()), prof

    if args.inner:                              # profiled inner run of the --pmc passes: the timed region and nothing else
        timed_region(args.steps, args.warmup)
        return

    runs = []
    m = sh = None
    for _ in range(max(args.repeats, 1)):
        if m is not None:
            if hasattr(sh, "close"):
                sh.close()
            m.close()
        m, sh, dt_max, cand_all, prof = timed_region(args.steps, args.warmup)
        runs.append((cand_all / dt_max, dt_max, cand_all, prof))
    order = sorted(range(len(runs)), key=lambda k: runs[k][0])
    med = order[len(order) // 2]
    value, dt_max, cand_all, prof = runs[med]
    info = m.info()
    n_q, n_p = info.n_sampled_q, info.n_sampled_p

    # k-bar (mean P points distance-tested per query) and the pass fractions of the three levels, from two extra,
    # untimed, instrumented bases
    m.profile_enable(False, True)
    m.profile_get(reset=True)
    q_before = m.info().candidates_verified
    sh.run_windows(2)
    pk = m.profile_get(reset=True)
    q_after = m.info().candidates_verified
    queries = max((q_after - q_before) * n_q, 1)
    kbar = pk.verify_point_tests / queries
    f_l0, f_l1, f_l2 = pk.verify_l0_pass / queries, pk.verify_l1_pass / queries, pk.verify_l2_pass / queries
    m.profile_enable(False, False)
    final_info = m.info()                       # state after the last repeat's windows (N > 1: compared across ranks below)
    m.close()

    # time-to-register (the metric's second half): one whole ComputeTransformation on the same pair, wall time from
    # call to return with inputs in host memory (sampling of both 1 M-point clouds, grid build, upload, all trials,
    # final apply).  Reported, never part of `value`.
    ttr = None
    M2 = None
    if world == 1 and args.time_to_register:
        m2 = capi.Matcher(opt, device=local_rank, max_pairs=MAX_PAIRS, max_quads=MAX_QUADS)
        m2.set_sharding(0, 1, True)
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

    parity = None
    if rank == 0 and world == 1 and args.parity:
        parity, om_full = parity_gate(P, Q, T_gt, opt, args.warmup, args.parity_bases, 2500, local_rank)
        if ttr is not None:
            # the registration's result, recounted by the oracle's kd-tree Verify on its own sampled clouds
            recount = int(om_full.verify_batch(T2c.reshape(1, 16))[0])
            ttr["oracle_recount_of_final_transform"] = recount
            if recount != ttr["best_count"]:
                parity["mismatches"] += 1
                parity.setdefault("failed", []).append("time-to-register: final LCP %d != oracle recount %d" % (ttr["best_count"], recount))

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
            trials = (args.warmup + args.steps + 2) * world      # the last repeat's windows + the two instrumented ones above
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

    traffic, traffic_note = None, "skipped"
    if rank == 0 and world == 1 and args.pmc:
        traffic, traffic_note = pmc_traffic(args)
    hbm_point = None
    if rank == 0 and world == 1 and args.hbm_point:
        try:
            hbm_point = hbm_bound_point(local_rank)
        except Exception as e:                                  # noqa: BLE001
            hbm_point = {"error": "%s: %s" % (type(e).__name__, e)}

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
    exclusive = None
    if world == 1 and rank == 0 and args.exclusive:
        saved = os.environ.get("S4P_LANES")
        os.environ["S4P_LANES"] = "1"
        try:
            m1 = capi.Matcher(opt, device=local_rank, max_pairs=MAX_PAIRS, max_quads=MAX_QUADS)
            m1.init_full(P, Q)
            m1.set_sharding(0, 1, True)
            m1.perform_n_steps(args.warmup)
            m1.profile_enable(True, False)
            m1.profile_get(reset=True)
            m1.perform_n_steps(min(args.steps, 60))
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

    if rank == 0:
        launches = max(prof.verify_launches, 1)
        avg_ms = prof.verify_ms_total / launches
        cand_per_launch = prof.verify_candidates / launches
        sweep_b, gather_b = structure_bytes_per_candidate(n_q, f_l0, f_l1, kbar)
        t = avg_ms * 1e-3
        achieved = cand_per_launch * gather_b / t / 1e9 if t > 0 else 0.0
        sweep_gbps = cand_per_launch * sweep_b / t / 1e9 if t > 0 else 0.0
        survey_b = survey_bytes_per_candidate(n_q, kbar)
        vals = sorted(r[0] for r in runs)
        out = {
            "metric": "candidate transforms verified/sec", "value": value, "unit": "candidates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "spread": {"repeats": len(runs), "min": vals[0], "median": vals[len(vals) // 2], "max": vals[-1],
                       "note": "each repeat: fresh matcher, same seed, same %d timed bases" % args.steps},
            "config": {"workload": "configs[2]: synthetic %d-point pair, 50%% overlap, Gaussian noise sigma=delta=%g, "
                                   "sample_size=%d (n_P=%d sampled P points, n_Q=%d); one step = one RANSAC base per GPU"
                                   % (args.points, DELTA, args.sample, n_p, n_q),
                       "n_P": n_p, "n_Q": n_q, "delta": DELTA, "overlap": OVERLAP, "seed": SEED,
                       "candidates_timed": cand_all, "point_queries_per_s": cand_all * n_q / dt_max,
                       "parallelism": "bases sharded over %d GPU(s), one 8-byte ncclAllReduce(max) per window (C++ loop, s4p_shard_run_windows)" % world,
                       "time_to_register": ttr},
            "parity": parity,
            "roofline": {
                "bound": "hbm", "kernel": "k_verify", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                "achieved": achieved, "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_note": traffic_note,
                "avg_launch_ms": avg_ms, "launches": int(prof.verify_launches), "candidates_per_launch": cand_per_launch,
                "algorithmic_bytes_per_candidate": gather_b,
                "definition": "achieved = gather bytes the three-level structure requires per candidate (8 B reach word per L0 survivor, "
                              "16 B header + 16 B query per L1 survivor, 16 B per exact point test, 64 B candidate record) x candidates "
                              "per launch / HIP-event launch time; these are dependent 8/16-B gathers into structures beyond one XCD's L2, "
                              "priced against the HBM peak.  The 16 B/query sweep of the 32 KB query array is L1/L2-resident and reported "
                              "separately (l2_sweep).  DESIGN.md section 7.",
                "pass_fractions": {"coarse_bitmap_L0": f_l0, "reach_bit_L1": f_l1, "subcell_mask_L2": f_l2}, "kbar": kbar,
                "l2_sweep": {"bytes_per_candidate": sweep_b, "achieved_GBps": sweep_gbps, "peak_GBps": L2_PEAK_GBS, "frac": sweep_gbps / L2_PEAK_GBS},
                "survey_8d_model": {"bytes_per_candidate": survey_b, "GBps": cand_per_launch * survey_b / t / 1e9 if t > 0 else 0.0,
                                    "note": "SURVEY.md 8d figure (27 cells x 8 B per query, no cache credit): a cell-probing kernel this one "
                                            "replaced; kept for reference, not a roofline fraction"},
                "hbm_bound_point": hbm_point,
                "exclusive": None if exclusive is None else dict(
                    exclusive, achieved=exclusive["candidates_per_launch"] * gather_b / (exclusive["avg_launch_ms"] * 1e-3) / 1e9,
                    frac=exclusive["candidates_per_launch"] * gather_b / (exclusive["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    note="same bases with one base in flight (S4P_LANES=1): k_verify's own launch time; achieved/frac above use the "
                         "launch time of the default configuration, where every launch shares the chip with the other lanes' kernels"),
                "aggregate": {"achieved": value * gather_b / 1e9, "frac": value * gather_b / 1e9 / HBM_PEAK_GBS,
                              "note": "value (candidates/s of the whole pipeline) x algorithmic bytes per candidate: what the chip "
                                      "sustains per unit of time with the default lanes"},
            },
            "k_apply": apply_row,
            "stage_ms_per_step": {"pairs_and_prep": prof.pairs_ms_total / max(prof.quads_launches, 1),
                                  "quads_and_gate": prof.quads_ms_total / max(prof.quads_launches, 1), "verify_and_select": avg_ms},
        }
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(P, Q, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
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
