"""rocprofv3 passes of the timed region (kernel trace + counters) and the HBM-bound point of the scoring kernel."""
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

from benchlib.workload import *          # noqa: F401,F403 -- the workload's constants and byte models
from benchlib.workload import BENCH_PY, GOLDEN_SCALE, ROOT


def pmc_passes(args, n_timed_launches, timeout_s=240):
    """rocprofv3 --pmc passes over an inner run of this script with the SAME warm-up + timed bases and the default lanes
    (counters serialise the launches: per-kernel numbers are the kernel's own).  Three passes -- the TCC slots do not hold
    FETCH_SIZE and WRITE_SIZE together (MI355X_MICROARCH.md "rocprofv3 PMC slots").  Returns {counter: (mean per k_verify
    launch, launches)}, notes, and the same counters + the kernel's own duration (kernel trace of the counter passes) for ALL
    FOUR kernels of a device pass ({kernel: {...}}: every launch covers a group of bases)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {}, ["rocprofv3 not found"]
    got, note = {}, []
    KERNELS = ("k_pairs2", "k_prep", "k_quads", "k_verify")
    per_kernel = {k: {} for k in KERNELS}

    def kname(row_name):
        for k in KERNELS:
            if ("s4p::%s<" % k) in row_name or ("s4p::%s(" % k) in row_name:
                return k
        return None
    for ctrs in (["FETCH_SIZE", "GRBM_GUI_ACTIVE", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY"],
                 ["WRITE_SIZE"], ["TCC_HIT_sum", "TCC_MISS_sum"]):
        d = tempfile.mkdtemp(prefix="s4p_pmc_", dir="/tmp")
        cmd = [exe, "--pmc"] + ctrs + ["--kernel-trace", "-d", d, "-o", "p", "--output-format", "csv", "--",
               sys.executable, BENCH_PY, "--inner", "--steps", str(args.steps), "--warmup", str(args.warmup),
               "--points", str(args.points), "--sample", str(args.sample)]
        env = dict(os.environ, TMPDIR="/tmp")
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=True)
            vals = {}
            allk = {k: {} for k in KERNELS}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") not in ctrs:
                        continue
                    if "k_verify<" in row.get("Kernel_Name", ""):
                        vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
                    kn = kname(row.get("Kernel_Name", ""))
                    if kn:
                        allk[kn].setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            for kn, cs in allk.items():
                for cname, v in cs.items():
                    per_kernel[kn][cname] = float(np.mean(v[len(v) // 4:]))      # (skip the warm-up quarter)
                    per_kernel[kn]["launches"] = len(v)
            if "GRBM_GUI_ACTIVE" in ctrs:                        # the kernels' own durations: launches are serialised under --pmc
                dur = {k: [] for k in KERNELS}
                for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
                    for row in csv.DictReader(open(f)):
                        kn = kname(row.get("Kernel_Name", ""))
                        if kn:
                            dur[kn].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
                for kn, v in dur.items():
                    if v:
                        per_kernel[kn]["avg_us"] = float(np.mean(v[len(v) // 4:]))
            # the inner run launches warm-up + timed bases (a launch covers a group of bases): keep the timed region's launches,
            # as many as the main run counted with its HIP events (the last ones)
            for k, v in vals.items():
                v = v[-n_timed_launches:] if (n_timed_launches and len(v) >= n_timed_launches) else v
                got[k] = (float(np.mean(v)), len(v))
            if args.profile_dir:
                os.makedirs(args.profile_dir, exist_ok=True)
                for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                    rows = [r for r in csv.DictReader(open(f)) if "k_verify<" in r.get("Kernel_Name", "")]
                    if rows:
                        with open(os.path.join(args.profile_dir, "pmc_%s_k_verify.csv" % "_".join(ctrs)[:60]), "w", newline="") as fo:
                            w = csv.DictWriter(fo, fieldnames=["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"], extrasaction="ignore")
                            w.writeheader(); w.writerows(rows)
        except Exception as e:                                  # noqa: BLE001 -- the bench line must still be printed
            note.append("%s pass failed: %s" % ("+".join(ctrs), type(e).__name__))
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return got, note, per_kernel



def part_in_whole_structure(device, n_transforms, seed=11):
    """BASELINE configs[4]'s structure (100 k-point query in a 10 M-point scene: n_P ~ 4.2 M sampled scene points -> ~1.4 GB
    of point lines, >> the 256 MB Infinity Cache) and a batch of transforms that slide the query over the WHOLE scene
    (uniform over the ground's extent, any yaw), so that consecutive transforms do not share cache lines."""
    from super4pcs_amd import capi, datasets
    delta = 0.05
    P, Q, T_gt = datasets.part_in_whole_pair(10_000_000, 100_000, delta=delta)
    opt = capi.make_options(delta, 0.2, 5000)
    m = capi.Matcher(opt, device=device, max_pairs=1 << 20, max_quads=1 << 20)
    m.init_full(P, Q)
    i = m.info()
    Ps, Qs = m.sampled(0), m.sampled(1)
    m.close()
    ctx = capi.Context(opt, device=device, max_pairs=1 << 20, max_quads=1 << 20)
    ctx.set_clouds(Ps, Qs)
    cP, cQ = np.array(i.centroid_p, np.float64), np.array(i.centroid_q, np.float64)
    Tg = np.asarray(T_gt, np.float64)
    Tc = np.eye(4)
    Tc[:3, :3] = Tg[:3, :3]
    Tc[:3, 3] = Tg[:3, :3] @ cQ + Tg[:3, 3] - cP
    lo, hi = Ps.min(axis=0).astype(np.float64), Ps.max(axis=0).astype(np.float64)
    rng = np.random.default_rng(seed)
    Ts = []
    for _ in range(n_transforms):
        Tp = np.eye(4)
        a = rng.uniform(-np.pi, np.pi)
        Tp[:3, :3] = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
        Tp[:3, 3] = [rng.uniform(0.8 * lo[0], 0.8 * hi[0]), rng.uniform(0.8 * lo[1], 0.8 * hi[1]), rng.uniform(-0.05, 0.05)]
        Ts.append((Tp @ Tc).astype(np.float32))
    return ctx, Ps, Qs, np.stack(Ts), opt


def hbm_point_inner(device, n_transforms):
    """(run under rocprofv3 by hbm_bound_point) ONE cold s4p_verify_transforms over the configs[4] structure; prints the counts."""
    ctx, Ps, Qs, Ts, _ = part_in_whole_structure(device, n_transforms)
    t0 = time.perf_counter()
    counts = ctx.verify_transforms(Ts)               # first and only scoring launch of this process: nothing is warm
    dt = time.perf_counter() - t0
    print(json.dumps({"seconds_wall": dt, "n_P": int(Ps.shape[0]), "n_Q": int(Qs.shape[0]), "counts_head": counts[:128].tolist(),
                      "mean_inliers": float(np.mean(counts))}))


def hbm_bound_point(args, device, n_transforms=4096, timeout_s=300):
    """One HBM-bound operating point of the same scoring code, measured by rocprofv3 in a process of its own: kernel
    duration from --kernel-trace and FETCH_SIZE from --pmc of the SINGLE, cold k_verify_T launch; the first 64 counts are
    recomputed by the oracle's kd-tree Verify on the same sampled clouds (parity of the timed launch itself)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    d = tempfile.mkdtemp(prefix="s4p_hbm_", dir="/tmp")
    cmd = [exe, "--pmc", "FETCH_SIZE", "--kernel-trace", "-d", d, "-o", "h", "--output-format", "csv", "--",
           sys.executable, BENCH_PY, "--hbm-point-inner", "--hbm-transforms", str(n_transforms)]
    out = {}
    try:
        pr = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout_s, check=True)
        inner = json.loads([l for l in pr.stdout.splitlines() if l.startswith("{")][-1])
        dur_ns, fetch = None, None
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if "k_verify_T<" in row.get("Kernel_Name", ""):
                    dur_ns = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if "k_verify_T<" in row.get("Kernel_Name", "") and row.get("Counter_Name") == "FETCH_SIZE":
                    fetch = float(row["Counter_Value"])
            if args.profile_dir:
                os.makedirs(args.profile_dir, exist_ok=True)
                shutil.copy(f, os.path.join(args.profile_dir, "hbm_point_counter_collection.csv"))
        if args.profile_dir:
            for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
                shutil.copy(f, os.path.join(args.profile_dir, "hbm_point_kernel_trace.csv"))
        if dur_ns is None or fetch is None:
            return {"error": "no k_verify_T row in the rocprofv3 output"}
        # the byte model's inputs for this structure: an instrumented pass over the same transforms
        from oracle import oracle as O
        ctx, Ps, Qs, Ts, _ = part_in_whole_structure(device, n_transforms)
        stats = ctx.verify_stats(Ts)                 # survivors per level, exact point tests
        n_q = Qs.shape[0]
        queries = float(len(Ts)) * n_q
        groups = stats["tests"] / 4.0 / queries      # (listed points of mask survivors / 4: an upper bound of the groups walked)
        _sweep, gathers = structure_bytes_per_candidate(n_q, stats["l0"] / queries, stats["l1"] / queries, stats["l2"] / queries, groups)
        fetch_b = fetch * 1024.0 * 2.0               # KB -> B, x2: gfx950 tallies 128-B read requests at 64 B (MI355X_MICROARCH.md, HBM)
        t = dur_ns * 1e-9
        out = {"workload": "configs[4] structure: n_P=%d sampled scene points, n_Q=%d, %d transforms spread over the whole scene, ONE cold launch"
                           % (inner["n_P"], n_q, len(Ts)),
               "kernel_ms": dur_ns * 1e-6, "transforms_per_s": len(Ts) / t, "mean_inliers": inner["mean_inliers"],
               "fetch_bytes": fetch_b, "measured_GBps": fetch_b / t / 1e9, "measured_frac": fetch_b / t / 1e9 / HBM_PEAK_GBS,
               "algorithmic_bytes": len(Ts) * gathers, "achieved_GBps": len(Ts) * gathers / t / 1e9, "peak_GBps": HBM_PEAK_GBS,
               "frac": len(Ts) * gathers / t / 1e9 / HBM_PEAK_GBS,
               "counts_checked_by_oracle": 0, "count_mismatches": None,
               "note": "`frac` is the ALGORITHMIC figure (dependent gathers the structure requires / kernel time); measured_frac doubles FETCH_SIZE, a factor "
                       "the guide calibrates for wide coalesced streams only -- uncalibrated for 16-byte gathers, an upper bound here.  "
                       "kernel duration and FETCH_SIZE (doubled: gfx950 tallies 128-B requests at 64 B) of the single cold k_verify_T launch, "
                       "rocprofv3 --kernel-trace --pmc FETCH_SIZE in a process of its own; algorithmic bytes = dependent gathers the structure "
                       "requires (reach words, 32-B headers, query re-reads, 48-B point groups) from the instrumented kernel's counters"}
        # parity of the TIMED launch: the oracle's kd-tree Verify recounts the first 64 transforms on exactly the sampled,
        # centred clouds the context was given (the inner process builds them the same way: same seeds)
        try:
            om = O.Matcher(O.make_options(0.05, 0.2, 5000), full_counts=True, use_kdtree=True)
            om.set_sampled(Ps, Qs)
            want = om.verify_batch(Ts[:64])
            got = np.array(inner["counts_head"][:64], np.int64)
            out["counts_checked_by_oracle"] = 64
            out["count_mismatches"] = int((want.astype(np.int64) != got).sum())
        except Exception as e:                                  # noqa: BLE001
            out["count_mismatches"] = "oracle recount failed: %s" % e
    except Exception as e:                                      # noqa: BLE001
        out = {"error": "%s: %s" % (type(e).__name__, e)}
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return out
