"""The `extra` figure: the benchmarked clouds at SURVEY 8d's GPU-scale sample (n = 20 000), in a process of its own."""
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


def scale_golden_inner(args, device):
    """The `extra` figure's own process: the benchmarked clouds at SURVEY 8d's GPU-scale sample (n = 20 000 sampled Q points:
    ~27 M ordered pairs and ~10^9 congruent quads per base, every base chunked).  The seeded bases the oracle's committed record
    covers (tests/golden/scale_config2_n20000.json: trials 0 and 1, written by tests/golden/make_scale_golden.py on the CPU --
    the oracle needs the better part of an hour for them) run one at a time through TryOneBase; the LAST one is the timed
    base.  Every base is checked against the record: pair counts, number of congruent quads and of gated candidates with
    their order-independent checksums, the inlier counts of a deterministic subsample of the gated quads (stage-level entry
    point), and the winner must not be beaten by any sampled candidate.  Prints one JSON object."""
    import hashlib
    from super4pcs_amd import capi, datasets
    G = json.load(open(GOLDEN_SCALE))
    P, Q, _ = datasets.bumpy_pair(args.points, overlap=OVERLAP, delta=DELTA, seed=SEED)
    opt = capi.make_options(DELTA, OVERLAP, args.sample)
    m = capi.Matcher(opt, device=device, max_pairs=32 << 20, max_quads=32 << 20)
    m.init_full(P, Q)
    info = m.info()
    mism = []

    def check(ok, what):
        if not ok:
            mism.append(what)

    check((info.n_sampled_p, info.n_sampled_q, info.number_of_trials) == (G["n_P"], G["n_Q"], G["number_of_trials"]), "sampled clouds / trial count")
    saved = os.environ.get("S4P_LANES")
    os.environ["S4P_LANES"] = "1"                       # the stage-level context needs one lane
    ctx = capi.Context(opt, device=device, max_pairs=1 << 20, max_quads=1 << 20)
    if saved is None:
        os.environ.pop("S4P_LANES", None)
    else:
        os.environ["S4P_LANES"] = saved
    ctx.set_clouds(m.sampled(0), m.sampled(1))
    m.loop_begin()                                      # (the trial loops' mode: commits refresh the early-exit bound)
    timed = None
    parity = {"bases": 0, "quads": 0, "candidates": 0, "candidates_count_checked": 0, "golden": os.path.relpath(GOLDEN_SCALE, ROOT),
              "golden_sha16": hashlib.sha256(open(GOLDEN_SCALE, "rb").read()).hexdigest()[:16]}
    for rec in G["bases"]:
        t0 = time.perf_counter()
        _ok, r = m.try_one_base()
        dt = time.perf_counter() - t0
        b = rec["trial"]
        check((r.n_pairs1, r.n_pairs2) == (rec["pairs"][0]["n"], rec["pairs"][1]["n"]), "base %d: pair counts" % b)
        check((r.n_quads, "%016x" % r.quad_checksum) == (rec["K"], rec["quad_sum"]), "base %d: quads %d / checksum vs the oracle's %d" % (b, r.n_quads, rec["K"]))
        check((r.n_verified, "%016x" % r.cand_checksum) == (rec["C"], rec["cand_sum"]), "base %d: candidates %d / checksum vs the oracle's %d" % (b, r.n_verified, rec["C"]))
        smp = np.array(rec["sample_quads"], np.int32).reshape(-1, 4)
        want = np.array(rec["sample_counts"], np.int32)
        ctx.set_base(m.sampled(0)[np.array(rec["base"])])
        _gr, g_per = ctx.try_congruent_set(np.array(rec["base"], np.int32), smp)
        check(np.array_equal(g_per, want), "base %d: inlier counts of %d sampled candidates" % (b, len(smp)))
        check(bool(r.has_best) and int(want.max()) <= int(r.best_count), "base %d: a sampled candidate beats the reported winner" % b)
        parity["bases"] += 1; parity["quads"] += int(r.n_quads); parity["candidates"] += int(r.n_verified); parity["candidates_count_checked"] += int(len(smp))
        timed = {"seconds": dt, "candidates": int(r.n_verified), "trial": b}
    m.loop_end()
    parity["mismatches"] = len(mism)
    parity["what"] = ("every base of this run (trials %s; the last one is the timed base): pair counts, number of congruent quads and of gated candidates "
                      "with their order-independent checksums, the inlier counts of a deterministic subsample of the gated quads, and no sampled "
                      "candidate beats the winner -- against the oracle's committed record" % [r_["trial"] for r_ in G["bases"]])
    if mism:
        parity["failed"] = mism[:20]
    st = m.chunk_stats()
    out = {"sample_size": args.sample, "value": timed["candidates"] / timed["seconds"], "unit": "candidates/s", "ms_per_step": timed["seconds"] * 1e3,
           "steps": 1, "warmup": len(G["bases"]) - 1, "n_Q": info.n_sampled_q, "candidates_timed": timed["candidates"], "timed_trial": timed["trial"],
           "chunked_bases": st["bases"], "chunk_passes": st["passes"], "k_verify": m.verify_kernel_info(), "parity": parity}
    m.close()
    print(json.dumps(out))
    return 1 if mism else 0


def extra_sample_start(args, sample=20000):
    """SURVEY 8d's other sample size of the benchmarked clouds (n = 20 000 sampled Q points) as a second, reported figure of the
    driver's own command, in a process of its own (scale_golden_inner): one warm-up base + one timed base, both checked against
    the oracle's committed record.  The live-oracle gate of that size is `bench.py --sample 20000`; the GPU test of the same
    record is tests/test_gpu_configs.py::test_config2_gpu_scale_sample_20000.  Started while the parent is in its host-bound legs
    (extra_sample_finish collects it)."""
    if args.points != N_POINTS or not os.path.exists(GOLDEN_SCALE):
        return {"sample": sample, "t0": time.perf_counter(), "proc": None, "error": "no golden record for this workload (%s)" % os.path.relpath(GOLDEN_SCALE, ROOT)}
    cmd = [sys.executable, BENCH_PY, "--sample", str(sample), "--points", str(args.points), "--scale-golden-inner"]
    try:
        return {"sample": sample, "t0": time.perf_counter(), "proc": subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)}
    except Exception as e:                                      # noqa: BLE001 -- the bench line must still be printed
        return {"sample": sample, "t0": time.perf_counter(), "proc": None, "error": type(e).__name__}


def extra_sample_finish(h, timeout_s=300):
    if h.get("proc") is None:
        return {"sample_size": h["sample"], "error": h.get("error", "not started"), "wall_s": 0.0}
    try:
        stdout, _ = h["proc"].communicate(timeout=timeout_s)
        line = [ln for ln in stdout.decode().splitlines() if ln.startswith("{")][-1]
        d = json.loads(line)
        d["wall_s"] = time.perf_counter() - h["t0"]
        d["note"] = ("same clouds, sample_size %d: one timed base after one warm-up base, separate process started after the GPU-exclusive "
                     "measurements of this run and running beside its host-bound legs (oracle replay, CPU baselines); reported, not `value`" % h["sample"])
        return d
    except Exception as e:                                      # noqa: BLE001
        try:
            h["proc"].kill()
        except Exception:                                       # noqa: BLE001
            pass
        return {"sample_size": h["sample"], "error": "%s" % type(e).__name__, "wall_s": time.perf_counter() - h["t0"]}
