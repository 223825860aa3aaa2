"""The parity gates of the bench line: the timed bases replayed on the oracle (N = 1), and against the committed golden record at the GPU-scale sample."""
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


def parity_gate(P, Q, opt, warmup, n_bases, full_bases, device, sample):
    """The W warm-up + n_bases timed bases of the seeded sequence, one by one, on a fresh GPU matcher and on the oracle.
    Returns (parity object, oracle state after the last base, oracle matcher for recounts)."""
    from oracle import oracle as O
    from super4pcs_amd import capi
    O.build()
    nproc = os.cpu_count() or 1
    oopt = O.make_options(DELTA, OVERLAP, sample)
    om_ref = O.Matcher(oopt, full_counts=False, use_kdtree=True, keep_trace=True)    # reference semantics (early exit)
    om_ref.set_threads(nproc)                                                         # candidates under OpenMP: same results as the serial loop
    om_full = O.Matcher(oopt, full_counts=True, use_kdtree=True, keep_trace=False)   # stage-wise, every inlier counted
    om_full.set_threads(nproc)
    om_ref.init(P, Q)
    om_full.init(P, Q)
    gm = capi.Matcher(opt, device=device, max_pairs=MAX_PAIRS, max_quads=MAX_QUADS)
    gm.init_full(P, Q)
    mism = []
    out = {"bases": 0, "warmup_bases": warmup, "quads": 0, "candidates": 0, "bases_with_every_candidate_counted": 0,
           "candidates_count_checked": 0}

    def check(ok, what):
        if not ok:
            mism.append(what)

    check(np.array_equal(gm.sampled(0), om_ref.cloud(0)) and np.array_equal(gm.sampled(1), om_ref.cloud(1)), "sampled clouds")
    gi, os_ = gm.info(), om_ref.stats()
    check((gi.n_sampled_p, gi.n_sampled_q, gi.number_of_trials) == (os_.n_P, os_.n_Q, os_.number_of_trials), "sizes / trial count")
    check(gi.best_lcp == os_.best_lcp, "initial LCP (Verify(identity))")
    eps = 2.0 * DELTA
    cand_before = 0
    for b in range(warmup + n_bases):
        timed = b >= warmup
        g_ok, r = gm.try_one_base()                                   # the fused device pass, as timed
        want_full = timed and out["bases_with_every_candidate_counted"] < full_bases and r.n_quads > 0
        if want_full:
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
        # the stage-wise oracle walks the same sequence (RNG + pair-octree permutation); full counts where asked for
        ok, i1, i2, obase, bx = om_full.select_quadrilateral()
        if ok:
            p1 = om_full.extract_pairs(seg_len32(bx[0], bx[1]), 0.0, eps, 0, 1)
            p2 = om_full.extract_pairs(seg_len32(bx[2], bx[3]), 0.0, eps, 2, 3)
            if want_full:
                o_quads = om_full.find_congruent(i1, i2, eps, p1, p2, cap=max(int(r.n_quads) + 16, 1 << 16)) if (len(p1) and len(p2)) else np.zeros((0, 4), np.int32)
                same = o_quads.shape == g_quads.shape and np.array_equal(o_quads, g_quads)
                check(same, "base %d: congruent quads (std::set order)" % b)
                if same and len(o_quads):
                    _nb, per, _bc, _bi = om_full.try_congruent_set(obase, o_quads)          # EVERY candidate, full counts
                    check(np.array_equal(per, g_counts), "base %d: per-candidate inlier counts" % b)
                    out["candidates_count_checked"] += int((per >= 0).sum())
                    out["bases_with_every_candidate_counted"] += 1
        if timed:
            out["quads"] += int(r.n_quads); out["candidates"] += int(r.n_verified); out["bases"] += 1
        else:
            cand_before += int(r.n_verified)
    T, lcp, base, cong, _c1, _c2 = om_ref.best()
    state = {"best_lcp": lcp, "base": base.tolist(), "congruent": cong.tolist(), "transform": T.copy(),
             "candidates_timed": int(om_ref.stats().n_verified) - cand_before}
    check(state["candidates_timed"] == out["candidates"], "candidates over the timed bases: GPU replay %d vs oracle %d" % (out["candidates"], state["candidates_timed"]))
    out["mismatches"] = len(mism)
    out["what"] = ("the %d warm-up + %d timed bases of the seeded sequence, base by base on a fresh GPU matcher and on the oracle in the "
                   "reference's mode (kd-tree Verify with early exit, candidates under OpenMP): pair / quad / candidate counts, TryOneBase's "
                   "return value, running best LCP, winning base + quad and 4x4; ordered quad list and the inlier count of EVERY candidate of "
                   "%d base(s) against the oracle in full-count mode; the final state and the candidate total of every timed repeat "
                   "against the oracle's" % (warmup, n_bases, out["bases_with_every_candidate_counted"]))
    if mism:
        out["failed"] = mism[:20]
    del gm
    return out, state, om_full


def parity_gate_scale(P, Q, opt, warmup, n_bases, device, sample, sample_mod=1 << 18):
    """Parity at the "GPU-scale" sample (n = 20 000): a base has ~10^9 congruent quads, which neither the reference's
    std::set nor the oracle's list form can hold.  Per base the oracle's STREAMING enumeration (OpenMP over the second pair
    set; pinned to the list form on small cases by tests/test_oracle.py) gives the number of quads, the number that pass
    the rms gate and order-independent checksums of both, plus a deterministic subsample of the gated quads; the GPU's fused
    (chunked) pass must reproduce all four numbers, its winner's gate + inlier count are recomputed by the oracle's kd-tree
    Verify, no sampled candidate may beat it, and the sampled candidates' counts through the stage-level entry point equal
    the oracle's."""
    from oracle import oracle as O
    from super4pcs_amd import capi
    O.build()
    om = O.Matcher(O.make_options(DELTA, OVERLAP, sample), full_counts=True, use_kdtree=True, keep_trace=False)
    om.set_threads(os.cpu_count() or 1)
    om.init(P, Q)
    gm = capi.Matcher(opt, device=device)
    gm.init_full(P, Q)
    ctx = capi.Context(opt, device=device, max_pairs=32 << 20, max_quads=1 << 20)
    ctx.set_clouds(om.cloud(0), om.cloud(1))
    mism = []
    out = {"bases": 0, "warmup_bases": warmup, "quads": 0, "candidates": 0, "candidates_count_checked": 0}

    def check(ok, what):
        if not ok:
            mism.append(what)

    check(np.array_equal(gm.sampled(0), om.cloud(0)) and np.array_equal(gm.sampled(1), om.cloud(1)), "sampled clouds")
    eps = 2.0 * DELTA
    bm = {"tests": 0, "l0": 0, "l1": 0, "l2": 0, "queries": 0}
    for b in range(warmup + n_bases):
        g_ok, r = gm.try_one_base()
        ok, i1, i2, base, bx = om.select_quadrilateral()
        if not ok:
            check(r.n_pairs1 == 0 and r.n_quads == 0, "base %d: no base found by the oracle" % b)
            continue
        p1 = om.extract_pairs_cap(seg_len32(bx[0], bx[1]), 0.0, eps, 0, 1, 1 << 25)
        p2 = om.extract_pairs_cap(seg_len32(bx[2], bx[3]), 0.0, eps, 2, 3, 1 << 25)
        check((r.n_pairs1, r.n_pairs2) == (len(p1), len(p2)), "base %d: pair counts" % b)
        if not (len(p1) and len(p2)):
            continue
        want = om.count_congruent(i1, i2, eps, p1, p2, base=base, sample_mod=sample_mod, sample_cap=1 << 15)
        check((r.n_quads, r.quad_checksum) == (want["K"], want["quad_sum"]), "base %d: quads %d / checksum vs oracle %d" % (b, r.n_quads, want["K"]))
        check((r.n_verified, r.cand_checksum) == (want["C"], want["cand_sum"]), "base %d: candidates %d / checksum vs oracle %d" % (b, r.n_verified, want["C"]))
        if r.n_verified:
            _nb, w_per, _bc, _bi = om.try_congruent_set(base, np.array([list(r.best_quad)], np.int32))
            check(int(w_per[0]) == int(r.best_count), "base %d: winner's inlier count %d vs oracle %d" % (b, r.best_count, int(w_per[0])))
            smp = want["sample"][:256]
            if len(smp):
                ctx.set_base(bx)
                _nb, o_per, _bc, _bi = om.try_congruent_set(base, smp)
                _gr, g_per = ctx.try_congruent_set(base, smp)
                check(np.array_equal(g_per, o_per), "base %d: inlier counts of %d sampled candidates" % (b, len(smp)))
                check(int(o_per.max()) <= int(r.best_count), "base %d: a sampled candidate beats the reported winner" % b)
                out["candidates_count_checked"] += int(len(smp))
                if b >= warmup:          # the byte model's inputs on this deterministic subsample of the base's candidates
                    Ts = np.stack([om.compute_rigid(base, q)[2] for q in smp])
                    st = ctx.verify_stats(Ts)
                    for k in ("tests", "l0", "l1", "l2"):
                        bm[k] += st[k]
                    bm["queries"] += len(smp) * int(om.cloud(1).shape[0])
        if b >= warmup:
            out["quads"] += int(r.n_quads); out["candidates"] += int(r.n_verified); out["bases"] += 1
    out["chunk_stats"] = gm.chunk_stats()
    out["mismatches"] = len(mism)
    out["what"] = ("%d timed bases at sample size %d: pair counts, number of congruent quads and of gated candidates with their "
                   "order-independent checksums against the oracle's streaming enumeration; the winner's gate and inlier count and the counts "
                   "of a deterministic subsample of the candidates against the oracle's kd-tree Verify" % (n_bases, sample))
    if mism:
        out["failed"] = mism[:20]
    state = {"candidates_timed": out["candidates"], "byte_model": bm}
    return out, state, None
