#!/usr/bin/env python
"""Golden records of the ORACLE (oracle/s4p_oracle.cpp, pinned to the reference's sources by tests/test_oracle_vs_reference.py)
for seeded bases at the sample sizes SURVEY.md 8d states -- sizes at which the oracle needs minutes to hours, so its answers
are computed ONCE here (CPU, no GPU involved) and committed; the -m gpu tests and bench.py's `extra` line compare the HIP path
against them.

  configs[2] @ n = 20 000   the bench clouds (datasets.bumpy_pair, seed 20140814), first base of the seeded sequence:
      ordered pair lists of both sets (count + SHA-256 over the int32 pairs in the reference's emission order), the streaming
      enumeration's K / C and order-independent checksums of quads and gated candidates, a deterministic subsample of the gated
      quads with the oracle's full inlier count of each (kd-tree Verify).  The base's ~5 10^8 candidates cannot be verified in
      full on any CPU in reasonable time: the WINNER is pinned at test time instead (the oracle recounts the GPU's winner, no
      sampled candidate may beat it).
  configs[4] @ n = 5000     datasets.part_in_whole_pair(10 M, 100 k), trial 21 of the seeded sequence (the first cheap base with
      quads): the same, plus the streaming WINNER (every gated candidate verified in full: count_congruent_best).

usage: python tests/golden/make_scale_golden.py [config2] [config4]     (writes tests/golden/scale_*.json; ~1 h on 8 cores)"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O   # noqa: E402
from super4pcs_amd import datasets as D   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def seg_len32(a, b):
    return float(np.float32(np.linalg.norm((np.asarray(a, np.float32) - np.asarray(b, np.float32)).astype(np.float32))))


def pair_digest(p):
    return hashlib.sha256(np.ascontiguousarray(p, np.int32).tobytes()).hexdigest()


def record(om, delta, skip, with_winner, sample_mod, n_sample_counts, log, trial=None):
    eps = 2.0 * delta
    t0 = time.time()
    for _ in range(skip):                       # advance the RNG and the pair-octree permutation exactly as the trial loop does
        ok, _i1, _i2, _b, bx = om.select_quadrilateral()
        if ok:
            from bench import seg_len32 as sl
            om.extract_pairs(sl(bx[0], bx[1]), 0.0, eps, 0, 1)
            om.extract_pairs(sl(bx[2], bx[3]), 0.0, eps, 2, 3)
    ok, i1, i2, base, bx = om.select_quadrilateral()
    assert ok
    from bench import seg_len32 as sl
    sets = []
    for a, b in ((0, 1), (2, 3)):
        sets.append(om.extract_pairs_cap(sl(bx[a], bx[b]), 0.0, eps, a, b, 1 << 26))
        log("pairs set %d: %d (%.0f s)" % (len(sets), sets[-1].shape[0], time.time() - t0))
    out = {"trial": skip if trial is None else trial, "base": [int(v) for v in base], "inv1": float(i1), "inv2": float(i2),
           "pairs": [{"n": int(s.shape[0]), "sha256": pair_digest(s)} for s in sets]}
    if with_winner:
        w = om.count_congruent_best(i1, i2, eps, sets[0], sets[1], base)
        out["winner"] = {"found": w["found"], "best_count": w["best_count"], "best_quad": w["best_quad"]}
        log("streaming winner: %s (%.0f s)" % (out["winner"], time.time() - t0))
    c = om.count_congruent(i1, i2, eps, sets[0], sets[1], base=base, sample_mod=sample_mod, sample_cap=1 << 16)
    log("K=%d C=%d sample=%d (%.0f s)" % (c["K"], c["C"], len(c["sample"]), time.time() - t0))
    smp = c["sample"][:n_sample_counts]
    _nb, per, _bc, _bi = om.try_congruent_set(base, smp)
    assert (per >= 0).all()                      # the sample is drawn from the GATED quads
    out.update({"K": c["K"], "quad_sum": "%016x" % c["quad_sum"], "C": c["C"], "cand_sum": "%016x" % c["cand_sum"],
                "sample_mod": sample_mod, "sample_quads": smp.tolist(), "sample_counts": [int(v) for v in per]})
    log("sample counts done: max %d (%.0f s)" % (int(per.max()) if len(per) else -1, time.time() - t0))
    return out


def main():
    which = sys.argv[1:] or ["config2", "config4"]
    threads = os.cpu_count() or 1
    if "config2" in which:
        import bench
        def log(m): print("[config2 n=20000]", m, flush=True)
        P, Q, _ = D.bumpy_pair(bench.N_POINTS, overlap=bench.OVERLAP, delta=bench.DELTA, seed=bench.SEED)
        om = O.Matcher(O.make_options(bench.DELTA, bench.OVERLAP, 20000), full_counts=True, use_kdtree=True, keep_trace=False)
        om.init(P, Q)
        om.L.s4po_set_threads(om.h, threads)
        st = om.stats()
        # trials 0 and 1: bench.py's `extra` line runs one warm-up base and one timed base at this sample
        recs = [record(om, bench.DELTA, 0, False, 1 << 18, 300, log, trial=0)]
        json.dump({"partial": True, "bases": recs}, open(os.path.join(HERE, "scale_config2_n20000.json"), "w"), indent=0)
        recs.append(record(om, bench.DELTA, 0, False, 1 << 18, 300, log, trial=1))
        json.dump({"workload": "configs[2]: datasets.bumpy_pair(%d, overlap=%g, delta=%g, seed=%d), sample_size 20000" % (bench.N_POINTS, bench.OVERLAP, bench.DELTA, bench.SEED),
                   "n_P": st.n_P, "n_Q": st.n_Q, "number_of_trials": st.number_of_trials, "bases": recs},
                  open(os.path.join(HERE, "scale_config2_n20000.json"), "w"), indent=0)
    if "config4" in which:
        def log(m): print("[config4 n=5000]", m, flush=True)
        os.environ["S4PO_SKIP_MEAN_DISTANCE"] = "1"
        delta = 0.05
        P, Q, _ = D.part_in_whole_pair(10_000_000, 100_000, delta=delta)
        om = O.Matcher(O.make_options(delta, 0.2, 5000), full_counts=True, use_kdtree=True, keep_trace=False)
        om.init(P, Q)
        om.L.s4po_set_threads(om.h, threads)
        st = om.stats()
        rec = record(om, delta, 21, True, 1 << 10, 400, log)
        json.dump({"workload": "configs[4]: datasets.part_in_whole_pair(10000000, 100000, delta=0.05), sample_size 5000, overlap 0.2",
                   "n_P": st.n_P, "n_Q": st.n_Q, "number_of_trials": st.number_of_trials, "bases": [rec]},
                  open(os.path.join(HERE, "scale_config4_n5000.json"), "w"), indent=0)


if __name__ == "__main__":
    main()
