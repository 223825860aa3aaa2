"""CPU: the committed golden records of the oracle at the stated sample sizes (tests/golden/scale_*.json) are well
formed, and the script that wrote them (tests/golden/make_scale_golden.py) records exactly what the oracle's stage calls return --
checked on a small pair, where its `record()` can be compared with the list forms (extract_pairs / find_congruent /
try_congruent_set), the streaming winner included."""
import hashlib
import importlib.util
import json
import os

import numpy as np

from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _mod():
    spec = importlib.util.spec_from_file_location("make_scale_golden", os.path.join(GOLD, "make_scale_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_committed_records_are_well_formed():
    for name, n_q, n_bases, has_winner in (("scale_config2_n20000.json", 20000, 2, False), ("scale_config4_n5000.json", 5000, 1, True)):
        G = json.load(open(os.path.join(GOLD, name)))
        assert G["n_Q"] == n_q and G["n_P"] > n_q and G["number_of_trials"] > 0 and len(G["bases"]) == n_bases and "partial" not in G
        for b in G["bases"]:
            assert len(b["base"]) == 4 and len(b["pairs"]) == 2 and all(len(p["sha256"]) == 64 and p["n"] > 0 for p in b["pairs"])
            assert 0 < b["C"] <= b["K"] and len(b["quad_sum"]) == 16 and len(b["cand_sum"]) == 16
            assert len(b["sample_quads"]) == len(b["sample_counts"]) >= 100 and min(b["sample_counts"]) >= 0 and max(b["sample_counts"]) <= n_q
            assert ("winner" in b) == has_winner
            if has_winner:
                assert b["winner"]["found"] and b["winner"]["best_count"] >= max(b["sample_counts"])


def test_the_generator_records_what_the_oracles_stage_calls_return(oracle_mod):
    M = _mod()
    O = oracle_mod
    delta, overlap, n_s = 0.01, 0.6, 220
    P, Q, _ = H.small_pair(20000, delta=delta, seed=41)
    om = O.Matcher(O.make_options(delta, overlap, n_s), full_counts=True, use_kdtree=True, keep_trace=False)
    om.init(P, Q)
    rec = M.record(om, delta, 2, True, 4, 200, lambda m: None)           # trial 2 of the seeded sequence, every 4th gated quad sampled
    # the same trial through the list forms on a second oracle matcher
    o2 = O.Matcher(O.make_options(delta, overlap, n_s), full_counts=True, use_kdtree=True, keep_trace=False)
    o2.init(P, Q)
    from bench import seg_len32
    eps = 2.0 * delta
    for _ in range(2):
        ok, _i1, _i2, _b, bx = o2.select_quadrilateral()
        if ok:
            o2.extract_pairs(seg_len32(bx[0], bx[1]), 0.0, eps, 0, 1)
            o2.extract_pairs(seg_len32(bx[2], bx[3]), 0.0, eps, 2, 3)
    ok, i1, i2, base, bx = o2.select_quadrilateral()
    assert ok and [int(v) for v in base] == rec["base"] and rec["trial"] == 2
    p1 = o2.extract_pairs(seg_len32(bx[0], bx[1]), 0.0, eps, 0, 1)
    p2 = o2.extract_pairs(seg_len32(bx[2], bx[3]), 0.0, eps, 2, 3)
    for p, r in ((p1, rec["pairs"][0]), (p2, rec["pairs"][1])):
        assert r["n"] == p.shape[0] and r["sha256"] == hashlib.sha256(np.ascontiguousarray(p, np.int32).tobytes()).hexdigest()
    quads = o2.find_congruent(i1, i2, eps, p1, p2)
    nb, per, bc, bi = o2.try_congruent_set(base, quads)
    assert rec["K"] == quads.shape[0] and rec["C"] == nb == int((per >= 0).sum())
    assert rec["K"] > 0 and nb > 0                          # (a base with candidates: the comparisons below are not vacuous)
    if nb:
        assert rec["winner"]["found"] and rec["winner"]["best_count"] == bc and rec["winner"]["best_quad"] == [int(v) for v in quads[bi]]
    # the sampled quads are gated quads of this base and their recorded counts are the list form's
    lookup = {tuple(int(v) for v in q): int(c) for q, c in zip(quads, per)}
    assert len(rec["sample_quads"]) > 0
    for q, c in zip(rec["sample_quads"], rec["sample_counts"]):
        assert lookup[tuple(q)] == c
