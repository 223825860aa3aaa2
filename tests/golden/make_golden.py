"""Generates tests/golden/*.json with the CPU oracle (run from the repo root):

    python tests/golden/make_golden.py

The reference ships no golden vectors for this path; these fixtures freeze the
oracle's answers on small seeded inputs so that (a) the oracle itself cannot drift silently and
(b) the GPU path is checked against committed numbers, not only against a live oracle run.
When /root/reference is present the sampler counts of its bundled hippo assets are recorded too
(the 5281 figure is the one number the reference publishes: doc/Usage.md:80).
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from tests import helpers as H  # noqa: E402


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load_obj_vertices(path):
    v = []
    with open(path) as f:
        for line in f:
            if line.startswith("v "):
                v.append([float(x) for x in line.split()[1:4]])
    return np.array(v, np.float32)


def main():
    out = {}
    # ---- registration trace on a small synthetic pair -------------------------------------------
    delta, overlap, n_s, seed = 0.01, 0.6, 200, 5489
    P, Q, T = H.small_pair(20000, delta=delta, seed=31)
    m = O.Matcher(O.make_options(delta, overlap, n_s, seed=seed), full_counts=False, use_kdtree=True, keep_trace=True)
    lcp, M, Qt = m.compute_transformation(P, Q)
    tr, inv = m.trace()
    s = m.stats()
    out["registration"] = {
        "input": {"generator": "tests.helpers.small_pair(20000, delta=0.01, seed=31)", "delta": delta, "overlap": overlap,
                  "sample_size": n_s, "seed": seed, "P_sha256": digest(P), "Q_sha256": digest(Q)},
        "n_P": s.n_P, "n_Q": s.n_Q, "number_of_trials": s.number_of_trials,
        "lcp": float(lcp), "best_count": int(round(lcp * s.n_Q)), "M": [float(x) for x in M.reshape(-1)],
        "candidates_verified": int(s.n_verified), "quads": int(s.n_quads), "pairs": int(s.n_pairs),
        "trace_first_20": tr[:20].tolist(), "trace_sha256": digest(tr), "Qt_sha256": digest(Qt),
    }
    # ---- stage vectors: one base, pairs / quads / per-candidate counts ---------------------------
    m2 = H.init_oracle(O, P, Q, delta, overlap, n_s, seed=seed)
    stage = None
    for _ in range(30):
        ok, i1, i2, base, bx = m2.select_quadrilateral()
        if not ok:
            continue
        eps = 2.0 * delta
        d1 = float(np.float32(np.linalg.norm(bx[0] - bx[1]))); d2 = float(np.float32(np.linalg.norm(bx[2] - bx[3])))
        p1 = m2.extract_pairs(d1, 0.0, eps, 0, 1); p2 = m2.extract_pairs(d2, 0.0, eps, 2, 3)
        if len(p1) == 0 or len(p2) == 0:
            continue
        quads = m2.find_congruent(i1, i2, eps, p1, p2)
        if len(quads) < 20:
            continue
        nb, per, bc, bi = m2.try_congruent_set(base, quads)
        if nb < 5:
            continue
        stage = {"base": base.tolist(), "inv": [float(np.float32(i1)), float(np.float32(i2))], "d": [d1, d2],
                 "n_pairs": [int(len(p1)), int(len(p2))], "pairs1_sha256": digest(p1), "pairs2_sha256": digest(p2),
                 "pairs1_head": p1[:8].tolist(), "n_quads": int(len(quads)), "quads_sha256": digest(quads),
                 "quads_head": quads[:8].tolist(), "n_verified": int(nb), "counts_sha256": digest(per),
                 "counts_head": per[:32].tolist(), "max_count": int(per.max()), "ids_sha256": digest(m2.ids())}
        break
    out["stage"] = stage
    # ---- trial-count known answers (match4pcsBase.hpp:175-185) ----------------------
    kat = {}
    for ov in (0.7, 0.5, 0.8, 0.2):
        mm = O.Matcher(O.make_options(0.05, ov, 50))
        mm.init(P[:500], Q[:500])
        kat[str(ov)] = mm.stats().number_of_trials
    out["number_of_trials"] = kat
    # ---- sampler counts on the reference's bundled assets ----------------------------------------
    ref = "/root/reference/assets"
    if os.path.isdir(ref):
        h1, h2 = load_obj_vertices(ref + "/hippo1.obj"), load_obj_vertices(ref + "/hippo2.obj")
        out["hippo_sampler"] = {"n_vertices": [int(len(h1)), int(len(h2))],
                                "delta_0.01": [int(len(O.sample(h1, 0.01))), int(len(O.sample(h2, 0.01)))],
                                "delta_0.005": [int(len(O.sample(h1, 0.005))), int(len(O.sample(h2, 0.005)))],
                                "source": "assets/hippo{1,2}.obj, doc/Usage.md:80 publishes 5281 for hippo1 @ 0.01"}
    else:
        prev = os.path.join(os.path.dirname(__file__), "oracle_golden.json")
        if os.path.exists(prev):
            out["hippo_sampler"] = json.load(open(prev)).get("hippo_sampler")
    with open(os.path.join(os.path.dirname(__file__), "oracle_golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote oracle_golden.json:", {k: (v if not isinstance(v, dict) else "...") for k, v in out.items() if k != "registration"})


if __name__ == "__main__":
    main()
