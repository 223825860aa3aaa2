#!/usr/bin/env python
"""One arm of an A/B run on the GPU box: the bench workload (configs[2]) through the library named by S4P_LIB.
Prints one JSON line: throughput, per-stage HIP-event times and a digest of the results (so that arms can be checked
against each other as well as timed).  Usage: S4P_LIB=<.so> [S4P_LANES=n] [S4P_FUSED=0|1] python tools/ab_one.py [steps]"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from super4pcs_amd import capi, datasets   # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
P, Q, _ = datasets.bumpy_pair(1_000_000, overlap=0.5, delta=0.004, seed=20140814)
opt = capi.make_options(0.004, 0.5, 2000)
res = []
dig = None
for rep in range(reps):
    m = capi.Matcher(opt, device=0, max_pairs=8 << 20, max_quads=64 << 20)
    m.init_full(P, Q)
    m.set_sharding(0, 1, True)
    m.perform_n_steps(5)
    m.profile_enable(True, False)
    m.profile_get(reset=True)
    c0 = m.info().candidates_verified
    t0 = time.perf_counter()
    m.perform_n_steps(steps)
    dt = time.perf_counter() - t0
    i = m.info()
    p = m.profile_get(reset=True)
    cand = i.candidates_verified - c0
    res.append({"cand_per_s": cand / dt, "ms_per_base": dt / steps * 1e3,
                "verify_ms": p.verify_ms_total / max(p.verify_launches, 1),
                "pairs_ms": p.pairs_ms_total / max(p.quads_launches, 1), "quads_ms": p.quads_ms_total / max(p.quads_launches, 1),
                "cand_per_launch": p.verify_candidates / max(p.verify_launches, 1)})
    h = hashlib.sha256()
    h.update(np.array([i.candidates_verified, i.quads_total, i.pairs_total, i.best_count], np.int64).tobytes())
    h.update(np.array(list(i.transform), np.float32).tobytes())
    h.update(np.array(list(i.base) + list(i.congruent), np.int32).tobytes())
    dig = h.hexdigest()[:16]
    m.close()
best = max(res, key=lambda r: r["cand_per_s"])
print(json.dumps({"lib": os.path.basename(os.environ.get("S4P_LIB", "default")), "lanes": os.environ.get("S4P_LANES", "3"),
                  "fused": os.environ.get("S4P_FUSED", "1"), "blocks": os.environ.get("S4P_VERIFY_BLOCKS", "512"), "steps": steps,
                  "best": {k: round(v, 4) for k, v in best.items()}, "all_cand_per_s": [round(r["cand_per_s"] / 1e6, 2) for r in res],
                  "digest": dig}))
