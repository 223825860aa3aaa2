#!/usr/bin/env python
"""configs[4] (100 k query in a 10 M scene) at SURVEY 8d's sample of 5000: one ComputeTransformation under a time cap (default 60 s)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from super4pcs_amd import capi, datasets as D
cap = int(os.environ.get("CAP_S", "60"))
P, Q, T = D.part_in_whole_pair(10_000_000, 100_000, delta=0.05)
gm = capi.Matcher(capi.make_options(0.05, 0.2, 5000, max_time_seconds=cap))
t0 = time.perf_counter()
lcp, M, _ = gm.compute_transformation(P, Q)
dt = time.perf_counter() - t0
i = gm.info()
print(json.dumps({"ttr_s": round(dt, 2), "trials": int(i.bases_tried), "of": int(i.number_of_trials), "candidates": int(i.candidates_verified), "Mcand_per_s": round(i.candidates_verified / dt / 1e6, 2), "lcp": float(lcp), "k_verify": gm.verify_kernel_info()}))
