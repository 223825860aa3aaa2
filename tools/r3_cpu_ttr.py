#!/usr/bin/env python
"""CPU time-to-register of the bench workload (BASELINE.md section 3 "Reported"; the reference's tests allow 600 s): the oracle's
whole ComputeTransformation on configs[2] (1 M-point pair, sample 2000) with its candidate loop under OpenMP on all host
cores (baseline B) and max_time_seconds = 600, inputs in memory.  One JSON line.  The 1-core reference-faithful run (baseline
A) needs ~1900 s at the sampled 4.1 k candidates/s, i.e. it is cut by the 600 s cap: not run.
Run from the repo root: python tools/r3_cpu_ttr.py > gpurun_out/r3_cpu_time_to_register.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402
from super4pcs_amd import datasets as D  # noqa: E402
import bench  # noqa: E402

P, Q, T_gt = D.bumpy_pair(bench.N_POINTS, overlap=bench.OVERLAP, delta=bench.DELTA, seed=bench.SEED)
om = O.Matcher(O.make_options(bench.DELTA, bench.OVERLAP, bench.SAMPLE, max_time_seconds=600), full_counts=False, use_kdtree=True)
om.set_threads(os.cpu_count() or 1)
t0 = time.perf_counter()
lcp, M, _ = om.compute_transformation(P, Q)
dt = time.perf_counter() - t0
s = om.stats()
print(json.dumps({"workload": "configs[2] 1 M-point pair, sample 2000", "kind": "port (oracle), candidate loop under OpenMP", "cores": os.cpu_count(),
                  "time_to_register_s": round(dt, 2), "cap_s": 600, "lcp": float(lcp), "trials_run": int(s.current_trial),
                  "candidates_verified": int(s.n_verified),
                  "stage_seconds": {"select": s.t_select, "pairs": s.t_pairs, "quads": s.t_quads, "verify": s.t_verify},
                  "rotation_error_vs_ground_truth": float(abs(M[:3, :3] - T_gt[:3, :3]).max())}))
