#!/usr/bin/env python
"""Throughput probe (lab aid): the bench workload through Perform_N_steps, one JSON line.  Several of these side by side on one
GPU tell whether the launch thread or the device bounds the pipeline.  Usage: tp_probe.py [steps] [tag]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from super4pcs_amd import capi, datasets   # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
tag = sys.argv[2] if len(sys.argv) > 2 else ""
P, Q, _ = datasets.bumpy_pair(1_000_000, overlap=0.5, delta=0.004, seed=20140814)
opt = capi.make_options(0.004, 0.5, 2000)
m = capi.Matcher(opt, device=0, max_pairs=8 << 20, max_quads=64 << 20)
if os.environ.get("TP_FULL") == "1":
    m.early_exit(False)
m.init_full(P, Q)
m.set_sharding(0, 1, 2)
m.perform_n_steps(int(os.environ.get("TP_WARMUP", "5")))
start_at = float(os.environ.get("TP_START_AT", "0"))
while time.time() < start_at:
    pass
res = []
for rep in range(3):
    m.profile_enable(False, False)
    m.profile_get(reset=True)
    i0 = m.info()
    t0 = time.perf_counter()
    m.perform_n_steps(steps)
    dt = time.perf_counter() - t0
    i1 = m.info()
    p = m.profile_get(reset=True)
    res.append({"us_per_base": round(dt / steps * 1e6, 2), "wait_us": round(p.host_wait_s / steps * 1e6, 2),
                "mcand_per_s": round((i1.candidates_verified - i0.candidates_verified) / dt / 1e6, 2), "t_end": time.time()})
i = m.info()
print(json.dumps({"tag": tag, "lanes": os.environ.get("S4P_LANES", "6"), "runs": res, "best_count": i.best_count, "cand": i.candidates_verified}))
m.close()
