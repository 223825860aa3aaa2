#!/usr/bin/env python
"""Where does the launch thread's time go?  N bases of the bench workload through Perform_N_steps; prints the wall time per
base next to the host's own timers: blocked in stream synchronisation (= waiting for the GPU), pair-octree builds, base
selection.  A small wait share means the HOST bounds the pipeline, not the device."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from super4pcs_amd import capi, datasets   # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
P, Q, _ = datasets.bumpy_pair(1_000_000, overlap=0.5, delta=0.004, seed=20140814)
opt = capi.make_options(0.004, 0.5, 2000)
out = []
for threads in (0, 1):
    m = capi.Matcher(opt, device=0, max_pairs=8 << 20, max_quads=64 << 20)
    m.init_full(P, Q)
    m.set_sharding(0, 1, bool(threads))
    m.perform_n_steps(5)
    m.profile_enable(False, False)
    m.profile_get(reset=True)
    i0 = m.info()
    t0 = time.perf_counter()
    m.perform_n_steps(steps)
    dt = time.perf_counter() - t0
    i1 = m.info()
    p = m.profile_get(reset=True)
    out.append({"helper_threads": threads, "us_per_base": dt / steps * 1e6, "wait_us_per_base": p.host_wait_s / steps * 1e6,
                "octree_us_per_base": p.host_octree_s / steps * 1e6, "select_us_per_base": (i1.seconds_select - i0.seconds_select) / steps * 1e6,
                "cand_per_s": (i1.candidates_verified - i0.candidates_verified) / dt})
    m.close()
print(json.dumps(out))
