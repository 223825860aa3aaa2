#!/usr/bin/env python
"""Fixed cost of a short Perform_N_steps call on the bench workload (the driver times 20 steps = 2.7 ms): calls of 20 and of
200 steps on one matcher, with and without the selection pipeline threads; S4P_TRACE_CALL=1 makes the engine print where
each call's fixed cost goes (first base enqueued / first result / loop / rewind).  Run on a GPU box from the repo root:
S4P_TRACE_CALL=1 python tools/r3_short_calls.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
from super4pcs_amd import capi, datasets as D  # noqa: E402
import bench  # noqa: E402

P, Q, _ = D.bumpy_pair(bench.N_POINTS, overlap=bench.OVERLAP, delta=bench.DELTA, seed=bench.SEED)
for producer in (True, False):
    gm = capi.Matcher(capi.make_options(bench.DELTA, bench.OVERLAP, bench.SAMPLE), max_pairs=bench.MAX_PAIRS, max_quads=bench.MAX_QUADS)
    gm.init_full(P, Q)
    gm.set_sharding(0, 1, producer)
    gm.perform_n_steps(5)
    for n in (20, 20, 20, 20, 200, 20, 20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gm.perform_n_steps(n)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("producer_threads=%s n=%d: %.3f ms total, %.4f ms/step" % (producer, n, dt * 1e3, dt * 1e3 / n), flush=True)
    gm.close()
