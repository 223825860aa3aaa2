#!/usr/bin/env python
"""n = 20 000 on the bench clouds: trial 0 and 1 through TryOneBase inside a loop (early-exit bound in force), with the sweep's
candidate / survivor counts and the chunk statistics."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from super4pcs_amd import capi, datasets
P, Q, _ = datasets.bumpy_pair(bench.N_POINTS, overlap=bench.OVERLAP, delta=bench.DELTA, seed=bench.SEED)
m = capi.Matcher(capi.make_options(bench.DELTA, bench.OVERLAP, int(os.environ.get("SAMPLE", "20000"))), max_pairs=32 << 20, max_quads=32 << 20)
m.init_full(P, Q)
m.profile_enable(False, False)
m.loop_begin()
for t in range(int(os.environ.get("BASES", "2"))):
    t0 = time.perf_counter()
    ok, r = m.try_one_base()
    dt = time.perf_counter() - t0
    p = m.profile_get(reset=True)
    print(json.dumps({"trial": t, "s": round(dt, 2), "pairs": [r.n_pairs1, r.n_pairs2], "K": r.n_quads, "C": r.n_verified, "Mcand_per_s": round(r.n_verified / dt / 1e6, 2), "best": m.info().best_count,
                      "swept": p.sweep_candidates, "survivors": p.sweep_survivors, "pruned": p.verify_pruned, "chunks": m.chunk_stats()}))
m.loop_end()
