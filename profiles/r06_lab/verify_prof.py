#!/usr/bin/env python
"""Lab aid: per-wave timeline of k_verify (build variant -DS4P_JOIN_PROF, S4P_LIB=scratch/libjoinprof.so, S4P_LANES=1): where the
lifetime of a wave goes (staging / ticket loop / final barrier) and what the candidates cost (mean, longest, share above 8 / 20 us)."""
import ctypes as C
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("S4P_LANES", "1")
os.environ.setdefault("S4P_LIB", os.path.join(ROOT, "scratch", "libjoinprof.so"))
import bench                                         # noqa: E402
from super4pcs_amd import capi, datasets             # noqa: E402

P, Q, _ = datasets.bumpy_pair(bench.N_POINTS, overlap=bench.OVERLAP, delta=bench.DELTA, seed=bench.SEED)
m = capi.Matcher(capi.make_options(bench.DELTA, bench.OVERLAP, bench.SAMPLE), max_pairs=bench.MAX_PAIRS, max_quads=bench.MAX_QUADS)
m.init_full(P, Q)
L = m.L
L.s4p_debug_verify_prof.restype = C.c_int32
L.s4p_debug_verify_prof.argtypes = [C.c_void_p, C.c_int32]
NWORDS, NW = 12, 512 * 16
buf = np.zeros(NWORDS * NW, np.uint64)
m.loop_begin()
q = lambda a: [round(float(np.percentile(a, p)) / 100.0, 2) for p in (5, 50, 95, 100)]
for t in range(int(os.environ.get("BASES", "14"))):
    ok, r = m.try_one_base()
    L.s4p_debug_verify_prof(buf.ctypes.data_as(C.c_void_p), buf.size)
    s = buf.reshape(NW, NWORDS).astype(np.int64)
    s = s[s[:, 0] > 0]
    if t < 5 or len(s) == 0:
        continue
    t0 = s[:, 0].min()
    busy = s[s[:, 5] > 0]
    print("base %d: C %d best %d | waves %d (with candidates %d) | first start -> last barrier passed %.2f us" % (t, r.n_verified, m.info().best_count, len(s), len(busy), (s[:, 3].max() - t0) / 100.0))
    print("   staging %s ; ticket loop %s ; waiting in the final barrier %s (us p5/50/95/max)" % (q(s[:, 1] - s[:, 0]), q(s[:, 2] - s[:, 1]), q(s[:, 3] - s[:, 2])))
    tot = s[:, 9].sum() / 100.0
    print("   candidates per wave %s ; all candidates %.0f us of wave time = %.2f us each ; above 8 us: %d candidates, %.0f us (%.0f %%) ; above 20 us: %d ; longest per wave (us) %s" % (
        [int(np.percentile(busy[:, 5], p)) for p in (5, 50, 95, 100)], tot, tot / max(s[:, 5].sum(), 1), int(s[:, 7].sum()), s[:, 8].sum() / 100.0, 100.0 * s[:, 8].sum() / max(s[:, 9].sum(), 1), int(s[:, 10].sum()), q(busy[:, 6])))
    wg = {}
m.loop_end()
