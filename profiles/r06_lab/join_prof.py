#!/usr/bin/env python
"""Lab aid: per-phase cycle stamps of k_quads (build variant -DS4P_JOIN_PROF, S4P_LIB=scratch/libjoinprof.so, S4P_LANES=1).
Runs the bench workload's first bases one at a time and prints the mean cycles between stamps over the waves that did work."""
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
L.s4p_debug_join_prof.restype = C.c_int32
L.s4p_debug_join_prof.argtypes = [C.c_void_p, C.c_int32]
NS, NW = 12, 256 * 8
buf = np.zeros(NS * NW, np.uint64)
rows = []
for t in range(12):
    found, base, r = m.next_base(True)
    if not found:
        continue
    L.s4p_debug_join_prof(buf.ctypes.data_as(C.c_void_p), buf.size)
    pb = np.zeros(8 * 256 * 8, np.uint64)
    L.s4p_debug_pairs_prof(pb.ctypes.data_as(C.c_void_p), pb.size)
    if t >= 4:
        z = pb.reshape(-1, 8).astype(np.int64)
        z = z[(z[:, :5] > 0).all(axis=1)]
        qq = lambda a: [round(float(np.percentile(a, p)) / 100.0, 2) for p in (5, 50, 95, 100)]
        print("k_pairs2: waves %d | start spread %.2f us, first start -> last end %.2f us" % (len(z), (z[:, 0].max() - z[:, 0].min()) / 100.0, (z[:, 4].max() - z[:, 0].min()) / 100.0))
        for a_, b_, nm in [(0, 1, "entry -> chunk gathered"), (1, 5, "item loop (box + pre-tests, batches)"), (5, 2, "last batches"), (2, 3, "barrier + counter atomic"), (3, 4, "write_out (pairs + join records)")]:
            print("   %-36s us p5/50/95/max %s" % (nm, qq(z[:, b_] - z[:, a_])))
    s = buf.reshape(NW, NS).astype(np.int64)
    if t < 4:
        continue                                     # the first bases go through k_bin (no estimate yet)
    started = s[:, 0] > 0
    full = (s[:, :9] > 0).all(axis=1)
    t0 = s[started][:, 9].min()
    q = lambda a: [round(float(np.percentile(a, p)) / 100.0, 2) for p in (5, 50, 95, 100)]
    print("m1 %d m2 %d K %d C %d | waves started %d, through all phases %d" % (r.n_pairs1, r.n_pairs2, r.n_quads, r.n_verified, int(started.sum()), int(full.sum())))
    print("   us: start spread %.2f ; first start -> last end %.2f ; lifetime p5/50/95/max %s" % ((s[started][:, 9].max() - t0) / 100.0, (s[started][:, 10].max() - t0) / 100.0, q(s[started][:, 10] - s[started][:, 9])))
    z = s[full]
    order = [(9, 0, "entry"), (0, 1, "init (cone table, barrier)"), (1, 8, "-> entering the (last) bucket"), (8, 2, "cursor loads"), (2, 3, "T1 load + sort"), (3, 4, "T2 lookup + setup"), (4, 5, "M masks"), (5, 6, "X items"), (6, 11, "barrier wait"), (11, 7, "joint flush")]
    for a_, b_, nm in order:
        print("   %-28s us p5/50/95/max %s" % (nm, q(z[:, b_] - z[:, a_])))
