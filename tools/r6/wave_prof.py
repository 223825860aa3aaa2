#!/usr/bin/env python
"""Lab aid: per-wave timelines of k_pairs2 / k_quads / k_verify from the -DS4P_PROF build (scratch/libprof.so), one base in
flight (S4P_LANES=1): REFCLK stamps kept in registers and written at wave end.  Prints, per base, where a wave's lifetime goes."""
import ctypes as C
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("S4P_LANES", "1")
os.environ.setdefault("S4P_LIB", os.path.join(ROOT, "scratch", "libprof.so"))
import bench                                         # noqa: E402
from super4pcs_amd import capi, datasets             # noqa: E402

P, Q, _ = datasets.bumpy_pair(bench.N_POINTS, overlap=bench.OVERLAP, delta=bench.DELTA, seed=bench.SEED)
m = capi.Matcher(capi.make_options(bench.DELTA, bench.OVERLAP, bench.SAMPLE), max_pairs=bench.MAX_PAIRS, max_quads=bench.MAX_QUADS)
m.init_full(P, Q)
L = m.L
L.s4p_debug_prof.restype = C.c_int32
L.s4p_debug_prof.argtypes = [C.c_int32, C.c_void_p, C.c_int32]
NWORDS, NW = 12, 8192
q = lambda a: [round(float(np.percentile(a, p)) / 100.0, 2) for p in (5, 50, 95, 100)]


def grab(which):
    buf = np.zeros(NWORDS * NW, np.uint64)
    L.s4p_debug_prof(which, buf.ctypes.data_as(C.c_void_p), buf.size)
    s = buf.reshape(NW, NWORDS).astype(np.int64)
    return s[s[:, 0] > 0]


m.loop_begin()
for t in range(int(os.environ.get("BASES", "12"))):
    ok, r = m.try_one_base()
    sp, sq, sv = grab(0), grab(1), grab(2)
    lean = grab(3).sum(axis=0).astype(np.float64)
    if t < 5:
        continue                                     # (the first bases still run k_prep; the bound is not in force yet)
    print("base %d: m1 %d m2 %d K %d C %d best %d" % (t, r.n_pairs1, r.n_pairs2, r.n_quads, r.n_verified, m.info().best_count))
    if len(sp):
        z = sp[(sp[:, :5] > 0).all(axis=1)]
        print("  k_pairs2: %d waves, first start -> last end %.2f us (start spread %.2f)" % (len(sp), (sp[:, 4].max() - sp[:, 0].min()) / 100.0, (sp[:, 0].max() - sp[:, 0].min()) / 100.0))
        if len(z):
            for a_, b_, nm in [(0, 1, "entry -> chunk gathered"), (1, 2, "item loop + batches"), (2, 3, "barrier + counter atomic"), (3, 4, "write_out (+ set-1 preparation)")]:
                print("     %-34s us p5/50/95/max %s" % (nm, q(z[:, b_] - z[:, a_])))
            print("     staged entries per wave %s" % [int(np.percentile(z[:, 5], p)) for p in (5, 50, 95, 100)])
    if len(sq):
        print("  k_quads: %d waves, first start -> last end %.2f us; wave lifetime us %s" % (len(sq), (sq[:, 1].max() - sq[:, 0].min()) / 100.0, q(sq[:, 1] - sq[:, 0])))
        w = sq[sq[:, 6] > 0]
        if len(w):
            for k, nm in [(2, "A hash lookup + compaction"), (3, "B world point + cone mask"), (4, "C chain walk"), (5, "flush (atomics + gate)")]:
                print("     %-34s us p5/50/95/max %s" % (nm, q(w[:, k])))
            print("     tiles per workgroup %s ; longest chain walked in a wave (hops) %s ; quads flushed per wave-tile %s" % (
                [int(np.percentile(w[:, 6], p)) for p in (5, 50, 100)], [int(np.percentile(w[:, 7], p)) for p in (5, 50, 95, 100)], [int(np.percentile(w[:, 8], p)) for p in (5, 50, 95, 100)]))
    if len(sv):
        t0 = sv[:, 0].min()
        busy = sv[sv[:, 5] > 0]
        print("  k_verify: %d waves (%d with candidates), first start -> last barrier passed %.2f us" % (len(sv), len(busy), (sv[:, 3].max() - t0) / 100.0))
        print("     staging %s ; ticket loop %s ; waiting in the final barrier %s" % (q(sv[:, 1] - sv[:, 0]), q(sv[:, 2] - sv[:, 1]), q(sv[:, 3] - sv[:, 2])))
        if len(busy):
            tot = sv[:, 9].sum() / 100.0
            print("     ticket, record fetch and bookkeeping per candidate %.2f us (whole iteration minus the lean sweep's own time)%s" % (
                (sv[:, 9].sum() - (lean[8] if lean is not None else 0)) / 100.0 / max(sv[:, 5].sum(), 1),
                " ; of that the record's round trip %.2f us" % (sv[:, 11].sum() / 100.0 / max(sv[:, 5].sum(), 1)) if sv[:, 11].sum() else ""))
            print("     candidates per wave %s ; %.2f us per candidate ; above 8 us: %d (%.0f %% of the candidate time) ; above 20 us: %d ; longest per wave (us) %s" % (
                [int(np.percentile(busy[:, 5], p)) for p in (5, 50, 95, 100)], tot / max(sv[:, 5].sum(), 1), int(sv[:, 7].sum()), 100.0 * sv[:, 8].sum() / max(sv[:, 9].sum(), 1), int(sv[:, 10].sum()), q(busy[:, 6])))
    if t >= 5 and lean is not None and lean.size >= 12 and lean[0]:
        n = float(lean[0]); us = lambda v: float(v) / 100.0
        print("     lean sweep, per candidate over %d: %.2f us in all | sweep %.2f us (%.1f steps) | alive after the sweep %.3f: queue fill %.2f us, drain %.2f us, exact batches %.2f us "
              "(%.2f batches; %.3f of the candidates reach one; counted in full %.4f; mean count %.1f)" % (
                  int(n), us(lean[8]) / n, us(lean[1]) / n, float(lean[2]) / n, float(lean[3]) / n, us(lean[4]) / max(float(lean[3]), 1), us(lean[5]) / max(float(lean[3]), 1),
                  us(lean[6]) / max(float(lean[9]), 1), float(lean[7]) / max(float(lean[9]), 1), float(lean[9]) / n, float(lean[10]) / n, float(lean[11]) / n))
m.loop_end()
