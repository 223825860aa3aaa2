#!/usr/bin/env python
"""Wall time of one s4p_select_base_points_batch call (upload of the draws, k_select_triangle, k_select_fourth, k_select_finish,
read-back, synchronise) on an idle GPU, per batch size, at n_P = 4.2 M (configs[4])."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from super4pcs_amd import capi, datasets as D  # noqa: E402


def probe(tag, P, Q, delta, overlap, n_s):
    m = capi.Matcher(capi.make_options(delta, overlap, n_s), max_pairs=8 << 20, max_quads=64 << 20)
    m.init_full(P, Q)
    info = m.info()
    n_p = int(info.n_sampled_p)
    diameter = np.float32(info.p_diameter)
    limit = float(diameter * diameter)
    too_small = float(np.float32(float(diameter * np.float32(0.2)) ** 2))
    rng = np.random.default_rng(1)
    out = {"workload": tag, "n_P": n_p}
    L, h = m.L, m.ctx_handle()
    for nb in (1, 4, 8, 16):
        draws = np.ascontiguousarray(rng.integers(0, n_p, (nb, 2001)).astype(np.uint32))
        ids = np.empty((nb, 4), np.int32); xyz = np.empty((nb, 12), np.float32); st = np.empty(nb, np.int32)

        def call():
            rc = L.s4p_select_base_points_batch(h, draws.ctypes.data_as(C.POINTER(C.c_uint32)), nb, limit, too_small,
                                                ids.ctypes.data_as(C.POINTER(C.c_int32)), xyz.ctypes.data_as(C.POINTER(C.c_float)),
                                                st.ctypes.data_as(C.POINTER(C.c_int32)))
            assert rc == 0, rc
        for _ in range(3):
            call()
        reps = 30
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        out["batch_%d_us" % nb] = round((time.perf_counter() - t0) / reps * 1e6, 1)
        out["batch_%d_status" % nb] = sorted(set(int(s) for s in st))
    print(json.dumps(out), flush=True)
    m.close()


if __name__ == "__main__":
    P, Q, _ = D.part_in_whole_pair(10_000_000, 100_000, delta=0.05)
    probe("configs[4] 10 M-point scene", P, Q, 0.05, 0.2, 2000)
