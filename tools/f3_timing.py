#!/usr/bin/env python
"""Host time per base of SelectQuadrilateral with its two searches on the host structures (mode 0) and as device
reductions (mode 1), at BASELINE configs[4]'s sampled scene (n_P = 4.2 M) and at the bench workload (n_P = 57 k).
Prints one JSON line per (workload, mode).  Run on a GPU box from the repo root."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from super4pcs_amd import capi, datasets as D  # noqa: E402
import bench  # noqa: E402


def run(tag, P, Q, delta, overlap, n_s, n_bases):
    ref = None
    for mode in (0, 1):
        gm = capi.Matcher(capi.make_options(delta, overlap, n_s), max_pairs=8 << 20, max_quads=64 << 20)
        gm.set_device_selection(mode)
        t0 = time.perf_counter()
        gm.init_full(P, Q)
        t_init = time.perf_counter() - t0
        for _ in range(3):
            gm.select_quadrilateral()
        t0 = time.perf_counter()
        seq = []
        for _ in range(n_bases):
            ok, i1, i2, base, _bx = gm.select_quadrilateral()
            seq.append((ok, tuple(int(b) for b in base)))
        dt = time.perf_counter() - t0
        if ref is None:
            ref = seq
        print(json.dumps({"workload": tag, "n_P": int(gm.info().n_sampled_p), "mode": "device" if mode else "host",
                          "init_s": round(t_init, 4), "us_per_base": round(1e6 * dt / n_bases, 2), "bases": n_bases,
                          "same_bases_as_host": seq == ref}), flush=True)
        del gm


if __name__ == "__main__":
    P, Q, _ = D.bumpy_pair(bench.N_POINTS, overlap=bench.OVERLAP, delta=bench.DELTA, seed=bench.SEED)
    run("configs[2] bench pair", P, Q, bench.DELTA, bench.OVERLAP, bench.SAMPLE, 200)
    P, Q, _ = D.part_in_whole_pair(10_000_000, 100_000, delta=0.05)
    run("configs[4] 10 M-point scene", P, Q, 0.05, 0.2, 2000, 100)
