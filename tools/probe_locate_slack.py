#!/usr/bin/env python
"""Adversarial check of the LCP structure's locating slack: queries quantised to 16 bit over a LONG bounding box (half a
quantisation step = 0.0039 cell, the largest the LDS path accepts), every query with exactly one P point at distance
delta * (1 - 1e-6) along -x.  A query that the quantised locate puts into the cell next to its true one must still
find that point.  Prints GPU counts against a brute-force float32 count for the given S4P_CELL_FACTOR values."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from super4pcs_amd import capi  # noqa: E402

F = np.float32
delta = 1.0
rng = np.random.default_rng(5)
n = 2000
h = 1.002 * delta
ext = 0.0078 * h * 65535.0                      # step = 0.0078 cell -> half step 0.0039 cell (< 0.004: the LDS path is taken)
Q = np.stack([rng.uniform(0, ext, n), rng.uniform(0, 3, n), rng.uniform(0, 3, n)], axis=1).astype(F)
Q[0, 0], Q[1, 0] = 0.0, ext                     # pin the bounding box
# transform k shifts the queries by (0.0137 k, 5 k, 0); its own copy of the partner points sits 5 k up in y, so every
# transform sees each query exactly delta * (1 - 1e-6) from one point, at a different offset to the cell faces
Ts, Ps = [], []
for k in range(64):
    T = np.eye(4, dtype=F)
    T[0, 3] = F(0.0137 * k); T[1, 3] = F(5.0 * k)
    Ts.append(T)
    Pk = Q.astype(np.float64).copy()
    Pk[:, 0] += float(T[0, 3]) - delta * (1 - 1e-6)
    Pk[:, 1] += float(T[1, 3])
    Ps.append(Pk.astype(F))
Ts = np.stack(Ts)
P = np.concatenate(Ps)


def brute(T):
    tx = ((T[0, 0] * Q[:, 0] + T[0, 1] * Q[:, 1]) + T[0, 2] * Q[:, 2]) + T[0, 3]
    ty = ((T[1, 0] * Q[:, 0] + T[1, 1] * Q[:, 1]) + T[1, 2] * Q[:, 2]) + T[1, 3]
    tz = ((T[2, 0] * Q[:, 0] + T[2, 1] * Q[:, 1]) + T[2, 2] * Q[:, 2]) + T[2, 3]
    near = P[np.abs(P[:, 1] - T[1, 3] - 1.5) < 4.0]          # only this transform's slab of P can be within delta
    cnt = 0
    for i in range(n):
        dx, dy, dz = tx[i] - near[:, 0], ty[i] - near[:, 1], tz[i] - near[:, 2]
        d2 = dx * dx + (dy * dy + dz * dz)
        cnt += bool((d2 <= F(delta) * F(delta)).any())
    return cnt


want = np.array([brute(T) for T in Ts], np.int64)
for cf in sys.argv[1:] or ["1.002", "1.02"]:
    os.environ["S4P_CELL_FACTOR"] = cf
    ctx = capi.Context(capi.make_options(delta, 0.5, n), max_pairs=1 << 16, max_quads=1 << 16)
    ctx.set_clouds(P, Q)
    got = ctx.verify_transforms(Ts).astype(np.int64)
    print(json.dumps({"cell_factor": cf, "transforms": len(Ts), "queries": n, "total_expected": int(want.sum()), "total_gpu": int(got.sum()),
                      "transforms_differing": int((got != want).sum()), "missing_inliers": int((want - got).sum())}), flush=True)
    ctx.close()
