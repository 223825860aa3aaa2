#!/usr/bin/env python
"""Offline study for the next kernel step (no GPU): how much of k_verify's sweep a CLUSTER pre-pass would remove.

Today every candidate transform sweeps every sampled query point: transform, coarse cell, one bit of the LDS bitmap
(~800 vector instructions per candidate at n_Q = 2000 after the early exit).  Idea: group the sampled Q into spatial clusters
(centre m_c, radius r_c) once per registration; per candidate, transform the centres only and look them up in a coarse lower
bound of the distance to P (a byte per cell): if dist(T m_c, P) > r_c + eps for sure, none of the cluster's points can be an
inlier and the whole cluster is skipped -- conservative, so inlier counts stay exact.

Measured here on the bench workload (configs[2]: 1 M-point pair, n_P = 57 207, n_Q = 2000) and on the 20 000-point sample, with
the real candidate transforms of the first bases of the seeded sequence (oracle): share of (candidate, query) pairs left after
the cluster pass, for several cluster sizes and field resolutions, and the share of candidates whose surviving queries cannot
exceed a given best count (they would be dismissed after n_clusters tests instead of n_Q).
usage: python tools/lab/r5/cluster_cull_study.py [sample_size] [n_bases] [larger_sample_for_the_clusters]"""
import json
import os
import sys
import time

import numpy as np
from scipy.spatial import cKDTree

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import oracle as O  # noqa: E402
from super4pcs_amd import datasets as D  # noqa: E402


def clusters_of(Q, size):
    """recursive median split along the widest axis until a node holds <= size points"""
    out = []
    stack = [np.arange(len(Q))]
    while stack:
        idx = stack.pop()
        if len(idx) <= size:
            out.append(idx)
            continue
        pts = Q[idx]
        ax = int(np.argmax(pts.max(0) - pts.min(0)))
        order = idx[np.argsort(pts[:, ax], kind="stable")]
        h = len(order) // 2
        stack.append(order[:h]); stack.append(order[h:])
    return out


def distance_field(P, cell, margin):
    lo = P.min(0) - margin
    hi = P.max(0) + margin
    n = np.maximum(1, np.ceil((hi - lo) / cell).astype(int))
    gx, gy, gz = [lo[k] + (np.arange(n[k]) + 0.5) * cell for k in range(3)]
    C = np.stack(np.meshgrid(gx, gy, gz, indexing="ij"), -1).reshape(-1, 3)
    d, _ = cKDTree(P).query(C)
    lower = np.maximum(0.0, d - 0.5 * np.sqrt(3.0) * cell)          # valid for every point of the cell
    return lo, n, lower.reshape(n)


def main():
    n_s = int(sys.argv[1]) if len(sys.argv) > 1 else bench.SAMPLE
    n_bases = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    P, Q, _ = D.bumpy_pair(bench.N_POINTS, overlap=bench.OVERLAP, delta=bench.DELTA, seed=bench.SEED)
    om = O.Matcher(O.make_options(bench.DELTA, bench.OVERLAP, n_s), full_counts=True, use_kdtree=True, keep_trace=False)
    om.init(P, Q)
    Ps, Qs = om.cloud(0).astype(np.float64), om.cloud(1).astype(np.float64)
    eps = float(bench.DELTA)                                         # Verify's radius (match4pcsBase.cc:512-516)
    # candidate transforms of the first bases that have any
    Ts, counts = [], []
    tried = 0
    while len(Ts) < n_bases and tried < 60:
        tried += 1
        ok, i1, i2, base, bx = om.select_quadrilateral()
        if not ok:
            continue
        d1, d2 = bench.seg_len32(bx[0], bx[1]), bench.seg_len32(bx[2], bx[3])
        p1 = om.extract_pairs(d1, 0.0, 2 * bench.DELTA, 0, 1); p2 = om.extract_pairs(d2, 0.0, 2 * bench.DELTA, 2, 3)
        if not len(p1) or not len(p2):
            continue
        quads = om.find_congruent(i1, i2, 2 * bench.DELTA, p1, p2, cap=1 << 23)
        if not len(quads):
            continue
        sel = quads if len(quads) <= 4000 else quads[np.linspace(0, len(quads) - 1, 4000).astype(int)]
        T = []
        for q in sel:
            good, _rms, M = om.compute_rigid(base, q)
            if good:
                T.append(M.astype(np.float64))
        if T:
            T = np.stack(T)
            Ts.append(T); counts.append(om.verify_batch(T.astype(np.float32)))
    T = np.concatenate(Ts); cnt = np.concatenate(counts).astype(np.int64)
    check = True
    if len(sys.argv) > 3:                                            # clusters of a LARGER sample of the same Q under the same transforms
        om2 = O.Matcher(O.make_options(bench.DELTA, bench.OVERLAP, int(sys.argv[3])), full_counts=True, use_kdtree=True, keep_trace=False)
        om2.init(P, Q)
        shift = np.array(om2.frame()[1], np.float64) - np.array(om.frame()[1], np.float64)      # centroid of the other Q sample
        Qs = om2.cloud(1).astype(np.float64) + shift                  # into the frame the transforms were computed in
        cnt = np.zeros(len(T), np.int64); check = False
    print(json.dumps({"n_P": len(Ps), "n_Q": len(Qs), "bases": len(Ts), "candidates": len(T), "inliers_mean": float(cnt.mean()),
                      "inliers_p50_p90_p99_max": [int(np.percentile(cnt, p)) for p in (50, 90, 99, 100)]}), flush=True)
    extent = float((Ps.max(0) - Ps.min(0)).max())
    for size in (8, 16, 32, 64):
        cl = clusters_of(Qs, size)
        centre = np.stack([Qs[c].mean(0) for c in cl])
        radius = np.array([np.linalg.norm(Qs[c] - m, axis=1).max() for c, m in zip(cl, centre)])
        members = np.array([len(c) for c in cl])
        for cells in (32, 48, 64):
            cell = extent / cells
            lo, n, field = distance_field(Ps, cell, margin=2 * cell)
            t0 = time.time()
            Cc = np.einsum("kij,cj->kci", T[:, :3, :3], centre) + T[:, None, :3, 3]          # (K, clusters, 3)
            ijk = np.floor((Cc - lo) / cell).astype(int)
            inside = np.all((ijk >= 0) & (ijk < n), axis=2)
            ijk = np.clip(ijk, 0, n - 1)
            lower = np.where(inside, field[ijk[..., 0], ijk[..., 1], ijk[..., 2]], np.inf)   # outside the box: further than the margin... treated as far
            keep = lower <= radius[None, :] + eps
            left_queries = (keep * members[None, :]).sum(1)                                   # per candidate
            share = float(left_queries.sum() / (len(T) * len(Qs)))
            row = {"cluster_size": size, "clusters": len(cl), "radius_mean_over_extent": float(radius.mean() / extent), "field_cells": int(cells),
                   "field_bytes": int(np.prod(n)), "queries_left_share": round(share, 4),
                   "candidates_with_all_clusters_culled": round(float((left_queries == 0).mean()), 4)}
            for best in (int(0.02 * len(Qs)), int(0.1 * len(Qs)), int(0.2 * len(Qs))):
                row["dismissed_after_cluster_pass_if_best_%d" % best] = round(float((left_queries <= best).mean()), 4)
            # sanity: culling never removes an inlier-bearing cluster beyond what the counts allow
            assert not check or np.all(left_queries >= cnt), "a culled cluster held inliers"
            row["seconds"] = round(time.time() - t0, 1)
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
