#!/usr/bin/env python
"""Initialisation time and time-to-register of the BASELINE configs at the sample sizes SURVEY.md 8d states for them: per
workload one JSON line with init_full wall time (sampler + engine init + s4p_set_clouds, split into its phases), the wall
time of one whole ComputeTransformation with inputs in host memory, and the recovered pose against the generator's.
Run on a GPU box from the repo root: python tools/init_timing.py > gpurun_out/r04_init_and_time_to_register.jsonl"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from super4pcs_amd import capi, datasets as D  # noqa: E402
import bench  # noqa: E402


def run(tag, P, Q, T_gt, delta, overlap, n_s, register=True, max_time_seconds=10 ** 6):
    out = {"workload": tag, "n_points_P": int(P.shape[0]), "n_points_Q": int(Q.shape[0]), "sample_size": n_s}
    opt = capi.make_options(delta, overlap, n_s, max_time_seconds=max_time_seconds)
    for rep in range(2):                                   # second pass: allocator and page cache warm
        gm = capi.Matcher(opt)
        t0 = time.perf_counter()
        gm.init_full(P, Q)
        out["init_full_s_pass%d" % rep] = round(time.perf_counter() - t0, 4)
        out["set_clouds_pass%d" % rep] = {k: round(v, 5) for k, v in gm.set_clouds_timing().items()}
        i = gm.info()
        out["n_P"], out["n_Q"], out["trials"] = int(i.n_sampled_p), int(i.n_sampled_q), int(i.number_of_trials)
        out["k_verify"] = gm.verify_kernel_info()
        gm.close()
    if register:
        gm = capi.Matcher(opt)
        t0 = time.perf_counter()
        lcp, M, _ = gm.compute_transformation(P, Q)
        out["time_to_register_s"] = round(time.perf_counter() - t0, 4)
        i = gm.info()
        out["lcp"] = float(lcp); out["trials_run"] = int(i.bases_tried); out["candidates_verified"] = int(i.candidates_verified)
        out["rotation_error_vs_ground_truth"] = float(np.max(np.abs(M[:3, :3] - T_gt[:3, :3])))
        out["translation_error_vs_ground_truth"] = float(np.max(np.abs(M[:3, 3] - T_gt[:3, 3])))
        out["chunk_stats"] = gm.chunk_stats(); out["lane_growths"] = gm.capacity_growths()
        if max_time_seconds < 10 ** 6:
            out["max_time_seconds"] = max_time_seconds
        gm.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["1", "2", "3", "4"]
    if "1" in which:                                       # configs[1]: the bunny-like partial-scan pair, n = 1000
        P, Q, T = D.bumpy_pair(40000, overlap=0.45, delta=0.008, noise_sigma=0.3 * 0.008, seed=31)      # the pair of tests/test_gpu_configs.py::test_config1_*
        run("configs[1] 40 k-point partial-scan pair, sample 350 (the size the oracle replays in the GPU test)", P, Q, T, 0.008, 0.45, 350)
        run("configs[1] 40 k-point partial-scan pair, sample 1000 (SURVEY 8d)", P, Q, T, 0.008, 0.45, 1000)
    if "2" in which:
        P, Q, T = D.bumpy_pair(bench.N_POINTS, overlap=bench.OVERLAP, delta=bench.DELTA, seed=bench.SEED)
        run("configs[2] 1 M-point pair", P, Q, T, bench.DELTA, bench.OVERLAP, bench.SAMPLE)
    if "3" in which:
        P, Q, T = D.lidar_pair(5_000_000, delta=0.05)
        run("configs[3] 5 M-point LiDAR pair, sample 20 000 (SURVEY 8d)", P, Q, T, 0.05, 0.4, 20000, max_time_seconds=90)
    if "4" in which or "4s" in which:                      # "4s": the sample of 2000 only (the 5000 leg runs into its 60-second cap)
        P, Q, T = D.part_in_whole_pair(10_000_000, 100_000, delta=0.05)
        run("configs[4] 100 k query in 10 M scene, sample 2000", P, Q, T, 0.05, 0.2, 2000, max_time_seconds=60)
        if "4" in which:
            run("configs[4] 100 k query in 10 M scene, sample 5000 (SURVEY 8d)", P, Q, T, 0.05, 0.2, 5000, max_time_seconds=60)
