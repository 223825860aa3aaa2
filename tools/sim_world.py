#!/usr/bin/env python
"""One rank of a simulated N-rank sharded job on ONE GPU (the collective is a stub: s4p_shard_use_null_collective): the rank
selects and stages every trial of every window and runs its own device passes through the C++ window loop
(s4p_shard_run_windows), exactly as a real rank does.  Prints, per workload and world size, the time per window and the
host-side split, i.e. whether the host chain or the GPU pass bounds a rank.  VERDICT r02 item 4 asks for: per-window host time
at world 8 <= 0.5 x GPU step at n_P = 57 k and at n_P = 4.2 M.
Run on a GPU box from the repo root: python tools/sim_world.py > gpurun_out/r3_sim_world.jsonl"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from super4pcs_amd import capi, datasets as D  # noqa: E402
import bench  # noqa: E402


def run(tag, P, Q, delta, overlap, n_s, worlds, windows, producer=True):
    for world in worlds:
        m = capi.Matcher(capi.make_options(delta, overlap, n_s), max_pairs=8 << 20, max_quads=64 << 20)
        m.init_full(P, Q)
        sh = capi.Shard(m, 0, world, producer)
        sh.use_null_collective()
        sh.run_windows(3)
        m.profile_enable(True, False); m.profile_get(reset=True)
        i0 = m.info()
        t0 = time.perf_counter()
        cand = sh.run_windows(windows)
        dt = time.perf_counter() - t0
        pr = m.profile_get(reset=True); i1 = m.info()
        trials = windows * world
        print(json.dumps({"workload": tag, "n_P": int(i1.n_sampled_p), "world": world, "windows": windows, "helper_threads": bool(producer),
                          "ms_per_window": round(dt / windows * 1e3, 4), "candidates_per_s_this_rank": round(cand / dt),
                          "host_select_us_per_trial": round((i1.seconds_select - i0.seconds_select) / trials * 1e6, 2),
                          "host_octree_us_per_trial": round(pr.host_octree_s / trials * 1e6, 2),
                          "launch_thread_wait_us_per_window": round(pr.host_wait_s / windows * 1e6, 2),
                          "k_verify_ms_per_launch": round(pr.verify_ms_total / max(pr.verify_launches, 1), 4),
                          "device_selection": m.device_selection(), "host_cores": os.cpu_count()}), flush=True)
        sh.close(); m.close()


if __name__ == "__main__":
    P, Q, _ = D.bumpy_pair(bench.N_POINTS, overlap=bench.OVERLAP, delta=bench.DELTA, seed=bench.SEED)
    if "--threads-ab" in sys.argv:                 # helper threads on / off at every world size, small n_P, three passes each
        for _ in range(3):
            for producer in (True, False):
                run("configs[2] 1 M-point pair (n_P = 57 k)", P, Q, bench.DELTA, bench.OVERLAP, bench.SAMPLE, (1, 2, 3, 4, 6, 8), 60, producer)
        sys.exit(0)
    if "--only-big" not in sys.argv:
        run("configs[2] 1 M-point pair (n_P = 57 k)", P, Q, bench.DELTA, bench.OVERLAP, bench.SAMPLE, (1, 2, 4, 8), 60)
    P, Q, _ = D.part_in_whole_pair(10_000_000, 100_000, delta=0.05)
    run("configs[4] 10 M-point scene (n_P = 4.2 M)", P, Q, 0.05, 0.2, 2000, (1, 8), 40)
