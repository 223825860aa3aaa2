#!/usr/bin/env python
"""Copies the summaries of the round-4 final pass (tools/r4/final_pass.sh -> gpurun_out/r04_final/) into profiles/ (tracked)."""
import glob
import os
import shutil

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out", "r04_final"), os.path.join(R, "profiles")


def json_line(src, dst):
    path = os.path.join(G, src)
    if not os.path.exists(path):
        print("missing", src)
        return
    lines = [l for l in open(path).read().splitlines() if l.startswith('{"metric')]
    if not lines:
        print("no JSON line in", src)
        return
    open(os.path.join(P, dst), "w").write(lines[-1] + "\n")
    print(dst)


def copy(src, dst):
    path = os.path.join(G, src)
    if os.path.exists(path) and os.path.getsize(path):
        shutil.copy(path, os.path.join(P, dst))
        print(dst)
    else:
        print("missing", src)


json_line("bench_final.json", "r04_bench_final.json")
json_line("bench_driver_command.json", "r04_bench_driver_command.json")
json_line("bench_under_rocprof.json", "r04_bench_under_rocprof_final.json")
os.makedirs(os.path.join(P, "r04_bench_final"), exist_ok=True)
for f in glob.glob(os.path.join(G, "bench_final", "*.csv")):
    shutil.copy(f, os.path.join(P, "r04_bench_final", os.path.basename(f)))
    print("r04_bench_final/" + os.path.basename(f))
for f in glob.glob(os.path.join(G, "stats", "**", "r_kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(P, "r04_kernel_stats_bench_final.csv"))
    print("r04_kernel_stats_bench_final.csv")
copy("init_and_time_to_register.jsonl", "r04_init_and_time_to_register_final.jsonl")
copy("gpu_tests.log", "r04_gpu_tests_final.log")
copy("facade_timing.json", "r04_facade_timing.json")
copy("kernels_lanes1.json", "r04_kernels_lanes1.json")
copy("BUILD_INFO.json", "r04_build_info.json")
os.makedirs(os.path.join(P, "r04_kernels_lanes1"), exist_ok=True)
for f in glob.glob(os.path.join(G, "pmc_*_lanes1.csv")) + glob.glob(os.path.join(G, "trace_lanes1.csv")):
    shutil.copy(f, os.path.join(P, "r04_kernels_lanes1", os.path.basename(f)))
    print("r04_kernels_lanes1/" + os.path.basename(f))
