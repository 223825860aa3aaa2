#!/usr/bin/env python
"""Copies the summaries of the round-3 final pass (tools/lab/r3_run13.sh -> gpurun_out/) into profiles/ (tracked)."""
import glob
import os
import shutil

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")


def json_line(src, dst):
    lines = [l for l in open(os.path.join(G, src)).read().splitlines() if l.startswith('{"metric')]
    if not lines:
        print("no JSON line in", src)
        return
    open(os.path.join(P, dst), "w").write(lines[-1] + "\n")
    print(dst)


json_line("r03_bench_final.json", "r03_bench_final.json")
json_line("r03_bench_driver_command.json", "r03_bench_driver_command.json")
json_line("r03_bench_under_rocprof_final.json", "r03_bench_under_rocprof_final.json")
os.makedirs(os.path.join(P, "r03_bench_final"), exist_ok=True)
for f in glob.glob(os.path.join(G, "r03_bench_final", "*.csv")):
    shutil.copy(f, os.path.join(P, "r03_bench_final", os.path.basename(f)))
    print("r03_bench_final/" + os.path.basename(f))
for f in glob.glob(os.path.join(G, "r03_stats_final", "**", "r_kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(P, "r03_kernel_stats_bench_final.csv"))
    print("r03_kernel_stats_bench_final.csv")
src = os.path.join(G, "r03_init_and_time_to_register_final.jsonl")
if os.path.exists(src) and os.path.getsize(src):
    shutil.copy(src, os.path.join(P, "r03_init_and_time_to_register_final.jsonl"))
    print("r03_init_and_time_to_register_final.jsonl")
src = os.path.join(G, "r3_run13_tests.log")
if os.path.exists(src):
    shutil.copy(src, os.path.join(P, "r03_gpu_tests_final.log"))
    print("r03_gpu_tests_final.log")
