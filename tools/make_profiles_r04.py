#!/usr/bin/env python
"""Copies the summaries of a round-4 full pass into profiles/ (tracked).
  python tools/make_profiles_r04.py                 tools/r4/final_pass.sh + final_pass2.sh -> gpurun_out/r04_final/ (commit 290d201)
  python tools/make_profiles_r04.py r04_final3      tools/r4/final_pass3.sh -> gpurun_out/r04_final3/ (the commit that ships);
                                                    the per-kernel counters of the first pass are left in place"""
import glob
import os
import shutil
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASS = sys.argv[1] if len(sys.argv) > 1 else "r04_final"
G, P = os.path.join(R, "gpurun_out", PASS), os.path.join(R, "profiles")


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
copy("facade_timing.json", "r04_facade_timing.json")
copy("BUILD_INFO.json", "r04_build_info.json")
if PASS == "r04_final":
    copy("gpu_tests.log", "r04_gpu_tests_final.log")
    copy("kernels_lanes1.json", "r04_kernels_lanes1.json")
    os.makedirs(os.path.join(P, "r04_kernels_lanes1"), exist_ok=True)
    for f in glob.glob(os.path.join(G, "pmc_*_lanes1.csv")) + glob.glob(os.path.join(G, "trace_lanes1.csv")):
        shutil.copy(f, os.path.join(P, "r04_kernels_lanes1", os.path.basename(f)))
        print("r04_kernels_lanes1/" + os.path.basename(f))
else:
    copy("gpu_tests_configs.log", "r04_gpu_tests_final_configs.log")
    copy("gpu_tests_rest.log", "r04_gpu_tests_final_rest.log")
    copy("sim_world.jsonl", "r04_sim_world.jsonl")
    copy("select_probe.json", "r04_select_probe.json")
    for src, dst in (("init_timing.err", "r04_init_breakdown_trace.log"), ("sim_world.err", "r04_sim_world_host_chain_trace.log")):
        path = os.path.join(G, src)
        if os.path.exists(path):
            lines = [l for l in open(path).read().splitlines() if "s4p_trace" in l]
            open(os.path.join(P, dst), "w").write("\n".join(lines) + "\n")
            print(dst)
