#!/usr/bin/env python
"""Copies the summaries of the round-6 full pass (tools/r6/final_pass_a.sh + final_pass_b.sh -> gpurun_out/r06_final/) into
profiles/ (tracked) and writes the instruction digests of the library that was measured.   python tools/make_profiles_r06.py"""
import glob
import json
import os
import shutil
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out", "r06_final"), os.path.join(R, "profiles")


def json_line(src, dst):
    path = os.path.join(G, src)
    lines = [l for l in open(path).read().splitlines() if l.startswith('{"metric')] if os.path.exists(path) else []
    if not lines:
        print("missing / no JSON line:", src)
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


json_line("bench_final.json", "r06_bench_final.json")
json_line("bench_driver_command.json", "r06_bench_driver_command.json")
json_line("bench_under_rocprof.json", "r06_bench_under_rocprof_final.json")
os.makedirs(os.path.join(P, "r06_bench_final"), exist_ok=True)
for f in glob.glob(os.path.join(G, "bench_final", "*.csv")):
    shutil.copy(f, os.path.join(P, "r06_bench_final", os.path.basename(f)))
    print("r06_bench_final/" + os.path.basename(f))
for f in glob.glob(os.path.join(G, "stats", "**", "r_kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(P, "r06_kernel_stats_bench_final.csv"))
    print("r06_kernel_stats_bench_final.csv")
copy("facade_timing.json", "r06_facade_timing.json")
copy("facade_timing_default_env.json", "r06_facade_timing_default_env.json")
copy("BUILD_INFO.json", "r06_build_info.json")
copy("gpu_tests_configs.log", "r06_gpu_tests_final_configs.log")
copy("gpu_tests_rest.log", "r06_gpu_tests_final_rest.log")
copy("sim_world.jsonl", "r06_sim_world.jsonl")
path = os.path.join(G, "sim_world.err")
if os.path.exists(path):
    lines = [l for l in open(path).read().splitlines() if "s4p_trace" in l]
    open(os.path.join(P, "r06_sim_world_host_chain_trace.log"), "w").write("\n".join(lines) + "\n")
    print("r06_sim_world_host_chain_trace.log")
# time-to-register of the other configs: out of the bench line's extra object into a file of its own as well
try:
    d = json.loads(open(os.path.join(P, "r06_bench_final.json")).read())
    rows = (d.get("extra") or {}).get("time_to_register_configs") or []
    if isinstance(rows, list) and rows:
        with open(os.path.join(P, "r06_init_and_time_to_register_final.jsonl"), "w") as fh:
            for r in rows:
                fh.write(json.dumps(r) + "\n")
        print("r06_init_and_time_to_register_final.jsonl")
except Exception as e:                                              # noqa: BLE001
    print("time-to-register rows:", repr(e))
# part C: the kernels alone (one base per launch) with their counter CSVs, the lab build's per-wave phase profile, the two-rank dry run
copy(os.path.join("lanes1", "kernels_lanes1.json"), "r06_kernels_lanes1.json")
os.makedirs(os.path.join(P, "r06_kernels_lanes1"), exist_ok=True)
for f in glob.glob(os.path.join(G, "lanes1", "*.csv")):
    shutil.copy(f, os.path.join(P, "r06_kernels_lanes1", os.path.basename(f)))
    print("r06_kernels_lanes1/" + os.path.basename(f))
copy("wave_profile.txt", "r06_wave_profile.txt")
json_line("bench_2ranks_one_gpu.json", "r06_bench_2ranks_dryrun_one_gpu.json")
sys.path.insert(0, os.path.join(R, "tools"))
import kernel_isa_digest  # noqa: E402
json.dump(kernel_isa_digest.digest(os.path.join(R, "super4pcs_amd", "lib", "libsuper4pcs_amd.so")),
          open(os.path.join(P, "r06_kernel_isa_final.json"), "w"), indent=1, sort_keys=True)
print("r06_kernel_isa_final.json")
