#!/usr/bin/env python
"""Collects the round-2 measurements that gpurun merged into gpurun_out/ (scratch, untracked) into profiles/ (tracked):
  profiles/r02_ab_runs.json          every A/B arm of tools/lab/gpu_run*.sh (tools/ab_one.py lines), per run script
  profiles/r02_pmc_k_verify.json     rocprofv3 --pmc means per k_verify launch, per variant / counter set
  profiles/r02_kernel_stats_*.csv    rocprofv3 --kernel-trace --stats summaries (1 and 3 lanes)
  profiles/r02_final_kernel_stats_bench_*.csv   the same for the bench.py command of the final pass
  profiles/r02_bench*.json           bench.py lines
Run from the repo root after the GPU passes."""
import collections
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")

ab = {}
for f in sorted(glob.glob(os.path.join(G, "r2_ab*.log"))):
    rows = []
    for line in open(f):
        line = line.strip()
        if line.startswith("{"):
            try:
                rows.append(json.loads(line))
            except ValueError:
                pass
    ab[os.path.basename(f)] = rows
json.dump({"note": "one line per arm: tools/ab_one.py on the bench workload (configs[2], 100 timed bases after 5, best of 3 "
                   "repeats; verify_ms / pairs_ms / quads_ms are HIP-event means per launch; equal digests = equal results); "
                   "the run scripts (tools/lab/gpu_run*.sh) name the library / environment of every arm, DESIGN.md section 5 "
                   "lists what each variant was", "runs": ab}, open(os.path.join(P, "r02_ab_runs.json"), "w"), indent=1)

pmc = {}
for f in sorted(glob.glob(os.path.join(G, "r*pmc_*", "**", "p_counter_collection.csv"), recursive=True)):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_verify<false" in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if d:
        pmc[os.path.relpath(f, G).split(os.sep)[0]] = {k: {"mean_per_launch": sum(v) / len(v), "launches": len(v)} for k, v in sorted(d.items())}
json.dump({"note": "rocprofv3 --pmc <set> --kernel-trace -- python tools/ab_one.py 30 1 (S4P_LANES=1), k_verify<false,...> rows only; "
                   "directory names: r<run>pmc_<library>_<set>; SQ_* are summed over all waves (quad-cycles)", "counters": pmc},
          open(os.path.join(P, "r02_pmc_k_verify.json"), "w"), indent=1)

for f in glob.glob(os.path.join(G, "r2stats_l*", "**", "r_kernel_stats.csv"), recursive=True):
    lanes = os.path.relpath(f, G).split(os.sep)[0].replace("r2stats_", "")
    shutil.copy(f, os.path.join(P, "r02_kernel_stats_%s.csv" % lanes))
# rocprofv3 --kernel-trace --stats of the bench.py command itself (tools/lab/gpu_run12.sh): default lanes and S4P_LANES=1
for tag, name in (("r2stats_bench", "r02_final_kernel_stats_bench_3lanes.csv"), ("r2stats_bench_l1", "r02_final_kernel_stats_bench_1lane.csv")):
    for f in glob.glob(os.path.join(G, tag, "**", "r_kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(P, name))
for f in glob.glob(os.path.join(G, "r2_bench*.json")):
    shutil.copy(f, os.path.join(P, os.path.basename(f).replace("r2_", "r02_")))
print("profiles updated:", sorted(x for x in os.listdir(P) if x.startswith("r02_")))
