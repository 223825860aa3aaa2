#!/usr/bin/env python
"""Recomputes every figure of a bench line's `roofline` object from the files committed next to it (no GPU needed):
  python tools/recompute_roofline.py [profiles/r04_bench_final.json [profiles/r04_bench_final]]      (default: the newest round)
  - frac (per step): value x algorithmic_bytes_per_candidate / 8 TB/s
  - traffic: mean FETCH_SIZE (KB, x 2: gfx950 tallies 128-B requests at 64 B) + mean WRITE_SIZE (KB) of the timed launches
  - valu: mean SQ_ACTIVE_INST_VALU / (1024 SIMDs x mean GRBM_GUI_ACTIVE / 8 XCDs / 4)
  - l2: (TCC_HIT_sum + TCC_MISS_sum) x 128 B / the kernel's own launch time / 34.5 TB/s
  - HBM-bound point: FETCH_SIZE x 2 of the one cold k_verify_T launch / its duration in the kernel trace
  - per-launch time: the rocprofv3 kernel-stats average next to the HIP-event average of the same command
and prints them beside the values in the JSON line."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_newest = next((r for r in ("r05", "r04", "r03") if os.path.exists(os.path.join(ROOT, "profiles", r + "_bench_final.json"))), "r03")
line = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", _newest + "_bench_final.json")
_round = os.path.basename(line)[:3]
pdir = sys.argv[2] if len(sys.argv) > 2 else os.path.splitext(line)[0]
d = json.load(open(line))
r = d["roofline"]
steps = d["steps"]


def show(name, mine, theirs):
    rel = "" if not theirs else "  (%+.2f %%)" % (100.0 * (mine - theirs) / theirs)
    print("%-46s recomputed %-16.6g in the line %-16.6g%s" % (name, mine, theirs if theirs is not None else float("nan"), rel))


n_timed = int(r["per_launch"]["launches"]) if _round >= "r05" else steps      # round 5: a launch covers a group of bases; the line says how many launches the timed region had


def counters(fname):
    out = {}
    for row in csv.DictReader(open(os.path.join(pdir, fname))):
        out.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    return {k: (v[-n_timed:] if len(v) >= n_timed else v) for k, v in out.items()}                   # the timed launches are the last ones


def mean(v):
    return sum(v) / len(v)


show("roofline.achieved (GB/s)", d["value"] * r["algorithmic_bytes_per_candidate"] / 1e9, r["achieved"])
show("roofline.frac", d["value"] * r["algorithmic_bytes_per_candidate"] / 1e9 / r["peak"], r["frac"])
files = {f: f for f in os.listdir(pdir)}
a = next((f for f in files if f.startswith("pmc_FETCH_SIZE")), None)
w = next((f for f in files if f.startswith("pmc_WRITE_SIZE")), None)
t = next((f for f in files if f.startswith("pmc_TCC_HIT")), None)
if a and w:
    ca, cw = counters(a), counters(w)
    show("roofline.traffic (bytes per launch)", mean(ca["FETCH_SIZE"]) * 1024 * 2 + mean(cw["WRITE_SIZE"]) * 1024, r["traffic"])
    avail = 1024 * mean(ca["GRBM_GUI_ACTIVE"]) / 8 / 4
    show("roofline.valu.frac", mean(ca["SQ_ACTIVE_INST_VALU"]) / avail, r["valu"]["frac"])
    show("  VALU instructions per candidate", mean(ca["SQ_INSTS_VALU"]) / r["per_launch"]["candidates_per_launch"], r["valu"]["valu_instructions_per_candidate"])
if t:
    ct = counters(t)
    req = mean(ct["TCC_HIT_sum"]) + mean(ct["TCC_MISS_sum"])
    ex = r["l2"].get("launch_ms") or (r["per_launch"]["exclusive"]["avg_launch_ms"] if r["per_launch"].get("exclusive") else r["per_launch"]["avg_launch_ms"])
    show("roofline.l2.hit_rate", mean(ct["TCC_HIT_sum"]) / req, r["l2"]["hit_rate"])
    show("roofline.l2.frac", req * 128 / (ex * 1e-3) / 1e9 / r["l2"]["peak_GBps"], r["l2"]["frac"])
kt, cc = os.path.join(pdir, "hbm_point_kernel_trace.csv"), os.path.join(pdir, "hbm_point_counter_collection.csv")
if os.path.exists(kt) and os.path.exists(cc) and r.get("hbm_bound_point"):
    rows = [x for x in csv.DictReader(open(kt)) if "k_verify_T" in x["Kernel_Name"]]
    big = max(rows, key=lambda x: int(x["Grid_Size_X"]))               # the one launch over the 4096 transforms
    ms = (int(big["End_Timestamp"]) - int(big["Start_Timestamp"])) * 1e-6
    fetch = [float(x["Counter_Value"]) for x in csv.DictReader(open(cc)) if x["Dispatch_Id"] == big["Dispatch_Id"] and x["Counter_Name"] == "FETCH_SIZE"]
    h = r["hbm_bound_point"]
    show("hbm_bound_point.kernel_ms", ms, h["kernel_ms"])
    show("hbm_bound_point.measured_GBps", fetch[0] * 1024 * 2 / (ms * 1e-3) / 1e9, h["measured_GBps"])
    show("hbm_bound_point.frac (algorithmic gathers)", h["algorithmic_bytes"] / (ms * 1e-3) / 1e9 / h["peak_GBps"], h["frac"])
stats = os.path.join(ROOT, "profiles", _round + "_kernel_stats_bench_final.csv")
under = os.path.join(ROOT, "profiles", _round + "_bench_under_rocprof_final.json")
if os.path.exists(stats) and os.path.exists(under):
    row = next(x for x in csv.DictReader(open(stats)) if "k_verify<" in x["Name"])
    hip = json.load(open(under))["roofline"]["per_launch"]["avg_launch_ms"]
    show("k_verify ms per launch: rocprofv3 vs HIP events", float(row["AverageNs"]) * 1e-6, hip)
