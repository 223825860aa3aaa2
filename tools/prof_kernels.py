#!/usr/bin/env python
"""rocprofv3 over the bench workload's timed region (bench.py --inner): per-kernel durations from --kernel-trace and
per-kernel counters from separate --pmc passes, for ALL kernels of a base (k_pairs, k_prep, k_quads, k_verify).
Usage: python tools/prof_kernels.py OUT_DIR [--lanes 1] [--steps 100]
Writes OUT_DIR/kernels_<tag>.json (means per launch over the timed launches) and keeps the raw CSV rows of those kernels."""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ("k_pairs", "k_prep", "k_quads", "k_sweep", "k_verify")
PASSES = {
    "sq": ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "GRBM_GUI_ACTIVE"],
    "sq2": ["SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_ANY"],
    "lds": ["SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS", "GRBM_GUI_ACTIVE"],
    "tcc": ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum"],
    "fetch": ["FETCH_SIZE"],
    "write": ["WRITE_SIZE"],
}


def short(name):
    for k in KERNELS:
        if k + "<" in name or name.startswith(k) or ("::" + k) in name:
            return k
    return None


def run(exe, extra, env, steps, warmup, d):
    cmd = [exe] + extra + ["--kernel-trace", "-d", d, "-o", "p", "--output-format", "csv", "--", sys.executable,
                           os.path.join(ROOT, "bench.py"), "--inner", "--steps", str(steps), "--warmup", str(warmup)]
    subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300, check=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--lanes", default=None)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--passes", default="trace,sq,sq2,tcc,fetch")
    ap.add_argument("--tag", default=None)
    a = ap.parse_args()
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    env = dict(os.environ, TMPDIR="/tmp")
    if a.lanes:
        env["S4P_LANES"] = a.lanes
    tag = a.tag or ("lanes%s" % (a.lanes or "default"))
    os.makedirs(a.out, exist_ok=True)
    res = {"tag": tag, "steps": a.steps, "warmup": a.warmup, "lanes": a.lanes or "default", "kernels": {k: {} for k in KERNELS}}
    for p in a.passes.split(","):
        d = tempfile.mkdtemp(prefix="s4p_prof_", dir="/tmp")
        try:
            run(exe, [] if p == "trace" else ["--pmc"] + PASSES[p], env, a.steps, a.warmup, d)
            if p == "trace":
                dur = {k: [] for k in KERNELS}
                for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
                    rows = list(csv.DictReader(open(f)))
                    for r in rows:
                        k = short(r.get("Kernel_Name", ""))
                        if k:
                            dur[k].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
                    keep = [r for r in rows if short(r.get("Kernel_Name", ""))]
                    with open(os.path.join(a.out, "trace_%s.csv" % tag), "w", newline="") as fo:
                        w = csv.DictWriter(fo, fieldnames=["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Workgroup_Size", "Grid_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count"], extrasaction="ignore")
                        w.writeheader(); w.writerows(keep)
                for k, v in dur.items():
                    v = sorted(v)[-a.steps * (2 if False else 1):] if v else v
                    if v:
                        res["kernels"][k]["launches"] = len(v)
                        res["kernels"][k]["avg_us"] = sum(e - s for s, e in v) / len(v) / 1e3
                allv = sorted(x for v in dur.values() for x in v)
                if allv:
                    t0, t1 = allv[0][0], max(e for _, e in allv)
                    busy, cur_s, cur_e = 0, None, None
                    for s, e in allv:                       # union of the intervals: time with at least one kernel running
                        if cur_e is None or s > cur_e:
                            if cur_e is not None:
                                busy += cur_e - cur_s
                            cur_s, cur_e = s, e
                        else:
                            cur_e = max(cur_e, e)
                    busy += cur_e - cur_s
                    res["span_us"] = (t1 - t0) / 1e3
                    res["busy_us"] = busy / 1e3
                    res["sum_kernel_us"] = sum(e - s for s, e in allv) / 1e3
            else:
                vals = {}
                for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                    rows = list(csv.DictReader(open(f)))
                    keep = []
                    for r in rows:
                        k = short(r.get("Kernel_Name", ""))
                        if k and r.get("Counter_Name") in PASSES[p]:
                            vals.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
                            keep.append(r)
                    with open(os.path.join(a.out, "pmc_%s_%s.csv" % (p, tag)), "w", newline="") as fo:
                        w = csv.DictWriter(fo, fieldnames=["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"], extrasaction="ignore")
                        w.writeheader(); w.writerows(keep)
                for (k, c), v in vals.items():
                    v = v[-a.steps:]
                    res["kernels"][k][c] = sum(v) / len(v)
        except Exception as e:                                  # noqa: BLE001
            res.setdefault("errors", []).append("%s: %s %s" % (p, type(e).__name__, e))
        finally:
            shutil.rmtree(d, ignore_errors=True)
    for k, v in res["kernels"].items():                         # derived: VALU issue utilisation, wait share
        if "SQ_ACTIVE_INST_VALU" in v and "GRBM_GUI_ACTIVE" in v and v["GRBM_GUI_ACTIVE"]:
            v["valu_util"] = v["SQ_ACTIVE_INST_VALU"] / (1024.0 * v["GRBM_GUI_ACTIVE"] / 8.0 / 4.0)
        if "SQ_WAIT_ANY" in v and v.get("SQ_WAVE_CYCLES"):
            v["wait_share"] = v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]
        if "TCC_HIT_sum" in v and (v["TCC_HIT_sum"] + v.get("TCC_MISS_sum", 0)):
            v["l2_hit"] = v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
    with open(os.path.join(a.out, "kernels_%s.json" % tag), "w") as fo:
        json.dump(res, fo, indent=1)
    print(json.dumps({k: {c: (round(x, 4) if isinstance(x, float) else x) for c, x in v.items() if c in ("avg_us", "valu_util", "wait_share", "l2_hit", "launches", "SQ_INSTS_VALU", "SQ_WAVES", "FETCH_SIZE")} for k, v in res["kernels"].items()}))
    print(json.dumps({k: res.get(k) for k in ("span_us", "busy_us", "sum_kernel_us", "errors")}))


if __name__ == "__main__":
    main()
