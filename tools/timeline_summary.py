#!/usr/bin/env python
"""Summary of a kernel timeline ([[name, start_ns, end_ns, queue, ...], ...] as tools/r5/run4.sh / run11.sh write it from a
rocprofv3 --kernel-trace CSV): per kernel launches / mean duration, share of the time some kernel is running, histogram of
the number of kernels running at once, over the steady middle of the run.   usage: timeline_summary.py timeline.json [label]"""
import collections
import json
import sys


def short(n):
    for k in ("copyBuffer", "k_pairs2", "k_prep", "k_quads", "k_verify", "k_reset"):
        if k in n:
            return k
    return n[:16]


def summary(path, label=""):
    T = json.load(open(path))
    T.sort(key=lambda r: r[1])
    ver = [r for r in T if "k_verify" in r[0]]
    lo, hi = ver[int(len(ver) * 0.15)][1], ver[int(len(ver) * 0.75)][2]
    W = [r for r in T if r[1] >= lo and r[2] <= hi]
    nv = sum(1 for r in W if "k_verify" in r[0])
    d = collections.defaultdict(list)
    for r in W:
        d[short(r[0])].append((r[2] - r[1]) / 1e3)
    ev = sorted([(r[1], 1) for r in W] + [(r[2], -1) for r in W])
    cur, last, hist = 0, None, collections.Counter()
    for t, dd in ev:
        if last is not None:
            hist[cur] += t - last
        cur += dd
        last = t
    tot = float(sum(hist.values()))
    return {"label": label, "window_us": (hi - lo) / 1e3, "k_verify_launches": nv, "us_per_k_verify_launch": (hi - lo) / 1e3 / max(nv, 1),
            "kernels": {k: {"launches": len(v), "mean_us": sum(v) / len(v)} for k, v in sorted(d.items())},
            "some_kernel_running": 1.0 - hist[0] / tot, "kernels_running_at_once": {str(k): v / tot for k, v in sorted(hist.items())}}


if __name__ == "__main__":
    print(json.dumps(summary(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")))
