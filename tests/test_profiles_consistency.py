"""The committed bench line of the last full pass must be recomputable from the rocprofv3 CSVs committed next to it
(tools/recompute_roofline.py): every roofline figure within 5 %, and the kernel trace's k_verify average within 5 % of the
HIP-event average of the same command."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_roofline_figures_recompute_from_committed_csvs():
    line = os.path.join(ROOT, "profiles", "r03_bench_final.json")
    assert os.path.exists(line) and os.path.isdir(os.path.join(ROOT, "profiles", "r03_bench_final"))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "recompute_roofline.py")], capture_output=True, text=True, check=True).stdout
    rows = [l for l in out.splitlines() if "recomputed" in l]
    assert len(rows) >= 10, out                                   # achieved, frac, traffic, valu, l2 (2), HBM point (3), trace vs events, ...
    for l in rows:
        m = re.search(r"\(([-+][0-9.]+) %\)", l)
        assert m is not None, l
        assert abs(float(m.group(1))) <= 5.0, l
