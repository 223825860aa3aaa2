"""The committed bench line of the last full pass must be recomputable from the rocprofv3 CSVs committed next to it
(tools/recompute_roofline.py): every roofline figure within 5 %, and the kernel trace's k_verify average within 5 % of the
HIP-event average of the same command."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


import pytest


def _round():
    return next(r for r in ("r04", "r03") if os.path.exists(os.path.join(ROOT, "profiles", r + "_bench_final.json")))


def test_roofline_figures_recompute_from_committed_csvs():
    line = os.path.join(ROOT, "profiles", _round() + "_bench_final.json")
    assert os.path.exists(line) and os.path.isdir(os.path.join(ROOT, "profiles", _round() + "_bench_final"))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "recompute_roofline.py"), line], capture_output=True, text=True, check=True).stdout
    rows = [l for l in out.splitlines() if "recomputed" in l]
    assert len(rows) >= 10, out                                   # achieved, frac, traffic, valu, l2 (2), HBM point (3), trace vs events, ...
    for l in rows:
        m = re.search(r"\(([-+][0-9.]+) %\)", l)
        assert m is not None, l
        assert abs(float(m.group(1))) <= 5.0, l


def test_the_committed_bench_line_names_the_code_that_produced_it():
    """provenance (round 4): the line of the final pass carries the commit it was built at, a digest of the device / host
    sources and the C-ABI headers, and the k_verify instantiation the trial loop launched.  The digest must be the one of that
    commit's sources (recomputed here with `git show`), the build must have been clean, the line of the driver's command and
    the kernel trace must come from the same build, and the kernel named must be the one rocprofv3 saw."""
    import csv
    import hashlib
    import json
    if _round() != "r04":
        pytest.skip("no round-4 line committed yet")
    P = os.path.join(ROOT, "profiles")
    d = json.loads(open(os.path.join(P, "r04_bench_final.json")).read())
    prov = d["provenance"]
    sha = prov["git_sha"]
    assert sha and prov["dirty"] is False and prov["source_sha16"] == prov["source_sha16_now"]
    if subprocess.run(["git", "cat-file", "-e", sha + "^{commit}"], cwd=ROOT).returncode != 0:
        pytest.skip("commit %s not in this clone" % sha)
    files = subprocess.run(["git", "ls-tree", "-r", "--name-only", sha, "super4pcs_amd/csrc"], cwd=ROOT, capture_output=True, text=True, check=True).stdout.split()
    h = hashlib.sha256()
    for rel in sorted(files) + ["include/s4p_capi.h", "include/s4p_matcher.h"]:
        h.update(rel.encode())
        h.update(subprocess.run(["git", "show", "%s:%s" % (sha, rel)], cwd=ROOT, capture_output=True, check=True).stdout)
    assert h.hexdigest()[:16] == prov["source_sha16"]
    for other in ("r04_bench_driver_command.json", "r04_bench_under_rocprof_final.json"):
        o = json.loads(open(os.path.join(P, other)).read())["provenance"]
        assert (o["git_sha"], o["source_sha16"], o["k_verify"]) == (sha, prov["source_sha16"], prov["k_verify"])
    inst = re.search(r"k_verify<[a-z, ]+>", prov["k_verify"]).group(0)              # the instantiation of the timed loop
    names = [r["Name"] for r in csv.DictReader(open(os.path.join(P, "r04_kernel_stats_bench_final.csv")))]
    assert any(inst.replace(" ", "") in n.replace(" ", "") for n in names), (inst, names[:6])
    rows = [r for f in os.listdir(os.path.join(P, "r04_bench_final")) if f.startswith("pmc_")
            for r in csv.DictReader(open(os.path.join(P, "r04_bench_final", f)))]
    assert rows and all(inst.replace(" ", "") in r["Kernel_Name"].replace(" ", "") for r in rows[-20:])

