"""Regenerates tests/golden/io/*: what the REFERENCE's IOManager (src/super4pcs/io/io.cc, compiled unmodified against
oracle/eigen_shim by `make -C oracle ref`) reads from / writes for the files of tests/io_cases.py.
Needs /root/reference; run from the repo root:  python tests/golden/make_io_golden.py"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import io_cases  # noqa: E402


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def run_all(binary, workdir):
    """-> {case: {"read": sha, "clean": sha, "write": {"file": name, "sha": sha}}, "matrices": [text, ...]}"""
    res = {}
    cases = io_cases.write_cases(workdir)
    for name, path in cases:
        r = {}
        for cmd in ("read", "clean"):
            out = os.path.join(workdir, "%s.%s.txt" % (name, cmd))
            subprocess.run([binary, cmd, path, out], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            r[cmd] = sha(out)
        outbase = os.path.join(workdir, "out_%s.xyz" % name)
        rc = subprocess.run([binary, "write", path, outbase], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode
        produced = [e for e in ("ply", "obj") if os.path.exists(outbase[:-3] + e)]
        r["write"] = {"rc": rc, "ext": produced[0] if produced else None, "sha": sha(outbase[:-3] + produced[0]) if produced else None}
        for e in produced:
            os.remove(outbase[:-3] + e)
        res[name] = r
    mats = []
    for m in io_cases.MATRICES:
        out = os.path.join(workdir, "m.txt")
        subprocess.run([binary, "matrix", out] + ["%r" % float(x) for x in m], check=True)
        mats.append(open(out).read())
    res["matrices"] = mats
    return res


if __name__ == "__main__":
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    with tempfile.TemporaryDirectory() as d:
        res = run_all(os.path.join(ROOT, "oracle", "_ref", "ref_io"), d)
    os.makedirs(os.path.join(ROOT, "tests", "golden", "io"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "tests", "golden", "io", "reference_io.json"), "w"), indent=1, sort_keys=True)
    print("cases:", len(res) - 1)
