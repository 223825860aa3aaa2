#!/usr/bin/env python
"""VGPR / SGPR / LDS / scratch of every kernel in a built libsuper4pcs_amd.so (from the code objects' metadata notes).
usage: python tools/kernel_resources.py [path/to/libsuper4pcs_amd.so] [name filter]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def resources(lib):
    tmp = tempfile.mkdtemp(prefix="res_")
    out = []
    try:
        work = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, work)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", work], cwd=tmp, check=True, capture_output=True)
        for f in sorted(os.listdir(tmp)):
            if not f.endswith("gfx950"):
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            cur = {}
            for line in notes.splitlines():
                m = re.match(r"\s*[-\s]\s*\.(\w+):\s*(.*)", line)
                if not m:
                    continue
                k, v = m.group(1), m.group(2).strip()
                if k == "name" and "kernel" not in cur.get("_ctx", ""):
                    pass
                if k in ("sgpr_count", "vgpr_count", "agpr_count", "group_segment_fixed_size", "private_segment_fixed_size", "sgpr_spill_count", "vgpr_spill_count", "kernarg_segment_size", "max_flat_workgroup_size"):
                    cur[k] = int(v)
                elif k == "symbol":
                    cur["symbol"] = v.strip("'")
                elif k == "wavefront_size":                  # (the keys of a kernel's map are sorted: this is its last one)
                    out.append(cur)
                    cur = {}
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else os.path.join(ROOT, "super4pcs_amd", "lib", "libsuper4pcs_amd.so")
    flt = sys.argv[-1] if len(sys.argv) > 1 and not os.path.exists(sys.argv[-1]) else ""
    for r in resources(lib):
        name = subprocess.run(["c++filt", r["symbol"].replace(".kd", "")], capture_output=True, text=True).stdout.strip()
        if flt and flt not in name:
            continue
        print("%-70s vgpr %3d agpr %3d sgpr %3d lds %6d scratch %4d spills v%d s%d kernarg %d" % (
            name[:70], r.get("vgpr_count", -1), r.get("agpr_count", 0), r.get("sgpr_count", -1), r.get("group_segment_fixed_size", 0),
            r.get("private_segment_fixed_size", 0), r.get("vgpr_spill_count", 0), r.get("sgpr_spill_count", 0), r.get("kernarg_segment_size", 0)))
