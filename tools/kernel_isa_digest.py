#!/usr/bin/env python
"""Per-kernel digest of the gfx950 instructions inside a built libsuper4pcs_amd.so: the code objects are pulled out of the
library (llvm-objdump --offloading), disassembled, and every kernel's instruction text (addresses and encodings stripped) is
hashed.  Two builds whose digests agree for a kernel run the same instructions for it -- used to show that the per-kernel
counters of one build still describe another (profiles/README.md, round 4).
usage: python tools/kernel_isa_digest.py path/to/libsuper4pcs_amd.so > digest.json"""
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def digest(lib):
    tmp = tempfile.mkdtemp(prefix="isa_")
    try:
        work = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, work)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", work], cwd=tmp, check=True, capture_output=True)
        out = {}
        for f in sorted(os.listdir(tmp)):
            if not f.endswith("gfx950"):
                continue
            asm = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            cur = None
            body = {}
            for line in asm.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
                if m:
                    cur = m.group(1)
                    body[cur] = []
                elif cur and line.strip():
                    body[cur].append(line.split("//")[0].strip())
            for k, v in body.items():
                name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip() or k
                out[name] = {"instructions": len(v), "sha16": hashlib.sha256("\n".join(v).encode()).hexdigest()[:16]}
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    print(json.dumps(digest(sys.argv[1]), indent=1, sort_keys=True))
