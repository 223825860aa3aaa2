"""Builds lib/libsuper4pcs_amd.so with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
LIBDIR = os.path.join(_HERE, "lib")
LIB = os.path.join(LIBDIR, "libsuper4pcs_amd.so")

# -ffp-contract=off: integer inlier counts are bit-exact only if no mul+add is fused.
# -fhip-fp32-correctly-rounded-divide-sqrt is the hipcc default; stated explicitly because
# parity depends on IEEE sqrt and divide.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
               "-fhip-fp32-correctly-rounded-divide-sqrt", "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]
SOURCES = ["s4p_capi.hip", "s4p_sampler.hip", "s4p_engine.cpp", "s4p_shard.cpp"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "s4p_capi.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(out, defines=(), verbose=False):
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = [_hipcc()] + HIPCC_FLAGS + ["-D" + d for d in defines] + ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    for s in srcs:
        cmd += ["-x", "hip", s]
    cmd += ["-ldl", "-o", out]              # s4p_shard.cpp binds RCCL at run time (dlopen)
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out


def source_digest():
    """sha256 over the device / host sources and the two C-ABI headers (sorted by name): ties a measurement to the code that
    produced it (the GPU box has no .git; tests/test_profiles_consistency.py recomputes it from the recorded commit)."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join("super4pcs_amd", "csrc", f) for f in os.listdir(CSRC)) + ["include/s4p_capi.h", "include/s4p_matcher.h"]
    for rel in files:
        h.update(rel.encode())
        with open(os.path.join(ROOT, rel), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


BUILD_INFO = os.path.join(LIBDIR, "BUILD_INFO.json")


def _stamp():
    """lib/BUILD_INFO.json: the commit the library was built at (dirty = sources differ from that commit) and the source digest."""
    import json
    sha, dirty = None, None
    try:
        sha = subprocess.check_output(["git", "rev-parse", "HEAD"], cwd=ROOT, stderr=subprocess.DEVNULL).decode().strip()
        dirty = bool(subprocess.check_output(["git", "status", "--porcelain", "--", "super4pcs_amd/csrc", "include"], cwd=ROOT,
                                             stderr=subprocess.DEVNULL).decode().strip())
    except Exception:                                              # noqa: BLE001 -- no git here (the GPU box): keep what the dev box stamped
        if os.path.exists(BUILD_INFO):
            return
    with open(BUILD_INFO, "w") as fh:
        json.dump({"git_sha": sha, "dirty": dirty, "source_sha16": source_digest()}, fh)


def build_info():
    import json
    try:
        with open(BUILD_INFO) as fh:
            info = json.load(fh)
    except Exception:                                              # noqa: BLE001
        info = {"git_sha": None, "dirty": None, "source_sha16": None}
    info["source_sha16_now"] = source_digest()                     # must equal source_sha16: the sources the .so was built from
    return info


def build(force=False, verbose=False):
    if not force and not needs_build():
        if not os.path.exists(BUILD_INFO):
            _stamp()
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    out = _compile(LIB, (), verbose)
    _stamp()
    return out


def build_variant(name, defines):
    """A/B aid: the same sources with extra -D switches (e.g. S4P_FINE_FLAT=1) -> scratch/lib<name>.so; select it on the
    GPU box with S4P_LIB=<path> (tools/ab_one.py)."""
    d = os.path.join(ROOT, "scratch")
    os.makedirs(d, exist_ok=True)
    return _compile(os.path.join(d, "lib%s.so" % name), defines)


BINDIR = os.path.join(_HERE, "bin")
CLI = os.path.join(BINDIR, "Super4PCS")
CLI_SRC = os.path.join(ROOT, "demos", "Super4PCS", "super4pcs_cli.cc")


def build_cli(force=False):
    """The command-line program (demos/Super4PCS) against the facade headers and the library: plain host C++."""
    build()
    deps = [CLI_SRC, os.path.join(ROOT, "demos", "cli_options.h"), LIB,
            os.path.join(ROOT, "include", "super4pcs", "io", "io.h"),
            os.path.join(ROOT, "include", "super4pcs", "algorithms", "match4pcsBase.h")]
    if not force and os.path.exists(CLI) and all(os.path.getmtime(d) <= os.path.getmtime(CLI) for d in deps):
        return CLI
    os.makedirs(BINDIR, exist_ok=True)
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "include"), CLI_SRC,
           "-L" + LIBDIR, "-lsuper4pcs_amd", "-Wl,-rpath,$ORIGIN/../lib", "-o", CLI]
    subprocess.check_call(cmd)
    return CLI


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_cli(force="--force" in sys.argv))
