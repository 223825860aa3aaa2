"""The command-line program (demos/Super4PCS): flags and exit codes on the CPU, a whole
registration of BASELINE configs[0] through files on the GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def cli(s4p_lib_built):
    from super4pcs_amd import build as B
    return B.build_cli()


def _write_obj(path, pts):
    with open(path, "w") as f:
        f.write("# points\n")
        for p in pts:
            f.write("v %.9g %.9g %.9g\n" % (p[0], p[1], p[2]))
        f.write("# End of File\n")


def test_cli_usage_and_exit_codes(cli, tmp_path):
    r = subprocess.run([cli], capture_output=True, text=True)
    assert r.returncode == 254 and "Usage:" in r.stderr                   # exit(-2), super4pcs_test.cc:67-70
    r = subprocess.run([cli, "-i", "a.obj", "b.obj", "--bogus"], capture_output=True, text=True)
    assert r.returncode == 1 and "Unknown flag" in r.stderr               # :71-76
    r = subprocess.run([cli, "-i", "a.obj", "b.obj", "-o", "1.5"], capture_output=True, text=True)
    assert r.returncode == 253                                            # invalid overlap: exit(-3), :82-85
    r = subprocess.run([cli, "-i", str(tmp_path / "a.obj"), str(tmp_path / "b.obj")], capture_output=True, text=True)
    assert r.returncode == 255 and "Can't read input set1" in r.stderr    # exit(-1), :91-95
    r = subprocess.run([cli, "-i", "a.obj", "b.obj", "-x"], capture_output=True, text=True)
    assert r.returncode == 253 and "4PCS" in r.stderr                     # legacy matcher: out of scope, refused loudly


def test_cli_without_a_gpu_fails_loudly(cli, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    pts = np.random.default_rng(0).normal(size=(50, 3)).astype(np.float32)
    _write_obj(tmp_path / "a.obj", pts)
    _write_obj(tmp_path / "b.obj", pts)
    r = subprocess.run([cli, "-i", str(tmp_path / "a.obj"), str(tmp_path / "b.obj"), "-r", str(tmp_path / "o.obj")],
                       capture_output=True, text=True)
    assert r.returncode == 254 and "no CPU fallback" in r.stderr          # the exception path, :147-151
    assert not (tmp_path / "o.ply").exists() and not (tmp_path / "o.obj").exists()


@pytest.mark.gpu
def test_cli_registers_config1_like_the_reference(cli, tmp_path):
    """hippo1 <-> hippo2 with the flags of scripts/run-example.sh:68, through files.  The fixture holds the two clouds
    after the sampler (sampling them again at the same delta keeps every point) and the reference's own result."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "hippo_config1.npz"))
    _write_obj(tmp_path / "P.obj", g["Ps"])
    _write_obj(tmp_path / "Q.obj", g["Qu"])
    r = subprocess.run([cli, "-i", str(tmp_path / "P.obj"), str(tmp_path / "Q.obj"), "-o", "0.7", "-d", "0.01", "-t", "1000",
                        "-n", "200", "-r", str(tmp_path / "out.obj"), "-m", str(tmp_path / "mat.txt"),
                        "--sampled1", str(tmp_path / "s1.ply"), "--sampled2", str(tmp_path / "s2.ply")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert ("Score: %.6g" % float(g["lcp"])) in r.stdout or ("Score: %g" % float(g["lcp"])) in r.stdout
    M = g["M"].astype(np.float64)
    fmt = lambda v: (" " if v >= 0.0 else "") + "%f" % v
    want = "VERSION\t=\t1\nMATRIX\t=\n" + "".join("  ".join(fmt(M[j, k]) for k in range(4)) + "\n" for j in range(4))
    assert (tmp_path / "mat.txt").read_text() == want                     # Polyworks matrix of the reference's 4x4
    # registered geometry: a point set, so a binary PLY named after -r with the extension replaced (io.cc:285-290)
    raw = (tmp_path / "out.ply").read_bytes()
    head, body = raw.split(b"end_header\n", 1)
    assert b"element vertex 3820" in head and b"format binary_little_endian 1.0" in head
    got = np.frombuffer(body, "<f4").reshape(-1, 3)
    Mf = g["M"]
    x, y, z = g["Qu"][:, 0], g["Qu"][:, 1], g["Qu"][:, 2]
    exp = np.stack([((Mf[r_, 0] * x + Mf[r_, 1] * y) + Mf[r_, 2] * z) + Mf[r_, 3] for r_ in range(3)], 1)
    assert got.shape == exp.shape and np.max(np.abs(got - exp)) <= 1e-4
    for name, n in (("s1.ply", 5281), ("s2.ply", 200)):
        assert ("element vertex %d" % n).encode() in (tmp_path / name).read_bytes().split(b"end_header")[0]
