"""CPU (-m "not gpu"): tests/eigen_probe.cpp, the program a user with a REAL Eigen >= 3.3 runs to confirm the fixed-size
evaluation orders the parity chain asserts (DESIGN.md section 2 "Numerics contract").  Eigen is
not in this image, so here the probe is compiled against oracle/eigen_shim: that proves (a) the probe's scalar restatements
of the asserted orders are the orders the shim -- and therefore the pinned oracle and the kernels -- implement, and (b) pins
the digests of the Eigen-side results, which a real-Eigen run must reproduce line for line."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# digest of the results computed THROUGH the Eigen API, per expression group, on the probe's fixed adversarial inputs
EXPECTED = ["61bb12109d806cb9", "79f5807ec2454468", "2c565b18438aba39", "83f0585bf4419e61", "320faded56adcd1c",
            "fcba5d4df8729feb", "23fb2e52d0106909"]


def test_probe_confirms_the_asserted_orders_against_the_shim(tmp_path):
    exe = str(tmp_path / "eigen_probe")
    subprocess.check_call(["g++", "-O2", "-std=c++14", "-ffp-contract=off", "-I" + os.path.join(ROOT, "oracle", "eigen_shim"),
                           os.path.join(ROOT, "tests", "eigen_probe.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout
    lines = [l for l in out.stdout.splitlines() if " digest " in l]
    assert len(lines) == 7 and all("confirmed" in l for l in lines)
    assert [l.split()[-1] for l in lines] == EXPECTED
    assert "every asserted evaluation order confirmed" in out.stdout
