"""The reference-API facade (include/super4pcs/...) as an external C++ application, like the reference's
tests/externalAppTest: compiled with g++ against the headers and libsuper4pcs_amd.so."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path, s4p_lib_built):
    exe = str(tmp_path / "facade_app")
    libdir = os.path.join(ROOT, "super4pcs_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "facade_app", "main.cpp"), "-L" + libdir, "-lsuper4pcs_amd",
                           "-Wl,-rpath," + libdir, "-o", exe])
    return exe


def test_facade_compiles_and_fails_loudly_without_gpu(tmp_path, s4p_lib_built):
    import torch
    exe = _build(tmp_path, s4p_lib_built)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    out = subprocess.run([exe, "0"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "no CPU fallback" in out.stdout


@pytest.mark.gpu
def test_facade_registers_on_gpu(tmp_path, s4p_lib_built):
    exe = _build(tmp_path, s4p_lib_built)
    out = subprocess.run([exe, "1"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "score" in out.stdout and "quads" in out.stdout
    assert "routes: staged" in out.stdout and "filtering subclass" in out.stdout     # overridden hooks are honoured (main.cpp)
    assert "samplers: user-supplied" in out.stdout                                   # both sampling routes, same registration
