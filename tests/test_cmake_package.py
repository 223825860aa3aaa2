"""The CMake package surface of SURVEY.md 8(b): Super4PCS_INCLUDE_DIR / Super4PCS_LIB_DIR / Super4PCS_LIBRARIES
(reference: cmake/Config.cmake.in:34-39), consumed by an application outside the tree through find_package the way the
reference's tests/externalAppTest is (tests/CMakeLists.txt:35-59, tests/externalAppTest/CMakeLists.txt:6-11) -- from the
source tree and from an installed prefix (tools/install.sh)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
APP = os.path.join(ROOT, "tests", "external_app")


def _configure_build_run(tmp_path, prefix_path, expect_lib_dir):
    if shutil.which("cmake") is None:
        pytest.skip("cmake not on PATH")
    b = tmp_path / "build"
    b.mkdir()
    cfg = subprocess.run(["cmake", "-S", APP, "-B", str(b), "-DCMAKE_PREFIX_PATH=" + str(prefix_path),
                          "-DEIGEN3_INCLUDE_DIR=" + os.path.join(ROOT, "oracle", "eigen_shim")], capture_output=True, text=True, timeout=300)
    assert cfg.returncode == 0, cfg.stdout[-3000:] + cfg.stderr[-3000:]
    assert "Super4PCS_LIBRARIES  : super4pcs_accel;super4pcs_io;super4pcs_algo" in cfg.stdout      # the reference's three names
    assert ("Super4PCS_LIB_DIR    : " + expect_lib_dir) in cfg.stdout
    bld = subprocess.run(["cmake", "--build", str(b)], capture_output=True, text=True, timeout=600)
    assert bld.returncode == 0, bld.stdout[-3000:] + bld.stderr[-3000:]
    run = subprocess.run([str(b / "external_app")], capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert run.returncode == 0, run.stdout + run.stderr
    assert "matrix written=1" in run.stdout
    text = (tmp_path / "external_app_output.map").read_text()
    assert text.startswith("VERSION\t=\t1\nMATRIX\t=\n")                                            # io.cc:461-482
    return run.stdout


def test_find_package_from_the_source_tree(tmp_path, s4p_lib_built):
    import torch
    out = _configure_build_run(tmp_path, ROOT, os.path.join(ROOT, "super4pcs_amd", "lib") + "/")
    if not torch.cuda.is_available():
        assert "no CPU fallback" in out                  # the product path refuses loudly without a device


def test_find_package_from_an_installed_prefix(tmp_path, s4p_lib_built):
    prefix = tmp_path / "prefix"
    subprocess.check_call(["bash", os.path.join(ROOT, "tools", "install.sh"), str(prefix)])
    assert (prefix / "include" / "super4pcs" / "algorithms" / "super4pcs.h").exists()
    # the reference's own external-app test points CMAKE_PREFIX_PATH at <install prefix>/lib/cmake (tests/CMakeLists.txt:46)
    _configure_build_run(tmp_path, prefix / "lib" / "cmake", str(prefix / "lib") + "/")


@pytest.mark.gpu
def test_external_app_registers_nothing_but_runs_on_gpu(tmp_path, s4p_lib_built):
    out = _configure_build_run(tmp_path, ROOT, os.path.join(ROOT, "super4pcs_amd", "lib") + "/")
    assert "score=1e+09" in out                          # empty clouds: kLargeNumber (match4pcsBase.hpp:69-70), matcher constructed on the device
