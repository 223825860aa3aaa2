"""CPU (-m "not gpu"): the IO row.  include/super4pcs/io/io.h against the REFERENCE's IOManager.

The same harness (tests/io_app/io_main.cpp) is compiled against the product's header and -- by `make -C oracle ref`,
where /root/reference exists -- against the reference's unmodified io.cc.  tests/golden/io/reference_io.json holds what
the reference build read and wrote for the deterministic files of tests/io_cases.py (digests of full dumps: every
vertex position, normal and colour as hex floats, normal / texture / face lists; the bytes of the files written back;
the Polyworks matrix text); the product must reproduce all of it.  With /root/reference present the fixture itself is
re-derived from the live reference and the two asset files are compared directly."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
GOLDEN = os.path.join(ROOT, "tests", "golden", "io", "reference_io.json")
REF_ASSETS = "/root/reference/assets"


@pytest.fixture(scope="module")
def product_io(tmp_path_factory):
    out = tmp_path_factory.mktemp("io_app") / "io_product"
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", "-DS4P_NO_EIGEN", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "io_app", "io_main.cpp"), "-o", str(out)], check=True)
    return str(out)


def test_product_io_reproduces_the_reference_fixture(product_io, tmp_path):
    import make_io_golden
    want = json.load(open(GOLDEN))
    got = make_io_golden.run_all(product_io, str(tmp_path))
    assert set(got) == set(want)
    for name in sorted(want):
        assert got[name] == want[name], name


def test_unreadable_inputs_fail_like_the_reference(product_io, tmp_path):
    bad = tmp_path / "nothing.xyz"
    bad.write_text("1 2 3\n")
    dump = tmp_path / "d.txt"
    subprocess.run([product_io, "read", str(bad), str(dump)], check=True, capture_output=True)
    assert dump.read_text().startswith("ok 0 ")                       # unsupported extension (io.cc:40-42)
    subprocess.run([product_io, "read", str(tmp_path / "missing.obj"), str(dump)], check=True, capture_output=True)
    assert dump.read_text().startswith("ok 0 ")                       # cannot open
    empty = tmp_path / "empty.obj"
    empty.write_text("# no vertices\n")
    subprocess.run([product_io, "read", str(empty), str(dump)], check=True, capture_output=True)
    assert dump.read_text().startswith("ok 0 v 0 ")                   # `if (v.size() == 0) return false` (io.cc:266)
    notply = tmp_path / "fake.ply"
    notply.write_text("plx\n")
    subprocess.run([product_io, "read", str(notply), str(dump)], check=True, capture_output=True)
    assert dump.read_text().startswith("ok 0 ")


@pytest.mark.skipif(not os.path.isdir(REF_ASSETS), reason="/root/reference not present (GPU box): fixture-only")
def test_fixture_and_assets_against_the_live_reference(product_io, tmp_path):
    import make_io_golden
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True, capture_output=True)
    ref_io = os.path.join(ROOT, "oracle", "_ref", "ref_io")
    live = make_io_golden.run_all(ref_io, str(tmp_path))
    assert live == json.load(open(GOLDEN)), "tests/golden/io is stale: rerun tests/golden/make_io_golden.py"
    for asset in ("hippo1.obj", "hippo2.obj"):                        # BASELINE configs[0] inputs
        a, b = tmp_path / "a.txt", tmp_path / "b.txt"
        subprocess.run([product_io, "read", os.path.join(REF_ASSETS, asset), str(a)], check=True)
        subprocess.run([ref_io, "read", os.path.join(REF_ASSETS, asset), str(b)], check=True)
        assert a.read_bytes() == b.read_bytes()
        assert a.read_text().splitlines()[0].startswith("ok 1 v %d " % (30519 if asset == "hippo1.obj" else 21935))
        subprocess.run([product_io, "write", os.path.join(REF_ASSETS, asset), str(tmp_path / "pa.obj")], check=True)
        subprocess.run([ref_io, "write", os.path.join(REF_ASSETS, asset), str(tmp_path / "ra.obj")], check=True)
        assert (tmp_path / "pa.obj").read_bytes() == (tmp_path / "ra.obj").read_bytes()
