"""CPU (-m "not gpu"): the product's host-side structures against the oracle.

PairOctree (super4pcs_amd/csrc/s4p_host_structs.hpp) replays loop 1 of IntersectionFunctor::process with two
shortcuts (remembered cell splits, distance-shell bins for the sphere/cell test); its persistent permutation and
the flattened sequence must stay identical to the oracle's literal restatement over a long series of calls.
FourthPointIndex must return what the literal 4th-point loop returns, ties and empty results included."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


@pytest.fixture(scope="module")
def host_app(tmp_path_factory):
    out = tmp_path_factory.mktemp("host_app") / "host_structs"
    src = os.path.join(ROOT, "tests", "host_app", "host_structs_main.cpp")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "super4pcs_amd", "csrc"),
                    src, "-o", str(out)], check=True)
    return str(out)


@pytest.mark.parametrize("n_s,seed", [(400, 7), (1500, 11)])
def test_pair_octree_matches_oracle_permutation(oracle_mod, host_app, tmp_path, n_s, seed):
    import helpers
    O = oracle_mod
    delta = 0.01
    P, Q, _ = helpers.small_pair(20000, delta=delta, seed=seed)
    om = helpers.init_oracle(O, P, Q, delta, 0.6, n_s)
    q = om.cloud(1)
    rng = np.random.default_rng(seed)
    diam = float(np.linalg.norm(q.max(0) - q.min(0)))
    # realistic base edges, plus tiny and over-sized radii (cells nothing touches) and a coarser epsilon
    calls = [(float(rng.uniform(0.15, 0.6) * diam), 2 * delta) for _ in range(40)]
    calls += [(0.02 * diam, 2 * delta), (3.0 * diam, 2 * delta), (0.3 * diam, 8 * delta), (0.45 * diam, 2 * delta)]
    path = tmp_path / "octree.txt"
    with open(path, "w") as f:
        f.write("%d\n" % q.shape[0])
        for p in q:
            f.write("%.9g %.9g %.9g\n" % (p[0], p[1], p[2]))
        f.write("%d\n" % len(calls))
        for d, e in calls:
            f.write("%.9g %.9g\n" % (np.float32(d), np.float32(e)))
    # synch3DContent (pairCreationFunctor.h:90-122): same centre and ratio, bit for bit
    fr = subprocess.run([host_app, "frame", str(path)], check=True, capture_output=True, text=True).stdout.split()
    _, _, g, ratio = om.frame()
    assert [float.fromhex(v) for v in fr[:3]] == [float(v) for v in g] and float.fromhex(fr[3]) == float(np.float32(ratio))
    out = subprocess.run([host_app, "octree", str(path)], check=True, capture_output=True, text=True).stdout.split("\n")
    n = q.shape[0]
    for c, (d, e) in enumerate(calls):
        pairs = om.extract_pairs(float(np.float32(d)), 0.0, float(np.float32(e)), 0, 1)
        want_ids = om.ids()
        n_seq, n_leaf = (int(v) for v in out[3 * c].split())
        got_ids = np.array(out[3 * c + 1].split(), np.uint32)
        got_seq = np.array(out[3 * c + 2].split(), np.uint32)
        assert got_ids.shape[0] == n and np.array_equal(got_ids, want_ids), "permutation differs after call %d" % c
        assert got_seq.shape[0] == n_seq and n_leaf <= max(n_seq, 1)
        # every point the oracle paired as the octree-side element must be in the flattened sequence
        assert set(np.unique(pairs[:, 0])).issubset(set(got_seq.tolist())) or pairs.shape[0] == 0


@pytest.mark.parametrize("seed", [1, 2])
def test_fourth_point_index_matches_the_literal_loop(host_app, seed):
    out = subprocess.run([host_app, "fourth", str(seed)], check=True, capture_output=True, text=True).stdout
    words = out.split()
    assert words[0] == "queries" and int(words[1]) > 4000 and words[2] == "mismatches" and int(words[3]) == 0, out
