"""Config 1 (BASELINE.json configs[0]): the reference's bundled hippo pair, `-o 0.7 -d 0.01 -n 200`
(scripts/run-example.sh:68).  Needs /root/reference (assets + oracle/_ref built from its sources); writes
tests/golden/hippo_config1.npz with
  * the two clouds after UniformDistSampler (what Match4PCSBase::init feeds to shuffle/centring), so the fixture
    is 110 KB instead of two OBJ files and the GPU box (which has no /root/reference) can replay the registration;
  * what the reference's own ComputeTransformation (oracle/_ref) returns for them: LCP, 4x4, candidates verified.
Run from the repo root:  python tests/golden/make_hippo_fixture.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O, reflib  # noqa: E402


def load_obj_vertices(path):
    v = []
    with open(path) as f:
        for line in f:
            if line.startswith("v "):
                v.append([float(x) for x in line.split()[1:4]])
    return np.array(v, np.float32)


def main():
    ref = "/root/reference/assets"
    P = load_obj_vertices(ref + "/hippo1.obj"); Q = load_obj_vertices(ref + "/hippo2.obj")
    delta, overlap, n_s = 0.01, 0.7, 200
    Ps, Qu = O.sample(P, delta), O.sample(Q, delta)
    assert (len(Ps), len(Qu)) == (5281, 3820)
    reflib.build()
    rm = reflib.RefMatcher(O.make_options(delta, overlap, n_s))
    lcp, M, Qt, n = rm.compute_transformation(P, Q)
    T, l2, base, cong = rm.best()
    np.savez_compressed(os.path.join(os.path.dirname(__file__), "hippo_config1.npz"),
                        Ps=Ps, Qu=Qu, lcp=np.float32(lcp), M=M, n_candidates=np.int64(n), transform=T, base=base, congruent=cong,
                        n_trials=np.int32(rm.stats()["number_of_trials"]))
    print("hippo: lcp %.6f candidates %d trials %d" % (lcp, n, rm.stats()["number_of_trials"]))


if __name__ == "__main__":
    main()
