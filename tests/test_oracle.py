"""CPU (-m "not gpu"): the oracle against everything the reference itself pins for this path
 and against the committed golden fixtures."""
import hashlib
import json
import os

import numpy as np
import pytest

from tests import helpers as H

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_golden.json")))
REF_ASSETS = "/root/reference/assets"


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _load_obj(path):
    v = []
    with open(path) as f:
        for line in f:
            if line.startswith("v "):
                v.append([float(x) for x in line.split()[1:4]])
    return np.array(v, np.float32)


@pytest.mark.skipif(not os.path.isdir(REF_ASSETS), reason="reference assets not present on this box")
def test_sampler_known_answer_on_reference_assets(oracle_mod):
    """doc/Usage.md:80 publishes 5281 kd-tree points for hippo1 @ delta=0.01 (scripts/run-example.sh:68)."""
    h1, h2 = _load_obj(REF_ASSETS + "/hippo1.obj"), _load_obj(REF_ASSETS + "/hippo2.obj")
    assert (len(h1), len(h2)) == (30519, 21935)
    assert len(oracle_mod.sample(h1, 0.01)) == 5281
    assert [len(oracle_mod.sample(h1, 0.01)), len(oracle_mod.sample(h2, 0.01))] == GOLD["hippo_sampler"]["delta_0.01"]
    assert [len(oracle_mod.sample(h1, 0.005)), len(oracle_mod.sample(h2, 0.005))] == GOLD["hippo_sampler"]["delta_0.005"] == [16676, 11960]


def test_sampler_keeps_first_point_per_voxel(oracle_mod):
    rng = np.random.default_rng(1)
    X = rng.uniform(-1, 1, size=(5000, 3)).astype(np.float32)
    delta = np.float32(0.1)
    kept = oracle_mod.sample(X, float(delta))
    vox = np.floor(X * (np.float32(1.0) / delta)).astype(np.int64)
    _, first = np.unique(vox, axis=0, return_index=True)
    assert np.array_equal(kept, X[np.sort(first)])
    assert len(oracle_mod.sample(X[:0], 0.1)) == 0                     # empty input


@pytest.mark.parametrize("overlap,trials", [(0.7, 139), (0.5, 594), (0.8, 72), (0.2, 23966)])
def test_number_of_trials_formula(oracle_mod, overlap, trials):
    """match4pcsBase.hpp:175-185: log(1e-5)/log(1-o^4)/0.3 (P_diameter cancels); SURVEY.md §8c known answers."""
    rng = np.random.default_rng(0)
    X = rng.normal(size=(300, 3)).astype(np.float32)
    m = oracle_mod.Matcher(oracle_mod.make_options(0.05, overlap, 1000))
    m.init(X, X + 0.01)
    assert m.stats().number_of_trials == trials == GOLD["number_of_trials"][str(overlap)]


def test_extract_pairs_equals_reference_test_bruteforce(oracle_mod):
    """tests/pair_extraction.cc:239-314: delta=0.1, overlap 0.5, 200/150 unit-sphere points (< sample_size, so
    only centring), d=0.3 / 0.5, eps = distance_factor*delta: sorted ExtractPairs == sorted brute force of
    tests/testing.h:172-194 (both orderings, |dist - d| <= eps)."""
    from super4pcs_amd import datasets as D
    for rep in range(5):
        P = D.sphere_cloud(200, 100 + rep); Q = D.sphere_cloud(150, 200 + rep)
        m = oracle_mod.Matcher(oracle_mod.make_options(0.1, 0.5, 200))
        m.init(P, Q)
        Qs = m.cloud(1)
        eps = np.float32(2.0) * np.float32(0.1)
        for d in (np.float32(0.3), np.float32(0.5)):
            got = m.extract_pairs(float(d), 0.6, float(eps), 0, 1)
            diff = Qs[:, None, :] - Qs[None, :, :]
            dist = np.sqrt(diff[..., 0] * diff[..., 0] + (diff[..., 1] * diff[..., 1] + diff[..., 2] * diff[..., 2]))
            ok = np.abs(dist.astype(np.float64) - np.float64(d)) <= np.float64(eps)
            want = sorted((int(a), int(b)) for a, b in zip(*np.nonzero(ok)) if a != b)
            assert sorted(map(tuple, got.tolist())) == want
            assert len(want) > 0


def test_kdtree_verify_equals_bruteforce_predicate(oracle_mod):
    """the faithful kd-tree query and the exhaustive predicate must agree on benchmark inputs."""
    P, Q, T = H.small_pair(20000, delta=0.01, seed=3)
    m = H.init_oracle(oracle_mod, P, Q, 0.01, 0.6, 300)
    rng = np.random.default_rng(2)
    Ts = np.stack([np.eye(4, dtype=np.float32)] + [H.random_rigid(rng, 0.05) for _ in range(30)])
    a = m.verify_batch(Ts)
    m.set_mode(True, use_kdtree=False)
    b = m.verify_batch(Ts)
    assert np.array_equal(a, b)


def test_registration_matches_golden_and_ground_truth(oracle_mod):
    g = GOLD["registration"]
    delta, overlap, n_s, seed = g["input"]["delta"], g["input"]["overlap"], g["input"]["sample_size"], g["input"]["seed"]
    P, Q, T_gt = H.small_pair(20000, delta=delta, seed=31)
    assert digest(P) == g["input"]["P_sha256"] and digest(Q) == g["input"]["Q_sha256"]
    m = oracle_mod.Matcher(oracle_mod.make_options(delta, overlap, n_s, seed=seed), keep_trace=True)
    lcp, M, Qt = m.compute_transformation(P, Q)
    s = m.stats()
    tr, _ = m.trace()
    assert (s.n_P, s.n_Q, s.number_of_trials) == (g["n_P"], g["n_Q"], g["number_of_trials"])
    assert float(lcp) == g["lcp"] and s.n_verified == g["candidates_verified"]
    assert np.array_equal(M.reshape(-1), np.array(g["M"], np.float32))
    assert digest(tr) == g["trace_sha256"] and digest(Qt) == g["Qt_sha256"]
    # it is a correct registration: rotation within ~2 degrees of the ground truth, points land on P
    assert np.max(np.abs(M[:3, :3] - T_gt[:3, :3])) < 0.05
    # determinism under a fixed seed; a different seed explores other bases
    m2 = oracle_mod.Matcher(oracle_mod.make_options(delta, overlap, n_s, seed=seed))
    lcp2, M2, _ = m2.compute_transformation(P, Q)
    assert lcp2 == lcp and np.array_equal(M2, M)


def test_stage_vectors_match_golden(oracle_mod):
    g = GOLD["stage"]
    P, Q, _ = H.small_pair(20000, delta=0.01, seed=31)
    m = H.init_oracle(oracle_mod, P, Q, 0.01, 0.6, 200)
    found = False
    for _ in range(30):
        ok, i1, i2, base, bx = m.select_quadrilateral()
        if not ok:
            continue
        d1 = float(np.float32(np.linalg.norm(bx[0] - bx[1]))); d2 = float(np.float32(np.linalg.norm(bx[2] - bx[3])))
        p1 = m.extract_pairs(d1, 0.0, 0.02, 0, 1); p2 = m.extract_pairs(d2, 0.0, 0.02, 2, 3)
        if base.tolist() != g["base"]:
            continue
        found = True
        assert digest(p1) == g["pairs1_sha256"] and digest(p2) == g["pairs2_sha256"]
        quads = m.find_congruent(i1, i2, 0.02, p1, p2)
        assert digest(quads) == g["quads_sha256"]
        nb, per, _, _ = m.try_congruent_set(base, quads)
        assert nb == g["n_verified"] and digest(per) == g["counts_sha256"] and digest(m.ids()) == g["ids_sha256"]
        break
    assert found


def test_empty_and_tiny_inputs(oracle_mod):
    m = oracle_mod.Matcher(oracle_mod.make_options(0.1, 0.5, 200))
    lcp, M, Q = m.compute_transformation(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))
    assert lcp == np.float32(1e9)                                       # kLargeNumber, match4pcsBase.hpp:69-70
    # identical tiny clouds: initial LCP is 1 and no RANSAC step runs (match4pcsBase.hpp:73)
    X = np.random.default_rng(0).normal(size=(50, 3)).astype(np.float32)
    m = oracle_mod.Matcher(oracle_mod.make_options(0.1, 0.5, 200))
    lcp, M, Q = m.compute_transformation(X, X.copy())
    assert lcp == 1.0 and np.array_equal(Q, X)


def test_config1_hippo_fixture_is_what_the_oracle_computes(oracle_mod):
    """The committed config-1 fixture (made from the reference's own run) against the oracle, from the fixture's
    sampled clouds alone: init (shuffle, centring, trials) + all trials."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "hippo_config1.npz"))
    # a cloud that is already one point per voxel passes through the sampler unchanged
    assert np.array_equal(oracle_mod.sample(g["Ps"], 0.01), g["Ps"]) and np.array_equal(oracle_mod.sample(g["Qu"], 0.01), g["Qu"])
    m = oracle_mod.Matcher(oracle_mod.make_options(0.01, 0.7, 200))
    lcp, M, Q = m.compute_transformation(g["Ps"], g["Qu"])
    assert lcp == np.float32(g["lcp"]) and m.stats().n_verified == int(g["n_candidates"])
    T, l2, base, cong, _, _ = m.best()
    assert np.array_equal(T, g["transform"]) and np.array_equal(base, g["base"]) and np.array_equal(cong, g["congruent"])


def test_streaming_quad_count_equals_the_list_enumeration(oracle_mod):
    """oracle.count_congruent (OpenMP, counts + order-independent checksums: the checker for bases whose ~10^9 quads cannot
    be listed) against find_congruent / try_congruent_set, which are pinned to the reference's own sources: same number of
    quads, same checksum, same gate decisions, and the subsample is exactly the gated quads whose mix is 0 mod sample_mod."""
    O = oracle_mod
    delta, overlap, n_s = 0.01, 0.6, 300
    P, Q, _ = H.small_pair(20000, delta=delta, seed=11)
    om = H.init_oracle(O, P, Q, delta, overlap, n_s)
    eps = 2.0 * delta
    seen = 0
    for _ in range(6):
        ok, i1, i2, base, bx = om.select_quadrilateral()
        if not ok:
            continue
        p1 = om.extract_pairs(float(np.float32(np.linalg.norm(bx[0] - bx[1]))), 0.0, eps, 0, 1)
        p2 = om.extract_pairs(float(np.float32(np.linalg.norm(bx[2] - bx[3]))), 0.0, eps, 2, 3)
        if not (len(p1) and len(p2)):
            continue
        quads = om.find_congruent(i1, i2, eps, p1, p2)
        _nb, per, _bc, _bi = om.try_congruent_set(base, quads)
        gated = quads[per >= 0]
        for threads in (1, 4):
            got = om.count_congruent(i1, i2, eps, p1, p2, base=base, threads=threads, sample_mod=7)
            assert (got["K"], got["quad_sum"]) == (len(quads), H.checksum(quads))
            assert (got["C"], got["cand_sum"]) == (len(gated), H.checksum(gated))
            want = gated[H.quad_mix(gated) % np.uint64(7) == 0] if len(gated) else gated
            want = want[np.lexsort(want.T[::-1])] if len(want) else want
            assert np.array_equal(got["sample"], want)
        assert om.L.s4po_quad_mix(1, 2, 3, 4) == int(H.quad_mix([[1, 2, 3, 4]])[0]) == 6837720401966776326
        # ... and the streaming WINNER (every gated candidate verified in full, greatest count, then first in the reference's
        # candidate order) is the one the list form keeps: the first maximum of per (try_congruent_set runs in full-count mode)
        for threads in (1, 4):
            wb = om.count_congruent_best(i1, i2, eps, p1, p2, base, threads=threads)
            assert (wb["K"], wb["C"]) == (len(quads), len(gated))
            if len(gated):
                k_first_max = int(np.argmax(np.where(per >= 0, per, -1)))
                assert wb["found"] and wb["best_count"] == int(per[k_first_max]) and wb["best_quad"] == list(quads[k_first_max])
            else:
                assert not wb["found"]
        seen += len(quads)
    assert seen > 1000


def test_sampler_keeps_the_first_point_of_every_voxel_on_a_scene_of_planes(oracle_mod):
    """sampling.h:59-122 on structured data (planes, boxes, a ground grid -- where the reference's own hash clusters and the
    oracle keeps its voxels under a mixing hash instead): the kept points are exactly the first point, in input order, of every
    delta-voxel, computed here with numpy from the same float32 arithmetic (floor(x * (1 / delta)))."""
    from super4pcs_amd import datasets
    P, _Q, _T = datasets.lidar_pair(300_000, delta=0.05)
    X = np.ascontiguousarray(P, np.float32)
    for delta in (0.05, 0.2):
        scale = np.float32(1.0) / np.float32(delta)
        vox = np.floor(X * scale).astype(np.int64)
        _u, first = np.unique(vox, axis=0, return_index=True)
        want = X[np.sort(first)]
        got = oracle_mod.sample(X, delta)
        assert got.shape == want.shape and np.array_equal(got, want), (delta, got.shape, want.shape)
