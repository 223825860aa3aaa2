"""-m gpu: end-to-end parity of the engine (host RANSAC driver + HIP hot path) with the CPU oracle.

Same inputs, same seed -> same sampled clouds, same bases, same per-trial counts, same winning
candidate; final R,t within 1e-4 (BASELINE.json north_star), LCP (integer inliers) exact.
"""
import os

import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


def _run_both(O, P, Q, delta, overlap, n_s, seed=5489, **kw):
    from super4pcs_amd import capi
    om = O.Matcher(O.make_options(delta, overlap, n_s, seed=seed, **kw), full_counts=False, use_kdtree=True, keep_trace=True)
    o_lcp, o_M, o_Q = om.compute_transformation(P, Q)
    gm = capi.Matcher(capi.make_options(delta, overlap, n_s, seed=seed, **kw))
    g_lcp, g_M, g_Q = gm.compute_transformation(P, Q)
    return om, (o_lcp, o_M, o_Q), gm, (g_lcp, g_M, g_Q)


@pytest.mark.parametrize("seed,n_s", [(5489, 300), (17, 200)])
def test_compute_transformation_matches_oracle(oracle_mod, s4p_lib_built, seed, n_s):
    delta, overlap = 0.01, 0.6
    P, Q, T_gt = H.small_pair(30000, delta=delta, seed=23)
    om, (o_lcp, o_M, o_Q), gm, (g_lcp, g_M, g_Q) = _run_both(oracle_mod, P, Q, delta, overlap, n_s, seed=seed)
    os_, gi = om.stats(), gm.info()
    assert (gi.n_sampled_p, gi.n_sampled_q, gi.number_of_trials) == (os_.n_P, os_.n_Q, os_.number_of_trials)
    assert gi.candidates_verified == os_.n_verified          # candidate transforms verified (the metric's unit)
    assert gi.quads_total == os_.n_quads and gi.pairs_total == os_.n_pairs
    assert g_lcp == o_lcp                                     # integer inlier count / n_Q: exact
    assert np.max(np.abs(g_M - o_M)) <= 1e-4
    assert np.array_equal(g_M, o_M)                           # in fact the same IEEE operations on both sides
    assert np.max(np.abs(g_Q - o_Q)) <= 1e-4
    # and the registration is a real one: close to the ground-truth motion
    assert g_lcp > 0.35
    assert np.max(np.abs(g_M[:3, :3] - T_gt[:3, :3])) < 0.05


def test_stepwise_trace_matches_oracle(oracle_mod, s4p_lib_built):
    from super4pcs_amd import capi
    O = oracle_mod
    delta, overlap, n_s = 0.01, 0.5, 250
    P, Q, _ = H.small_pair(30000, delta=delta, seed=5)
    om = O.Matcher(O.make_options(delta, overlap, n_s), full_counts=False, use_kdtree=True, keep_trace=True)
    om.init(P, Q)
    gm = capi.Matcher(capi.make_options(delta, overlap, n_s))
    gm.init_full(P, Q)
    assert np.array_equal(gm.sampled(0), om.cloud(0))
    assert np.array_equal(gm.sampled(1), om.cloud(1))
    assert gm.info().best_lcp == om.stats().best_lcp           # initial LCP = Verify(identity)
    for t in range(25):
        o_ok = om.try_one_base()
        g_ok, r = gm.try_one_base()
        tr, inv = om.trace()
        rec = tr[-1]
        assert g_ok == o_ok
        if rec[0]:
            assert (r.n_pairs1, r.n_pairs2) == (rec[5], rec[6])
            if rec[5] and rec[6]:
                assert (r.n_quads, r.n_verified) == (rec[7], rec[8])
        assert gm.info().best_lcp == om.stats().best_lcp
    T, lcp, base, cong, c1, c2 = om.best()
    gi = gm.info()
    assert list(gi.base) == base.tolist() and list(gi.congruent) == cong.tolist()
    assert np.array_equal(np.array(gi.transform, np.float32).reshape(4, 4), T)


def test_transform_points_matches_oracle_order(oracle_mod, s4p_lib_built):
    from super4pcs_amd import capi
    rng = np.random.default_rng(9)
    ctx = capi.Context(capi.make_options(0.01, 0.5, 200))
    M = H.random_rigid(rng, 0.3)
    for n in (100003, 131072, 300007, 1):                         # one chunk, exactly one, three with a ragged tail, one point
        X = rng.normal(size=(n, 3)).astype(np.float32)
        got = ctx.transform_points(M, X)
        x, y, z = X[:, 0], X[:, 1], X[:, 2]
        want = np.stack([((M[r, 0] * x + M[r, 1] * y) + M[r, 2] * z) + M[r, 3] for r in range(3)], axis=1)
        assert np.array_equal(got, want), n


def test_options_rejected_loudly(s4p_lib_built):
    from super4pcs_amd import capi
    with pytest.raises(capi.S4PError) as e:
        capi.Context(capi.make_options(-1.0, 0.5, 200))
    assert e.value.code == -1


@pytest.mark.parametrize("max_angle,tol", [(30.0, None), (12.0, None), (30.0, "0.02")])
def test_max_angle_matches_oracle(oracle_mod, s4p_lib_built, max_angle, tol, monkeypatch):
    """options.max_angle on the device path: the segment-angle pair filter (pairCreationFunctor.h:203-212; an exact cosine
    threshold) and the Euler-angle bound of ComputeRigidTransformation (match4pcsBase.cc:457-472; decided on the device up
    to a margin, settled by the host with libm inside it) -- ordered pair lists, per-quad gate decisions and inlier counts,
    and the whole registration against the oracle, which is pinned to the reference's own code for this option
    (tests/test_oracle_vs_reference.py::test_max_angle_matches_reference).  tol = 0.02: the device margin widened 20 000 x,
    so that hundreds of candidates take the host route instead of one in a million -- same results."""
    from super4pcs_amd import capi
    if tol:
        monkeypatch.setenv("S4P_ANGLE_TOL", tol)
    O = oracle_mod
    delta, overlap, n_s = 0.01, 0.6, 200
    P, Q, _ = H.small_rotation_pair(20000, delta=delta, seed=31)
    oopt = O.make_options(delta, overlap, n_s, max_angle=max_angle)
    gopt = capi.make_options(delta, overlap, n_s, max_angle=max_angle)
    om = O.Matcher(oopt, full_counts=True, use_kdtree=True)
    om.init(P, Q)
    ctx = capi.Context(gopt)
    ctx.set_clouds(om.cloud(0), om.cloud(1))
    eps = 2.0 * delta
    n_pairs = n_gated = n_rejected = 0
    for _ in range(10):
        ok, i1, i2, base, bx = om.select_quadrilateral()
        if not ok:
            continue
        ctx.set_base(bx)
        sets = []
        for a, b in ((0, 1), (2, 3)):
            d = float(np.float32(np.linalg.norm(bx[a] - bx[b])))
            want, got = om.extract_pairs(d, 0.0, eps, a, b), ctx.extract_pairs(d, 0.0, eps, a, b)
            assert np.array_equal(got, want)                 # the filter drops ordered pairs one by one: same list, same order
            sets.append(want)
            n_pairs += len(want)
        if not (len(sets[0]) and len(sets[1])):
            continue
        quads = om.find_congruent(i1, i2, eps, sets[0], sets[1])
        assert np.array_equal(ctx.find_congruent(i1, i2, eps, sets[0], sets[1]), quads)
        if len(quads):
            _nb, per, _bc, _bi = om.try_congruent_set(base, quads)
            r, g_per = ctx.try_congruent_set(base, quads)
            assert np.array_equal(g_per, per)
            assert r.n_verified == int((per >= 0).sum())
            n_gated += int((per >= 0).sum()); n_rejected += int((per < 0).sum())
    assert n_pairs > 0 and n_rejected > 0
    om2 = O.Matcher(oopt)
    o_lcp, o_M, _ = om2.compute_transformation(P, Q)
    gm = capi.Matcher(gopt)
    g_lcp, g_M, _ = gm.compute_transformation(P, Q)
    assert g_lcp == o_lcp and np.array_equal(g_M, o_M)
    gi, os_ = gm.info(), om2.stats()
    assert (gi.pairs_total, gi.quads_total, gi.candidates_verified) == (os_.n_pairs, os_.n_quads, os_.n_verified)
    assert os_.n_verified > 0
    if tol:
        settled, rejected = gm.border_stats()
        assert settled > 20 and 0 < rejected < settled       # the host route was taken, both ways


def test_empty_inputs_return_large_number(s4p_lib_built):
    """match4pcsBase.hpp:69-70 / tests/externalAppTest/main.cpp: empty sets -> kLargeNumber."""
    from super4pcs_amd import capi
    m = capi.Matcher(capi.make_options(0.01, 0.5, 200))
    lcp, M, Q = m.compute_transformation(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))
    assert lcp == np.float32(1e9)


def test_gpu_against_committed_golden_vectors(s4p_lib_built):
    """GPU path vs tests/golden/oracle_golden.json (no live oracle involved): registration result and
    the stage vectors (pairs / quads / per-candidate counts) of one base."""
    import hashlib
    import json
    import os
    from super4pcs_amd import capi
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_golden.json")))

    def digest(a):
        return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()

    g = gold["registration"]
    P, Q, _ = H.small_pair(20000, delta=g["input"]["delta"], seed=31)
    assert digest(P) == g["input"]["P_sha256"]
    m = capi.Matcher(capi.make_options(g["input"]["delta"], g["input"]["overlap"], g["input"]["sample_size"], seed=g["input"]["seed"]))
    lcp, M, Qt = m.compute_transformation(P, Q)
    i = m.info()
    assert (i.n_sampled_p, i.n_sampled_q, i.number_of_trials) == (g["n_P"], g["n_Q"], g["number_of_trials"])
    assert float(lcp) == g["lcp"] and i.best_count == g["best_count"]
    assert i.candidates_verified == g["candidates_verified"] and i.quads_total == g["quads"] and i.pairs_total == g["pairs"]
    assert np.array_equal(M.reshape(-1), np.array(g["M"], np.float32))
    assert digest(Qt) == g["Qt_sha256"]
    # stage vectors
    s = gold["stage"]
    m2 = capi.Matcher(capi.make_options(g["input"]["delta"], g["input"]["overlap"], g["input"]["sample_size"], seed=g["input"]["seed"]))
    m2.init_full(P, Q)
    ctx = capi.Context(capi.make_options(g["input"]["delta"], g["input"]["overlap"], g["input"]["sample_size"]))
    ctx.set_clouds(m2.sampled(0), m2.sampled(1))
    hit = False
    for _ in range(30):
        found, i1, i2, base, bx = m2.select_quadrilateral()
        if not found:
            continue
        ctx.set_base(bx)
        d1 = float(np.float32(np.linalg.norm(bx[0] - bx[1]))); d2 = float(np.float32(np.linalg.norm(bx[2] - bx[3])))
        p1 = ctx.extract_pairs(d1, 0.0, 0.02, 0, 1); p2 = ctx.extract_pairs(d2, 0.0, 0.02, 2, 3)
        if base.tolist() != s["base"]:
            continue
        hit = True
        assert digest(p1) == s["pairs1_sha256"] and digest(p2) == s["pairs2_sha256"]
        quads = ctx.find_congruent(i1, i2, 0.02, p1, p2)
        assert digest(quads) == s["quads_sha256"]
        r, per = ctx.try_congruent_set(base, quads)
        assert r.n_verified == s["n_verified"] and digest(per) == s["counts_sha256"] and r.best_count == s["max_count"]
        break
    assert hit


def test_gpu_against_the_reference_sources_directly(s4p_lib_built):
    """GPU path vs oracle/_ref (the reference's own match4pcsBase.cc / super4pcs.cc compiled against the Eigen
    shim): same LCP (integer inliers), same number of verified candidates, R and t within 1e-4."""
    reflib = pytest.importorskip("oracle.reflib")
    if not reflib.available():
        pytest.skip("oracle/_ref/libs4p_ref.so not in the tree")
    from oracle import oracle as O
    from super4pcs_amd import capi
    delta, overlap, n_s = 0.01, 0.6, 200
    for seed, cloud_seed in ((5489, 31), (7, 12)):
        P, Q, _ = H.small_pair(20000, delta=delta, seed=cloud_seed)
        rm = reflib.RefMatcher(O.make_options(delta, overlap, n_s, seed=seed))
        r_lcp, r_M, r_Q, r_n = rm.compute_transformation(P, Q)
        gm = capi.Matcher(capi.make_options(delta, overlap, n_s, seed=seed))
        g_lcp, g_M, g_Q = gm.compute_transformation(P, Q)
        assert g_lcp == r_lcp
        assert gm.info().candidates_verified == r_n
        assert np.array_equal(g_M[:3, :3], r_M[:3, :3])
        assert np.max(np.abs(g_M - r_M)) <= 1e-4 and np.max(np.abs(g_Q - r_Q)) <= 1e-4
        rT, rl, rb, rc = rm.best()
        gi = gm.info()
        assert list(gi.base) == rb.tolist() and list(gi.congruent) == rc.tolist()
        assert np.array_equal(np.array(gi.transform, np.float32).reshape(4, 4), rT)


@pytest.mark.parametrize("n_s", [200, 300])
def test_producer_threads_do_not_change_the_result(oracle_mod, s4p_lib_built, n_s):
    """Base selection + octree staging on helper threads (s4p_matcher_set_sharding): same registration, and the
    speculative run-ahead is rewound exactly (a second Perform_N_steps continues like the sequential code)."""
    from super4pcs_amd import capi
    O = oracle_mod
    delta, overlap = 0.01, 0.6
    P, Q, _ = H.small_pair(30000, delta=delta, seed=23)
    om = O.Matcher(O.make_options(delta, overlap, n_s), keep_trace=True)
    om.init(P, Q)
    gm = capi.Matcher(capi.make_options(delta, overlap, n_s))
    gm.set_sharding(0, 1, True)
    gm.init_full(P, Q)
    # two chunks of trials: the producer is stopped and rewound between them
    for chunk in (7, 9):
        for _ in range(chunk):
            om.try_one_base()
        gm.perform_n_steps(chunk)
        assert gm.info().best_lcp == om.stats().best_lcp
        assert gm.info().candidates_verified == om.stats().n_verified
    # and step-wise calls through the queues
    for _ in range(5):
        o_ok = om.try_one_base()
        g_ok, r = gm.try_one_base()
        assert g_ok == o_ok and gm.info().candidates_verified == om.stats().n_verified
    T, lcp, base, cong, c1, c2 = om.best()
    gi = gm.info()
    assert list(gi.base) == base.tolist() and list(gi.congruent) == cong.tolist()
    assert np.array_equal(np.array(gi.transform, np.float32).reshape(4, 4), T)


@pytest.mark.parametrize("opts", [dict(max_normal_difference=20.0), dict(max_color_distance=0.3),
                                  dict(max_translation_distance=0.25), dict(max_normal_difference=30.0, max_color_distance=0.5)])
def test_attribute_filters_on_gpu_match_oracle(oracle_mod, s4p_lib_built, opts):
    """-a / -c / max_translation_distance (pairCreationFunctor.h:166-200) evaluated in k_pairs: same pair counts,
    same candidates, same registration as the CPU path (which is itself pinned to the reference sources)."""
    from super4pcs_amd import capi
    O = oracle_mod
    delta, overlap, n_s = 0.01, 0.6, 200
    P, Q, T_gt = H.small_pair(20000, delta=delta, seed=31)
    Pn, Pc, Qn, Qc = H.attributes_for(P, Q, T_gt)
    om = O.Matcher(O.make_options(delta, overlap, n_s, **opts))
    o_lcp, o_M, o_Q = om.compute_transformation(P, Q, Pn, Pc, Qn, Qc)
    gm = capi.Matcher(capi.make_options(delta, overlap, n_s, **opts))
    g_lcp, g_M, g_Q = gm.compute_transformation(P, Q, Pn, Pc, Qn, Qc)
    gi, os_ = gm.info(), om.stats()
    assert gi.pairs_total == os_.n_pairs and gi.quads_total == os_.n_quads and gi.candidates_verified == os_.n_verified
    assert g_lcp == o_lcp and np.array_equal(g_M, o_M) and np.max(np.abs(g_Q - o_Q)) <= 1e-4
    # the filters really filter: fewer pairs than the unfiltered run
    om0 = O.Matcher(O.make_options(delta, overlap, n_s))
    om0.compute_transformation(P, Q)
    assert os_.n_pairs < om0.stats().n_pairs


def test_small_clouds_use_whole_cloud_without_shuffle(oracle_mod, s4p_lib_built):
    """|P|,|Q| <= sample_size: no sampling, no shuffle (match4pcsBase.hpp:115-119,134-138) -- the reference's own
    pair_extraction test runs in this regime (200/150 points)."""
    from super4pcs_amd import capi, datasets
    O = oracle_mod
    P = datasets.sphere_cloud(200, 1)
    Q = (datasets.sphere_cloud(150, 2) * np.float32(1.0)).astype(np.float32)
    om = O.Matcher(O.make_options(0.1, 0.5, 200), keep_trace=True)
    gm = capi.Matcher(capi.make_options(0.1, 0.5, 200))
    om.init(P, Q)
    gm.init_full(P, Q)
    assert np.array_equal(gm.sampled(0), om.cloud(0)) and np.array_equal(gm.sampled(1), om.cloud(1))
    for _ in range(6):
        o_ok = om.try_one_base()
        g_ok, r = gm.try_one_base()
        assert g_ok == o_ok
        assert gm.info().best_lcp == om.stats().best_lcp and gm.info().candidates_verified == om.stats().n_verified


def test_config1_hippo_pair_matches_reference_run(s4p_lib_built):
    """BASELINE.json configs[0]: hippo1 <-> hippo2, -o 0.7 -d 0.01 -n 200 (scripts/run-example.sh:68).  The fixture
    holds the sampler outputs and the result of the reference's own ComputeTransformation (oracle/_ref)."""
    import os
    from super4pcs_amd import capi
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "hippo_config1.npz"))
    assert g["Ps"].shape[0] == 5281                       # doc/Usage.md:80
    m = capi.Matcher(capi.make_options(0.01, 0.7, 200))
    m.init_sampled(g["Ps"], g["Qu"], q_needs_shuffle=True)
    i = m.info()
    assert (i.n_sampled_p, i.n_sampled_q, i.number_of_trials) == (5281, 200, int(g["n_trials"])) == (5281, 200, 139)
    M, improved, done = m.perform_n_steps(i.number_of_trials)
    i = m.info()
    assert i.best_lcp == np.float32(g["lcp"]) and i.candidates_verified == int(g["n_candidates"])
    assert list(i.base) == g["base"].tolist() and list(i.congruent) == g["congruent"].tolist()
    assert np.array_equal(np.array(i.transform, np.float32).reshape(4, 4), g["transform"])
    assert np.array_equal(M[:3, :3], g["M"][:3, :3]) and np.max(np.abs(M - g["M"])) <= 1e-4


@pytest.mark.parametrize("chunked", [False, True])
def test_per_candidate_visitor_calls(oracle_mod, s4p_lib_built, chunked):
    """match4pcsBase.hpp:458-465: the visitor sees every verified candidate (fraction == -1) in candidate order -- also when a
    base takes several device passes (quad buffers of 1500 entries that may not grow: the passes are then cut along the
    set-1 order key, so the records come out in reference order chunk by chunk)."""
    from super4pcs_amd import capi
    O = oracle_mod
    delta, overlap, n_s = 0.01, 0.6, 200
    P, Q, _ = H.small_pair(20000, delta=delta, seed=31)
    om = H.init_oracle(O, P, Q, delta, overlap, n_s)          # full counts
    gm = capi.Matcher(capi.make_options(delta, overlap, n_s), **({"max_quads": 1500} if chunked else {}))
    if chunked:
        gm.set_quad_chunking(True, 1500)
    gm.init_full(P, Q)
    gm.visit_candidates(True)
    seen, per_trial = [], []

    def vis(fraction, lcp, T):
        if fraction < 0:
            seen.append((np.float32(lcp), T.copy()))
        else:
            per_trial.append(fraction)
    n_trials = 8
    gm.perform_n_steps(n_trials, visitor=vis)
    want = []
    for _ in range(n_trials):
        ok, i1, i2, base, bx = om.select_quadrilateral()
        if not ok:
            continue
        d1 = float(np.float32(np.linalg.norm(bx[0] - bx[1]))); d2 = float(np.float32(np.linalg.norm(bx[2] - bx[3])))
        p1 = om.extract_pairs(d1, 0.0, 2 * delta, 0, 1); p2 = om.extract_pairs(d2, 0.0, 2 * delta, 2, 3)
        if len(p1) == 0 or len(p2) == 0:
            continue
        quads = om.find_congruent(i1, i2, 2 * delta, p1, p2)
        if len(quads) == 0:
            continue
        nb, per, _, _ = om.try_congruent_set(base, quads)
        for k in np.nonzero(per >= 0)[0]:
            okk, rms, T = om.compute_rigid(base, quads[k])
            want.append((np.float32(per[k]) / np.float32(n_s), T))
    assert len(seen) == len(want) == gm.info().candidates_verified and len(seen) > 50
    assert len(per_trial) == n_trials + 1                      # v(0, ...) once, then once per trial
    for (gl, gT), (wl, wT) in zip(seen, want):
        assert gl == wl and np.array_equal(gT, wT)
    if chunked:
        st = gm.chunk_stats()
        assert st["bases"] >= 3 and st["passes"] >= 2 * st["bases"]


def test_records_of_a_multi_pass_base_through_the_stage_calls(oracle_mod, s4p_lib_built):
    """s4p_last_candidates / s4p_last_verified after a base that was processed in chunks (replayed once in reference-ordered
    chunks when nobody asked for the records beforehand, kept on the fly with s4p_keep_candidate_records), and
    s4p_try_congruent_set with more quads than the lane's buffers hold (scored in slices): the reference's std::vectors simply
    grow (super4pcs.cc:166-174, match4pcsBase.hpp:340-351, :458-465), so none of these may refuse."""
    from super4pcs_amd import capi
    O = oracle_mod
    delta, overlap, n_s = 0.01, 0.6, 300
    P, Q, _ = H.small_pair(25000, delta=delta, seed=17)
    m = H.init_oracle(O, P, Q, delta, overlap, n_s)
    big = capi.Context(capi.make_options(delta, overlap, n_s))
    small = capi.Context(capi.make_options(delta, overlap, n_s), max_quads=2000)
    small.set_quad_chunking(True, 2000)
    kept = capi.Context(capi.make_options(delta, overlap, n_s), max_quads=2000)
    kept.set_quad_chunking(True, 2000)
    kept.keep_candidate_records(True)
    sunk = capi.Context(capi.make_options(delta, overlap, n_s), max_quads=2000)
    sunk.set_quad_chunking(True, 2000)
    # the DEFAULT growth policy: small buffers, grow cap left at 32 Mi -- after a chunked base the lane widens, so
    # the replay for its records fits one pass and must be read back as a single-pass base (it used to report zero records)
    grown = capi.Context(capi.make_options(delta, overlap, n_s), max_quads=2000)
    for c in (big, small, kept, sunk, grown):
        c.set_clouds(m.cloud(0), m.cloud(1))
    checked = 0
    for t in range(10):
        ok, i1, i2, base, bx = m.select_quadrilateral()
        if not ok:
            continue
        eps = 2.0 * delta
        d1 = float(np.float32(np.linalg.norm(bx[0] - bx[1]))); d2 = float(np.float32(np.linalg.norm(bx[2] - bx[3])))
        p1 = m.extract_pairs(d1, 0.0, eps, 0, 1); p2 = m.extract_pairs(d2, 0.0, eps, 2, 3)
        got = []
        sunk.set_candidate_sink(lambda cnt, T: got.append((cnt, T)))
        res = []
        for c in (big, small, kept, sunk, grown):
            c.set_base(bx)
            res.append(c.try_base(base, i1, i2))
        sunk.set_candidate_sink(None)
        rb, rs, rk, rn, rg = res
        for r in (rs, rk, rn, rg):
            assert (r.n_quads, r.n_verified, r.quad_checksum, r.cand_checksum) == (rb.n_quads, rb.n_verified, rb.quad_checksum, rb.cand_checksum)
            assert (r.has_best, r.best_count, list(r.best_quad)) == (rb.has_best, rb.best_count, list(rb.best_quad))
        if rb.n_quads <= 2000:
            continue
        checked += 1
        bq, bc = big.last_candidates(rb.n_quads)
        bv, bT = big.last_verified(max(rb.n_verified, 1))
        for c in (small, kept, grown):                           # replay in chunks / kept on the fly / replay in ONE pass after the growth
            q_, c_ = c.last_candidates(rb.n_quads)
            assert np.array_equal(q_, bq) and np.array_equal(c_, bc)
            v_, T_ = c.last_verified(max(rb.n_verified, 1))
            assert np.array_equal(v_, bv) and np.array_equal(T_, bT)
        assert len(got) >= 2                                     # the sink saw the base pass by pass, in reference order
        assert np.array_equal(np.concatenate([g[0] for g in got]), bv) and np.array_equal(np.concatenate([g[1] for g in got]), bT)
        # a caller's quad list longer than the buffers: slices
        rr_b, per_b = big.try_congruent_set(base, bq)
        rr_s, per_s = small.try_congruent_set(base, bq)
        assert np.array_equal(per_b, per_s) and np.array_equal(per_b, bc)
        assert (rr_s.n_verified, rr_s.best_count, list(rr_s.best_quad)) == (rr_b.n_verified, rr_b.best_count, list(rr_b.best_quad))
        assert np.array_equal(np.frombuffer(bytes(rr_s.best_transform), np.float32), np.frombuffer(bytes(rr_b.best_transform), np.float32))
        v2, T2 = small.last_verified(max(rb.n_verified, 1))
        assert np.array_equal(v2, bv) and np.array_equal(T2, bT)
        # FindCongruentQuadrilaterals returning a list longer than the device buffers (super4pcs.cc:166-174): enumerated in
        # ranges of the first pair set, in the reference's std::set order
        sq = small.find_congruent(i1, i2, eps, p1, p2, cap=len(bq) + 8)
        assert np.array_equal(sq, bq)
    assert checked >= 2


@pytest.mark.parametrize("producer", [False, True])
def test_early_termination_is_exact(oracle_mod, s4p_lib_built, producer):
    """terminate_threshold < 1 (configureOverlap's second argument): the loop stops at the first trial whose best LCP
    exceeds it (match4pcsBase.hpp:255).  The pipelined engine has speculative bases in flight at that moment; they
    must not leak into the result, the candidate count, or the host state a following Perform_N_steps starts from."""
    from super4pcs_amd import capi
    O = oracle_mod
    delta, overlap, n_s, thr = 0.01, 0.6, 200, 0.45
    P, Q, _ = H.small_pair(20000, delta=delta, seed=31)
    om = O.Matcher(O.make_options(delta, overlap, n_s, terminate_threshold=thr))
    o_lcp, o_M, o_Q = om.compute_transformation(P, Q)
    gm = capi.Matcher(capi.make_options(delta, overlap, n_s, terminate_threshold=thr))
    if producer:
        gm.set_sharding(0, 1, True)
    g_lcp, g_M, g_Q = gm.compute_transformation(P, Q)
    assert o_lcp > thr and g_lcp == o_lcp and np.array_equal(g_M, o_M)
    assert gm.info().candidates_verified == om.stats().n_verified
    assert om.stats().n_verified < 571588                     # it really stopped early (the full run verifies 571 588)
    # continue both for a few more trials: same bases, same state afterwards
    for _ in range(4):
        o_ok = om.try_one_base()
        g_ok, _r = gm.try_one_base()
        assert g_ok == o_ok
    assert gm.info().candidates_verified == om.stats().n_verified and gm.info().best_lcp == om.stats().best_lcp


@pytest.mark.parametrize("producer", [False, True])
def test_capacity_growth_replays_the_overflowing_base(oracle_mod, s4p_lib_built, producer):
    """Device buffers far too small for a base (256 pairs / 256 quads): ComputeTransformation rolls the speculation back
    to just before the overflowing base, grows the buffers to what the base's counters ask for and resumes -- the same
    registration as the oracle's (whose std::vectors simply grow), to the candidate.  With growth switched off the
    same run fails loudly."""
    from super4pcs_amd import capi
    O = oracle_mod
    delta, overlap, n_s = 0.01, 0.6, 200
    P, Q, _ = H.small_pair(20000, delta=delta, seed=3)
    om = O.Matcher(O.make_options(delta, overlap, n_s), full_counts=False, use_kdtree=True)
    o_lcp, o_M, _ = om.compute_transformation(P, Q)
    gm = capi.Matcher(capi.make_options(delta, overlap, n_s), max_pairs=256, max_quads=256)
    if producer:
        gm.set_sharding(0, 1, True)
    g_lcp, g_M, _ = gm.compute_transformation(P, Q)
    assert g_lcp == o_lcp and np.max(np.abs(g_M - o_M)) <= 1e-4
    assert gm.info().candidates_verified == om.stats().n_verified
    assert gm.capacity_growths() >= 1
    mp, mq = gm.limits()
    assert mp > 256 and mq > 256
    strict = capi.Matcher(capi.make_options(delta, overlap, n_s), max_pairs=256, max_quads=256)
    strict.grow_on_overflow(False)
    with pytest.raises(capi.S4PError) as e:
        strict.compute_transformation(P, Q)
    assert e.value.code == capi.S4P_ERR_CAPACITY


def test_chunked_bases_equal_the_unbounded_registration(oracle_mod, s4p_lib_built):
    """Quad buffers of 1500 entries that may not grow (chunk cap = 1500) against bases with tens of thousands of congruent
    quads: every such base is enumerated, gated and scored in CHUNKS (ranges of the second pair set), the chunk bests folded
    with the first-maximum rule -- and the registration equals the oracle's (whose std::vectors simply grow) to the
    candidate, as do the per-base quad / candidate counts and their order-independent checksums."""
    from super4pcs_amd import capi
    from bench import seg_len32
    O = oracle_mod
    delta, overlap, n_s = 0.01, 0.6, 200
    P, Q, _ = H.small_pair(20000, delta=delta, seed=3)
    om = O.Matcher(O.make_options(delta, overlap, n_s), full_counts=False, use_kdtree=True)
    o_lcp, o_M, _ = om.compute_transformation(P, Q)
    gm = capi.Matcher(capi.make_options(delta, overlap, n_s), max_quads=1500)
    gm.set_quad_chunking(True, 1500)
    g_lcp, g_M, _ = gm.compute_transformation(P, Q)
    assert g_lcp == o_lcp and np.array_equal(g_M, o_M)
    gi, os_ = gm.info(), om.stats()
    assert (gi.pairs_total, gi.quads_total, gi.candidates_verified) == (os_.n_pairs, os_.n_quads, os_.n_verified)
    st = gm.chunk_stats()
    assert st["bases"] > 10 and st["passes"] >= 2 * st["bases"] and gm.limits()[1] == 1500
    # base by base: counts and checksums of chunked bases against the oracle's lists
    of = O.Matcher(O.make_options(delta, overlap, n_s), full_counts=True, use_kdtree=True)
    of.init(P, Q)
    g2 = capi.Matcher(capi.make_options(delta, overlap, n_s), max_quads=1500)
    g2.set_quad_chunking(True, 1500)
    g2.init_full(P, Q)
    eps = 2.0 * delta
    chunked = 0
    for _ in range(12):
        before = g2.chunk_stats()["bases"]
        _ok, r = g2.try_one_base()
        ok, i1, i2, base, bx = of.select_quadrilateral()
        if not ok:
            continue
        p1 = of.extract_pairs(seg_len32(bx[0], bx[1]), 0.0, eps, 0, 1)
        p2 = of.extract_pairs(seg_len32(bx[2], bx[3]), 0.0, eps, 2, 3)
        assert (r.n_pairs1, r.n_pairs2) == (len(p1), len(p2))
        if not (len(p1) and len(p2)):
            continue
        quads = of.find_congruent(i1, i2, eps, p1, p2)
        _nb, per, _bc, _bi = of.try_congruent_set(base, quads)
        assert (r.n_quads, r.quad_checksum) == (len(quads), H.checksum(quads))
        assert (r.n_verified, r.cand_checksum) == (int((per >= 0).sum()), H.checksum(quads[per >= 0]))
        if r.n_verified:
            k = int(np.flatnonzero(per == per.max())[0])           # first maximum in std::set order
            assert r.best_count == per.max() and list(r.best_quad) == quads[k].tolist()
        if g2.chunk_stats()["bases"] > before:
            chunked += 1
            # the per-candidate records of a chunked base: replayed once in reference-ordered chunks (round 4; rounds 1-3 refused)
            gq, gc = g2.last_candidates(len(quads))
            assert np.array_equal(gq, quads) and np.array_equal(gc, per)
    assert chunked >= 3
    # chunking off: the same base fails loudly
    strict = capi.Matcher(capi.make_options(delta, overlap, n_s), max_quads=1500)
    strict.set_quad_chunking(False)
    strict.grow_on_overflow(False)
    with pytest.raises(capi.S4PError) as e:
        strict.compute_transformation(P, Q)
    assert e.value.code == capi.S4P_ERR_CAPACITY


def test_early_exit_changes_no_result(oracle_mod, s4p_lib_built):
    """The trial loops let the device abandon candidates that can no longer EXCEED the best inlier count so far (the reference's
    Verify early exit, match4pcsBase.cc:520,558-560).  With it on (the default) and off: same best LCP, winner, transform and
    number of verified candidates as the oracle -- and it really prunes.  Outside a loop (try_one_base) counts stay full."""
    from super4pcs_amd import capi
    O = oracle_mod
    delta, overlap, n_s = 0.01, 0.6, 300
    P, Q, _ = H.small_pair(20000, delta=delta, seed=5)
    om = O.Matcher(O.make_options(delta, overlap, n_s), full_counts=False, use_kdtree=True)
    o_lcp, o_M, _ = om.compute_transformation(P, Q)
    res = {}
    for on in (True, False):
        gm = capi.Matcher(capi.make_options(delta, overlap, n_s))
        gm.early_exit(on)
        gm.profile_enable(True, False)
        lcp, M, _ = gm.compute_transformation(P, Q)
        i = gm.info()
        res[on] = (lcp, M.tobytes(), tuple(i.base), tuple(i.congruent), i.candidates_verified, i.quads_total)
        pruned = gm.profile_get().verify_pruned
        assert (pruned > 0.3 * i.candidates_verified) if on else (pruned == 0)
        assert lcp == o_lcp and np.array_equal(M, o_M) and i.candidates_verified == om.stats().n_verified
    assert res[True] == res[False]
    # stage-level / single-base calls are never pruned: full per-candidate counts against the oracle
    of = O.Matcher(O.make_options(delta, overlap, n_s), full_counts=True, use_kdtree=True)
    of.init(P, Q)
    gm = capi.Matcher(capi.make_options(delta, overlap, n_s))
    gm.init_full(P, Q)
    gm.perform_n_steps(3)                                    # a loop ran (and left): the hint must be gone afterwards
    for _ in range(3):
        of.try_one_base()
    _ok, r = gm.try_one_base()
    g_quads, g_counts = gm.last_candidates(r.n_quads)
    ok, i1, i2, base, bx = of.select_quadrilateral()
    assert ok
    from bench import seg_len32
    p1 = of.extract_pairs(seg_len32(bx[0], bx[1]), 0.0, 2 * delta, 0, 1); p2 = of.extract_pairs(seg_len32(bx[2], bx[3]), 0.0, 2 * delta, 2, 3)
    quads = of.find_congruent(i1, i2, 2 * delta, p1, p2)
    _nb, per, _bc, _bi = of.try_congruent_set(base, quads)
    assert np.array_equal(quads, g_quads) and np.array_equal(per, g_counts)


@pytest.mark.parametrize("parts,chunk", [(2, False), (3, False), (2, True)])
def test_quad_slices_are_a_partition_of_the_base(oracle_mod, s4p_lib_built, parts, chunk):
    """s4p_set_quad_slice (one base over several GPUs): the shares are defined on the pairs' order keys, so whatever order
    each context's pair kernel appended them in, every quad and every candidate of a base belongs to exactly one share --
    counts and order-independent checksums of the shares add up to the oracle's lists (checksums modulo 2^64), and the best
    of the shares' winners under (count, smallest tag) is the base's first maximum.  With chunk=True the shares are chunked
    on top (quad buffers of 600 entries that may not grow)."""
    from super4pcs_amd import capi
    from bench import seg_len32
    O = oracle_mod
    delta, overlap, n_s = 0.01, 0.6, 200
    P, Q, _ = H.small_pair(20000, delta=delta, seed=3)
    of = O.Matcher(O.make_options(delta, overlap, n_s), full_counts=True, use_kdtree=True)
    of.init(P, Q)
    ms = []
    for k in range(parts):
        g = capi.Matcher(capi.make_options(delta, overlap, n_s), **({"max_quads": 600} if chunk else {}))
        if chunk:
            g.set_quad_chunking(True, 600)
        g.early_exit(False)
        g.init_full(P, Q)
        g.set_quad_slice(k, parts)
        ms.append(g)
    eps = 2.0 * delta
    seen = 0
    M64 = (1 << 64) - 1
    for _ in range(10):
        rs = [g.try_one_base()[1] for g in ms]
        ok, i1, i2, base, bx = of.select_quadrilateral()
        if not ok:
            continue
        p1 = of.extract_pairs(seg_len32(bx[0], bx[1]), 0.0, eps, 0, 1)
        p2 = of.extract_pairs(seg_len32(bx[2], bx[3]), 0.0, eps, 2, 3)
        assert all((r.n_pairs1, r.n_pairs2) == (len(p1), len(p2)) for r in rs)
        if not (len(p1) and len(p2)):
            continue
        quads = of.find_congruent(i1, i2, eps, p1, p2)
        _nb, per, _bc, _bi = of.try_congruent_set(base, quads)
        assert sum(r.n_quads for r in rs) == len(quads)
        assert sum(r.quad_checksum for r in rs) & M64 == H.checksum(quads)
        assert sum(r.n_verified for r in rs) == int((per >= 0).sum())
        assert sum(r.cand_checksum for r in rs) & M64 == H.checksum(quads[per >= 0])
        if len(quads) > 50 * parts:
            assert all(r.n_quads > 0 for r in rs)                     # (the shares are of comparable size)
        have = [r for r in rs if r.has_best]
        if (per >= 0).any():
            k = int(np.flatnonzero(per == per.max())[0])              # first maximum in std::set order
            w = min(have, key=lambda r: (-int(r.best_count), int(r.best_rank)))
            assert w.best_count == per.max() and list(w.best_quad) == quads[k].tolist()
            seen += 1
    assert seen >= 4
    if chunk:
        assert any(g.chunk_stats()["bases"] > 0 for g in ms)


@pytest.mark.parametrize("lanes,group", [(1, 1), (6, 1), (5, 2), (7, 3), (14, 2), (9, 3)])
def test_lanes_and_base_groups_change_no_result(oracle_mod, s4p_lib_built, lanes, group, monkeypatch):
    """Round 5: consecutive bases in flight form GROUPS that go through every kernel in one launch (k_pairs2 / k_quads by
    blockIdx.y, k_verify over all their candidate lists, result records written into pinned host memory).  Whatever the
    packing -- one base per launch, full groups, the partial groups a wait flushes, lane counts that are no multiple of the
    group size -- a whole registration is the oracle's: same totals, same LCP, same 4x4, same transformed cloud; and with
    TryOneBase one base at a time (every group launched with a single base) the per-base records are the oracle's too."""
    from super4pcs_amd import capi
    monkeypatch.setenv("S4P_LANES", str(lanes))
    monkeypatch.setenv("S4P_GROUP", str(group))
    delta, overlap, n_s = 0.01, 0.6, 250
    P, Q, _ = H.small_pair(30000, delta=delta, seed=29)
    om, (o_lcp, o_M, o_Q), gm, (g_lcp, g_M, g_Q) = _run_both(oracle_mod, P, Q, delta, overlap, n_s)
    os_, gi = om.stats(), gm.info()
    assert "%d lanes in groups of %d" % (lanes, group) in gm.verify_kernel_info()
    assert (gi.pairs_total, gi.quads_total, gi.candidates_verified) == (os_.n_pairs, os_.n_quads, os_.n_verified)
    assert g_lcp == o_lcp and np.array_equal(g_M, o_M) and np.max(np.abs(g_Q - o_Q)) <= 1e-4
    # one base at a time through the same context shape
    om2 = oracle_mod.Matcher(oracle_mod.make_options(delta, overlap, n_s), full_counts=False, use_kdtree=True, keep_trace=True)
    om2.init(P, Q)
    g2 = capi.Matcher(capi.make_options(delta, overlap, n_s))
    g2.init_full(P, Q)
    for _ in range(12):
        g_ok, r = g2.try_one_base()
        assert om2.try_one_base() == g_ok
        rec = om2.trace()[0][-1]
        if rec[0]:
            assert (r.n_pairs1, r.n_pairs2) == (rec[5], rec[6])
            if rec[5] and rec[6]:
                assert (r.n_quads, r.n_verified) == (rec[7], rec[8])
    assert g2.info().best_lcp == om2.stats().best_lcp


@pytest.mark.gpu
def test_a_device_pass_that_stalls_is_an_error_not_a_hang(s4p_lib_built, monkeypatch):
    """The host waits for a base by polling the launch number k_verify writes last into the pinned result record, with the
    stream's event as the fallback and a watchdog on top (S4P_WAIT_TIMEOUT_S).  Round 5 found a build whose k_verify never
    finished by perturbation, not by a test: here a launch that stalls on purpose (S4P_ABLATE=3: the last workgroup sleeps ~3 s
    before it writes the records) must come back as an error within the watchdog's second -- and the context must survive to be
    destroyed once the launch has drained."""
    import time
    from super4pcs_amd import capi
    monkeypatch.setenv("S4P_ABLATE", "3")
    monkeypatch.setenv("S4P_WAIT_TIMEOUT_S", "1")
    delta, overlap, n_s = 0.01, 0.6, 200
    P, Q, _ = H.small_pair(20000, delta=delta, seed=3)
    gm = capi.Matcher(capi.make_options(delta, overlap, n_s))
    gm.init_full(P, Q)
    t0 = time.perf_counter()
    with pytest.raises(capi.S4PError) as e:
        for _ in range(4):
            gm.try_one_base()
    dt = time.perf_counter() - t0
    assert "did not finish within" in str(e.value) and "S4P_WAIT_TIMEOUT_S" in str(e.value)
    assert 0.9 <= dt < 2.9                                   # the watchdog, not the end of the stall
    gm.close()                                              # (waits for the stalled launch: the stall ends after ~3 s)


@pytest.mark.gpu
@pytest.mark.parametrize("force,n_s,max_angle,own_bitmap,tiled", [("1", 300, -1.0, True, True), ("1", 250, 30.0, True, True), (None, 2700, -1.0, True, True),
                                                                   (None, 2700, -1.0, False, True), (None, 2700, -1.0, True, False)])
def test_the_sweep_pass_changes_no_result(oracle_mod, s4p_lib_built, monkeypatch, force, n_s, max_angle, own_bitmap, tiled):
    """Round 6: with an early-exit bound in force k_sweep counts, for every gated candidate, the sampled-Q points whose coarse
    cube is marked under its transform -- an upper bound of its inlier count -- and only the candidates whose bound exceeds the
    registration's best go on to k_verify (samples that do not fit LDS by default; S4P_SWEEP_PASS=1 forces the pass for any
    sample).  Forced on a sample that fits LDS, with the Euler-angle gate in force (undecided candidates must reach the host
    whatever their count), and on a sample of several LDS tiles (2700 points: two tiles): the registration is the oracle's --
    LCP, 4x4, transformed cloud, totals -- and the pass really abandons candidates.  The pass indexes a bitmap of its own, as fine
    as a workgroup with a CU to itself can hold (k_sweep_bitmap; the block's four transforms by one MFMA per 16 queries);
    S4P_SWEEP_COARSE=1 makes it use k_verify's coarse bitmap instead: the same registration either way.  The survivors of a sample
    beyond LDS are scored through LDS tiles in rounds (k_verify<., true, true, true>: L0 survivors per tile first, then the lean sweep
    per tile against "confirmed + pending + the tiles to come"); S4P_VERIFY_TILED=0 keeps round 5's form: the same registration."""
    from super4pcs_amd import capi
    if force is not None:
        monkeypatch.setenv("S4P_SWEEP_PASS", force)
    if not own_bitmap:
        monkeypatch.setenv("S4P_SWEEP_COARSE", "1")
    if not tiled:
        monkeypatch.setenv("S4P_VERIFY_TILED", "0")             # the survivors scored with the queries from global memory (round 5's form)
    if max_angle >= 0:
        monkeypatch.setenv("S4P_ANGLE_TOL", "0.02")             # many candidates with an undecided gate
    delta, overlap = 0.01, 0.6
    if n_s > 2000:
        delta, overlap = 0.004, 0.8                           # (fewer trials: the oracle replays the whole registration on the host)
    P, Q, _ = H.small_pair(60000 if n_s > 2000 else 30000, delta=delta, seed=41, overlap=overlap)
    kw = {"max_angle": max_angle} if max_angle >= 0 else {}
    O = oracle_mod
    om = O.Matcher(O.make_options(delta, overlap, n_s, **kw), full_counts=False, use_kdtree=True)
    if n_s > 2000:
        om.set_threads(os.cpu_count() or 1)
    o_lcp, o_M, o_Q = om.compute_transformation(P, Q)
    gm = capi.Matcher(capi.make_options(delta, overlap, n_s, **kw))
    gm.profile_enable(True, False)
    g_lcp, g_M, g_Q = gm.compute_transformation(P, Q)
    os_, gi = om.stats(), gm.info()
    if n_s > 2000:
        assert gi.n_sampled_q > 2560 and ("query tiles through LDS" if tiled else "queries from global memory") in gm.verify_kernel_info()
    assert (gi.pairs_total, gi.quads_total, gi.candidates_verified) == (os_.n_pairs, os_.n_quads, os_.n_verified)
    assert g_lcp == o_lcp and np.array_equal(g_M, o_M) and np.max(np.abs(g_Q - o_Q)) <= 1e-4
    assert gm.profile_get().verify_pruned > 0.3 * gi.candidates_verified
