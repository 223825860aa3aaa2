"""-m gpu: end-to-end parity of the engine (host RANSAC driver + HIP hot path) with the CPU oracle.

Same inputs, same seed -> same sampled clouds, same bases, same per-trial counts, same winning
candidate; final R,t within 1e-4 (BASELINE.json north_star), LCP (integer inliers) exact.
"""
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
    X = rng.normal(size=(100003, 3)).astype(np.float32)
    M = H.random_rigid(rng, 0.3)
    got = ctx.transform_points(M, X)
    x, y, z = X[:, 0], X[:, 1], X[:, 2]
    want = np.stack([((M[r, 0] * x + M[r, 1] * y) + M[r, 2] * z) + M[r, 3] for r in range(3)], axis=1)
    assert np.array_equal(got, want)


def test_options_rejected_loudly(s4p_lib_built):
    from super4pcs_amd import capi
    with pytest.raises(capi.S4PError) as e:
        capi.Context(capi.make_options(0.01, 0.5, 200, max_angle=30.0))
    assert e.value.code == -6
    with pytest.raises(capi.S4PError):
        capi.Context(capi.make_options(-1.0, 0.5, 200))


def test_empty_inputs_return_large_number(s4p_lib_built):
    """match4pcsBase.hpp:69-70 / tests/externalAppTest/main.cpp: empty sets -> kLargeNumber."""
    from super4pcs_amd import capi
    m = capi.Matcher(capi.make_options(0.01, 0.5, 200))
    lcp, M, Q = m.compute_transformation(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))
    assert lcp == np.float32(1e9)
