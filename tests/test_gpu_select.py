"""SelectRandomTriangle + the 4th-point scan of SelectQuadrilateral as device reductions
(k_select_triangle / k_select_fourth / k_select_finish behind s4p_select_base_points), against
  * the literal loops of match4pcsBase.cc:185-218 and :303-338 restated in numpy float32 (one attempt, given draws),
    including exact ties (lattice clouds), a degenerate plane, no wide triangle and no admissible 4th point;
  * the host search structures of the engine (mode 0) and the oracle's SelectQuadrilateral, over consecutive bases of the
    seeded stream (same draws => same bases => same RNG position afterwards)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F = np.float32


def _literal_attempt(P, draws, limit_sq, too_small):
    """One pass of SelectQuadrilateral's loop body (match4pcsBase.cc:285-338) for the given draws, float32 like the reference's
    Eigen expressions (x + (y + z) reductions), plane coefficients in double."""
    P = P.astype(F)
    o = P[draws[0]]
    sec, thr = draws[1::2], draws[2::2]
    u, w = P[sec] - o, P[thr] - o

    def sqn(v):
        return v[:, 0] * v[:, 0] + (v[:, 1] * v[:, 1] + v[:, 2] * v[:, 2])
    c = np.stack([u[:, 1] * w[:, 2] - u[:, 2] * w[:, 1], u[:, 2] * w[:, 0] - u[:, 0] * w[:, 2], u[:, 0] * w[:, 1] - u[:, 1] * w[:, 0]], axis=1)
    wide = np.sqrt(sqn(c))
    ok = (sqn(u) < F(limit_sq)) & (sqn(w) < F(limit_sq)) & (wide > 0)
    if not ok.any():
        return 1, [-1, -1, -1, -1]
    wide = np.where(ok, wide, F(-1))
    t = int(np.argmax(wide))                                     # first of the widest: "wide > widest" keeps the earliest
    b1, b2, b3 = int(draws[0]), int(sec[t]), int(thr[t])
    (x1, y1, z1), (x2, y2, z2), (x3, y3, z3) = [tuple(float(v) for v in P[b]) for b in (b1, b2, b3)]
    denom = F(-x3 * y2 * z1 + x2 * y3 * z1 + x3 * y1 * z2 - x1 * y3 * z2 - x2 * y1 * z3 + x1 * y2 * z3)
    if not denom != 0:
        return 2, [b1, b2, b3, -1]
    d = float(denom)
    pa = F((-y2 * z1 + y3 * z1 + y1 * z2 - y3 * z2 - y1 * z3 + y2 * z3) / d)
    pb = F((x2 * z1 - x3 * z1 - x1 * z2 + x3 * z2 + x1 * z3 - x2 * z3) / d)
    pc = F((-x2 * y1 + x3 * y1 + x1 * y2 - x3 * y2 - x1 * y3 + x2 * y3) / d)
    with np.errstate(over="ignore", invalid="ignore"):
        dist = np.abs(((pa * P[:, 0] + pb * P[:, 1]) + pc * P[:, 2]) - F(1))
        far = np.ones(len(P), bool)
        for b in (b1, b2, b3):
            far &= sqn(P - P[b]) >= F(too_small)
    dist = np.where(far & (dist < np.finfo(F).max), dist, np.inf)
    if not np.isfinite(dist).any():
        return 3, [b1, b2, b3, -1]
    return 0, [b1, b2, b3, int(np.argmin(dist))]                 # first of the closest


def _ctx_for(capi, P, delta):
    Q = P[:: max(1, len(P) // 300)][:300].copy()
    ctx = capi.Context(capi.make_options(delta, 0.5, max(len(Q), 10)), max_pairs=1 << 16, max_quads=1 << 16)
    ctx.set_clouds(P, Q)
    return ctx


def _check_attempts(capi, P, delta, diameter, attempts, seed, expect=None):
    ctx = _ctx_for(capi, P, delta)
    rng = np.random.default_rng(seed)
    limit_sq = float(F(diameter) * F(diameter))
    too_small = float(F(float(F(diameter) * F(0.2)) ** 2))
    seen = set()
    for _ in range(attempts):
        draws = rng.integers(0, len(P), 2001).astype(np.uint32)
        st, ids, xyz = ctx.select_base_points(draws, limit_sq, too_small)
        w_st, w_ids = _literal_attempt(P, draws, limit_sq, too_small)
        assert st == w_st and list(ids) == w_ids, (st, list(ids), w_st, w_ids)
        for k in range(4):
            if ids[k] >= 0:
                assert np.array_equal(xyz[k], P[ids[k]].astype(F))
        seen.add(st)
    if expect is not None:
        assert seen == expect, seen
    return ctx


def test_attempt_matches_literal_loops_on_a_surface(s4p_lib_built):
    from super4pcs_amd import capi, datasets
    P, _Q, _ = datasets.bumpy_pair(60000, overlap=0.6, delta=0.01, noise_sigma=0.003, seed=11)
    P = P.astype(F) - P.astype(F).mean(axis=0)
    _check_attempts(capi, P, 0.01, 0.7, 25, 5, expect={0})


def test_attempt_ties_resolve_to_the_first_index(s4p_lib_built):
    """Integer lattice offset from the origin: many triangles share an area and many points share a plane distance
    exactly; the reference keeps the first."""
    from super4pcs_amd import capi
    g = np.arange(12, dtype=F)
    P = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3) * F(0.25) + F(0.5)
    P = np.concatenate([P, P[::-1]])                           # every point twice: guaranteed ties, far apart in index
    _check_attempts(capi, P, 0.25, 2.0, 40, 9, expect={0})


def test_attempt_status_codes(s4p_lib_built):
    from super4pcs_amd import capi
    rng = np.random.default_rng(3)
    # 1: collinear cloud -> every cross product is exactly zero -> SelectRandomTriangle fails
    line = np.zeros((4096, 3), F); line[:, 0] = np.arange(4096, dtype=F) * F(0.125)
    _check_attempts(capi, line, 0.125, 1e6, 3, 1, expect={1})
    # 2: all points in the plane z = 0 -> the plane's denominator (a 3x3 determinant with a zero column) is 0
    flat = np.zeros((4096, 3), F); flat[:, :2] = rng.integers(-64, 64, (4096, 2)).astype(F) * F(0.5)
    _check_attempts(capi, flat, 0.5, 1e6, 3, 2, expect={2})
    # 3: every point closer than too_small to the triangle -> no admissible 4th point
    ball = rng.normal(size=(4096, 3)).astype(F)
    ball /= np.linalg.norm(ball, axis=1, keepdims=True).astype(F)
    ball = (ball + F(3.0)).astype(F)
    _check_attempts(capi, ball, 0.05, 1e3, 3, 4, expect={3})


def test_a_batch_of_attempts_equals_the_attempts_one_by_one(s4p_lib_built):
    """k_select_fourth scans P once for a whole batch: every batch size up to the maximum, attempts that have no triangle
    (all draws equal: every cross product is zero) mixed with attempts that find a base, against the literal loops and against
    the same attempts evaluated singly."""
    from super4pcs_amd import capi, datasets
    P, _Q, _ = datasets.bumpy_pair(60000, overlap=0.6, delta=0.01, noise_sigma=0.003, seed=13)
    P = P.astype(F) - P.astype(F).mean(axis=0)
    ctx = _ctx_for(capi, P, 0.01)
    rng = np.random.default_rng(17)
    limit_sq = float(F(0.7) * F(0.7))
    too_small = float(F(float(F(0.7) * F(0.2)) ** 2))
    bmax = int(ctx.L.s4p_select_batch_max())
    assert bmax >= 16
    seen = set()
    for n in (1, 2, 3, 5, 8, 13, bmax):
        draws = rng.integers(0, len(P), (n, 2001)).astype(np.uint32)
        for a in range(1, n, 3):
            draws[a, :] = draws[a, 0]                           # SelectRandomTriangle fails for this attempt only
        st, ids, xyz = ctx.select_base_points_batch(draws, limit_sq, too_small)
        for a in range(n):
            w_st, w_ids = _literal_attempt(P, draws[a], limit_sq, too_small)
            assert int(st[a]) == w_st and list(ids[a]) == w_ids, (n, a, int(st[a]), list(ids[a]), w_st, w_ids)
            s1, i1, x1 = ctx.select_base_points(draws[a], limit_sq, too_small)
            assert s1 == int(st[a]) and np.array_equal(i1, ids[a]) and np.array_equal(x1, xyz[a])
            seen.add(w_st)
    assert seen == {0, 1}, seen
    # ties and the "no admissible fourth point" answer inside a batch
    g = np.arange(12, dtype=F)
    L = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3) * F(0.25) + F(0.5)
    L = np.concatenate([L, L[::-1]])
    ctx = _ctx_for(capi, L, 0.25)
    draws = rng.integers(0, len(L), (bmax, 2001)).astype(np.uint32)
    st, ids, _ = ctx.select_base_points_batch(draws, float(F(2.0) * F(2.0)), float(F(float(F(2.0) * F(0.2)) ** 2)))
    for a in range(bmax):
        w_st, w_ids = _literal_attempt(L, draws[a], float(F(2.0) * F(2.0)), float(F(float(F(2.0) * F(0.2)) ** 2)))
        assert int(st[a]) == w_st and list(ids[a]) == w_ids, (a, int(st[a]), list(ids[a]), w_st, w_ids)
    ball = rng.normal(size=(4096, 3)).astype(F)
    ball /= np.linalg.norm(ball, axis=1, keepdims=True).astype(F)
    ball = (ball + F(3.0)).astype(F)
    ctx = _ctx_for(capi, ball, 0.05)
    draws = rng.integers(0, len(ball), (7, 2001)).astype(np.uint32)
    st, ids, _ = ctx.select_base_points_batch(draws, 1e3 * 1e3, float(F(float(F(1e3) * F(0.2)) ** 2)))
    assert list(st) == [3] * 7 and all(int(i[3]) == -1 for i in ids)


def test_out_of_range_draw_is_refused(s4p_lib_built):
    from super4pcs_amd import capi
    P = np.random.default_rng(0).random((1000, 3)).astype(F)
    ctx = _ctx_for(capi, P, 0.05)
    draws = np.zeros(2001, np.uint32); draws[7] = 1000
    with pytest.raises(capi.S4PError):
        ctx.select_base_points(draws, 1.0, 0.01)


def _select_sequence(m, n):
    out = []
    for _ in range(n):
        ok, i1, i2, base, bx = m.select_quadrilateral()
        out.append((ok, F(i1), F(i2), tuple(int(b) for b in base), bx.copy()))
    return out


def _same(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert x[0] == y[0] and x[3] == y[3], (x[:4], y[:4])
        assert x[1] == y[1] and x[2] == y[2], (x[:4], y[:4])
        assert np.array_equal(x[4], y[4])


def test_matcher_device_selection_equals_host_and_oracle(oracle_mod, s4p_lib_built):
    """The whole SelectQuadrilateral (retries, the 12-pairing order, invariants) with the searches on the device: 60
    consecutive bases of the seeded stream against the host structures and against the oracle."""
    from super4pcs_amd import capi, datasets
    O = oracle_mod
    delta, overlap, n_s = 0.01, 0.6, 300
    P, Q, _ = datasets.bumpy_pair(40000, overlap=overlap, delta=delta, noise_sigma=0.3 * delta, seed=21)
    om = O.Matcher(O.make_options(delta, overlap, n_s), full_counts=False, use_kdtree=True)
    om.init(P, Q)
    want = []
    for _ in range(60):
        ok, i1, i2, base, bx = om.select_quadrilateral()
        want.append((ok, F(i1), F(i2), tuple(int(b) for b in base), np.asarray(bx, F).reshape(4, 3).copy()))
    got = {}
    for mode in (0, 1):
        gm = capi.Matcher(capi.make_options(delta, overlap, n_s))
        gm.set_device_selection(mode)
        gm.init_full(P, Q)
        assert gm.device_selection() == bool(mode)
        got[mode] = _select_sequence(gm, 60)
    _same(got[0], got[1])
    _same(got[1], want)


def test_registration_with_device_selection_is_the_same_registration(oracle_mod, s4p_lib_built):
    """ComputeTransformation with the selector thread calling the device searches while bases are in flight."""
    from super4pcs_amd import capi, datasets
    O = oracle_mod
    delta, overlap, n_s = 0.01, 0.6, 200
    P, Q, _ = datasets.bumpy_pair(20000, overlap=overlap, delta=delta, noise_sigma=0.3 * delta, seed=3)
    om = O.Matcher(O.make_options(delta, overlap, n_s), full_counts=False, use_kdtree=True)
    o_lcp, o_M, _ = om.compute_transformation(P, Q)
    gm = capi.Matcher(capi.make_options(delta, overlap, n_s))
    gm.set_device_selection(1)
    g_lcp, g_M, _ = gm.compute_transformation(P, Q)
    assert gm.device_selection()
    assert g_lcp == o_lcp and np.max(np.abs(g_M - o_M)) <= 1e-4
    assert gm.info().candidates_verified == om.stats().n_verified


def test_large_sampled_p_selects_on_the_device_by_default(oracle_mod, s4p_lib_built):
    """From 2^20 sampled P points the matcher picks the device searches by itself; bases equal the host structures'."""
    from super4pcs_amd import capi
    rng = np.random.default_rng(17)
    n = (1 << 20) + 4096
    # a wavy sheet sampled on a jittered lattice: points at least ~delta apart, extent 1024 x 1024 cells
    gx, gy = np.meshgrid(np.arange(1026, dtype=F), np.arange(1026, dtype=F), indexing="ij")
    xy = np.stack([gx.ravel(), gy.ravel()], axis=1)[:n] * F(0.01)
    xy += rng.uniform(-0.002, 0.002, xy.shape).astype(F)
    z = (0.3 * np.sin(xy[:, 0] * 1.7) * np.cos(xy[:, 1] * 2.3)).astype(F)
    Ps = np.concatenate([xy, z[:, None]], axis=1).astype(F)
    Qs = Ps[rng.choice(n, 400, replace=False)].copy()
    seqs = {}
    for mode in (-1, 0):
        gm = capi.Matcher(capi.make_options(0.01, 0.5, 400), max_pairs=1 << 18, max_quads=1 << 18)
        gm.set_device_selection(mode)
        gm.init_sampled(Ps, Qs, False)
        assert gm.device_selection() == (mode == -1)
        seqs[mode] = _select_sequence(gm, 12)
    _same(seqs[-1], seqs[0])
