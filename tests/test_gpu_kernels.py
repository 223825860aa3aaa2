"""-m gpu: kernel-level parity of the HIP path (through the C ABI) against the CPU oracle.

Bar: bit-exact for every integer / index result (pairs, quads, inlier counts, winner),
exact float equality for the winning transform (same IEEE operations on both sides).
"""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(oracle_mod, s4p_lib_built):
    from super4pcs_amd import capi
    O = oracle_mod
    delta, overlap, n_s = 0.01, 0.6, 400
    P, Q, T = H.small_pair(30000, delta=delta, seed=11)
    m = H.init_oracle(O, P, Q, delta, overlap, n_s)
    ctx = capi.Context(capi.make_options(delta, overlap, n_s))
    Ps, Qs = m.cloud(0), m.cloud(1)
    ctx.set_clouds(Ps, Qs)
    return dict(O=O, m=m, ctx=ctx, Ps=Ps, Qs=Qs, delta=delta, T_gt=T)


def test_device_is_gfx950(setup):
    assert "gfx950" in setup["ctx"].device_name()


def test_ieee_sqrt_div_muladd(setup):
    rng = np.random.default_rng(3)
    a = (rng.normal(size=1 << 16) * 10 ** rng.uniform(-6, 6, 1 << 16)).astype(np.float32)
    b = (rng.normal(size=1 << 16) * 10 ** rng.uniform(-6, 6, 1 << 16)).astype(np.float32)
    b[b == 0] = 1
    s, d, ma = setup["ctx"].selftest_ieee(a, b)
    assert np.array_equal(s, np.sqrt(np.abs(a)))
    assert np.array_equal(d, a / b)
    with np.errstate(over="ignore"):
        ref = a * b + (a * a + b * b)
    assert np.array_equal(ma, ref, equal_nan=True)


def test_verify_transforms_counts_bit_exact(setup):
    m, ctx = setup["m"], setup["ctx"]
    rng = np.random.default_rng(5)
    Ts = [np.eye(4, dtype=np.float32)]
    # the ground-truth motion expressed in the centred frames: T_c = [R | R*cQ + t - cP]
    cP, cQ, _, _ = m.frame()
    Tg = setup["T_gt"].astype(np.float64)
    Tc = np.eye(4)
    Tc[:3, :3] = Tg[:3, :3]
    Tc[:3, 3] = Tg[:3, :3] @ cQ + Tg[:3, 3] - cP
    Ts.append(Tc.astype(np.float32))
    for _ in range(60):
        Ts.append(H.random_rigid(rng, 0.05))
    for _ in range(20):   # small perturbations of the good transform: counts in the interesting range
        Tp = H.random_rigid(rng, 0.0)
        Tp[:3, :3] = np.eye(3) + 0.02 * rng.normal(size=(3, 3))
        Ts.append((Tp.astype(np.float64) @ Tc).astype(np.float32))
    Ts.append(np.full((4, 4), 1e30, np.float32))          # far away / overflow
    Ts = np.stack(Ts)
    got = ctx.verify_transforms(Ts)
    want = m.verify_batch(Ts)                              # kd-tree (reference-faithful)
    m.set_mode(True, use_kdtree=False, keep_trace=True)
    want_bf = m.verify_batch(Ts)                           # brute force predicate
    m.set_mode(True, use_kdtree=True, keep_trace=True)
    assert np.array_equal(want, want_bf)
    assert np.array_equal(got, want)
    assert got[1] > 0.3 * setup["Qs"].shape[0]            # the ground-truth pose is a real match


def _bases(m, n):
    out = []
    while len(out) < n:
        ok, i1, i2, base, bx = m.select_quadrilateral()
        if ok:
            out.append((i1, i2, base.copy(), bx.copy()))
    return out


def test_extract_pairs_order_exact_and_ids_persist(setup):
    m, ctx = setup["m"], setup["ctx"]
    eps = 2.0 * setup["delta"]
    for (i1, i2, base, bx) in _bases(m, 3):
        ctx.set_base(bx)
        m.set_base(base)
        d1 = float(np.float32(np.linalg.norm(bx[0] - bx[1])))
        d2 = float(np.float32(np.linalg.norm(bx[2] - bx[3])))
        for d, a, b in ((d1, 0, 1), (d2, 2, 3)):
            want = m.extract_pairs(d, 0.0, eps, a, b)
            got = ctx.extract_pairs(d, 0.0, eps, a, b)
            assert got.shape == want.shape
            assert np.array_equal(got, want)      # same pairs, same emission order
            assert want.shape[0] > 0


def test_pairs_match_reference_test_predicate(setup):
    """tests/pair_extraction.cc:239-314 + tests/testing.h:172-194: sorted ExtractPairs output ==
    sorted brute force { (j,i),(i,j) : |‖q_i-q_j‖ - d| <= eps }."""
    ctx, Qs = setup["ctx"], setup["Qs"]
    eps = np.float32(2.0 * setup["delta"])
    ctx.set_base(np.zeros((4, 3), np.float32))
    for d in (np.float32(0.3), np.float32(0.12)):
        got = ctx.extract_pairs(float(d), 0.0, float(eps), 0, 1)
        diff = Qs[:, None, :] - Qs[None, :, :]
        dist = np.sqrt((diff[..., 0] * diff[..., 0] + (diff[..., 1] * diff[..., 1] + diff[..., 2] * diff[..., 2])).astype(np.float32))
        ok = np.abs(dist.astype(np.float64) - np.float64(d)) <= np.float64(eps)
        ii, jj = np.nonzero(ok)
        want = sorted((int(a), int(b)) for a, b in zip(ii, jj) if a != b)
        assert sorted(map(tuple, got.tolist())) == want


def test_find_congruent_quads_exact(setup):
    m, ctx = setup["m"], setup["ctx"]
    eps = 2.0 * setup["delta"]
    nonempty = 0
    for (i1, i2, base, bx) in _bases(m, 6):
        ctx.set_base(bx)
        m.set_base(base)      # _bases() ran ahead: put the oracle's base_3D_ back on this base
        d1 = float(np.float32(np.linalg.norm(bx[0] - bx[1])))
        d2 = float(np.float32(np.linalg.norm(bx[2] - bx[3])))
        p1 = m.extract_pairs(d1, 0.0, eps, 0, 1)
        p2 = m.extract_pairs(d2, 0.0, eps, 2, 3)
        if len(p1) == 0 or len(p2) == 0:
            continue
        want = m.find_congruent(i1, i2, eps, p1, p2)
        got = ctx.find_congruent(i1, i2, eps, p1, p2)
        assert np.array_equal(got, want)
        nonempty += len(want) > 0
    assert nonempty > 0


def test_try_congruent_set_counts_and_winner(setup):
    m, ctx = setup["m"], setup["ctx"]
    eps = 2.0 * setup["delta"]
    tested = 0
    for (i1, i2, base, bx) in _bases(m, 8):
        m.set_base(base)
        d1 = float(np.float32(np.linalg.norm(bx[0] - bx[1])))
        d2 = float(np.float32(np.linalg.norm(bx[2] - bx[3])))
        p1 = m.extract_pairs(d1, 0.0, eps, 0, 1)
        p2 = m.extract_pairs(d2, 0.0, eps, 2, 3)
        if len(p1) == 0 or len(p2) == 0:
            continue
        quads = m.find_congruent(i1, i2, eps, p1, p2)
        if len(quads) == 0:
            continue
        nb, per, bc, bi = m.try_congruent_set(base, quads)
        r, got_per = ctx.try_congruent_set(base, quads)
        assert np.array_equal(got_per, per)
        assert r.n_verified == nb
        if nb:
            v = per[per >= 0]
            assert r.best_count == v.max()
            first = int(np.nonzero(per == v.max())[0][0])
            assert r.best_rank == first
            ok, rms, T = m.compute_rigid(base, quads[first])
            assert ok
            assert np.array_equal(np.array(r.best_transform, np.float32).reshape(4, 4), T)
        tested += 1
    assert tested > 0


def test_fused_try_base_matches_oracle_trace(setup):
    O = setup["O"]
    delta, overlap, n_s = 0.01, 0.6, 400
    from super4pcs_amd import capi
    P, Q, _ = H.small_pair(30000, delta=delta, seed=11)
    m = H.init_oracle(O, P, Q, delta, overlap, n_s)
    ctx = capi.Context(capi.make_options(delta, overlap, n_s))
    ctx.set_clouds(m.cloud(0), m.cloud(1))
    total_c = 0
    for t in range(12):
        ok, i1, i2, base, bx = m.select_quadrilateral()
        if not ok:
            continue
        ctx.set_base(bx)
        d1 = float(np.float32(np.linalg.norm(bx[0] - bx[1])))
        d2 = float(np.float32(np.linalg.norm(bx[2] - bx[3])))
        eps = 2.0 * delta
        p1 = m.extract_pairs(d1, 0.0, eps, 0, 1)
        p2 = m.extract_pairs(d2, 0.0, eps, 2, 3)
        quads = m.find_congruent(i1, i2, eps, p1, p2) if len(p1) and len(p2) else np.zeros((0, 4), np.int32)
        nb, per, bc, bi = (m.try_congruent_set(base, quads) if len(quads) else (0, np.zeros(0, np.int32), 0, -1))
        r = ctx.try_base(base, i1, i2)
        assert (r.n_pairs1, r.n_pairs2) == (len(p1), len(p2))
        if len(p1) and len(p2):
            assert r.n_quads == len(quads)
            assert r.n_verified == nb
            gq, gc = ctx.last_candidates(max(len(quads), 1))
            assert np.array_equal(gq, quads)
            assert np.array_equal(gc, per)
            if nb:
                v = per[per >= 0]
                first = int(np.nonzero(per == v.max())[0][0])
                assert r.best_count == v.max()
                assert list(r.best_quad) == quads[first].tolist()
        total_c += nb
    assert total_c > 0


def test_gpu_sampler_equals_host_and_oracle(oracle_mod, s4p_lib_built, monkeypatch):
    """UniformDistSampler on the device (s4p_sampler.hip) vs the host hash vs the oracle: identical kept indices,
    in input order, including duplicates, negative coordinates and a ragged size."""
    from super4pcs_amd import capi, datasets
    P, Q, _ = datasets.bumpy_pair(300007, 0.5, 0.004, seed=9)
    X = np.concatenate([P, P[:1000], Q - 0.7]).astype(np.float32)          # exact duplicates + negative octants
    for delta in (0.004, 0.02):
        dev = capi.uniform_dist_sample(X, delta)
        monkeypatch.setenv("S4P_SAMPLER", "host")
        host = capi.uniform_dist_sample(X, delta)
        monkeypatch.delenv("S4P_SAMPLER")
        assert np.array_equal(dev, host) and np.all(np.diff(dev) > 0)
        assert np.array_equal(X[dev], oracle_mod.sample(X, delta))
    # coordinates whose voxel index does not fit the 21-bit device key fall back to the host path, same answer
    Y = (X[:40000] * np.float32(1e7)).astype(np.float32)
    assert np.array_equal(Y[capi.uniform_dist_sample(Y, 0.004)], oracle_mod.sample(Y, 0.004))


@pytest.mark.parametrize("cell_factor", [None, 1.6, 2.5])
def test_quantised_locate_cannot_lose_an_inlier_at_distance_delta(s4p_lib_built, cell_factor, monkeypatch):
    """Adversarial case for the LCP structure's locating slack (LcpGridHost::plan): queries quantised to 16 bit over a long
    bounding box (half a quantisation step = 0.0039 cell, the largest the LDS path accepts), every query exactly one P point
    at distance delta (1 - 1e-6) straight along -x, 64 transforms that move the pairs across the cell faces.  A query that
    the quantised locate puts into the cell next to its true one must still find its point: counts equal a brute-force
    float32 count.  (With cells of 1.002 delta this lost 8 inliers of 127 902.)  cell_factor 1.6 / 2.5: the same with the
    enlarged cells LcpGridHost::plan falls back to for huge extents -- the locate error is a fraction of the CELL, so the
    reach of the lists must be too (delta + 0.01 h; a fixed 1.01 delta stops covering it from h ~ 1.4 delta on)."""
    from super4pcs_amd import capi
    F = np.float32
    delta, n = 1.0, 2000
    rng = np.random.default_rng(5)
    h = (cell_factor or 1.002) * delta
    if cell_factor:
        monkeypatch.setenv("S4P_CELL_FACTOR", str(cell_factor))
    ext = 0.0078 * h * 65535.0
    Q = np.stack([rng.uniform(0, ext, n), rng.uniform(0, 3, n), rng.uniform(0, 3, n)], axis=1).astype(F)
    Q[0, 0], Q[1, 0] = 0.0, ext
    Ts, Ps = [], []
    for k in range(64):
        T = np.eye(4, dtype=F)
        T[0, 3] = F(0.0137 * k); T[1, 3] = F(5.0 * k)
        Ts.append(T)
        Pk = Q.astype(np.float64).copy()
        Pk[:, 0] += float(T[0, 3]) - delta * (1 - 1e-6)
        Pk[:, 1] += float(T[1, 3])
        Ps.append(Pk.astype(F))
    Ts = np.stack(Ts)
    P = np.concatenate(Ps)
    want = []
    for T in Ts:
        tx = ((T[0, 0] * Q[:, 0] + T[0, 1] * Q[:, 1]) + T[0, 2] * Q[:, 2]) + T[0, 3]
        ty = ((T[1, 0] * Q[:, 0] + T[1, 1] * Q[:, 1]) + T[1, 2] * Q[:, 2]) + T[1, 3]
        tz = ((T[2, 0] * Q[:, 0] + T[2, 1] * Q[:, 1]) + T[2, 2] * Q[:, 2]) + T[2, 3]
        near = P[np.abs(P[:, 1] - T[1, 3] - 1.5) < 4.0]
        cnt = 0
        for i in range(n):
            dx, dy, dz = tx[i] - near[:, 0], ty[i] - near[:, 1], tz[i] - near[:, 2]
            cnt += bool(((dx * dx + (dy * dy + dz * dz)) <= F(delta) * F(delta)).any())
        want.append(cnt)
    ctx = capi.Context(capi.make_options(delta, 0.5, n), max_pairs=1 << 16, max_quads=1 << 16)
    ctx.set_clouds(P, Q)
    got = ctx.verify_transforms(Ts)
    assert sum(want) > 120000
    assert np.array_equal(got.astype(np.int64), np.array(want, np.int64))
