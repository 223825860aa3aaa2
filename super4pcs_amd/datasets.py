"""Synthetic point-cloud pairs for the BASELINE.json configs.

All generators are seeded numpy, float32 output, and jitter coordinates so that no two
points are lattice-aligned (exact delta^2 distances must be measure-zero).
"""
import numpy as np


def _bumpy_radius(dirs, rng_params):
    a, f, g, al, be = rng_params
    theta = np.arccos(np.clip(dirs[:, 2], -1.0, 1.0))
    phi = np.arctan2(dirs[:, 1], dirs[:, 0])
    r = np.ones(dirs.shape[0])
    for k in range(len(a)):
        r += a[k] * np.sin(f[k] * theta + al[k]) * np.sin(g[k] * phi + be[k])
    return r


def _random_rotation(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def bumpy_pair(n_points, overlap=0.5, delta=0.004, noise_sigma=None, seed=20140814):
    """Config-3 style pair: a closed bumpy surface r(theta,phi) = 1 + sum a_k sin(f_k theta + al_k) sin(g_k phi + be_k),
    scaled to unit bounding-box diagonal.  P and Q are independent draws; P keeps directions with
    z >= -c, Q keeps z <= c, with c chosen so the shared band is `overlap` of each cloud.
    Q is then moved by a random rigid motion and perturbed by Gaussian noise (sigma = delta by default).

    Returns (P, Q, T_gt) with T_gt the 4x4 that maps the *moved* Q back onto P.
    """
    rng = np.random.default_rng(seed)
    K = 8
    params = (rng.uniform(0.02, 0.08, K), rng.integers(1, 7, K), rng.integers(1, 7, K),
              rng.uniform(0, 2 * np.pi, K), rng.uniform(0, 2 * np.pi, K))
    # each cloud covers a fraction (1+c)/2 of the sphere; the shared band covers c: c/((1+c)/2) = overlap
    c = overlap / (2.0 - overlap)

    def draw(n, keep):
        out = []
        got = 0
        while got < n:
            d = rng.normal(size=(int((n - got) * 2.2) + 16, 3))
            d /= np.linalg.norm(d, axis=1, keepdims=True)
            d = d[keep(d[:, 2])]
            out.append(d)
            got += d.shape[0]
        d = np.concatenate(out)[:n]
        return d * _bumpy_radius(d, params)[:, None]

    P = draw(n_points, lambda z: z >= -c)
    Q = draw(n_points, lambda z: z <= c)
    both = np.concatenate([P, Q])
    diag = np.linalg.norm(both.max(0) - both.min(0))
    P /= diag
    Q /= diag
    R = _random_rotation(rng)
    t = rng.uniform(-0.5, 0.5, 3)
    sigma = delta if noise_sigma is None else noise_sigma
    Qm = Q @ R.T + t + rng.normal(scale=sigma, size=Q.shape)
    Qm = Qm[rng.permutation(Qm.shape[0])]
    T = np.eye(4)
    T[:3, :3] = R.T
    T[:3, 3] = -R.T @ t
    return P.astype(np.float32), Qm.astype(np.float32), T


def sphere_cloud(n, seed):
    """tests/testing.h:157-168 generateSphereCloud analogue (random points on the unit sphere)."""
    rng = np.random.default_rng(seed)
    d = rng.uniform(-1, 1, size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return d.astype(np.float32)


def lidar_pair(n_points, delta=0.05, seed=5, extent=30.0, n_boxes=200, n_cyl=50, yaw_deg=30.0, shift=10.0,
               first_scan_points=None, second_scan_points=None, ground_fraction=0.3):
    """Config-4 style pair: two simulated terrestrial scans of one scene (ground plane, 200 axis-aligned boxes
    with roofs, 50 vertical cylinders) from poses `shift` metres apart and `yaw_deg` apart, ~1/r^2 density, range noise
    sigma = delta.  Returns (P, Q, T_gt) with Q expressed in the second scanner's frame.

    Round 4: the scene is 30 m across instead of 60 m and the ground carries 30 % of the returns instead of 50 %.  With
    delta = 0.05 m the sampled Q (sample_size 20 000) has to be about as dense as delta for a congruent base to exist at all
    (4PCS assumes sample spacing ~ delta): on the 60 m scene a sampled point had a partner within 2 delta with probability
    ~0.01 per base point, 1466 trials verified 403 candidates and the registration ended on a pose that slid along the ground
    plane.  On the denser scene the reference's estimator recovers the generator's pose (tests/test_gpu_configs.py).
    Feature sizes scale with extent / 30, so a reduced-scale case (extent ~ sqrt(scale), sample ~ scale) keeps the density."""
    rng = np.random.default_rng(seed)
    k = extent / 30.0
    boxes = np.concatenate([rng.uniform(-extent / 2, extent / 2, (n_boxes, 2)), rng.uniform(0.8 * k, 3.5 * k, (n_boxes, 2)),
                            rng.uniform(1.5 * k, 8.0 * k, (n_boxes, 1))], axis=1)
    cyls = np.concatenate([rng.uniform(-extent / 2, extent / 2, (n_cyl, 2)), rng.uniform(0.15 * k, 0.6 * k, (n_cyl, 1)),
                           rng.uniform(3 * k, 10 * k, (n_cyl, 1))], axis=1)
    eye = 1.8 * k

    def surface_samples(n):
        pts = []
        ng = int(n * ground_fraction)
        r = extent / 2 * np.sqrt(rng.uniform(0, 1, ng))
        a = rng.uniform(0, 2 * np.pi, ng)
        pts.append(np.stack([r * np.cos(a), r * np.sin(a), np.zeros(ng)], 1))
        nb = n - ng
        which = rng.integers(0, n_boxes + n_cyl, nb)
        u, v, w = rng.uniform(-0.5, 0.5, nb), rng.uniform(-0.5, 0.5, nb), rng.uniform(0, 1, nb)
        out = np.zeros((nb, 3))
        isb = which < n_boxes
        b = boxes[np.minimum(which, n_boxes - 1)]
        face = rng.integers(0, 5, nb)                                   # four walls and the roof
        bx = np.where(face < 2, (face * 2 - 1) * 0.5 * b[:, 2], u * b[:, 2])
        by = np.where(face < 2, v * b[:, 3], np.where(face < 4, ((face - 2) * 2 - 1) * 0.5 * b[:, 3], v * b[:, 3]))
        bz = np.where(face == 4, b[:, 4], w * b[:, 4])
        out[isb] = np.stack([b[:, 0] + bx, b[:, 1] + by, bz], 1)[isb]
        c = cyls[np.clip(which - n_boxes, 0, n_cyl - 1)]
        th = 2 * np.pi * (u + 0.5)
        out[~isb] = np.stack([c[:, 0] + c[:, 2] * np.cos(th), c[:, 1] + c[:, 2] * np.sin(th), w * c[:, 3]], 1)[~isb]
        pts.append(out)
        return np.concatenate(pts)

    def scan(pose_xy, n):
        if n <= 0:
            return np.zeros((0, 3))
        got, have = [], 0
        while have < n:                                              # range-dependent thinning: draw until n
            S = surface_samples(min(max(n, 1 << 16), 1 << 22))
            d = np.linalg.norm(S - np.array([pose_xy[0], pose_xy[1], eye]), axis=1)
            keep = rng.uniform(0, 1, len(S)) < np.clip((6.0 * k / np.maximum(d, 1.0 * k)) ** 1.2, 0, 1)
            got.append(S[keep]); have += int(keep.sum())
        S = np.concatenate(got)[:n]
        dirs = S - np.array([pose_xy[0], pose_xy[1], eye])
        dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        return S + dirs * rng.normal(scale=delta, size=(len(S), 1))

    P = scan((-shift / 2, 0.0), n_points if first_scan_points is None else first_scan_points)
    Qw = scan((shift / 2, 0.0), n_points if second_scan_points is None else second_scan_points)
    yaw = np.deg2rad(yaw_deg)
    R = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]])
    t = np.array([shift, 0.0, 0.0])
    Q = (Qw - t) @ R            # second scanner frame: q = R^T (w - t)
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
    return P.astype(np.float32), Q.astype(np.float32), T


def lidar_pair_scaled(scale, delta=0.05, seed=5, **kw):
    """The configs[3] pair at a fraction of its size, geometrically similar (every length x sqrt(scale): scene 30 m, poses
    10 m apart, box and cylinder sizes; same 200 boxes and 50 cylinders) with 5 M x scale returns per scan: same point
    density, same structure per square metre of sample.  Only delta (the noise) is not scaled.  Use with a sample of
    20 000 x scale."""
    r = float(np.sqrt(scale))
    return lidar_pair(int(5_000_000 * scale), delta=delta, seed=seed, extent=30.0 * r, shift=10.0 * r, **kw)


def part_in_whole_scaled(scale, delta=0.05, seed=9):
    """configs[4] at a fraction of its size, geometrically similar (lengths x sqrt(scale), incl. the 3 m ball of the query):
    10 M x scale scene points, 100 k x scale query points.  Use with a sample of 5000 x scale."""
    r = float(np.sqrt(scale))
    return part_in_whole_pair(int(10_000_000 * scale), int(100_000 * scale), delta=delta, seed=seed, radius=3.0 * r,
                              spot=(2.0 * r, 1.5 * r, 1.0 * r), extent=30.0 * r, shift=10.0 * r)


def part_in_whole_pair(n_scene, n_query, delta=0.05, seed=9, radius=3.0, spot=(2.0, 1.5, 1.0), extent=30.0, shift=10.0, n_boxes=200, n_cyl=50):
    """Config-5 style pair (BASELINE.json configs[4]): a `n_query`-point query cut out of a second scan (everything
    within `radius` metres of `spot`, in world coordinates: SURVEY.md 8d's 3 m ball) against a `n_scene`-point scene;
    P = scene, Q = query in the second scanner's frame.  Returns (P, Q, T_gt)."""
    P, _, T = lidar_pair(n_scene, delta=delta, seed=seed, second_scan_points=0, extent=extent, shift=shift, n_boxes=n_boxes, n_cyl=n_cyl)
    _, Qfull, _ = lidar_pair(max(24 * n_query, 1 << 16), delta=delta, seed=seed, first_scan_points=0, extent=extent, shift=shift,
                             n_boxes=n_boxes, n_cyl=n_cyl)   # same scene, same poses
    Qw = Qfull.astype(np.float64) @ T[:3, :3].T + T[:3, 3]
    near = np.nonzero(np.linalg.norm(Qw - np.asarray(spot), axis=1) < radius)[0]
    if len(near) > n_query:
        near = np.sort(np.random.default_rng(seed + 1).choice(near, n_query, replace=False))
    return P, Qfull[near], T
