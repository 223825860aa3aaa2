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


def lidar_pair(n_points, delta=0.05, seed=5, extent=60.0, n_boxes=120, n_cyl=30, yaw_deg=30.0, shift=10.0,
               first_scan_points=None, second_scan_points=None):
    """Config-4 style pair: two simulated terrestrial scans of one scene (ground plane, axis-aligned
    boxes, vertical cylinders) from poses `shift` metres apart and `yaw_deg` apart, ~1/r^2 density, range noise
    sigma = delta.  Returns (P, Q, T_gt) with Q expressed in the second scanner's frame."""
    rng = np.random.default_rng(seed)
    boxes = np.concatenate([rng.uniform(-extent / 2, extent / 2, (n_boxes, 2)), rng.uniform(1.5, 6.0, (n_boxes, 3))], axis=1)
    cyls = np.concatenate([rng.uniform(-extent / 2, extent / 2, (n_cyl, 2)), rng.uniform(0.2, 0.8, (n_cyl, 1)), rng.uniform(3, 10, (n_cyl, 1))], axis=1)

    def surface_samples(n):
        pts = []
        ng = n // 2
        r = extent / 2 * np.sqrt(rng.uniform(0, 1, ng)) ** 1.5          # denser near the centre
        a = rng.uniform(0, 2 * np.pi, ng)
        pts.append(np.stack([r * np.cos(a), r * np.sin(a), np.zeros(ng)], 1))
        nb = n - ng
        which = rng.integers(0, n_boxes + n_cyl, nb)
        u, v, w = rng.uniform(-0.5, 0.5, nb), rng.uniform(-0.5, 0.5, nb), rng.uniform(0, 1, nb)
        out = np.zeros((nb, 3))
        isb = which < n_boxes
        b = boxes[np.minimum(which, n_boxes - 1)]
        face = rng.integers(0, 4, nb)
        bx = np.where(face < 2, (face * 2 - 1) * 0.5 * b[:, 2], u * b[:, 2])
        by = np.where(face < 2, v * b[:, 3], ((face - 2) * 2 - 1) * 0.5 * b[:, 3])
        out[isb] = np.stack([b[:, 0] + bx, b[:, 1] + by, w * b[:, 4]], 1)[isb]
        c = cyls[np.clip(which - n_boxes, 0, n_cyl - 1)]
        th = 2 * np.pi * (u + 0.5)
        out[~isb] = np.stack([c[:, 0] + c[:, 2] * np.cos(th), c[:, 1] + c[:, 2] * np.sin(th), w * c[:, 3]], 1)[~isb]
        pts.append(out)
        return np.concatenate(pts)

    def scan(pose_xy, n):
        if n <= 0:
            return np.zeros((0, 3))
        got, have = [], 0
        while have < n:                                              # range-dependent thinning keeps ~1/3: draw until n
            S = surface_samples(min(max(n, 1 << 16), 1 << 22))
            d = np.linalg.norm(S - np.array([pose_xy[0], pose_xy[1], 1.8]), axis=1)
            keep = rng.uniform(0, 1, len(S)) < np.clip((6.0 / np.maximum(d, 1.0)) ** 1.2, 0, 1)
            got.append(S[keep]); have += int(keep.sum())
        S = np.concatenate(got)[:n]
        dirs = S - np.array([pose_xy[0], pose_xy[1], 1.8])
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


def part_in_whole_pair(n_scene, n_query, delta=0.05, seed=9, radius=7.0, spot=(4.0, 3.0, 1.0)):
    """Config-5 style pair (BASELINE.json configs[4]): a `n_query`-point query cut out of a second scan (everything
    within `radius` metres of `spot`, in world coordinates) against a `n_scene`-point scene; P = scene, Q = query in
    the second scanner's frame.  Returns (P, Q, T_gt)."""
    P, _, T = lidar_pair(n_scene, delta=delta, seed=seed, second_scan_points=0)
    _, Qfull, _ = lidar_pair(max(24 * n_query, 1 << 16), delta=delta, seed=seed, first_scan_points=0)   # same scene, same poses
    Qw = Qfull.astype(np.float64) @ T[:3, :3].T + T[:3, 3]
    near = np.nonzero(np.linalg.norm(Qw - np.asarray(spot), axis=1) < radius)[0]
    if len(near) > n_query:
        near = np.sort(np.random.default_rng(seed + 1).choice(near, n_query, replace=False))
    return P, Qfull[near], T
