"""Shared helpers of the parity tests."""
import numpy as np

from super4pcs_amd import datasets as D


def small_pair(n_points=30000, delta=0.01, seed=7, overlap=0.6):
    P, Q, T = D.bumpy_pair(n_points, overlap=overlap, delta=delta, noise_sigma=0.3 * delta, seed=seed)
    return P, Q, T


def init_oracle(O, P, Q, delta, overlap, sample_size, seed=5489, **kw):
    opt = O.make_options(delta, overlap, sample_size, seed=seed, **kw)
    m = O.Matcher(opt, full_counts=True, use_kdtree=True, keep_trace=True)
    m.init(P, Q)
    return m


def random_rigid(rng, scale_t=0.2):
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = R
    T[:3, 3] = rng.uniform(-scale_t, scale_t, 3)
    return T


def attributes_for(P, Q, T_gt, seed=0):
    """Per-point normals (radial direction, rotated consistently for Q) and colours (smooth function of position in
    P's frame) so that normal / colour filters are satisfiable by the true correspondences."""
    rng = np.random.default_rng(seed)
    R = np.asarray(T_gt, np.float64)[:3, :3]
    t = np.asarray(T_gt, np.float64)[:3, 3]
    Pn = P / np.linalg.norm(P, axis=1, keepdims=True)
    Qp = Q.astype(np.float64) @ R.T + t                      # Q mapped into P's frame
    Qn_p = Qp / np.linalg.norm(Qp, axis=1, keepdims=True)
    Qn = Qn_p @ R                                            # back-rotate the normal into Q's frame
    def col(X):
        c = 0.5 + 0.5 * np.sin(6.0 * X[:, [0, 1, 2]] + np.array([0.3, 1.1, 2.0]))
        return c
    Pc = col(P.astype(np.float64)) + 0.01 * rng.normal(size=P.shape)
    Qc = col(Qp) + 0.01 * rng.normal(size=Q.shape)
    return Pn.astype(np.float32), np.clip(Pc, 0, 1).astype(np.float32), Qn.astype(np.float32), np.clip(Qc, 0, 1).astype(np.float32)


def quad_mix(quads):
    """numpy form of s4p_quad_mix (include/s4p_capi.h): uint64 per quad; checksums are sums mod 2^64."""
    q = np.asarray(quads, np.int64).reshape(-1, 4)
    m = np.uint64(0xFFFFFFFF)
    a, b, c, d = [(q[:, k].astype(np.uint64) & m) for k in range(4)]
    x = (a << np.uint64(32)) | b
    y = (c << np.uint64(32)) | d
    with np.errstate(over="ignore"):
        x = x * np.uint64(0x9E3779B97F4A7C15); x ^= x >> np.uint64(29)
        y = y * np.uint64(0xC2B2AE3D27D4EB4F); y ^= y >> np.uint64(31)
        h = (x + y) * np.uint64(0xD6E8FEB86659FD93)
    return h ^ (h >> np.uint64(32))


def checksum(quads):
    with np.errstate(over="ignore"):
        return int(np.sum(quad_mix(quads), dtype=np.uint64)) if len(quads) else 0


def small_rotation_pair(n_points=20000, delta=0.01, seed=31, degrees=(8.0, -11.0, 6.0)):
    """A pair whose true motion has small Euler angles (for max_angle): P ~ R_s Q2 + t_s with R_s = Rz Ry Rx of `degrees`."""
    P, Q, T = small_pair(n_points, delta=delta, seed=seed)
    T = np.asarray(T, np.float64)
    ax, ay, az = np.deg2rad(degrees)
    Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    Ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    Rz = np.array([[np.cos(az), -np.sin(az), 0], [np.sin(az), np.cos(az), 0], [0, 0, 1]])
    Rs = Rz @ Ry @ Rx
    Qp = Q.astype(np.float64) @ T[:3, :3].T + T[:3, 3]          # Q in P's frame
    Q2 = (Qp - np.array([0.05, -0.03, 0.02])) @ Rs               # x = Rs^T (p - t_s)  <=>  p = Rs x + t_s
    Tn = np.eye(4)
    Tn[:3, :3] = Rs
    Tn[:3, 3] = [0.05, -0.03, 0.02]
    return P, Q2.astype(np.float32), Tn.astype(np.float32)
