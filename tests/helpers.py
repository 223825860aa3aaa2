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
