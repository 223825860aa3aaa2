"""k_prep (set 2) takes the direction bucket of a rotated cone sample from the UN-normalised vector when that is safe
(s4p_k_prep.hip.hpp, cone_mask_row): the rotated vector is unit to rounding, so int((x / 2 + 0.5) / neps) of the normalised
vector equals int(fma(x, 0.5, 0.5) * (1 / neps)) of the raw one unless a coordinate lies within 4e-4 of an integer.  This
test restates both sequences in numpy float32 (same operation order as the kernel / normalset.hpp:110-127, 186-196) on
a few million random rotations and checks the two facts the kernel relies on: the bucket coordinates differ by far less
than the 2e-4 the margin assumes, and wherever the kernel would take the fast path the buckets are identical."""
import numpy as np

F = np.float32


def _cross(ax, ay, az, bx, by, bz):
    return ay * bz - az * by, az * bx - ax * bz, ax * by - ay * bx


def test_fast_bucket_equals_exact_bucket_wherever_it_is_taken():
    rng = np.random.default_rng(2014)
    n = 3_000_000
    q = rng.normal(size=(n, 4)).astype(F)
    q /= np.sqrt((q.astype(np.float64) ** 2).sum(1, keepdims=True)).astype(F)          # unit quaternions (w, x, y, z), float32
    # cone samples as the reference builds them: (sin A cos t, sin A sin t, cos A)
    ang = rng.uniform(0.0, np.pi / 2, n)
    th = rng.uniform(0.0, 2 * np.pi, n)
    vx, vy, vz = (np.sin(ang) * np.cos(th)).astype(F), (np.sin(ang) * np.sin(th)).astype(F), np.cos(ang).astype(F)
    q0, q1, q2, q3 = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    ux, uy, uz = _cross(q1, q2, q3, vx, vy, vz)                                       # QuaternionBase::_transformVector
    ux, uy, uz = ux + ux, uy + uy, uz + uz
    cx, cy, cz = _cross(q1, q2, q3, ux, uy, uz)
    dx, dy, dz = (vx + q0 * ux) + cx, (vy + q0 * uy) + cy, (vz + q0 * uz) + cz
    neps = F(np.float64(F(1.0) / F(7.0)) + 0.00001)
    # exact sequence: normalize3, then index_normal
    s = np.sqrt(dx * dx + (dy * dy + dz * dz))
    ex = [((c / s) / F(2.0) + F(0.5)) / neps for c in (dx, dy, dz)]
    # fast sequence: rotation by the quaternion's 3x3 matrix (float32 entries, fused multiply-adds), one fused
    # multiply-add and one multiply by 1 / neps
    inv = F(1.0) / neps
    w_, x_, y_, z_ = q0, q1, q2, q3
    xx, yy, zz, xy, xz, yz, wx, wy, wz = x_ * x_, y_ * y_, z_ * z_, x_ * y_, x_ * z_, y_ * z_, w_ * x_, w_ * y_, w_ * z_
    R = [F(1) - F(2) * (yy + zz), F(2) * (xy - wz), F(2) * (xz + wy),
         F(2) * (xy + wz), F(1) - F(2) * (xx + zz), F(2) * (yz - wx),
         F(2) * (xz - wy), F(2) * (yz + wx), F(1) - F(2) * (xx + yy)]

    def fma(a, b, c):
        return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F)
    fx = fma(R[0], vx, fma(R[1], vy, R[2] * vz)); fy = fma(R[3], vx, fma(R[4], vy, R[5] * vz)); fz = fma(R[6], vx, fma(R[7], vy, R[8] * vz))
    fa = [(c.astype(np.float64) * 0.5 + 0.5).astype(F) * inv for c in (fx, fy, fz)]
    dx, dy, dz = fx, fy, fz                                       # (the norm check below is on the fast vector)
    worst = max(float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))) for a, b in zip(ex, fa))
    assert worst < 2e-5, worst                                   # the kernel's margin assumes < 2e-4
    n2 = (dx.astype(np.float64) ** 2 + dy.astype(np.float64) ** 2 + dz.astype(np.float64) ** 2)
    frac = [t - np.floor(t) for t in fa]
    edge = np.minimum.reduce([np.minimum(f, F(1.0) - f) for f in frac])
    safe = (np.abs(n2 - 1.0) < 1e-4) & (edge > F(4e-4))
    assert 0.99 < safe.mean() < 1.0                              # ~0.2 % of the samples take the exact sequence
    same = np.ones(n, bool)
    for a, b in zip(ex, fa):
        same &= a.astype(np.int32) == b.astype(np.int32)
    assert same[safe].all()
