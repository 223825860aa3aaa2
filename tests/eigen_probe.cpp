// eigen_probe.cpp -- closes the one "unpinned" residue of the parity chain.
//
// The reference evaluates its per-point arithmetic through Eigen 3.3 fixed-size expressions (shared4pcs.h:49,
// match4pcsBase.cc:416-497,532, pairCreationFunctor.h:161-209, super4pcs.cc:110-160, normalset.hpp:181-191).  Eigen is an
// un-vendored, un-pinned submodule of the reference and is absent from the build image, so the oracle, the device kernels
// and oracle/eigen_shim all implement the evaluation ORDERS listed below, derived from Eigen 3.3's sources
// (redux_novec_unroller, the lazy coefficient-based product, the 4-row packet path of Matrix4f * Vector4f).  A real Eigen
// that evaluated differently could move a last bit and flip an inlier sitting exactly on the delta^2 boundary.
//
// This program evaluates each expression THROUGH THE EIGEN API, written exactly as the reference's call sites write it, on
// adversarial inputs (full mantissas, mixed exponents, cancellation), and compares the bits with the asserted order
// written out in scalar code.  Anyone with a real Eigen >= 3.3 runs
//     g++ -O2 -ffp-contract=off -I/path/to/eigen tests/eigen_probe.cpp -o eigen_probe && ./eigen_probe
// (the flags of a plain CMake Release build of the reference on x86-64 plus -ffp-contract=off, which GCC needs on targets
// with FMA; add -DNDEBUG freely) and reads one line per expression; exit status 0 = every order confirmed.  Here it is
// compiled against oracle/eigen_shim by tests/test_eigen_probe.py, which proves the probe itself and the shim agree with
// the asserted orders, and pins the printed digest.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include <Eigen/Core>
#include <Eigen/Geometry>

namespace {
using V3 = Eigen::Matrix<float, 3, 1>;
using M3 = Eigen::Matrix<float, 3, 3>;
using M4 = Eigen::Matrix<float, 4, 4>;

uint32_t g_state = 0x2545F491u;
uint32_t lcg() { g_state = g_state * 1664525u + 1013904223u; return g_state; }
// full-mantissa float in [-2^e, 2^e), e drawn from {-6 .. 3}: sums of such terms round differently under reassociation
float adversarial() {
  const uint32_t m = lcg() >> 9;
  const int e = int(lcg() >> 28) % 10 - 6;
  const float f = std::ldexp(1.0f + float(m) * (1.0f / 8388608.0f), e);
  return (lcg() & 0x80000000u) ? -f : f;
}
V3 vec() { return V3(adversarial(), adversarial(), adversarial()); }
M3 mat3() { M3 m; for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) m(r, c) = adversarial(); return m; }
uint32_t bits(float f) { uint32_t b; std::memcpy(&b, &f, 4); return b; }

struct Tally { const char* name; const char* where; long n = 0, bad = 0; uint64_t digest = 1469598103934665603ull; };
void record(Tally& t, float eigen, float asserted) {
  ++t.n;
  if (bits(eigen) != bits(asserted)) ++t.bad;
  t.digest = (t.digest ^ bits(eigen)) * 1099511628211ull;
}
int report(const Tally& t) {
  std::printf("%-34s %-58s %6ld cases  %s  digest %016llx\n", t.name, t.where, t.n, t.bad ? "MISMATCH" : "confirmed",
              (unsigned long long)t.digest);
  if (t.bad) std::printf("    -> %ld of %ld results differ from the asserted order: this Eigen evaluates it differently\n", t.bad, t.n);
  return t.bad ? 1 : 0;
}
}  // namespace

int main() {
  const int N = 20000;
  int failures = 0;

  {  // 1. squaredNorm / dot / norm of a 3-vector: x + (y + z)
    Tally t{"1 squaredNorm, dot, norm", "match4pcsBase.cc:176,200-210; kdtree.h:417; super4pcs.cc:160"};
    for (int i = 0; i < N; ++i) {
      const V3 a = vec(), b = vec();
      record(t, a.squaredNorm(), a(0) * a(0) + (a(1) * a(1) + a(2) * a(2)));
      record(t, a.dot(b), a(0) * b(0) + (a(1) * b(1) + a(2) * b(2)));
      record(t, (a - b).norm(), std::sqrt((a(0) - b(0)) * (a(0) - b(0)) + ((a(1) - b(1)) * (a(1) - b(1)) + (a(2) - b(2)) * (a(2) - b(2)))));
    }
    failures += report(t);
  }
  {  // 2. normalized(): v / sqrt(squaredNorm), three correctly rounded divisions (not a multiplication by the reciprocal)
    Tally t{"2 normalized", "match4pcsBase.cc:416-434; super4pcs.cc:110-111,121,144"};
    for (int i = 0; i < N; ++i) {
      const V3 a = vec();
      const V3 n = a.normalized();
      const float s = std::sqrt(a(0) * a(0) + (a(1) * a(1) + a(2) * a(2)));
      for (int k = 0; k < 3; ++k) record(t, n(k), a(k) / s);
    }
    failures += report(t);
  }
  {  // 3. cross product, as written in Eigen's cross()
    Tally t{"3 cross", "match4pcsBase.cc:200-210,430; normalset.hpp:181"};
    for (int i = 0; i < N; ++i) {
      const V3 a = vec(), b = vec();
      const V3 c = a.cross(b);
      record(t, c(0), a(1) * b(2) - a(2) * b(1));
      record(t, c(1), a(2) * b(0) - a(0) * b(2));
      record(t, c(2), a(0) * b(1) - a(1) * b(0));
    }
    failures += report(t);
  }
  {  // 4. 3x3 products and Matrix3f * Vector3f: every coefficient a0*b0 + (a1*b1 + a2*b2)
    Tally t{"4 Matrix3f*Matrix3f, *Vector3f", "match4pcsBase.cc:449,453,480"};
    for (int i = 0; i < N / 4; ++i) {
      const M3 A = mat3(), B = mat3();
      const V3 v = vec();
      const M3 C = A.transpose() * B;                        // rotation = rotate_p.transpose() * rotate_q   (:449)
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) record(t, C(r, c), A(0, r) * B(0, c) + (A(1, r) * B(1, c) + A(2, r) * B(2, c)));
      const V3 d = (A * A).diagonal();                        // (rotation * rotation).diagonal()            (:453)
      for (int k = 0; k < 3; ++k) record(t, d(k), A(k, 0) * A(0, k) + (A(k, 1) * A(1, k) + A(k, 2) * A(2, k)));
      const V3 w = A * v;                                     // rotation * (scale * q - centroid2)          (:480)
      for (int r = 0; r < 3; ++r) record(t, w(r), A(r, 0) * v(0) + (A(r, 1) * v(1) + A(r, 2) * v(2)));
    }
    failures += report(t);
  }
  {  // 5. (Matrix4f * v.homogeneous()).head<3>(): ((m0*x + m1*y) + m2*z) + m3  -- Verify and the final apply
    Tally t{"5 Matrix4f*homogeneous (Verify)", "match4pcsBase.cc:532; match4pcsBase.hpp:266"};
    for (int i = 0; i < N; ++i) {
      M4 M = M4::Identity();
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) M(r, c) = adversarial();
      const V3 q = vec();
      const V3 p = (M * q.homogeneous()).head<3>();
      for (int r = 0; r < 3; ++r) record(t, p(r), ((M(r, 0) * q(0) + M(r, 1) * q(1)) + M(r, 2) * q(2)) + M(r, 3));
    }
    failures += report(t);
  }
  {  // 6. Transform chain of ComputeRigidTransformation: t = c1 + R * (-c2), linear part = R
    Tally t{"6 Transform scale/translate/rotate", "match4pcsBase.cc:491-497"};
    for (int i = 0; i < N / 4; ++i) {
      const M3 R = mat3();
      const V3 c1 = vec(), c2 = vec();
      Eigen::Transform<float, 3, Eigen::Affine> etrans(Eigen::Transform<float, 3, Eigen::Affine>::Identity());
      const M4 T = etrans.scale(1.0f).translate(c1).rotate(R).translate(-c2).matrix();
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) record(t, T(r, c), R(r, c));
        record(t, T(r, 3), c1(r) + (R(r, 0) * (-c2(0)) + (R(r, 1) * (-c2(1)) + R(r, 2) * (-c2(2)))));
      }
    }
    failures += report(t);
  }
  {  // 7. Quaternion::setFromTwoVectors(zhat, n) (regular branch) and quaternion * vector
    Tally t{"7 Quaternion from two vectors, *v", "normalset.hpp:181-191"};
    for (int i = 0; i < N / 2; ++i) {
      V3 n = vec();
      if (n.normalized()(2) < -0.9f) n(2) = -n(2);            // stay clear of the near-opposite branch (deviation D1)
      const V3 v = vec();
      Eigen::Quaternion<float> q;
      q.setFromTwoVectors(V3(0.f, 0.f, 1.f), n);
      const V3 v1 = n.normalized();
      const float c = 0.f * v1(0) + (0.f * v1(1) + 1.f * v1(2));
      const float ax = 0.f * v1(2) - 1.f * v1(1), ay = 1.f * v1(0) - 0.f * v1(2), az = 0.f * v1(1) - 0.f * v1(0);
      const float s = std::sqrt((1.f + c) * 2.f), invs = 1.f / s;
      const float qx = q.vec()(0), qy = q.vec()(1), qz = q.vec()(2);
      record(t, q.w(), s * 0.5f); record(t, qx, ax * invs); record(t, qy, ay * invs); record(t, qz, az * invs);
      const V3 r = q * v;                                     // QuaternionBase::_transformVector: v + w * uv + q.vec().cross(uv), uv = 2 q.vec().cross(v)
      float uv[3] = {qy * v(2) - qz * v(1), qz * v(0) - qx * v(2), qx * v(1) - qy * v(0)};
      for (float& u : uv) u += u;
      const float cr[3] = {qy * uv[2] - qz * uv[1], qz * uv[0] - qx * uv[2], qx * uv[1] - qy * uv[0]};
      for (int k = 0; k < 3; ++k) record(t, r(k), (v(k) + q.w() * uv[k]) + cr[k]);
    }
    failures += report(t);
  }
  std::printf(failures ? "RESULT: %d expression group(s) evaluate differently here: counts ON the delta^2 boundary may differ from this Eigen's binary\n"
                       : "RESULT: every asserted evaluation order confirmed for this Eigen (%d)\n", failures);
  return failures ? 1 : 0;
}
