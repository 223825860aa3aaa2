// s4p_k_rigid.hip.hpp -- part of the gfx950 device code (included by s4p_kernels.hip.hpp, in this order; one translation unit):
// ComputeRigidTransformation + the rms / Euler-angle gate.
#pragma once

namespace s4p {

// ---------------------------------------------------------------------------
// ComputeRigidTransformation (match4pcsBase.cc:365-500), computeScale == false.
// Returns true iff "ok && rms >= 0 && rms < 2*delta" (match4pcsBase.hpp:436-439).
// T is row-major 3x4 (R | t).
// ---------------------------------------------------------------------------
struct BaseFrame {          // per-base constants, computed once on the host side of the ABI
  float p[3][3];            // first three base points (sampled P, centred)
  float c1[3];              // centroid1 = ((b1+b2)+b3)/3
  float gate;               // distance_factor * delta = 2*delta
  float max_angle_rad;      // options.max_angle * pi / 180 as a float (match4pcsBase.hpp:426)
  int angle_gate;           // options.max_angle >= 0: the Euler-angle bound of match4pcsBase.cc:457-472 is in force
  float angle_tol;          // device margin of that bound, 1e-6 rad (S4P_ANGLE_TOL widens it: a test aid that sends more candidates to the host)
};

__host__ __device__ __forceinline__ bool gs_frame(const float* a0, const float* a1, const float* a2, float e[3][3]) {
  e[0][0] = a1[0] - a0[0]; e[0][1] = a1[1] - a0[1]; e[0][2] = a1[2] - a0[2];
  if (sqn3(e[0][0], e[0][1], e[0][2]) == 0.f) return false;
  normalize3(e[0][0], e[0][1], e[0][2]);
  const float tx = a2[0] - a0[0], ty = a2[1] - a0[1], tz = a2[2] - a0[2];
  const float dd = dot3(tx, ty, tz, e[0][0], e[0][1], e[0][2]);
  e[1][0] = tx - dd * e[0][0]; e[1][1] = ty - dd * e[0][1]; e[1][2] = tz - dd * e[0][2];
  if (sqn3(e[1][0], e[1][1], e[1][2]) == 0.f) return false;
  normalize3(e[1][0], e[1][1], e[1][2]);
  cross3(e[0][0], e[0][1], e[0][2], e[1][0], e[1][1], e[1][2], e[2][0], e[2][1], e[2][2]);
  if (sqn3(e[2][0], e[2][1], e[2][2]) == 0.f) return false;
  normalize3(e[2][0], e[2][1], e[2][2]);
  return true;
}

// The Euler-angle bound (match4pcsBase.cc:457-472):
//   |atan2f(R21, R22)| <= a  &&  |atan2(-R20, sqrt(R21^2 + R22^2))| <= a (double)  &&  |atan2(R10, R00)| <= a (double).
// Its outcome depends on libm's last bits, which device code cannot reproduce.  HOST: the reference's expression, libm --
// exact.  DEVICE: the three angles in double with a margin of 1e-6 rad (libm's float atan2f is within 1.5 ulp of the true
// angle: < 6e-7 at pi): 1 = passes for certain, 0 = fails for certain, 2 = within the margin -- such a candidate is
// scored but left out of the device's selection, and the host settles it with the exact expression (s4p_capi.hip,
// settle_borderline): about one candidate in 10^6.
constexpr uint32_t kBorderFlag = 0x80000000u;      // in cand_idx: the gate of this candidate is undecided
constexpr uint32_t kBorderCap = 1024;              // undecided candidates a pass can hand to the host
__host__ __device__ inline int euler_verdict(const float R[3][3], const float max_angle, const float angle_tol) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double m = double(max_angle), tol = double(angle_tol);
  const double r21 = double(R[2][1]), r22 = double(R[2][2]);
  const double a1 = fabs(atan2(r21, r22));
  const double a2 = fabs(atan2(double(-R[2][0]), sqrt(r21 * r21 + r22 * r22)));
  const double a3 = fabs(atan2(double(R[1][0]), double(R[0][0])));
  const double hi = fmax(a1, fmax(a2, a3));
  if (hi > m + tol) return 0;
  return hi <= m - tol ? 1 : 2;
#else
  const bool ok = std::abs(std::atan2(R[2][1], R[2][2])) <= max_angle &&
                  std::abs(std::atan2(double(-R[2][0]), std::sqrt(std::pow(double(R[2][1]), 2) + std::pow(double(R[2][2]), 2)))) <= double(max_angle) &&
                  std::abs(::atan2(double(R[1][0]), double(R[0][0]))) <= double(max_angle);
  return ok ? 1 : 0;
#endif
}

// 0: rejected ("!ok || !(0 <= rms < 2 delta)", match4pcsBase.hpp:436-439); 1: a candidate; 2 (device only): a candidate if
// the Euler-angle bound holds, which the host has to settle.
// ANGLE: compiled with the Euler-angle bound (three double-precision atan2: ~70 extra registers in a kernel that inlines
// it, so the variants without it stay the ones launched when options.max_angle < 0).
template <bool ANGLE>
__host__ __device__ __forceinline__ int rigid_verdict(const BaseFrame& b, const float q[3][3], float T[12], float c2[3]) {
  for (int k = 0; k < 3; ++k) c2[k] = ((q[0][k] + q[1][k]) + q[2][k]) / 3.f;    // match4pcsBase.hpp:415-417
  float vp[3][3], vq[3][3];
  if (!gs_frame(b.p[0], b.p[1], b.p[2], vp)) return 0;       // rms = 1e9 -> gate fails (quirk .cc:417-433)
  if (!gs_frame(q[0], q[1], q[2], vq)) return 0;
  float R[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) R[r][c] = vp[0][r] * vq[0][c] + (vp[1][r] * vq[1][c] + vp[2][r] * vq[2][c]);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float dg = R[i][0] * R[0][i] + (R[i][1] * R[1][i] + R[i][2] * R[2][i]);   // (R*R).diagonal() .cc:453
    if (dg - 1.f > 1e-6f) return 0;
  }
  int verdict = 1;
  if (ANGLE && b.angle_gate) {                                                     // .cc:457-472 (uniform over the launch)
    verdict = euler_verdict(R, b.max_angle_rad, b.angle_tol);
    if (verdict == 0) return 0;
  }
  float rms = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float f0 = 1.f * q[i][0] - c2[0], f1 = 1.f * q[i][1] - c2[1], f2 = 1.f * q[i][2] - c2[2];
    float d[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float tr = R[r][0] * f0 + (R[r][1] * f1 + R[r][2] * f2);
      d[r] = (tr - b.p[i][r]) + b.c1[r];
    }
    rms += sqrtf(sqn3(d[0], d[1], d[2]));
  }
  rms /= 4.f;                                                                      // .cc:489 (quirk: /4 over 3 terms)
  if (!(rms >= 0.f && rms < b.gate)) return 0;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float rc = R[r][0] * (-c2[0]) + (R[r][1] * (-c2[1]) + R[r][2] * (-c2[2]));
    T[r * 4 + 0] = R[r][0]; T[r * 4 + 1] = R[r][1]; T[r * 4 + 2] = R[r][2];
    T[r * 4 + 3] = b.c1[r] + rc;
  }
  return verdict;
}
__host__ __device__ __forceinline__ bool rigid_gate(const BaseFrame& b, const float q[3][3], float T[12], float c2[3]) {
  return rigid_verdict<false>(b, q, T, c2) != 0;            // (without the Euler-angle bound: the transform of a candidate that is known to have passed)
}

}  // namespace s4p
