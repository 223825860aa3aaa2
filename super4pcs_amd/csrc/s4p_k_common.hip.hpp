// s4p_k_common.hip.hpp -- part of the gfx950 device code (included by s4p_kernels.hip.hpp, in this order; one translation unit):
// exact-order float helpers, the per-base counters / result record, the LCP structure (LcpGrid) and its locate helpers; the -DS4P_PROF stamps.
#pragma once

namespace s4p {

// Lab build (-DS4P_PROF=1, super4pcs_amd/build.py build_variant; tools/r6/wave_prof.py): every wave of k_pairs2 / k_quads /
// k_verify keeps REFCLK stamps (s_memrealtime, 100 MHz: comparable across CUs and XCDs, which s_memtime is not) and phase
// accumulators in registers and writes them when it ends -- a store per stamp perturbs what it measures.  Read back by
// s4p_debug_prof.  Absent from the shipped library.
#if defined(S4P_PROF)
constexpr int kProfWords = 12, kProfWaves = 8192;
__device__ unsigned long long g_prof[4][kProfWords * kProfWaves];      // [0] k_pairs2 [1] k_quads [2] k_verify [3] k_verify's lean sweep: phase sums per wave (s4p_k_lcp.hip.hpp)
#define PROF_DECL unsigned long long tp_[kProfWords] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define PROF_NOW(v) do { __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_sched_barrier(0); v = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PROF_STAMP(k) PROF_NOW(tp_[k])
#define PROF_WRITE(which, wave_id) do { if ((threadIdx.x & 63u) == 0 && (wave_id) < uint32_t(kProfWaves)) for (int k_ = 0; k_ < kProfWords; ++k_) g_prof[which][(wave_id) * kProfWords + k_] = tp_[k_]; } while (0)
#else
#define PROF_DECL do { } while (0)
#define PROF_NOW(v) do { } while (0)
#define PROF_STAMP(k) do { } while (0)
#define PROF_WRITE(which, wave_id) do { } while (0)
#endif

constexpr uint32_t kNil = 0xFFFFFFFFu;
constexpr int kGroupMax = 3;             // bases one launch of each kernel of a device pass may cover ("BASE GROUPS" below)
constexpr uint32_t kGateFailed = 0xFFFFFFFFu;
constexpr int kMaskWords = 11;   // 343 direction buckets (7^3) -> 11 x 32 bit
constexpr int kMaxConeSamples = 56;

// ---------------------------------------------------------------------------
// exact-order float helpers (Eigen 3-vector reductions: x + (y + z))
// ---------------------------------------------------------------------------
__host__ __device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
  return ax * bx + (ay * by + az * bz);
}
__host__ __device__ __forceinline__ float sqn3(float x, float y, float z) { return x * x + (y * y + z * z); }
__host__ __device__ __forceinline__ void normalize3(float& x, float& y, float& z) {
  const float s2 = sqn3(x, y, z);
  if (s2 > 0.f) { const float s = sqrtf(s2); x /= s; y /= s; z /= s; }
}
__host__ __device__ __forceinline__ void cross3(float ax, float ay, float az, float bx, float by, float bz,
                                       float& ox, float& oy, float& oz) {
  ox = ay * bz - az * by;
  oy = az * bx - ax * bz;
  oz = ax * by - ay * bx;
}

// 64-bit mix of a congruent quad (indices into the sampled Q): the term of the order-independent checksums the fused
// path keeps per base (DevCounters::quad_sum / cand_sum).  Exported as s4p_quad_mix so that a checker can form the same sums.
__host__ __device__ inline unsigned long long quad_mix(int a, int b, int c, int d) {
  unsigned long long x = (static_cast<unsigned long long>(uint32_t(a)) << 32) | uint32_t(b);
  unsigned long long y = (static_cast<unsigned long long>(uint32_t(c)) << 32) | uint32_t(d);
  x *= 0x9E3779B97F4A7C15ull; x ^= x >> 29;
  y *= 0xC2B2AE3D27D4EB4Full; y ^= y >> 31;
  const unsigned long long h = (x + y) * 0xD6E8FEB86659FD93ull;
  return h ^ (h >> 32);
}

// ---------------------------------------------------------------------------
// device-resident per-base counters / result
// ---------------------------------------------------------------------------
struct DevCounters {
  uint32_t m1, m2;                  // appended pairs1 / pairs2 (contiguous: restored with one 8-byte copy by the chunk loop)
  uint32_t C;                       // verified candidates
  uint32_t best_count;              // max inlier count (verified candidates only)
  unsigned long long K;             // congruent quads FOUND: keeps counting past the capacity (64 bit: a base of a 20 000-point
                                    // sample has ~10^9), so an overflowing pass reports what the base needs
  uint32_t overflow;                // bit0 pairs1, bit1 pairs2, bit2 quads
  uint32_t n_border;                // max_angle >= 0: candidates whose Euler-angle gate the device could not decide (scored, not selected; the host settles them)
  unsigned long long best_tag;      // min tag among candidates with best_count
  unsigned long long quad_sum;      // order-independent checksums (sum of quad_mix mod 2^64) over all quads found ...
  unsigned long long cand_sum;      // ... and over the quads that passed the rms gate (fused path): parity at sizes where lists cannot be compared
  unsigned long long point_tests;   // optional instrumentation (COUNT kernels only)
  unsigned long long l0_pass, l1_pass, l2_pass;
  uint32_t done;                    // k_verify: workgroups that have published their best (last one selects the winner)
  uint32_t pruned;                  // k_sweep + k_verify: candidates abandoned because they could not beat the bound (VerifyParams::prune)
  uint32_t S;                       // k_sweep: candidates that survived the coarse sweep (the list k_verify scores when a bound is in force)
  // winner record
  int32_t best_quad[4];
  float best_T[16];
  float best_c2[3];
  uint32_t has_best;
  // written LAST into a result record, after a system-scope fence: the number of the launch that produced it.  The result
  // records live in pinned host memory (k_verify writes them there itself: no read-back copy), the host polls this word.
  uint32_t seq;
};

// ---------------------------------------------------------------------------
// LCP structure over sampled P  (replaces kd_tree_, match4pcsBase.cc:353-363).
// Uniform grid of edge h >= 1.02*delta (LcpGridHost::plan says why), three levels, all conservative supersets of the
// exact predicate "some P point with fl(dx^2+(dy^2+dz^2)) <= fl(delta^2)" (kdtree.h:417-421):
//   L0  coarse bitmap (OR of 2^s-cubes of the reach bitmap), <= 48 KB, staged in LDS;
//   L1  reach bitmap: bit(c) = some P point lies within delta + 0.01 h of the box of cell c,
//       stored as {bits, rank-prefix} records (one 8 B load gives the bit and the rank);
//   L2  per reachable cell, a 32 B header {first line, point count, 64-bit mask of the 4x4x4 sub-cells (edge h/4) that
//       some listed point can reach | the cell's integer coordinates as floats} and the list of exactly those P points
//       in 128-byte LINES of 8 points each: [x0..x3][y0..y3][z0..z3][x4..x7][y4..y7][z4..z7][32 B unused], unused slots
//       hold a far-away point.  One header load (both halves arrive together), a sub-cell bit test that drops most
//       near-misses, then per dependent step THREE 16 B loads = four points whose exact distance tests run on the
//       packed FP32 pipe (v_pk_add/mul_f32: two points per instruction).
// The reach records and range table (~2 MB per 10^5 points) are L2-cache resident; the
// point lists (~400 B per P point) stream from Infinity Cache / HBM.
// ---------------------------------------------------------------------------
struct LcpGrid {
  const uint2* reach;           // per 32-cell word: {reach bits, number of reachable cells before this word}
  const uint4* list_hdr;        // per reachable cell TWO records: {first line, point count, sub-cell reach mask lo, hi}, {float(ix), float(iy), float(iz), -}
  const float4* nbr;            // point lines: 8 float4 (128 B) per line, see above
  const uint32_t* coarse;       // coarse bitmap (global copy, staged to LDS by the kernels)
  uint32_t coarse_words;
  int cshift, cnx, cny;
  float ox, oy, oz, inv_h;
  int nx, ny, nz;
  float sq_eps;                 // fl(delta*delta)
};

// The exact stage takes 128 queued queries at a time, two per lane, so that two point lists are in flight per lane (every
// batch runs for as many dependent steps as its longest list; two lists per lane halve the steps per query: round 2,
// k_verify 0.161 -> 0.151 ms alone); the sweep takes two chunks per step so that the queue fits the LDS budget.
// 64-query chunks a sweep step locates together, i.e. reach-word gathers in flight per lane before the first is consumed.  With
// the early exit the sweep is most of the kernel and a wave spends it waiting on one dependent chain per step (LDS query ->
// LDS bitmap word -> 8-byte gather) with three waves per SIMD to hide it: 4 chunks per step against 2, same box
// (tools/r3_run15.sh): k_verify alone 0.0967 -> 0.0863 ms, 121.6 -> 125.1 M candidates/s; every candidate counted in full
// (no early exit: the exact stage dominates again) 83.5 -> 81.0 M.  The same change cut 12 of 118 vector instructions per step
// (packed locate, direct ballots) and that alone moved nothing (tools/r3_run14.sh): the sweep waits, it does not compute.
constexpr uint32_t kSweepChunks = 4;
constexpr uint32_t kSweepStep = 64u * kSweepChunks;             // queries per sweep step; the LDS query copy is padded to a multiple of it
static_assert(kSweepChunks == 2 || kSweepChunks == 4 || kSweepChunks == 8, "sweep steps of 2, 4 or 8 chunks");
// Queue of the fused / staged sweeps (wave_lcp_count, wave_lcp_count_staged).  The staged sweep touches LDS only (locate +
// coarse bitmap) and queues the L0 survivors; the reach-word gather runs later, on the queued entries, and only for candidates
// the early exit has not dismissed by then: the queue holds both kinds of entries.
constexpr int kQueueEntries = 512 + int(kSweepStep);            // reach-tested entries below, L0 survivors of the sweep above them
constexpr uint32_t kQueueHold = 512;              // the queue is drained (reach test, then exact batches) once more than this many wait
constexpr uint32_t kExactHold = 256;              // ... down to this many reach-tested entries
constexpr int kQueueWordsPerWave = kQueueEntries + kQueueEntries / 2;     // 32-bit rank + 16-bit query index per entry: 3 KB
constexpr int kCoarseMaxWords = 9216;              // 36 KB (two k_verify workgroups per CU share 160 KB: 80 KB each)
// LDS per k_verify workgroup: coarse bitmap + survivor queues (3 KB per wave) [+ quantised queries].  Two workgroups per CU
// (structure streaming from HBM, chunk passes) share 160 KB: 80 KB each.  With ONE workgroup per CU (structure cache
// resident: verify_blocks <= 256) the quantised query copy may take more -- measured with 512-entry queues at 768 threads,
// where it no longer fits 80 KB (tools/r3_run16.sh): float queries from global memory 125.5 M candidates/s, LDS copy 130.2 M.
constexpr int kVerifyLdsBudget = 80 * 1024 - 1024;      // (minus VerifyShared, 0.8 KB)
constexpr int kVerifyLdsOnePerCu = 112 * 1024 - 1024;

// value held by every lane of the wave -> SGPR
__device__ __forceinline__ float wave_uniform(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

// (mat * q.homogeneous()).head<3>() : ((m0*x + m1*y) + m2*z) + m3   (match4pcsBase.cc:532)
__device__ __forceinline__ void transform_point(const float* T, const float4 q, float& tx, float& ty, float& tz) {
  tx = ((T[0] * q.x + T[1] * q.y) + T[2] * q.z) + T[3];
  ty = ((T[4] * q.x + T[5] * q.y) + T[6] * q.z) + T[7];
  tz = ((T[8] * q.x + T[9] * q.y) + T[10] * q.z) + T[11];
}

// Candidate transform in GRID units: U = diag(1/h) * (T - origin), so that floor(U * [q;1]) is the cell of the
// transformed query.  Evaluated with fused multiply-adds: nine instructions per query instead of the 30 of "exact
// transform, subtract origin, scale", and that is what stage 1 spends most of its time on.  It only LOCATES the query:
// the result may differ from the exactly rounded cell coordinate by ~1e-5 cell, which the structure absorbs by
// construction (a cell lists every P point within delta + 0.01 h of its box, LcpGridHost::plan; the 4x4x4 sub-cell masks
// carry the same 1 % slack).  The inlier predicate itself (exact_setup) uses the exact, un-fused transform_point.
// Only the coarse copy lives across the query loop (12 registers); the exact 3x4 and the fine-unit transform are
// re-derived from the candidate's record where the dense stages need them -- keeping all 36 values live cost ~25 % of
// the loop's instructions in scalar-register spills.
struct GridXf { float u[12]; };
__device__ __forceinline__ void load_rows(const float4* Tsrc, float T[12]) {      // 3x4 row-major; same address in every lane
  const float4 r0 = Tsrc[0], r1 = Tsrc[1], r2 = Tsrc[2];
  T[0] = r0.x; T[1] = r0.y; T[2] = r0.z; T[3] = r0.w; T[4] = r1.x; T[5] = r1.y; T[6] = r1.z; T[7] = r1.w;
  T[8] = r2.x; T[9] = r2.y; T[10] = r2.z; T[11] = r2.w;
}
__device__ __forceinline__ GridXf make_grid_xf(const LcpGrid& g, const float* T, const float scale) {   // scale: 1 or 2^-cshift
  GridXf X;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float o = r == 0 ? g.ox : (r == 1 ? g.oy : g.oz);
    X.u[4 * r + 0] = (T[4 * r + 0] * g.inv_h) * scale;
    X.u[4 * r + 1] = (T[4 * r + 1] * g.inv_h) * scale;
    X.u[4 * r + 2] = (T[4 * r + 2] * g.inv_h) * scale;
    X.u[4 * r + 3] = ((T[4 * r + 3] - o) * g.inv_h) * scale;
  }
  return X;
}
__device__ __forceinline__ float coarse_scale(const LcpGrid& g) { return __builtin_bit_cast(float, (127u - uint32_t(g.cshift)) << 23); }   // 2^-cshift
// a * b + c on the low 24 bits of a and b, full rate (the compiler turns __umul24(a, b) + c into the quarter-rate
// v_mad_u64_u32 here)
__device__ __forceinline__ uint32_t mad24(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r;
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// the same with a wave-uniform multiplier taken straight from a scalar register (no v_mov per use)
__device__ __forceinline__ uint32_t mad24_s(uint32_t a, uint32_t b_uniform, uint32_t c) {
  uint32_t r;
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b_uniform), "v"(c));
  return r;
}
__device__ __forceinline__ uint32_t mul24_s(uint32_t a, uint32_t b_uniform) {
  uint32_t r;
  asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "s"(b_uniform), "v"(a));
  return r;
}
// 16-bit pairs (round 6, the lean sweep's cube coordinates): {lo, hi} -> {u16(RNE(clamp(lo, 0, 1) * 65535)), u16(...hi...)} in one
// instruction (NaN -> 0); per-half unsigned minimum against a wave-uniform pair; a.lo * k.lo + a.hi * k.hi + c.
__device__ __forceinline__ uint32_t cvt_pknorm_u16(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pknorm_u16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
__device__ __forceinline__ uint32_t pk_min_u16_s(uint32_t a, uint32_t b_uniform) {
  uint32_t r;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b_uniform));
  return r;
}
__device__ __forceinline__ uint32_t dot2_u16_s(uint32_t a, uint32_t k_uniform, uint32_t c) {
  uint32_t r;
  asm("v_dot2_u32_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(k_uniform), "v"(c));
  return r;
}
// inclusive prefix sum over the wave on the DPP pipe: row_shr 1, 2, 4, 8 inside the rows of 16 lanes, then row_bcast 15 and 31
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
  v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x111, 0xf, 0xf, false));
  v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x112, 0xf, 0xf, false));
  v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x114, 0xf, 0xf, false));
  v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x118, 0xf, 0xf, false));
  v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x142, 0xa, 0xf, false));
  v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x143, 0xc, 0xf, false));
  return v;
}
// floor(x) as an integer in one instruction (V_CVT_FLR_I32_F32; the compiler only emits v_floor + v_cvt)
__device__ __forceinline__ int floor_to_int(float x) {
  int i;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(i) : "v"(x));
  return i;
}
// Integer cell coordinates of a query: floor of its position in grid units (u = X.u: fine cells, X.uc: coarse cubes).
// ONE definition for every stage, so all of them see bit-identical cells (plain IEEE fma, no reassociation:
// -ffp-contract=off only forbids *implicit* fusing).
__device__ __forceinline__ void grid_cell(const float* u, const float4 q, int& ix, int& iy, int& iz) {
  ix = floor_to_int(__builtin_fmaf(u[0], q.x, __builtin_fmaf(u[1], q.y, __builtin_fmaf(u[2], q.z, u[3]))));
  iy = floor_to_int(__builtin_fmaf(u[4], q.x, __builtin_fmaf(u[5], q.y, __builtin_fmaf(u[6], q.z, u[7]))));
  iz = floor_to_int(__builtin_fmaf(u[8], q.x, __builtin_fmaf(u[9], q.y, __builtin_fmaf(u[10], q.z, u[11]))));
}

// Two queries at once on the packed-FP32 pipe (v_pk_fma_f32: one issue slot for both): the same three IEEE fma per axis and
// query in the same order, so the cells are bit-identical to grid_cell's.  The coefficients of an axis travel as two register
// pairs (a, b) and (c, d); the instruction's operand selectors broadcast one half of a pair to both lanes (op_sel picks the
// half the LOW result reads, op_sel_hi the half the HIGH result reads), so the transform occupies twelve registers as in the
// scalar form -- splatting every coefficient into a pair of its own cost twelve more and spilled.
typedef float v2f_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f_t pk_fma_lo(const v2f_t coef, const v2f_t v, const v2f_t acc) {       // coef.x * v + acc
  v2f_t r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(coef), "v"(v), "v"(acc));
  return r;
}
__device__ __forceinline__ v2f_t pk_fma_hi(const v2f_t coef, const v2f_t v, const v2f_t acc) {       // coef.y * v + acc
  v2f_t r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(coef), "v"(v), "v"(acc));
  return r;
}
__device__ __forceinline__ v2f_t pk_fma_lo_hi(const v2f_t coef, const v2f_t v) {                      // coef.x * v + coef.y
  v2f_t r;
  asm("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,0,1] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(coef), "v"(v));
  return r;
}
__device__ __forceinline__ void grid_cell2(const float* u, const float4 q0, const float4 q1, int& ix0, int& iy0, int& iz0, int& ix1, int& iy1, int& iz1) {
  const v2f_t x = {q0.x, q1.x}, y = {q0.y, q1.y}, z = {q0.z, q1.z};
  auto axis = [&](const int r) -> v2f_t {
    const v2f_t ab = {u[4 * r], u[4 * r + 1]}, cd = {u[4 * r + 2], u[4 * r + 3]};
    return pk_fma_lo(ab, x, pk_fma_hi(ab, y, pk_fma_lo_hi(cd, z)));       // fma(a, x, fma(b, y, fma(c, z, d))) for both queries
  };
  const v2f_t px = axis(0), py = axis(1), pz = axis(2);
  ix0 = floor_to_int(px.x); ix1 = floor_to_int(px.y);
  iy0 = floor_to_int(py.x); iy1 = floor_to_int(py.y);
  iz0 = floor_to_int(pz.x); iz1 = floor_to_int(pz.y);
}

}  // namespace s4p
