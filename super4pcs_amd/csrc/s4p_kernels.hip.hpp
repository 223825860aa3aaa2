// s4p_kernels.hip.hpp -- gfx950 device code of the Super4PCS hot path (HIP, wave64).
//
// Compiled ONLY for gfx950 with -ffp-contract=off: every float expression below is
// evaluated as separate IEEE mul/add (no FMA), with correctly rounded sqrtf and '/',
// in the exact association order the reference's Eigen 3.3 fixed-size expressions
// use (see DESIGN.md "Numerics contract").  That is what makes integer inlier counts
// bit-exact against the CPU path.
//
// Reference functions restated here (paths under /root/reference/src/super4pcs/):
//   k_pairs2       accelerators/pairExtraction/intersectionFunctor.h:197-233 (loop 2),
//                  intersectionPrimitive.h:117-157, algorithms/pairCreationFunctor.h:151-218
//   k_prep (and the append step of k_pairs2)  algorithms/super4pcs.cc:118-146, accelerators/normalset.hpp:110-127,162-203
//   k_quads        algorithms/super4pcs.cc:151-163
//   k_gate/k_quads algorithms/match4pcsBase.cc:365-500 (ComputeRigidTransformation + rms gate, match4pcsBase.hpp:436-439)
//   k_verify       match4pcsBase.cc:508-567 (Verify), accelerators/kdtree.h:417-421 (predicate),
//                  match4pcsBase.hpp:467-484 (first strictly greater LCP wins)
//   k_apply        algorithms/match4pcsBase.hpp:265-267
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace s4p {

// Lab build (-DS4P_PROF=1, super4pcs_amd/build.py build_variant; tools/r6/wave_prof.py): every wave of k_pairs2 / k_quads /
// k_verify keeps REFCLK stamps (s_memrealtime, 100 MHz: comparable across CUs and XCDs, which s_memtime is not) and phase
// accumulators in registers and writes them when it ends -- a store per stamp perturbs what it measures.  Read back by
// s4p_debug_prof.  Absent from the shipped library.
#if defined(S4P_PROF)
constexpr int kProfWords = 12, kProfWaves = 8192;
__device__ unsigned long long g_prof[3][kProfWords * kProfWaves];      // [0] k_pairs2 [1] k_quads [2] k_verify
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

// ---------------------------------------------------------------------------
// Device build of the LCP structure (s4p_set_clouds; replaces KdTree::finalize, kdtree.h:349-364,554-635).
// Counting formulation, no sort: (1) count (cell, point) incidences into a dense per-cell array, (2) per 32-cell
// word: reach bits + popcount, (3) scan -> rank prefix, (4) per reachable cell: header count + cell id, (5) scan ->
// list starts, (6) second incidence pass fills the lists through per-cell cursors, (7) sub-cell masks, coarse bitmap.
// The order of the points inside a list is whatever the atomics produce; the predicate "some listed point within
// delta" does not depend on it.
// ---------------------------------------------------------------------------
struct GridBuildParams {
  const float* px; const float* py; const float* pz; uint32_t n_p;
  float ox, oy, oz, h, inv_h; int nx, ny, nz; double reach2;
  uint32_t* cell_count;          // dense, one per cell (temporary)
  uint2* reach; uint32_t n_words;
  uint4* list_hdr; uint32_t* cell_id; uint32_t* cursor; float4* nbr;
  uint32_t* coarse; int cshift, cnx, cny;
};

// incidence (point i, neighbour k of its cell): true if the point can reach that cell's box
__device__ __forceinline__ bool grid_incidence(const GridBuildParams& P, uint32_t i, int k, uint32_t& cell) {
  const float x = P.px[i], y = P.py[i], z = P.pz[i];
  const int ix = int(floorf((x - P.ox) * P.inv_h)) + (k % 3) - 1, iy = int(floorf((y - P.oy) * P.inv_h)) + ((k / 3) % 3) - 1,
            iz = int(floorf((z - P.oz) * P.inv_h)) + (k / 9) - 1;
  if (ix < 0 || iy < 0 || iz < 0 || ix >= P.nx || iy >= P.ny || iz >= P.nz) return false;
  const double v[3] = {double(x), double(y), double(z)};
  const double lo[3] = {double(P.ox) + double(ix) * double(P.h), double(P.oy) + double(iy) * double(P.h), double(P.oz) + double(iz) * double(P.h)};
  double d2 = 0;
#pragma unroll
  for (int a = 0; a < 3; ++a) { const double hi = lo[a] + double(P.h); const double d = v[a] < lo[a] ? lo[a] - v[a] : (v[a] > hi ? v[a] - hi : 0.0); d2 += d * d; }
  cell = (uint32_t(iz) * uint32_t(P.ny) + uint32_t(iy)) * uint32_t(P.nx) + uint32_t(ix);
  return d2 <= P.reach2;
}
__global__ __launch_bounds__(256) void k_grid_count(GridBuildParams P) {
  const uint64_t total = uint64_t(P.n_p) * 27u;
  for (uint64_t t = blockIdx.x * uint64_t(blockDim.x) + threadIdx.x; t < total; t += uint64_t(gridDim.x) * blockDim.x) {
    uint32_t cell;
    if (grid_incidence(P, uint32_t(t / 27u), int(t % 27u), cell)) atomicAdd(&P.cell_count[cell], 1u);
  }
}
__global__ __launch_bounds__(256) void k_grid_words(GridBuildParams P, uint32_t* word_pop) {
  for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < P.n_words; w += gridDim.x * blockDim.x) {
    uint32_t bits = 0;
    const uint64_t ncell = uint64_t(P.nx) * P.ny * P.nz;
    for (uint32_t b = 0; b < 32; ++b) { const uint64_t c = uint64_t(w) * 32u + b; if (c < ncell && P.cell_count[c] != 0u) bits |= (1u << b); }
    P.reach[w].x = bits;
    word_pop[w] = uint32_t(__popc(bits));
  }
}
// Exclusive scan of n values (in place), *total = sum: three launches over tiles of kScanTile values -- per-tile sums,
// a single-workgroup scan of those (<= a few thousand), per-tile scan seeded with the tile's offset.  (The former single
// workgroup walked the whole array: 0.15 ms at 2 10^6 cells, linear in the grid; SURVEY 8 f1 sizes have 10^8 cells.)
constexpr uint32_t kScanTile = 4096;                 // 1024 threads x 4 values
__device__ __forceinline__ uint32_t block_exclusive_scan_1024(uint32_t v, uint32_t* s_wave, uint32_t* block_total) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const uint32_t up = uint32_t(__shfl_up(int(incl), o)); if (lane >= uint32_t(o)) incl += up; }
  if (lane == 63u) s_wave[wave] = incl;
  __syncthreads();
  if (wave == 0) {
    uint32_t w = lane < 16u ? s_wave[lane] : 0u, wi = w;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { const uint32_t up = uint32_t(__shfl_up(int(wi), o)); if (lane >= uint32_t(o)) wi += up; }
    if (lane < 16u) s_wave[lane] = wi - w;           // exclusive prefix of the wave sums
    if (lane == 15u) *block_total = wi;
  }
  __syncthreads();
  return s_wave[wave] + incl - v;
}
__global__ __launch_bounds__(1024) void k_scan_tile_sums(const uint32_t* v, uint32_t n, uint32_t* tile_sum) {
  __shared__ uint32_t s_wave[16], s_total;
  const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * 4u;
  uint32_t sum = 0;
#pragma unroll
  for (uint32_t k = 0; k < 4u; ++k) if (base + k < n) sum += v[base + k];
  (void)block_exclusive_scan_1024(sum, s_wave, &s_total);
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = s_total;
}
__global__ __launch_bounds__(1024) void k_scan_tiles(uint32_t* v, uint32_t n, const uint32_t* tile_offset) {
  __shared__ uint32_t s_wave[16], s_total;
  const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * 4u;
  uint32_t x[4], sum = 0;
#pragma unroll
  for (uint32_t k = 0; k < 4u; ++k) { x[k] = base + k < n ? v[base + k] : 0u; sum += x[k]; }
  uint32_t run = tile_offset[blockIdx.x] + block_exclusive_scan_1024(sum, s_wave, &s_total);
#pragma unroll
  for (uint32_t k = 0; k < 4u; ++k) { if (base + k < n) v[base + k] = run; run += x[k]; }
}
// single-workgroup exclusive scan of n values (in place); *total = sum: the tile sums of the scan above, small arrays
__global__ __launch_bounds__(1024) void k_scan_exclusive(uint32_t* v, uint32_t n, uint32_t* total) {
  __shared__ uint32_t s_part[1024];
  const uint32_t per = (n + 1023u) / 1024u;
  const uint32_t b0 = min(threadIdx.x * per, n), b1 = min(b0 + per, n);
  uint32_t sum = 0;
  for (uint32_t b = b0; b < b1; ++b) sum += v[b];
  s_part[threadIdx.x] = sum;
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {
    const uint32_t t = threadIdx.x >= off ? s_part[threadIdx.x - off] : 0u;
    __syncthreads();
    s_part[threadIdx.x] += t;
    __syncthreads();
  }
  uint32_t run = s_part[threadIdx.x] - sum;
  for (uint32_t b = b0; b < b1; ++b) { const uint32_t c = v[b]; v[b] = run; run += c; }
  if (threadIdx.x == 1023) *total = s_part[1023];
}
__global__ __launch_bounds__(256) void k_grid_headers(GridBuildParams P, const uint32_t* word_prefix, uint32_t* hdr_count) {
  for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < P.n_words; w += gridDim.x * blockDim.x) {
    uint32_t bits = P.reach[w].x, rank = word_prefix[w];
    P.reach[w].y = rank;
    while (bits) {
      const uint32_t b = uint32_t(__ffs(int(bits))) - 1u; bits &= bits - 1u;
      const uint32_t c = w * 32u + b;
      hdr_count[rank] = P.cell_count[c];
      P.cell_id[rank] = c;
      // coarse level: OR of the 2^cshift-cubes
      const int ix = int(c % uint32_t(P.nx)), iy = int((c / uint32_t(P.nx)) % uint32_t(P.ny)), iz = int(c / (uint32_t(P.nx) * uint32_t(P.ny)));
      const uint32_t cc = (uint32_t(iz >> P.cshift) * uint32_t(P.cny) + uint32_t(iy >> P.cshift)) * uint32_t(P.cnx) + uint32_t(ix >> P.cshift);
      atomicOr(&P.coarse[cc >> 5], 1u << (cc & 31u));
      ++rank;
    }
  }
}
constexpr float kFarAway = 3.0e38f;             // unused slots of a point line: (t - 3e38)^2 = inf, never an inlier, never a NaN
constexpr uint32_t kLinePoints = 8;              // points per 128-byte line
// float index of point slot `slot` (0..7), axis `axis` (0..2) inside a line of 32 floats
__host__ __device__ __forceinline__ uint32_t line_float(uint32_t slot, uint32_t axis) { return (slot >> 2) * 12u + axis * 4u + (slot & 3u); }

// point counts -> line counts (the scan of these places the lists)
__global__ __launch_bounds__(256) void k_lines_of(uint32_t* v, uint32_t n) {
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) v[r] = (v[r] + kLinePoints - 1u) / kLinePoints;
}
// every point slot of every line: far away
__global__ __launch_bounds__(256) void k_lines_clear(float4* nbr, uint64_t n_lines) {
  const float4 far4 = make_float4(kFarAway, kFarAway, kFarAway, kFarAway);
  for (uint64_t t = blockIdx.x * uint64_t(blockDim.x) + threadIdx.x; t < n_lines * 8u; t += uint64_t(gridDim.x) * blockDim.x) nbr[t] = far4;
}
__global__ __launch_bounds__(256) void k_grid_hdr_pack(GridBuildParams P, const uint32_t* list_start, const uint32_t* hdr_count, uint32_t n_reach) {
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n_reach; r += gridDim.x * blockDim.x) {
    const uint32_t c = P.cell_id[r];
    const int ix = int(c % uint32_t(P.nx)), iy = int((c / uint32_t(P.nx)) % uint32_t(P.ny)), iz = int(c / (uint32_t(P.nx) * uint32_t(P.ny)));
    P.list_hdr[2u * r] = make_uint4(list_start[r], hdr_count[r], 0u, 0u);
    P.list_hdr[2u * r + 1u] = make_uint4(__float_as_uint(float(ix)), __float_as_uint(float(iy)), __float_as_uint(float(iz)), 0u);
    P.cursor[r] = 0u;
  }
}
__global__ __launch_bounds__(256) void k_grid_fill(GridBuildParams P) {
  const uint64_t total = uint64_t(P.n_p) * 27u;
  float* lines = reinterpret_cast<float*>(P.nbr);
  for (uint64_t t = blockIdx.x * uint64_t(blockDim.x) + threadIdx.x; t < total; t += uint64_t(gridDim.x) * blockDim.x) {
    uint32_t cell;
    const uint32_t i = uint32_t(t / 27u);
    if (grid_incidence(P, i, int(t % 27u), cell)) {
      const uint2 w = P.reach[cell >> 5];
      const uint32_t rank = w.y + uint32_t(__popc(w.x & ((1u << (cell & 31u)) - 1u)));
      const uint32_t at = atomicAdd(&P.cursor[rank], 1u);
      float* line = lines + (size_t(P.list_hdr[2u * rank].x) + at / kLinePoints) * 32u;
      const uint32_t slot = at % kLinePoints;
      line[line_float(slot, 0)] = P.px[i]; line[line_float(slot, 1)] = P.py[i]; line[line_float(slot, 2)] = P.pz[i];
    }
  }
}

// Fills hdr.z/.w, the 4x4x4 sub-cell reach masks, from the point lists.
// bit(sx,sy,sz) = some listed point lies within `reach` of the sub-box; double precision, same slack as the lists.
struct MaskParams {
  uint4* list_hdr; const float4* nbr; const uint32_t* cell_id; uint32_t n_reach;
  float ox, oy, oz, h; int nx, ny; double reach2;
};
// One wave per reachable cell, lane = one of its 64 sub-boxes: every lane walks the cell's list (the same address in all
// lanes: one broadcast load per point) and tests its own sub-box; the mask is the ballot.  (A thread per cell ran
// 64 x list length double-precision box tests serially and loaded its list uncoalesced: 0.37 ms at 1.4 10^5 cells.)
__global__ __launch_bounds__(256) void k_build_masks(MaskParams P) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t waves = (gridDim.x * blockDim.x) >> 6;
  const float* lines = reinterpret_cast<const float*>(P.nbr);
  for (uint32_t r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; r < P.n_reach; r += waves) {
    uint4 hdr = P.list_hdr[2u * r];
    const uint32_t c = P.cell_id[r];
    const int ix = int(c % uint32_t(P.nx)), iy = int((c / uint32_t(P.nx)) % uint32_t(P.ny)), iz = int(c / (uint32_t(P.nx) * uint32_t(P.ny)));
    const double q = double(P.h) * 0.25;
    const double lo[3] = {double(P.ox) + double(ix) * double(P.h) + double(lane & 3u) * q,
                          double(P.oy) + double(iy) * double(P.h) + double((lane >> 2) & 3u) * q,
                          double(P.oz) + double(iz) * double(P.h) + double(lane >> 4) * q};
    bool hit = false;
    for (uint32_t p = 0; p < hdr.y && !__all(hit); ++p) {
      const float* line = lines + (size_t(hdr.x) + p / kLinePoints) * 32u;
      const uint32_t slot = p % kLinePoints;
      const double v[3] = {double(line[line_float(slot, 0)]), double(line[line_float(slot, 1)]), double(line[line_float(slot, 2)])};
      double d2 = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) { const double d = v[k] < lo[k] ? lo[k] - v[k] : (v[k] > lo[k] + q ? v[k] - (lo[k] + q) : 0.0); d2 += d * d; }
      hit = hit || d2 <= P.reach2;
    }
    const unsigned long long mask = __ballot(hit);
    if (lane == 0) { hdr.z = uint32_t(mask); hdr.w = uint32_t(mask >> 32); P.list_hdr[2u * r] = hdr; }
  }
}

// ---------------------------------------------------------------------------
// LCP scoring: Verify() (match4pcsBase.cc:508-567) without the early exit.
//
// Structure per wave and candidate:
//   sweep (kSweepChunks 64-query chunks per step, their gathers in flight together): position in grid units (3 converts
//     + 9 fma + 3 floor-converts, two queries per packed instruction), L0 test of the cell's coarse cube against the LDS
//     bitmap, L1 reach word (8 B gather; rejected lanes read word 0, one broadcast line); queries whose cell is reachable
//     are compacted (ballot/prefix) into the wave's LDS queue as {query, rank of the cell among the reachable ones};
//   exact stage whenever 128 entries wait (and once at the end): TWO entries per lane -- the candidate's exact 3x4, list
//     headers, 4x4x4 sub-cell masks, then the exact predicate sqdist <= delta^2 (kdtree.h:417-421) against the listed
//     points, both lists advancing together with four 16-byte loads in flight per lane and dependent step.
// The sampled-Q points the SWEEP reads live in LDS, quantised to 3 x 16 bit over Q's bounding box (8 B per query, 16 KB
// for n_Q = 2000): the 32 KB float array that every wave re-streamed for every candidate through a 32 KB L1 it shares
// with the gathers is gone from the sweep (-18 % L1 accesses).  The sweep only LOCATES a query; the quantisation moves it
// by < 2e-3 cell, inside the 1 % slack the structure is built with (LcpGridHost::plan).  The exact stage still reads
// the exact float query for the inlier predicate, and uses the SAME quantised value for the cell, so both stages
// agree bit for bit.  Clouds whose sample does not fit (n_Q > kLdsQueries) or whose extent needs more than 16 bits keep
// the float array in global memory (QLDS = false).
// What the round-2 measurements say about this kernel (profiles/r02_*): it is VALU-issue bound first
// (the vector pipes are busy 55 % of a launch: ~3200 instructions per candidate, half of them the sweep at 50 per 64
// queries, most of the rest the exact stage at ~70 per step of four point tests) and waits on its gathers second.
// Software-pipelining the sweep, flattening the exact stage into (query, point) pairs or 4-point items, finer work units,
// a queue persisting across candidates and 4-byte packed point records (a third of the walk's loads, more instructions
// per point) were all built, verified bit-exact and measured slower or equal -- they are documented there, not kept
// here.  What did help: list starts on 128-byte lines, 768-thread workgroups, two lists per lane.
// ---------------------------------------------------------------------------
constexpr int kLdsQueries = 2560;                  // sampled-Q points that fit the LDS copy (20 KB)

struct QuantQ {                  // 16-bit fixed point over the bounding box of the sampled Q (centred coordinates)
  float lo[3], step[3];          // q~ = lo + step * u, u in [0, 65535]
  const uint2* packed;           // per query (sweep order): {x | y << 16, z}
};

struct LcpTask {                 // what the scoring loop needs besides the grid
  const float4* q4;              // sampled Q (centred), packed (x,y,z,0), in the order of the sweep
  uint32_t n_q;
  QuantQ qq;                     // (QLDS kernels)
  const float4* T;               // candidate transforms: row-major 3x4 at T + t_stride * candidate
  uint32_t t_stride;             // in float4: 3 (cand_T records) or 4 (caller's 4x4 matrices)
  unsigned long long* point_tests;   // instrumentation (COUNT kernels only)
  // A candidate whose inlier count cannot EXCEED `prune` (the best count of the registration when the base was launched)
  // cannot become the best (match4pcsBase.hpp:468: strictly greater wins) and may be abandoned -- what the reference's
  // Verify does sequentially (match4pcsBase.cc:520,558-560), here with the stronger bound "confirmed inliers + queries
  // still waiting for their exact test + queries not swept yet".  Its reported count is then a lower bound (as the
  // reference's is for every candidate it abandons).  0 = every candidate is counted in full.
  uint32_t prune;
  uint32_t* pruned;                  // (k_verify) per-workgroup LDS counter of abandoned candidates, or nullptr
};

// The locating transform of a candidate: grid units, and for QLDS folded with the de-quantisation
// (X * (lo + step * u) + t = (X * diag(step)) * u + (X * lo + t)), so a query costs 3 converts + 9 fma + 3 floor-converts.
template <bool QLDS>
__device__ __forceinline__ GridXf locating_xf(const LcpGrid& g, const LcpTask& K, const float* T) {
  GridXf X = make_grid_xf(g, T, 1.f);
  if (QLDS) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float t = __builtin_fmaf(X.u[4 * r], K.qq.lo[0], __builtin_fmaf(X.u[4 * r + 1], K.qq.lo[1], __builtin_fmaf(X.u[4 * r + 2], K.qq.lo[2], X.u[4 * r + 3])));
      X.u[4 * r] *= K.qq.step[0]; X.u[4 * r + 1] *= K.qq.step[1]; X.u[4 * r + 2] *= K.qq.step[2];
      X.u[4 * r + 3] = t;
    }
  }
  return X;
}
// the sweep's view of query i: quantised coordinates as floats (QLDS) or the float point itself
template <bool QLDS>
__device__ __forceinline__ float4 sweep_query(const LcpTask& K, const uint2* s_q, const uint32_t i) {
  if (QLDS) {
    const uint2 w = s_q[i];
    return make_float4(float(w.x & 0xFFFFu), float(w.x >> 16), float(w.y), 0.f);
  }
  return K.q4[i];
}

// One queue entry prepared for the point loop: the exact transformed query and its range of point GROUPS (four points,
// three 16-byte loads; group gi lives at float4 index (gi >> 1) * 8 + (gi & 1) * 3) -- empty if the entry is not valid or
// its sub-cell cannot be reached.
struct ExactEntry { float tx, ty, tz; uint32_t p, e; };
// (exact_setup_q: the query point handed over by the caller; exact_setup: read from the float array by index)
template <bool COUNT>
__device__ __forceinline__ ExactEntry exact_setup_q(const LcpGrid& g, const LcpTask& K, const float* T, const bool valid, const float4 q, const uint32_t rank) {
  ExactEntry E;
  E.tx = E.ty = E.tz = 0.f; E.p = E.e = 0u;
  if (valid) {
    const uint4 hdr = g.list_hdr[2u * rank], cel = g.list_hdr[2u * rank + 1u];     // one 32-byte record: both halves arrive together
    transform_point(T, q, E.tx, E.ty, E.tz);                    // exact (reference order, no fma)
    const float rx = (E.tx - g.ox) * g.inv_h - __uint_as_float(cel.x), ry = (E.ty - g.oy) * g.inv_h - __uint_as_float(cel.y),
                rz = (E.tz - g.oz) * g.inv_h - __uint_as_float(cel.z);
    const uint32_t sx = uint32_t(min(max(int(rx * 4.f), 0), 3)), sy = uint32_t(min(max(int(ry * 4.f), 0), 3)),
                   sz = uint32_t(min(max(int(rz * 4.f), 0), 3));
    const uint32_t sb = sz * 16u + sy * 4u + sx;
    const uint32_t mword = sb < 32u ? hdr.z : hdr.w;
    if ((mword >> (sb & 31u)) & 1u) {
      if (COUNT) { atomicAdd(K.point_tests + 3, 1ull); atomicAdd(K.point_tests, (unsigned long long)hdr.y); }   // l2_pass, listed points
      E.p = 2u * hdr.x; E.e = E.p + (hdr.y + 3u) / 4u;
    }
  }
  return E;
}
template <bool COUNT>
__device__ __forceinline__ ExactEntry exact_setup(const LcpGrid& g, const LcpTask& K, const float* T, const bool valid, const uint32_t i, const uint32_t rank) {
  ExactEntry E;
  E.tx = E.ty = E.tz = 0.f; E.p = E.e = 0u;
  if (valid) {
    const uint4 hdr = g.list_hdr[2u * rank], cel = g.list_hdr[2u * rank + 1u];     // one 32-byte record: both halves arrive together
    const float4 q = K.q4[i];
    transform_point(T, q, E.tx, E.ty, E.tz);                    // exact (reference order, no fma)
    // sub-cell of the exact point inside the cell the sweep LOCATED (its coordinates travel in the header), clamped: the
    // exact point can sit a few 1e-3 cell outside it, which the masks' slack covers (LcpGridHost::plan)
    const float rx = (E.tx - g.ox) * g.inv_h - __uint_as_float(cel.x), ry = (E.ty - g.oy) * g.inv_h - __uint_as_float(cel.y),
                rz = (E.tz - g.oz) * g.inv_h - __uint_as_float(cel.z);
    const uint32_t sx = uint32_t(min(max(int(rx * 4.f), 0), 3)), sy = uint32_t(min(max(int(ry * 4.f), 0), 3)),
                   sz = uint32_t(min(max(int(rz * 4.f), 0), 3));
    const uint32_t sb = sz * 16u + sy * 4u + sx;
    const uint32_t mword = sb < 32u ? hdr.z : hdr.w;
    if ((mword >> (sb & 31u)) & 1u) {
      if (COUNT) { atomicAdd(K.point_tests + 3, 1ull); atomicAdd(K.point_tests, (unsigned long long)hdr.y); }   // l2_pass, listed points
      E.p = 2u * hdr.x; E.e = E.p + (hdr.y + 3u) / 4u;
    }
  }
  return E;
}
// sqdist <= delta^2 (kdtree.h:417-421) of one transformed query against the four points of a group: the reference's
// x*x + (y*y + z*z) per point, two points per instruction on the packed FP32 pipe (separately rounded mul / add)
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bool group_hit(const float4 X, const float4 Y, const float4 Z, const float tx, const float ty, const float tz, const float sq_eps) {
  const v2f tx2 = {tx, tx}, ty2 = {ty, ty}, tz2 = {tz, tz};
  const v2f dx0 = tx2 - v2f{X.x, X.y}, dx1 = tx2 - v2f{X.z, X.w};
  const v2f dy0 = ty2 - v2f{Y.x, Y.y}, dy1 = ty2 - v2f{Y.z, Y.w};
  const v2f dz0 = tz2 - v2f{Z.x, Z.y}, dz1 = tz2 - v2f{Z.z, Z.w};
  const v2f s0 = dx0 * dx0 + (dy0 * dy0 + dz0 * dz0), s1 = dx1 * dx1 + (dy1 * dy1 + dz1 * dz1);
  return (s0.x <= sq_eps) | (s0.y <= sq_eps) | (s1.x <= sq_eps) | (s1.y <= sq_eps);
}
// Exact stage for up to 128 queue entries, two per lane (A, B): both lists advance together, one group (four points) of
// each per dependent step, six 16-byte loads in flight per lane.  Returns which of this lane's two queries are inliers.
template <bool COUNT>
__device__ __forceinline__ uint32_t exact_pair(const LcpGrid& g, const LcpTask& K, const float4* Tsrc,
                                               const bool validA, const uint32_t iA, const uint32_t rankA,
                                               const bool validB, const uint32_t iB, const uint32_t rankB) {
  ExactEntry A, B;
  { float T[12]; load_rows(Tsrc, T);
    A = exact_setup<COUNT>(g, K, T, validA, iA, rankA);
    B = exact_setup<COUNT>(g, K, T, validB, iB, rankB); }
  uint32_t hits = 0;
  while (A.p < A.e || B.p < B.e) {
    const bool la = A.p < A.e, lb = B.p < B.e;
    const uint32_t ia = la ? (A.p >> 1) * 8u + (A.p & 1u) * 3u : 0u, ib = lb ? (B.p >> 1) * 8u + (B.p & 1u) * 3u : 0u;
    const float4 ax = g.nbr[ia], ay = g.nbr[ia + 1u], az = g.nbr[ia + 2u];
    const float4 bx = g.nbr[ib], by = g.nbr[ib + 1u], bz = g.nbr[ib + 2u];
    const bool ha = la && group_hit(ax, ay, az, A.tx, A.ty, A.tz, g.sq_eps);
    const bool hb = lb && group_hit(bx, by, bz, B.tx, B.ty, B.tz, g.sq_eps);
    if (ha) { hits |= 1u; A.p = A.e; } else if (la) A.p += 1u;
    if (hb) { hits |= 2u; B.p = B.e; } else if (lb) B.p += 1u;
  }
  return hits;                                           // bit 0: this lane's first query is an inlier, bit 1: its second
}

// Number of sampled-Q points the candidate at Tsrc brings within delta of a sampled-P point, for one wave64.
//   s_coarse: LDS copy of the coarse bitmap; s_q: LDS copy of the quantised queries (QLDS); s_queue: this wave's
//   private LDS queue (kQueueEntries entries: 32-bit ranks, then 16-bit query indices)
template <bool COUNT, bool SKIP_FINE, bool QLDS>
__device__ __forceinline__ uint32_t wave_lcp_count(const LcpGrid& g, const LcpTask& K, const uint32_t* s_coarse, const uint2* s_q,
                                                   uint32_t* s_queue, const float4* Tsrc) {
  constexpr uint32_t kNone = 0xFFFFFFFFu;
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t* q_rank = s_queue;                                                     // queue: ranks (32 bit) ...
  uint16_t* q_idx = reinterpret_cast<uint16_t*>(s_queue + kQueueEntries);         // ... and query indices (n_Q < 65536)
  uint32_t cnt = 0, qn = 0;
  const uint32_t cmax = g.coarse_words * 32u - 1u;
  const uint32_t unx = uint32_t(g.nx), uny = uint32_t(g.ny), unz = uint32_t(g.nz);
  const uint32_t ucx = uint32_t(g.cnx), ucy = uint32_t(g.cny);
  GridXf X;                                              // locating transform: the only one live across the sweep
  { float T[12]; load_rows(Tsrc, T); X = locating_xf<QLDS>(g, K, T); }
  // cell of query i under the candidate, or kNone if it falls outside the grid or into a coarse cube nothing can reach
  // (float -> int conversion saturates and one unsigned compare per axis covers both bounds; a NaN coordinate maps to
  // cell 0 and then fails every exact distance test, so it cannot create an inlier)
  auto locate = [&](const int ix, const int iy, const int iz, const uint32_t i) -> uint32_t {
    const bool inb = (uint32_t(ix) < unx) & (uint32_t(iy) < uny) & (uint32_t(iz) < unz) & (i < K.n_q);
    const uint32_t cc = min(mad24(mad24(uint32_t(iz) >> g.cshift, ucy, uint32_t(iy) >> g.cshift), ucx, uint32_t(ix) >> g.cshift), cmax);
    const uint32_t bit = (s_coarse[cc >> 5] >> (cc & 31u)) & 1u;
    // 24-bit multiplies are full rate (a 32-bit v_mul_lo is not); LcpGridHost::plan keeps nx and ny*nz below 2^24
    const uint32_t c = mad24(mad24(uint32_t(iz), uny, uint32_t(iy)), unx, uint32_t(ix));
    return (inb & (bit != 0u)) ? c : kNone;
  };
  auto lds_fence = [&]() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };
  // L1 for one chunk + compaction
  auto push = [&](const uint32_t c, const uint2 w, const uint32_t i) {
    const uint32_t sh = c & 31u;
    const bool reach = (c != kNone) & (((w.x >> sh) & 1u) != 0u);
    const unsigned long long m = __builtin_amdgcn_ballot_w64(reach);       // (the mask itself: __ballot goes through a 0/1 select and a second compare)
    if (COUNT) {
      const unsigned long long m0 = __ballot(c != kNone);
      if (lane == 0) { atomicAdd(K.point_tests + 1, (unsigned long long)__popcll(m0)); atomicAdd(K.point_tests + 2, (unsigned long long)__popcll(m)); }
    }
    if (m == 0ull) return;
    if (reach) {
      const uint32_t at = qn + __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));
      q_rank[at] = w.y + uint32_t(__popc(w.x & ((1u << sh) - 1u)));
      q_idx[at] = uint16_t(i);
    }
    qn += uint32_t(__popcll(m));
  };
  const uint32_t last = K.n_q - 1u;
  bool abandoned = false;
  for (uint32_t base = 0;; base += kSweepStep) {
    const bool more = base < K.n_q;                      // wave-uniform
    if (more) {                                          // one step: kSweepChunks chunks, all their loads in flight together
      uint32_t ii[kSweepChunks], cc[kSweepChunks];
      uint2 ww[kSweepChunks];
#pragma unroll
      for (uint32_t k = 0; k < kSweepChunks; k += 2u) {
        ii[k] = base + lane + 64u * k; ii[k + 1u] = ii[k] + 64u;
        // (the LDS copy is padded to a multiple of a step: no index clamp; locate() rejects i >= n_q)
        const float4 q0 = sweep_query<QLDS>(K, s_q, QLDS ? ii[k] : min(ii[k], last));
        const float4 q1 = sweep_query<QLDS>(K, s_q, QLDS ? ii[k + 1u] : min(ii[k + 1u], last));
        int ix0, iy0, iz0, ix1, iy1, iz1;
        grid_cell2(X.u, q0, q1, ix0, iy0, iz0, ix1, iy1, iz1);
        cc[k] = locate(ix0, iy0, iz0, ii[k]); cc[k + 1u] = locate(ix1, iy1, iz1, ii[k + 1u]);
      }
#pragma unroll
      for (uint32_t k = 0; k < kSweepChunks; ++k) ww[k] = g.reach[cc[k] == kNone ? 0u : cc[k] >> 5];
#pragma unroll
      for (uint32_t k = 0; k < kSweepChunks; ++k) push(cc[k], ww[k], ii[k]);
      lds_fence();
      // upper bound of what this candidate can still reach: confirmed + waiting + not swept yet (all wave-uniform)
      const uint32_t swept = min(base + kSweepStep, K.n_q);
      if (cnt + qn + (K.n_q - swept) <= K.prune) { abandoned = true; break; }
    }
    // exact stage, ONE code site: 128 entries at a time once more than kQueueHold wait (the queue then still takes a
    // sweep step), the rest after the last step -- unless the waiting entries can no longer lift the candidate above the bound
    while (qn > kQueueHold || (!more && qn != 0u)) {
      if (!more && cnt + qn <= K.prune) { abandoned = true; break; }
      const uint32_t n = min(qn, 128u);
      const bool va = lane < n, vb = lane + 64u < n;
      const uint32_t aa = qn - n + min(lane, n - 1u), ab = qn - n + min(lane + 64u, n - 1u);
      if (!SKIP_FINE) {
        const uint32_t h = exact_pair<COUNT>(g, K, Tsrc, va, uint32_t(q_idx[aa]), q_rank[aa], vb, uint32_t(q_idx[ab]), q_rank[ab]);
        cnt += uint32_t(__popcll(__builtin_amdgcn_ballot_w64((h & 1u) != 0u))) + uint32_t(__popcll(__builtin_amdgcn_ballot_w64((h & 2u) != 0u)));
      }
      qn -= n;
      lds_fence();
    }
    if (!more || abandoned) break;
  }
  if (abandoned && K.prune != 0u && K.pruned != nullptr && lane == 0) atomicAdd(K.pruned, 1u);
  __builtin_amdgcn_wave_barrier();
  return cnt;                                            // wave-uniform
}

// The same count with a sweep that stays inside the CU.  Queue layout: [0, nb) entries that passed the reach test {rank of the
// cell among the reachable ones, query}, [nb, nb + na) L0 survivors of the sweep {cell, query}.
//   sweep step: locate, coarse bitmap, compaction of the L0 survivors -- no global access;
//   bound after every step: confirmed + nb + na + not swept yet <= prune -> abandoned (an L0 survivor is a possible inlier);
//   drain (queue filling up, or sweep over and the candidate still alive): the reach word of every L0 survivor, 64 entries per
//     gather on dense lanes; survivors become reach-tested entries in place (they are written below the read position);
//     then the bound again with the reach-tested entries, then exact batches as in wave_lcp_count.
template <bool COUNT, bool SKIP_FINE, bool QLDS>
__device__ __forceinline__ uint32_t wave_lcp_count_staged(const LcpGrid& g, const LcpTask& K, const uint32_t* s_coarse, const uint2* s_q,
                                                          uint32_t* s_queue, const float4* Tsrc) {
  constexpr uint32_t kNone = 0xFFFFFFFFu;
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t* q_rank = s_queue;                                                     // 32-bit word of an entry: rank, or cell
  uint16_t* q_idx = reinterpret_cast<uint16_t*>(s_queue + kQueueEntries);         // its query index
  uint32_t cnt = 0, nb = 0, na = 0;
  const uint32_t cmax = g.coarse_words * 32u - 1u;
  const uint32_t unx = uint32_t(g.nx), uny = uint32_t(g.ny), unz = uint32_t(g.nz);
  const uint32_t ucx = uint32_t(g.cnx), ucy = uint32_t(g.cny);
  GridXf X;
  { float T[12]; load_rows(Tsrc, T); X = locating_xf<QLDS>(g, K, T); }
  auto locate = [&](const int ix, const int iy, const int iz, const uint32_t i) -> uint32_t {
    const bool inb = (uint32_t(ix) < unx) & (uint32_t(iy) < uny) & (uint32_t(iz) < unz) & (i < K.n_q);
    const uint32_t cc = min(mad24(mad24(uint32_t(iz) >> g.cshift, ucy, uint32_t(iy) >> g.cshift), ucx, uint32_t(ix) >> g.cshift), cmax);
    const uint32_t bit = (s_coarse[cc >> 5] >> (cc & 31u)) & 1u;
    const uint32_t c = mad24(mad24(uint32_t(iz), uny, uint32_t(iy)), unx, uint32_t(ix));
    return (inb & (bit != 0u)) ? c : kNone;
  };
  auto lds_fence = [&]() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };
  auto push_cell = [&](const uint32_t c, const uint32_t i) {
    const bool hit = c != kNone;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
    if (COUNT) { if (lane == 0) atomicAdd(K.point_tests + 1, (unsigned long long)__popcll(m)); }
    if (m == 0ull) return;
    if (hit) {
      const uint32_t at = nb + na + __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));
      q_rank[at] = c;
      q_idx[at] = uint16_t(i);
    }
    na += uint32_t(__popcll(m));
  };
  auto drain = [&]() {                                     // reach test of [nb, nb + na), survivors appended to [0, nb)
    uint32_t rd = nb;
    const uint32_t end = nb + na;
    while (rd < end) {                                     // wave-uniform
      const uint32_t n = min(end - rd, 64u);
      const bool v = lane < n;
      const uint32_t c = q_rank[rd + min(lane, n - 1u)];
      const uint32_t i = uint32_t(q_idx[rd + min(lane, n - 1u)]);
      lds_fence();                                         // every lane holds its entry before any slot of this batch is rewritten
      const uint2 w = g.reach[c >> 5];
      const uint32_t sh = c & 31u;
      const bool reach = v & (((w.x >> sh) & 1u) != 0u);
      const unsigned long long m = __builtin_amdgcn_ballot_w64(reach);
      if (COUNT) { if (lane == 0) atomicAdd(K.point_tests + 2, (unsigned long long)__popcll(m)); }
      if (reach) {                                         // nb <= rd and at most n survivors: the writes stay below rd + n
        const uint32_t at = nb + __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));
        q_rank[at] = w.y + uint32_t(__popc(w.x & ((1u << sh) - 1u)));
        q_idx[at] = uint16_t(i);
      }
      nb += uint32_t(__popcll(m));
      rd += n;
      lds_fence();
    }
    na = 0u;
  };
  const uint32_t last = K.n_q - 1u;
  bool abandoned = false;
  for (uint32_t base = 0;; base += kSweepStep) {
    const bool more = base < K.n_q;                        // wave-uniform
    const uint32_t unswept = K.n_q - min(base + kSweepStep, K.n_q);
    if (more) {
      uint32_t ii[kSweepChunks], cc[kSweepChunks];
#pragma unroll
      for (uint32_t k = 0; k < kSweepChunks; k += 2u) {
        ii[k] = base + lane + 64u * k; ii[k + 1u] = ii[k] + 64u;
        const float4 q0 = sweep_query<QLDS>(K, s_q, QLDS ? ii[k] : min(ii[k], last));
        const float4 q1 = sweep_query<QLDS>(K, s_q, QLDS ? ii[k + 1u] : min(ii[k + 1u], last));
        int ix0, iy0, iz0, ix1, iy1, iz1;
        grid_cell2(X.u, q0, q1, ix0, iy0, iz0, ix1, iy1, iz1);
        cc[k] = locate(ix0, iy0, iz0, ii[k]); cc[k + 1u] = locate(ix1, iy1, iz1, ii[k + 1u]);
      }
#pragma unroll
      for (uint32_t k = 0; k < kSweepChunks; ++k) push_cell(cc[k], ii[k]);
      lds_fence();
      if (cnt + nb + na + unswept <= K.prune) { abandoned = true; break; }
    }
    if (nb + na > kQueueHold || (!more && (nb + na) != 0u)) {
      drain();
      if (cnt + nb + (more ? unswept : 0u) <= K.prune) { abandoned = true; break; }
      while (nb > kExactHold || (!more && nb != 0u)) {
        if (!more && cnt + nb <= K.prune) { abandoned = true; break; }
        const uint32_t n = min(nb, 128u);
        const bool va = lane < n, vb = lane + 64u < n;
        const uint32_t aa = nb - n + min(lane, n - 1u), ab = nb - n + min(lane + 64u, n - 1u);
        if (!SKIP_FINE) {
          const uint32_t h = exact_pair<COUNT>(g, K, Tsrc, va, uint32_t(q_idx[aa]), q_rank[aa], vb, uint32_t(q_idx[ab]), q_rank[ab]);
          cnt += uint32_t(__popcll(__builtin_amdgcn_ballot_w64((h & 1u) != 0u))) + uint32_t(__popcll(__builtin_amdgcn_ballot_w64((h & 2u) != 0u)));
        }
        nb -= n;
        lds_fence();
      }
    }
    if (!more || abandoned) break;
  }
  if (abandoned && K.prune != 0u && K.pruned != nullptr && lane == 0) atomicAdd(K.pruned, 1u);
  __builtin_amdgcn_wave_barrier();
  return cnt;
}
// With the exit off every candidate pays for all stages and the fused sweep is ahead again (86.4 vs 89.7 M candidates/s, round 3):
// picked per candidate -- staged when a bound is in force, fused otherwise.  (Round 5: the builds that fixed one of the two,
// -DS4P_SWEEP_STAGED=0/1, and the MFMA locate of the lean sweep, -DS4P_LEAN_MFMA=1, are gone; profiles/HISTORY.md has their numbers.)
template <bool COUNT, bool SKIP_FINE, bool QLDS>
__device__ __forceinline__ uint32_t wave_lcp_count_auto(const LcpGrid& g, const LcpTask& K, const uint32_t* s_coarse, const uint2* s_q,
                                                        uint32_t* s_queue, const float4* Tsrc) {
  if (K.prune != 0u) return wave_lcp_count_staged<COUNT, SKIP_FINE, QLDS>(g, K, s_coarse, s_q, s_queue, Tsrc);      // wave-uniform
  return wave_lcp_count<COUNT, SKIP_FINE, QLDS>(g, K, s_coarse, s_q, s_queue, Tsrc);
}

// ---------------------------------------------------------------------------
// The LEAN sweep: what k_verify runs when an early-exit bound is in force (LcpTask::prune > 0), i.e. inside the trial loops,
// where all but one candidate in 10^4 are abandoned after a sweep that never needs more than "how many queries COULD still
// be inliers".  Per 64 queries the fused sweep above issues ~50 vector instructions and the staged one ~37; this one ~16:
//   * the sampled Q lives in LDS as three float arrays (no 16-bit unpack: 3 conversions per query gone);
//   * the 3x4 locating transform runs in COARSE units (2^cshift cells: an exact power-of-two scaling of the grid-unit
//     transform) on the packed-FP32 pipe: grid_cell2, two queries per v_pk_fma_f32, nine per pair of queries (the round-4
//     variant on the matrix pipe -- v_mfma_f32_4x4x1, lane = query -- was measured slower and removed in round 5);
//   * only the coarse cube is located (floor, bounds, linear index, one LDS word, one bit): the fine cell and the rank among
//     the reachable cells are needed only for queries whose candidate survives the sweep, so they are computed THERE;
//   * a queue entry is the 16-bit query index alone.
// Everything here only LOCATES: positions may be off by ~1e-5 cell (coarse-unit rounding vs the fine-unit fma chain of grid_cell),
// which the structure absorbs at every level independently (a coarse cube is marked if any of its cells is reachable, a cell
// if a P point lies within delta + 0.01 h of its box: LcpGridHost::plan).  The inlier predicate itself is exact_setup /
// group_hit, untouched: counts stay bit-exact.
// Queue of one wave: kLeanQueue 16-bit entries: [0, nb) passed the reach test, [nb, nb + na) L0 survivors of the sweep.
// Padding queries (index >= n_q) sit at 1e18: a rigid transform sends them outside the grid on at least one axis, so the
// sweep needs no index test.
// ---------------------------------------------------------------------------
constexpr uint32_t kLeanQueue = 768;                       // entries per wave (2 B each)
static_assert(kLeanQueue >= 2u * kSweepStep + 128u && kLeanQueue % 64u == 0u, "lean queue: two sweep steps + one exact batch");
constexpr float kLeanPad = 1.0e18f;                        // coordinates of the padding queries
constexpr int kLeanMaxQueries = 2560;                      // sampled-Q points the float LDS copy takes (30 KB)
typedef float f4_t __attribute__((ext_vector_type(4)));

struct LeanLds { const uint32_t* coarse; const float* qx; const float* qy; const float* qz; uint16_t* queue; };


// LDS word `index` of the array at byte address `base` (wave-uniform): one shift-add for the address (the compiler's own
// form of base + 4 * (x >> 5) is shift, mask, add)
__device__ __forceinline__ uint32_t lds_word(const uint32_t base, const uint32_t index) {
  uint32_t addr;
  asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(addr) : "v"(index), "s"(base));
  return *(const __attribute__((address_space(3))) uint32_t*)(uintptr_t(addr));
}
__device__ __forceinline__ uint32_t bfe1(const uint32_t word, const uint32_t pos) {      // (word >> (pos & 31)) & 1: the hardware masks pos itself
  uint32_t r;
  asm("v_bfe_u32 %0, %1, %2, 1" : "=v"(r) : "v"(word), "v"(pos));
  return r;
}

// exact stage for entries that carry only the query index: fine cell (same function, same inputs as the drain's reach test),
// reach word -> rank, then exact_setup / the point walk as in exact_pair
// query i of the sweep order: from the LDS copy (QL) or from the padded float4 array in global memory (samples that do not
// fit LDS: the 20 000-point sample of SURVEY 8d; same values, so both kernels locate and count identically)
template <bool QL>
__device__ __forceinline__ float4 lean_query(const LcpTask& K, const LeanLds& L, const uint32_t i) {
  if (QL) return make_float4(L.qx[i], L.qy[i], L.qz[i], 0.f);
  return K.q4[i];
}
template <bool COUNT, bool QL>
__device__ __forceinline__ uint32_t exact_pair_lean(const LcpGrid& g, const LcpTask& K, const LeanLds& L, const float4* Tsrc,
                                                    const bool validA, const uint32_t iA, const bool validB, const uint32_t iB) {
  ExactEntry A, B;
  { float T[12]; load_rows(Tsrc, T);
    const GridXf X = make_grid_xf(g, T, 1.f);
    auto one = [&](const bool valid, const uint32_t i) -> ExactEntry {
      const float4 q = lean_query<QL>(K, L, i);
      int ix, iy, iz;
      grid_cell(X.u, q, ix, iy, iz);
      const uint32_t c = mad24(mad24(uint32_t(iz), uint32_t(g.ny), uint32_t(iy)), uint32_t(g.nx), uint32_t(ix));
      const uint2 w = g.reach[valid ? c >> 5 : 0u];                   // (valid entries passed the bounds + reach test with this very c)
      const uint32_t rank = w.y + uint32_t(__popc(w.x & ((1u << (c & 31u)) - 1u)));
      return exact_setup_q<COUNT>(g, K, T, valid, q, rank);
    };
    A = one(validA, iA);
    B = one(validB, iB); }
  uint32_t hits = 0;
  while (A.p < A.e || B.p < B.e) {
    const bool la = A.p < A.e, lb = B.p < B.e;
    const uint32_t ia = la ? (A.p >> 1) * 8u + (A.p & 1u) * 3u : 0u, ib = lb ? (B.p >> 1) * 8u + (B.p & 1u) * 3u : 0u;
    const float4 ax = g.nbr[ia], ay = g.nbr[ia + 1u], az = g.nbr[ia + 2u];
    const float4 bx = g.nbr[ib], by = g.nbr[ib + 1u], bz = g.nbr[ib + 2u];
    const bool ha = la && group_hit(ax, ay, az, A.tx, A.ty, A.tz, g.sq_eps);
    const bool hb = lb && group_hit(bx, by, bz, B.tx, B.ty, B.tz, g.sq_eps);
    if (ha) { hits |= 1u; A.p = A.e; } else if (la) A.p += 1u;
    if (hb) { hits |= 2u; B.p = B.e; } else if (lb) B.p += 1u;
  }
  return hits;
}

template <bool COUNT, bool SKIP_FINE, bool QL>
__device__ __forceinline__ uint32_t wave_lcp_count_lean(const LcpGrid& g, const LcpTask& K, const LeanLds& L, const float4* Tsrc, const float4 t0, const float4 t1, const float4 t2) {
  // t0..t2: the rows at Tsrc, already in registers (k_verify fetches a candidate's record while the previous one is swept); the
  // rare drain / exact batches read them again through Tsrc
  const uint32_t lane = threadIdx.x & 63u;
  uint16_t* q = L.queue;
  uint32_t cnt = 0, nb = 0, na = 0;
  // pitches of the coarse bitmap (g.cnx, g.cny) include one empty border cube per axis (LcpGridHost::plan); mx, my, mz = its index
  const uint32_t ucx = uint32_t(g.cnx), ucy = uint32_t(g.cny);
  const uint32_t mx = ucx - 1u, my = ucy - 1u, mz = uint32_t(((g.nz - 1) >> g.cshift) + 1);
  // locating transform in coarse units (exact power-of-two scaling of the grid-unit transform)
  GridXf Xc;
  { const float T[12] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w, t2.x, t2.y, t2.z, t2.w};
    Xc = make_grid_xf(g, T, coarse_scale(g)); }
  typedef const __attribute__((address_space(3))) uint32_t* lds_u32_ptr;
  const uint32_t coarse_base = uint32_t(uintptr_t((lds_u32_ptr)L.coarse));      // byte address of the bitmap inside LDS (0 in k_verify)
  auto lds_fence = [&]() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };
  // Reach test of the L0 survivors [nb, nb + na), the survivors compacted in place onto [0, nb).  The reach word is the one
  // global access of the lean path, a dependent ~1 us round trip: kSweepChunks batches of 64 entries are located and gathered
  // TOGETHER (one exposure per 256 entries instead of four), and after every round the bound is applied to "survivors so far
  // + entries not tested yet (+ rest: queries not swept yet)" -- a candidate whose L0 survivors exceeded the bound is usually
  // dismissed before all of them have been tested (measured on the bench workload: one candidate in five reaches this point and
  // its serial 64-entry batches were ~40 % of the kernel's wave time).  Returns true if the candidate is dismissed.
  auto drain = [&](const uint32_t rest) -> bool {
    lds_fence();
    float T[12]; load_rows(Tsrc, T);
    const GridXf X = make_grid_xf(g, T, 1.f);
    uint32_t rd = nb;
    const uint32_t end = nb + na;
    bool dead = false;
    while (rd < end) {                                     // wave-uniform
      uint32_t ii[kSweepChunks], cc[kSweepChunks]; bool vv[kSweepChunks]; uint2 ww[kSweepChunks];
#pragma unroll
      for (uint32_t k = 0; k < kSweepChunks; ++k) { const uint32_t at = rd + 64u * k + lane; vv[k] = at < end; ii[k] = uint32_t(q[min(at, end - 1u)]); }
      lds_fence();                                         // every lane holds its entries before any slot of this round is rewritten
#pragma unroll
      for (uint32_t k = 0; k < kSweepChunks; ++k) {
        int ix, iy, iz;
        grid_cell(X.u, lean_query<QL>(K, L, ii[k]), ix, iy, iz);
        vv[k] = vv[k] & (uint32_t(ix) < uint32_t(g.nx)) & (uint32_t(iy) < uint32_t(g.ny)) & (uint32_t(iz) < uint32_t(g.nz));
        cc[k] = mad24(mad24(uint32_t(iz), uint32_t(g.ny), uint32_t(iy)), uint32_t(g.nx), uint32_t(ix));
        ww[k] = g.reach[vv[k] ? cc[k] >> 5 : 0u];
      }
      const uint32_t n_round = min(end - rd, kSweepStep);
#pragma unroll
      for (uint32_t k = 0; k < kSweepChunks; ++k) {
        const bool reach = vv[k] & (((ww[k].x >> (cc[k] & 31u)) & 1u) != 0u);
        const unsigned long long m = __builtin_amdgcn_ballot_w64(reach);
        if (reach) q[__builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), nb))] = uint16_t(ii[k]);   // nb <= rd: below the entries read
        nb += uint32_t(__popcll(m));
      }
      if (COUNT) { if (lane == 0) atomicAdd(K.point_tests + 1, (unsigned long long)n_round); }      // reach words gathered (l0_pass)
      rd += n_round;
      lds_fence();
      if (cnt + nb + (end - rd) + rest <= K.prune) { dead = true; break; }
    }
    na = 0u;
    return dead;
  };
  const uint32_t n_pad = (K.n_q + kSweepStep - 1u) & ~(kSweepStep - 1u);
  bool abandoned = false;
  for (uint32_t base = 0;; base += kSweepStep) {
    const bool more = base < n_pad;                        // wave-uniform
    const uint32_t unswept = K.n_q - min(base + kSweepStep, K.n_q);
    if (more) {
      // one step = kSweepChunks chunks in four phases, so that the LDS reads of all chunks are in flight together and
      // the dependent packed FMAs of one chunk are separated by the other chunks'
      uint32_t ii[kSweepChunks], cc[kSweepChunks], ww[kSweepChunks];
      float x[kSweepChunks], y[kSweepChunks], z[kSweepChunks];
#pragma unroll
      for (uint32_t k = 0; k < kSweepChunks; ++k) { ii[k] = base + 64u * k + lane; const float4 p = lean_query<QL>(K, L, ii[k]); x[k] = p.x; y[k] = p.y; z[k] = p.z; }
      int cx[kSweepChunks], cy[kSweepChunks], cz[kSweepChunks];
#pragma unroll
      for (uint32_t k = 0; k < kSweepChunks; k += 2u)
        grid_cell2(Xc.u, make_float4(x[k], y[k], z[k], 0.f), make_float4(x[k + 1u], y[k + 1u], z[k + 1u], 0.f), cx[k], cy[k], cz[k], cx[k + 1u], cy[k + 1u], cz[k + 1u]);
#pragma unroll
      for (uint32_t k = 0; k < kSweepChunks; ++k) {
        // the bitmap has one more (empty) cube per axis: a coordinate outside the grid on either side clamps onto it
        cc[k] = mad24_s(mad24_s(min(uint32_t(cz[k]), mz), ucy, min(uint32_t(cy[k]), my)), ucx, min(uint32_t(cx[k]), mx));
        ww[k] = lds_word(coarse_base, cc[k] >> 5);
      }
#pragma unroll
      for (uint32_t k = 0; k < kSweepChunks; ++k) {
        const uint32_t t = bfe1(ww[k], cc[k]);
        const unsigned long long m = __builtin_amdgcn_ballot_w64(t != 0u);
        if (m != 0ull) {
          uint16_t* qw = q + (nb + na);                       // (uniform: the lane's slot is one shift-add from its mbcnt)
          if (t != 0u) qw[__builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u))] = uint16_t(ii[k]);
          na += uint32_t(__popcll(m));
        }
      }
      // (instrumentation counts what is FETCHED: L0 survivors when their reach word is gathered -- drain --, reach survivors
      // when their list header is read -- exact batch; an abandoned candidate has touched neither)
      // upper bound of what this candidate can still reach: confirmed + waiting (either kind) + not swept yet
      if (cnt + nb + na + unswept <= K.prune) { abandoned = true; break; }
    }
    if (!more || nb + na + kSweepStep > kLeanQueue) {
      const uint32_t rest = more ? unswept : 0u;
      const bool dead = drain(rest);
      if (dead) { abandoned = true; break; }
      while ((more && nb + 2u * kSweepStep > kLeanQueue) || (!more && nb != 0u)) {
        if (cnt + nb + rest <= K.prune) { abandoned = true; break; }
        const uint32_t n = min(nb, 128u);
        const bool va = lane < n, vb = lane + 64u < n;
        const uint32_t ia = uint32_t(q[nb - n + min(lane, n - 1u)]), ib = uint32_t(q[nb - n + min(lane + 64u, n - 1u)]);
        if (COUNT) { if (lane == 0) atomicAdd(K.point_tests + 2, (unsigned long long)n); }      // list headers read (l1_pass)
        if (!SKIP_FINE) {
          const uint32_t h = exact_pair_lean<COUNT, QL>(g, K, L, Tsrc, va, ia, vb, ib);
          cnt += uint32_t(__popcll(__builtin_amdgcn_ballot_w64((h & 1u) != 0u))) + uint32_t(__popcll(__builtin_amdgcn_ballot_w64((h & 2u) != 0u)));
        }
        nb -= n;
        lds_fence();
      }
    }
    if (!more || abandoned) break;
  }
  if (abandoned && K.pruned != nullptr && lane == 0) atomicAdd(K.pruned, 1u);
  __builtin_amdgcn_wave_barrier();
  return cnt;
}

// Global -> LDS copy of the float queries (SoA, padded on the device side to a multiple of a sweep step with kLeanPad)
__device__ __forceinline__ void stage_queries_f(const float* src, float* dst, const uint32_t n_words) {
  const float4* s4 = reinterpret_cast<const float4*>(src);
  float4* d4 = reinterpret_cast<float4*>(dst);
  for (uint32_t w = threadIdx.x; w < (n_words >> 2); w += blockDim.x) d4[w] = s4[w];      // n_words = 3 * n_pad, a multiple of 4
}

// Global -> LDS copy of the quantised queries (8 B each), padded to a multiple of a sweep step with the last entry
__device__ __forceinline__ void stage_queries(const LcpTask& K, uint2* s_q) {
  const uint32_t n_pad = (K.n_q + kSweepStep - 1u) & ~(kSweepStep - 1u);
  for (uint32_t w = threadIdx.x; w < n_pad; w += blockDim.x) s_q[w] = K.qq.packed[min(w, K.n_q - 1u)];
}

// Global -> LDS copy of the coarse bitmap: 16 B per lane and four independent loads in flight per thread (a
// word-at-a-time loop serialises ~13 L2 round trips per thread and cost ~50 us per launch).
__device__ __forceinline__ void stage_coarse(const LcpGrid& g, uint32_t* s_coarse) {
  const uint32_t n4 = g.coarse_words >> 2;                       // the host pads coarse_words to a multiple of 4
  const uint4* src = reinterpret_cast<const uint4*>(g.coarse);
  uint4* dst = reinterpret_cast<uint4*>(s_coarse);
  for (uint32_t w = threadIdx.x; w < n4; w += 4u * blockDim.x) {
    const uint32_t w1 = w + blockDim.x, w2 = w1 + blockDim.x, w3 = w2 + blockDim.x;
    const uint4 a = src[w];
    const uint4 b = src[min(w1, n4 - 1u)];
    const uint4 c = src[min(w2, n4 - 1u)];
    const uint4 d = src[min(w3, n4 - 1u)];
    dst[w] = a;
    if (w1 < n4) dst[w1] = b;
    if (w2 < n4) dst[w2] = c;
    if (w3 < n4) dst[w3] = d;
  }
  __syncthreads();
}

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

// ---------------------------------------------------------------------------
// Congruent-quad enumeration, preparation side (FindCongruentQuadrilaterals, super4pcs.cc:80-177).
// set 1 entries are chained per euclidean cell in an epoch-tagged hash table
// (no per-base clearing); set 2 entries carry a 343-bit cone mask of direction buckets.
// ---------------------------------------------------------------------------
struct QuadGrid {             // IndexedNormalSet parameters (normalset.h:114-124)
  float gepsilon;             // 1.f / egSize
  float nepsilon;             // 1/7 + 1e-5
  int egSize;
};
struct ConeTable {            // getNeighbors constants (normalset.hpp:174-191), host-computed with libm
  int nb;
  float v[kMaxConeSamples][3];   // (sinA*cos(theta_a), sinA*sin(theta_a), cosA)
};

__device__ __forceinline__ uint32_t index_normal(float x, float y, float z, float neps) {
  const int c0 = int((x / 2.f + 0.5f) / neps);
  const int c1 = int((y / 2.f + 0.5f) / neps);
  const int c2 = int((z / 2.f + 0.5f) / neps);
  return uint32_t(c2 * 49 + c1 * 7 + c0);
}
__device__ __forceinline__ uint32_t index_pos(float x, float y, float z, const QuadGrid& g) {
  const int c0 = int(x / g.gepsilon), c1 = int(y / g.gepsilon), c2 = int(z / g.gepsilon);
  return (uint32_t(c2) * uint32_t(g.egSize) + uint32_t(c1)) * uint32_t(g.egSize) + uint32_t(c0);
}
__device__ __forceinline__ uint32_t hash_cell(uint32_t c) {
  c ^= c >> 16; c *= 0x7feb352du; c ^= c >> 15; c *= 0x846ca68bu; c ^= c >> 16;
  return c;
}

struct HashTable {
  unsigned long long* keys;    // (epoch << 32) | cell
  unsigned long long* heads;   // (epoch << 32) | entry index
  uint32_t mask;               // allocated size - 1 (power of two; sized for the pair capacity: 128 MB + 128 MB at 8 M pairs)
  uint32_t epoch;
  const uint32_t* m1_dev;      // device count of the set-1 pairs of this base (final when k_prep / k_quads run)
  uint32_t cap1;
  uint32_t fixed_mask;         // != 0: the slots this base uses, fixed by the host from the registration's recent bases (hash_mask)
};
// Slots a base really uses: 4 x its set-1 pairs, rounded up to a power of two (>= 4096) -- a few MB that stay in L2 instead
// of random probes over the whole allocation (HBM + TLB misses on every hop).  Builder (k_prep) and reader (k_quads)
// derive the same mask from the same device counter; entries of earlier epochs, wherever they lie, read as empty.
// When the set-1 records are prepared INSIDE k_pairs2 (PairParams::prep_on) the count is not final yet while the table is being
// built: the host then fixes the size from what the registration's recent bases needed (fixed_mask), with head-room, and an
// entry whose index would push the load above one half is refused (overflow bit 8: the host redoes the base with the exact size).
__device__ __forceinline__ uint32_t hash_mask(const HashTable& ht) {
  if (ht.fixed_mask != 0u) return ht.fixed_mask;
  const uint32_t m = min(*ht.m1_dev, ht.cap1);
  if (m > (ht.mask >> 2)) return ht.mask;                 // (also keeps 4 * m inside 32 bits)
  const uint32_t want = max(4u * m, 4096u);
  const uint32_t size = 1u << (32 - __clz(int(want - 1u)));
  return min(size - 1u, ht.mask);
}

// Preparation parameters of set 1 (k_prep).
struct PrepParams {
  const float* ux; const float* uy; const float* uz;
  const float* qx; const float* qy; const float* qz;
  const int2* ab; const uint32_t* m_dev; uint32_t cap;
  float invariant;
  QuadGrid qg;
  uint32_t* cell; uint32_t* bucket; float4* ew; uint32_t* next;
  HashTable ht;
};

// set 1, one pair (entry e = (ab.x, ab.y)): invariant point, cell, direction bucket, world point, hash insert
__device__ __forceinline__ void prep1_item(const PrepParams& P, const uint32_t e, const int2 ab) {
  const float p1x = P.ux[ab.x], p1y = P.uy[ab.x], p1z = P.uz[ab.x];
  const float p2x = P.ux[ab.y], p2y = P.uy[ab.y], p2z = P.uz[ab.y];
  float nx = p2x - p1x, ny = p2y - p1y, nz = p2z - p1z;
  const float posx = p1x + P.invariant * nx, posy = p1y + P.invariant * ny, posz = p1z + P.invariant * nz;  // super4pcs.cc:123
  normalize3(nx, ny, nz);                                                                                  // :121
  const uint32_t cell = index_pos(posx, posy, posz, P.qg);
  P.cell[e] = cell;
  P.bucket[e] = index_normal(nx, ny, nz, P.qg.nepsilon);
  const float w1x = P.qx[ab.x], w1y = P.qy[ab.x], w1z = P.qz[ab.x];
  const float w2x = P.qx[ab.y], w2y = P.qy[ab.y], w2z = P.qz[ab.y];
  P.ew[e] = make_float4(w1x + (w2x - w1x) * P.invariant, w1y + (w2y - w1y) * P.invariant,
                        w1z + (w2z - w1z) * P.invariant, 0.f);                                            // :157
  // insert into the cell hash (find-or-claim slot, then push on the chain)
  const unsigned long long mykey = ((unsigned long long)P.ht.epoch << 32) | cell;
  const uint32_t hmask = hash_mask(P.ht);
  uint32_t h = hash_cell(cell) & hmask;
  while (true) {
    const unsigned long long k = __hip_atomic_load(&P.ht.keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k == mykey) break;
    if (uint32_t(k >> 32) != P.ht.epoch) {
      const unsigned long long old = atomicCAS(&P.ht.keys[h], k, mykey);
      if (old == k || old == mykey) break;
      continue;   // somebody claimed it for another cell: re-read the same slot
    }
    h = (h + 1u) & hmask;
  }
  const unsigned long long prev = atomicExch(&P.ht.heads[h], ((unsigned long long)P.ht.epoch << 32) | e);
  P.next[e] = (uint32_t(prev >> 32) == P.ht.epoch) ? uint32_t(prev) : kNil;
}

// Quaternion::setFromTwoVectors(zhat, n) (Eigen/Geometry) + closed-form replacement of
// its JacobiSVD branch (deviation D1, identical in the oracle).  q = (w, x, y, z).
__device__ __forceinline__ void quat_from_z_to(float nx, float ny, float nz, float q[4]) {
  normalize3(nx, ny, nz);
  float c = 0.f * nx + (0.f * ny + 1.f * nz);
  float ax, ay, az;
  cross3(0.f, 0.f, 1.f, nx, ny, nz, ax, ay, az);
  if (c < -1.f + 1e-5f) {
    c = fmaxf(c, -1.f);
    const float s = sqn3(ax, ay, az);
    if (s > 0.f) { const float r = sqrtf(s); ax /= r; ay /= r; az /= r; }
    else { ax = 1.f; ay = 0.f; az = 0.f; }
    const float w2 = (1.f + c) * 0.5f;
    q[0] = sqrtf(w2);
    const float sv = sqrtf(1.f - w2);
    q[1] = ax * sv; q[2] = ay * sv; q[3] = az * sv;
    return;
  }
  const float s = sqrtf((1.f + c) * 2.f);
  const float invs = 1.f / s;
  q[1] = ax * invs; q[2] = ay * invs; q[3] = az * invs;
  q[0] = s * 0.5f;
}

// The 343-bit cone mask of one set-2 pair (getNeighbors, normalset.hpp:174-196): the direction buckets hit by the nb cone
// samples rotated onto the pair's direction (nx, ny, nz: p2 - p1 in unit coordinates, not normalised), OR-ed into the
// caller's private row of kMaskWords words (zeroed here).
// Only the BUCKET of a rotated, normalised cone sample is needed: int((x / 2 + 0.5) / neps) per axis.  The exact
// sequence -- the quaternion product as Eigen writes it, a square root and six correctly rounded divisions -- is ~150
// instructions per sample.  The fast path rotates with the quaternion's 3x3 matrix (9 fma; equal to the exact product
// to ~1e-7), does not normalise (the rotated vector is unit to rounding) and multiplies by 1/neps: with
// | |d|^2 - 1 | < 1e-4 its bucket coordinates differ from the exact ones by < 2e-4 (measured < 2e-5,
// tests/test_prep_bucket_fast_path.py), so if every coordinate lies further than 4e-4 from an integer the truncations
// agree; otherwise (0.2 % of the samples) the exact sequence runs.
__device__ __forceinline__ void cone_mask_row(const ConeTable& cone, const float nepsilon, float nx, float ny, float nz, uint32_t* row) {
  float q[4];
  normalize3(nx, ny, nz);                             // queryn = (p2-p1).normalized()            super4pcs.cc:144
  quat_from_z_to(nx, ny, nz, q);                      // setFromTwoVectors normalises it again    normalset.hpp:181
  const float inv_neps = 1.0f / nepsilon;
  float R[9];
  { const float w = q[0], x = q[1], y = q[2], z = q[3];
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
    R[0] = 1.f - 2.f * (yy + zz); R[1] = 2.f * (xy - wz); R[2] = 2.f * (xz + wy);
    R[3] = 2.f * (xy + wz); R[4] = 1.f - 2.f * (xx + zz); R[5] = 2.f * (yz - wx);
    R[6] = 2.f * (xz - wy); R[7] = 2.f * (yz + wx); R[8] = 1.f - 2.f * (xx + yy); }
#pragma unroll
  for (int w = 0; w < kMaskWords; ++w) row[w] = 0u;
  for (int a = 0; a < cone.nb; ++a) {
    const float vx = cone.v[a][0], vy = cone.v[a][1], vz = cone.v[a][2];
    const float fx = __builtin_fmaf(R[0], vx, __builtin_fmaf(R[1], vy, R[2] * vz)), fy = __builtin_fmaf(R[3], vx, __builtin_fmaf(R[4], vy, R[5] * vz)),
                fz = __builtin_fmaf(R[6], vx, __builtin_fmaf(R[7], vy, R[8] * vz));
    const float t0 = __builtin_fmaf(fx, 0.5f, 0.5f) * inv_neps, t1 = __builtin_fmaf(fy, 0.5f, 0.5f) * inv_neps,
                t2 = __builtin_fmaf(fz, 0.5f, 0.5f) * inv_neps;
    const float f0 = __builtin_amdgcn_fractf(t0), f1 = __builtin_amdgcn_fractf(t1), f2 = __builtin_amdgcn_fractf(t2);
    const float edge = fminf(fminf(fminf(f0, 1.f - f0), fminf(f1, 1.f - f1)), fminf(f2, 1.f - f2));
    const float n2 = __builtin_fmaf(fx, fx, __builtin_fmaf(fy, fy, fz * fz));
    uint32_t id;
    if (fabsf(n2 - 1.f) < 1e-4f && edge > 4e-4f) {
      id = uint32_t(int(t2) * 49 + int(t1) * 7 + int(t0));
    } else {
      float ux_, uy_, uz_;
      cross3(q[1], q[2], q[3], vx, vy, vz, ux_, uy_, uz_);            // QuaternionBase::_transformVector
      ux_ += ux_; uy_ += uy_; uz_ += uz_;
      float cx, cy, cz;
      cross3(q[1], q[2], q[3], ux_, uy_, uz_, cx, cy, cz);
      float dx = (vx + q[0] * ux_) + cx, dy = (vy + q[0] * uy_) + cy, dz = (vz + q[0] * uz_) + cz;
      normalize3(dx, dy, dz);
      id = index_normal(dx, dy, dz, nepsilon);
    }
    if (id < 343u) row[id >> 5] |= (1u << (id & 31u));
  }
}

// Preparation of set 1 (one thread per pair): invariant point, cell, direction bucket, world point, hash insert.  Set 2 is
// prepared where it is consumed (k_quads): only the pairs whose cell holds a set-1 pair need their world point and cone mask.
struct PrepGroup { PrepParams base[kGroupMax]; };
static_assert(sizeof(PrepGroup) <= 4096, "PrepGroup travels by value in the 4 KB kernel-argument segment");
__global__ __launch_bounds__(256) void k_prep(PrepGroup PG) {
  const PrepParams& P = PG.base[blockIdx.y];
  const uint32_t m = min(*P.m_dev, P.cap);
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < m; e += gridDim.x * blockDim.x) prep1_item(P, e, P.ab[e]);
}

// ---------------------------------------------------------------------------
// k_pairs: loop 2 of IntersectionFunctor::process (intersectionFunctor.h:197-233) + PairCreationFunctor::process
// (pairCreationFunctor.h:151-218), and -- on the fused path -- the per-pair preparation of FindCongruentQuadrilaterals.
//
// The host hands over the leaves of its octree (loop 1) as a flat sequence of point ids plus, per leaf, its box and
// its slot range.  As the reference does, a primitive first tests the leaf BOX (:205) and only the points of leaves
// its sphere touches are examined (:208-220), so the work is (primitives x touched leaves x their points), not n_Q^2.
// One wave64 per primitive pId.  Leaves are taken 64 at a time (lane = leaf): box test, then the (leaf, point) slots of
// the touched leaves are flattened with a prefix over the leaf sizes -- a binary search for the owning leaf over that prefix -- so every
// round of 64 lanes tests 64 real points.  Accepted (i = pId, j) are compacted by ballot/prefix into the wave's private
// LDS stage (no LDS atomics) and appended with ONE global atomic per workgroup at the end (per wave if its stage
// fills up first): (j,i) then (i,j) with order keys 2*(pId*n_seq + slot) + {0,1}, monotone in the reference's
// emission order.
// ---------------------------------------------------------------------------
struct PairParams {
  const float* ux; const float* uy; const float* uz;     // unit-cube coordinates of sampled Q
  const float* qx; const float* qy; const float* qz;     // world (centred) coordinates
  const float* nx; const float* ny; const float* nz;     // normals or nullptr
  const float* cr; const float* cg; const float* cb;     // rgb or nullptr
  const uint32_t* seq_id; uint32_t n_seq;                  // point ids, leaf-major
  const uint32_t* leaf_off; const float4* leaves; uint32_t n_leaf;   // slot range and (cx, cy, cz, halfEdge argument) per leaf
  uint32_t n_q;
  float nRadius, eps_unit;
  double pair_distance, pair_distance_eps, pair_normals_angle;
  float max_normal_difference, max_color_distance, max_translation_distance, norm_threshold;
  float b1pos[3], b2pos[3], b1rgb[3], b2rgb[3];
  int2* ab; uint32_t* okey; uint32_t* counter; uint32_t cap; uint32_t* overflow; uint32_t overflow_bit;
  uint32_t split;                                          // waves per (tile, chunk): wave `part` takes the slots [part, part + 1) * 64 / split of the chunk (1, 2 or 4)
  // max_angle > 0 (pairCreationFunctor.h:203-212): (j,i) is emitted iff acosf(segment1 . segment2) <= max_angle * pi / 180,
  // (i,j) iff the same holds for -segment2.  acosf is decreasing, so the test is d >= cos_min with cos_min = the smallest
  // float whose libm acosf passes (found by the host with libm itself, s4p_capi.hip angle_threshold); |d| > 1 gives NaN in
  // the reference, i.e. no pair.
  float seg1[3]; float cos_min;
  // Fused pass, first pair set: the FindCongruentQuadrilaterals preparation of every appended pair (invariant point, cell,
  // direction bucket, world point, insert into the cell hash: what k_prep does in a launch of its own) runs where the pair is
  // appended -- one launch and one dependent pass over the pair list less per base (round 5).
  int prep_on; PrepParams prep; uint32_t* prep_overflow;
};

__device__ __forceinline__ bool sphere_box(float cx, float cy, float cz, float r, float4 leaf) {
  const float h = leaf.w;
  float dmin[3], dmax[3];
  const float c[3] = {cx, cy, cz};
  const float nc[3] = {leaf.x, leaf.y, leaf.z};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float mn = nc[k] - h, mx = nc[k] + h;
    const float sqmin = (c[k] - mn) * (c[k] - mn);
    const float sqmax = (c[k] - mx) * (c[k] - mx);
    dmin[k] = (c[k] < mn) ? sqmin : ((c[k] > mx) ? sqmax : 0.f);
    dmax[k] = (sqmin < sqmax) ? sqmax : sqmin;
  }
  const float r2 = r * r;
  return (dmin[0] + (dmin[1] + dmin[2])) < r2 && r2 < (dmax[0] + (dmax[1] + dmax[2]));
}

// the same test with the squared radius precomputed (identical arithmetic: r2 = r * r is what sphere_box forms itself)
__device__ __forceinline__ bool sphere_box_r2(float cx, float cy, float cz, float r2, float4 leaf) {
  const float h = leaf.w;
  float dmin[3], dmax[3];
  const float c[3] = {cx, cy, cz};
  const float nc[3] = {leaf.x, leaf.y, leaf.z};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float mn = nc[k] - h, mx = nc[k] + h;
    const float sqmin = (c[k] - mn) * (c[k] - mn);
    const float sqmax = (c[k] - mx) * (c[k] - mx);
    dmin[k] = (c[k] < mn) ? sqmin : ((c[k] > mx) ? sqmax : 0.f);
    dmax[k] = (sqmin < sqmax) ? sqmax : sqmin;
  }
  return (dmin[0] + (dmin[1] + dmin[2])) < r2 && r2 < (dmax[0] + (dmax[1] + dmax[2]));
}

// PairCreationFunctor::process(i = pId, j) filters (pairCreationFunctor.h:151-218): p = Q[j], q = Q[i]
__device__ __forceinline__ bool pair_filters_w(const PairParams& P, const uint32_t pId, const uint32_t j,
                                               const float wxi, const float wyi, const float wzi, const float wxj, const float wyj, const float wzj);
__device__ __forceinline__ bool pair_filters(const PairParams& P, const uint32_t pId, const uint32_t j,
                                             const float wxi, const float wyi, const float wzi) {
  return pair_filters_w(P, pId, j, wxi, wyi, wzi, P.qx[j], P.qy[j], P.qz[j]);
}
// (the world point of j handed in: k_pairs2 holds it in registers)
__device__ __forceinline__ bool pair_filters_w(const PairParams& P, const uint32_t pId, const uint32_t j,
                                               const float wxi, const float wyi, const float wzi, const float wxj, const float wyj, const float wzj) {
  const float wx = wxi - wxj, wy = wyi - wyj, wz = wzi - wzj;
  const float distance = sqrtf(sqn3(wx, wy, wz));
  bool acc = !(fabs(double(distance) - P.pair_distance) > P.pair_distance_eps);   // :162
  if (acc && P.max_normal_difference > 0.f && P.nx != nullptr) {              // :166-180
    const float qn0 = P.nx[pId], qn1 = P.ny[pId], qn2 = P.nz[pId];
    const float pn0 = P.nx[j], pn1 = P.ny[j], pn2 = P.nz[j];
    if (sqn3(qn0, qn1, qn2) > 0.f && sqn3(pn0, pn1, pn2) > 0.f) {
      const double a1 = double(sqrtf(sqn3(qn0 - pn0, qn1 - pn1, qn2 - pn2)));
      const double a2 = double(sqrtf(sqn3(qn0 + pn0, qn1 + pn1, qn2 + pn2)));
      const float fnd = float(fmin(fabs(a1 - P.pair_normals_angle), fabs(a2 - P.pair_normals_angle)));
      if (fnd > P.norm_threshold) acc = false;
    }
  }
  if (acc && P.max_color_distance > 0.f) {                                    // :182-192
    float pr0 = -1.f, pr1 = -1.f, pr2 = -1.f, qr0 = -1.f, qr1 = -1.f, qr2 = -1.f;
    if (P.cr != nullptr) { pr0 = P.cr[j]; pr1 = P.cg[j]; pr2 = P.cb[j]; qr0 = P.cr[pId]; qr1 = P.cg[pId]; qr2 = P.cb[pId]; }
    const bool use_rgb = (pr0 >= 0.f && qr0 >= 0.f && P.b1rgb[0] >= 0.f && P.b2rgb[0] >= 0.f);
    const bool good = sqrtf(sqn3(pr0 - P.b1rgb[0], pr1 - P.b1rgb[1], pr2 - P.b1rgb[2])) < P.max_color_distance &&
                      sqrtf(sqn3(qr0 - P.b2rgb[0], qr1 - P.b2rgb[1], qr2 - P.b2rgb[2])) < P.max_color_distance;
    if (use_rgb && !good) acc = false;
  }
  if (acc && P.max_translation_distance > 0.f) {                              // :194-200
    const bool good =
        sqrtf(sqn3(wxj - P.b1pos[0], wyj - P.b1pos[1], wzj - P.b1pos[2])) < P.max_translation_distance &&
        sqrtf(sqn3(wxi - P.b2pos[0], wyi - P.b2pos[1], wzi - P.b2pos[2])) < P.max_translation_distance;
    if (!good) acc = false;
  }
  return acc;
}

struct PairSet { PairParams pair; };
struct PairParams2 { PairSet set[2]; };

// BASE GROUPS (round 5).  Every kernel of a base's device pass takes the parameter records of up to kGroupMax bases and one
// launch covers them all: blockIdx.y picks the base (k_pairs2: base and pair set), k_verify walks the candidate lists of all
// of them with ONE staging of its LDS tables.  The kernels of one base are a chain of short, latency-bound launches (21 + 14 +
// 55 + 62 us alone, most of it fixed cost: k_pairs2 takes 29 us whatever the base, k_verify 30 us + 1.9 ns per candidate,
// profiles/r05_verify_vs_candidates.json); a launch that covers three bases pays those fixed costs once.  The records travel
// by value in the kernel argument segment (3 x QuadParams = 3.2 KB of the 4 KB it holds: kGroupMax = 3).
struct PairGroup { PairParams2 base[kGroupMax]; };
static_assert(sizeof(PairGroup) <= 4096, "PairGroup travels by value in the 4 KB kernel-argument segment");

// ---------------------------------------------------------------------------
// k_pairs2: loop 2 TRANSPOSED (round 4).  Round 3's k_pairs walked (primitive -> touched leaves -> their points) with one
// wave per primitive: every round of 64 point slots pays a 6-step cross-lane binary search for the owning leaf and two
// dependent gathers (slot -> id -> coordinates), ~1.5 us of latency per round.  Here a wave owns a TILE of 64 primitives
// (lane = primitive, its centre in registers) times a CHUNK of 64 consecutive point slots of the leaf-major sequence: the
// chunk's ids, points and leaf records are gathered ONCE (lane = slot, all loads independent), and the wave then runs over
// the chunk's leaves and points with UNIFORM control flow and no memory access:
//   per leaf run: box test per lane (intersect, intersectionPrimitive.h:117-142; intersectionFunctor.h:205);
//   per point: its record broadcast by v_readlane, squared distance to the 64 centres and a conservative PRE-test
//     (r - E)^2 <= s2 <= (r + E)^2 of the point test -- no square root; the few lanes that pass (a few % of the tests) are
//     queued as (lane, slot) in LDS;
//   per 64 queued candidates, on dense lanes: the exact point test (intersectPoint, :154-157: correctly rounded sqrt) and
//     the world-space filters of PairCreationFunctor::process (pairCreationFunctor.h:151-218), operands fetched across lanes
//     with ds_bpermute; accepted pairs go to the wave's LDS stage and are appended as in k_pairs.
// Every (primitive, point) test of the reference's loop is decided exactly once; the pre-test only removes tests whose
// outcome is certain (margin E - eps covers the rounding of the exact expression with three orders of magnitude to spare).
// The leaf of a slot travels in the upper half of its sequence word (PairOctree::flatten: id | leaf << 16; both < 2^16
// because n_Q <= 46 340).  Order keys as in k_pairs: 2 * (pId * n_seq + slot) + {0, 1}.
// ---------------------------------------------------------------------------
constexpr int kPair2Waves = 8;      // waves per workgroup: few workgroups = few appends on the one pair counter
constexpr int kPair2StageW = 256;   // staged accepted (primitive, slot) per wave between two flushes
constexpr int kPair2Queue = 128;    // queued (lane, slot-in-chunk) candidates per wave: a batch of 64 runs when 64 wait

template <bool ANGLE>
__global__ __launch_bounds__(64 * kPair2Waves) void k_pairs2(PairGroup PG) {
  const PairParams& P = PG.base[blockIdx.y >> 1].set[blockIdx.y & 1u].pair;      // blockIdx.y = 2 * base + pair set
  __shared__ uint32_t st_e[kPair2Waves][kPair2StageW];   // primitive | slot << 16
  __shared__ uint8_t st_f[kPair2Waves][ANGLE ? kPair2StageW : 4];   // ANGLE: 1 = the second of the two ordered pairs
  __shared__ uint16_t s_qc[kPair2Waves][kPair2Queue];    // candidate queue: lane | k << 6
  __shared__ uint32_t s_cnt[kPair2Waves], s_base;
  const uint32_t lane = threadIdx.x & 63u, wave = uint32_t(__builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)));
  uint32_t n_st = 0;                                     // staged entries of this wave (wave-uniform)
  PROF_DECL;
  PROF_STAMP(0);
  auto wave_fence = [&]() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };
  auto write_out = [&](const uint32_t base) {            // entries -> ordered pairs at positions base, base + 1, ...
    for (uint32_t pe = lane; pe < (ANGLE ? n_st : 2u * n_st); pe += 64u) {
      const uint32_t e = ANGLE ? pe : pe >> 1, second = ANGLE ? uint32_t(st_f[wave][e]) : pe & 1u;
      const uint32_t at = base + pe;
      if ((ANGLE ? at : (at | 1u)) < P.cap) {              // both pairs of an entry fit, or neither is written
        const uint32_t w = st_e[wave][e], pId = w & 0xFFFFu, sl = w >> 16;
        const uint32_t j = P.seq_id[sl] & 0xFFFFu;
        // pairs->emplace_back(j, i) then pairs->emplace_back(i, j)   pairCreationFunctor.h:214-215
        const int2 ab = second ? make_int2(int(pId), int(j)) : make_int2(int(j), int(pId));
        P.ab[at] = ab;
        P.okey[at] = 2u * (pId * P.n_seq + sl) + second;
        if (P.prep_on) {                                     // (uniform) first pair set of a fused pass: its preparation, here
          if (at <= (P.prep.ht.fixed_mask >> 1)) prep1_item(P.prep, at, ab);
          else atomicOr(P.prep_overflow, 8u);                // more pairs than the table was sized for: the host redoes the base
        }
      } else {
        atomicOr(P.overflow, P.overflow_bit);
      }
    }
    wave_fence();
  };
  const uint32_t n_tiles = (P.n_q + 63u) >> 6, n_chunks = (P.n_seq + 63u) >> 6;
  const uint32_t gw = blockIdx.x * kPair2Waves + wave, nw = gridDim.x * kPair2Waves;
  const float r2 = P.nRadius * P.nRadius, e2 = P.eps_unit * P.eps_unit;
  // pre-test bounds on the squared distance: |sqrt(s2) - r| < eps can only hold inside [(r - E)^2, (r + E)^2], E = eps
  // widened by 1e-4 relative + 1e-6 absolute (the exact expression rounds three times at 6e-8 relative each)
  const float E = P.eps_unit * 1.0001f + 1e-6f * (P.nRadius + 1.f);
  const float lo_r = fmaxf(P.nRadius - E, 0.f), hi_r = P.nRadius + E;
  const float lo2 = lo_r * lo_r * 0.9999f, hi2 = hi_r * hi_r * 1.0001f;
  // item = (chunk, tile), tile fastest: the waves of a workgroup share their chunk's gathers in cache
  // Items of very different weight (a tile and a chunk that lie at the base's distance from each other hold most of the
  // candidates: 40 us against a median of 9, profiles/r06_wave_profile_before.log) set the launch's duration, so an item is
  // shared by `split` waves, each taking a contiguous part of the chunk's slots (all of them gather the whole chunk: cheap).
  const uint32_t split = P.split, part_slots = 64u / split;
  for (uint32_t item = gw; item < n_tiles * n_chunks * split; item += nw) {
    const uint32_t part = item % split, ct = item / split;
    const uint32_t chunk = ct / n_tiles, tile = ct - chunk * n_tiles;
    const uint32_t s0 = chunk * 64u, n_in = min(64u, P.n_seq - s0);
    const uint32_t k_lo = part * part_slots, k_hi = min(k_lo + part_slots, n_in);
    if (k_lo >= n_in) continue;                                // (uniform) the chunk's tail part is empty
    // the chunk: lane = slot -- id, unit point, world point, and the record of the slot's leaf (box, end of its slot range)
    const uint32_t sw = P.seq_id[s0 + min(lane, n_in - 1u)];
    const uint32_t jl = sw & 0xFFFFu, leaf_l = sw >> 16;
    const float pux = P.ux[jl], puy = P.uy[jl], puz = P.uz[jl];
    const float pwx = P.qx[jl], pwy = P.qy[jl], pwz = P.qz[jl];
    const float4 box_l = P.leaves[leaf_l];
    const uint32_t end_l = P.leaf_off[leaf_l + 1u];
    PROF_STAMP(1);
    // the tile: lane = primitive
    const uint32_t pId = tile * 64u + lane;
    const bool pvalid = pId < P.n_q;
    const uint32_t pi = min(pId, P.n_q - 1u);
    const float cx = P.ux[pi], cy = P.uy[pi], cz = P.uz[pi];
    const float wxi = P.qx[pi], wyi = P.qy[pi], wzi = P.qz[pi];
    auto bcast = [&](const float v, const uint32_t k) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), int(k))); };
    auto fetch = [&](const float v, const uint32_t from) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(int(from << 2), __builtin_bit_cast(int, v))); };
    uint32_t nq = 0;                                        // queued candidates (wave-uniform)
    // exact tests of up to 64 queued candidates, one per lane
    auto run_batch = [&]() {
      wave_fence();
      const uint32_t n = min(nq, 64u);
      const bool v = lane < n;
      const uint32_t ent = uint32_t(s_qc[wave][nq - n + min(lane, n - 1u)]);
      const uint32_t L = ent & 63u, k = ent >> 6;
      nq -= n;
      // the candidate's primitive (lane L of the tile) and point (slot k of the chunk)
      const float ccx = fetch(cx, L), ccy = fetch(cy, L), ccz = fetch(cz, L);
      const float cwx = fetch(wxi, L), cwy = fetch(wyi, L), cwz = fetch(wzi, L);
      const uint32_t j = uint32_t(__builtin_amdgcn_ds_bpermute(int(k << 2), int(jl)));
      const float qx_ = fetch(pux, k), qy_ = fetch(puy, k), qz_ = fetch(puz, k);
      const float wxj = fetch(pwx, k), wyj = fetch(pwy, k), wzj = fetch(pwz, k);
      const uint32_t cp = tile * 64u + L;
      bool acc = false;
      if (v) {
        const float dx = qx_ - ccx, dy = qy_ - ccy, dz = qz_ - ccz;
        const float d = sqrtf(sqn3(dx, dy, dz)) - P.nRadius;
        if (d * d < e2) acc = pair_filters_w(P, cp, j, cwx, cwy, cwz, wxj, wyj, wzj);      // intersectPoint intersectionPrimitive.h:154-157
      }
      bool emit_a = acc, emit_b = false;                    // ANGLE: (j,i) / (i,j) separately
      if (ANGLE) {
        emit_a = false;
        if (acc) {                                          // pairCreationFunctor.h:203-212
          float sx = cwx - wxj, sy = cwy - wyj, sz = cwz - wzj;
          normalize3(sx, sy, sz);
          const float dd = dot3(P.seg1[0], P.seg1[1], P.seg1[2], sx, sy, sz), nd = -dd;
          emit_a = dd >= P.cos_min && dd <= 1.f;
          emit_b = nd >= P.cos_min && nd <= 1.f;
        }
      }
      const unsigned long long m = __builtin_amdgcn_ballot_w64(emit_a);
      const unsigned long long mb = ANGLE ? __builtin_amdgcn_ballot_w64(emit_b) : 0ull;
      if ((m | mb) == 0ull) return;
      const uint32_t word = cp | ((s0 + k) << 16);
      if (emit_a) {
        const uint32_t e = n_st + __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));
        st_e[wave][e] = word; if (ANGLE) st_f[wave][e] = 0;
      }
      n_st += uint32_t(__popcll(m));
      if (ANGLE) {
        if (emit_b) {
          const uint32_t e = n_st + __builtin_amdgcn_mbcnt_hi(uint32_t(mb >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mb), 0u));
          st_e[wave][e] = word; st_f[wave][e] = 1;
        }
        n_st += uint32_t(__popcll(mb));
      }
      if (n_st + (ANGLE ? 128u : 64u) > uint32_t(kPair2StageW)) {   // stage full before the end: this wave appends on its own
        wave_fence();
        uint32_t b = 0;
        if (lane == 0) b = atomicAdd(P.counter, ANGLE ? n_st : 2u * n_st);
        write_out(uint32_t(__builtin_amdgcn_readfirstlane(int(b))));
        n_st = 0;
      }
    };
    for (uint32_t k = k_lo; k < k_hi;) {                    // uniform: one run of slots = the part of one leaf inside this wave's part of the chunk
      const float4 box = make_float4(bcast(box_l.x, k), bcast(box_l.y, k), bcast(box_l.z, k), bcast(box_l.w, k));
      const uint32_t k1 = min(uint32_t(__builtin_amdgcn_readlane(int(end_l), int(k))) - s0, k_hi);
      const bool touch = pvalid && sphere_box_r2(cx, cy, cz, r2, box);     // intersect, intersectionPrimitive.h:117-142
      if (__builtin_amdgcn_ballot_w64(touch) == 0ull) { k = k1; continue; }
      for (; k < k1; ++k) {                                 // uniform: one point of the leaf against the 64 primitives
        const uint32_t j = uint32_t(__builtin_amdgcn_readlane(int(jl), int(k)));
        const float dx = bcast(pux, k) - cx, dy = bcast(puy, k) - cy, dz = bcast(puz, k) - cz;
        const float s2 = sqn3(dx, dy, dz);
        const bool pre = touch & (pId > j) & (s2 >= lo2) & (s2 <= hi2);      // intersectionFunctor.h:210 + the certain part of :211
        const unsigned long long m = __builtin_amdgcn_ballot_w64(pre);
        if (m == 0ull) continue;
        if (pre) s_qc[wave][nq + __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u))] = uint16_t(lane | (k << 6));
        nq += uint32_t(__popcll(m));
        if (nq >= 64u) run_batch();
      }
    }
    while (nq != 0u) run_batch();                           // (the queue refers to this item's registers: empty it before the next)
  }
  wave_fence();
  PROF_STAMP(2);
  // end of the workgroup's items: ONE global atomic for the waves' leftovers
  if (lane == 0) s_cnt[wave] = n_st;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot = 0;
    for (int w = 0; w < kPair2Waves; ++w) tot += s_cnt[w];
    s_base = tot ? atomicAdd(P.counter, ANGLE ? tot : 2u * tot) : 0u;
  }
  __syncthreads();
  PROF_STAMP(3);
  uint32_t before = 0;
  for (uint32_t w = 0; w < wave; ++w) before += s_cnt[w];
  if (n_st) write_out(s_base + (ANGLE ? before : 2u * before));
  PROF_STAMP(4);
#if defined(S4P_PROF)
  tp_[5] = n_st;
#endif
  PROF_WRITE(0, (blockIdx.y * gridDim.x + blockIdx.x) * kPair2Waves + wave);
}

// ---------------------------------------------------------------------------
// ComputeRigidTransformation + rms gate of one congruent quad (match4pcsBase.cc:365-500, match4pcsBase.hpp:436-439)
// and the compaction of the passing candidates.  Shared by k_gate (stage-level entry point) and k_quads (fused path).
// ---------------------------------------------------------------------------
struct GateParams {
  const float4* q4;                                     // sampled Q (centred), packed (x,y,z,0), original order (quad indices)
  BaseFrame base;
  uint32_t* counts;                                     // per quad: kGateFailed, later the inlier count
  uint32_t* cand_idx; float4* cand_T;                   // gated candidates: quad index + 64-byte record {3x4 transform | tag, quad index}
  uint32_t* C_dev;
};
constexpr uint32_t kCandStride = 4;                     // float4 per candidate record: one 64-byte line, everything k_verify needs of a candidate
// 0 = rejected, 1 = candidate, 2 = candidate whose Euler-angle bound the host settles (rigid_verdict)
template <bool ANGLE>
__device__ __forceinline__ int gate_quad(const GateParams& G, const int4 qd, float T[12]) {
  const float4 a = G.q4[qd.x], b = G.q4[qd.y], c = G.q4[qd.z];
  const float q[3][3] = {{a.x, a.y, a.z}, {b.x, b.y, b.z}, {c.x, c.y, c.z}};
  float c2[3];
  return rigid_verdict<ANGLE>(G.base, q, T, c2);
}
// k: index of the quad, with kBorderFlag set if its gate is undecided
__device__ __forceinline__ void store_candidate(const GateParams& G, const uint32_t at, const uint32_t k, const float T[12], const unsigned long long tag) {
  G.cand_idx[at] = k;
  float4* dst = G.cand_T + kCandStride * size_t(at);
  dst[0] = make_float4(T[0], T[1], T[2], T[3]);
  dst[1] = make_float4(T[4], T[5], T[6], T[7]);
  dst[2] = make_float4(T[8], T[9], T[10], T[11]);
  dst[3] = make_float4(__uint_as_float(uint32_t(tag)), __uint_as_float(uint32_t(tag >> 32)), __uint_as_float(k), 0.f);
}

// k_gate: one thread per congruent quad (s4p_try_congruent_set, where the quads come from the caller).  Passing
// candidates are compacted (wave-aggregated append) into cand_idx / cand_T so that the scoring kernel sees a dense,
// perfectly balanceable list; failing ones get counts[k] = kGateFailed.
struct GateKernelParams { GateParams g; const int4* quads; const unsigned long long* tags; const unsigned long long* K_dev; uint32_t K_cap; };
template <bool ANGLE>
__global__ __launch_bounds__(256) void k_gate(GateKernelParams P) {
  const uint32_t K = uint32_t(min(*P.K_dev, (unsigned long long)P.K_cap));
  const uint32_t lane = threadIdx.x & 63u;
  for (uint32_t k0 = (blockIdx.x * blockDim.x + threadIdx.x) & ~63u; k0 < K; k0 += gridDim.x * blockDim.x) {
    const uint32_t k = k0 + lane;
    float T[12];
    int vd = 0;
    if (k < K) {
      vd = gate_quad<ANGLE>(P.g, P.quads[k], T);
      if (!vd) P.g.counts[k] = kGateFailed;
    }
    const bool ok = vd != 0;
    const unsigned long long pass = __ballot(ok);
    if (pass == 0ull) continue;
    const uint32_t leader = __ffsll((long long)pass) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(P.g.C_dev, uint32_t(__popcll(pass)));
    base = __shfl(base, leader);
    if (ok) store_candidate(P.g, base + uint32_t(__popcll(pass & ((1ull << lane) - 1ull))), k | (vd == 2 ? kBorderFlag : 0u), T, P.tags[k]);
  }
}

struct QuadParams {
  // set 1
  const int2* ab1; const uint32_t* okey1; const uint32_t* bucket1; const float4* ew1; const uint32_t* next1;
  // set 2: prepared here, and only where needed (cell with a set-1 pair): invariant point + cell, world point, cone mask
  const int2* ab2; const uint32_t* okey2;
  const float* ux; const float* uy; const float* uz; const float* qx; const float* qy; const float* qz;
  float invariant2; QuadGrid qg; ConeTable cone;
  const uint32_t* m2_dev; uint32_t cap2;
  HashTable ht;
  float thr;                   // distance_threshold2 (compared against a SQUARED norm: quirk super4pcs.cc:160)
  int4* quads; unsigned long long* tags; unsigned long long* K_dev; uint32_t K_cap; uint32_t* overflow;
  uint32_t r0, r1;                                       // set-2 entries [r0, min(r1, m2)): the whole set, or one chunk of a base whose quads do not fit
  uint32_t slice_num, slice_den;                         // slice_den != 0: only the pairs whose order key = slice_num mod slice_den (one GPU's share of a base)
  uint32_t k1_lo, k1_hi;                                 // only set-1 pairs with order key in [k1_lo, k1_hi): a chunk of a base in REFERENCE order (whole base: 0, 2^32 - 1 with k1_all)
  int k1_all;                                            // 1: no filter on the set-1 order key (the default; saves its gather per hop)
  unsigned long long* qsum_dev; unsigned long long* csum_dev;   // checksums (DevCounters::quad_sum / cand_sum)
  int do_gate; GateParams gate;                          // fused path: gate every quad as it is appended
};

// Threads per workgroup of k_quads (one set-2 pair per thread and tile).  Every flush of a workgroup is two dependent atomics on
// the base's quad and candidate counters, and same-address atomics are served one after the other (~17 ns each): with 256
// threads a base's ~900 workgroups spent 10-24 us of their ~45 in the flush (profiles/r06_wave_profile_before.log).
#ifndef S4P_QUAD_THREADS
#define S4P_QUAD_THREADS 256
#endif
constexpr int kQuadThreads = S4P_QUAD_THREADS, kQuadWaves = kQuadThreads / 64;
constexpr int kQuadStage = 2 * kQuadThreads;      // quads per workgroup between two flushes (24 B each)

// One thread per pairs2 entry: hash lookup of its euclidean cell, walk of the set-1 chain (super4pcs.cc:151-163).
// Matches are staged in LDS and flushed with one global atomic per workgroup round; on the fused path the flush also
// runs ComputeRigidTransformation + the rms gate on the staged quads -- one thread per quad, all 256 lanes busy,
// instead of a separate launch that re-reads them -- and appends the survivors to the candidate list (one more
// atomic per 256 quads).
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t lo = uint32_t(__shfl_xor(int(uint32_t(v)), o)), hi = uint32_t(__shfl_xor(int(uint32_t(v >> 32)), o));
    v += (static_cast<unsigned long long>(hi) << 32) | lo;
  }
  return v;
}

struct QuadGroup { QuadParams base[kGroupMax]; };
static_assert(sizeof(QuadGroup) <= 4096, "QuadGroup travels by value in the 4 KB kernel-argument segment");
template <bool ANGLE>
__global__ __launch_bounds__(kQuadThreads) void k_quads(QuadGroup QG) {
  const QuadParams& P = QG.base[blockIdx.y];
  __shared__ int4 st_q[kQuadStage];
  __shared__ unsigned long long st_t[kQuadStage];
  __shared__ unsigned long long st_base, s_qsum, s_csum;
  __shared__ uint32_t st_n, s_wc[kQuadWaves], s_cbase, s_ic[kQuadWaves];
  __shared__ uint32_t s_item_i[kQuadThreads], s_item_e[kQuadThreads];        // the tile's pairs whose cell holds a set-1 pair, compacted
  __shared__ uint32_t s_mask[kQuadThreads * kMaskWords];   // each thread's direction mask (row stride 11: conflict-free)
  const uint32_t m2 = min(*P.m2_dev, P.cap2);
  const uint32_t begin = P.r0, end = min(m2, P.r1);
  const uint32_t hmask = hash_mask(P.ht);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) { st_n = 0; s_qsum = 0ull; s_csum = 0ull; }
  PROF_DECL;
  PROF_STAMP(0);
  __syncthreads();
  for (uint32_t i0 = begin + blockIdx.x * blockDim.x; i0 < end; i0 += gridDim.x * blockDim.x) {
#if defined(S4P_PROF)
    unsigned long long pa_, pb_, pc_, pd_, pe_;
    PROF_NOW(pa_);
#endif
    // phase A, one thread per set-2 pair of the tile: invariant point -> cell -> head of the cell's set-1 chain (super4pcs.cc:141,
    // normalset.hpp:162-171).  Typically well under half of the pairs fall into a cell that holds a set-1 pair; those are
    // compacted (ballot + per-wave offsets) so that the expensive part below runs on DENSE waves.
    {
      const uint32_t i = i0 + threadIdx.x;
      uint32_t e = kNil;
      // A share of the set (one GPU's part of a base) is defined on the pairs' ORDER KEYS, not on their positions: the
      // position of a pair in the list is whatever the appends of k_pairs made it on this device, its key is the same everywhere.
      if (i < end && (P.slice_den == 0u || P.okey2[i] % P.slice_den == P.slice_num)) {
        const int2 ab = P.ab2[i];
        const float p1x = P.ux[ab.x], p1y = P.uy[ab.x], p1z = P.uz[ab.x];
        const float p2x = P.ux[ab.y], p2y = P.uy[ab.y], p2z = P.uz[ab.y];
        const float nx = p2x - p1x, ny = p2y - p1y, nz = p2z - p1z;
        const uint32_t cell = index_pos(p1x + P.invariant2 * nx, p1y + P.invariant2 * ny, p1z + P.invariant2 * nz, P.qg);
        const unsigned long long mykey = ((unsigned long long)P.ht.epoch << 32) | cell;
        uint32_t h = hash_cell(cell) & hmask;
        while (true) {
          const unsigned long long k = P.ht.keys[h];
          if (k == mykey) { const unsigned long long hd = P.ht.heads[h]; e = (uint32_t(hd >> 32) == P.ht.epoch) ? uint32_t(hd) : kNil; break; }
          if (uint32_t(k >> 32) != P.ht.epoch) break;
          h = (h + 1u) & hmask;
        }
      }
      const unsigned long long m = __ballot(e != kNil);
      if (lane == 0) s_ic[wave] = uint32_t(__popcll(m));
      __syncthreads();
      uint32_t before = 0;
      for (uint32_t w = 0; w < wave; ++w) before += s_ic[w];
      if (e != kNil) {
        const uint32_t at = before + __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));
        s_item_i[at] = i; s_item_e[at] = e;
      }
      __syncthreads();
    }
    uint32_t n_items = 0;
#pragma unroll
    for (int w = 0; w < kQuadWaves; ++w) n_items += s_ic[w];
    PROF_NOW(pb_);
    uint32_t hops_ = 0; (void)hops_;
    if (threadIdx.x < n_items) {
      // phase B, one thread per pair with a chain: world point (super4pcs.cc:142) and the cone mask of its direction
      // (normalset.hpp:174-196) into the thread's LDS row -- what a separate preparation launch used to do for EVERY pair
      const uint32_t i = s_item_i[threadIdx.x];
      uint32_t e = s_item_e[threadIdx.x];
      const int2 ab2 = P.ab2[i];
      const uint32_t ok2 = P.okey2[i];
      uint32_t* row = s_mask + threadIdx.x * kMaskWords;
      float4 eq;
      { const float p1x = P.ux[ab2.x], p1y = P.uy[ab2.x], p1z = P.uz[ab2.x];
        const float p2x = P.ux[ab2.y], p2y = P.uy[ab2.y], p2z = P.uz[ab2.y];
        const float w1x = P.qx[ab2.x], w1y = P.qy[ab2.x], w1z = P.qz[ab2.x];
        const float w2x = P.qx[ab2.y], w2y = P.qy[ab2.y], w2z = P.qz[ab2.y];
        eq = make_float4(w1x + P.invariant2 * (w2x - w1x), w1y + P.invariant2 * (w2y - w1y), w1z + P.invariant2 * (w2z - w1z), 0.f);
        cone_mask_row(P.cone, P.qg.nepsilon, p2x - p1x, p2y - p1y, p2z - p1z, row); }
      PROF_NOW(pc_);
      // phase C: the walk is a chain of dependent gathers (one set-1 pair per hop), so each hop is ONE round trip: the hop's
      // direction bucket, world point and successor are requested together; the bucket test reads the LDS row.
      while (e != kNil) {
        const uint32_t b = P.bucket1[e];
        const float4 ep = P.ew1[e];
        const uint32_t nxt = P.next1[e];
        const float dx = eq.x - ep.x, dy = eq.y - ep.y, dz = eq.z - ep.z;
        if (((row[b >> 5] >> (b & 31u)) & 1u) && sqn3(dx, dy, dz) <= P.thr &&       // super4pcs.cc:160
            (P.k1_all || (P.okey1[e] >= P.k1_lo && P.okey1[e] < P.k1_hi))) {
          const int2 ab1 = P.ab1[e];
          const int4 quad = make_int4(ab1.x, ab1.y, ab2.x, ab2.y);                 // :171-172
          const unsigned long long tag = ((unsigned long long)P.okey1[e] << 32) | ok2;
          const uint32_t slot = atomicAdd(&st_n, 1u);
          if (slot < uint32_t(kQuadStage)) { st_q[slot] = quad; st_t[slot] = tag; }
          else {                                                                    // stage full (rare): direct append
            const unsigned long long at = atomicAdd(P.K_dev, 1ull);
            const unsigned long long mix = quad_mix(quad.x, quad.y, quad.z, quad.w);
            atomicAdd(&s_qsum, mix);
            if (at < P.K_cap) {
              P.quads[at] = quad; P.tags[at] = tag;
              if (P.do_gate) {
                float T[12];
                const int vd = gate_quad<ANGLE>(P.gate, quad, T);
                if (vd) { store_candidate(P.gate, atomicAdd(P.gate.C_dev, 1u), uint32_t(at) | (vd == 2 ? kBorderFlag : 0u), T, tag); atomicAdd(&s_csum, mix); }
                else P.gate.counts[at] = kGateFailed;
              }
            } else atomicOr(P.overflow, 4u);
          }
        }
        e = nxt; ++hops_;
      }
    }
#if defined(S4P_PROF)
    else { pc_ = pb_; }
#endif
    PROF_NOW(pd_);
    __syncthreads();
    const uint32_t n = min(st_n, uint32_t(kQuadStage));
    if (n) {                                                   // uniform
      if (threadIdx.x == 0) st_base = atomicAdd(P.K_dev, (unsigned long long)n);
      __syncthreads();
      const unsigned long long base = st_base;
      for (uint32_t c0 = 0; c0 < n; c0 += blockDim.x) {        // uniform trip count
        const uint32_t e = c0 + threadIdx.x;
        const unsigned long long at = base + e;
        const bool live = e < n && at < P.K_cap;
        if (e < n && at >= P.K_cap) atomicOr(P.overflow, 4u);
        int4 quad = make_int4(0, 0, 0, 0);
        unsigned long long mix = 0ull;
        if (e < n) { quad = st_q[e]; mix = quad_mix(quad.x, quad.y, quad.z, quad.w); }   // counted (and summed) even when it does not fit
        if (live) { P.quads[at] = quad; P.tags[at] = st_t[e]; }
        { const unsigned long long ws = wave_sum_u64(mix); if (lane == 0 && ws) atomicAdd(&s_qsum, ws); }
        if (P.do_gate) {                                        // uniform
          float T[12];
          const int vd = live ? gate_quad<ANGLE>(P.gate, quad, T) : 0;
          const bool ok = vd != 0;
          if (live && !ok) P.gate.counts[at] = kGateFailed;
          const unsigned long long pass = __ballot(ok);
          { const unsigned long long ws = wave_sum_u64(ok ? mix : 0ull); if (lane == 0 && ws) atomicAdd(&s_csum, ws); }
          if (lane == 0) s_wc[wave] = uint32_t(__popcll(pass));
          __syncthreads();
          if (threadIdx.x == 0) { uint32_t tot = 0; for (int w = 0; w < kQuadWaves; ++w) tot += s_wc[w]; s_cbase = tot ? atomicAdd(P.gate.C_dev, tot) : 0u; }
          __syncthreads();
          if (ok) {
            uint32_t before = 0;
            for (uint32_t w = 0; w < wave; ++w) before += s_wc[w];
            store_candidate(P.gate, s_cbase + before + uint32_t(__popcll(pass & ((1ull << lane) - 1ull))), uint32_t(at) | (vd == 2 ? kBorderFlag : 0u), T, e < n ? st_t[e] : 0ull);
          }
          __syncthreads();                                      // s_wc / s_cbase are rewritten by the next chunk
        }
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) st_n = 0;
    __syncthreads();
#if defined(S4P_PROF)
    PROF_NOW(pe_);
    tp_[2] += pb_ - pa_; tp_[3] += pc_ - pb_; tp_[4] += pd_ - pc_; tp_[5] += pe_ - pd_; tp_[6] += 1;
    { uint32_t hm_ = hops_;
      for (int o_ = 32; o_ > 0; o_ >>= 1) { const uint32_t v_ = uint32_t(__shfl_xor(int(hm_), o_)); hm_ = v_ > hm_ ? v_ : hm_; }
      if (hm_ > tp_[7]) tp_[7] = hm_; tp_[8] += n; }
#endif
  }
  PROF_STAMP(1);
  PROF_WRITE(1, (blockIdx.y * gridDim.x + blockIdx.x) * uint32_t(kQuadWaves) + wave);
  if (threadIdx.x == 0) {                                      // one pair of global atomics per workgroup that found anything
    if (s_qsum) atomicAdd(P.qsum_dev, s_qsum);
    if (s_csum) atomicAdd(P.csum_dev, s_csum);
  }
}

// ---------------------------------------------------------------------------
// k_sweep (round 6): the FIRST PASS of Verify when an early-exit bound is in force (VerifyParams::prune > 0: the trial loops).
// All but a few candidates of a base are dismissed by "how many queries COULD still be inliers": the number of sampled-Q points
// whose coarse cube (L0 bit) is marked under the candidate's transform is an upper bound of its inlier count, and a candidate
// whose bound does not EXCEED prune cannot become the best (match4pcsBase.hpp:468).  Until round 5 that count fell
// out of k_verify's lean sweep, which also queued every L0 survivor for the levels below and so carried a queue per wave (18 KB of
// LDS), scalar bookkeeping per chunk (SALU 0.6 x VALU), a candidate-record round trip per candidate -- and, for samples that do
// not fit LDS (n_Q > 2560: the 20 000-point sample), re-streamed the whole query array from L2 for EVERY candidate (4.6 TB/s of
// L2 reads at 14 M candidates/s).  This pass only COUNTS:
//   * one wave per BLOCK of kSweepCands candidates; their records are fetched together (one exposure), their coarse-unit
//     transforms live in registers;
//   * the sampled Q goes through LDS in TILES of tile_q points (x | y | z floats, padded with far-away points), staged once per
//     workgroup and tile and swept by every wave for all its block's candidates: query traffic / (waves x kSweepCands);
//   * per 64 queries and candidate: 3 LDS reads, the packed locate (grid_cell2), clamp, one LDS word, bit extract, ADD -- no
//     ballot, no queue, no scalar work; one wave reduction per candidate and tile;
//   * a candidate whose count + unswept queries <= prune is dead: its per-quad count reads 0 (a lower bound, as the
//     reference's is for what it abandons); the others are copied -- record and candidate index -- to the survivor list, one
//     global atomic per workgroup and base, and k_verify scores exactly those.
// Identical results by construction: k_verify applies the same bound again, with the same locate.
// ---------------------------------------------------------------------------
constexpr int kSweepCands = 4;                            // candidates per wave and pass over the query tiles
constexpr uint32_t kSweepTileMax = 2560;                  // queries per LDS tile (30 KB); larger samples take several tiles of 2048
constexpr int kSweepSurvCap = 1024;                       // survivors a workgroup stages between two flushes
struct SweepBase {
  const float4* cand_T; float4* surv_T;                   // gated candidates (64-byte records) -> survivors (same records, .w of the last row = candidate index)
  DevCounters* ctr; uint32_t* counts;
};
struct SweepParams {
  LcpGrid grid;
  const float* qtiles;                                    // sampled Q in sweep order: per tile x[tile_q] | y[tile_q] | z[tile_q], padded with kLeanPad
  uint32_t n_q, tile_q, n_tiles;
  SweepBase b[kGroupMax]; uint32_t n_bases;
  uint32_t prune;
};
struct SweepShared {
  uint32_t end[kGroupMax], next, n_surv, dead[kGroupMax], base_pos[kGroupMax];
  uint32_t surv[kSweepSurvCap];                           // base << 28 | candidate index
};
static_assert(sizeof(SweepParams) <= 4096, "SweepParams travels by value in the 4 KB kernel-argument segment");
static_assert(kGroupMax <= 8, "a staged survivor keeps its base in 3 bits... (28-bit candidate index)");

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += uint32_t(__shfl_xor(int(v), o));
  return v;
}

__global__ __launch_bounds__(1024, 4) void k_sweep(SweepParams P) {      // (launched with k_verify's grid and block: <= kVerifyMaxThreads)
  extern __shared__ uint32_t s_mem[];
  uint32_t* s_coarse = s_mem;                               // LDS: coarse bitmap (address 0) | query tile x | y | z | SweepShared
  float* s_qx = reinterpret_cast<float*>(s_mem + P.grid.coarse_words);
  float* s_qy = s_qx + P.tile_q; float* s_qz = s_qy + P.tile_q;
  SweepShared& S = *reinterpret_cast<SweepShared*>(s_qz + P.tile_q);
  const uint32_t lane = threadIdx.x & 63u, wave = uint32_t(__builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6))), n_waves = blockDim.x >> 6;
  const uint32_t nb = P.n_bases;
  if (threadIdx.x == 0) {
    uint32_t acc = 0;
    for (uint32_t b = 0; b < uint32_t(kGroupMax); ++b) {
      const uint32_t Cb = (b < nb && !(P.b[b].ctr->overflow & 4u)) ? P.b[b].ctr->C : 0u;      // (a pass whose quads overflowed is redone in chunks: nothing to score)
      acc += blockIdx.x < Cb ? (Cb - blockIdx.x + gridDim.x - 1u) / gridDim.x : 0u;
      S.end[b] = acc; S.dead[b] = 0u;
    }
    S.next = 0u; S.n_surv = 0u;
  }
  __syncthreads();
  const uint32_t hi = uint32_t(__builtin_amdgcn_readfirstlane(int(S.end[kGroupMax - 1])));
  if (hi == 0u) return;                                     // (uniform) more workgroups than candidates
  stage_coarse(P.grid, s_coarse);                           // ends with a workgroup barrier
  auto stage_tile = [&](const uint32_t t) {
    const float4* src = reinterpret_cast<const float4*>(P.qtiles + size_t(t) * 3u * P.tile_q);
    float4* dst = reinterpret_cast<float4*>(s_qx);
    for (uint32_t w = threadIdx.x; w < (3u * P.tile_q) >> 2; w += blockDim.x) dst[w] = src[w];      // (tile_q is a multiple of 256)
  };
  if (P.n_tiles == 1u) { stage_tile(0u); __syncthreads(); }
  // the survivors staged so far -> the bases' survivor lists (all threads; between barriers)
  auto flush = [&]() {
    const uint32_t n = min(S.n_surv, uint32_t(kSweepSurvCap));
    if (n == 0u) return;                                    // (uniform)
    if (threadIdx.x < uint32_t(kGroupMax)) {
      uint32_t cnt = 0;
      for (uint32_t e = 0; e < n; ++e) cnt += (S.surv[e] >> 28) == threadIdx.x ? 1u : 0u;
      S.base_pos[threadIdx.x] = cnt ? atomicAdd(&P.b[threadIdx.x].ctr->S, cnt) : 0u;
    }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < n; e += blockDim.x) {
      const uint32_t w = S.surv[e], b = w >> 28, i = w & 0x0FFFFFFFu;
      uint32_t before = 0;
      for (uint32_t f = 0; f < e; ++f) before += (S.surv[f] >> 28) == b ? 1u : 0u;      // (a few dozen entries per flush)
      const float4* src = P.b[b].cand_T + kCandStride * size_t(i);
      float4* dst = P.b[b].surv_T + kCandStride * size_t(S.base_pos[b] + before);
      const float4 r3 = src[3];
      dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
      dst[3] = make_float4(r3.x, r3.y, r3.z, __uint_as_float(i));
    }
    __syncthreads();
    if (threadIdx.x == 0) S.n_surv = 0u;
    __syncthreads();
  };
  // pitches of the coarse bitmap include one empty border cube per axis (LcpGridHost::plan): a coordinate outside clamps onto it
  const uint32_t ucx = uint32_t(P.grid.cnx), ucy = uint32_t(P.grid.cny);
  const uint32_t mx = ucx - 1u, my = ucy - 1u, mz = uint32_t(((P.grid.nz - 1) >> P.grid.cshift) + 1);
  const float cs = coarse_scale(P.grid);
  // every wave of the workgroup takes part in every ROUND (the tile staging is a workgroup affair); a round = kSweepCands tickets per wave
  const uint32_t per_round = n_waves * uint32_t(kSweepCands), rounds = (hi + per_round - 1u) / per_round;
  for (uint32_t r = 0; r < rounds; ++r) {
    const uint32_t t_first = (r * n_waves + wave) * uint32_t(kSweepCands);
    GridXf X[kSweepCands]; uint32_t bsel[kSweepCands], ci[kSweepCands], kq[kSweepCands], cnt[kSweepCands];
    bool valid[kSweepCands], alive[kSweepCands], keep[kSweepCands];
#pragma unroll
    for (int k = 0; k < kSweepCands; ++k) {
      const uint32_t t = t_first + uint32_t(k);
      valid[k] = t < hi; alive[k] = valid[k]; keep[k] = false; cnt[k] = 0u; kq[k] = 0u;
      uint32_t bs = 0u, t0 = 0u;
#pragma unroll
      for (int b = 1; b < kGroupMax; ++b) { const uint32_t e = S.end[b - 1]; if (t >= e) { bs = uint32_t(b); t0 = e; } }
      bsel[k] = uint32_t(__builtin_amdgcn_readfirstlane(int(bs)));
      ci[k] = blockIdx.x + (t - uint32_t(__builtin_amdgcn_readfirstlane(int(t0)))) * gridDim.x;
    }
#pragma unroll
    for (int k = 0; k < kSweepCands; ++k) {                 // the block's records: all loads in flight together
      const float4* src = P.b[valid[k] ? bsel[k] : 0u].cand_T + kCandStride * size_t(valid[k] ? ci[k] : 0u);
      const float4 r0 = src[0], r1 = src[1], r2 = src[2], r3 = src[3];
      const float T[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
      X[k] = make_grid_xf(P.grid, T, cs);
      const uint32_t kraw = uint32_t(__builtin_amdgcn_readfirstlane(int(__float_as_uint(r3.z))));
      kq[k] = kraw & ~kBorderFlag;
      // a candidate whose Euler-angle gate the host still has to settle goes to k_verify whatever its count: only there is it
      // entered into the list the host works through (it may turn out not to be a candidate at all)
      keep[k] = valid[k] && (kraw & kBorderFlag) != 0u;
    }
    for (uint32_t tile = 0; tile < P.n_tiles; ++tile) {     // uniform
      if (P.n_tiles > 1u) { __syncthreads(); stage_tile(tile); __syncthreads(); }
      const uint32_t swept_after = min((tile + 1u) * P.tile_q, P.n_q);
#pragma unroll
      for (int k = 0; k < kSweepCands; ++k) {
        if (!alive[k] || keep[k]) continue;                 // (wave-uniform)
        uint32_t hits = 0u;
        for (uint32_t base = 0; base < P.tile_q; base += kSweepStep) {
          float x[kSweepChunks], y[kSweepChunks], z[kSweepChunks];
          int cx[kSweepChunks], cy[kSweepChunks], cz[kSweepChunks];
          uint32_t cc[kSweepChunks], ww[kSweepChunks];
#pragma unroll
          for (uint32_t c = 0; c < kSweepChunks; ++c) { const uint32_t i = base + 64u * c + lane; x[c] = s_qx[i]; y[c] = s_qy[i]; z[c] = s_qz[i]; }
#pragma unroll
          for (uint32_t c = 0; c < kSweepChunks; c += 2u)
            grid_cell2(X[k].u, make_float4(x[c], y[c], z[c], 0.f), make_float4(x[c + 1u], y[c + 1u], z[c + 1u], 0.f), cx[c], cy[c], cz[c], cx[c + 1u], cy[c + 1u], cz[c + 1u]);
#pragma unroll
          for (uint32_t c = 0; c < kSweepChunks; ++c) {
            cc[c] = mad24_s(mad24_s(min(uint32_t(cz[c]), mz), ucy, min(uint32_t(cy[c]), my)), ucx, min(uint32_t(cx[c]), mx));
            ww[c] = lds_word(0u, cc[c] >> 5);              // (the bitmap starts at LDS address 0)
          }
#pragma unroll
          for (uint32_t c = 0; c < kSweepChunks; ++c) hits += bfe1(ww[c], cc[c]);
        }
        cnt[k] += wave_sum_u32(hits);
        if (cnt[k] + (P.n_q - swept_after) <= P.prune) alive[k] = false;      // cannot exceed the bound any more
      }
    }
    // verdicts of the block: dead -> its quad's count reads 0; alive -> staged for the survivor list
#pragma unroll
    for (int k = 0; k < kSweepCands; ++k) {
      if (!valid[k]) continue;                              // (wave-uniform)
      if (!alive[k]) {
        if (lane == 0) { P.b[bsel[k]].counts[kq[k]] = 0u; atomicAdd(&S.dead[bsel[k]], 1u); }
      } else if (lane == 0) {
        const uint32_t slot = atomicAdd(&S.n_surv, 1u);
        if (slot < uint32_t(kSweepSurvCap)) S.surv[slot] = (bsel[k] << 28) | ci[k];
        else {                                              // stage full (a base whose candidates nearly all survive): straight to the list
          const uint32_t pos = atomicAdd(&P.b[bsel[k]].ctr->S, 1u);
          const float4* src = P.b[bsel[k]].cand_T + kCandStride * size_t(ci[k]);
          float4* dst = P.b[bsel[k]].surv_T + kCandStride * size_t(pos);
          const float4 r3 = src[3];
          dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = make_float4(r3.x, r3.y, r3.z, __uint_as_float(ci[k]));
        }
      }
    }
    // (uniform: every wave runs the same rounds)  several tiles: a barrier per round anyway; one tile: every 16th round -- at most
    // 16 x 16 waves x kSweepCands = 1024 survivors between two looks at the stage, so the direct path above stays a safety net
    if (P.n_tiles > 1u || (r & 15u) == 15u) { __syncthreads(); if (S.n_surv > uint32_t(kSweepSurvCap) / 4u) flush(); }
  }
  __syncthreads();
  flush();
  if (threadIdx.x < nb && S.dead[threadIdx.x]) atomicAdd(&P.b[threadIdx.x].ctr->pruned, S.dead[threadIdx.x]);
}

// ---------------------------------------------------------------------------
// k_verify: Verify() (match4pcsBase.cc:508-567) of every gated candidate, then -- in the same launch -- the selection of
// the base's winner (match4pcsBase.hpp:467-484: the first candidate in reference order with the strictly greatest LCP),
// its transform, and the result record the host reads.  With VerifyParams::prune > 0 a candidate that can no longer
// EXCEED that count is abandoned (the order-independent form of match4pcsBase.cc:558-560); with
// prune == 0 every candidate is counted in full.
// Persistent workgroups of up to 1024 threads, one wave64 per gated candidate; the length of the gated list lives in
// device memory (no host round trip).  LDS per workgroup: coarse bitmap (<= 34 KB) + one private survivor queue per wave.
// ---------------------------------------------------------------------------
// k_verify / k_verify_T are launched with kVerifyThreadsCached threads per workgroup (measured 0.1496 / 0.1443 / 0.1424 /
// 0.1433 / 0.1458 ms at 512 / 640 / 768 / 896 / 1024 threads, tools/gpu_run19.sh; S4P_VERIFY_THREADS overrides) and with one
// workgroup per CU while the structure is cache resident, two when the point lines stream from HBM or a chunk pass has the
// chip to itself (s4p_capi.hip: verify_blocks / verify_grid).  The block size is a launch parameter; the kernels only assume
// blockDim.x <= kVerifyMaxThreads.
constexpr int kVerifyMaxThreads = 1024;       // (k_sweep's launch bound above says the same)
constexpr int kVerifyThreadsCached = 768;
constexpr int kVerifyMaxBlocks = 4096;
struct VerifyBase {                                     // one base of the launch
  BaseFrame base;
  const int4* quads; const unsigned long long* tags; uint32_t* counts;
  const uint32_t* cand_idx; const float4* cand_T;       // gated candidates: quad index + 3x4 transform
  const float4* surv_T;                                 // (VerifyParams::use_surv) the candidates k_sweep let through: same records, the candidate's index in .w of the last row
  DevCounters* ctr;                                     // live counters of the base (reset by the last workgroup)
  DevCounters* res;                                     // result record of the base: pinned host memory, written by the last workgroup
  uint4* slots;                                         // per workgroup: {best count, its candidate, tag lo, tag hi}
  uint32_t* border;                                     // candidates (positions in cand_idx) with an undecided gate, kBorderCap entries
};
struct VerifyParams {
  LcpGrid grid;
  const float4* q4;                                     // sampled Q (centred), packed (x,y,z,0), original order (quad indices)
  const float4* q4v;                                    // the same points in Morton order, for the LCP sweep
  QuantQ qq;                                            // ... and their 16-bit quantisation (QLDS kernels)
  const float* qsoa;                                    // ... and as x[n_pad] | y[n_pad] | z[n_pad], padded with kLeanPad (LEAN kernels)
  uint32_t n_q;
  VerifyBase b[kGroupMax]; uint32_t n_bases;            // the bases of this launch (1 .. kGroupMax)
  uint32_t* group_done;                                 // workgroups that have published their bests (the last one selects the winners); left at 0
  uint32_t seq;                                         // launch number, written last into every result record
  uint32_t prune;                                       // best inlier count of the registration at launch (LcpTask::prune), 0 = count every candidate in full
  uint32_t use_surv;                                    // 1: k_sweep ran first -- score the survivor lists (ctr->S entries of surv_T) instead of all gated candidates
  int count_tests;                                      // instrumentation counters are live: carry them into res
  int ablate;                                           // S4P_ABLATE debugging only (0 = full kernel)
};

static_assert(sizeof(VerifyParams) <= 4096, "VerifyParams travels by value in the 4 KB kernel-argument segment");
struct VerifyShared {                                   // k_verify's workgroup scalars, at the end of its dynamic LDS
  unsigned long long wtag[kGroupMax][kVerifyMaxThreads / 64];
  uint32_t wcnt[kGroupMax][kVerifyMaxThreads / 64], wcand[kGroupMax][kVerifyMaxThreads / 64];
  uint32_t pruned[kGroupMax];
  uint32_t end[kGroupMax];                              // end of base b's tickets in this workgroup's ticket space
  uint32_t next, last;
};

// better(a, b): a wins over b if its count is greater, or equal with a smaller tag (= earlier in reference order)
__device__ __forceinline__ bool slot_better(const uint32_t ca, const unsigned long long ta, const uint32_t cb, const unsigned long long tb, const bool b_valid) {
  return !b_valid || ca > cb || (ca == cb && ta < tb);
}
struct WaveBest { uint32_t c, i; unsigned long long t; };      // (count, candidate, tag); i == kNil: none yet

// One ticket from an LDS counter for the whole wave: lane 0 alone performs the add (exec is narrowed around the ONE
// instruction, inside the asm statement), every lane gets the old value.  No lane-dependent control flow the compiler can
// see or move: an `if (lane == 0)` around the atomic lets it thread lane 0's path through a loop's back edge (k_verify),
// and an all-lanes atomicAdd(lane == 0 ? 1 : 0) is turned into a 64-step scalar scan per ticket.
__device__ __forceinline__ uint32_t wave_ticket(uint32_t* counter) {
  typedef __attribute__((address_space(3))) uint32_t* lds_ptr;
  const uint32_t addr = uint32_t(uintptr_t((lds_ptr)counter));
  uint32_t r; unsigned long long saved;
  asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tds_add_rtn_u32 %0, %2, %3\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, %1"
               : "=&v"(r), "=&s"(saved) : "v"(addr), "v"(1u) : "memory");
  return uint32_t(__builtin_amdgcn_readfirstlane(int(r)));
}

// LEAN (launched when an early-exit bound is in force): wave_lcp_count_lean, LDS = coarse bitmap | float queries x, y, z (QLDS:
// the sample fits) | one 16-bit queue per wave.  Without QLDS the lean sweep reads the queries from the padded float4 array in
// global memory (VerifyParams::q4v holds n_pad entries): the 20 000-point sample.
// One launch scores the candidate lists of up to kGroupMax bases: the LDS tables (58 KB per workgroup) are staged once, the
// waves draw candidates from ONE ticket counter over the workgroup's shares of all lists (a heavy candidate of one base
// overlaps the cheap ones of the others), every wave keeps one best per base, and the last workgroup to finish writes one
// result record per base.
template <bool COUNT, bool QLDS, bool LEAN>
__global__ __launch_bounds__(kVerifyMaxThreads, 6) void k_verify(VerifyParams P) {   // <= 80 VGPRs: six waves per SIMD, i.e. two 768-thread workgroups per CU (of one launch, or of two)
  PROF_DECL;
  PROF_STAMP(0);
  extern __shared__ uint32_t s_mem[];
  uint32_t* s_coarse = s_mem;                              // LDS: coarse bitmap | quantised queries (QLDS) | 16 survivor queues
  uint2* s_q = reinterpret_cast<uint2*>(s_mem + P.grid.coarse_words);
  uint32_t* s_queue = reinterpret_cast<uint32_t*>(s_q + (QLDS ? ((P.n_q + kSweepStep - 1u) & ~(kSweepStep - 1u)) : 0u)) + (threadIdx.x >> 6) * kQueueWordsPerWave;
  const uint32_t n_pad = (P.n_q + kSweepStep - 1u) & ~(kSweepStep - 1u);
  LeanLds LL;
  LL.coarse = s_coarse;
  const uint32_t lean_q_words = QLDS ? 3u * n_pad : 0u;     // lean kernels: QLDS = the float copy of the queries is staged in LDS
  { float* f = reinterpret_cast<float*>(s_mem + P.grid.coarse_words);
    LL.qx = f; LL.qy = f + n_pad; LL.qz = f + 2u * n_pad;
    LL.queue = reinterpret_cast<uint16_t*>(f + lean_q_words) + uint32_t(__builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6))) * kLeanQueue; }   // (uniform: scalar register)
  // The workgroup's few scalars live at the END of the dynamic segment (VerifyShared), not in static __shared__: the coarse
  // bitmap then starts at LDS address 0 and the sweep's word address needs no base added (one vector instruction per chunk).
  VerifyShared& S = *reinterpret_cast<VerifyShared*>(LEAN
      ? reinterpret_cast<uint32_t*>(reinterpret_cast<uint16_t*>(s_mem + P.grid.coarse_words + lean_q_words) + (blockDim.x >> 6) * kLeanQueue)
      : reinterpret_cast<uint32_t*>(s_q + (QLDS ? n_pad : 0u)) + (blockDim.x >> 6) * kQueueWordsPerWave);
  uint32_t& s_next = S.next; uint32_t& s_last = S.last;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t nb = P.n_bases;
  // Work split: every workgroup owns a fixed share of every base's gated candidate list (static: a single-address global
  // cursor caps at ~90 dequeues/us, MI355X_MICROARCH "dequeue"); inside the shares its waves take candidates from an LDS
  // counter, so a wave that drew cheap candidates (few L0 survivors) simply takes more.
  // Share of this workgroup in a list: candidates blockIdx.x, blockIdx.x + gridDim.x, ...  Neighbours in the candidate order
  // are neighbours in quad order and cost about the same, so contiguous slices made some workgroups consistently slower than
  // others, and the launch ends with its slowest workgroup (0.145 -> 0.136 ms alone with the strided share).  A global
  // counter for the tail of the list was tried on top and is not here: even ~8000 single-address atomics per launch cost
  // more than the imbalance they remove (0.136 -> 0.172 ms, profiles/HISTORY.md).
  // (the per-base bookkeeping -- ticket ranges, every wave's best per base -- lives in LDS, not in registers: the sweep leaves no
  // scalar registers to spare, and a candidate costs microseconds against one LDS round trip)
  if (threadIdx.x == 0) {
    uint32_t acc = 0;
    for (uint32_t b = 0; b < uint32_t(kGroupMax); ++b) {
      const uint32_t Cb = b < nb ? (P.use_surv ? P.b[b].ctr->S : P.b[b].ctr->C) : 0u;
      acc += blockIdx.x < Cb ? (Cb - blockIdx.x + gridDim.x - 1u) / gridDim.x : 0u;
      S.end[b] = acc; S.pruned[b] = 0u;
    }
    s_next = 0u;
  }
  if (lane == 0)
    for (uint32_t b = 0; b < uint32_t(kGroupMax); ++b) { S.wcnt[b][wave] = 0u; S.wcand[b][wave] = kNil; S.wtag[b][wave] = ~0ull; }
  __syncthreads();
  // (read back through readfirstlane: the compiler cannot know that an LDS word is the same in every lane, and a ticket loop
  // whose exit it takes for divergent is structurised into nested loops with partial exec masks around the lane-0 atomic --
  // measured: that build hung on the device)
  const uint32_t hi = uint32_t(__builtin_amdgcn_readfirstlane(int(S.end[kGroupMax - 1])));
  if (hi != 0u && P.ablate != 2) {                         // (uniform) otherwise: more workgroups than candidates
    LcpTask K;
    K.q4 = P.q4v; K.n_q = P.n_q; K.qq = P.qq; K.T = P.b[0].cand_T; K.t_stride = kCandStride; K.point_tests = &P.b[0].ctr->point_tests;
    K.prune = P.prune; K.pruned = &S.pruned[0];
    if (LEAN) { if (QLDS) stage_queries_f(P.qsoa, const_cast<float*>(LL.qx), 3u * n_pad); }
    else if (QLDS) stage_queries(K, s_q);
    stage_coarse(P.grid, s_coarse);                        // ends with a workgroup barrier
    PROF_STAMP(1);
    // A candidate is one 64-byte record {3x4 transform | tag, quad index}: one line, everything the wave needs of it.  (Holding
    // the NEXT candidate's record in registers while the current one is swept was measured in round 4: 16 more VGPRs, no gain;
    // so was finishing "heavy" candidates -- those that stay alive through the whole sweep -- by the whole workgroup after the
    // ticket loop: slower, most of them die in their first exact batch.  profiles/HISTORY.md.)
    // The loop body has NO lane-dependent control flow of its own: the ticket is drawn by wave_ticket (lane 0 adds, inside one asm
    // statement), the candidate's bookkeeping afterwards is done by all lanes with identical values.  With `if (lane == 0)` regions on both sides of the loop's back edge the compiler threaded lane 0's
    // path through the edge and re-entered the loop with lanes 1..63 alone -- readfirstlane then read THEIR (zero) ticket: the
    // group build hung on the device (round 5; the breadcrumb build that perturbed the code did not).
    while (true) {
      const uint32_t t = wave_ticket(&s_next);
      if (t >= hi) break;
      // ticket -> (base, position in this workgroup's share of its list)
      uint32_t bsel = 0u, t0 = 0u;
#pragma unroll
      for (int b = 1; b < kGroupMax; ++b) { const uint32_t e = S.end[b - 1]; if (t >= e) { bsel = uint32_t(b); t0 = e; } }
      bsel = uint32_t(__builtin_amdgcn_readfirstlane(int(bsel))); t0 = uint32_t(__builtin_amdgcn_readfirstlane(int(t0)));
      const VerifyBase& B = P.b[bsel];
      const uint32_t li = blockIdx.x + (t - t0) * gridDim.x;           // position in the list this launch scores
      const float4* src = (P.use_surv ? B.surv_T : B.cand_T) + kCandStride * size_t(li);         // one candidate per wave
      const float4 r0 = src[0], r1 = src[1], r2 = src[2], r3 = src[3];
#if defined(S4P_PROF)
      const unsigned long long tc0_ = __builtin_amdgcn_s_memrealtime();
#endif
      K.point_tests = &B.ctr->point_tests; K.pruned = &S.pruned[bsel];
      uint32_t cnt;
      if (LEAN) cnt = P.ablate == 1 ? wave_lcp_count_lean<COUNT, true, QLDS>(P.grid, K, LL, src, r0, r1, r2) : wave_lcp_count_lean<COUNT, false, QLDS>(P.grid, K, LL, src, r0, r1, r2);
      else cnt = P.ablate == 1 ? wave_lcp_count_auto<COUNT, true, QLDS>(P.grid, K, s_coarse, s_q, s_queue, src)
                               : wave_lcp_count_auto<COUNT, false, QLDS>(P.grid, K, s_coarse, s_q, s_queue, src);
      { // (the record's last 16 bytes are re-read: a line this wave has just held; nothing lives in registers across the sweep)
        const float4 rr = src[3];
        const uint32_t i = P.use_surv ? uint32_t(__builtin_amdgcn_readfirstlane(int(__float_as_uint(rr.w)))) : li;      // the candidate's index in the gated list
        const uint32_t kraw = uint32_t(__builtin_amdgcn_readfirstlane(int(__float_as_uint(rr.z)))), k = kraw & ~kBorderFlag;
        const unsigned long long tag = (unsigned long long)uint32_t(__builtin_amdgcn_readfirstlane(int(__float_as_uint(rr.x)))) |
                                       ((unsigned long long)uint32_t(__builtin_amdgcn_readfirstlane(int(__float_as_uint(rr.y)))) << 32);
        B.counts[k] = cnt;                                   // (every lane, same address, same value)
        if (kraw & kBorderFlag) {                            // (uniform) scored, but the host decides whether it is a candidate at all
          uint32_t n = 0;
          if (lane == 0) n = atomicAdd(&B.ctr->n_border, 1u);
          n = uint32_t(__builtin_amdgcn_readfirstlane(int(n)));
          if (n < kBorderCap) B.border[n] = i;
        } else {
          const uint32_t oc = uint32_t(__builtin_amdgcn_readfirstlane(int(S.wcnt[bsel][wave]))), oi = uint32_t(__builtin_amdgcn_readfirstlane(int(S.wcand[bsel][wave])));
          const unsigned long long ot = S.wtag[bsel][wave];
          const unsigned long long otu = (unsigned long long)uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(ot)))) | ((unsigned long long)uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(ot >> 32)))) << 32);
          if (slot_better(cnt, tag, oc, otu, oi != kNil)) { S.wcnt[bsel][wave] = cnt; S.wtag[bsel][wave] = tag; S.wcand[bsel][wave] = i; }   // (uniform)
        }
      }
      (void)r3;
      __builtin_amdgcn_wave_barrier();
#if defined(S4P_PROF)
      { __builtin_amdgcn_s_waitcnt(0); const unsigned long long d_ = __builtin_amdgcn_s_memrealtime() - tc0_;
        tp_[5] += 1; if (d_ > tp_[6]) tp_[6] = d_; if (d_ > 800ull) { tp_[7] += 1; tp_[8] += d_; } tp_[9] += d_; if (d_ > 2000ull) tp_[10] += 1; }
#endif
    }
  }
  PROF_STAMP(2);
  // ---- selection: wave bests (LDS) -> workgroup best -> slot; the last workgroup to finish reduces the slots ----
  auto wave_reduce = [&](WaveBest& w) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint32_t oc = uint32_t(__shfl_xor(int(w.c), o)), oi = uint32_t(__shfl_xor(int(w.i), o));
      const uint32_t tl = uint32_t(__shfl_xor(int(uint32_t(w.t)), o)), th = uint32_t(__shfl_xor(int(uint32_t(w.t >> 32)), o));
      const unsigned long long ot = (unsigned long long)tl | ((unsigned long long)th << 32);
      if (oi != kNil && slot_better(oc, ot, w.c, w.t, w.i != kNil)) { w.c = oc; w.t = ot; w.i = oi; }
    }
  };
  auto block_best = [&](const uint32_t b) -> WaveBest {     // thread 0, after a barrier: the workgroup's best of base b
    WaveBest r{0u, kNil, ~0ull};
    for (uint32_t w = 0; w < (blockDim.x >> 6); ++w)
      if (S.wcand[b][w] != kNil && slot_better(S.wcnt[b][w], S.wtag[b][w], r.c, r.t, r.i != kNil)) { r.c = S.wcnt[b][w]; r.t = S.wtag[b][w]; r.i = S.wcand[b][w]; }
    return r;
  };
  __syncthreads();
  PROF_STAMP(3);
  PROF_WRITE(2, blockIdx.x * 16u + wave);
  if (threadIdx.x == 0) {
    for (uint32_t b = 0; b < nb; ++b) {
      if (S.pruned[b]) atomicAdd(&P.b[b].ctr->pruned, S.pruned[b]);
      const WaveBest r = block_best(b);
      P.b[b].slots[blockIdx.x] = make_uint4(r.c, r.i, uint32_t(r.t), uint32_t(r.t >> 32));
    }
    __threadfence();                                       // release (agent scope): the slots are visible before the ticket
    const uint32_t ticket = atomicAdd(P.group_done, 1u);
    s_last = (ticket == gridDim.x - 1u) ? 1u : 0u;
  }
  __syncthreads();
  if (s_last == 0u) return;
  __threadfence();                                         // acquire: the other workgroups' slots
  for (uint32_t b = 0; b < nb; ++b) {                      // (S.w* are reused: every thread is past the barrier above)
    WaveBest r{0u, kNil, ~0ull};
    for (uint32_t w = threadIdx.x; w < gridDim.x; w += blockDim.x) {
      const uint4 sl = P.b[b].slots[w];
      const unsigned long long t = (unsigned long long)sl.z | ((unsigned long long)sl.w << 32);
      if (sl.y != kNil && slot_better(sl.x, t, r.c, r.t, r.i != kNil)) { r.c = sl.x; r.t = t; r.i = sl.y; }
    }
    wave_reduce(r);
    if (lane == 0) { S.wcnt[b][wave] = r.c; S.wcand[b][wave] = r.i; S.wtag[b][wave] = r.t; }
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  for (uint32_t b = 0; b < nb; ++b) {
    const VerifyBase& B = P.b[b];
    DevCounters* c = B.ctr;
    DevCounters* r = B.res;
    const WaveBest w = block_best(b);
    r->m1 = c->m1; r->m2 = c->m2; r->K = c->K; r->C = c->C; r->S = c->S; r->overflow = c->overflow;
    r->quad_sum = c->quad_sum; r->cand_sum = c->cand_sum; r->n_border = c->n_border; r->pruned = c->pruned;
    r->best_count = w.c; r->best_tag = w.t; r->has_best = 0u;
    if (COUNT || P.count_tests) { r->point_tests = c->point_tests; r->l0_pass = c->l0_pass; r->l1_pass = c->l1_pass; r->l2_pass = c->l2_pass; }
    if (w.i != kNil) {                                       // recompute the winner's 4x4 (ComputeRigidTransformation)
      const uint32_t k = B.cand_idx[w.i] & ~kBorderFlag;
      const int4 qd = B.quads[k];
      const float4 a = P.q4[qd.x], bq = P.q4[qd.y], cc = P.q4[qd.z];
      const float q[3][3] = {{a.x, a.y, a.z}, {bq.x, bq.y, bq.z}, {cc.x, cc.y, cc.z}};
      float T[12], c2[3];
      rigid_gate(B.base, q, T, c2);
      for (int i = 0; i < 12; ++i) r->best_T[i] = T[i];
      r->best_T[12] = 0.f; r->best_T[13] = 0.f; r->best_T[14] = 0.f; r->best_T[15] = 1.f;
      for (int i = 0; i < 3; ++i) r->best_c2[i] = c2[i];
      r->best_quad[0] = qd.x; r->best_quad[1] = qd.y; r->best_quad[2] = qd.z; r->best_quad[3] = qd.w;
      r->has_best = 1u;
    }
    // the live counters are ready for the next base on this lane (no separate reset launch)
    c->m1 = 0; c->m2 = 0; c->K = 0; c->C = 0; c->best_count = 0; c->overflow = 0; c->best_tag = ~0ull; c->has_best = 0;
    c->quad_sum = 0; c->cand_sum = 0; c->n_border = 0; c->pruned = 0; c->S = 0;
    c->point_tests = 0; c->l0_pass = 0; c->l1_pass = 0; c->l2_pass = 0;
    c->done = 0;
  }
  *P.group_done = 0u;
  if (P.ablate == 3) {                                     // S4P_ABLATE=3 (test aid): the launch stalls for ~3 s before its result records appear -- what the
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();      // host's watchdog (S4P_WAIT_TIMEOUT_S) has to turn into an error instead of a hang
    while (__builtin_amdgcn_s_memrealtime() - t0 < 300000000ull) __builtin_amdgcn_s_sleep(127);
  }
  __threadfence_system();                                  // the records (host memory) and the cleared counters before the launch number
  for (uint32_t b = 0; b < nb; ++b) __hip_atomic_store(&P.b[b].res->seq, P.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// k_verify_T: Verify() for explicit row-major 4x4 transforms (one wave per transform).
struct VerifyTParams {
  LcpGrid grid; const float4* q4; QuantQ qq; uint32_t n_q;
  const float* T; uint32_t B; uint32_t* counts; DevCounters* ctr;
};
template <bool COUNT, bool QLDS>
__global__ __launch_bounds__(kVerifyMaxThreads) void k_verify_T(VerifyTParams P) {
  extern __shared__ uint32_t s_mem[];
  uint32_t* s_coarse = s_mem;
  uint2* s_q = reinterpret_cast<uint2*>(s_mem + P.grid.coarse_words);
  uint32_t* s_queue = reinterpret_cast<uint32_t*>(s_q + (QLDS ? ((P.n_q + kSweepStep - 1u) & ~(kSweepStep - 1u)) : 0u)) + (threadIdx.x >> 6) * kQueueWordsPerWave;
  LcpTask K;
  K.q4 = P.q4; K.n_q = P.n_q; K.qq = P.qq; K.T = reinterpret_cast<const float4*>(P.T); K.t_stride = 4u;
  K.point_tests = COUNT ? &P.ctr->point_tests : nullptr;
  K.prune = 0u; K.pruned = nullptr;                        // explicit transforms are always counted in full
  if (QLDS) stage_queries(K, s_q);
  stage_coarse(P.grid, s_coarse);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t k = wave; k < P.B; k += nwaves) {
    const uint32_t cnt = wave_lcp_count_auto<COUNT, false, QLDS>(P.grid, K, s_coarse, s_q, s_queue, K.T + 4 * size_t(k));
    if (lane == 0) P.counts[k] = cnt;
  }
}

// ---------------------------------------------------------------------------
// k_apply: final rigid apply on the full-resolution cloud (match4pcsBase.hpp:265-267).
// 24 B/point of HBM traffic for 18 flop: bandwidth-bound; plain VALU keeps the
// reference's (non-fused) rounding, which an MFMA fma-chain would not.
// ---------------------------------------------------------------------------
struct ApplyParams { float M[12]; float* x; float* y; float* z; uint64_t n; };
__global__ __launch_bounds__(256) void k_apply(ApplyParams P) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < P.n; i += (uint64_t)gridDim.x * blockDim.x) {
    const float x = P.x[i], y = P.y[i], z = P.z[i];
    P.x[i] = ((P.M[0] * x + P.M[1] * y) + P.M[2] * z) + P.M[3];
    P.y[i] = ((P.M[4] * x + P.M[5] * y) + P.M[6] * z) + P.M[7];
    P.z[i] = ((P.M[8] * x + P.M[9] * y) + P.M[10] * z) + P.M[11];
  }
}

// The same contraction on the matrix cores, kept ONLY as the measured alternative:
// v_mfma_f32_4x4x1_16b_f32 = 16 independent 4x4x1 outer products, so with lane l <-> point l (block l/4, column l%4) four
// accumulating instructions (k = x, y, z, 1) leave rows 0..2 of [R|t] * [p;1] for point l in lane l's own registers.
// Each step is a FUSED multiply-add, so the result differs from the reference's separately rounded
// ((m0*x + m1*y) + m2*z) + m3 in the last bit of many coordinates -- which is why the product path does not use it.
__global__ __launch_bounds__(256) void k_apply_mfma(ApplyParams P) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const uint32_t lane = threadIdx.x & 63u, row = lane & 3u;
  // A operand of step k: lane (block, i) holds M[i][k]; row 3 is the homogeneous row (0 0 0 1), never stored
  const float a0 = row < 3u ? P.M[4 * row + 0] : 0.f, a1 = row < 3u ? P.M[4 * row + 1] : 0.f,
              a2 = row < 3u ? P.M[4 * row + 2] : 0.f, a3 = row < 3u ? P.M[4 * row + 3] : 1.f;
  const uint64_t nround = (P.n + 63ull) & ~63ull;            // whole waves: the MFMA needs all 64 lanes
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < nround; i += (uint64_t)gridDim.x * blockDim.x) {
    const bool live = i < P.n;
    const float x = live ? P.x[i] : 0.f, y = live ? P.y[i] : 0.f, z = live ? P.z[i] : 0.f;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a1, y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a2, z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a3, 1.f, acc, 0, 0, 0);
    if (live) { P.x[i] = acc[0]; P.y[i] = acc[1]; P.z[i] = acc[2]; }
  }
}

// ---------------------------------------------------------------------------
// Base selection on the device: SelectRandomTriangle's 1000-draw search (match4pcsBase.cc:185-218) and the
// 4th-point scan of SelectQuadrilateral (match4pcsBase.cc:303-338) as reductions over the sampled P resident in HBM
// (float4 records in sampling order).  The random stream stays on the host (std::mt19937, 2001 draws per attempt, none
// of them depends on a point); everything that reads points runs here.  Both loops of the reference keep "the first
// strictly better item", i.e. the lexicographic optimum of (value, position): packed into 64-bit keys
//   triangle:  max over  bits(area) << 32 | (0xFFFFFFFF - draw)      area = |u x w| > 0, both edges below the limit
//   4th point: min over  bits(dist) << 32 | index                     dist = |a x + b y + c z - 1| < FLT_MAX
// (non-negative floats order like their bit patterns).
// ---------------------------------------------------------------------------
struct SelectRecord {
  unsigned long long tri_key, fourth_key;
  int32_t ids[4];
  int32_t status;                 // kSelect*
  float pa, pb, pc;
  float xyz[12];
};
constexpr int32_t kSelectFound = 0, kSelectNoTriangle = 1, kSelectDegenerate = 2, kSelectNoFourth = 3;
constexpr int kSelectTriangles = 1000;      // kNumberOfDiameterTrials, match4pcsBase.cc:58
constexpr int kSelectDraws = 1 + 2 * kSelectTriangles;
constexpr int kSelectBatch = 16;            // attempts one s4p_select_base_points_batch call evaluates at most

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int o) {
  const uint32_t lo = uint32_t(__shfl_xor(int(uint32_t(v)), o)), hi = uint32_t(__shfl_xor(int(uint32_t(v >> 32)), o));
  return (static_cast<unsigned long long>(hi) << 32) | lo;
}

// One workgroup per ATTEMPT (blockIdx.x): a batch of attempts of consecutive draws is evaluated by one set of launches.
__global__ __launch_bounds__(1024) void k_select_triangle(const float4* __restrict__ p4, const uint32_t* __restrict__ draws_all,
                                                          float limit_sq, SelectRecord* rec_all) {
  __shared__ unsigned long long s_key[16];
  const uint32_t* draws = draws_all + size_t(blockIdx.x) * kSelectDraws;
  SelectRecord* rec = rec_all + blockIdx.x;
  const uint32_t t = threadIdx.x;
  unsigned long long key = 0;
  const float4 o = p4[draws[0]];
  if (t < uint32_t(kSelectTriangles)) {
    const float4 ps = p4[draws[1 + 2 * t]], pt = p4[draws[2 + 2 * t]];
    const float ux = ps.x - o.x, uy = ps.y - o.y, uz = ps.z - o.z;
    const float wx = pt.x - o.x, wy = pt.y - o.y, wz = pt.z - o.z;
    const float cx = uy * wz - uz * wy, cy = uz * wx - ux * wz, cz = ux * wy - uy * wx;
    const float wide = sqrtf(sqn3(cx, cy, cz));
    if (sqn3(ux, uy, uz) < limit_sq && sqn3(wx, wy, wz) < limit_sq && wide > 0.f)
      key = (static_cast<unsigned long long>(__float_as_uint(wide)) << 32) | (0xFFFFFFFFu - t);
  }
  for (int s = 32; s > 0; s >>= 1) { const unsigned long long k2 = shfl_xor_u64(key, s); key = k2 > key ? k2 : key; }
  if ((t & 63u) == 0) s_key[t >> 6] = key;
  __syncthreads();
  if (t != 0) return;
  for (int w = 1; w < 16; ++w) key = s_key[w] > key ? s_key[w] : key;
  rec->tri_key = key;
  rec->fourth_key = ~0ull;
  rec->ids[0] = rec->ids[1] = rec->ids[2] = rec->ids[3] = -1;
  rec->pa = rec->pb = rec->pc = 0.f;
  if (key == 0) { rec->status = kSelectNoTriangle; return; }
  const uint32_t win = 0xFFFFFFFFu - uint32_t(key);
  const uint32_t b1 = draws[0], b2 = draws[1 + 2 * win], b3 = draws[2 + 2 * win];
  rec->ids[0] = int32_t(b1); rec->ids[1] = int32_t(b2); rec->ids[2] = int32_t(b3);
  // plane through the three points, a x + b y + c z = 1, in double as match4pcsBase.cc:303-316 writes it
  const float4 A = p4[b1], B = p4[b2], Cc = p4[b3];
  const double x1 = A.x, y1 = A.y, z1 = A.z, x2 = B.x, y2 = B.y, z2 = B.z, x3 = Cc.x, y3 = Cc.y, z3 = Cc.z;
  const float denom = float(-x3 * y2 * z1 + x2 * y3 * z1 + x3 * y1 * z2 - x1 * y3 * z2 - x2 * y1 * z3 + x1 * y2 * z3);
  if (!(denom != 0)) { rec->status = kSelectDegenerate; return; }
  rec->pa = float((-y2 * z1 + y3 * z1 + y1 * z2 - y3 * z2 - y1 * z3 + y2 * z3) / denom);
  rec->pb = float((x2 * z1 - x3 * z1 - x1 * z2 + x3 * z2 + x1 * z3 - x2 * z3) / denom);
  rec->pc = float((-x2 * y1 + x3 * y1 + x1 * y2 - x3 * y2 - x1 * y3 + x2 * y3) / denom);
  rec->status = kSelectNoFourth;                // until k_select_fourth finds one
}

// One pass over the sampled P for ALL attempts of a batch: a thread holds a tile of points in registers and the best
// (distance, index) key of every attempt so far, so the 16 bytes of a point are read once per batch instead of once per
// attempt (at n_P = 4.2 M: 67 MB per scan).  A point further from an attempt's plane than the best admissible point the
// WORKGROUP has seen so far (s_bound, kept with LDS atomics) cannot become the minimum and skips the three sphere tests; equal
// distances still compete on the index (the reference keeps the first, match4pcsBase.cc:324-338).  Trip 0 takes one point per
// thread so that the bound exists after 1024 points; the other trips take tiles.  A workgroup reduces its keys in LDS and
// issues ONE atomicMin per attempt: same-address device atomics are served one after the other (~35 ns each on this part --
// with one per wave, 8192 of them, they WERE the kernel: ~300 us whatever the batch size, profiles/r04_select_probe*).
constexpr int kSelectTile = 4;
constexpr int kSelectThreads = 1024;
__global__ __launch_bounds__(kSelectThreads) void k_select_fourth(const float4* __restrict__ p4, uint32_t n_p, float too_small,
                                                                  SelectRecord* rec_all, int32_t n_attempts) {
  __shared__ float4 s_par[kSelectBatch][3];        // {pa, pb, pc, A.x} {A.y, A.z, B.x, B.y} {B.z, C.x, C.y, C.z}
  __shared__ uint32_t s_bound[kSelectBatch];       // bits of the smallest admissible distance any thread of the workgroup has seen
  __shared__ unsigned long long s_red[kSelectBatch][kSelectThreads / 64];
  __shared__ uint32_t s_live;                      // bit a: attempt a has a triangle and a plane and waits for its fourth point
  const uint32_t t = threadIdx.x;
  if (t == 0) s_live = 0u;
  if (t < uint32_t(kSelectBatch)) s_bound[t] = 0xFFFFFFFFu;
  __syncthreads();
  if (t < uint32_t(n_attempts) && rec_all[t].status == kSelectNoFourth) {
    const SelectRecord* rec = rec_all + t;
    const float4 A = p4[rec->ids[0]], B = p4[rec->ids[1]], Cc = p4[rec->ids[2]];
    s_par[t][0] = make_float4(rec->pa, rec->pb, rec->pc, A.x);
    s_par[t][1] = make_float4(A.y, A.z, B.x, B.y);
    s_par[t][2] = make_float4(B.z, Cc.x, Cc.y, Cc.z);
    atomicOr(&s_live, 1u << t);
  }
  __syncthreads();
  const uint32_t live = s_live;
  if (live == 0u) return;
  unsigned long long key[kSelectBatch];
#pragma unroll
  for (int a = 0; a < kSelectBatch; ++a) key[a] = ~0ull;
  const uint32_t nthreads = gridDim.x * blockDim.x, gid = blockIdx.x * blockDim.x + t;
  const uint32_t rest = n_p > nthreads ? n_p - nthreads : 0u;
  const uint32_t trips = 1u + (rest + uint32_t(kSelectTile) * nthreads - 1u) / (uint32_t(kSelectTile) * nthreads);
  const float not_a_point = __uint_as_float(0x7FC00000u);          // no point in this slot: the distance is a NaN and fails `d < FLT_MAX`
  for (uint32_t trip = 0; trip < trips; ++trip) {
    uint32_t idx[kSelectTile];
    float4 p[kSelectTile];
#pragma unroll
    for (int k = 0; k < kSelectTile; ++k)
      idx[k] = trip == 0u ? (k == 0 ? gid : 0xFFFFFFFFu) : nthreads + ((trip - 1u) * uint32_t(kSelectTile) + uint32_t(k)) * nthreads + gid;
#pragma unroll
    for (int k = 0; k < kSelectTile; ++k) p[k] = p4[min(idx[k], n_p - 1u)];      // (the loads of a tile in flight together)
#pragma unroll
    for (int k = 0; k < kSelectTile; ++k) p[k].x = idx[k] < n_p ? p[k].x : not_a_point;
    // the parameters and the bound of an attempt are re-read from LDS for every tile (3 ds_read_b128 + 1 ds_read_b32 against
    // the vector instructions of four points): hoisted out of the scan the parameters alone would be 192 registers
    uint32_t zero = 0u;
    asm volatile("" : "+v"(zero));
#pragma unroll
    for (int a = 0; a < kSelectBatch; ++a) {
      if (!((live >> a) & 1u)) continue;
      const float4* par = s_par[uint32_t(a) + zero];
      const float4 q0 = par[0], q1 = par[1], q2 = par[2];
      const uint32_t bound = s_bound[uint32_t(a) + zero];
#pragma unroll
      for (int k = 0; k < kSelectTile; ++k) {
        const float d = fabsf(((q0.x * p[k].x + q0.y * p[k].y) + q0.z * p[k].z) - 1.0f);
        // one rarely-taken branch per point and attempt: `d < FLT_MAX` drops infinities and NaNs (whose bit patterns would pass
        // the comparison with the initial bound), the bit comparison everything further from the plane than the bound
        if (__float_as_uint(d) <= bound && d < 3.402823466e+38f) {
          const bool far = sqn3(p[k].x - q0.w, p[k].y - q1.x, p[k].z - q1.y) >= too_small &&
                           sqn3(p[k].x - q1.z, p[k].y - q1.w, p[k].z - q2.x) >= too_small &&
                           sqn3(p[k].x - q2.y, p[k].y - q2.z, p[k].z - q2.w) >= too_small;
          const unsigned long long k2 = (static_cast<unsigned long long>(__float_as_uint(d)) << 32) | idx[k];
          if (far) {
            key[a] = k2 < key[a] ? k2 : key[a];
            atomicMin(&s_bound[a], __float_as_uint(d));
          }
        }
      }
    }
  }
  // workgroup minimum per attempt, one device atomic each (every lane of the workgroup is here: `live` and `trips` are uniform)
#pragma unroll
  for (int a = 0; a < kSelectBatch; ++a) {
    if (!((live >> a) & 1u)) continue;
    unsigned long long best = key[a];
    for (int s = 32; s > 0; s >>= 1) { const unsigned long long k2 = shfl_xor_u64(best, s); best = k2 < best ? k2 : best; }
    if ((t & 63u) == 0) s_red[a][t >> 6] = best;
  }
  __syncthreads();
  if (t < uint32_t(kSelectBatch) && ((live >> t) & 1u)) {
    unsigned long long best = ~0ull;
    for (uint32_t w = 0; w < blockDim.x / 64u; ++w) best = s_red[t][w] < best ? s_red[t][w] : best;
    if (best != ~0ull) atomicMin(&rec_all[t].fourth_key, best);
  }
}

__global__ void k_select_finish(const float4* __restrict__ p4, SelectRecord* rec_all) {
  SelectRecord* rec = rec_all + blockIdx.x;
  const uint32_t t = threadIdx.x;
  int32_t id = t < 3 ? rec->ids[t] : -1;
  if (t == 3 && rec->status == kSelectNoFourth && rec->fourth_key != ~0ull) id = int32_t(uint32_t(rec->fourth_key));
  if (t < 4) {
    const float4 p = id >= 0 ? p4[id] : make_float4(0.f, 0.f, 0.f, 0.f);
    rec->xyz[3 * t] = p.x; rec->xyz[3 * t + 1] = p.y; rec->xyz[3 * t + 2] = p.z;
  }
  if (t == 3 && id >= 0) { rec->ids[3] = id; rec->status = kSelectFound; }
}

__global__ void k_pack_points(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, uint32_t n, float4* __restrict__ out) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = make_float4(x[i], y[i], z[i], 0.f);
}

__global__ void k_selftest(const float* a, const float* b, uint64_t n, float* o_sqrt, float* o_div, float* o_ma) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const float x = a[i], y = b[i];
    o_sqrt[i] = sqrtf(fabsf(x));
    o_div[i] = x / y;
    o_ma[i] = x * y + (x * x + y * y);
  }
}

// Stage-level entry points start from explicitly cleared counters; on the fused path the last workgroup of k_verify
// leaves them cleared for the next base of the lane.
__global__ void k_reset_counters(DevCounters* c) {
  c->m1 = 0; c->m2 = 0; c->K = 0; c->C = 0; c->best_count = 0; c->overflow = 0;
  c->best_tag = ~0ull; c->has_best = 0; c->done = 0; c->quad_sum = 0; c->cand_sum = 0; c->n_border = 0; c->pruned = 0; c->S = 0;
  c->point_tests = 0; c->l0_pass = 0; c->l1_pass = 0; c->l2_pass = 0;
}

}  // namespace s4p
