// s4p_kernels.hip.hpp -- gfx950 device code of the Super4PCS hot path (HIP, wave64).
//
// Compiled ONLY for gfx950 with -ffp-contract=off: every float expression below is
// evaluated as separate IEEE mul/add (no FMA), with correctly rounded sqrtf and '/',
// in the exact association order the reference's Eigen 3.3 fixed-size expressions
// use (see DESIGN.md "Numerics contract").  That is what makes integer inlier counts
// bit-exact against the CPU path.
//
// Reference functions restated here (paths under /root/reference/src/super4pcs/):
//   k_pairs        accelerators/pairExtraction/intersectionFunctor.h:197-233 (loop 2),
//                  intersectionPrimitive.h:117-157, algorithms/pairCreationFunctor.h:151-218
//   k_prep         algorithms/super4pcs.cc:118-146, accelerators/normalset.hpp:110-127,162-203
//   k_quads        algorithms/super4pcs.cc:151-163
//   k_verify       algorithms/match4pcsBase.cc:365-500 (ComputeRigidTransformation),
//                  match4pcsBase.cc:508-567 (Verify), accelerators/kdtree.h:417-421 (predicate)
//   k_apply        algorithms/match4pcsBase.hpp:265-267
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace s4p {

constexpr uint32_t kNil = 0xFFFFFFFFu;
constexpr uint32_t kGateFailed = 0xFFFFFFFFu;
constexpr int kMaskWords = 11;   // 343 direction buckets (7^3) -> 11 x 32 bit
constexpr int kMaxConeSamples = 56;

// ---------------------------------------------------------------------------
// exact-order float helpers (Eigen 3-vector reductions: x + (y + z))
// ---------------------------------------------------------------------------
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
  return ax * bx + (ay * by + az * bz);
}
__device__ __forceinline__ float sqn3(float x, float y, float z) { return x * x + (y * y + z * z); }
__device__ __forceinline__ void normalize3(float& x, float& y, float& z) {
  const float s2 = sqn3(x, y, z);
  if (s2 > 0.f) { const float s = sqrtf(s2); x /= s; y /= s; z /= s; }
}
__device__ __forceinline__ void cross3(float ax, float ay, float az, float bx, float by, float bz,
                                       float& ox, float& oy, float& oz) {
  ox = ay * bz - az * by;
  oy = az * bx - ax * bz;
  oz = ax * by - ay * bx;
}

// ---------------------------------------------------------------------------
// device-resident per-base counters / result
// ---------------------------------------------------------------------------
struct DevCounters {
  uint32_t m1, m2, K, C;            // appended pairs1 / pairs2 / quads, verified candidates
  uint32_t best_count;              // max inlier count (verified candidates only)
  uint32_t overflow;                // bit0 pairs1, bit1 pairs2, bit2 quads
  unsigned long long best_tag;      // min tag among candidates with best_count
  unsigned long long point_tests;   // optional instrumentation (COUNT kernels only)
  unsigned long long l0_pass, l1_pass, l2_pass;
  uint32_t cursor;                  // k_verify work cursor
  // winner record
  int32_t best_quad[4];
  float best_T[16];
  float best_c2[3];
  uint32_t has_best;
};

// ---------------------------------------------------------------------------
// LCP structure over sampled P  (replaces kd_tree_, match4pcsBase.cc:353-363).
// Uniform grid of edge h >= 1.002*delta, three levels, all conservative supersets of the
// exact predicate "some P point with fl(dx^2+(dy^2+dz^2)) <= fl(delta^2)" (kdtree.h:417-421):
//   L0  coarse bitmap (OR of 2^s-cubes of the reach bitmap), <= 48 KB, staged in LDS;
//   L1  reach bitmap: bit(c) = some P point lies within 1.01*delta of the box of cell c,
//       stored as {bits, rank-prefix} records (one 8 B load gives the bit and the rank);
//   L2  per reachable cell, a 16 B header {list start, count, 64-bit mask of the 4x4x4 sub-cells (edge h/4) that
//       some listed point can reach} and the contiguous list of exactly those P points (float4 copies): one
//       header load, a sub-cell bit test that drops most near-misses, then one 16 B load per exact distance test.
// The reach records and range table (~2 MB per 10^5 points) are L2-cache resident; the
// point lists (~400 B per P point) stream from Infinity Cache / HBM.
// ---------------------------------------------------------------------------
struct LcpGrid {
  const uint2* reach;           // per 32-cell word: {reach bits, number of reachable cells before this word}
  const uint4* list_hdr;        // per reachable cell: {first entry in nbr, entry count, 4x4x4 sub-cell reach mask lo, hi}
  const float4* nbr;            // P points (x,y,z,0) grouped by reachable cell
  const uint32_t* coarse;       // coarse bitmap (global copy, staged to LDS by the kernels)
  uint32_t coarse_words;
  int cshift, cnx, cny;
  float ox, oy, oz, inv_h;
  int nx, ny, nz;
  float sq_eps;                 // fl(delta*delta)
};

constexpr int kQueueEntries = 128;                 // per-wave survivor queue ({query, rank} per entry)
constexpr int kCoarseMaxWords = 12288;             // 48 KB

__device__ __forceinline__ bool cell_coords(const LcpGrid& g, float tx, float ty, float tz, int& ix, int& iy, int& iz) {
  const float fx = floorf((tx - g.ox) * g.inv_h);
  const float fy = floorf((ty - g.oy) * g.inv_h);
  const float fz = floorf((tz - g.oz) * g.inv_h);
  if (!(fx >= 0.f && fx < float(g.nx) && fy >= 0.f && fy < float(g.ny) && fz >= 0.f && fz < float(g.nz))) return false;
  ix = int(fx); iy = int(fy); iz = int(fz);
  return true;
}

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

// L2 + exact point tests for query i, whose cell is the `rank`-th reachable one (L0 and L1 already passed).
// The transformed point is recomputed here -- the same three expressions, hence the same bits -- rather than carried
// through the queue: 18 flops for the ~10 % of queries that get this far against two LDS words per queued query.
template <bool COUNT>
__device__ __forceinline__ bool fine_test(const LcpGrid& g, const float4* q4, const float* T, uint32_t i, uint32_t rank,
                                          unsigned long long* point_tests) {
  const uint4 hdr = g.list_hdr[rank];
  float tx, ty, tz;
  transform_point(T, q4[i], tx, ty, tz);
  // sub-cell of the query inside its cell (conservative: the mask was built with 1 % slack, rounding here is ~1e-5 cell)
  const float ux = (tx - g.ox) * g.inv_h, uy = (ty - g.oy) * g.inv_h, uz = (tz - g.oz) * g.inv_h;
  const float rx = ux - floorf(ux), ry = uy - floorf(uy), rz = uz - floorf(uz);
  const uint32_t sx = min(uint32_t(max(int(rx * 4.f), 0)), 3u), sy = min(uint32_t(max(int(ry * 4.f), 0)), 3u),
                 sz = min(uint32_t(max(int(rz * 4.f), 0)), 3u);
  const uint32_t sb = sz * 16u + sy * 4u + sx;
  const uint32_t mword = sb < 32u ? hdr.z : hdr.w;
  if (!((mword >> (sb & 31u)) & 1u)) return false;
  if (COUNT) atomicAdd(point_tests + 3, 1ull);     // l2_pass
  const uint32_t s = hdr.x, e = hdr.x + hdr.y;
  for (uint32_t p = s; p < e; p += 2) {                       // two independent 16 B loads per dependent step (four: slower)
    const float4 pa = g.nbr[p];
    const float4 pb = g.nbr[min(p + 1u, e - 1u)];            // (predicating this load away on odd tails was measured slower)
    if (COUNT) atomicAdd(point_tests, (p + 1u < e) ? 2ull : 1ull);
    const bool ha = sqn3(tx - pa.x, ty - pa.y, tz - pa.z) <= g.sq_eps;   // kdtree.h:417-421  sqdist <= cl_dist
    const bool hb = sqn3(tx - pb.x, ty - pb.y, tz - pb.z) <= g.sq_eps;
    if (ha | hb) return true;
  }
  return false;
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
// single-workgroup exclusive scan of n values (in place); *total = sum.  n up to a few million (one-off use).
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
__global__ __launch_bounds__(256) void k_grid_hdr_pack(GridBuildParams P, const uint32_t* list_start, const uint32_t* hdr_count, uint32_t n_reach) {
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n_reach; r += gridDim.x * blockDim.x) {
    P.list_hdr[r] = make_uint4(list_start[r], hdr_count[r], 0u, 0u);
    P.cursor[r] = 0u;
  }
}
__global__ __launch_bounds__(256) void k_grid_fill(GridBuildParams P) {
  const uint64_t total = uint64_t(P.n_p) * 27u;
  for (uint64_t t = blockIdx.x * uint64_t(blockDim.x) + threadIdx.x; t < total; t += uint64_t(gridDim.x) * blockDim.x) {
    uint32_t cell;
    const uint32_t i = uint32_t(t / 27u);
    if (grid_incidence(P, i, int(t % 27u), cell)) {
      const uint2 w = P.reach[cell >> 5];
      const uint32_t rank = w.y + uint32_t(__popc(w.x & ((1u << (cell & 31u)) - 1u)));
      const uint32_t at = P.list_hdr[rank].x + atomicAdd(&P.cursor[rank], 1u);
      P.nbr[at] = make_float4(P.px[i], P.py[i], P.pz[i], 0.f);
    }
  }
}

// Fills hdr.z/.w, the 4x4x4 sub-cell reach masks, from the point lists.
// bit(sx,sy,sz) = some listed point lies within `reach` of the sub-box; double precision, same slack as the lists.
struct MaskParams {
  uint4* list_hdr; const float4* nbr; const uint32_t* cell_id; uint32_t n_reach;
  float ox, oy, oz, h; int nx, ny; double reach2;
};
__global__ __launch_bounds__(256) void k_build_masks(MaskParams P) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= P.n_reach) return;
  uint4 hdr = P.list_hdr[r];
  const uint32_t c = P.cell_id[r];
  const int ix = int(c % uint32_t(P.nx)), iy = int((c / uint32_t(P.nx)) % uint32_t(P.ny)), iz = int(c / (uint32_t(P.nx) * uint32_t(P.ny)));
  const double q = double(P.h) * 0.25;
  const double bx = double(P.ox) + double(ix) * double(P.h), by = double(P.oy) + double(iy) * double(P.h), bz = double(P.oz) + double(iz) * double(P.h);
  unsigned long long mask = 0ull;
  for (uint32_t p = hdr.x; p < hdr.x + hdr.y; ++p) {
    const float4 pp = P.nbr[p];
    for (int s = 0; s < 64; ++s) {
      const double lo[3] = {bx + (s & 3) * q, by + ((s >> 2) & 3) * q, bz + (s >> 4) * q};
      const double v[3] = {double(pp.x), double(pp.y), double(pp.z)};
      double d2 = 0;
      for (int k = 0; k < 3; ++k) { const double d = v[k] < lo[k] ? lo[k] - v[k] : (v[k] > lo[k] + q ? v[k] - (lo[k] + q) : 0.0); d2 += d * d; }
      if (d2 <= P.reach2) mask |= (1ull << s);
    }
  }
  hdr.z = uint32_t(mask); hdr.w = uint32_t(mask >> 32);
  P.list_hdr[r] = hdr;
}

// Number of sampled-Q points that T brings within delta of a sampled-P point: Verify()
// (match4pcsBase.cc:508-567) without the early exit, for one wave64.
//   s_coarse : LDS copy of the coarse bitmap (workgroup-shared)
//   s_queue  : this wave's private LDS queue (kQueueEntries entries of {query index, reachable-cell rank})
// Phase 1 (every query, four 64-query chunks per step): transform, cell, L0 test out of LDS; the L0 survivors then
// read their reach word (L1, one 8-byte gather, issued for all four chunks back to back so the round trips overlap)
// and the queries whose cell is reachable are compacted into the queue with a ballot/prefix.
// Phase 2 (64 queued queries at a time, one per lane): list header + sub-cell mask (L2), then the exact tests.
// Doing L1 in phase 1 means phase 2 runs on ~10 % of the queries instead of the ~25 % that pass L0, and its
// dependent chain is header -> points instead of reach word -> header -> points.
template <bool COUNT, bool SKIP_FINE = false>
__device__ __forceinline__ uint32_t wave_lcp_count(const LcpGrid& g, const uint32_t* s_coarse, uint2* s_queue,
                                                   const float4* q4, uint32_t n_q, const float* T,
                                                   unsigned long long* point_tests) {
  constexpr uint32_t kNone = 0xFFFFFFFFu;
  const uint32_t lane = threadIdx.x & 63u;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  uint32_t cnt = 0, qn = 0;
  const uint32_t cmax = g.coarse_words * 32u - 1u;
  const uint32_t unx = uint32_t(g.nx), uny = uint32_t(g.ny), unz = uint32_t(g.nz);
  const bool dims24 = unx < (1u << 24) && uny * unz < (1u << 24);      // uniform: 24-bit multiplies are full rate
  // cell of query i under T, or kNone if it falls outside the grid or into a coarse cube nothing can reach
  auto locate = [&](const float4 q, const uint32_t i) -> uint32_t {
    float tx, ty, tz;
    transform_point(T, q, tx, ty, tz);
    // float -> int conversion saturates and one unsigned compare per axis covers both bounds (a NaN coordinate maps
    // to cell 0 and then fails every exact distance test, so it cannot create an inlier)
    const int ix = int(floorf((tx - g.ox) * g.inv_h)), iy = int(floorf((ty - g.oy) * g.inv_h)), iz = int(floorf((tz - g.oz) * g.inv_h));
    const bool inb = (uint32_t(ix) < unx) & (uint32_t(iy) < uny) & (uint32_t(iz) < unz) & (i < n_q);
    const uint32_t cc = min(__umul24(__umul24(uint32_t(iz) >> g.cshift, uint32_t(g.cny)) + (uint32_t(iy) >> g.cshift), uint32_t(g.cnx)) +
                            (uint32_t(ix) >> g.cshift), cmax);
    const bool surv = inb & (((s_coarse[cc >> 5] >> (cc & 31u)) & 1u) != 0u);
    const uint32_t c = dims24 ? __umul24(__umul24(uint32_t(iz), uny) + uint32_t(iy), unx) + uint32_t(ix)
                              : (uint32_t(iz) * uny + uint32_t(iy)) * unx + uint32_t(ix);
    return surv ? c : kNone;
  };
  auto fine = [&](const uint2 e) -> uint32_t {
    if (SKIP_FINE) return uint32_t(e.y == kNone);
    return fine_test<COUNT>(g, q4, T, e.x, e.y, point_tests) ? 1u : 0u;
  };
  // L1 for one chunk + compaction; drains 64 queued queries when that many are waiting
  auto push = [&](const uint32_t c, const uint2 w, const uint32_t i) {
    const uint32_t sh = c & 31u;
    const bool reach = (c != kNone) & (((w.x >> sh) & 1u) != 0u);
    const unsigned long long m = __ballot(reach);
    const unsigned long long m0 = COUNT ? __ballot(c != kNone) : 0ull;
    if (COUNT && lane == 0) {
      atomicAdd(point_tests + 1, (unsigned long long)__popcll(m0));                      // l0_pass
      atomicAdd(point_tests + 2, (unsigned long long)__popcll(m));                       // l1_pass
    }
    if (reach) s_queue[qn + uint32_t(__popcll(m & lt_mask))] = make_uint2(i, w.y + uint32_t(__popc(w.x & ((1u << sh) - 1u))));
    qn += uint32_t(__popcll(m));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (qn >= 64u) {
      cnt += fine(s_queue[qn - 64u + lane]);
      qn -= 64u;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  };
  const uint32_t last = n_q - 1u;
  for (uint32_t base = 0; base < n_q; base += 256) {
    const uint32_t i0 = base + lane, i1 = i0 + 64u, i2 = i0 + 128u, i3 = i0 + 192u;
    const float4 q0 = q4[min(i0, last)];
    const float4 q1 = q4[min(i1, last)];
    const float4 q2 = q4[min(i2, last)];
    const float4 q3 = q4[min(i3, last)];
    const uint32_t c0 = locate(q0, i0), c1 = locate(q1, i1), c2 = locate(q2, i2), c3 = locate(q3, i3);
    // reach words of the L0 survivors; rejected lanes read word 0 (one broadcast line)
    const uint2 w0 = g.reach[c0 == kNone ? 0u : c0 >> 5];
    const uint2 w1 = g.reach[c1 == kNone ? 0u : c1 >> 5];
    const uint2 w2 = g.reach[c2 == kNone ? 0u : c2 >> 5];
    const uint2 w3 = g.reach[c3 == kNone ? 0u : c3 >> 5];
    push(c0, w0, i0); push(c1, w1, i1); push(c2, w2, i2); push(c3, w3, i3);
  }
  if (lane < qn) cnt += fine(s_queue[lane]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
  __builtin_amdgcn_wave_barrier();
  return cnt;
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
// ComputeRigidTransformation (match4pcsBase.cc:365-500), computeScale == false,
// max_angle < 0 (the Euler-angle gate is rejected at s4p_create).
// Returns true iff "ok && rms >= 0 && rms < 2*delta" (match4pcsBase.hpp:436-439).
// T is row-major 3x4 (R | t).
// ---------------------------------------------------------------------------
struct BaseFrame {          // per-base constants, computed once on the host side of the ABI
  float p[3][3];            // first three base points (sampled P, centred)
  float c1[3];              // centroid1 = ((b1+b2)+b3)/3
  float gate;               // distance_factor * delta = 2*delta
};

__device__ __forceinline__ bool gs_frame(const float* a0, const float* a1, const float* a2, float e[3][3]) {
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

__device__ __forceinline__ bool rigid_gate(const BaseFrame& b, const float q[3][3], float T[12], float c2[3]) {
  for (int k = 0; k < 3; ++k) c2[k] = ((q[0][k] + q[1][k]) + q[2][k]) / 3.f;    // match4pcsBase.hpp:415-417
  float vp[3][3], vq[3][3];
  if (!gs_frame(b.p[0], b.p[1], b.p[2], vp)) return false;   // rms = 1e9 -> gate fails (quirk .cc:417-433)
  if (!gs_frame(q[0], q[1], q[2], vq)) return false;
  float R[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) R[r][c] = vp[0][r] * vq[0][c] + (vp[1][r] * vq[1][c] + vp[2][r] * vq[2][c]);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float dg = R[i][0] * R[0][i] + (R[i][1] * R[1][i] + R[i][2] * R[2][i]);   // (R*R).diagonal() .cc:453
    if (dg - 1.f > 1e-6f) return false;
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
  if (!(rms >= 0.f && rms < b.gate)) return false;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float rc = R[r][0] * (-c2[0]) + (R[r][1] * (-c2[1]) + R[r][2] * (-c2[2]));
    T[r * 4 + 0] = R[r][0]; T[r * 4 + 1] = R[r][1]; T[r * 4 + 2] = R[r][2];
    T[r * 4 + 3] = b.c1[r] + rc;
  }
  return true;
}

// ---------------------------------------------------------------------------
// k_gate / k_verify parameters.  k_verify: persistent 1024-thread workgroups, one wave64 per gated candidate; the
// lengths of the quad list and of the gated list live in device memory (no host round trip).
// LDS per workgroup: coarse bitmap (<= 48 KB) + 16 private survivor queues (1 KB each).
// ---------------------------------------------------------------------------
constexpr int kVerifyThreads = 1024;
struct VerifyParams {
  LcpGrid grid;
  const float4* q4;                                     // sampled Q (centred), packed (x,y,z,0), original order (quad indices)
  const float4* q4v;                                    // the same points in Morton order, for the LCP sweep
  uint32_t n_q;
  BaseFrame base;
  const int4* quads; const unsigned long long* tags; uint32_t* counts;
  const uint32_t* K_dev; uint32_t K_cap;
  uint32_t* cand_idx; float4* cand_T;                   // gated candidates: quad index + 3x4 transform
  DevCounters* ctr;
  int ablate;                                           // S4P_ABLATE debugging only (0 = full kernel)
};

// k_gate: one thread per congruent quad: ComputeRigidTransformation + rms gate
// (match4pcsBase.cc:365-500, match4pcsBase.hpp:436-439).  Passing candidates are compacted
// (wave-aggregated append) into cand_idx / cand_T so that the scoring kernel sees a dense,
// perfectly balanceable list; failing ones get counts[k] = kGateFailed.
__global__ __launch_bounds__(256) void k_gate(VerifyParams P) {
  const uint32_t K = min(*P.K_dev, P.K_cap);
  const uint32_t lane = threadIdx.x & 63u;
  for (uint32_t k0 = (blockIdx.x * blockDim.x + threadIdx.x) & ~63u; k0 < K; k0 += gridDim.x * blockDim.x) {
    const uint32_t k = k0 + lane;
    float T[12]; float c2[3];
    bool ok = false;
    if (k < K) {
      const int4 qd = P.quads[k];
      const float4 a = P.q4[qd.x], b = P.q4[qd.y], c = P.q4[qd.z];
      float q[3][3] = {{a.x, a.y, a.z}, {b.x, b.y, b.z}, {c.x, c.y, c.z}};
      ok = rigid_gate(P.base, q, T, c2);
      if (!ok) P.counts[k] = kGateFailed;
    }
    const unsigned long long pass = __ballot(ok);
    if (pass == 0ull) continue;
    const uint32_t leader = __ffsll((long long)pass) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(&P.ctr->C, uint32_t(__popcll(pass)));
    base = __shfl(base, leader);
    if (ok) {
      const uint32_t at = base + uint32_t(__popcll(pass & ((1ull << lane) - 1ull)));
      P.cand_idx[at] = k;
      float4* dst = P.cand_T + 3 * size_t(at);
      dst[0] = make_float4(T[0], T[1], T[2], T[3]);
      dst[1] = make_float4(T[4], T[5], T[6], T[7]);
      dst[2] = make_float4(T[8], T[9], T[10], T[11]);
    }
  }
}

// k_verify: Verify() (match4pcsBase.cc:508-567, no early exit) of every gated candidate.
// Persistent 1024-thread workgroups, each working through its slice of the gated candidate list (the list's
// length lives in device memory: no host round trip).
// LDS: coarse bitmap (<= 48 KB) + 16 private survivor queues (1 KB each).
template <bool COUNT>
__global__ __launch_bounds__(kVerifyThreads, 8) void k_verify(VerifyParams P) {   // 8 waves/SIMD: two 1024-thread workgroups per CU
  extern __shared__ uint32_t s_mem[];
  uint32_t* s_coarse = s_mem;
  uint2* s_queue = reinterpret_cast<uint2*>(s_mem + P.grid.coarse_words) + (threadIdx.x >> 6) * kQueueEntries;
  const uint32_t C = P.ctr->C;
  // Work split: every workgroup owns a contiguous slice of the gated candidate list (static: a single-address global
  // cursor caps at ~90 dequeues/us, MI355X_MICROARCH "dequeue"); inside the slice its 16 waves take candidates from an
  // LDS counter, so a wave that drew cheap candidates (few L0 survivors) simply takes more.  Slices differ by at most
  // one candidate, against "one or two candidates per wave" for a static stride over waves.
  const uint32_t lo = uint32_t((uint64_t(C) * blockIdx.x) / gridDim.x), hi = uint32_t((uint64_t(C) * (blockIdx.x + 1u)) / gridDim.x);
  if (lo >= hi) return;                                    // more workgroups than candidates (uniform)
  __shared__ uint32_t s_next;
  if (threadIdx.x == 0) s_next = lo;
  stage_coarse(P.grid, s_coarse);                          // ends with a workgroup barrier
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t local_best = 0;
  bool any = false;
  while (true) {
    uint32_t i = 0;
    if (lane == 0) i = atomicAdd(&s_next, 1u);
    i = uint32_t(__builtin_amdgcn_readfirstlane(int(i)));
    if (i >= hi) break;
    const float4* src = P.cand_T + 3 * size_t(i);
    const float4 r0 = src[0], r1 = src[1], r2 = src[2];
    // one candidate per wave: its 3x4 lives in scalar registers
    const float T[12] = {wave_uniform(r0.x), wave_uniform(r0.y), wave_uniform(r0.z), wave_uniform(r0.w),
                         wave_uniform(r1.x), wave_uniform(r1.y), wave_uniform(r1.z), wave_uniform(r1.w),
                         wave_uniform(r2.x), wave_uniform(r2.y), wave_uniform(r2.z), wave_uniform(r2.w)};
    uint32_t cnt = 0;
    if (P.ablate == 2) cnt = uint32_t(T[3] > 1e30f);
    else if (P.ablate == 1) cnt = wave_lcp_count<COUNT, true>(P.grid, s_coarse, s_queue, P.q4v, P.n_q, T, &P.ctr->point_tests);
    else cnt = wave_lcp_count<COUNT, false>(P.grid, s_coarse, s_queue, P.q4v, P.n_q, T, &P.ctr->point_tests);
    if (lane == 0) P.counts[P.cand_idx[i]] = cnt;
    local_best = max(local_best, cnt);
    any = true;
  }
  // One global atomic per workgroup, and only if it can raise the maximum: a per-wave atomicMax on the single
  // result word serialised at ~90 ops/us (8192 waves = 0.09 ms per launch, measured with S4P_ABLATE=2).
  __shared__ uint32_t s_best;
  if (threadIdx.x == 0) s_best = 0;
  __syncthreads();
  if (lane == 0 && any) atomicMax(&s_best, local_best);
  __syncthreads();
  if (threadIdx.x == 0 && s_best > 0) {
    const uint32_t cur = __hip_atomic_load(&P.ctr->best_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (s_best > cur) atomicMax(&P.ctr->best_count, s_best);
  }
}

// k_select: best_tag = min tag among candidates whose count == best_count (first in
// reference order wins: match4pcsBase.hpp:468 strict '>').
struct SelectParams {
  const unsigned long long* tags; const uint32_t* counts; const int4* quads;
  const uint32_t* K_dev; uint32_t K_cap; DevCounters* ctr;
  const float4* q4; BaseFrame base;
};
__global__ __launch_bounds__(256) void k_select(SelectParams P) {
  const uint32_t K = min(*P.K_dev, P.K_cap);
  if (P.ctr->C == 0) return;
  const uint32_t best = P.ctr->best_count;
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < K; k += gridDim.x * blockDim.x)
    if (P.counts[k] == best) atomicMin(&P.ctr->best_tag, P.tags[k]);
}
__global__ __launch_bounds__(256) void k_winner(SelectParams P) {
  const uint32_t K = min(*P.K_dev, P.K_cap);
  if (P.ctr->C == 0) return;
  const unsigned long long bt = P.ctr->best_tag;
  const uint32_t best = P.ctr->best_count;
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < K; k += gridDim.x * blockDim.x) {
    if (P.tags[k] == bt && P.counts[k] == best) {
      const int4 qd = P.quads[k];
      const float4 a = P.q4[qd.x], b = P.q4[qd.y], c = P.q4[qd.z];
      float q[3][3] = {{a.x, a.y, a.z}, {b.x, b.y, b.z}, {c.x, c.y, c.z}};
      float T[12], c2[3];
      rigid_gate(P.base, q, T, c2);
      for (int i = 0; i < 12; ++i) P.ctr->best_T[i] = T[i];
      P.ctr->best_T[12] = 0.f; P.ctr->best_T[13] = 0.f; P.ctr->best_T[14] = 0.f; P.ctr->best_T[15] = 1.f;
      for (int i = 0; i < 3; ++i) P.ctr->best_c2[i] = c2[i];
      P.ctr->best_quad[0] = qd.x; P.ctr->best_quad[1] = qd.y; P.ctr->best_quad[2] = qd.z; P.ctr->best_quad[3] = qd.w;
      P.ctr->has_best = 1u;
    }
  }
}

// k_verify_T: Verify() for explicit transforms (one wave per transform).
struct VerifyTParams {
  LcpGrid grid; const float4* q4; uint32_t n_q;
  const float* T; uint32_t B; uint32_t* counts; DevCounters* ctr;
};
__global__ __launch_bounds__(kVerifyThreads) void k_verify_T(VerifyTParams P) {
  extern __shared__ uint32_t s_mem[];
  uint32_t* s_coarse = s_mem;
  uint2* s_queue = reinterpret_cast<uint2*>(s_mem + P.grid.coarse_words) + (threadIdx.x >> 6) * kQueueEntries;
  stage_coarse(P.grid, s_coarse);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t k = wave; k < P.B; k += nwaves) {
    float T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = wave_uniform(P.T[16 * size_t(k) + i]);
    const uint32_t cnt = wave_lcp_count<false, false>(P.grid, s_coarse, s_queue, P.q4, P.n_q, T, nullptr);
    if (lane == 0) P.counts[k] = cnt;
  }
}

// ---------------------------------------------------------------------------
// k_pairs: loop 2 of IntersectionFunctor::process + PairCreationFunctor::process.
// One thread per (primitive pId, sequence slot s); the sequence is the concatenation
// of the leaf ranges of the host-built octree, so (pId, s) order == reference order.
// Accepted (i=pId, j) appends (j,i) then (i,j) with order keys 2*(pId*n_seq+s)+{0,1}.
// ---------------------------------------------------------------------------
struct PairParams {
  const float* ux; const float* uy; const float* uz;     // unit-cube coordinates of sampled Q
  const float* qx; const float* qy; const float* qz;     // world (centred) coordinates
  const float* nx; const float* ny; const float* nz;     // normals or nullptr
  const float* cr; const float* cg; const float* cb;     // rgb or nullptr
  const uint32_t* seq_id; const uint32_t* seq_leaf; uint32_t n_seq;
  const float4* leaves;                                    // (cx, cy, cz, halfEdge argument)
  uint32_t n_q;
  float nRadius, eps_unit;
  double pair_distance, pair_distance_eps, pair_normals_angle;
  float max_normal_difference, max_color_distance, max_translation_distance, norm_threshold;
  float b1pos[3], b2pos[3], b1rgb[3], b2rgb[3];
  int2* ab; uint32_t* okey; uint32_t* counter; uint32_t cap; uint32_t* overflow; uint32_t overflow_bit;
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

constexpr int kPairStage = 2048;   // accepted (pId, j) per workgroup between two flushes

// One workgroup per primitive pId; threads sweep the sequence in chunks of 256.  Accepted slots are
// staged in LDS and flushed with ONE global atomic per workgroup (a per-wave atomic on the single
// output cursor serialises at ~90 ops/us and dominated the first version of this kernel).
// The two pair sets of a base are independent: one launch, blockIdx.y picks the set (gridDim.y = 1 for a single set).
struct PairParams2 { PairParams set[2]; };
__global__ __launch_bounds__(256) void k_pairs(PairParams2 PP) {
  const PairParams& P = PP.set[blockIdx.y];
  __shared__ uint32_t st_j[kPairStage];
  __shared__ uint32_t st_s[kPairStage];
  __shared__ uint32_t st_p[kPairStage];
  __shared__ uint32_t st_n, st_base;
  if (threadIdx.x == 0) st_n = 0;
  __syncthreads();
  auto flush = [&]() {     // called by all threads, between barriers
    const uint32_t n = st_n;
    if (n == 0) return;
    if (threadIdx.x == 0) st_base = atomicAdd(P.counter, 2u * n);
    __syncthreads();
    const uint32_t base = st_base;
    for (uint32_t e = threadIdx.x; e < n; e += blockDim.x) {
      const uint32_t at = base + 2u * e;
      if (at + 1u < P.cap) {
        const uint32_t j = st_j[e], pId = st_p[e];
        const uint32_t ok = 2u * (pId * P.n_seq + st_s[e]);
        P.ab[at] = make_int2(int(j), int(pId));     P.okey[at] = ok;          // pairs->emplace_back(j, i)  :214
        P.ab[at + 1] = make_int2(int(pId), int(j)); P.okey[at + 1] = ok + 1u; // pairs->emplace_back(i, j)  :215
      } else {
        atomicOr(P.overflow, P.overflow_bit);
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) st_n = 0;
    __syncthreads();
  };
  // each workgroup takes primitives blockIdx.x, blockIdx.x + gridDim.x, ... so the number of global atomics is
  // ~ the number of workgroups (a few hundred), not the number of primitives
  for (uint32_t pId = blockIdx.x; pId < P.n_q; pId += gridDim.x) {
  const float cx = P.ux[pId], cy = P.uy[pId], cz = P.uz[pId];
  const float wxi = P.qx[pId], wyi = P.qy[pId], wzi = P.qz[pId];
  for (uint32_t s0 = 0; s0 < P.n_seq; s0 += blockDim.x) {
    const uint32_t s = s0 + threadIdx.x;
    bool acc = false;
    uint32_t j = 0;
    if (s < P.n_seq) {
      j = P.seq_id[s];
      if (pId > j) {
        const float dx = P.ux[j] - cx, dy = P.uy[j] - cy, dz = P.uz[j] - cz;
        const float t = sqrtf(sqn3(dx, dy, dz)) - P.nRadius;
        if (t * t < P.eps_unit * P.eps_unit) {                               // intersectPoint, intersectionPrimitive.h:154-157
          if (sphere_box(cx, cy, cz, P.nRadius, P.leaves[P.seq_leaf[s]])) {   // intersect, :117-142
            // PairCreationFunctor::process(i = pId, j): p = Q[j], q = Q[i]
            const float wx = wxi - P.qx[j], wy = wyi - P.qy[j], wz = wzi - P.qz[j];
            const float distance = sqrtf(sqn3(wx, wy, wz));
            acc = !(fabs(double(distance) - P.pair_distance) > P.pair_distance_eps);   // :162
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
                  sqrtf(sqn3(P.qx[j] - P.b1pos[0], P.qy[j] - P.b1pos[1], P.qz[j] - P.b1pos[2])) < P.max_translation_distance &&
                  sqrtf(sqn3(wxi - P.b2pos[0], wyi - P.b2pos[1], wzi - P.b2pos[2])) < P.max_translation_distance;
              if (!good) acc = false;
            }
          }
        }
      }
    }
    if (acc) {
      const uint32_t e = atomicAdd(&st_n, 1u);     // LDS atomic; room for 256 more is guaranteed by the flush rule below
      st_j[e] = j; st_s[e] = s; st_p[e] = pId;
    }
    __syncthreads();
    if (st_n + blockDim.x > uint32_t(kPairStage)) flush();   // uniform: st_n is read after the barrier
  }
  }
  flush();
}

// ---------------------------------------------------------------------------
// Congruent-quad enumeration (FindCongruentQuadrilaterals, super4pcs.cc:80-177).
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
  uint32_t mask;               // size - 1 (power of two)
  uint32_t epoch;
};

struct PrepParams {
  const float* ux; const float* uy; const float* uz;
  const float* qx; const float* qy; const float* qz;
  const int2* ab; const uint32_t* m_dev; uint32_t cap;
  float invariant;
  QuadGrid qg;
  uint32_t* cell; uint32_t* bucket; float4* ew; uint32_t* next;   // set 1: bucket,next used; set 2: unused
  uint32_t* mask;                                                  // set 2: kMaskWords per entry
  HashTable ht;
  ConeTable cone;
};

__device__ __forceinline__ void prep1_body(const PrepParams& P) {
  const uint32_t m = min(*P.m_dev, P.cap);
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < m; e += gridDim.x * blockDim.x) {
    const int2 ab = P.ab[e];
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
    uint32_t h = hash_cell(cell) & P.ht.mask;
    while (true) {
      const unsigned long long k = __hip_atomic_load(&P.ht.keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (k == mykey) break;
      if (uint32_t(k >> 32) != P.ht.epoch) {
        const unsigned long long old = atomicCAS(&P.ht.keys[h], k, mykey);
        if (old == k || old == mykey) break;
        continue;   // somebody claimed it for another cell: re-read the same slot
      }
      h = (h + 1u) & P.ht.mask;
    }
    const unsigned long long prev = atomicExch(&P.ht.heads[h], ((unsigned long long)P.ht.epoch << 32) | e);
    P.next[e] = (uint32_t(prev >> 32) == P.ht.epoch) ? uint32_t(prev) : kNil;
  }
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

__device__ __forceinline__ void prep2_body(const PrepParams& P, uint32_t* smask) {
  const uint32_t m = min(*P.m_dev, P.cap);
  const uint32_t nthreads = gridDim.x * blockDim.x;
  uint32_t* my = smask + threadIdx.x * kMaskWords;
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < m; e += nthreads) {
    const int2 ab = P.ab[e];
    const float p1x = P.ux[ab.x], p1y = P.uy[ab.x], p1z = P.uz[ab.x];
    const float p2x = P.ux[ab.y], p2y = P.uy[ab.y], p2z = P.uz[ab.y];
    const float nx = p2x - p1x, ny = p2y - p1y, nz = p2z - p1z;
    const float qx_ = p1x + P.invariant * nx, qy_ = p1y + P.invariant * ny, qz_ = p1z + P.invariant * nz;   // super4pcs.cc:141
    P.cell[e] = index_pos(qx_, qy_, qz_, P.qg);
    const float w1x = P.qx[ab.x], w1y = P.qy[ab.x], w1z = P.qz[ab.x];
    const float w2x = P.qx[ab.y], w2y = P.qy[ab.y], w2z = P.qz[ab.y];
    P.ew[e] = make_float4(w1x + P.invariant * (w2x - w1x), w1y + P.invariant * (w2y - w1y),
                          w1z + P.invariant * (w2z - w1z), 0.f);                                          // :142
    float q[4];
    float qnx = nx, qny = ny, qnz = nz;
    normalize3(qnx, qny, qnz);                          // queryn = (p2-p1).normalized()            super4pcs.cc:144
    quat_from_z_to(qnx, qny, qnz, q);                   // setFromTwoVectors normalises it again    normalset.hpp:181
#pragma unroll
    for (int w = 0; w < kMaskWords; ++w) my[w] = 0u;
    for (int a = 0; a < P.cone.nb; ++a) {               // normalset.hpp:186-196
      const float vx = P.cone.v[a][0], vy = P.cone.v[a][1], vz = P.cone.v[a][2];
      float ux_, uy_, uz_;
      cross3(q[1], q[2], q[3], vx, vy, vz, ux_, uy_, uz_);            // QuaternionBase::_transformVector
      ux_ += ux_; uy_ += uy_; uz_ += uz_;
      float cx, cy, cz;
      cross3(q[1], q[2], q[3], ux_, uy_, uz_, cx, cy, cz);
      float dx = (vx + q[0] * ux_) + cx, dy = (vy + q[0] * uy_) + cy, dz = (vz + q[0] * uz_) + cz;
      normalize3(dx, dy, dz);
      const uint32_t id = index_normal(dx, dy, dz, P.qg.nepsilon);
      if (id < 343u) my[id >> 5] |= (1u << (id & 31u));
    }
#pragma unroll
    for (int w = 0; w < kMaskWords; ++w) P.mask[size_t(e) * kMaskWords + w] = my[w];
  }
}
// Set 1 (hash insert) and set 2 (cone masks) are prepared by one launch: blockIdx.y == 0 -> set 1, 1 -> set 2.
__global__ __launch_bounds__(256) void k_prep(PrepParams P1, PrepParams P2) {
  __shared__ uint32_t smask[256 * kMaskWords];
  if (blockIdx.y == 0) prep1_body(P1);
  else prep2_body(P2, smask);
}

struct QuadParams {
  // set 1
  const int2* ab1; const uint32_t* okey1; const uint32_t* bucket1; const float4* ew1; const uint32_t* next1;
  // set 2
  const int2* ab2; const uint32_t* okey2; const uint32_t* cell2; const float4* ew2; const uint32_t* mask2;
  const uint32_t* m2_dev; uint32_t cap2;
  HashTable ht;
  float thr;                   // distance_threshold2 (compared against a SQUARED norm: quirk super4pcs.cc:160)
  int4* quads; unsigned long long* tags; uint32_t* K_dev; uint32_t K_cap; uint32_t* overflow;
};

constexpr int kQuadStage = 1536;   // quads per workgroup between two flushes (24 B each)

// One thread per pairs2 entry: hash lookup of its euclidean cell, walk of the set-1 chain.
// Matches are staged in LDS and flushed with one global atomic per workgroup round.
__global__ __launch_bounds__(256) void k_quads(QuadParams P) {
  __shared__ int4 st_q[kQuadStage];
  __shared__ unsigned long long st_t[kQuadStage];
  __shared__ uint32_t st_n, st_base;
  const uint32_t m2 = min(*P.m2_dev, P.cap2);
  if (threadIdx.x == 0) st_n = 0;
  __syncthreads();
  for (uint32_t i0 = blockIdx.x * blockDim.x; i0 < m2; i0 += gridDim.x * blockDim.x) {
    const uint32_t i = i0 + threadIdx.x;
    if (i < m2) {
      const uint32_t cell = P.cell2[i];
      const unsigned long long mykey = ((unsigned long long)P.ht.epoch << 32) | cell;
      uint32_t h = hash_cell(cell) & P.ht.mask;
      uint32_t e = kNil;
      while (true) {
        const unsigned long long k = P.ht.keys[h];
        if (k == mykey) { const unsigned long long hd = P.ht.heads[h]; e = (uint32_t(hd >> 32) == P.ht.epoch) ? uint32_t(hd) : kNil; break; }
        if (uint32_t(k >> 32) != P.ht.epoch) break;
        h = (h + 1u) & P.ht.mask;
      }
      if (e != kNil) {
        const float4 eq = P.ew2[i];
        const uint32_t* mk = P.mask2 + size_t(i) * kMaskWords;
        const int2 ab2 = P.ab2[i];
        const uint32_t ok2 = P.okey2[i];
        while (e != kNil) {
          const uint32_t b = P.bucket1[e];
          if ((mk[b >> 5] >> (b & 31u)) & 1u) {
            const float4 ep = P.ew1[e];
            const float dx = eq.x - ep.x, dy = eq.y - ep.y, dz = eq.z - ep.z;
            if (sqn3(dx, dy, dz) <= P.thr) {                                           // super4pcs.cc:160
              const int2 ab1 = P.ab1[e];
              const int4 quad = make_int4(ab1.x, ab1.y, ab2.x, ab2.y);                 // :171-172
              const unsigned long long tag = ((unsigned long long)P.okey1[e] << 32) | ok2;
              const uint32_t slot = atomicAdd(&st_n, 1u);
              if (slot < uint32_t(kQuadStage)) { st_q[slot] = quad; st_t[slot] = tag; }
              else {                                                                    // stage full: direct append
                const uint32_t at = atomicAdd(P.K_dev, 1u);
                if (at < P.K_cap) { P.quads[at] = quad; P.tags[at] = tag; } else atomicOr(P.overflow, 4u);
              }
            }
          }
          e = P.next1[e];
        }
      }
    }
    __syncthreads();
    const uint32_t n = min(st_n, uint32_t(kQuadStage));
    if (n) {
      if (threadIdx.x == 0) st_base = atomicAdd(P.K_dev, n);
      __syncthreads();
      const uint32_t base = st_base;
      for (uint32_t e = threadIdx.x; e < n; e += blockDim.x) {
        const uint32_t at = base + e;
        if (at < P.K_cap) { P.quads[at] = st_q[e]; P.tags[at] = st_t[e]; } else atomicOr(P.overflow, 4u);
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) st_n = 0;
    __syncthreads();
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

__global__ void k_selftest(const float* a, const float* b, uint64_t n, float* o_sqrt, float* o_div, float* o_ma) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const float x = a[i], y = b[i];
    o_sqrt[i] = sqrtf(fabsf(x));
    o_div[i] = x / y;
    o_ma[i] = x * y + (x * x + y * y);
  }
}

__global__ void k_reset_counters(DevCounters* c) {
  c->m1 = 0; c->m2 = 0; c->K = 0; c->C = 0; c->best_count = 0; c->overflow = 0;
  c->best_tag = ~0ull; c->has_best = 0; c->cursor = 0;
}

}  // namespace s4p
