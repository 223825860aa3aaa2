// s4p_k_lcp.hip.hpp -- part of the gfx950 device code (included by s4p_kernels.hip.hpp, in this order; one translation unit):
// LCP scoring of one candidate by one wave: the fused / staged sweeps (full counts) and the lean sweep (an early-exit bound in force), LDS staging.
#pragma once

namespace s4p {

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
  uint32_t l0_only = 0u;             // (lean sweep of an LDS-resident sample) 1: return the number of L0 survivors of the sweep and stop there
#if defined(S4P_PROF)
  unsigned long long* lp;            // lab build: the calling wave's phase sums of the lean sweep (kProfWords words, in registers)
#endif
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
__device__ __forceinline__ uint32_t wave_lcp_count_lean(const LcpGrid& g, const LcpTask& K, const LeanLds& L, const float4* Tsrc, const float4 t0, const float4 t1, const float4 t2, bool* dead_out = nullptr) {
  // (dead_out: whether the candidate was abandoned -- the tiled form of k_verify calls this once per tile of the sample and keeps the books itself)
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
  float Tpre[12];                                          // (the lane-parallel path below requests the candidate's rows before it fills the queue)
  auto drain = [&](const uint32_t rest, auto have_rows) -> bool {
    lds_fence();
    float T[12];
    if constexpr (decltype(have_rows)::value) {
#pragma unroll
      for (int i = 0; i < 12; ++i) T[i] = Tpre[i];
    } else load_rows(Tsrc, T);
    const GridXf X = make_grid_xf(g, T, 1.f);
    uint32_t rd = nb;
    const uint32_t end = nb + na;
    bool dead = false;
    while (rd < end) {                                     // wave-uniform
      uint32_t ii[kSweepChunks], cc[kSweepChunks]; bool vv[kSweepChunks]; uint2 ww[kSweepChunks];
#pragma unroll
      for (uint32_t k = 0; k < kSweepChunks; ++k) { const uint32_t at = rd + 64u * k + lane; vv[k] = at < end; ii[k] = uint32_t(q[min(at, end - 1u)]); }
      lds_fence();                                         // every lane holds its entries before any slot of this round is rewritten
      const uint32_t n_round = min(end - rd, kSweepStep);
#pragma unroll
      for (uint32_t k = 0; k < kSweepChunks; ++k) {
        cc[k] = 0u; ww[k] = make_uint2(0u, 0u);
        if (k != 0u && 64u * k >= n_round) continue;        // (uniform) a round is rarely full: most candidates bring ~150 entries
        int ix, iy, iz;
        grid_cell(X.u, lean_query<QL>(K, L, ii[k]), ix, iy, iz);
        vv[k] = vv[k] & (uint32_t(ix) < uint32_t(g.nx)) & (uint32_t(iy) < uint32_t(g.ny)) & (uint32_t(iz) < uint32_t(g.nz));
        cc[k] = mad24(mad24(uint32_t(iz), uint32_t(g.ny), uint32_t(iy)), uint32_t(g.nx), uint32_t(ix));
        ww[k] = g.reach[vv[k] ? cc[k] >> 5 : 0u];
      }
#pragma unroll
      for (uint32_t k = 0; k < kSweepChunks; ++k) {
        if (k != 0u && 64u * k >= n_round) continue;        // (uniform; vv[k] is false in every lane)
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
#if defined(S4P_PROF)
  unsigned long long lp_[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // [0] start [1] sweep end [2] queue filled [3] drain time [4] exact time [5] exact batches [6] sweep steps [7] alive after the sweep
  PROF_NOW(lp_[0]);
#endif
  // drain + exact batches once the queue cannot take another step (or the candidate's last entries are in): `rest` = what has not
  // been queued yet and may still count.  Returns true if the candidate is abandoned.
  auto settle = [&](const bool more, const uint32_t rest, auto have_rows) -> bool {
#if defined(S4P_PROF)
    unsigned long long sa_, sb_; PROF_NOW(sa_);
    const bool dd_ = drain(rest, have_rows);
    PROF_NOW(sb_); lp_[3] += sb_ - sa_;
    if (dd_) return true;
#else
    if (drain(rest, have_rows)) return true;
#endif
    while ((more && nb + 2u * kSweepStep > kLeanQueue) || (!more && nb != 0u)) {
      if (cnt + nb + rest <= K.prune) return true;
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
#if defined(S4P_PROF)
      lp_[5] += 1;
#endif
    }
#if defined(S4P_PROF)
    { unsigned long long sc_; PROF_NOW(sc_); lp_[4] += sc_ - sb_; }
#endif
    return false;
  };
  if constexpr (QL) {
    // ---- round 6: the sweep of a sample that lives in LDS COUNTS and REMEMBERS, it does not queue ----
    // With three waves per SIMD the sweep is bound by VECTOR ISSUE: a wave issues one instruction per four cycles, a SIMD one
    // vector instruction per four cycles whichever of its waves it comes from, and a step of 256 queries was ~90 vector
    // instructions (measured per candidate with the lab build's stamps: 3.4 us for its eight steps = ~1000 cycles per step and
    // wave; more waves per CU change nothing: profiles/r06_lab).  So the sweep is priced in vector instructions per 64 queries:
    // ~22 until round 5, ~18 here.
    //   * x and y of two queries leave the packed FMAs already scaled by 1/65535 and offset by half a cube, so ONE
    //     v_cvt_pknorm_u16_f32 per query rounds (to nearest = floor of the cube coordinate + 1), clamps below at 0 and packs both
    //     axes (measured on the device: RNE of the exact product, NaN and negatives -> 0, +inf -> 65535);
    //     ONE v_pk_min_u16 clamps both above, onto the pitch (the bitmap's empty border cube); ONE v_dot2_u32_u16 forms
    //     x' + pitch_x * y' on top of the z term.  (Before: three floor-converts, three clamps, two multiply-adds.)
    //     The bitmap this indexes is the copy SHIFTED by pitch_x + 1 bits behind a few zero words (LcpGrid::coarse of the lean
    //     launches, built once per cloud by k_coarse_shift): cube -1 of a row is the border cube of the row before it, row -1
    //     of a slab the border row of the slab before it, and in front of the first slab lie the zero words.
    //   * z keeps the floor-convert + unsigned clamp onto the border slab (a packed form needs pitch_x * pitch_y < 65536).
    //   * an L0 survivor costs its lane one v_lshl_or_b32 -- the lane's 64-bit shift register holds one bit per chunk of the
    //     sample -- and the wave one compare for the running total; no prefix sum, no LDS write, no scalar bookkeeping per chunk.
    //     The queue is filled from the shift registers only for the candidates that are still alive after the whole sweep (one
    //     in four on the bench workload), with the tighter bound "survivors not queued yet" in place of "queries not swept yet"
    //     for everything that follows.
    // The locate may move a query by ~1e-5 cube against grid_cell2 (coefficients rounded after the scaling by 1/65535); the
    // structure absorbs 1e-2 cell at every level (LcpGridHost::plan), and the exact stage below re-locates in fine units.
    // (The same transform on the matrix pipe -- v_mfma_f32_4x4x1_16b_f32, lane = query, thirteen MFMAs per step for the eighteen
    // packed FMAs -- was measured again on this form: no faster, MFMAs take the same issue slots; and this compiler leaves out
    // the wait states between such an MFMA and a vector instruction that reads its result: wrong counts until s_nop by hand.)
    static_assert(kLeanMaxQueries <= 64 * 64, "a 64-bit shift register per lane: at most 64 chunks");
    const uint32_t lim_xy = (ucy << 16) | ucx, kdot = (ucx << 16) | 1u, cnxy = ucx * ucy;
    float un[8];
    { const float kn = 1.0f / 65535.0f;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        un[4 * r + 0] = Xc.u[4 * r + 0] * kn; un[4 * r + 1] = Xc.u[4 * r + 1] * kn; un[4 * r + 2] = Xc.u[4 * r + 2] * kn;
        un[4 * r + 3] = (Xc.u[4 * r + 3] + 0.5f) * kn;
      } }
    auto axis2 = [&](const float a, const float b, const float c, const float d, const v2f_t x, const v2f_t y, const v2f_t z) -> v2f_t {
      const v2f_t ab = {a, b}, cd = {c, d};
      return pk_fma_lo(ab, x, pk_fma_hi(ab, y, pk_fma_lo_hi(cd, z)));       // fma(a, x, fma(b, y, fma(c, z, d))) for both queries
    };
    uint32_t acc0 = 0u, acc1 = 0u, total = 0u;
    for (uint32_t base = 0; base < n_pad; base += kSweepStep) {          // wave-uniform
      const uint32_t unswept = uint32_t(max(int(K.n_q) - int(base + kSweepStep), 0));
      float x[kSweepChunks], y[kSweepChunks], z[kSweepChunks];
      uint32_t bb[kSweepChunks], ww[kSweepChunks];
#pragma unroll
      for (uint32_t k = 0; k < kSweepChunks; ++k) { const uint32_t i = base + 64u * k + lane; x[k] = L.qx[i]; y[k] = L.qy[i]; z[k] = L.qz[i]; }
#pragma unroll
      for (uint32_t k = 0; k < kSweepChunks; k += 2u) {
        const v2f_t vx = {x[k], x[k + 1u]}, vy = {y[k], y[k + 1u]}, vz = {z[k], z[k + 1u]};
        const v2f_t px = axis2(un[0], un[1], un[2], un[3], vx, vy, vz), py = axis2(un[4], un[5], un[6], un[7], vx, vy, vz);
        const v2f_t pz = axis2(Xc.u[8], Xc.u[9], Xc.u[10], Xc.u[11], vx, vy, vz);
        const uint32_t z0 = min(uint32_t(floor_to_int(pz.x)), mz), z1 = min(uint32_t(floor_to_int(pz.y)), mz);
        bb[k] = dot2_u16_s(pk_min_u16_s(cvt_pknorm_u16(px.x, py.x), lim_xy), kdot, mul24_s(z0, cnxy));
        bb[k + 1u] = dot2_u16_s(pk_min_u16_s(cvt_pknorm_u16(px.y, py.y), lim_xy), kdot, mul24_s(z1, cnxy));
      }
      __builtin_amdgcn_sched_barrier(0);                                 // (the step's four bitmap words are requested together ...
#pragma unroll
      for (uint32_t k = 0; k < kSweepChunks; ++k) ww[k] = lds_word(coarse_base, bb[k] >> 5);
      __builtin_amdgcn_sched_barrier(0);                                 // ... before the first is consumed: one exposure per step, not four)
      uint32_t bits = 0u;
#pragma unroll
      for (uint32_t k = 0; k < kSweepChunks; ++k) {
        const uint32_t t = bfe1(ww[k], bb[k]);
        total += uint32_t(__popcll(__builtin_amdgcn_ballot_w64(t != 0u)));
        bits = k == 0u ? t : ((bits << 1) | t);
      }
      // the lane's 64-bit shift register {acc1, acc0}: after the sweep, chunk c of n sits at bit n - 1 - c
      acc1 = __builtin_amdgcn_alignbit(acc1, acc0, 32u - kSweepChunks);
      acc0 = (acc0 << kSweepChunks) | bits;
#if defined(S4P_PROF)
      lp_[6] += 1;
#endif
      if (total + unswept <= K.prune) { abandoned = true; break; }          // cannot exceed the bound any more: no global access made
    }
#if defined(S4P_PROF)
    PROF_NOW(lp_[1]); lp_[2] = lp_[1]; lp_[7] = abandoned ? 0 : 1;
#endif
    const uint32_t n_chunks = n_pad >> 6;
    if (K.l0_only != 0u) {                                             // (uniform) the tiled form's first look at a tile: its L0 survivors, nothing else
      if (dead_out != nullptr) *dead_out = false;
      return total;
    }
    if (!abandoned && total <= kLeanQueue) {
      // The survivors' query indices -> the queue, every lane its own (one candidate in four gets here -- the coarse level alone
      // dismisses only the far-off ones -- so this is priced like the sweep: a chunk-by-chunk expansion with a ballot per chunk
      // cost ~250 vector instructions and took back what the sweep had gained).  Lane l writes its popcount(acc) entries behind
      // those of the lanes below it (inclusive scan on the DPP pipe: row_shr 1, 2, 4, 8 inside the rows of 16, then row_bcast 15
      // and 31 across them), one entry per trip of a loop that runs for the busiest lane's count: ~6 trips.
      load_rows(Tsrc, Tpre);                                           // (the drain's transform: its round trip runs beside the queue fill)
      const uint32_t pc = uint32_t(__popc(acc0)) + uint32_t(__popc(acc1));
      uint32_t incl = pc;
      incl += uint32_t(__builtin_amdgcn_update_dpp(0, int(incl), 0x111, 0xf, 0xf, false));
      incl += uint32_t(__builtin_amdgcn_update_dpp(0, int(incl), 0x112, 0xf, 0xf, false));
      incl += uint32_t(__builtin_amdgcn_update_dpp(0, int(incl), 0x114, 0xf, 0xf, false));
      incl += uint32_t(__builtin_amdgcn_update_dpp(0, int(incl), 0x118, 0xf, 0xf, false));
      incl += uint32_t(__builtin_amdgcn_update_dpp(0, int(incl), 0x142, 0xa, 0xf, false));
      incl += uint32_t(__builtin_amdgcn_update_dpp(0, int(incl), 0x143, 0xc, 0xf, false));
      uint16_t* qw = q + (incl - pc);
      const uint32_t top = (n_chunks - 1u) * 64u + lane;               // bit p of {acc1, acc0} is chunk n_chunks - 1 - p: query top - 64 p
      uint32_t w = acc0;
      while (w != 0u) { const uint32_t b = uint32_t(__builtin_ctz(w)); w &= w - 1u; *qw++ = uint16_t(top - 64u * b); }
      w = acc1;
      while (w != 0u) { const uint32_t b = uint32_t(__builtin_ctz(w)) + 32u; w &= w - 1u; *qw++ = uint16_t(top - 64u * b); }
      na = total;
#if defined(S4P_PROF)
      PROF_NOW(lp_[2]);
#endif
      if (settle(false, 0u, std::true_type{})) abandoned = true;
    } else if (!abandoned) {
      // (more L0 survivors than the queue takes: a candidate that really aligns the clouds)  chunk by chunk, a step's worth at a
      // time, settled whenever the queue cannot take another step
      uint32_t rest = total;                                           // L0 survivors not queued yet
      for (uint32_t c0 = 0;; c0 += kSweepChunks) {
        const bool more = c0 < n_chunks;                               // wave-uniform
        if (more) {
#pragma unroll
          for (uint32_t k = 0; k < kSweepChunks; ++k) {
            const uint32_t ch = c0 + k;                                // (uniform; n_chunks is a multiple of kSweepChunks)
            const uint32_t pos = n_chunks - 1u - ch, word = pos < 32u ? acc0 : acc1;
            const uint32_t t = (word >> (pos & 31u)) & 1u;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(t != 0u);
            if (m != 0ull) {
              uint16_t* qw = q + (nb + na);
              if (t != 0u) qw[__builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u))] = uint16_t(ch * 64u + lane);
              const uint32_t n = uint32_t(__popcll(m));
              na += n; rest -= n;
            }
          }
          if (cnt + nb + na + rest <= K.prune) { abandoned = true; break; }
        }
        if (!more || nb + na + kSweepStep > kLeanQueue) {
          if (settle(more, more ? rest : 0u, std::false_type{})) { abandoned = true; break; }
        }
        if (!more) break;
      }
    }
  } else {
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
      if (settle(more, more ? unswept : 0u, std::false_type{})) { abandoned = true; break; }
    }
    if (!more || abandoned) break;
  }
  }
#if defined(S4P_PROF)
  if (QL) {
    unsigned long long e_; PROF_NOW(e_);
    K.lp[0] += 1; K.lp[1] += lp_[1] - lp_[0]; K.lp[2] += lp_[6]; K.lp[3] += lp_[7]; K.lp[4] += lp_[2] - lp_[1];
    K.lp[5] += lp_[3]; K.lp[6] += lp_[4]; K.lp[7] += lp_[5]; K.lp[8] += e_ - lp_[0]; K.lp[9] += lp_[5] ? 1ull : 0ull; K.lp[10] += abandoned ? 0ull : 1ull; K.lp[11] += cnt;
  }
#endif
  if (dead_out != nullptr) *dead_out = abandoned;
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

}  // namespace s4p
