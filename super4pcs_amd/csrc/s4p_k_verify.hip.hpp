// s4p_k_verify.hip.hpp -- part of the gfx950 device code (included by s4p_kernels.hip.hpp, in this order; one translation unit):
// k_sweep (counting first pass), k_verify (scoring + winner + result record), k_verify_T.
#pragma once

namespace s4p {

// ---------------------------------------------------------------------------
// k_sweep (round 6): the FIRST PASS of Verify when an early-exit bound is in force (VerifyParams::prune > 0: the trial loops).
// All but a few candidates of a base are dismissed by "how many queries COULD still be inliers": the number of sampled-Q points
// whose coarse cube (L0 bit) is marked under the candidate's transform is an upper bound of its inlier count, and a candidate
// whose bound does not EXCEED prune cannot become the best (match4pcsBase.hpp:468).  Until round 5 that count fell
// out of k_verify's lean sweep, which also queued every L0 survivor for the levels below and so carried a queue per wave (18 KB of
// LDS), scalar bookkeeping per chunk (SALU 0.6 x VALU), a candidate-record round trip per candidate -- and, for samples that do
// not fit LDS (n_Q > 2560: the 20 000-point sample), re-streamed the whole query array from L2 for EVERY candidate (4.6 TB/s of
// L2 reads at 14 M candidates/s).  This pass only COUNTS:
//   * one wave per BLOCK of kSweepCands = 4 candidates: lane l works on candidate (l >> 4) and query (l & 15) of a group of 16, and
//     the block's four transforms are applied to the group by ONE v_mfma_f32_16x16x4_f32 (rows = candidate x coordinate, columns =
//     queries, k = x, y, z, 1): the one place of the path where the arithmetic is a small dense contraction;
//   * the sampled Q goes through LDS in TILES of tile_q points (x | y | z floats, padded with far-away points), staged once per
//     workgroup and tile and swept by every wave for its block: query traffic / (waves x kSweepCands);
//   * per 64 (candidate, query) pairs: one LDS read, one MFMA, then the lean sweep's packed form -- v_cvt_pknorm_u16_f32,
//     v_pk_min_u16, the z floor and clamp, v_dot2_u32_u16, one LDS word, bit extract, ADD -- no ballot, no queue, no scalar work; one
//     row reduction (DPP) per candidate and tile;
//   * the bitmap is the pass's OWN when the workgroup has its CU to itself (samples beyond LDS, chunk passes): one bit per
//     2^sx x 2^sy x 2^sz cells, the finest shifts that fit 160 KB beside the tile (k_sweep_bitmap, planned by s4p_set_clouds) -- at
//     the bench clouds 2 x 1 x 1 cells against k_verify's 2 x 2 x 2: half the survivors;
//   * a candidate whose count + unswept queries <= prune is dead: its per-quad count reads 0 (a lower bound, as the
//     reference's is for what it abandons); the others are copied -- record and candidate index -- to the survivor list, one
//     global atomic per workgroup and base, and k_verify scores exactly those.
// Identical results by construction: the count is an upper bound of the inlier count under any locate inside the structure's slack,
// and k_verify applies its own bound again to what comes through.
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
  // the bitmap this pass indexes (shifted copy, k_coarse_shift): k_verify's coarse bitmap, or the finer one of k_sweep_bitmap when the
  // workgroup has its CU to itself; pitches px, py (border cube included), index of the border slab, cells -> cubes per axis (2^-s)
  const uint32_t* bm; uint32_t bm_words, px, py, mzc; float scx, scy, scz;
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
  uint32_t* s_coarse = s_mem;                               // LDS: coarse bitmap, shifted copy (address 0) | query tile x | y | z | 256 x 1.0f | SweepShared
  float* s_qx = reinterpret_cast<float*>(s_mem + P.bm_words);
  float* s_qy = s_qx + P.tile_q; float* s_qz = s_qy + P.tile_q;
  SweepShared& S = *reinterpret_cast<SweepShared*>(s_qz + P.tile_q + 256u);      // (256 floats of 1.0 lie between the tile and these scalars)
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
  { LcpGrid gb = P.grid; gb.coarse = P.bm; gb.coarse_words = P.bm_words; stage_coarse(gb, s_coarse); }      // ends with a workgroup barrier
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
  const uint32_t ucx = P.px, ucy = P.py, mz = P.mzc;
  const uint32_t lim_xy = (ucy << 16) | ucx, kdot = (ucx << 16) | 1u, cnxy = ucx * ucy;
  // (round 6) The block's FOUR transforms are applied by ONE matrix instruction per 16 queries: v_mfma_f32_16x16x4_f32 with
  //   A[i = 4 c + r][k] = row r of candidate c's locating transform (k = x, y, z, translation): lane l supplies A[l & 15][l >> 4],
  //   B[k][j]           = coordinate k of query j of the group (k = 3: 1.0):                       lane l supplies B[l >> 4][l & 15],
  //   D[4 c + r][j]     -> lane l holds rows 4 (l >> 4) .. + 3 of column l & 15: the cube coordinates of query (l & 15) under
  //                        candidate (l >> 4), in its own registers -- an fmaf chain per element, bit for bit.
  // One issue slot instead of the eighteen packed FMAs the four candidates took for the same 64 (candidate, query) pairs, and the
  // sweep is bound by vector issue (s4p_k_lcp.hip.hpp).  What follows a result is the lean sweep's packed form: rows 0 and 1 carry the
  // 1/65535 scale and the half-cube offset of v_cvt_pknorm_u16_f32 (floor + 1, clamped below, both axes packed), v_pk_min_u16 clamps
  // above onto the pitch, v_dot2_u32_u16 forms x' + pitch_x y' on the z term, and the bitmap in LDS is the copy shifted by pitch_x + 1
  // bits (SweepParams::grid.coarse).  Eleven issue slots per 64 pairs against 16.5.
  // The results of a batch of four MFMAs are read only after the NEXT batch has been issued (the matrix pipe works in order: they
  // are complete by then), never straight after their own issue -- this compiler leaves out the wait states between an MFMA and a
  // vector instruction that reads its result (measured with the 4x4x1 form: wrong counts until s_nop by hand).
  typedef float f4v_t __attribute__((ext_vector_type(4)));
  const uint32_t kk = lane >> 4, jj = lane & 15u;             // B: coordinate kk of the group's query jj; D: candidate kk of the block, query jj
  const uint32_t ar = lane & 3u, ac = (lane & 15u) >> 2;      // A: row ar of candidate ac, column kk
  const float kn = 1.0f / 65535.0f;
  const float a_org = kk == 3u ? (ar == 0u ? P.grid.ox : (ar == 1u ? P.grid.oy : P.grid.oz)) : 0.f;
  const float a_scl = ar == 0u ? P.scx * kn : (ar == 1u ? P.scy * kn : (ar == 2u ? P.scz : 0.f));
  const float a_half = (kk == 3u && ar < 2u) ? 0.5f * kn : 0.f;
  float* s_ones = s_qz + P.tile_q;                             // 256 x 1.0f behind the tile (the k = 3 lanes read their "coordinate" there)
  for (uint32_t w = threadIdx.x; w < 256u; w += blockDim.x) s_ones[w] = 1.0f;
  typedef __attribute__((address_space(3))) float* lds_f32_ptr;
  const uint32_t q_addr0 = kk < 3u ? uint32_t(uintptr_t((lds_f32_ptr)(s_qx + kk * P.tile_q + jj))) : uint32_t(uintptr_t((lds_f32_ptr)(s_ones + jj)));
  const uint32_t q_inc = kk < 3u ? 4u * kSweepStep : 0u;       // bytes per step of 256 queries
  __syncthreads();
  // every wave of the workgroup takes part in every ROUND (the tile staging is a workgroup affair); a round = kSweepCands tickets per wave
  const uint32_t per_round = n_waves * uint32_t(kSweepCands), rounds = (hi + per_round - 1u) / per_round;
  for (uint32_t r = 0; r < rounds; ++r) {
    const uint32_t t_first = (r * n_waves + wave) * uint32_t(kSweepCands);
    uint32_t bsel[kSweepCands], ci[kSweepCands], kq[kSweepCands], cnt[kSweepCands];
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
    // the block's records: the lane's A element (one dword of candidate ac's record) and every record's flag word, all in flight together
    float a_val;
    { const float* rec[kSweepCands];
#pragma unroll
      for (int k = 0; k < kSweepCands; ++k) rec[k] = reinterpret_cast<const float*>(P.b[valid[k] ? bsel[k] : 0u].cand_T + kCandStride * size_t(valid[k] ? ci[k] : 0u));
      const float* mine = ac == 0u ? rec[0] : (ac == 1u ? rec[1] : (ac == 2u ? rec[2] : rec[3]));
      const float raw = mine[min(ar, 2u) * 4u + kk];          // (row 3 of the 4 x 4 block does not exist: scaled by 0)
      uint32_t kraw[kSweepCands];
#pragma unroll
      for (int k = 0; k < kSweepCands; ++k) kraw[k] = __float_as_uint(rec[k][14]);      // {tag lo, tag hi, quad index | border flag, -}
      a_val = ((raw - a_org) * P.grid.inv_h) * a_scl + a_half;
#pragma unroll
      for (int k = 0; k < kSweepCands; ++k) {
        const uint32_t kr = uint32_t(__builtin_amdgcn_readfirstlane(int(kraw[k])));
        kq[k] = kr & ~kBorderFlag;
        // a candidate whose Euler-angle gate the host still has to settle goes to k_verify whatever its count: only there is it
        // entered into the list the host works through (it may turn out not to be a candidate at all)
        keep[k] = valid[k] && (kr & kBorderFlag) != 0u;
      } }
    for (uint32_t tile = 0; tile < P.n_tiles; ++tile) {     // uniform
      if (P.n_tiles > 1u) { __syncthreads(); stage_tile(tile); __syncthreads(); }
      const uint32_t swept_after = min((tile + 1u) * P.tile_q, P.n_q);
      bool any = false;
#pragma unroll
      for (int k = 0; k < kSweepCands; ++k) any = any || (alive[k] && !keep[k]);
      if (!any) continue;                                   // (uniform; the staging barriers above are passed by every wave)
      uint32_t hits = 0u, q_addr = q_addr0;
      auto load4 = [&](const uint32_t g0, float b[4]) {      // the B operands of groups g0 .. g0 + 3 of the current step
#pragma unroll
        for (uint32_t g = 0; g < 4u; ++g) b[g] = *(const __attribute__((address_space(3))) float*)(uintptr_t(q_addr + 64u * (g0 + g)));
      };
      auto mfma4 = [&](const float b[4], f4v_t d[4]) {
#pragma unroll
        for (uint32_t g = 0; g < 4u; ++g) d[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_val, b[g], f4v_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      };
      auto score4 = [&](const f4v_t d[4]) {
        uint32_t bb[4], ww[4];
        // (row 3 of a block is never read; without this the compiler hands its register to some temporary while the MFMA that
        // writes it is still in flight)
#pragma unroll
        for (uint32_t g = 0; g < 4u; ++g) asm volatile("" :: "v"(d[g][3]));
#pragma unroll
        for (uint32_t g = 0; g < 4u; ++g) {
          const uint32_t zc = min(uint32_t(floor_to_int(d[g][2])), mz);
          bb[g] = dot2_u16_s(pk_min_u16_s(cvt_pknorm_u16(d[g][0], d[g][1]), lim_xy), kdot, mul24_s(zc, cnxy));
        }
#pragma unroll
        for (uint32_t g = 0; g < 4u; ++g) ww[g] = lds_word(0u, bb[g] >> 5);              // (the bitmap starts at LDS address 0)
#pragma unroll
        for (uint32_t g = 0; g < 4u; ++g) hits += bfe1(ww[g], bb[g]);
      };
      f4v_t da[4], db[4];
      float ba[4], bq[4];
      load4(0u, ba); mfma4(ba, da);                         // prologue: batch 0 (64 queries = 4 groups of 16) of the first step
      for (uint32_t base = 0; base < P.tile_q; base += kSweepStep) {      // a step = 256 queries = four batches, A B A B
        load4(4u, bq);
        __builtin_amdgcn_sched_barrier(0);
        mfma4(bq, db);
        __builtin_amdgcn_sched_barrier(0);
        score4(da);
        __builtin_amdgcn_sched_barrier(0);
        load4(8u, ba);
        __builtin_amdgcn_sched_barrier(0);
        mfma4(ba, da);
        __builtin_amdgcn_sched_barrier(0);
        score4(db);
        __builtin_amdgcn_sched_barrier(0);
        load4(12u, bq);
        __builtin_amdgcn_sched_barrier(0);
        mfma4(bq, db);
        __builtin_amdgcn_sched_barrier(0);
        score4(da);
        __builtin_amdgcn_sched_barrier(0);
        q_addr += q_inc;
        if (base + kSweepStep < P.tile_q) {                 // (uniform) batch 0 of the next step
          load4(0u, ba);
          __builtin_amdgcn_sched_barrier(0);
          mfma4(ba, da);
        }
        __builtin_amdgcn_sched_barrier(0);
        score4(db);
        __builtin_amdgcn_sched_barrier(0);
      }
      // per-candidate totals of the tile: the sixteen lanes of a row -> its last lane (row_shr 1, 2, 4, 8 on the DPP pipe)
      uint32_t rs = hits;
      rs += uint32_t(__builtin_amdgcn_update_dpp(0, int(rs), 0x111, 0xf, 0xf, false));
      rs += uint32_t(__builtin_amdgcn_update_dpp(0, int(rs), 0x112, 0xf, 0xf, false));
      rs += uint32_t(__builtin_amdgcn_update_dpp(0, int(rs), 0x114, 0xf, 0xf, false));
      rs += uint32_t(__builtin_amdgcn_update_dpp(0, int(rs), 0x118, 0xf, 0xf, false));
#pragma unroll
      for (int k = 0; k < kSweepCands; ++k) {
        const uint32_t tot = uint32_t(__builtin_amdgcn_readlane(int(rs), 16 * k + 15));
        if (!alive[k] || keep[k]) continue;                 // (wave-uniform)
        cnt[k] += tot;
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
  const float* qtiles; uint32_t tile_q, n_tiles;        // ... and tile by tile, x | y | z per tile (k_sweep's view): the TILED form scores a sample beyond LDS through them
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
constexpr int kVerifyMaxTiles = 24;       // tiles of 2048 points: the largest sample the order keys allow (46 340 points) is 23
struct VerifyShared {                                   // k_verify's workgroup scalars, at the end of its dynamic LDS
  unsigned long long wtag[kGroupMax][kVerifyMaxThreads / 64];
  uint32_t wcnt[kGroupMax][kVerifyMaxThreads / 64], wcand[kGroupMax][kVerifyMaxThreads / 64];
  uint32_t pruned[kGroupMax];
  uint32_t end[kGroupMax];                              // end of base b's tickets in this workgroup's ticket space
  uint32_t next, last;
  uint16_t l0t[kVerifyMaxThreads / 64][kVerifyMaxTiles];      // (TILED) every wave's candidate: L0 survivors per tile of the sample
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
// TILED (round 6; with QLDS and LEAN): the survivors of k_sweep for a sample that does NOT fit LDS.  Until now their sweep read
// the queries from global memory -- 320 KB per candidate at the 20 000-point sample, every wave its own stream: 56 % of the GPU time
// of a base, bound by L2 bandwidth (~1 ms per candidate and wave).  Here the workgroup works in ROUNDS: every wave holds one
// candidate, the sample passes through LDS tile by tile (k_sweep's tiles) and every wave runs the lean sweep of an LDS-resident
// sample on its candidate and the current tile -- count, remember, fill, drain, exact batches -- with the bound "confirmed so far +
// this tile's pending + queries of the later tiles" (the single-pass bound, so the counts and the winner are the single pass's).
template <bool COUNT, bool QLDS, bool LEAN, bool TILED = false>
__global__ __launch_bounds__(kVerifyMaxThreads, 6) void k_verify(VerifyParams P) {   // <= 80 VGPRs: six waves per SIMD, i.e. two 768-thread workgroups per CU (of one launch, or of two)
  PROF_DECL;
  PROF_STAMP(0);
  extern __shared__ uint32_t s_mem[];
  uint32_t* s_coarse = s_mem;                              // LDS: coarse bitmap | quantised queries (QLDS) | 16 survivor queues
  uint2* s_q = reinterpret_cast<uint2*>(s_mem + P.grid.coarse_words);
  uint32_t* s_queue = reinterpret_cast<uint32_t*>(s_q + (QLDS ? ((P.n_q + kSweepStep - 1u) & ~(kSweepStep - 1u)) : 0u)) + (threadIdx.x >> 6) * kQueueWordsPerWave;
  const uint32_t n_pad = TILED ? P.tile_q : ((P.n_q + kSweepStep - 1u) & ~(kSweepStep - 1u));      // (TILED: the LDS copy is one tile)
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
#if defined(S4P_PROF)
    unsigned long long lpa_[kProfWords] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    K.lp = lpa_;
#endif
    if (LEAN) { if (QLDS && !TILED) stage_queries_f(P.qsoa, const_cast<float*>(LL.qx), 3u * n_pad); }
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
    // a scored candidate: its count into the per-quad array, into the border list (undecided gate) or into the wave's best of its base
    auto commit = [&](const uint32_t bsel, const uint32_t li, const float4 rr, const uint32_t cnt) {
      const VerifyBase& B = P.b[bsel];
      { // (round 6: the record's last row stays in registers across the sweep -- re-reading it was a global round trip per candidate)
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
    };
    if constexpr (TILED) {
      // ROUNDS: every wave holds one candidate of the workgroup's share, the tiles pass through LDS in lockstep, twice:
      //   pass A (all tiles): the candidate's L0 survivors per tile (the same sweep, counting only).  Their sum over the tiles NOT
      //     YET SCORED is what those can still add: against "every query of the later tiles" it dismisses a survivor of k_sweep
      //     after a tile or two of pass B instead of after nine of ten (measured with the weaker bound: ~430 us per survivor and
      //     wave, ten tiles of drains);
      //   pass B (until no wave of the workgroup has a live candidate): count, remember, fill, drain, exact batches on the tile,
      //     bound = confirmed + this tile's pending + pass A's counts of the tiles still to come.
      // Pass A is worth its sweeps only when the bound has teeth: against a best of a few per cent of the sample (the first bases of
      // a registration) nearly every candidate runs through all its tiles anyway, and "every query of the tiles to come" is used.
      // (A carousel -- every wave its own candidate and pass, synchronised only on the tile in LDS -- was measured slower: the cheap
      // pass-A steps of one wave wait for the drains of another's pass B at every tile.)
      const bool per_tile = P.n_tiles <= uint32_t(kVerifyMaxTiles) && P.prune * 10u >= P.n_q;
      auto stage_tile = [&](const uint32_t tile) {             // (all threads, between barriers)
        const float4* tsrc = reinterpret_cast<const float4*>(P.qtiles + size_t(tile) * 3u * P.tile_q);
        float4* tdst = reinterpret_cast<float4*>(const_cast<float*>(LL.qx));
        for (uint32_t w = threadIdx.x; w < (3u * P.tile_q) >> 2; w += blockDim.x) tdst[w] = tsrc[w];
      };
      auto queries_in = [&](const uint32_t tile) { const uint32_t first = tile * P.tile_q; return first < P.n_q ? min(P.tile_q, P.n_q - first) : 0u; };
      const uint32_t n_waves = blockDim.x >> 6, rounds = (hi + n_waves - 1u) / n_waves;
      const uint32_t wv = uint32_t(__builtin_amdgcn_readfirstlane(int(wave)));
      for (uint32_t r = 0; r < rounds; ++r) {                 // (uniform over the workgroup: the tile staging is a workgroup affair)
        const uint32_t t = r * n_waves + wv;
        const bool valid = t < hi;                            // (wave-uniform)
        uint32_t bsel = 0u, li = 0u;
        { uint32_t bs = 0u, t0 = 0u;
          const uint32_t tt = valid ? t : 0u;
#pragma unroll
          for (int b = 1; b < kGroupMax; ++b) { const uint32_t e = S.end[b - 1]; if (tt >= e) { bs = uint32_t(b); t0 = e; } }
          bsel = uint32_t(__builtin_amdgcn_readfirstlane(int(bs))); t0 = uint32_t(__builtin_amdgcn_readfirstlane(int(t0)));
          li = blockIdx.x + (tt - t0) * gridDim.x; }
        const VerifyBase& B = P.b[bsel];
        const float4* src = (P.use_surv ? B.surv_T : B.cand_T) + kCandStride * size_t(valid ? li : 0u);
        const float4 r0 = src[0], r1 = src[1], r2 = src[2], r3 = src[3];
        uint32_t later = 0u;
        if (per_tile) {
          for (uint32_t tile = 0; tile < P.n_tiles; ++tile) {   // uniform
            __syncthreads();
            stage_tile(tile);
            __syncthreads();
            const uint32_t in_tile = queries_in(tile);
            uint32_t h = 0u;
            if (valid && in_tile != 0u) {
              LcpTask Kc = K;
              Kc.n_q = in_tile; Kc.prune = 0u; Kc.pruned = nullptr; Kc.l0_only = 1u;
              h = wave_lcp_count_lean<COUNT, false, true>(P.grid, Kc, LL, src, r0, r1, r2);
            }
            S.l0t[wave][tile] = uint16_t(h);                    // (every lane, same value)
            later += h;
          }
        } else {
          later = P.n_q;
        }
        uint32_t cnt = 0u;
        bool alive = valid, was_abandoned = false;
        if (alive && later <= P.prune) { alive = false; was_abandoned = true; }      // (the coarse count alone: cannot exceed the best)
        for (uint32_t tile = 0; tile < P.n_tiles; ++tile) {   // uniform
          if (__syncthreads_or(alive ? 1 : 0) == 0) break;      // (uniform) nobody in the workgroup has a live candidate any more
          stage_tile(tile);
          __syncthreads();
          const uint32_t in_tile = queries_in(tile);
          if (alive && in_tile != 0u) {                       // (wave-uniform)
            later -= per_tile ? uint32_t(S.l0t[wave][tile]) : in_tile;
            LcpTask Kt = K;
            Kt.n_q = in_tile; Kt.point_tests = &B.ctr->point_tests; Kt.pruned = nullptr;
            // "cannot exceed the best" for this tile: its pending entries may still add at most prune - confirmed - (what the tiles
            // to come can add); while that is negative the candidate cannot be dismissed yet and the tile is simply counted
            const bool bounded = P.prune >= cnt + later;
            Kt.prune = bounded ? P.prune - cnt - later : 0u;
            bool dead = false;
            cnt += P.ablate == 1 ? wave_lcp_count_lean<COUNT, true, true>(P.grid, Kt, LL, src, r0, r1, r2, &dead)
                                 : wave_lcp_count_lean<COUNT, false, true>(P.grid, Kt, LL, src, r0, r1, r2, &dead);
            if (bounded && dead) { alive = false; was_abandoned = true; }
          }
        }
        if (valid) {
          if (was_abandoned && lane == 0) atomicAdd(&S.pruned[bsel], 1u);
          commit(bsel, li, r3, cnt);
        }
        __builtin_amdgcn_wave_barrier();
      }
    } else
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
      commit(bsel, li, r3, cnt);
      __builtin_amdgcn_wave_barrier();
#if defined(S4P_PROF)
      { __builtin_amdgcn_s_waitcnt(0); const unsigned long long d_ = __builtin_amdgcn_s_memrealtime() - tc0_;
        tp_[5] += 1; if (d_ > tp_[6]) tp_[6] = d_; if (d_ > 800ull) { tp_[7] += 1; tp_[8] += d_; } tp_[9] += d_; if (d_ > 2000ull) tp_[10] += 1; }
#endif
    }
#if defined(S4P_PROF)
    { const uint32_t wid_ = blockIdx.x * 16u + wave;
      if (lane == 0 && wid_ < uint32_t(kProfWaves)) for (int k_ = 0; k_ < kProfWords; ++k_) g_prof[3][wid_ * kProfWords + k_] = lpa_[k_]; }
#endif
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

}  // namespace s4p
