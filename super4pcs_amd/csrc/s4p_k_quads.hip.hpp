// s4p_k_quads.hip.hpp -- part of the gfx950 device code (included by s4p_kernels.hip.hpp, in this order; one translation unit):
// the gate of one quad, k_gate, k_quads (enumeration + gate).
#pragma once

namespace s4p {

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
  __shared__ uint32_t s_pair_i[kQuadThreads];                // the tile's pairs whose cell holds set-1 pairs, compacted: pair slot -> index of the pair
  __shared__ uint32_t s_item_p[kQuadThreads * kSubChains], s_item_e[kQuadThreads * kSubChains];      // work items: (pair slot, head of ONE of its cell's chains)
  __shared__ float4 s_eq[kQuadThreads];                      // per pair slot: the world point of the pair's invariant point,
  __shared__ int2 s_ab2[kQuadThreads];                       // ... its two point ids
  __shared__ uint32_t s_ok2[kQuadThreads], s_pc[kQuadWaves]; // ... its order key; pairs with a chain per wave (s_ic: items per wave)
  __shared__ uint32_t s_mask[kQuadThreads * kMaskWords];   // each pair slot's direction mask (row stride 11: conflict-free)
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
    // phase A, one thread per set-2 pair of the tile: invariant point -> cell -> the heads of the cell's set-1 chains (super4pcs.cc:141,
    // normalset.hpp:162-171).  Typically well under half of the pairs fall into a cell that holds a set-1 pair; those are
    // compacted (pair slots), and every non-empty chain of their cells becomes a WORK ITEM of its own (round 6: kSubChains chains
    // per cell, so that a cell with fifty pairs is four walks of a dozen hops on four threads instead of one of fifty).
    uint32_t n_pairs = 0, n_items = 0;
    {
      const uint32_t i = i0 + threadIdx.x;
      uint32_t he[kSubChains], cnt = 0u;
#pragma unroll
      for (uint32_t sc = 0; sc < kSubChains; ++sc) he[sc] = kNil;
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
          if (k == mykey) {
            const ulonglong2* hp = reinterpret_cast<const ulonglong2*>(P.ht.heads + size_t(h) * kSubChains);      // (32 bytes: one segment)
            const ulonglong2 h01 = hp[0], h23 = hp[1];
            const unsigned long long hd[kSubChains] = {h01.x, h01.y, h23.x, h23.y};
#pragma unroll
            for (uint32_t sc = 0; sc < kSubChains; ++sc)
              if (uint32_t(hd[sc] >> 32) == P.ht.epoch) { he[sc] = uint32_t(hd[sc]); ++cnt; }
            break;
          }
          if (uint32_t(k >> 32) != P.ht.epoch) break;
          h = (h + 1u) & hmask;
        }
      }
      const unsigned long long m = __ballot(cnt != 0u);
      const uint32_t incl = wave_incl_scan_u32(cnt);
      if (lane == 63u) s_ic[wave] = incl;
      if (lane == 0) s_pc[wave] = uint32_t(__popcll(m));
      __syncthreads();
      uint32_t pbefore = 0, ibefore = 0;
      for (uint32_t w = 0; w < wave; ++w) { pbefore += s_pc[w]; ibefore += s_ic[w]; }
      if (cnt != 0u) {
        const uint32_t slot = pbefore + __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));
        s_pair_i[slot] = i;
        uint32_t at = ibefore + incl - cnt;
#pragma unroll
        for (uint32_t sc = 0; sc < kSubChains; ++sc)
          if (he[sc] != kNil) { s_item_p[at] = slot; s_item_e[at] = he[sc]; ++at; }
      }
      __syncthreads();
#pragma unroll
      for (int w = 0; w < kQuadWaves; ++w) { n_pairs += s_pc[w]; n_items += s_ic[w]; }
    }
    PROF_NOW(pb_);
    uint32_t hops_ = 0; (void)hops_;
    // phase B, per pair with a chain: world point (super4pcs.cc:142) and the cone mask of its direction (normalset.hpp:174-196) into
    // the pair slot's LDS row -- what a separate preparation launch used to do for EVERY pair.  The 56 cone samples of a pair are
    // the phase's cost (~10 us for one thread) and the pairs with a chain rarely fill the workgroup: a pair gets 1, 2 or 4 threads
    // (uniform per tile), each a share of the samples, all setting bits of the same row by LDS atomics.
    for (uint32_t w = threadIdx.x; w < n_pairs * uint32_t(kMaskWords); w += blockDim.x) s_mask[w] = 0u;
    __syncthreads();
    {
      const uint32_t tpp = n_pairs * 4u <= blockDim.x ? 4u : (n_pairs * 2u <= blockDim.x ? 2u : 1u);      // threads per pair
      const uint32_t ps = threadIdx.x / tpp, part = threadIdx.x % tpp;
      if (ps < n_pairs) {
        const uint32_t i = s_pair_i[ps];
        const int2 ab2 = P.ab2[i];
        const float p1x = P.ux[ab2.x], p1y = P.uy[ab2.x], p1z = P.uz[ab2.x];
        const float p2x = P.ux[ab2.y], p2y = P.uy[ab2.y], p2z = P.uz[ab2.y];
        if (part == 0u) {
          const float w1x = P.qx[ab2.x], w1y = P.qy[ab2.x], w1z = P.qz[ab2.x];
          const float w2x = P.qx[ab2.y], w2y = P.qy[ab2.y], w2z = P.qz[ab2.y];
          s_eq[ps] = make_float4(w1x + P.invariant2 * (w2x - w1x), w1y + P.invariant2 * (w2y - w1y), w1z + P.invariant2 * (w2z - w1z), 0.f);
          s_ab2[ps] = ab2; s_ok2[ps] = P.okey2[i];
        }
        cone_mask_row(P.cone, P.qg.nepsilon, p2x - p1x, p2y - p1y, p2z - p1z, s_mask + ps * kMaskWords, int(part), int(tpp));
      }
    }
    PROF_NOW(pc_);
    __syncthreads();                                           // (the items of a pair are walked by other threads than the one that prepared it)
    // phase C: the walk is a chain of dependent gathers (one set-1 pair per hop), so each hop is ONE round trip: the hop's
    // direction bucket, world point and successor are requested together; the bucket test reads the pair slot's LDS row.
    for (uint32_t it = threadIdx.x; it < n_items; it += blockDim.x) {
      const uint32_t ps = s_item_p[it];
      uint32_t e = s_item_e[it];
      const uint32_t* row = s_mask + ps * kMaskWords;
      const float4 eq = s_eq[ps];
      const int2 ab2 = s_ab2[ps];
      const uint32_t ok2 = s_ok2[ps];
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
    PROF_NOW(pd_);
    __syncthreads();
    const uint32_t n = min(st_n, uint32_t(kQuadStage));
    if (n) {                                                   // uniform
      // The atomic that takes the quads' positions is issued first and its value consumed only after the gate of the first chunk
      // has been computed (ComputeRigidTransformation needs nothing but the staged quad): with ~900 workgroups per base the two
      // counters are the flush -- same-address atomics are served one after the other, 10-24 us of a workgroup's ~45
      // (profiles/r06_wave_profile_before.log) -- and this takes the gate's arithmetic out of that wait.
      unsigned long long kb = 0ull;
      if (threadIdx.x == 0) kb = atomicAdd(P.K_dev, (unsigned long long)n);
      for (uint32_t c0 = 0; c0 < n; c0 += blockDim.x) {        // uniform trip count
        const uint32_t e = c0 + threadIdx.x;
        int4 quad = make_int4(0, 0, 0, 0);
        unsigned long long mix = 0ull, tag = 0ull;
        if (e < n) { quad = st_q[e]; tag = st_t[e]; mix = quad_mix(quad.x, quad.y, quad.z, quad.w); }   // counted (and summed) even when it does not fit
        { const unsigned long long ws = wave_sum_u64(mix); if (lane == 0 && ws) atomicAdd(&s_qsum, ws); }
        float T[12];
        int vd = 0;
        if (P.do_gate && e < n) vd = gate_quad<ANGLE>(P.gate, quad, T);
        if (c0 == 0u) { if (threadIdx.x == 0) st_base = kb; __syncthreads(); }
        const unsigned long long at = st_base + e;
        const bool live = e < n && at < P.K_cap;
        if (e < n && at >= P.K_cap) atomicOr(P.overflow, 4u);
        if (live) { P.quads[at] = quad; P.tags[at] = tag; }
        if (P.do_gate) {                                        // uniform
          const bool ok = vd != 0 && live;                      // (a quad beyond the capacity is no candidate: the pass is redone in chunks)
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
            store_candidate(P.gate, s_cbase + before + uint32_t(__popcll(pass & ((1ull << lane) - 1ull))), uint32_t(at) | (vd == 2 ? kBorderFlag : 0u), T, tag);
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

}  // namespace s4p
