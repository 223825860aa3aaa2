// s4p_k_pairs.hip.hpp -- part of the gfx950 device code (included by s4p_kernels.hip.hpp, in this order; one translation unit):
// ExtractPairs loop 2 + PairCreationFunctor::process: k_pairs2.
#pragma once

namespace s4p {

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
  // pair pe of wave sw's staged entries -> the ordered pair at position `at` (and, for the first set of a fused pass, its preparation)
  auto write_pair = [&](const uint32_t sw, const uint32_t pe, const uint32_t at) {
    {
      const uint32_t e = ANGLE ? pe : pe >> 1, second = ANGLE ? uint32_t(st_f[sw][e]) : pe & 1u;
      if ((ANGLE ? at : (at | 1u)) < P.cap) {              // both pairs of an entry fit, or neither is written
        const uint32_t w = st_e[sw][e], pId = w & 0xFFFFu, sl = w >> 16;
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
  };
  auto write_out = [&](const uint32_t base) {            // this wave's entries -> ordered pairs at positions base, base + 1, ...
    for (uint32_t pe = lane; pe < (ANGLE ? n_st : 2u * n_st); pe += 64u) write_pair(wave, pe, base + pe);
    wave_fence();
  };
  const uint32_t n_tiles = (P.n_q + 63u) >> 6, n_chunks = (P.n_seq + 63u) >> 6;
  // (round 6) consecutive items go to different WORKGROUPS: the heavy items -- a tile and a chunk at the base's distance from each other --
  // are neighbours, and a workgroup that held eight of them ended the launch writing (and preparing) three pairs per thread while
  // the others had none; the chunk's gathers a workgroup's waves used to share in cache are a few KB per wave out of L2
  const uint32_t gw = wave * gridDim.x + blockIdx.x, nw = gridDim.x * kPair2Waves;
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
  // (round 6) The leftovers of ALL the workgroup's waves, flattened over all its threads: the waves' shares differ widely (0 .. 190
  // entries) and a pair of the first set costs its thread a chain of dependent device-scope accesses (prep1_item: probe, CAS,
  // exchange) -- with every wave writing its own the launch ended with a few waves doing ~6 pairs per lane one after the other
  // (10-19 us, profiles/r06_wave_profile.txt).  Same positions as before: wave-major, entry order inside a wave.
  {
    const uint32_t per = ANGLE ? 1u : 2u;
    uint32_t tot_pairs = 0;
#pragma unroll
    for (int w = 0; w < kPair2Waves; ++w) tot_pairs += per * s_cnt[w];
    for (uint32_t p = threadIdx.x; p < tot_pairs; p += blockDim.x) {
      uint32_t sw = 0, start = 0;
#pragma unroll
      for (int w = 0; w < kPair2Waves - 1; ++w) { const uint32_t c = per * s_cnt[w]; if (sw == uint32_t(w) && p >= start + c) { start += c; sw = uint32_t(w) + 1u; } }
      write_pair(sw, p - start, s_base + p);
    }
  }
  PROF_STAMP(4);
#if defined(S4P_PROF)
  tp_[5] = n_st;
#endif
  PROF_WRITE(0, (blockIdx.y * gridDim.x + blockIdx.x) * kPair2Waves + wave);
}

}  // namespace s4p
