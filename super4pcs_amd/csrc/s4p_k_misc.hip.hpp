// s4p_k_misc.hip.hpp -- part of the gfx950 device code (included by s4p_kernels.hip.hpp, in this order; one translation unit):
// k_apply, device base selection (k_select_*), k_pack_points, k_selftest, k_reset_counters.
#pragma once

namespace s4p {

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
