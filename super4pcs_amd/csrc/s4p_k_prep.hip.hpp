// s4p_k_prep.hip.hpp -- part of the gfx950 device code (included by s4p_kernels.hip.hpp, in this order; one translation unit):
// FindCongruentQuadrilaterals, preparation side: cell hash, set-1 records, cone masks; k_prep.
#pragma once

namespace s4p {

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

// (round 6) A cell's set-1 pairs hang on kSubChains chains, not on one: an entry goes to chain (entry index mod kSubChains), and
// k_quads walks the chains of a cell as separate work items.  The walk is one dependent round trip per hop and a workgroup of
// k_quads waits for its longest walk: with one chain per cell that was ~50 hops (~35 us of the kernel's 52 alone).
constexpr uint32_t kSubChains = 4;
struct HashTable {
  unsigned long long* keys;    // (epoch << 32) | cell
  unsigned long long* heads;   // kSubChains per slot: (epoch << 32) | entry index
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
  const unsigned long long prev = atomicExch(&P.ht.heads[size_t(h) * kSubChains + (e & (kSubChains - 1u))], ((unsigned long long)P.ht.epoch << 32) | e);
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
// caller's row of kMaskWords words (zeroed by the caller; bits are set by LDS atomics, so several threads may share a row).
// Only the BUCKET of a rotated, normalised cone sample is needed: int((x / 2 + 0.5) / neps) per axis.  The exact
// sequence -- the quaternion product as Eigen writes it, a square root and six correctly rounded divisions -- is ~150
// instructions per sample.  The fast path rotates with the quaternion's 3x3 matrix (9 fma; equal to the exact product
// to ~1e-7), does not normalise (the rotated vector is unit to rounding) and multiplies by 1/neps: with
// | |d|^2 - 1 | < 1e-4 its bucket coordinates differ from the exact ones by < 2e-4 (measured < 2e-5,
// tests/test_prep_bucket_fast_path.py), so if every coordinate lies further than 4e-4 from an integer the truncations
// agree; otherwise (0.2 % of the samples) the exact sequence runs.
// (round 6) The caller zeroes the row; samples a0, a0 + astep, ... are this thread's share: k_quads gives a pair to 1, 2 or 4 threads.
__device__ __forceinline__ void cone_mask_row(const ConeTable& cone, const float nepsilon, float nx, float ny, float nz, uint32_t* row, const int a0 = 0, const int astep = 1) {
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
  // (round 6: four samples per trip, independent chains for a wave that is alone on its SIMD; the row's bits are set by LDS
  // atomics without a return value -- a read-modify-write per sample was a dependent LDS round trip per sample: phase B of
  // k_quads 11-15 us of a workgroup's ~45, profiles/r06_wave_profile.txt)
#pragma unroll 4
  for (int a = a0; a < cone.nb; a += astep) {
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
    if (id < 343u) atomicOr(&row[id >> 5], 1u << (id & 31u));
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

}  // namespace s4p
