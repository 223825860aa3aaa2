// s4p_k_gridbuild.hip.hpp -- part of the gfx950 device code (included by s4p_kernels.hip.hpp, in this order; one translation unit):
// device build of the LCP structure (s4p_set_clouds).
#pragma once

namespace s4p {

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
// k_sweep's own bitmap (round 6): one bit per 2^sx x 2^sy x 2^sz cells, as fine as the LDS of a workgroup that has its CU to
// itself takes (s4p_set_clouds plans the shifts) -- k_verify's coarse bitmap has to leave room for a second workgroup, queues and
// the sample.  Built from the reach words (a cell is reachable if a P point lies within delta + 0.01 h of its box), pitches px, py
// with one empty border cube per axis as in LcpGridHost::plan.
struct SweepBitmapParams { const uint2* reach; uint32_t n_words; int nx, ny; int sx, sy, sz; uint32_t px, py; uint32_t* out; };
__global__ __launch_bounds__(256) void k_sweep_bitmap(SweepBitmapParams P) {
  for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < P.n_words; w += gridDim.x * blockDim.x) {
    uint32_t bits = P.reach[w].x;
    while (bits) {
      const uint32_t b = uint32_t(__ffs(int(bits))) - 1u; bits &= bits - 1u;
      const uint32_t c = w * 32u + b;
      const uint32_t ix = c % uint32_t(P.nx), iy = (c / uint32_t(P.nx)) % uint32_t(P.ny), iz = c / (uint32_t(P.nx) * uint32_t(P.ny));
      const uint32_t bit = ((iz >> P.sz) * P.py + (iy >> P.sy)) * P.px + (ix >> P.sx);
      atomicOr(&P.out[bit >> 5], 1u << (bit & 31u));
    }
  }
}
// The coarse bitmap moved up by `shift` bits behind zero bits, for the lean sweep of k_verify (s4p_k_lcp.hip.hpp): its index is
// (cube + 1) on x and y, so cube -1 of a row must read the (empty) border cube of the row before it and the cubes in front of the
// first row must read zero words.  dst bit b = src bit b - shift; one thread per destination word, once per cloud.
__global__ __launch_bounds__(256) void k_coarse_shift(const uint32_t* src, uint32_t n_src, uint32_t shift, uint32_t* dst, uint32_t n_dst) {
  for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < n_dst; w += gridDim.x * blockDim.x) {
    const long long first = 32ll * w - (long long)shift;          // source bit behind destination bit 32 w
    const long long sw = first >> 5;                              // (floor)
    const uint32_t so = uint32_t(first & 31ll);
    const uint32_t lo = (sw >= 0 && sw < (long long)n_src) ? src[sw] : 0u, hi = (sw + 1 >= 0 && sw + 1 < (long long)n_src) ? src[sw + 1] : 0u;
    dst[w] = so ? (lo >> so) | (hi << (32u - so)) : lo;
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

}  // namespace s4p
