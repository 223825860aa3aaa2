// s4p_sampler.hip -- UniformDistSampler (src/super4pcs/sampling.h:59-122) on the GPU.
//
// The reference keeps, in input order, the first point that falls into each delta-voxel
// (voxel = int(floor(coord * (1.0f / delta))), sampling.h:84-86); its open-addressing table only decides where a
// voxel is stored.  On the device:
//   k_vox_insert  one thread per point: voxel key -> slot of an open-addressing table (atomicCAS on the key),
//                 atomicMin of the point index into that slot  => per voxel, the smallest input index wins;
//   k_vox_flag    keep[i] = (winner of my voxel == i);  per-1024-block counts;
//   k_vox_scan    exclusive scan of the block counts (single workgroup);
//   k_vox_write   order-preserving compaction: out[offset(block) + rank in block] = i.
// 16 B/point in (SoA x,y,z + slot), 4 B/kept point out: HBM-bound, a few hundred microseconds per million points
// against ~30-100 ms for the host hash.  Used for clouds >= kGpuSamplerMin points when a device is visible;
// smaller clouds and device-less processes (facade users sampling before a matcher exists) take the host path.
// Both paths are required to return identical index lists (tests/test_gpu_kernels.py).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

namespace s4p {

constexpr unsigned long long kEmptyKey = ~0ull;
constexpr int kScanBlock = 1024;

__device__ __forceinline__ unsigned long long vox_hash(unsigned long long k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
  return k;
}

__global__ __launch_bounds__(256) void k_vox_insert(const float* x, const float* y, const float* z, uint32_t n, float scale,
                                                    unsigned long long* keys, uint32_t* vals, uint32_t mask, uint32_t* slot,
                                                    uint32_t* range_error) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float fx = floorf(x[i] * scale), fy = floorf(y[i] * scale), fz = floorf(z[i] * scale);     // sampling.h:84-86
    if (!(fabsf(fx) < 1048576.f && fabsf(fy) < 1048576.f && fabsf(fz) < 1048576.f)) { atomicOr(range_error, 1u); slot[i] = 0; continue; }
    const unsigned long long key = (unsigned long long)(uint32_t(int(fx) + 1048576) & 0x1FFFFFu) |
                                   ((unsigned long long)(uint32_t(int(fy) + 1048576) & 0x1FFFFFu) << 21) |
                                   ((unsigned long long)(uint32_t(int(fz) + 1048576) & 0x1FFFFFu) << 42);
    uint32_t h = uint32_t(vox_hash(key)) & mask;
    while (true) {
      const unsigned long long k = atomicCAS(&keys[h], kEmptyKey, key);
      if (k == kEmptyKey || k == key) break;
      h = (h + 1u) & mask;
    }
    atomicMin(&vals[h], i);
    slot[i] = h;
  }
}

__global__ __launch_bounds__(kScanBlock) void k_vox_flag(const uint32_t* vals, const uint32_t* slot, uint32_t n, uint32_t* block_count) {
  __shared__ uint32_t s_cnt;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  const uint32_t i = blockIdx.x * kScanBlock + threadIdx.x;
  const bool keep = i < n && vals[slot[i]] == i;
  const unsigned long long m = __ballot(keep);
  if ((threadIdx.x & 63u) == 0 && m) atomicAdd(&s_cnt, uint32_t(__popcll(m)));
  __syncthreads();
  if (threadIdx.x == 0) block_count[blockIdx.x] = s_cnt;
}

__global__ __launch_bounds__(1024) void k_vox_scan(uint32_t* block_count, uint32_t nblocks, uint32_t* total) {
  __shared__ uint32_t s_part[1024];
  // each thread owns a contiguous chunk of block counts
  const uint32_t per = (nblocks + 1023u) / 1024u;
  const uint32_t b0 = threadIdx.x * per, b1 = min(b0 + per, nblocks);
  uint32_t sum = 0;
  for (uint32_t b = b0; b < b1; ++b) sum += block_count[b];
  s_part[threadIdx.x] = sum;
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {       // Hillis-Steele inclusive scan
    const uint32_t v = threadIdx.x >= off ? s_part[threadIdx.x - off] : 0u;
    __syncthreads();
    s_part[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t run = s_part[threadIdx.x] - sum;             // exclusive prefix of my chunk
  for (uint32_t b = b0; b < b1; ++b) { const uint32_t c = block_count[b]; block_count[b] = run; run += c; }
  if (threadIdx.x == 1023) *total = s_part[1023];
}

__global__ __launch_bounds__(kScanBlock) void k_vox_write(const uint32_t* vals, const uint32_t* slot, uint32_t n,
                                                         const uint32_t* block_offset, uint32_t* out) {
  __shared__ uint32_t s_wave[kScanBlock / 64];
  const uint32_t i = blockIdx.x * kScanBlock + threadIdx.x;
  const bool keep = i < n && vals[slot[i]] == i;
  const unsigned long long m = __ballot(keep);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (lane == 0) s_wave[wave] = uint32_t(__popcll(m));
  __syncthreads();
  uint32_t before = 0;
  for (uint32_t w = 0; w < wave; ++w) before += s_wave[w];
  if (keep) out[block_offset[blockIdx.x] + before + uint32_t(__popcll(m & ((1ull << lane) - 1ull)))] = i;
}

namespace {
struct SamplerState {
  std::mutex mu;
  bool probed = false, usable = false;
  int device = -1;                 // the device the stream and buffers below live on
  hipStream_t stream = nullptr;
  float* d_xyz = nullptr; size_t cap_pts = 0;
  unsigned long long* d_keys = nullptr; uint32_t* d_vals = nullptr; size_t cap_tab = 0;
  uint32_t* d_slot = nullptr; uint32_t* d_out = nullptr; uint32_t* d_blocks = nullptr; size_t cap_blocks = 0;
  uint32_t* d_misc = nullptr;   // [0] total, [1] range error
};
SamplerState g_sampler;
}  // namespace

// Returns the number of kept points; -1 if the device path is NOT APPLICABLE (no HIP device visible -- a facade user may
// sample before any matcher exists -- or a voxel coordinate outside +-2^20: the caller's host hash is the same function);
// -2 on a HIP ERROR, which the caller reports loudly instead of hiding it behind the host path.
long long gpu_uniform_dist_sample(const float* x, const float* y, const float* z, long long n, float delta, long long* out_index, int device) {
  if (n <= 0 || n >= 0x7FFFFFF0ll) return -1;
  SamplerState& S = g_sampler;
  std::lock_guard<std::mutex> lk(S.mu);
  if (S.probed && S.usable && device >= 0 && device != S.device) {      // another GPU than last time: start over on it
    (void)hipSetDevice(S.device);
    if (S.d_xyz) { (void)hipFree(S.d_xyz); (void)hipFree(S.d_slot); (void)hipFree(S.d_out); }
    if (S.d_keys) { (void)hipFree(S.d_keys); (void)hipFree(S.d_vals); }
    if (S.d_blocks) (void)hipFree(S.d_blocks);
    if (S.d_misc) (void)hipFree(S.d_misc);
    if (S.stream) (void)hipStreamDestroy(S.stream);
    S.d_xyz = nullptr; S.d_slot = S.d_out = S.d_blocks = S.d_misc = nullptr; S.d_keys = nullptr; S.d_vals = nullptr;
    S.cap_pts = S.cap_tab = S.cap_blocks = 0; S.stream = nullptr; S.probed = false; S.usable = false;
  }
  if (!S.probed) {
    S.probed = true;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0) {
      if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
      if (device < ndev && hipSetDevice(device) == hipSuccess && hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking) == hipSuccess) { S.usable = true; S.device = device; }
    }
  }
  if (!S.usable) return -1;
  device = S.device;
  hipError_t last = hipSuccess;
  auto ok = [&last](hipError_t e) { if (e != hipSuccess) { last = e; fprintf(stderr, "super4pcs_amd: device sampler: %s\n", hipGetErrorString(e)); } return e == hipSuccess; };
  if (device >= 0 && !ok(hipSetDevice(device))) return -2;
  const size_t un = size_t(n);
  size_t tab = 1; while (tab < 2 * un) tab <<= 1;
  const uint32_t nblocks = uint32_t((un + kScanBlock - 1) / kScanBlock);
  if (S.cap_pts < un) {
    if (S.d_xyz) { (void)hipFree(S.d_xyz); (void)hipFree(S.d_slot); (void)hipFree(S.d_out); }
    S.d_xyz = nullptr; S.cap_pts = 0;
    if (!ok(hipMalloc((void**)&S.d_xyz, un * 12)) || !ok(hipMalloc((void**)&S.d_slot, un * 4)) || !ok(hipMalloc((void**)&S.d_out, un * 4))) return -2;
    S.cap_pts = un;
  }
  if (S.cap_tab < tab) {
    if (S.d_keys) { (void)hipFree(S.d_keys); (void)hipFree(S.d_vals); }
    S.d_keys = nullptr; S.cap_tab = 0;
    if (!ok(hipMalloc((void**)&S.d_keys, tab * 8)) || !ok(hipMalloc((void**)&S.d_vals, tab * 4))) return -2;
    S.cap_tab = tab;
  }
  if (S.cap_blocks < nblocks) {
    if (S.d_blocks) (void)hipFree(S.d_blocks);
    S.d_blocks = nullptr; S.cap_blocks = 0;
    if (!ok(hipMalloc((void**)&S.d_blocks, size_t(nblocks) * 4))) return -2;
    S.cap_blocks = nblocks;
  }
  if (!S.d_misc && !ok(hipMalloc((void**)&S.d_misc, 8))) return -2;
  // S4P_TRACE_INIT=1 (lab aid): upload / kernels / read-back of this call on stderr; costs two extra synchronisations
  static const bool trace = std::getenv("S4P_TRACE_INIT") != nullptr;
  using clk = std::chrono::steady_clock;
  const auto t_begin = clk::now();
  auto ms_since = [](clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); };
  float* dx = S.d_xyz; float* dy = dx + un; float* dz = dy + un;
  if (!ok(hipMemcpyAsync(dx, x, un * 4, hipMemcpyHostToDevice, S.stream)) || !ok(hipMemcpyAsync(dy, y, un * 4, hipMemcpyHostToDevice, S.stream)) ||
      !ok(hipMemcpyAsync(dz, z, un * 4, hipMemcpyHostToDevice, S.stream))) return -2;
  if (!ok(hipMemsetAsync(S.d_keys, 0xFF, tab * 8, S.stream)) || !ok(hipMemsetAsync(S.d_vals, 0xFF, tab * 4, S.stream)) ||
      !ok(hipMemsetAsync(S.d_misc, 0, 8, S.stream))) return -2;
  double ms_upload = 0.0;
  if (trace) { if (!ok(hipStreamSynchronize(S.stream))) return -2; ms_upload = ms_since(t_begin); }
  const auto t_kernels = clk::now();
  const float scale = 1.0f / delta;                                                      // sampling.h:76
  hipLaunchKernelGGL(k_vox_insert, dim3(2048), dim3(256), 0, S.stream, dx, dy, dz, uint32_t(un), scale, S.d_keys, S.d_vals, uint32_t(tab - 1), S.d_slot, S.d_misc + 1);
  hipLaunchKernelGGL(k_vox_flag, dim3(nblocks), dim3(kScanBlock), 0, S.stream, S.d_vals, S.d_slot, uint32_t(un), S.d_blocks);
  hipLaunchKernelGGL(k_vox_scan, dim3(1), dim3(1024), 0, S.stream, S.d_blocks, nblocks, S.d_misc);
  hipLaunchKernelGGL(k_vox_write, dim3(nblocks), dim3(kScanBlock), 0, S.stream, S.d_vals, S.d_slot, uint32_t(un), S.d_blocks, S.d_out);
  uint32_t misc[2] = {0, 0};
  if (!ok(hipGetLastError()) || !ok(hipMemcpyAsync(misc, S.d_misc, 8, hipMemcpyDeviceToHost, S.stream)) || !ok(hipStreamSynchronize(S.stream))) return -2;
  if (misc[1]) return -1;                         // a voxel coordinate outside +-2^20: host path handles it
  const double ms_kernels = ms_since(t_kernels);
  const auto t_back = clk::now();
  const uint32_t kept = misc[0];
  std::vector<uint32_t> idx(kept);
  if (kept && !ok(hipMemcpy(idx.data(), S.d_out, size_t(kept) * 4, hipMemcpyDeviceToHost))) return -2;
  for (uint32_t i = 0; i < kept; ++i) out_index[i] = (long long)idx[i];
  if (trace) std::fprintf(stderr, "{\"s4p_trace\": \"device sampler\", \"points\": %lld, \"kept\": %u, \"alloc_and_upload_ms\": %.3f, \"kernels_ms\": %.3f, \"read_back_ms\": %.3f}\n",
                          n, kept, ms_upload, ms_kernels, ms_since(t_back));
  return (long long)kept;
}

}  // namespace s4p
