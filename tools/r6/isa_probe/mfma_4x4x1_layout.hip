#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4v_t __attribute__((ext_vector_type(4)));
__global__ void k(float* o) {
  const int l = threadIdx.x;
  float a = float(l + 1), b = float(100 + l);
  f4v_t d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, f4v_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
  for (int r = 0; r < 4; ++r) o[l * 4 + r] = d[r];
}
int main() {
  float* d; (void)hipMalloc(&d, 64 * 4 * 4); k<<<1, 64>>>(d); float h[256]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l : {1, 6, 63}) { printf("lane %2d:", l); for (int r = 0; r < 4; ++r) printf("  d[%d]=%.0f (A(4b+r)*B(l) would be %d, A(l)*B(4b+r) %d)", r, h[l * 4 + r], (l / 4 * 4 + r + 1) * (100 + l), (l + 1) * (100 + l / 4 * 4 + r)); printf("\n"); }
  return 0;
}
