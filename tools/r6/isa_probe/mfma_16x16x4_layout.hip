#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4v_t __attribute__((ext_vector_type(4)));
// A[i][k] = 100*i + k (i<16,k<4), B[k][j] = (k==K0 ? 1 : 0) * (j+1)  -> D[i][j] = A[i][K0]*(j+1)
__global__ void k(float* o, int K0) {
  const int l = threadIdx.x;
  const int ai = l & 15, ak = l >> 4, bk = l >> 4, bj = l & 15;
  float a = float(100 * ai + ak), b = (bk == K0) ? float(bj + 1) : 0.f;
  f4v_t d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, f4v_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
  for (int r = 0; r < 4; ++r) o[l * 4 + r] = d[r];
}
int main() {
  float* d; (void)hipMalloc(&d, 64 * 4 * 4); float h[256];
  for (int K0 : {0, 2}) {
    k<<<1, 64>>>(d, K0); (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) { int i = 4 * (l >> 4) + r, j = l & 15; float want = float(100 * i + K0) * float(j + 1); if (h[l * 4 + r] != want) { if (bad < 5) printf("K0 %d lane %d r %d got %.0f want %.0f\n", K0, l, r, h[l * 4 + r], want); ++bad; } }
    printf("K0=%d: %d mismatches against D[4*(l>>4)+r][l&15] with A[l&15][l>>4], B[l>>4][l&15]\n", K0, bad);
  }
  return 0;
}
