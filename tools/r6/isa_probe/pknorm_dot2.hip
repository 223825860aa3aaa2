#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>
__global__ void k(const float* a, const float* b, uint32_t* o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t p;
  asm("v_cvt_pknorm_u16_f32 %0, %1, %2" : "=v"(p) : "v"(a[i]), "v"(b[i]));
  o[i] = p;
}
__global__ void k2(uint32_t* o) {
  uint32_t pk = (7u << 16) | 5u, K = (100u << 16) | 1u, pz = (9u << 16) | 3u, r1, r2, r3, r4;
  asm("v_dot2_u32_u16 %0, %1, %2, %3" : "=v"(r1) : "v"(pk), "v"(K), "v"(1000u));
  asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(r2) : "v"(pz), "v"(10u), "v"(r1));
  asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(r3) : "v"(pz), "v"(10u), "v"(r1));
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(r4) : "v"(pk), "v"((6u << 16) | 6u));
  o[0] = r1; o[1] = r2; o[2] = r3; o[3] = r4;
}
int main() {
  const int n = 1 << 20;
  float *ha = new float[n], *hb = new float[n];
  // values u in [-2, 70): x = (u + 0.5) / 65535  -> expect floor(u) + 1 for u >= -1 (ties aside), 0 below
  for (int i = 0; i < n; ++i) { double u = -2.0 + 72.0 * (double)i / n; ha[i] = (float)((u + 0.5) / 65535.0); hb[i] = (float)((u + 0.5) * (1.0 / 65535.0)); }
  ha[0] = NAN; ha[1] = INFINITY; ha[2] = -INFINITY; ha[3] = 1e18f; ha[4] = -1e18f; ha[5] = 1.0f; ha[6] = 0.99999f;
  float *da, *db; uint32_t* dout;
  hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dout, n * 4);
  hipMemcpy(da, ha, n * 4, hipMemcpyHostToDevice); hipMemcpy(db, hb, n * 4, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(da, db, dout, n);
  uint32_t* ho = new uint32_t[n];
  hipMemcpy(ho, dout, n * 4, hipMemcpyDeviceToHost);
  printf("special: nan %u inf %u -inf %u 1e18 %u -1e18 %u 1.0 %u 0.99999 %u\n", ho[0] & 0xffff, ho[1] & 0xffff, ho[2] & 0xffff, ho[3] & 0xffff, ho[4] & 0xffff, ho[5] & 0xffff, ho[6] & 0xffff);
  long bad_near = 0, bad_far = 0, bad_rne = 0, bad_half = 0;
  for (int i = 8; i < n; ++i) {
    double x = (double)ha[i] * 65535.0;             // exact product in double
    double cl = x < 0 ? 0 : (x > 65535 ? 65535 : x);
    uint32_t rne = (uint32_t)nearbyint(cl), half = (uint32_t)floor(cl + 0.5);
    uint32_t got = ho[i] & 0xffff;
    if (got != rne) bad_rne++;
    if (got != half) bad_half++;
    double u = x - 0.5;                             // the coordinate this encodes
    long want = u < -1 ? 0 : (long)floor(u) + 1;
    if ((long)got != want) { double frac = u - floor(u); if (frac < 1e-4 || frac > 1 - 1e-4) bad_near++; else bad_far++; }
  }
  printf("vs exact product: != RNE %ld, != floor(x+0.5) %ld ; vs floor(u)+1: boundary cases %ld, real errors %ld\n", bad_rne, bad_half, bad_near, bad_far);
  uint32_t* d4; hipMalloc(&d4, 16); k2<<<1, 1>>>(d4); uint32_t h4[4]; hipMemcpy(h4, d4, 16, hipMemcpyDeviceToHost);
  printf("dot2 %u (expect 5*1+7*100+1000=1705) mad hi %u (expect 9*10+1705=1795) mad lo %u (expect 3*10+1705=1735) pk_min %08x (expect 00060005)\n", h4[0], h4[1], h4[2], h4[3]);
  return 0;
}
