// Is v_mfma_f32_16x16x4_f32 bit-identical to a chain of fmaf over k (k = lane group 0..3 inside one instruction)? Decides whether small-K
// products (layer-3 data gradients, K = 1..4) may be formed on the VALU inside the fused dense-engine kernels (dense_fused.h) without changing bits.
// hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/mfma_chain_test.hip -o /tmp/mfma_chain_test && /tmp/mfma_chain_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, float* D, int nmf) {   // A[n][16][4] (i,k), B[n][4][16] (k,j); one wave
  const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int n = 0; n < nmf; ++n) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(n * 16 + c) * 4 + g], B[(n * 4 + g) * 16 + c], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + c] = acc[r];
}
int main() {
  const int nmf = 8; size_t na = nmf * 64;
  float *hA = (float*)malloc(na * 4), *hB = (float*)malloc(na * 4), hD[256];
  float *dA, *dB, *dD; hipMalloc(&dA, na * 4); hipMalloc(&dB, na * 4); hipMalloc(&dD, 1024);
  int bad_seq = 0, bad_rev = 0, bad_pair = 0, bad_g0 = 0, bad_dbl = 0; srand(1);
  for (int trial = 0; trial < 200; ++trial) {
    const bool g0only = trial & 1;
    for (size_t i = 0; i < na; ++i) { hA[i] = (float)((rand() / (double)RAND_MAX - 0.5) * exp((rand() % 9) - 4.0)); hB[i] = (float)((rand() / (double)RAND_MAX - 0.5) * exp((rand() % 9) - 4.0)); }
    if (g0only) for (int n = 0; n < nmf; ++n) for (int i = 0; i < 16; ++i) for (int kk = 1; kk < 4; ++kk) { hA[(n * 16 + i) * 4 + kk] = 0.f; hB[(n * 4 + kk) * 16 + i] = 0.f; }
    hipMemcpy(dA, hA, na * 4, hipMemcpyHostToDevice); hipMemcpy(dB, hB, na * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, nmf); hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
      float s = 0.f, rv = 0.f, pr = 0.f; double dd = 0.0;
      for (int n = 0; n < nmf; ++n) {
        for (int kk = 0; kk < 4; ++kk) s = fmaf(hA[(n * 16 + i) * 4 + kk], hB[(n * 4 + kk) * 16 + j], s);
        for (int kk = 3; kk >= 0; --kk) rv = fmaf(hA[(n * 16 + i) * 4 + kk], hB[(n * 4 + kk) * 16 + j], rv);
        { double t = (double)pr; for (int kk = 0; kk < 4; ++kk) t += (double)hA[(n * 16 + i) * 4 + kk] * (double)hB[(n * 4 + kk) * 16 + j]; pr = (float)t; }
        for (int kk = 0; kk < 4; ++kk) dd += (double)hA[(n * 16 + i) * 4 + kk] * (double)hB[(n * 4 + kk) * 16 + j];
      }
      const float d = hD[i * 16 + j];
      if (g0only) { if (memcmp(&d, &s, 4)) ++bad_g0; }
      else { if (memcmp(&d, &s, 4)) ++bad_seq; if (memcmp(&d, &rv, 4)) ++bad_rev; if (memcmp(&d, &pr, 4)) ++bad_pair; float f = (float)dd; if (memcmp(&d, &f, 4)) ++bad_dbl; }
    } }
  printf("mismatches of 25600 elements each: sequential-fma k ascending %d | k descending %d | one rounding per instruction %d | one rounding overall %d | one-slice (g = 0 only) chain vs fmaf %d\n", bad_seq, bad_rev, bad_pair, bad_dbl, bad_g0);
  return 0;
}
