// Tagged-chunk exchange with compiler-generated agent-scope dword loads only (no inline-asm loads): latency and correctness, 2 workgroups, same XCD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NCH>
__global__ __launch_bounds__(256) void k(float* buf, int iters, float* out, unsigned* fail) {
  if (blockIdx.x & 7) return;
  const int p = blockIdx.x >> 3, tid = threadIdx.x;
  float acc[3 * NCH];
  for (int q = 0; q < 3 * NCH; ++q) acc[q] = (float)(p + 1);
  for (int it = 0; it < iters; ++it) {
    const int tagi = 0x40000000 | (it + 1); const float tagf = __builtin_bit_cast(float, tagi);
    float* mine = buf + (size_t)((it & 1) * 2 + p) * 8192 + tid * 32;
    const float* oth = buf + (size_t)((it & 1) * 2 + (1 - p)) * 8192 + tid * 32;
    for (int q = 0; q < NCH; ++q) *(f32x4*)&mine[4 * q] = (f32x4){acc[3 * q], acc[3 * q + 1], acc[3 * q + 2], tagf};
    float pv[4 * NCH]; unsigned spins = 0;
    for (;;) {
      for (int q = 0; q < 4 * NCH; ++q) pv[q] = __hip_atomic_load(oth + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      bool ok = true;
      for (int q = 0; q < NCH; ++q) ok = ok && (__builtin_bit_cast(int, pv[4 * q + 3]) == tagi);
      if (__builtin_amdgcn_ballot_w64(ok) == ~0ull) break;
      if (++spins > (1u << 16) || __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { __hip_atomic_store(fail, (unsigned)it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
    for (int q = 0; q < NCH; ++q) for (int r = 0; r < 3; ++r) acc[3 * q + r] = 0.5f * (acc[3 * q + r] + pv[4 * q + r]);
    __syncthreads();
    if (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
  }
  float s = 0; for (int q = 0; q < 3 * NCH; ++q) s += acc[q];
  out[p * 256 + tid] = s / (3 * NCH);
}
int main() {
  float* buf; float* out; unsigned* fail; hipMalloc(&buf, 4 * 8192 * 4); hipMalloc(&out, 2048); hipMalloc(&fail, 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); const int iters = 20000;
  for (int n : {2, 7}) { float ms = 0; unsigned hf = 0;
    for (int rep = 0; rep < 2; ++rep) { hipMemset(buf, 0, 4 * 8192 * 4); hipMemset(fail, 0, 256); hipEventRecord(e0);
      if (n == 2) hipLaunchKernelGGL(k<2>, dim3(16), dim3(256), 0, 0, buf, iters, out, fail); else hipLaunchKernelGGL(k<7>, dim3(16), dim3(256), 0, 0, buf, iters, out, fail);
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); }
    float h[512]; hipMemcpy(h, out, 2048, hipMemcpyDeviceToHost); hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost);
    float mn = 1e30f, mx = -1e30f; for (float v : h) { mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
    printf("tagged exchange, %d chunks/lane, dword agent-scope loads: %.3f us/iter, result in [%g, %g] expect 1.5, timeout-at-iter %u\n", n, 1e3 * ms / iters, mn, mx, hf); }
  return 0;
}
