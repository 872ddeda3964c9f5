// Micro-benchmark: cost of one cross-workgroup gradient exchange (18 KB per workgroup) + barrier per iteration, for P workgroups
// placed on the same XCD (blockIdx stride 8) or on neighbouring XCDs (stride 1). Decides whether splitting the serial PPO learner
// step over several CUs can pay. Build: hipcc --offload-arch=gfx950 -O3 tools/xcu_barrier_bench.hip -o xcu_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define NF 4608
#define PER 18

template <int MODE>
__global__ __launch_bounds__(256) void k(float* buf, unsigned* ctr, int P, int stride, int iters, float* out) {
  if (blockIdx.x % stride) return;
  const int p = blockIdx.x / stride; if (p >= P) return;
  const int tid = threadIdx.x;
  float acc[PER];
  for (int q = 0; q < PER; ++q) acc[q] = (float)(p + 1);
  const float invP = 1.f / (float)P;
  for (int it = 0; it < iters; ++it) {
    float* mine = buf + (size_t)((it & 1) * P + p) * NF;
    for (int q = 0; q < PER; ++q) {
      if (MODE == 0) __hip_atomic_store(&mine[tid + 256 * q], acc[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else mine[tid + 256 * q] = acc[q];
    }
    if (MODE == 1) __threadfence();
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(ctr, 1u, MODE == 2 ? __ATOMIC_RELAXED : __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)P * (unsigned)(it + 1);
      while (__hip_atomic_load(ctr, MODE == 2 ? __ATOMIC_RELAXED : __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    if (MODE == 1) __threadfence();
    for (int o = 0; o < P; ++o) { if (o == p) continue;
      const float* oth = buf + (size_t)((it & 1) * P + o) * NF;
      for (int q = 0; q < PER; ++q) {
        float v;
        if (MODE == 1) v = oth[tid + 256 * q]; else v = __hip_atomic_load(&oth[tid + 256 * q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        acc[q] += v; } }
    for (int q = 0; q < PER; ++q) acc[q] *= invP;
  }
  for (int q = 0; q < PER; ++q) out[(size_t)p * NF + tid + 256 * q] = acc[q];
}

int main() {
  float* buf; unsigned* ctr; float* out;
  hipMalloc(&buf, sizeof(float) * NF * 2 * 8); hipMalloc(&ctr, 256); hipMalloc(&out, sizeof(float) * NF * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int mode = 0; mode < 3; ++mode)
    for (int stride : {8, 1})
      for (int P : {1, 2, 4}) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
          hipMemset(ctr, 0, 256); hipMemset(buf, 0, sizeof(float) * NF * 2 * 8);
          hipEventRecord(e0);
          if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(stride * P), dim3(256), 0, 0, buf, ctr, P, stride, iters, out);
          else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(stride * P), dim3(256), 0, 0, buf, ctr, P, stride, iters, out);
          else hipLaunchKernelGGL(k<2>, dim3(stride * P), dim3(256), 0, 0, buf, ctr, P, stride, iters, out);
          hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        std::vector<float> h((size_t)NF * P); hipMemcpy(h.data(), out, sizeof(float) * NF * P, hipMemcpyDeviceToHost);
        float mn = 1e30f, mx = -1e30f; for (float v : h) { mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
        printf("mode %d (%s) stride %d P %d: %.3f us/iter   result in [%g, %g] expect %g\n", mode,
               mode == 0 ? "agent-scope relaxed atomics for data, acq/rel flag" : mode == 1 ? "plain data + __threadfence" : "all relaxed",
               stride, P, 1e3 * ms / iters, mn, mx, (P + 1) / 2.0);
      }
  return 0;
}
