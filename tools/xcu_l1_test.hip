// Does a `global_load_dwordx4 ... sc1` (inline asm) re-read after another CU's store see the new data, or can it hit a stale line in the
// reader's vector L1? Footprint is tiny (4 KB per slot) so nothing is evicted by capacity. Compare with compiler-generated agent-scope
// atomic dword loads. Build: hipcc --offload-arch=gfx950 -O3 tools/xcu_l1_test.hip -o xcu_l1
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>   // 0: asm b128 sc1   1: asm b128 sc0 sc1   2: 4 x atomic dword loads   3: asm b128 sc1 after buffer_inv sc1
__global__ __launch_bounds__(256) void k(float* buf, unsigned* ctr, int iters, unsigned* bad) {
  if (blockIdx.x & 7) return;
  const int p = blockIdx.x >> 3, tid = threadIdx.x; unsigned nbad = 0;
  for (int it = 0; it < iters; ++it) {
    float* mine = buf + (size_t)((it & 1) * 2 + p) * 1024 + tid * 4;
    const float* oth = buf + (size_t)((it & 1) * 2 + (1 - p)) * 1024 + tid * 4;
    *(f32x4*)mine = (f32x4){(float)it, (float)(it + 1), (float)(it + 2), (float)(it + 3)};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) { __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 2u * (unsigned)(it + 1)) __builtin_amdgcn_s_sleep(1); }
    __syncthreads();
    f32x4 v;
    if (MODE == 0) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(oth) : "memory");
    else if (MODE == 1) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(oth) : "memory");
    else if (MODE == 3) asm volatile("buffer_inv sc1\n\tglobal_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(oth) : "memory");
    else { for (int q = 0; q < 4; ++q) v[q] = __hip_atomic_load(oth + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    if (v[0] != (float)it || v[3] != (float)(it + 3)) nbad += 1;
  }
  if (nbad) atomicAdd(bad, nbad);
}
int main() {
  float* buf; unsigned* ctr; unsigned* bad; hipMalloc(&buf, 4 * 4096 * 4); hipMalloc(&ctr, 256); hipMalloc(&bad, 256);
  const char* nm[4] = {"asm b128 sc1", "asm b128 sc0 sc1", "4 x atomic dword (compiler, sc1)", "buffer_inv sc1 + asm b128 sc1"};
  for (int m = 0; m < 4; ++m) { hipMemset(buf, 0, 4 * 4096 * 4); hipMemset(ctr, 0, 256); hipMemset(bad, 0, 256);
    if (m == 0) hipLaunchKernelGGL(k<0>, dim3(16), dim3(256), 0, 0, buf, ctr, 2000, bad);
    if (m == 1) hipLaunchKernelGGL(k<1>, dim3(16), dim3(256), 0, 0, buf, ctr, 2000, bad);
    if (m == 2) hipLaunchKernelGGL(k<2>, dim3(16), dim3(256), 0, 0, buf, ctr, 2000, bad);
    if (m == 3) hipLaunchKernelGGL(k<3>, dim3(16), dim3(256), 0, 0, buf, ctr, 2000, bad);
    unsigned h = 0; hipDeviceSynchronize(); hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    printf("%-36s stale reads: %u of %u\n", nm[m], h, 2000u * 512u); }
  return 0;
}
