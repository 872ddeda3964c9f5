// Which tagged-granule formats survive an exchange between four workgroups behind one L2 (gfx950), and what a round costs. Every iteration each thread publishes NV values of
// the step in its slot and reads the same thread's values of the three peers, polling until every tag is the step's; a value that does not belong to the step whose tag it
// carries is a TORN granule. Formats:
//   A  16-byte granule {v, v, v, step}: one global_store_dwordx4, read by global_load_dwordx4 sc1
//   B   8-byte granule {v, step}: global_store_dwordx2, read by an agent-scope 8-byte atomic load (global_load_dwordx2 sc1)
//   C  two 8-byte granules per global_store_dwordx4 / global_load_dwordx4 sc1 (each half checked by its own tag)
//   D  8-byte granules stored with dwordx2, read in pairs with global_load_dwordx4 sc1
// build: hipcc --offload-arch=gfx950 -O3 -o tools/xcu_granule_test tools/xcu_granule_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#define SLOT 16384   // dwords per (parity, workgroup)
__device__ __forceinline__ unsigned val(unsigned it, unsigned p, unsigned tid, unsigned q) { return (it * 2654435761u) ^ (p * 40503u + tid * 97u + q * 7919u) ^ 0x5bd1e995u; }

template <int FMT, int NV>
__global__ __launch_bounds__(256) void k(unsigned* buf, int iters, unsigned* torn, unsigned* fail) {
  if (blockIdx.x & 7) return;
  const unsigned p = blockIdx.x >> 3, tid = threadIdx.x;
  unsigned my_torn = 0;
  for (int it = 0; it < iters; ++it) {
    const unsigned tag = (unsigned)it + 1u;
    unsigned* mine = buf + (size_t)((it & 1) * 4 + p) * SLOT;
    if (FMT == 0) { for (int j = 0; j < (NV + 2) / 3; ++j) { u32x4 q = {val(it, p, tid, 3 * j), val(it, p, tid, 3 * j + 1), val(it, p, tid, 3 * j + 2), tag}; *(u32x4*)(mine + 4 * (j * 256 + tid)) = q; } }
    if (FMT == 1 || FMT == 3) { for (int j = 0; j < NV; ++j) { u32x2 q = {val(it, p, tid, j), tag}; *(u32x2*)(mine + 2 * (j * 256 + tid)) = q; } }
    if (FMT == 2) { for (int j = 0; j < NV / 2; ++j) { u32x4 q = {val(it, p, tid, 2 * j), tag, val(it, p, tid, 2 * j + 1), tag}; *(u32x4*)(mine + 4 * (j * 256 + tid)) = q; } }
    unsigned spins = 0; bool dead = false;
    for (;;) {
      bool in = true; unsigned bad = 0;
      for (unsigned d = 1; d < 4; ++d) { const unsigned q_ = (p + d) & 3; const unsigned* peer = buf + (size_t)((it & 1) * 4 + q_) * SLOT;
        if (FMT == 0) { for (int j = 0; j < (NV + 2) / 3; ++j) { u32x4 g; asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=&v"(g) : "v"(peer + 4 * (j * 256 + tid)) : "memory");
            in = in && g.w == tag; if (g.w == tag) bad += (g.x != val(it, q_, tid, 3 * j)) + (g.y != val(it, q_, tid, 3 * j + 1)) + (g.z != val(it, q_, tid, 3 * j + 2)); } }
        if (FMT == 1) { for (int j = 0; j < NV; ++j) { const unsigned long long g = __hip_atomic_load((const unsigned long long*)(peer + 2 * (j * 256 + tid)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            in = in && (unsigned)(g >> 32) == tag; if ((unsigned)(g >> 32) == tag) bad += (unsigned)g != val(it, q_, tid, j); } }
        if (FMT == 2) { for (int j = 0; j < NV / 2; ++j) { u32x4 g; asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=&v"(g) : "v"(peer + 4 * (j * 256 + tid)) : "memory");
            in = in && g.y == tag && g.w == tag; if (g.y == tag) bad += g.x != val(it, q_, tid, 2 * j); if (g.w == tag) bad += g.z != val(it, q_, tid, 2 * j + 1); } }
        if (FMT == 3) { for (int j = 0; j < NV; ++j) { u32x2 g; asm volatile("global_load_dwordx2 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=&v"(g) : "v"(peer + 2 * (j * 256 + tid)) : "memory");
            in = in && g.y == tag; if (g.y == tag) bad += g.x != val(it, q_, tid, j); } }
      }
      if (in) { my_torn += bad; }
      if (__builtin_amdgcn_ballot_w64(in) == ~0ull) break;
      if (in) my_torn -= bad;      // (counted again on the round that ends the poll)
      if (++spins > (1u << 18) || __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { __hip_atomic_store(fail, (unsigned)it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); dead = true; break; }
    }
    if (dead) break;
    __syncthreads();
  }
  if (my_torn) atomicAdd(torn, my_torn);
}
template <int FMT, int NV> void run(const char* name, unsigned* buf, unsigned* torn, unsigned* fail) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); const int iters = 200000; float ms = 0;
  for (int rep = 0; rep < 2; ++rep) { hipMemset(buf, 0, 8 * SLOT * 4); hipMemset(torn, 0, 4); hipMemset(fail, 0, 4); hipEventRecord(e0);
    hipLaunchKernelGGL((k<FMT, NV>), dim3(32), dim3(256), 0, 0, buf, iters, torn, fail); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); }
  unsigned ht = 0, hf = 0; hipMemcpy(&ht, torn, 4, hipMemcpyDeviceToHost); hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost);
  printf("%-70s NV=%2d: %.3f us/iter, torn values %u, timeout-at-iter %u\n", name, NV, 1e3 * ms / iters, ht, hf);
}
int main() {
  unsigned *buf, *torn, *fail; hipMalloc(&buf, 8 * SLOT * 4); hipMalloc(&torn, 256); hipMalloc(&fail, 256);
  run<0, 6>("A 16-byte {v,v,v,step}, store dwordx4, load dwordx4 sc1", buf, torn, fail);
  run<0, 15>("A 16-byte {v,v,v,step}, store dwordx4, load dwordx4 sc1", buf, torn, fail);
  run<1, 8>("B 8-byte {v,step}, store dwordx2, 8-byte agent-scope atomic load", buf, torn, fail);
  run<1, 16>("B 8-byte {v,step}, store dwordx2, 8-byte agent-scope atomic load", buf, torn, fail);
  run<2, 16>("C two 8-byte granules per store dwordx4 / load dwordx4 sc1", buf, torn, fail);
  run<3, 16>("D 8-byte {v,step}, store dwordx2, asm load dwordx2 sc1", buf, torn, fail);
  return 0;
}
