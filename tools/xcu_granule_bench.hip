// Micro-benchmark of the cross-workgroup gradient exchange of the learner step (4 608 floats per workgroup, P workgroups behind one L2), two transports:
//   mode 0  "flag":    plain b128 stores -> s_waitcnt -> barrier -> one atomic arrive -> poll -> barrier -> sc1 loads            (the protocol of train_mfma_kernel.h)
//   mode 1  "granule": every value travels as an 8-byte {value, tag = exchange number} written by ONE global_store_dwordx2 sc1; the consumer lane polls its own
//                      granules with global_load_dwordx2 sc1 until every tag matches -- no store acknowledgement wait, no counter, no workgroup barrier
//   mode 2  "granule16": 16-byte {v0, v1, v2, tag} chunks, dwordx4 sc1
// Build: hipcc --offload-arch=gfx950 -O3 tools/xcu_granule_bench.hip -o xcu_granule
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define PER 18
#define NF (256 * PER)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* buf, unsigned* ctr, int P, int iters, float* out, unsigned* fail, int work) {
  if (blockIdx.x & 7) return;
  const int p = blockIdx.x >> 3; if (p >= P) return;
  const int tid = threadIdx.x;
  float acc[PER];
  for (int q = 0; q < PER; ++q) acc[q] = (float)(p + 1);
  const float invP = 1.f / (float)P;
  unsigned long long polls = 0;
  for (int it = 0; it < iters; ++it) {
    // some dependent work between exchanges (a stand-in for the step's compute: `work` dependent fmas)
    for (int w = 0; w < work; ++w) for (int q = 0; q < PER; ++q) acc[q] = __builtin_fmaf(acc[q], 1.0f, 0.0f);
    const int par = it & 1;
    if (MODE == 0) {
      float* mine = buf + (size_t)(par * P + p) * 8192;
      for (int q = 0; q < 4; ++q) *(f32x4*)&mine[tid * 16 + 4 * q] = (f32x4){acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
      mine[4096 + tid] = acc[16]; mine[4096 + 256 + tid] = acc[17];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) { __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned want = (unsigned)P * (unsigned)(it + 1); unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) { __builtin_amdgcn_s_sleep(1); if (++spins > (1u << 22)) { *fail = it + 1; break; } } }
      __syncthreads();
      float tot[PER]; for (int q = 0; q < PER; ++q) tot[q] = 0.f;
      for (int o = 0; o < P; ++o) {
        if (o == p) { for (int q = 0; q < PER; ++q) tot[q] += acc[q]; continue; }
        const float* oth = buf + (size_t)(par * P + o) * 8192;
        f32x4 pw[4]; float s0, s1;
        s0 = __hip_atomic_load(oth + 4096 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s1 = __hip_atomic_load(oth + 4096 + 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:16 sc1\n\tglobal_load_dwordx4 %2, %4, off offset:32 sc1\n\t"
                     "global_load_dwordx4 %3, %4, off offset:48 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(pw[0]), "=&v"(pw[1]), "=&v"(pw[2]), "=&v"(pw[3]) : "v"(oth + tid * 16) : "memory");
        for (int q = 0; q < 16; ++q) tot[q] += pw[q >> 2][q & 3];
        tot[16] += s0; tot[17] += s1; }
      for (int q = 0; q < PER; ++q) acc[q] = tot[q] * invP;
    } else if (MODE == 1) {
      const unsigned tag = 0x40000000u | (unsigned)(it + 1);
      // granule q of lane tid at slot[par][p] + (q * 256 + tid) * 2 floats: a wave's store / load covers 512 contiguous bytes
      float* mine = buf + (size_t)(par * P + p) * (2 * NF);
      for (int q = 0; q < PER; ++q) { f32x2 gq = {acc[q], __builtin_bit_cast(float, tag)};
        asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(mine + (size_t)(q * 256 + tid) * 2), "v"(gq) : "memory"); }
      float tot[PER]; for (int q = 0; q < PER; ++q) tot[q] = 0.f;
      for (int o = 0; o < P; ++o) {
        if (o == p) { for (int q = 0; q < PER; ++q) tot[q] += acc[q]; continue; }
        const float* oth = buf + (size_t)(par * P + o) * (2 * NF);
        f32x2 gv[PER]; unsigned spins = 0;
        for (;;) {
#pragma unroll
          for (int q = 0; q < PER; ++q) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=&v"(gv[q]) : "v"(oth + (size_t)(q * 256 + tid) * 2) : "memory");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          bool ok = true;
#pragma unroll
          for (int q = 0; q < PER; ++q) { ok = ok && (__builtin_bit_cast(unsigned, gv[q][1]) == tag); }
          ++polls;
          if (__builtin_amdgcn_ballot_w64(ok) == ~0ull) break;
          if (++spins > (1u << 20)) { *fail = it + 1; break; }
        }
        for (int q = 0; q < PER; ++q) tot[q] += gv[q][0]; }
      for (int q = 0; q < PER; ++q) acc[q] = tot[q] * invP;
    } else {
      const unsigned tag = 0x40000000u | (unsigned)(it + 1);
      float* mine = buf + (size_t)(par * P + p) * 8192;
      for (int q = 0; q < 6; ++q) { f32x4 gq = {acc[3 * q], acc[3 * q + 1], acc[3 * q + 2], __builtin_bit_cast(float, tag)};
        asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(mine + (size_t)(q * 256 + tid) * 4), "v"(gq) : "memory"); }
      float tot[PER]; for (int q = 0; q < PER; ++q) tot[q] = 0.f;
      for (int o = 0; o < P; ++o) {
        if (o == p) { for (int q = 0; q < PER; ++q) tot[q] += acc[q]; continue; }
        const float* oth = buf + (size_t)(par * P + o) * 8192;
        f32x4 gv[6]; unsigned spins = 0;
        for (;;) {
#pragma unroll
          for (int q = 0; q < 6; ++q) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(gv[q]) : "v"(oth + (size_t)(q * 256 + tid) * 4) : "memory");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          bool ok = true;
#pragma unroll
          for (int q = 0; q < 6; ++q) { ok = ok && (__builtin_bit_cast(unsigned, gv[q][3]) == tag); }
          ++polls;
          if (__builtin_amdgcn_ballot_w64(ok) == ~0ull) break;
          if (++spins > (1u << 20)) { *fail = it + 1; break; }
        }
        for (int q = 0; q < 6; ++q) { tot[3 * q] += gv[q][0]; tot[3 * q + 1] += gv[q][1]; tot[3 * q + 2] += gv[q][2]; } }
      for (int q = 0; q < PER; ++q) acc[q] = tot[q] * invP;
    }
    if (*(volatile unsigned*)fail) break;
  }
  for (int q = 0; q < PER; ++q) out[(size_t)p * NF + tid + 256 * q] = acc[q];
  if (tid == 0) out[(size_t)8 * NF + p] = (float)((double)polls / (double)iters);
}

int main() {
  float* buf; unsigned* ctr; float* out; unsigned* fail;
  const size_t bufb = sizeof(float) * 8192 * 2 * 8 * 2;
  hipMalloc(&buf, bufb); hipMalloc(&ctr, 256); hipMalloc(&out, sizeof(float) * (NF * 8 + 64)); hipMalloc(&fail, 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  const char* nm[3] = {"flag (stores, waitcnt, barrier, arrive, poll, barrier, sc1 loads)", "granule 8 B {value, tag}, dwordx2 sc1", "granule 16 B {v0, v1, v2, tag}, dwordx4 sc1"};
  for (int work : {0, 64})
    for (int mode = 0; mode < 3; ++mode)
      for (int P : {2, 4}) {
        float ms = 0; unsigned hf = 0;
        for (int rep = 0; rep < 2; ++rep) {
          hipMemset(ctr, 0, 256); hipMemset(buf, 0, bufb); hipMemset(fail, 0, 256);
          hipEventRecord(e0);
          if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(8 * P), dim3(256), 0, 0, buf, ctr, P, iters, out, fail, work);
          else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(8 * P), dim3(256), 0, 0, buf, ctr, P, iters, out, fail, work);
          else hipLaunchKernelGGL(k<2>, dim3(8 * P), dim3(256), 0, 0, buf, ctr, P, iters, out, fail, work);
          hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        std::vector<float> h((size_t)NF * 8 + 64); hipMemcpy(h.data(), out, sizeof(float) * h.size(), hipMemcpyDeviceToHost); hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost);
        float mn = 1e30f, mx = -1e30f; for (size_t i = 0; i < (size_t)NF * P; ++i) { mn = h[i] < mn ? h[i] : mn; mx = h[i] > mx ? h[i] : mx; }
        printf("work %3d  P %d  %-70s %.3f us/iter  result [%g, %g] expect %g  polls/exchange %.2f  timeout-at %u\n", work, P, nm[mode], 1e3 * ms / iters, mn, mx, (P + 1) / 2.0,
               h[(size_t)8 * NF], hf);
      }
  return 0;
}
