// Which store / load / invalidate combination lets a POLLING consumer see 8-byte {value, tag} granules written by another CU behind the same L2 (gfx950)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define PER 18
#define NF (256 * PER)
// ST: 0 plain store, 1 sc1, 2 sc0 sc1      LD: 0 asm sc1, 1 asm sc0 sc1, 2 compiler 64-bit agent atomic load, 3 asm plain      INV: 0 none, 1 buffer_inv sc0, 2 buffer_inv sc1, 3 s_sleep 1
template <int ST, int LD, int INV>
__global__ __launch_bounds__(256) void k(float* buf, int P, int iters, float* out, unsigned* fail) {
  if (blockIdx.x & 7) return;
  const int p = blockIdx.x >> 3; if (p >= P) return;
  const int tid = threadIdx.x;
  float acc[PER];
  for (int q = 0; q < PER; ++q) acc[q] = (float)(p + 1);
  const float invP = 1.f / (float)P;
  unsigned long long polls = 0;
  for (int it = 0; it < iters; ++it) {
    const int par = it & 1;
    const unsigned tag = 0x40000000u | (unsigned)(it + 1);
    float* mine = buf + (size_t)(par * P + p) * (2 * NF);
    for (int q = 0; q < PER; ++q) { float* ad = mine + (size_t)(q * 256 + tid) * 2;
      const unsigned long long pk = (unsigned long long)__builtin_bit_cast(unsigned, acc[q]) | ((unsigned long long)tag << 32);
      if (ST == 0) *(volatile unsigned long long*)ad = pk;
      else if (ST == 1) __hip_atomic_store((unsigned long long*)ad, pk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_store((unsigned long long*)ad, pk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
    float tot[PER]; for (int q = 0; q < PER; ++q) tot[q] = 0.f;
    for (int o = 0; o < P; ++o) {
      if (o == p) { for (int q = 0; q < PER; ++q) tot[q] += acc[q]; continue; }
      const float* oth = buf + (size_t)(par * P + o) * (2 * NF);
      unsigned long long gv[PER]; unsigned spins = 0;
      for (;;) {
#pragma unroll
        for (int q = 0; q < PER; ++q) { const float* ad = oth + (size_t)(q * 256 + tid) * 2;
          if (LD == 0) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=&v"(gv[q]) : "v"(ad) : "memory");
          else if (LD == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=&v"(gv[q]) : "v"(ad) : "memory");
          else if (LD == 3) asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(gv[q]) : "v"(ad) : "memory");
          else { const unsigned long long w = __hip_atomic_load((const unsigned long long*)ad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            gv[q] = w; } }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bool ok = true;
#pragma unroll
        for (int q = 0; q < PER; ++q) { ok = ok && ((unsigned)(gv[q] >> 32) == tag); }
        ++polls;
        if (__builtin_amdgcn_ballot_w64(ok) == ~0ull) break;
        if (++spins > (1u << 14)) { *fail = it + 1;
          if (!ok) { for (int q = 0; q < PER; ++q) if ((unsigned)(gv[q] >> 32) != tag) { unsigned* d = fail + 8; if (atomicAdd(fail + 1, 1u) < 6u) { const unsigned s_ = atomicAdd(fail + 2, 4u); d[s_] = (unsigned)(p * 1000 + tid); d[s_ + 1] = (unsigned)q; d[s_ + 2] = (unsigned)(gv[q] >> 32); d[s_ + 3] = (unsigned)gv[q]; } break; } }
          break; }
        if (INV == 1) asm volatile("buffer_inv sc0" ::: "memory");
        if (INV == 2) asm volatile("buffer_inv sc1" ::: "memory");
        if (INV == 3) __builtin_amdgcn_s_sleep(1);
      }
      for (int q = 0; q < PER; ++q) tot[q] += __builtin_bit_cast(float, (unsigned)gv[q]); }
    for (int q = 0; q < PER; ++q) acc[q] = tot[q] * invP;
    if (*(volatile unsigned*)fail) break;
  }
  for (int q = 0; q < PER; ++q) out[(size_t)p * NF + tid + 256 * q] = acc[q];
  if (tid == 0) out[(size_t)8 * NF + p] = (float)((double)polls / (double)iters);
}
template <int ST, int LD, int INV> void run(float* buf, float* out, unsigned* fail, size_t bufb, const char* nm) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); const int iters = 20000;
  for (int P : {2, 4}) { float ms = 0; unsigned hf = 0;
    for (int rep = 0; rep < 2; ++rep) { (void)hipMemset(buf, 0, bufb); (void)hipMemset(fail, 0, 256); (void)hipEventRecord(e0);
      hipLaunchKernelGGL((k<ST, LD, INV>), dim3(8 * P), dim3(256), 0, 0, buf, P, iters, out, fail);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1); }
    std::vector<float> h((size_t)NF * 8 + 64); (void)hipMemcpy(h.data(), out, sizeof(float) * h.size(), hipMemcpyDeviceToHost); (void)hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost);
    float mn = 1e30f, mx = -1e30f; for (size_t i = 0; i < (size_t)NF * P; ++i) { mn = h[i] < mn ? h[i] : mn; mx = h[i] > mx ? h[i] : mx; }
    { unsigned hb[8]; (void)hipMemcpy(hb, buf, 32, hipMemcpyDeviceToHost); printf("     memory slot[0][0] granules: %08x %08x | %08x %08x | %08x %08x\n", hb[0], hb[1], hb[2], hb[3], hb[4], hb[5]); }
    { unsigned hd[64]; (void)hipMemcpy(hd, fail, 256, hipMemcpyDeviceToHost); for (unsigned i = 0; i < hd[2] && i < 24; i += 4) printf("     lane %u granule %u: tag seen %08x value bits %08x\n", hd[8 + i], hd[9 + i], hd[10 + i], hd[11 + i]); }
    printf("P %d  %-60s %8.3f us/iter  result [%g, %g] expect %g  polls/exchange %.2f  timeout-at %u\n", P, nm, 1e3 * ms / iters, mn, mx, (P + 1) / 2.0, h[(size_t)8 * NF], hf); }
}
int main() {
  float* buf; float* out; unsigned* fail; const size_t bufb = sizeof(float) * 2 * NF * 2 * 8;
  (void)hipMalloc(&buf, bufb); (void)hipMalloc(&out, sizeof(float) * (NF * 8 + 64)); (void)hipMalloc(&fail, 256);
  run<1, 0, 0>(buf, out, fail, bufb, "st sc1, ld sc1");
  run<0, 2, 0>(buf, out, fail, bufb, "st plain, ld compiler agent atomic u64");
  run<1, 2, 0>(buf, out, fail, bufb, "st agent atomic, ld compiler agent atomic u64");
  run<1, 2, 3>(buf, out, fail, bufb, "st agent atomic, ld agent atomic u64, s_sleep 1 per re-poll");
  run<1, 0, 0>(buf, out, fail, bufb, "st agent atomic, ld asm sc1 x2");
  return 0;
  run<2, 1, 0>(buf, out, fail, bufb, "st sc0 sc1, ld sc0 sc1");
  run<1, 1, 0>(buf, out, fail, bufb, "st sc1, ld sc0 sc1");
  run<1, 2, 0>(buf, out, fail, bufb, "st sc1, ld compiler agent atomic u64");
  run<0, 2, 0>(buf, out, fail, bufb, "st plain, ld compiler agent atomic u64");
  run<1, 0, 1>(buf, out, fail, bufb, "st sc1, ld sc1, buffer_inv sc0 per re-poll");
  run<1, 0, 2>(buf, out, fail, bufb, "st sc1, ld sc1, buffer_inv sc1 per re-poll");
  run<1, 0, 3>(buf, out, fail, bufb, "st sc1, ld sc1, s_sleep 1 per re-poll");
  run<1, 3, 1>(buf, out, fail, bufb, "st sc1, ld plain, buffer_inv sc0 per re-poll");
  return 0;
}
