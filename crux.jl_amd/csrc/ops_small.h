// ops_small.h -- device bodies of small kernels that live in buffer.hip / train.hip / mlp.hip and are ALSO executed by the fused-step executor
// (exec.hip) as ops: one definition, two callers. Reference lines are cited at the kernels' home files.
#pragma once
#include "common.h"

struct PerUpdateOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, float* __restrict__ pr, float* pminmax, const int64_t* __restrict__ I,
                             const double* __restrict__ v64, const float* __restrict__ v32, const float* vconst_from_max,
                             float alpha, int64_t n) {
  __shared__ int64_t ids_s[512];           // small calls (a sampled minibatch): the "last write wins" scan over the ids runs out of LDS (from global memory it cost 27 us for n = 128)
  bool sorted_ids = false;
  if (n <= 512 && !vconst_from_max) { for (int64_t j = threadIdx.x; j < n; j += blockDim.x) ids_s[j] = I[j]; __syncthreads();
    int ok = 1; for (int64_t j = threadIdx.x; j + 1 < n; j += blockDim.x) ok &= ids_s[j] <= ids_s[j + 1] ? 1 : 0;
    sorted_ids = __syncthreads_and(ok) != 0; }          // stratified samples arrive in ascending order: duplicates are then neighbours
  int wmax = (int)0x80000000, wmin = 0x7fffffff;
  for (int64_t i = (int64_t)bid_ * blockDim.x + threadIdx.x; i < n; i += (int64_t)nb_ * blockDim.x) {
    double val;
    if (vconst_from_max) val = (double)vconst_from_max[0] + (double)1.1920928955078125e-07f;   // push!: max_priority*ones(N) (Float64)
    else if (v64) val = v64[i] + (double)1.1920928955078125e-07f;
    else { const float vf = __fadd_rn(v32[i], 1.1920928955078125e-07f); val = (double)vf; }
    // priorities[I] = val.^alpha is a sequential scatter in the reference (:297): with repeated indices the LAST value wins. Small calls (the
    // sampled-batch case) resolve that exactly; large calls are ring pushes, whose repeats (N > capacity) carry the same value anyway.
    bool later = false;
    if (n <= 512 && !vconst_from_max) { const int64_t me = ids_s[i];
      if (sorted_ids) later = i + 1 < n && ids_s[i + 1] == me;
      else for (int64_t j = i + 1; j < n; ++j) if (ids_s[j] == me) { later = true; break; } }
    if (!later) pr[I[i]] = (float)pow(val, (double)alpha);
    const float vf32 = (float)val;
    wmax = max(wmax, __float_as_int(vf32)); wmin = min(wmin, __float_as_int(vf32));
  }
  // positive floats order like their bit patterns: one atomic pair per WAVE (256 same-address atomics from one block cost 13 us at the L2)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { wmax = max(wmax, __shfl_xor(wmax, o, 64)); wmin = min(wmin, __shfl_xor(wmin, o, 64)); }
  if ((threadIdx.x & 63) == 0 && wmax != (int)0x80000000) { atomicMax((int*)&pminmax[0], wmax); atomicMin((int*)&pminmax[1], wmin); }
} };
struct DqnTargetOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, const float* __restrict__ q, int nout, const float* __restrict__ r, const uint8_t* __restrict__ done, float gamma, int64_t n, float* __restrict__ y) {
#pragma clang fp contract(off)      // the reference evaluates r .+ gamma .* (1 .- done) .* q un-fused; train.hip is not built with -ffp-contract=off
  const int64_t s = (int64_t)bid_ * blockDim.x + threadIdx.x; if (s >= n) return;
  float mx = q[s * nout]; for (int k = 1; k < nout; ++k) mx = q[s * nout + k] > mx ? q[s * nout + k] : mx;
  const float nd = 1.f - (done[s] ? 1.f : 0.f); const float gn = gamma * nd; const float t = gn * mx; y[s] = r[s] + t;      // plain operators: the pragma above governs them (the __f*_rn intrinsics are inline functions compiled under the unit's own contraction mode)
   // r .+ gamma .* (1 .- done) .* max  (dqn.jl:5)
} };
struct SoftqTargetOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, const float* __restrict__ q, int nout, const float* __restrict__ r, const uint8_t* __restrict__ done, float gamma, float alpha, int64_t n, float* __restrict__ y) {
#pragma clang fp contract(off)      // softq_target(alpha) (rl/softq.jl:1-13): r .+ gamma .* (1 .- done) .* (alpha .* logsumexp(Q ./ alpha)), un-fused
  const int64_t s = (int64_t)bid_ * blockDim.x + threadIdx.x; if (s >= n) return;
  float mx = q[s * nout] / alpha; for (int k = 1; k < nout; ++k) { const float v = q[s * nout + k] / alpha; mx = v > mx ? v : mx; }
  float sum = 0.f; for (int k = 0; k < nout; ++k) sum = sum + expf(q[s * nout + k] / alpha - mx);
  const float lse = mx + logf(sum); const float sv = alpha * lse;                                 // soft_value (softq.jl:1)
  const float nd = 1.f - (done[s] ? 1.f : 0.f); const float gn = gamma * nd; const float t = gn * sv; y[s] = r[s] + t;
} };
struct TdErrorOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, const float* __restrict__ q, int nout, const uint8_t* __restrict__ a, const float* __restrict__ y, int64_t n, float* __restrict__ err) {
#pragma clang fp contract(off)      // the reference evaluates r .+ gamma .* (1 .- done) .* q un-fused; train.hip is not built with -ffp-contract=off
  const int64_t s = (int64_t)bid_ * blockDim.x + threadIdx.x; if (s >= n) return;
  float Q = 0.f; for (int k = 0; k < nout; ++k) { const float t = q[s * nout + k] * (a[s * nout + k] ? 1.f : 0.f); Q = Q + t; }
  err[s] = fabsf(Q - y[s]);
} };
struct PolyakOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, float* __restrict__ to, const float* __restrict__ from, float tau, int64_t n) {
  const float omt = __fsub_rn(1.0f, tau);
  for (int64_t i = (int64_t)bid_ * blockDim.x + threadIdx.x; i < n; i += (int64_t)nb_ * blockDim.x)      // element-wise: any grid gives the same result
    to[i] = __fadd_rn(__fmul_rn(tau, from[i]), __fmul_rn(omt, to[i]));   // tau .* from .+ (1f0 - tau) .* to, no contraction
} };
struct CopyF32Op { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, float* __restrict__ dst, const float* __restrict__ src, int64_t n) {
  for (int64_t i = (int64_t)bid_ * blockDim.x + threadIdx.x; i < n; i += (int64_t)nb_ * blockDim.x) dst[i] = src[i];
} };
