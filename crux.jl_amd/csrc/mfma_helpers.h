// mfma_helpers.h -- lane-level helpers shared by the MFMA learner kernels (gfx950).
#pragma once
#include "train_args.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MF_HID 64
#define MF_LD 68            // padded row stride (floats) of the 64x64 LDS masters (16-B aligned rows)
#define MF_TLD 36           // row stride of the exchange tiles [64 features][32 samples + 4 pad]
#define EPS32F 1.1920928955078125e-07f

enum { MFK_CATEGORICAL = 0, MFK_GAUSSIAN = 1, MFK_VALUE = 2 };

__device__ __forceinline__ void wave_sync() {   // order LDS traffic between the lanes of ONE wave (LDS is in-order per wave)
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
// DPP lane exchange inside a 16-lane row (pure VALU, no LDS crossbar): CTRL = row_mirror 0x140 (c <-> 15-c),
// row_half_mirror 0x141 (c <-> c^7), quad_perm[3,2,1,0] 0x1B (c <-> c^3), quad_perm[1,0,3,2] 0xB1 (c <-> c^1).
template <int CTRL> __device__ __forceinline__ float dpp_x(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// Reduce-scatter of 16 per-lane values over the 16 lanes of a row: lane c returns sum over the row of v[c].
// 15 exchanges instead of 64: after the step on bit b each lane keeps the half of the indices whose bit b equals its own.
__device__ __forceinline__ float row16_reduce_scatter(const float (&v)[16], int c) {
  const bool b3 = c & 8, b2 = c & 4, b1 = c & 2, b0 = c & 1;
  float v8[8], v4[4], v2[2];
#pragma unroll
  for (int j = 0; j < 8; ++j) { const float snd = b3 ? v[j] : v[j + 8], kp = b3 ? v[j + 8] : v[j]; v8[j] = kp + dpp_x<0x140>(snd); }
#pragma unroll
  for (int j = 0; j < 4; ++j) { const float snd = b2 ? v8[j] : v8[j + 4], kp = b2 ? v8[j + 4] : v8[j]; v4[j] = kp + dpp_x<0x141>(snd); }
#pragma unroll
  for (int j = 0; j < 2; ++j) { const float snd = b1 ? v4[j] : v4[j + 2], kp = b1 ? v4[j + 2] : v4[j]; v2[j] = kp + dpp_x<0x1B>(snd); }
  const float snd = b0 ? v2[0] : v2[1], kp = b0 ? v2[1] : v2[0];
  return kp + dpp_x<0xB1>(snd);
}
// Sum over the 4 lanes that share c (one per 16-lane row); every lane gets the total. v_permlane16/32_swap are written as
// inline asm: the hipcc (ROCm 7.2) builtins return the same register for both results when both inputs are one value.
// s_nop 1 before = the VALU-write -> permlane-read hazard (2 wait states); the trailing s_nop covers the consumer.
__device__ __forceinline__ float g4_sum(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  a = a + b; b = a;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}
// relu as ONE instruction: fmaxf() makes hipcc canonicalise the MFMA result first (v_max x,x,x), doubling the count. On the bit
// pattern relu is a signed-integer max with 0 (negative floats and -0 are negative ints); a NaN keeps its bits and propagates.
__device__ __forceinline__ float relu1(float z) { const int b = __builtin_bit_cast(int, z); return __builtin_bit_cast(float, b > 0 ? b : 0); }
// tanh through the hardware exp/rcp units: 1 - 2/(exp(2z) + 1), absolute error <= 2e-7 (saturates correctly at +-1); libm tanhf costs ~40 instructions
// per element and the 17-64-64-6 tanh family evaluates 128 of them per sample and step.
__device__ __forceinline__ float tanh_fast(float z) { return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * z) + 1.f); }
template <int ACT> __device__ __forceinline__ float actf(float z) { return ACT == CRUX_ACT_RELU ? relu1(z) : (ACT == CRUX_ACT_TANH ? tanh_fast(z) : z); }
template <int ACT> __device__ __forceinline__ float actg(float y, float d) { return ACT == CRUX_ACT_RELU ? (y > 0.f ? d : 0.f) : (ACT == CRUX_ACT_TANH ? d * (1.f - y * y) : d); }

// Adam on one element with f32 arithmetic; c1 = 1/(1-b1^t), c2 = 1/(1-b2^t) come from Float64 (Flux keeps Float64 scalars;
// the f32 evaluation differs from the reference's per-element Float64 evaluation by < 1e-7 relative in the step).
struct AdamK { float b1, b2, omb1, omb2, eps, eta, c1, c2; };
__device__ __forceinline__ float adam1(float g, float& m, float& v, const AdamK& k) {
  // (explicit fma: which of the two products of `b1 m + (1 - b1) g` the compiler fuses into the addition otherwise differs from one kernel instantiation to the next, and with it
  //  the last bit of m and v -- the forms of the learner kernel are meant to be bit-identical to each other)
  m = fmaf(k.b1, m, k.omb1 * g); v = fmaf(k.b2, v, (k.omb2 * g) * g);
  return (m * k.c1) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v * k.c2) + k.eps) * k.eta;
}

