// mlp_forward.h -- value(pi, x) for narrow Chain(Dense...) networks (src/policies.jl:94,120): one workgroup = tiles of FWD_TS samples, activations ping-pong in LDS.
// Shared by mlp.hip (k_mlp_forward) and env.hip (the small-network solve kernel).
#pragma once
#include "common.h"

// ---- generic forward: one block = TS samples, activations ping-pong in LDS as [sample][feature] ----
#define FWD_TS 32
__device__ __forceinline__ void mlp_forward_run(const NetDesc& nd, const float* __restrict__ p, const float* __restrict__ x, int64_t B, float* __restrict__ y, float* sm, const unsigned bid_, const unsigned nb_) {
  float* h0 = sm; float* h1 = sm + (size_t)nd.maxdim * FWD_TS;
  const int tid = threadIdx.x;
  for (int64_t s0 = (int64_t)bid_ * FWD_TS; s0 < B; s0 += (int64_t)nb_ * FWD_TS) {
    const int ns = (int)((B - s0) < FWD_TS ? (B - s0) : FWD_TS);
    const int in0 = nd.dims[0];
    for (int idx = tid; idx < in0 * ns; idx += 256) h0[idx] = x[s0 * in0 + idx];
    __syncthreads();
    for (int l = 0; l < nd.L; ++l) {
      const int in = nd.dims[l], out = nd.dims[l + 1], act = nd.acts[l];
      const float* W = p + nd.woff[l]; const float* b = p + nd.boff[l];
      for (int idx = tid; idx < out * ns; idx += 256) {
        const int o = idx % out, s = idx / out;
        float acc = 0.f;
        for (int k = 0; k < in; ++k) acc = fmaf(W[o + out * k], h0[s * in + k], acc);
        h1[s * out + o] = crux_act(act, acc + b[o]);
      }
      __syncthreads();
      float* t = h0; h0 = h1; h1 = t;
    }
    const int outL = nd.dims[nd.L];
    for (int idx = tid; idx < outL * ns; idx += 256) y[s0 * outL + idx] = h0[idx];
    __syncthreads();
  }
}
