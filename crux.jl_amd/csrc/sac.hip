// sac.hip -- off-policy actor-critic learner steps (SAC, DDPG, TD3) on the dense engine (dense.hip).
// Reference: src/model_free/rl/sac.jl:4-9 (sac_target), :34-40 (sac_actor_loss), :45-52 (sac_temp_loss); double_Q_loss
// src/utils.jl:89-96; GaussianPolicy exploration / gaussian_logpdf src/policies.jl:333-344; value(pi, s, a) = net(vcat(s, a))
// src/policies.jl:96; train! src/training.jl:13-25 (gradient norm, NaN => error before the update, Adam).
//
// Each step is a short stream-ordered chain: tile GEMMs for the networks, small elementwise/reduction kernels for the heads,
// and a device-side "norm is NaN => skip Adam" so that one host read-back per step (the info row) is the only synchronisation.
#include "common.h"
#include "exec.h"

// Box-Muller standard normal, first output; identical to the rollout's definition (env.hip, oracle randn_f32).
__device__ __forceinline__ float sac_randn(uint64_t seed, uint64_t ctr, uint32_t stream) {
  const crux_u32x4 x = crux_philox(seed, ctr, stream, CRUX_RNG_NOISE);
  const double u1 = crux_u32x2_to_f64(x.v[0], x.v[1]), u2 = crux_u32x2_to_f64(x.v[2], x.v[3]);
  return (float)(sqrt(-2.0 * log(1.0 - u1)) * cos(2.0 * M_PI * u2));
}

// exploration(pi::GaussianPolicy, s) (policies.jl:338-344) from the cached means: a = eps*sigma + mu, logprob, eps; and sa = vcat(s, a).
struct GaussExploreOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, const float* __restrict__ mu, const float* __restrict__ ls, const float* __restrict__ s, int od, int ad, int64_t B,
                                uint64_t seed, uint64_t counter, float* __restrict__ sa, float* __restrict__ lp, float* __restrict__ eps) {
  const int64_t j = (int64_t)bid_ * blockDim.x + threadIdx.x; if (j >= B) return;
  float acc = 0.f;
  for (int d = 0; d < ad; ++d) {
    const float sg = expf(ls[d]); const float e = sac_randn(seed, counter, (uint32_t)(j * ad + d));
    const float m = mu[j * ad + d]; const float a = __fadd_rn(__fmul_rn(e, sg), m);
    const float s2 = __fmul_rn(sg, sg), df = __fsub_rn(a, m);
    acc = __fadd_rn(acc, __fsub_rn(__fsub_rn(-(__fmul_rn(df, df)) / __fmul_rn(2.f, s2), 0.9189385332046727f), ls[d]));
    if (sa) sa[j * (od + ad) + od + d] = a;
    if (eps) eps[j * ad + d] = e;
  }
  if (sa) for (int k = 0; k < od; ++k) sa[j * (od + ad) + k] = s[j * od + k];
  lp[j] = acc;
} };
__global__ void k_gauss_explore(const float* __restrict__ mu, const float* __restrict__ ls, const float* __restrict__ s, int od, int ad, int64_t B,
                                uint64_t seed, uint64_t counter, float* __restrict__ sa, float* __restrict__ lp, float* __restrict__ eps) { GaussExploreOp::run(blockIdx.x, gridDim.x, mu, ls, s, od, ad, B, seed, counter, sa, lp, eps); }
struct ConcatSaOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, const float* __restrict__ s, const float* __restrict__ a, int od, int ad, int64_t B, float* __restrict__ sa) {
  const int64_t i = (int64_t)bid_ * blockDim.x + threadIdx.x; if (i >= B * (od + ad)) return;
  const int64_t j = i / (od + ad); const int k = (int)(i - j * (od + ad));
  sa[i] = k < od ? s[j * od + k] : a[j * ad + (k - od)];
} };
__global__ void k_concat_sa(const float* __restrict__ s, const float* __restrict__ a, int od, int ad, int64_t B, float* __restrict__ sa) { ConcatSaOp::run(blockIdx.x, gridDim.x, s, a, od, ad, B, sa); }
struct SacTargetOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, const float* __restrict__ r, const uint8_t* __restrict__ done, const float* __restrict__ q1, const float* __restrict__ q2,
                             const float* __restrict__ lp, const float* __restrict__ log_alpha, float gamma, int64_t B, float* __restrict__ y) {
  const int64_t j = (int64_t)bid_ * blockDim.x + threadIdx.x; if (j >= B) return;
  const float alpha = expf(log_alpha[0]); const float mn = q2[j] < q1[j] ? q2[j] : q1[j];
  y[j] = __fadd_rn(r[j], __fmul_rn(__fmul_rn(gamma, __fsub_rn(1.f, done[j] ? 1.f : 0.f)), __fsub_rn(mn, __fmul_rn(alpha, lp[j]))));
} };
__global__ void k_sac_target(const float* __restrict__ r, const uint8_t* __restrict__ done, const float* __restrict__ q1, const float* __restrict__ q2,
                             const float* __restrict__ lp, const float* __restrict__ log_alpha, float gamma, int64_t B, float* __restrict__ y) { SacTargetOp::run(blockIdx.x, gridDim.x, r, done, q1, q2, lp, log_alpha, gamma, B, y); }

// DDPG / TD3 target actions (ddpg.jl:6-18, td3.jl:4-7): a' = mu(sp) [smoothed: clamp(a' + clamp(sigma*randn, eps_min, eps_max), a_min, a_max), policies.jl:510-514]; sa = vcat(sp, a')
struct DpgActionOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, const float* __restrict__ mu, const float* __restrict__ s, int od, int ad, int64_t B, float sigma, float emin, float emax, float amin, float amax,
                             uint64_t seed, uint64_t counter, float* __restrict__ sa) {
  const int64_t i = (int64_t)bid_ * blockDim.x + threadIdx.x; if (i >= B * (od + ad)) return;
  const int64_t j = i / (od + ad); const int k = (int)(i - j * (od + ad));
  if (k < od) { sa[i] = s[j * od + k]; return; }
  const int d = k - od; float a = mu[j * ad + d];
  if (sigma >= 0.f) { float e = __fmul_rn(sac_randn(seed, counter, (uint32_t)(j * ad + d)), sigma); e = fminf(fmaxf(e, emin), emax); a = fminf(fmaxf(__fadd_rn(a, e), amin), amax); }
  sa[i] = a;
} };
__global__ void k_dpg_action(const float* __restrict__ mu, const float* __restrict__ s, int od, int ad, int64_t B, float sigma, float emin, float emax, float amin, float amax,
                             uint64_t seed, uint64_t counter, float* __restrict__ sa) { DpgActionOp::run(blockIdx.x, gridDim.x, mu, s, od, ad, B, sigma, emin, emax, amin, amax, seed, counter, sa); }
struct DpgTargetOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, const float* __restrict__ r, const uint8_t* __restrict__ done, const float* __restrict__ q1, const float* __restrict__ q2, float gamma, int64_t B, float* __restrict__ y) {
  const int64_t j = (int64_t)bid_ * blockDim.x + threadIdx.x; if (j >= B) return;
  const float q = q2 ? (q2[j] < q1[j] ? q2[j] : q1[j]) : q1[j];
  y[j] = __fadd_rn(r[j], __fmul_rn(__fmul_rn(gamma, __fsub_rn(1.f, done[j] ? 1.f : 0.f)), q));
} };
__global__ void k_dpg_target(const float* __restrict__ r, const uint8_t* __restrict__ done, const float* __restrict__ q1, const float* __restrict__ q2, float gamma, int64_t B, float* __restrict__ y) { DpgTargetOp::run(blockIdx.x, gridDim.x, r, done, q1, q2, gamma, B, y); }
struct FillOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, float* __restrict__ p, float v, int64_t n) { const int64_t i = (int64_t)bid_ * blockDim.x + threadIdx.x; if (i < n) p[i] = v; } };
__global__ void k_fill(float* __restrict__ p, float v, int64_t n) { FillOp::run(blockIdx.x, gridDim.x, p, v, n); }
struct SliceRowsOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, const float* __restrict__ src, int ld, int off, int rows, int64_t B, float* __restrict__ dst) {   // dst[r + rows*j] = src[off + r + ld*j]
  const int64_t i = (int64_t)bid_ * blockDim.x + threadIdx.x; if (i >= B * rows) return;
  const int64_t j = i / rows; const int r = (int)(i - j * rows); dst[i] = src[off + r + (int64_t)ld * j];
} };
__global__ void k_slice_rows(const float* __restrict__ src, int ld, int off, int rows, int64_t B, float* __restrict__ dst) { SliceRowsOp::run(blockIdx.x, gridDim.x, src, ld, off, rows, B, dst); }
#define SUMSQ_BLOCKS 64
// out[0] = the 64 partials added in block order. Called by thread 0 of the single-block info kernel that follows k_sumsq2 in every step sequence
// (stream order makes the partials visible): the earlier "last block to arrive combines" form needed two device-scope fences and took 11 us.
__device__ __forceinline__ void ssq_finalize(const double* ssq_c) { double* ssq = const_cast<double*>(ssq_c); double t = 0; for (int k = 0; k < SUMSQ_BLOCKS; ++k) t += ssq[1 + k]; ssq[0] = t; }
struct MeanInfoOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, const float* __restrict__ q, int64_t B, float sign, const double* __restrict__ ssq, float* __restrict__ dinfo) {   // single thread block of 256
  __shared__ double red[4];
  double s = 0; for (int64_t j = threadIdx.x; j < B; j += 256) s += (double)q[j];
  s = wave_sum_d(s); __syncthreads(); if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s; __syncthreads();
  if (threadIdx.x == 0) { ssq_finalize(ssq); dinfo[CRUX_INFO_LOSS] = sign * (float)((((red[0] + red[1]) + red[2]) + red[3]) / (double)B); dinfo[CRUX_INFO_GRAD_NORM] = (float)sqrt(ssq[0]); }
} };
__global__ void k_mean_info(const float* __restrict__ q, int64_t B, float sign, const double* __restrict__ ssq, float* __restrict__ dinfo) { MeanInfoOp::run(blockIdx.x, gridDim.x, q, B, sign, ssq, dinfo); }

// deterministic single-block reductions (256 threads; double accumulators like the oracle)
__device__ __forceinline__ double block_sum256(double v, double* red) {
  v = wave_sum_d(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return ((red[0] + red[1]) + red[2]) + red[3];
}
// sac_temp_loss: loss = -mean(alpha*(lp + H)); d/dlog_alpha = the same value. dev_info: [LOSS, GRAD_NORM, ALPHA]; ssq = grad^2 (NaN gate)
struct TempHeadOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, const float* __restrict__ lp, int64_t B, float H, const float* __restrict__ log_alpha, float* __restrict__ g,
                                                   float* __restrict__ dinfo, double* __restrict__ ssq) {
  __shared__ double red[4];
  const float alpha = expf(log_alpha[0]); double st = 0;
  for (int64_t j = threadIdx.x; j < B; j += 256) st += (double)(alpha * (lp[j] + H));
  st = block_sum256(st, red);
  if (threadIdx.x == 0) { const float m = (float)(st / (double)B); g[0] = -m; dinfo[CRUX_INFO_LOSS] = -m; dinfo[CRUX_INFO_GRAD_NORM] = fabsf(m); dinfo[CRUX_INFO_ALPHA] = alpha;
    ssq[0] = (double)m * (double)m; }
} };
__global__ __launch_bounds__(256) void k_temp_head(const float* __restrict__ lp, int64_t B, float H, const float* __restrict__ log_alpha, float* __restrict__ g,
                                                   float* __restrict__ dinfo, double* __restrict__ ssq) { TempHeadOp::run(blockIdx.x, gridDim.x, lp, B, H, log_alpha, g, dinfo, ssq); }
// td head of one Q network: dy = 0.5 * 2 (Q - y) w / B; stats: sum (Q-y)^2 w, sum Q
struct QHeadOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, const float* __restrict__ Q, const float* __restrict__ y, const float* __restrict__ w, int64_t B, float scale,
                                                float* __restrict__ dy, double* __restrict__ stats /* [2] */) {
  __shared__ double red[4];
  const float invB = 1.f / (float)B; double sl = 0, sq = 0;
  for (int64_t j = threadIdx.x; j < B; j += 256) { const float d = Q[j] - y[j]; const float ww = w ? w[j] : 1.f; sl += (double)(d * d * ww); sq += (double)Q[j];
    dy[j] = scale * (2.f * d * ww * invB); }
  sl = block_sum256(sl, red); sq = block_sum256(sq, red);
  if (threadIdx.x == 0) { stats[0] = sl; stats[1] = sq; }
} };
__global__ __launch_bounds__(256) void k_q_head(const float* __restrict__ Q, const float* __restrict__ y, const float* __restrict__ w, int64_t B, float scale,
                                                float* __restrict__ dy, double* __restrict__ stats /* [2] */) { QHeadOp::run(blockIdx.x, gridDim.x, Q, y, w, B, scale, dy, stats); }
// td_loss head for a DiscreteNetwork critic (utils.jl:76-87, policies.jl:122): Q = sum(value .* onehot); dy = onehot * 2 (Q - y) w / B
struct TdHeadOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, const float* __restrict__ z, const uint8_t* __restrict__ a, int nout, const float* __restrict__ y, const float* __restrict__ w, int64_t B,
                                                 float* __restrict__ dy, double* __restrict__ stats /* [2] */, float* __restrict__ err /* td_error(pi, D, y) of the same forward pass, or NULL */) {
  __shared__ double red[4];
  const float invB = 1.f / (float)B; double sl = 0, sq = 0;
  for (int64_t j = threadIdx.x; j < B; j += 256) { float Q = 0.f;
    for (int k = 0; k < nout; ++k) Q += z[j * nout + k] * (a[j * nout + k] ? 1.f : 0.f);
    const float d = Q - y[j]; if (err) err[j] = fabsf(d); const float ww = w ? w[j] : 1.f; sl += (double)(d * d * ww); sq += (double)Q;
    for (int k = 0; k < nout; ++k) dy[j * nout + k] = a[j * nout + k] ? 2.f * d * ww * invB : 0.f; }
  sl = block_sum256(sl, red); sq = block_sum256(sq, red);
  if (threadIdx.x == 0) { stats[0] = sl; stats[1] = sq; }
} };
__global__ __launch_bounds__(256) void k_td_head(const float* __restrict__ z, const uint8_t* __restrict__ a, int nout, const float* __restrict__ y, const float* __restrict__ w, int64_t B,
                                                 float* __restrict__ dy, double* __restrict__ stats /* [2] */, float* __restrict__ err /* td_error(pi, D, y) of the same forward pass, or NULL */) { TdHeadOp::run(blockIdx.x, gridDim.x, z, a, nout, y, w, B, dy, stats, err); }
struct TdInfoOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, const double* __restrict__ st, const double* __restrict__ ssq, int64_t B, float* __restrict__ dinfo) { if (threadIdx.x != 0) return;
  ssq_finalize(ssq);
  dinfo[CRUX_INFO_LOSS] = (float)(st[0] / (double)B); dinfo[2] = (float)(st[1] / (double)B); dinfo[CRUX_INFO_GRAD_NORM] = (float)sqrt(ssq[0]);
} };
__global__ void k_td_info(const double* __restrict__ st, const double* __restrict__ ssq, int64_t B, float* __restrict__ dinfo) { TdInfoOp::run(blockIdx.x, gridDim.x, st, ssq, B, dinfo); }
// sum of squares of up to two flat gradients (norm(grads), utils.jl:49-55: sqrt of the sum of per-tensor squared norms): 64 blocks of partial
// sums, combined in block order by the last block to arrive (deterministic: the combine order is fixed, only who performs it varies)
// Layer-0 gradient entries a fused pullback left as quarter partials (Sumsq2Fix, common.h): element i of flat gradient `slot` is formed here -- the additions Gemm16's
// split-K combine would have done, in its order -- and stored, by the one thread that sums its square.
// one slot of a Sumsq2Fix as scalars, picked with selects: indexing the struct's two-element arrays with a run-time slot put the whole struct into private memory (64 B of scratch
// per thread in every phase kernel, read back per gradient element by AdamSelfOp; round 6)
struct Sumsq2Slot { const float* part; int32_t out1, in0, woff, boff; float scale; };
__device__ __forceinline__ Sumsq2Slot ssq_slot(const Sumsq2Fix& fx, int slot) {      // (device side: compile-time slots only -- Sumsq2Op's 0 and 1)
  Sumsq2Slot s; const bool z = slot == 0;
  s.part = z ? fx.part[0] : fx.part[1]; s.out1 = z ? fx.out1[0] : fx.out1[1]; s.in0 = z ? fx.in0[0] : fx.in0[1]; s.woff = z ? fx.woff[0] : fx.woff[1]; s.boff = z ? fx.boff[0] : fx.boff[1];
  s.scale = z ? fx.scale[0] : fx.scale[1]; return s;
}
__device__ __forceinline__ float ssq_elem(float* g, int64_t i, const Sumsq2Slot& fs) {
  const float* part = fs.part;
  if (!part) return g[i];
  const int out1 = fs.out1, in0 = fs.in0, ps = in0 + 4; const int64_t w0 = fs.woff, b0 = fs.boff, qs = (int64_t)out1 * ps;
  if (i >= w0 && i < w0 + (int64_t)out1 * in0) { const int64_t e = i - w0; const int f = (int)(e % out1), qc = (int)(e / out1); const float* p = part + (int64_t)f * ps + qc;
    const float v = (((p[0] + p[qs]) + p[2 * qs]) + p[3 * qs]) * fs.scale; g[i] = v; return v; }
  if (i >= b0 && i < b0 + out1) { const int f = (int)(i - b0); const float* p = part + (int64_t)f * ps + in0; float R[4];
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) R[gg] = ((p[gg] + p[qs + gg]) + p[2 * qs + gg]) + p[3 * qs + gg];
    const float v = fs.scale * ((R[0] + R[1]) + (R[2] + R[3])); g[i] = v; return v; }
  return g[i];
}
__device__ __forceinline__ float ssq_elem(float* g, int64_t i, const Sumsq2Fix& fx, int slot) { return ssq_elem(g, i, ssq_slot(fx, slot)); }
struct Sumsq2Op { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, float* __restrict__ g1, int64_t n1, float* __restrict__ g2, int64_t n2, double* __restrict__ out /* [1..64] per-block partials; [0] is filled by ssq_finalize */, Sumsq2Fix fx) {
  __shared__ double red[4];
  double s = 0;
  for (int64_t i = (int64_t)bid_ * 256 + threadIdx.x; i < n1; i += (int64_t)SUMSQ_BLOCKS * 256) { const double v = (double)ssq_elem(g1, i, fx, 0); s += v * v; }
  for (int64_t i = (int64_t)bid_ * 256 + threadIdx.x; i < n2; i += (int64_t)SUMSQ_BLOCKS * 256) { const double v = (double)ssq_elem(g2, i, fx, 1); s += v * v; }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[1 + bid_] = ((red[0] + red[1]) + red[2]) + red[3];
} };
__global__ __launch_bounds__(256) void k_sumsq2(float* __restrict__ g1, int64_t n1, float* __restrict__ g2, int64_t n2, double* __restrict__ out /* [1..64] per-block partials; [0] is filled by ssq_finalize */, Sumsq2Fix fx) { Sumsq2Op::run(blockIdx.x, gridDim.x, g1, n1, g2, n2, out, fx); }
struct CriticInfoOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, const double* __restrict__ st1, const double* __restrict__ st2, const double* __restrict__ ssq, int64_t B, float* __restrict__ dinfo) { if (threadIdx.x != 0) return;
  ssq_finalize(ssq);
  dinfo[CRUX_INFO_LOSS] = (float)(0.5 * (st1[0] / (double)B) + 0.5 * (st2[0] / (double)B));
  dinfo[CRUX_INFO_Q1AVG] = (float)(st1[1] / (double)B); dinfo[CRUX_INFO_Q2AVG] = (float)(st2[1] / (double)B);
  dinfo[CRUX_INFO_GRAD_NORM] = (float)sqrt(ssq[0]);
} };
__global__ void k_critic_info(const double* __restrict__ st1, const double* __restrict__ st2, const double* __restrict__ ssq, int64_t B, float* __restrict__ dinfo) { CriticInfoOp::run(blockIdx.x, gridDim.x, st1, st2, ssq, B, dinfo); }
// sac_actor_loss head: which Q is the minimum, d(-mean(min Q))/dQ, loss statistics
struct ActorHeadOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, const float* __restrict__ q1, const float* __restrict__ q2, const float* __restrict__ lp, const float* __restrict__ log_alpha,
                                                    int64_t B, float* __restrict__ dy1, float* __restrict__ dy2, double* __restrict__ stats /* [2] */) {
  __shared__ double red[4];
  const float alpha = expf(log_alpha[0]), invB = 1.f / (float)B; double sl = 0, slp = 0;
  for (int64_t j = threadIdx.x; j < B; j += 256) { const bool second = q2[j] < q1[j]; const float mn = second ? q2[j] : q1[j];
    sl += (double)(alpha * lp[j] - mn); slp += (double)lp[j]; dy1[j] = second ? 0.f : -invB; dy2[j] = second ? -invB : 0.f; }
  sl = block_sum256(sl, red); slp = block_sum256(slp, red);
  if (threadIdx.x == 0) { stats[0] = sl; stats[1] = slp; }
} };
__global__ __launch_bounds__(256) void k_actor_head(const float* __restrict__ q1, const float* __restrict__ q2, const float* __restrict__ lp, const float* __restrict__ log_alpha,
                                                    int64_t B, float* __restrict__ dy1, float* __restrict__ dy2, double* __restrict__ stats /* [2] */) { ActorHeadOp::run(blockIdx.x, gridDim.x, q1, q2, lp, log_alpha, B, dy1, dy2, stats); }
// reverse pass through exploration(): mubar [ad x B] and the per-sample logSigma contributions [ad x B] (see the oracle for the accumulation order)
struct ActorGradOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, const float* __restrict__ sa, const float* __restrict__ mu, const float* __restrict__ eps, const float* __restrict__ ls,
                             const float* __restrict__ dsa1, const float* __restrict__ dsa2, const float* __restrict__ log_alpha, int od, int ad, int64_t B,
                             float* __restrict__ dmu, float* __restrict__ dls) {
  const int64_t i = (int64_t)bid_ * blockDim.x + threadIdx.x; if (i >= B * ad) return;
  const int64_t j = i / ad; const int d = (int)(i - j * ad);
  const float alpha = expf(log_alpha[0]); const float clp = alpha * (1.f / (float)B);
  const float sg = expf(ls[d]), s2 = sg * sg, a = sa[j * (od + ad) + od + d], df = a - mu[i];
  const float dq = dsa1[j * (od + ad) + od + d] + dsa2[j * (od + ad) + od + d];   // only the selected network's entry is non-zero
  const float abar = clp * (-(df / s2)) + dq;
  dmu[i] = clp * (df / s2) + abar;
  dls[i] = clp * ((df * df) / s2 - 1.f) + abar * (eps[i] * sg);
} };
__global__ void k_actor_grad(const float* __restrict__ sa, const float* __restrict__ mu, const float* __restrict__ eps, const float* __restrict__ ls,
                             const float* __restrict__ dsa1, const float* __restrict__ dsa2, const float* __restrict__ log_alpha, int od, int ad, int64_t B,
                             float* __restrict__ dmu, float* __restrict__ dls) { ActorGradOp::run(blockIdx.x, gridDim.x, sa, mu, eps, ls, dsa1, dsa2, log_alpha, od, ad, B, dmu, dls); }
struct RowsumOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, const float* __restrict__ v, int ad, int64_t B, float* __restrict__ out, int32_t* __restrict__ nf = nullptr) {   // out[d] = sum_j v[d + ad*j]; one block per d, fixed-order combine
  __shared__ float red[4];
  const int d = bid_; float acc = 0.f;
  for (int64_t j = threadIdx.x; j < B; j += 256) acc += v[d + (int64_t)ad * j];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) { const float r_ = ((red[0] + red[1]) + red[2]) + red[3]; out[d] = r_; if (nf && r_ != r_) atomicOr((int*)nf, 1); }
} };
__global__ __launch_bounds__(256) void k_rowsum(const float* __restrict__ v, int ad, int64_t B, float* __restrict__ out, int32_t* __restrict__ nf) { RowsumOp::run(blockIdx.x, gridDim.x, v, ad, B, out, nf); }
struct ActorInfoOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, const double* __restrict__ st, const double* __restrict__ ssq, int64_t B, float* __restrict__ dinfo) { if (threadIdx.x != 0) return;
  ssq_finalize(ssq);
  dinfo[CRUX_INFO_LOSS] = (float)(st[0] / (double)B); dinfo[CRUX_INFO_ENTROPY] = (float)(-(st[1] / (double)B)); dinfo[CRUX_INFO_GRAD_NORM] = (float)sqrt(ssq[0]);
} };
__global__ void k_actor_info(const double* __restrict__ st, const double* __restrict__ ssq, int64_t B, float* __restrict__ dinfo) { ActorInfoOp::run(blockIdx.x, gridDim.x, st, ssq, B, dinfo); }
// Flux.update!(Adam) gated on the gradient norm: NaN => parameters untouched, status set (training.jl:20)
struct AdamGatedOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, double* __restrict__ bp,
                                                    double eta, double b1, double b2, double eps, int64_t n, const double* __restrict__ ssq, int32_t* __restrict__ status, int advance, int from_partials) {
  // from_partials (fused executor): the 64 partials of k_sumsq2 are added here, in the order ssq_finalize adds them, so that the info op -- which writes ssq[0] --
  // can share this op's phase instead of preceding it by a barrier
  // The gate only asks whether the norm is NaN. The partials are sums of squares (no cancellation), so their sum is NaN exactly when one of them is: each lane
  // looks at its own partials and the wave votes -- instead of every thread adding all 64 in ssq_finalize's order (64 loads + 64 dependent Float64 adds per thread)
  bool bad;
  if (!from_partials) bad = isnan(ssq[0]);
  else { bool b_ = false; for (int k = (int)(threadIdx.x & 63); k < SUMSQ_BLOCKS; k += 64) b_ = b_ || isnan(ssq[1 + k]); bad = __ballot(b_) != 0ull; }
  if (status[0] == CRUX_ENAN) return;      // an earlier step of this launch sequence already stopped with "NaN detected!": no further updates (training.jl:20)
  if (bad) { if (bid_ == 0 && threadIdx.x == 0) status[0] = CRUX_ENAN; return; }
  const double c1 = 1.0 - bp[0], c2 = 1.0 - bp[1];
  for (int64_t i = (int64_t)bid_ * blockDim.x + threadIdx.x; i < n; i += (int64_t)nb_ * blockDim.x) {     // element-wise: any grid gives the same result
    const double gd = (double)g[i];
    const float mi = (float)(b1 * (double)m[i] + (1.0 - b1) * gd);
    const float vi = (float)(b2 * (double)v[i] + ((1.0 - b2) * gd) * gd);
    const float d = (float)((double)mi / c1 / (sqrt((double)vi / c2) + eps) * eta);
    m[i] = mi; v[i] = vi; p[i] = p[i] - d;
  }
  if (!advance) return;      // the fused executor advances the beta powers with AdamAdvanceOp in the next phase instead
  // the beta powers advance once every block has used them: the last block to FINISH (ticket in bp[2]) does it -- no second launch
  __syncthreads();
  if (threadIdx.x == 0) { __threadfence(); unsigned* ticket = (unsigned*)(bp + 2);
    if (atomicAdd(ticket, 1u) == nb_ - 1) { bp[0] *= b1; bp[1] *= b2; *ticket = 0u; } }
} };
__global__ __launch_bounds__(256) void k_adam_gated(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, double* __restrict__ bp,
                                                    double eta, double b1, double b2, double eps, int64_t n, const double* __restrict__ ssq, int32_t* __restrict__ status, int advance, int from_partials) { AdamGatedOp::run(blockIdx.x, gridDim.x, p, g, m, v, bp, eta, b1, b2, eps, n, ssq, status, advance, from_partials); }
struct AdamAdvanceOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, double* __restrict__ bp, double b1, double b2, const double* __restrict__ ssq) {
  if (bid_ == 0 && threadIdx.x == 0 && !isnan(ssq[0])) { bp[0] *= b1; bp[1] *= b2; }      // ssq[0] was written by the info op one phase earlier
} };

// ---- the self-gated form (round 4, the tile plans of exec.hip): Adam in the SAME phase as the norm ------------------------------------------------------------------
// training.jl:20 skips the update when the gradient norm is NaN -- exactly when a gradient element is NaN (the norm is a Float64 sum of squares: no overflow, no
// cancellation). The producers of the gradient raise flags[0] when a value they store is NaN (Gemm16's EPI_WGRAD epilogue, Wgrad2Op, RowsumOp) and flags[1] when a deferred
// layer-0 partial is not finite (Dgrad2W1Op: the final sum of four partials may then be NaN -- Inf - Inf -- without any partial being NaN). AdamSelfOp gates on flags[0] and,
// in the flags[1] case only, on the deferred sums themselves, which every block then forms and inspects; so the update needs no reduced norm and can run beside Sumsq2Op
// (which still forms the norm for the info row) instead of one dependent launch after it. Deferred layer-0 entries are formed here as Sumsq2Op forms them (ssq_elem: the same
// additions; both store the same value).
__device__ __forceinline__ bool adam_self_bad(const int32_t* __restrict__ flags, const Sumsq2Slot& fs) {
  bool bad = flags[0] != 0;
  if (flags[1] != 0 && fs.part) {      // rare: look at the sums
    const int out1 = fs.out1, in0 = fs.in0, ps = in0 + 4; const int64_t qs = (int64_t)out1 * ps; const float* part = fs.part; bool b_ = false;
    for (int64_t e = threadIdx.x; e < (int64_t)out1 * in0; e += blockDim.x) { const int f = (int)(e % out1), qc = (int)(e / out1); const float* p = part + (int64_t)f * ps + qc;
      const float v = (((p[0] + p[qs]) + p[2 * qs]) + p[3 * qs]) * fs.scale; b_ = b_ || v != v; }
    for (int f = threadIdx.x; f < out1; f += blockDim.x) { const float* p = part + (int64_t)f * ps + in0; float R[4];
#pragma unroll
      for (int gg = 0; gg < 4; ++gg) R[gg] = ((p[gg] + p[qs + gg]) + p[2 * qs + gg]) + p[3 * qs + gg];
      const float v = fs.scale * ((R[0] + R[1]) + (R[2] + R[3])); b_ = b_ || v != v; }
    bad = __syncthreads_or(b_ ? 1 : 0) != 0 || bad;
  }
  return bad;
}
struct AdamSelfOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, const double* __restrict__ bp,
                                                   double eta, double b1, double b2, double eps, int64_t n, const int32_t* __restrict__ flags, int32_t* __restrict__ status, Sumsq2Slot fs) {
  // (the slot of the Sumsq2Fix is resolved by the HOST when the op is recorded: a run-time index into the struct's two-element arrays -- and even a select between their
  //  elements, which the compiler turns back into an indexed load -- put the whole argument pack into private memory, 176 bytes per thread of every phase kernel; round 6)
  const bool bad = adam_self_bad(flags, fs);
  if (status[0] == CRUX_ENAN) return;      // an earlier step of this launch sequence already stopped with "NaN detected!" (training.jl:20)
  if (bad) { if (bid_ == 0 && threadIdx.x == 0) status[0] = CRUX_ENAN; return; }
  const double c1 = 1.0 - bp[0], c2 = 1.0 - bp[1];
  for (int64_t i = (int64_t)bid_ * blockDim.x + threadIdx.x; i < n; i += (int64_t)nb_ * blockDim.x) {
    const double gd = (double)ssq_elem(g, i, fs);
    const float mi = (float)(b1 * (double)m[i] + (1.0 - b1) * gd);
    const float vi = (float)(b2 * (double)v[i] + ((1.0 - b2) * gd) * gd);
    const float d = (float)((double)mi / c1 / (sqrt((double)vi / c2) + eps) * eta);
    m[i] = mi; v[i] = vi; p[i] = p[i] - d;
  }
} };
// the beta powers advance one phase later (every block of AdamSelfOp has used them), on the same evidence
struct AdamAdvanceSelfOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, double* __restrict__ bp, double b1, double b2, const int32_t* __restrict__ flags, Sumsq2Slot fs) {
  const bool bad = adam_self_bad(flags, fs);
  if (bid_ == 0 && threadIdx.x == 0 && !bad) { bp[0] *= b1; bp[1] *= b2; }
} };
static int32_t adam_self(crux_mlp* n, const int32_t* d_flags, int32_t* d_status, const Sumsq2Fix& fx, int slot) {      // recorded sequences only
  crux_ctx* c = n->ctx;
  if (!n->has_adam) return crux_fail(c, CRUX_EINVAL, "train!: crux_adam_init was not called on this handle");
  if (!crux_exec_recording(c)) return crux_fail(c, CRUX_EHIP, "adam_self: outside a recording");
  const int64_t cnt = n->nd.n_params; const unsigned nbk = (unsigned)((cnt + 255) / 256);
  const Sumsq2Slot fs{fx.part[slot], fx.out1[slot], fx.in0[slot], fx.woff[slot], fx.boff[slot], fx.scale[slot]};
  crux_exec_push<AdamSelfOp, OP_ADAM_SELF>(c, nbk < 64u ? nbk : 64u, n->p, n->g, n->m, n->v, (const double*)n->bp, n->eta, n->b1, n->b2, n->eps, cnt, d_flags, d_status, fs);
  crux_exec_push<AdamAdvanceSelfOp, OP_ADAM_ADVANCE_SELF>(c, 1u, n->bp, n->b1, n->b2, d_flags, fs);
  return CRUX_OK;
}

#include "sac_fused.h"

// ---- OnPolicyGAIL pieces (src/model_free/il/on_policy_gail.jl:1-5,49-54; src/extras/gans.jl:7-9) ---------------------------------------
// vcat(a, s) of buffer rows [off, off + n): the ACTION first (D(x, y) convention, on_policy_gail.jl:50); one-hot Bool actions become 0/1
struct ConcatAsOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, const void* __restrict__ a, int a_is_u8, const float* __restrict__ s, int od, int ad, int64_t off, int64_t n, float* __restrict__ out) {
  const int64_t i = (int64_t)bid_ * blockDim.x + threadIdx.x; if (i >= n * (od + ad)) return;
  const int64_t j = i / (od + ad); const int k = (int)(i - j * (od + ad)); const int64_t row = off + j;
  out[i] = k < ad ? (a_is_u8 ? (((const uint8_t*)a)[row * ad + k] ? 1.f : 0.f) : ((const float*)a)[row * ad + k]) : s[row * od + (k - ad)];
} };
__global__ void k_concat_as(const void* __restrict__ a, int a_is_u8, const float* __restrict__ s, int od, int ad, int64_t off, int64_t n, float* __restrict__ out) { ConcatAsOp::run(blockIdx.x, gridDim.x, a, a_is_u8, s, od, ad, off, n, out); }
__device__ __forceinline__ float logsigmoid_f(float x) { const float nx = -x; return -(log1pf(expf(-fabsf(nx))) + (nx > 0.f ? nx : 0.f)); }   // NNlib: -softplus(-x)
// logitbinarycrossentropy heads of the two halves of the batch: columns [0, n_ex) carry label 1, [n_ex, n_ex + n_pi) label 0
struct GailHeadOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, const float* __restrict__ z, int64_t n_ex, int64_t n_pi, float* __restrict__ dz, double* __restrict__ stats /* [2] */) {
  __shared__ double red[4];
  double le = 0, lp = 0;
  for (int64_t j = threadIdx.x; j < n_ex + n_pi; j += 256) { const float v = z[j]; const float ls = logsigmoid_f(v); const float sg = 1.f / (1.f + expf(-v));
    if (j < n_ex) { le += (double)(-ls); dz[j] = (sg - 1.f) / (float)n_ex; } else { lp += (double)(v - ls); dz[j] = sg / (float)n_pi; } }
  le = block_sum256(le, red); lp = block_sum256(lp, red);
  if (threadIdx.x == 0) { stats[0] = le; stats[1] = lp; }
} };
__global__ __launch_bounds__(256) void k_gail_head(const float* __restrict__ z, int64_t n_ex, int64_t n_pi, float* __restrict__ dz, double* __restrict__ stats /* [2] */) { GailHeadOp::run(blockIdx.x, gridDim.x, z, n_ex, n_pi, dz, stats); }
struct GailInfoOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, const double* __restrict__ st, const double* __restrict__ ssq, int64_t n_ex, int64_t n_pi, float* __restrict__ dinfo) { if (threadIdx.x != 0) return;
  ssq_finalize(ssq);
  dinfo[CRUX_INFO_LOSS] = (float)(st[0] / (double)n_ex) + (float)(st[1] / (double)n_pi); dinfo[CRUX_INFO_GRAD_NORM] = (float)sqrt(ssq[0]);
} };
__global__ void k_gail_info(const double* __restrict__ st, const double* __restrict__ ssq, int64_t n_ex, int64_t n_pi, float* __restrict__ dinfo) { GailInfoOp::run(blockIdx.x, gridDim.x, st, ssq, n_ex, n_pi, dinfo); }
struct GailRewardOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, const float* __restrict__ z, int64_t n, float alpha_r, float rscale, float* __restrict__ r, double* __restrict__ partial) {
  __shared__ double red[4];
  double s = 0;
  for (int64_t j = (int64_t)bid_ * 256 + threadIdx.x; j < n; j += (int64_t)nb_ * 256) { const float v = z[j]; const float ls = logsigmoid_f(v), lc = ls - v;
    const float rr = alpha_r * ls - (1.f - alpha_r) * lc; s += (double)rr; r[j] = rr * rscale; }
  s = block_sum256(s, red);
  if (threadIdx.x == 0) partial[bid_] = s;
} };
__global__ __launch_bounds__(256) void k_gail_reward(const float* __restrict__ z, int64_t n, float alpha_r, float rscale, float* __restrict__ r, double* __restrict__ partial) { GailRewardOp::run(blockIdx.x, gridDim.x, z, n, alpha_r, rscale, r, partial); }

// partials: the squared norm arrives as k_sumsq2's per-block partials in d_ssq[1..] (finalised by the info op into d_ssq[0]); false: d_ssq[0] was written directly
static int32_t adam_gated(crux_mlp* n, const double* d_ssq, int32_t* d_status, bool partials = true, hipStream_t strm = nullptr) {
  crux_ctx* c = n->ctx; if (!strm) strm = c->stream;
  if (!n->has_adam) return crux_fail(c, CRUX_EINVAL, "train!: crux_adam_init was not called on this handle");
  const int64_t cnt = n->nd.n_params;
  if (crux_exec_recording(c)) {      // fused sequence: at most one round of blocks, beta powers advanced by a one-thread op of the next phase (no device-scope fence per block)
    const unsigned nbk = (unsigned)((cnt + 255) / 256);
    crux_exec_push<AdamGatedOp, OP_ADAM_GATED>(c, nbk < 64u ? nbk : 64u, n->p, (const float*)n->g, n->m, n->v, n->bp, n->eta, n->b1, n->b2, n->eps, cnt, d_ssq, d_status, 0, partials ? 1 : 0);
    crux_exec_push<AdamAdvanceOp, OP_ADAM_ADVANCE>(c, 1u, n->bp, n->b1, n->b2, d_ssq);
    return CRUX_OK;
  }
  CRUX_RUN(c, AdamGatedOp, OP_ADAM_GATED, k_adam_gated, (unsigned)((cnt + 255) / 256), 256, strm, n->p, n->g, n->m, n->v, n->bp, n->eta, n->b1, n->b2, n->eps, cnt, d_ssq, d_status, 1, 0);
  return crux_launch_check(c, "k_adam_gated");
}

// scratch carve-up: one crux_scratch block per call
struct Carve { char* p; size_t off; template <class T> T* take(size_t n) { T* r = (T*)(p + off); off += ((n * sizeof(T) + 255) / 256) * 256; return r; } };
static inline unsigned nblk(int64_t n) { return (unsigned)((n + 255) / 256); }
// info row / statistics / status words of one step: in a recorded (fused) sequence they must survive until the read-back at the end of the launch, so they
// come from the executor's own region instead of the scratch block the next piece will carve again
static inline Carve small_carve(crux_ctx* c, Carve& cv, size_t bytes) {
  if (crux_exec_recording(c)) return Carve{(char*)crux_exec_small(c, bytes), 0};
  Carve sv{cv.p + cv.off, 0}; cv.off += bytes; return sv;
}

static int32_t check_sac(crux_ctx* c, crux_mlp* actor, crux_mlp* q1, crux_mlp* q2, crux_mlp* la, crux_buffer* b, const char* who) {
  if (actor && actor->squash > 0.f) return crux_fail(c, CRUX_EUNSUP, "%s: SquashedGaussianPolicy actors are implemented for the on-policy learners (PPO / A2C / REINFORCE / BC) only", who);
  const int od = b->obs_dim, ad = b->act_dim;
  if (b->elements < 1) return crux_fail(c, CRUX_EINVAL, "%s: empty batch", who);
  if (b->act_kind != CRUX_ACTION_CONTINUOUS) return crux_fail(c, CRUX_EINVAL, "%s: needs a continuous action column", who);
  if (actor && (actor->nd.L < 1 || actor->nd.dims[0] != od || actor->nd.dims[actor->nd.L] != ad || actor->nd.n_extra != ad || ad > 64))
    return crux_fail(c, CRUX_EINVAL, "%s: actor must be a GaussianPolicy handle %d -> %d with %d logSigma extras", who, od, ad, ad);
  crux_mlp* qs[2] = {q1, q2};
  for (int t = 0; t < 2; ++t) if (qs[t] && (qs[t]->nd.L < 1 || qs[t]->nd.dims[0] != od + ad || qs[t]->nd.dims[qs[t]->nd.L] != 1))
    return crux_fail(c, CRUX_EINVAL, "%s: Q%d must map vcat(s, a) (%d) -> 1", who, t + 1, od + ad);
  if (la && la->nd.n_params < 1) return crux_fail(c, CRUX_EINVAL, "%s: log_alpha handle has no parameters", who);
  return CRUX_OK;
}
static int32_t finish_step(crux_ctx* c, const float* d_info, const int32_t* d_status, float* info_out, const char* who) {
  if (crux_exec_recording(c)) { crux_exec_add_readback(c, info_out, d_info, d_status, who); return CRUX_OK; }   // fulfilled by crux_exec_run after the fused launch
  float* h = (float*)crux_pinned(c, sizeof(float) * CRUX_INFO_N + 16); if (!h) return crux_fail(c, CRUX_ENOMEM, "%s: pinned staging", who);
  HIPCHK(c, hipMemcpyAsync(h, d_info, sizeof(float) * CRUX_INFO_N, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(h + CRUX_INFO_N, d_status, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (info_out) memcpy(info_out, h, sizeof(float) * CRUX_INFO_N);
  int32_t st; memcpy(&st, h + CRUX_INFO_N, sizeof st);
  if (st == CRUX_ENAN) return crux_fail(c, CRUX_ENAN, "NaN detected! (grad norm is NaN, src/training.jl:20) in %s", who);
  return CRUX_OK;
}

// train!(critic, td_loss) for wide DiscreteNetwork critics (C3: 8-256-256-4) on the dense engine; crux_td_step (train.hip) routes here.
int32_t crux_td_step_dense(crux_mlp* net, crux_buffer* b, const float* d_y, int32_t use_weight, float* info_out, float* d_err) {
  crux_ctx* c = net->ctx; const int64_t B = b->elements; const int nout = net->nd.dims[net->nd.L];
  Carve cv{(char*)crux_scratch(c, 4 * (size_t)B * nout + 8192), 0}; if (!cv.p) return crux_fail(c, CRUX_ENOMEM, "td_step: scratch");
  float* dy = cv.take<float>((size_t)B * nout); Carve sv = small_carve(c, cv, 256 * 6); if (!sv.p) return crux_fail(c, CRUX_ENOMEM, "td_step: executor region");
  float* dinfo = sv.take<float>(CRUX_INFO_N); double* st = sv.take<double>(2); double* ssq = sv.take<double>(2 + SUMSQ_BLOCKS); int32_t* status = sv.take<int32_t>(1);
  { const int32_t rz = crux_exec_zero(c, dinfo, 256 * 6, c->stream); if (rz) return rz; }
  const float* S = (const float*)b->col[CRUX_COL_S]; const float* w = use_weight ? (const float*)b->col[CRUX_COL_WEIGHT] : nullptr;
  int32_t rc = crux_dense_forward(net, S, B, c->stream); if (rc) return rc;
  CRUX_RUN(c, TdHeadOp, OP_TD_HEAD, k_td_head, 1, 256, c->stream, crux_dense_act(net, net->nd.L), (const uint8_t*)b->col[CRUX_COL_A], nout, d_y, w, B, dy, st, d_err);
  Sumsq2Fix fx{};
  rc = crux_dense_backward(net, S, B, dy, 1.0f, true, nullptr, c->stream, &fx, 0); if (rc) return rc;
  CRUX_RUN(c, Sumsq2Op, OP_SUMSQ2, k_sumsq2, SUMSQ_BLOCKS, 256, c->stream, net->g, (int64_t)net->nd.n_params, (float*)nullptr, (int64_t)0, ssq, fx);
  CRUX_RUN(c, TdInfoOp, OP_TD_INFO, k_td_info, 1, 1, c->stream, st, ssq, B, dinfo);
  rc = adam_gated(net, ssq, status); if (rc) return rc;
  return finish_step(c, dinfo, status, info_out, "td_loss");
}

extern "C" {

int32_t crux_sac_target(crux_mlp* actor, crux_mlp* q1t, crux_mlp* q2t, crux_mlp* la, crux_buffer* b, float gamma, uint64_t seed, uint64_t counter, float* d_y) {
  if (!actor || !q1t || !q2t || !la || !b || !d_y) return CRUX_EINVAL;
  crux_ctx* c = actor->ctx; int32_t rc = check_sac(c, actor, q1t, q2t, la, b, "sac_target"); if (rc) return rc;
  const int64_t B = b->elements; const int od = b->obs_dim, ad = b->act_dim;
  Carve cv{(char*)crux_scratch(c, 4 * (size_t)B * (od + ad + 1) + 1024), 0}; if (!cv.p) return crux_fail(c, CRUX_ENOMEM, "sac_target: scratch");
  float* sa = cv.take<float>((size_t)B * (od + ad)); float* lp = cv.take<float>((size_t)B);
  const float* SP = (const float*)b->col[CRUX_COL_SP];
  rc = crux_dense_forward(actor, SP, B, c->stream); if (rc) return rc;
  CRUX_RUN(c, GaussExploreOp, OP_GAUSS_EXPLORE, k_gauss_explore, nblk(B), 256, c->stream, crux_dense_act(actor, actor->nd.L), actor->p + actor->nd.xoff, SP, od, ad, B, seed, counter, sa, lp, (float*)nullptr);
  rc = crux_dense_forward(q1t, sa, B, c->stream); if (rc) return rc;
  rc = crux_dense_forward(q2t, sa, B, c->stream); if (rc) return rc;
  CRUX_RUN(c, SacTargetOp, OP_SAC_TARGET, k_sac_target, nblk(B), 256, c->stream, (const float*)b->col[CRUX_COL_R], (const uint8_t*)b->col[CRUX_COL_DONE], crux_dense_act(q1t, q1t->nd.L), crux_dense_act(q2t, q2t->nd.L), lp, la->p, gamma, B, d_y);
  return crux_launch_check(c, "sac_target");
}

int32_t crux_sac_temp_step(crux_mlp* actor, crux_mlp* la, crux_buffer* b, float H_target, uint64_t seed, uint64_t counter, float* info_out) {
  if (!actor || !la || !b) return CRUX_EINVAL;
  crux_ctx* c = actor->ctx; int32_t rc = check_sac(c, actor, nullptr, nullptr, la, b, "sac_temp_loss"); if (rc) return rc;
  const int64_t B = b->elements; const int od = b->obs_dim, ad = b->act_dim;
  Carve cv{(char*)crux_scratch(c, 4 * (size_t)B + 4096), 0}; if (!cv.p) return crux_fail(c, CRUX_ENOMEM, "sac_temp: scratch");
  float* lp = cv.take<float>((size_t)B); Carve sv = small_carve(c, cv, 256 * 5); if (!sv.p) return crux_fail(c, CRUX_ENOMEM, "sac_temp: executor region");
  float* dinfo = sv.take<float>(CRUX_INFO_N); double* ssq = sv.take<double>(2 + SUMSQ_BLOCKS); int32_t* st = sv.take<int32_t>(1);
  { const int32_t rz = crux_exec_zero(c, dinfo, 256 * 5, c->stream); if (rz) return rz; }
  const float* S = (const float*)b->col[CRUX_COL_S];
  rc = crux_dense_forward(actor, S, B, c->stream); if (rc) return rc;
  CRUX_RUN(c, GaussExploreOp, OP_GAUSS_EXPLORE, k_gauss_explore, nblk(B), 256, c->stream, crux_dense_act(actor, actor->nd.L), actor->p + actor->nd.xoff, S, od, ad, B, seed, counter, (float*)nullptr, lp, (float*)nullptr);
  { const int32_t rz = crux_exec_zero(c, la->g, sizeof(float) * (size_t)la->nd.n_params, c->stream); if (rz) return rz; }
  CRUX_RUN(c, TempHeadOp, OP_TEMP_HEAD, k_temp_head, 1, 256, c->stream, lp, B, H_target, la->p, la->g, dinfo, ssq);
  rc = adam_gated(la, ssq, st, false); if (rc) return rc;
  return finish_step(c, dinfo, st, info_out, "sac_temp_loss");
}

static int32_t q_step_impl(crux_mlp* q1, crux_mlp* q2, crux_buffer* b, const float* d_y, int32_t use_weight, float* info_out, const char* who) {
  crux_ctx* c = q1->ctx; int32_t rc = check_sac(c, nullptr, q1, q2, nullptr, b, who); if (rc) return rc;
  if (use_weight && !has_col(b, CRUX_COL_WEIGHT)) return crux_fail(c, CRUX_EINVAL, "%s(weight=:weight): batch has no :weight column", who);
  const int64_t B = b->elements; const int od = b->obs_dim, ad = b->act_dim; const int nq = q2 ? 2 : 1;
  Carve cv{(char*)crux_scratch(c, 4 * (size_t)B * (od + ad + 2) + 8192), 0}; if (!cv.p) return crux_fail(c, CRUX_ENOMEM, "%s: scratch", who);
  float* sa = cv.take<float>((size_t)B * (od + ad)); float* dys[2] = {cv.take<float>((size_t)B), cv.take<float>((size_t)B)};      // one dL/dQ per critic: the two chains may run side by side (exec.hip)
  Carve sv = small_carve(c, cv, 256 * 7); if (!sv.p) return crux_fail(c, CRUX_ENOMEM, "%s: executor region", who);
  float* dinfo = sv.take<float>(CRUX_INFO_N);
  double* st1 = sv.take<double>(2); double* st2 = sv.take<double>(2); double* ssq = sv.take<double>(2 + SUMSQ_BLOCKS); int32_t* st = sv.take<int32_t>(1);
  { const int32_t rz = crux_exec_zero(c, dinfo, 256 * 7, c->stream); if (rz) return rz; }
  const float* w = use_weight ? (const float*)b->col[CRUX_COL_WEIGHT] : nullptr;
  CRUX_RUN(c, ConcatSaOp, OP_CONCAT_SA, k_concat_sa, nblk(B * (od + ad)), 256, c->stream, (const float*)b->col[CRUX_COL_S], (const float*)b->col[CRUX_COL_A], od, ad, B, sa);
  crux_mlp* qs[2] = {q1, q2}; double* sts[2] = {st1, st2}; Sumsq2Fix fx{};
  for (int t = 0; t < nq; ++t) {
    rc = crux_dense_forward(qs[t], sa, B, c->stream); if (rc) return rc;
    CRUX_RUN(c, QHeadOp, OP_Q_HEAD, k_q_head, 1, 256, c->stream, crux_dense_act(qs[t], qs[t]->nd.L), d_y, w, B, nq == 2 ? 0.5f : 1.0f, dys[t], sts[t]);
    rc = crux_dense_backward(qs[t], sa, B, dys[t], 1.0f, true, nullptr, c->stream, &fx, t); if (rc) return rc;
  }
  CRUX_RUN(c, Sumsq2Op, OP_SUMSQ2, k_sumsq2, SUMSQ_BLOCKS, 256, c->stream, q1->g, (int64_t)q1->nd.n_params, q2 ? q2->g : (float*)nullptr, (int64_t)(q2 ? q2->nd.n_params : 0), ssq, fx);
  CRUX_RUN(c, CriticInfoOp, OP_CRITIC_INFO, k_critic_info, 1, 1, c->stream, st1, nq == 2 ? st2 : st1, ssq, B, dinfo);   // single Q: 0.5 l + 0.5 l = l
  rc = adam_gated(q1, ssq, st); if (rc) return rc;
  if (q2) { rc = adam_gated(q2, ssq, st); if (rc) return rc; }
  return finish_step(c, dinfo, st, info_out, who);
}
int32_t crux_double_q_step(crux_mlp* q1, crux_mlp* q2, crux_buffer* b, const float* d_y, int32_t use_weight, float* info_out) {
  if (!q1 || !q2 || !b || !d_y) return CRUX_EINVAL;
  return q_step_impl(q1, q2, b, d_y, use_weight, info_out, "double_Q_loss");
}
int32_t crux_q_step(crux_mlp* q, crux_buffer* b, const float* d_y, int32_t use_weight, float* info_out) {
  if (!q || !b || !d_y) return CRUX_EINVAL;
  return q_step_impl(q, nullptr, b, d_y, use_weight, info_out, "td_loss");
}

int32_t crux_gail_d_step(crux_mlp* D, crux_buffer* ex, int64_t off_ex, int64_t n_ex, crux_buffer* pi, int64_t off_pi, int64_t n_pi, float* info_out) {
  if (!D || !ex || !pi) return CRUX_EINVAL;
  crux_ctx* c = D->ctx; const int od = ex->obs_dim, ad = ex->act_dim, sd = od + ad;
  if (n_ex <= 0 || n_pi <= 0 || off_ex < 0 || off_pi < 0 || off_ex + n_ex > ex->elements || off_pi + n_pi > pi->elements) return crux_fail(c, CRUX_EINVAL, "gail_d_loss: row ranges outside the buffers");
  if (pi->obs_dim != od || pi->act_dim != ad || pi->act_kind != ex->act_kind) return crux_fail(c, CRUX_EINVAL, "gail_d_loss: expert and policy buffers differ in shape");
  if (D->nd.L < 1 || D->nd.dims[0] != sd || D->nd.dims[D->nd.L] != 1) return crux_fail(c, CRUX_EINVAL, "gail_d_loss: discriminator must map vcat(a, s) (%d) -> 1", sd);
  const int64_t B = n_ex + n_pi;
  Carve cv{(char*)crux_scratch(c, 4 * (size_t)B * (sd + 1) + 8192), 0}; if (!cv.p) return crux_fail(c, CRUX_ENOMEM, "gail_d_loss: scratch");
  float* x = cv.take<float>((size_t)B * sd); float* dz = cv.take<float>((size_t)B); float* dinfo = cv.take<float>(CRUX_INFO_N);
  double* st2 = cv.take<double>(2); double* ssq = cv.take<double>(2 + SUMSQ_BLOCKS); int32_t* st = cv.take<int32_t>(1);
  HIPCHK(c, hipMemsetAsync(dinfo, 0, 256 * 5, c->stream));
  const int u8 = ex->act_kind == CRUX_ACTION_DISCRETE ? 1 : 0;
  hipLaunchKernelGGL(k_concat_as, dim3(nblk(n_ex * sd)), dim3(256), 0, c->stream, (const void*)ex->col[CRUX_COL_A], u8, (const float*)ex->col[CRUX_COL_S], od, ad, off_ex, n_ex, x);
  hipLaunchKernelGGL(k_concat_as, dim3(nblk(n_pi * sd)), dim3(256), 0, c->stream, (const void*)pi->col[CRUX_COL_A], u8, (const float*)pi->col[CRUX_COL_S], od, ad, off_pi, n_pi, x + (size_t)n_ex * sd);
  int32_t rc = crux_dense_forward(D, x, B, c->stream); if (rc) return rc;
  hipLaunchKernelGGL(k_gail_head, dim3(1), dim3(256), 0, c->stream, crux_dense_act(D, D->nd.L), n_ex, n_pi, dz, st2);
  rc = crux_dense_backward(D, x, B, dz, 1.0f, true, nullptr, c->stream); if (rc) return rc;
  CRUX_RUN(c, Sumsq2Op, OP_SUMSQ2, k_sumsq2, SUMSQ_BLOCKS, 256, c->stream, D->g, (int64_t)D->nd.n_params, (float*)nullptr, (int64_t)0, ssq, Sumsq2Fix{});
  hipLaunchKernelGGL(k_gail_info, dim3(1), dim3(1), 0, c->stream, st2, ssq, n_ex, n_pi, dinfo);
  rc = adam_gated(D, ssq, st); if (rc) return rc;
  return finish_step(c, dinfo, st, info_out, "gail_d_loss");
}

int32_t crux_gail_reward(crux_mlp* D, crux_buffer* b, float alpha_r, float rscale, float* mean_r) {
  if (!D || !b) return CRUX_EINVAL;
  crux_ctx* c = D->ctx; const int od = b->obs_dim, ad = b->act_dim, sd = od + ad; const int64_t n = b->elements;
  if (n <= 0) return crux_fail(c, CRUX_EINVAL, "GAIL reward: empty buffer");
  if (D->nd.L < 1 || D->nd.dims[0] != sd || D->nd.dims[D->nd.L] != 1) return crux_fail(c, CRUX_EINVAL, "GAIL reward: discriminator must map vcat(a, s) (%d) -> 1", sd);
  const int nb = 64;
  Carve cv{(char*)crux_scratch(c, 4 * (size_t)n * sd + 4096), 0}; if (!cv.p) return crux_fail(c, CRUX_ENOMEM, "GAIL reward: scratch");
  float* x = cv.take<float>((size_t)n * sd); double* part = cv.take<double>(nb);
  hipLaunchKernelGGL(k_concat_as, dim3(nblk(n * sd)), dim3(256), 0, c->stream, (const void*)b->col[CRUX_COL_A], b->act_kind == CRUX_ACTION_DISCRETE ? 1 : 0, (const float*)b->col[CRUX_COL_S], od, ad, (int64_t)0, n, x);
  int32_t rc = crux_dense_forward(D, x, n, c->stream); if (rc) return rc;
  hipLaunchKernelGGL(k_gail_reward, dim3(nb), dim3(256), 0, c->stream, crux_dense_act(D, D->nd.L), n, alpha_r, rscale, (float*)b->col[CRUX_COL_R], part);
  rc = crux_launch_check(c, "k_gail_reward"); if (rc) return rc;
  double h[64];
  HIPCHK(c, hipMemcpyAsync(h, part, sizeof h, hipMemcpyDeviceToHost, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream));
  double s = 0; for (int k = 0; k < nb; ++k) s += h[k];
  if (mean_r) *mean_r = (float)(s / (double)n);
  return CRUX_OK;
}

int32_t crux_dpg_target(crux_mlp* actor_t, crux_mlp* q1t, crux_mlp* q2t, crux_buffer* b, float gamma, float sigma, float eps_min, float eps_max, float a_min, float a_max,
                        uint64_t seed, uint64_t counter, float* d_y) {
  if (!actor_t || !q1t || !b || !d_y) return CRUX_EINVAL;
  crux_ctx* c = actor_t->ctx; int32_t rc = check_sac(c, nullptr, q1t, q2t, nullptr, b, "ddpg_target"); if (rc) return rc;
  const int64_t B = b->elements; const int od = b->obs_dim, ad = b->act_dim;
  if (actor_t->nd.L < 1 || actor_t->nd.dims[0] != od || actor_t->nd.dims[actor_t->nd.L] != ad) return crux_fail(c, CRUX_EINVAL, "ddpg_target: actor must map %d -> %d", od, ad);
  Carve cv{(char*)crux_scratch(c, 4 * (size_t)B * (od + ad) + 1024), 0}; if (!cv.p) return crux_fail(c, CRUX_ENOMEM, "ddpg_target: scratch");
  float* sa = cv.take<float>((size_t)B * (od + ad));
  const float* SP = (const float*)b->col[CRUX_COL_SP];
  rc = crux_dense_forward(actor_t, SP, B, c->stream); if (rc) return rc;
  CRUX_RUN(c, DpgActionOp, OP_DPG_ACTION, k_dpg_action, nblk(B * (od + ad)), 256, c->stream, crux_dense_act(actor_t, actor_t->nd.L), SP, od, ad, B, sigma, eps_min, eps_max, a_min, a_max, seed, counter, sa);
  rc = crux_dense_forward(q1t, sa, B, c->stream); if (rc) return rc;
  if (q2t) { rc = crux_dense_forward(q2t, sa, B, c->stream); if (rc) return rc; }
  CRUX_RUN(c, DpgTargetOp, OP_DPG_TARGET, k_dpg_target, nblk(B), 256, c->stream, (const float*)b->col[CRUX_COL_R], (const uint8_t*)b->col[CRUX_COL_DONE], crux_dense_act(q1t, q1t->nd.L), q2t ? crux_dense_act(q2t, q2t->nd.L) : (const float*)nullptr, gamma, B, d_y);
  return crux_launch_check(c, "ddpg_target");
}

int32_t crux_dpg_actor_step(crux_mlp* actor, crux_mlp* q, crux_buffer* b, float* info_out) {
  if (!actor || !q || !b) return CRUX_EINVAL;
  crux_ctx* c = actor->ctx; int32_t rc = check_sac(c, nullptr, q, nullptr, nullptr, b, "ddpg_actor_loss"); if (rc) return rc;
  const int64_t B = b->elements; const int od = b->obs_dim, ad = b->act_dim, sd = od + ad;
  if (actor->nd.L < 1 || actor->nd.dims[0] != od || actor->nd.dims[actor->nd.L] != ad) return crux_fail(c, CRUX_EINVAL, "ddpg_actor_loss: actor must map %d -> %d", od, ad);
  Carve cv{(char*)crux_scratch(c, 4 * (size_t)B * (2 * sd + ad + 1) + 8192), 0}; if (!cv.p) return crux_fail(c, CRUX_ENOMEM, "ddpg_actor: scratch");
  float* sa = cv.take<float>((size_t)B * sd); float* dsa = cv.take<float>((size_t)B * sd); float* da = cv.take<float>((size_t)B * ad); float* dy = cv.take<float>((size_t)B);
  Carve sv = small_carve(c, cv, 256 * 5); if (!sv.p) return crux_fail(c, CRUX_ENOMEM, "ddpg_actor: executor region");
  float* dinfo = sv.take<float>(CRUX_INFO_N); double* ssq = sv.take<double>(2 + SUMSQ_BLOCKS); int32_t* st = sv.take<int32_t>(1);
  { const int32_t rz = crux_exec_zero(c, dinfo, 256 * 5, c->stream); if (rz) return rz; }
  const float* S = (const float*)b->col[CRUX_COL_S];
  rc = crux_dense_forward(actor, S, B, c->stream); if (rc) return rc;
  CRUX_RUN(c, DpgActionOp, OP_DPG_ACTION, k_dpg_action, nblk(B * sd), 256, c->stream, crux_dense_act(actor, actor->nd.L), S, od, ad, B, -1.f, 0.f, 0.f, 0.f, 0.f, (uint64_t)0, (uint64_t)0, sa);
  rc = crux_dense_forward(q, sa, B, c->stream); if (rc) return rc;
  CRUX_RUN(c, FillOp, OP_FILL, k_fill, nblk(B), 256, c->stream, dy, -1.f / (float)B, B);                    // d(-mean(Q))/dQ
  rc = crux_dense_backward(q, sa, B, dy, 1.0f, false, dsa, c->stream); if (rc) return rc;                        // the critic's parameters are not trained here
  CRUX_RUN(c, SliceRowsOp, OP_SLICE_ROWS, k_slice_rows, nblk(B * ad), 256, c->stream, dsa, sd, od, ad, B, da);
  Sumsq2Fix fx{};
  rc = crux_dense_backward(actor, S, B, da, 1.0f, true, nullptr, c->stream, &fx, 0); if (rc) return rc;
  CRUX_RUN(c, Sumsq2Op, OP_SUMSQ2, k_sumsq2, SUMSQ_BLOCKS, 256, c->stream, actor->g, (int64_t)actor->nd.n_params, (float*)nullptr, (int64_t)0, ssq, fx);
  CRUX_RUN(c, MeanInfoOp, OP_MEAN_INFO, k_mean_info, 1, 256, c->stream, crux_dense_act(q, q->nd.L), B, -1.f, ssq, dinfo);
  rc = adam_gated(actor, ssq, st); if (rc) return rc;
  return finish_step(c, dinfo, st, info_out, "ddpg_actor_loss");
}

int32_t crux_sac_actor_step(crux_mlp* actor, crux_mlp* q1, crux_mlp* q2, crux_mlp* la, crux_buffer* b, uint64_t seed, uint64_t counter, float* info_out) {
  if (!actor || !q1 || !q2 || !la || !b) return CRUX_EINVAL;
  crux_ctx* c = actor->ctx; int32_t rc = check_sac(c, actor, q1, q2, la, b, "sac_actor_loss"); if (rc) return rc;
  const int64_t B = b->elements; const int od = b->obs_dim, ad = b->act_dim, sd = od + ad;
  Carve cv{(char*)crux_scratch(c, 4 * (size_t)B * (3 * sd + 3 * ad + 3) + 16384), 0}; if (!cv.p) return crux_fail(c, CRUX_ENOMEM, "sac_actor: scratch");
  float* sa = cv.take<float>((size_t)B * sd); float* dsa1 = cv.take<float>((size_t)B * sd); float* dsa2 = cv.take<float>((size_t)B * sd);
  float* eps = cv.take<float>((size_t)B * ad); float* dmu = cv.take<float>((size_t)B * ad); float* dls = cv.take<float>((size_t)B * ad);
  float* lp = cv.take<float>((size_t)B); float* dy1 = cv.take<float>((size_t)B); float* dy2 = cv.take<float>((size_t)B);
  Carve sv = small_carve(c, cv, 256 * 6); if (!sv.p) return crux_fail(c, CRUX_ENOMEM, "sac_actor: executor region");
  float* dinfo = sv.take<float>(CRUX_INFO_N); double* stats = sv.take<double>(2); double* ssq = sv.take<double>(2 + SUMSQ_BLOCKS); int32_t* st = sv.take<int32_t>(1);
  { const int32_t rz = crux_exec_zero(c, dinfo, 256 * 6, c->stream); if (rz) return rz; }
  const float* S = (const float*)b->col[CRUX_COL_S];
  rc = crux_dense_forward(actor, S, B, c->stream); if (rc) return rc;
  float* mu = crux_dense_act(actor, actor->nd.L);
  CRUX_RUN(c, GaussExploreOp, OP_GAUSS_EXPLORE, k_gauss_explore, nblk(B), 256, c->stream, mu, actor->p + actor->nd.xoff, S, od, ad, B, seed, counter, sa, lp, eps);
  rc = crux_dense_forward(q1, sa, B, c->stream); if (rc) return rc;
  rc = crux_dense_forward(q2, sa, B, c->stream); if (rc) return rc;
  CRUX_RUN(c, ActorHeadOp, OP_ACTOR_HEAD, k_actor_head, 1, 256, c->stream, crux_dense_act(q1, q1->nd.L), crux_dense_act(q2, q2->nd.L), lp, la->p, B, dy1, dy2, stats);
  rc = crux_dense_backward(q1, sa, B, dy1, 1.0f, false, dsa1, c->stream); if (rc) return rc;     // gradient w.r.t. vcat(s, a) only: the Q parameters are not trained here
  rc = crux_dense_backward(q2, sa, B, dy2, 1.0f, false, dsa2, c->stream); if (rc) return rc;
  CRUX_RUN(c, ActorGradOp, OP_ACTOR_GRAD, k_actor_grad, nblk(B * ad), 256, c->stream, sa, mu, eps, actor->p + actor->nd.xoff, dsa1, dsa2, la->p, od, ad, B, dmu, dls);
  Sumsq2Fix fx{};
  rc = crux_dense_backward(actor, S, B, dmu, 1.0f, true, nullptr, c->stream, &fx, 0); if (rc) return rc;
  CRUX_RUN(c, RowsumOp, OP_ROWSUM, k_rowsum, ad, 256, c->stream, dls, ad, B, actor->g + actor->nd.xoff, (int32_t*)nullptr);
  CRUX_RUN(c, Sumsq2Op, OP_SUMSQ2, k_sumsq2, SUMSQ_BLOCKS, 256, c->stream, actor->g, (int64_t)actor->nd.n_params, (float*)nullptr, (int64_t)0, ssq, fx);
  CRUX_RUN(c, ActorInfoOp, OP_ACTOR_INFO, k_actor_info, 1, 1, c->stream, stats, ssq, B, dinfo);
  rc = adam_gated(actor, ssq, st); if (rc) return rc;
  return finish_step(c, dinfo, st, info_out, "sac_actor_loss");
}

}  // extern "C"
