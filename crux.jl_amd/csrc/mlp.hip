// mlp.hip -- Chain(Dense...) networks: parameters, generic forward, polyak, Adam.
// Reference: Flux Chain/Dense wrapped by ContinuousNetwork / DiscreteNetwork (src/policies.jl:68-157),
// polyak_average! (src/policies.jl:48-59), Flux.update!(Adam) (src/training.jl:21).
#include "common.h"
#include "exec.h"
#include "ops_small.h"

static void fill_desc(NetDesc& nd, int32_t L, const int32_t* dims, const int32_t* acts, int32_t n_extra) {
  nd.L = L; int off = 0; nd.maxdim = 0;
  for (int i = 0; i <= L; ++i) { nd.dims[i] = dims[i]; if (dims[i] > nd.maxdim) nd.maxdim = dims[i]; }
  for (int l = 0; l < L; ++l) {
    nd.acts[l] = acts[l]; nd.woff[l] = off; off += dims[l + 1] * dims[l]; nd.boff[l] = off; off += dims[l + 1];
  }
  nd.xoff = off; nd.n_extra = n_extra; nd.n_params = off + n_extra;
}

#include "mlp_forward.h"
__device__ __forceinline__ void mlp_forward_body(const NetDesc& nd, const float* __restrict__ p, const float* __restrict__ x, int64_t B, float* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  mlp_forward_run(nd, p, x, B, y, sm, blockIdx.x, gridDim.x);
}
__global__ __launch_bounds__(256) void k_mlp_forward(NetDesc nd, const float* __restrict__ p, const float* __restrict__ x, int64_t B, float* __restrict__ y) {
  mlp_forward_body(nd, p, x, B, y);
}
// the same network shape evaluated for many (parameters, input, output) triples in one launch: grid.y = job
__global__ __launch_bounds__(256) void k_mlp_forward_multi(NetDesc nd, const crux_fwd_job* __restrict__ jobs, int64_t B) {
  const crux_fwd_job j = jobs[blockIdx.y];
  mlp_forward_body(nd, j.p, j.x, B, j.y);
}
// ---- value(pi, x) for IN -> 64 -> 64 -> OUT chains at large batches on the matrix pipes (round 6) -------------------------------------------------------------------
// fill_gae! evaluates the critic on every s and sp of the buffer (sampler.jl:264-266: 2 x 65 536 rows per PPO iteration; x 128 learners in a population run, where the scalar
// kernel above took 16 of an iteration's 472 ms). One WAVE per 16-sample tile, the three layers as v_mfma_f32_16x16x4_f32 chains with the weights of all layers in registers
// for the whole launch; instruction ks of a layer carries k = 4 ks + (lane group) -- ASCENDING k, the order of the scalar loop above, and the MFMA is a sequential fma chain
// over its four k slices (tools/mfma_chain_test.hip) -- then `acc + b`, then the activation through the same crux_act: the results are the scalar kernel's BIT FOR BIT
// (tests/test_gpu_components.py compares under CRUX_FORCE_GENERIC=1). Activations pass from one layer's D layout [feature 4g + r][sample c] to the next layer's B layout
// [k = 4 ks + g][sample c] through a per-wave LDS tile.
typedef float f32x4_ __attribute__((ext_vector_type(4)));
#define FWH_LD 17
__device__ __forceinline__ void mlp_forward_h64_run(const NetDesc& nd, const float* __restrict__ p, const float* __restrict__ x, int64_t B, float* __restrict__ y, float* tile,
                                                    const int64_t wave0, const int64_t n_waves) {
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int in0 = nd.dims[0], out3 = nd.dims[3], ks0 = (in0 + 3) >> 2;
  const float* W1 = p + nd.woff[0]; const float* W2 = p + nd.woff[1]; const float* W3 = p + nd.woff[2];
  float w1f[4][8], w2f[4][16], w3f[16]; f32x4_ b1v[4], b2v[4], b3v;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) { const int k = 4 * ks + g; w1f[m][ks] = (ks < ks0 && k < in0) ? W1[16 * m + c + 64 * k] : 0.f; }
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) w2f[m][ks] = W2[16 * m + c + 64 * (4 * ks + g)];
#pragma unroll
    for (int r = 0; r < 4; ++r) { b1v[m][r] = p[nd.boff[0] + 16 * m + 4 * g + r]; b2v[m][r] = p[nd.boff[1] + 16 * m + 4 * g + r]; } }
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) w3f[ks] = c < out3 ? W3[c + out3 * (4 * ks + g)] : 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) b3v[r] = 4 * g + r < out3 ? p[nd.boff[2] + 4 * g + r] : 0.f;
  const int a1 = nd.acts[0], a2 = nd.acts[1], a3 = nd.acts[2];
  const int64_t n_tiles = (B + 15) >> 4;
  for (int64_t t = wave0; t < n_tiles; t += n_waves) {
    const int64_t s = 16 * t + c; const bool vs = s < B; const int64_t sc = vs ? s : B - 1;
    float xB[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) { const int k = 4 * ks + g; xB[ks] = (ks < ks0 && k < in0) ? x[sc * in0 + k] : 0.f; }
#pragma unroll
    for (int m = 0; m < 4; ++m) { f32x4_ acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) if (ks < ks0) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1f[m][ks], xB[ks], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) tile[(16 * m + 4 * g + r) * FWH_LD + c] = crux_act(a1, acc[r] + b1v[m][r]); }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    float hB[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) hB[ks] = tile[(4 * ks + g) * FWH_LD + c];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int m = 0; m < 4; ++m) { f32x4_ acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w2f[m][ks], hB[ks], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) tile[(16 * m + 4 * g + r) * FWH_LD + c] = crux_act(a2, acc[r] + b2v[m][r]); }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) hB[ks] = tile[(4 * ks + g) * FWH_LD + c];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    f32x4_ acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w3f[ks], hB[ks], acc, 0, 0, 0);
    if (vs) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int o = 4 * g + r; if (o < out3) y[s * out3 + o] = crux_act(a3, acc[r] + b3v[r]); } }
  }
}
static inline bool fwd_h64_ok(const NetDesc& nd, int64_t B) {
  return nd.L == 3 && nd.dims[1] == 64 && nd.dims[2] == 64 && nd.dims[0] >= 1 && nd.dims[0] <= 32 && nd.dims[3] >= 1 && nd.dims[3] <= 16 && B >= 1024 && !crux_sw().force_generic;
}
__global__ __launch_bounds__(256) void k_mlp_forward_h64(NetDesc nd, const float* __restrict__ p, const float* __restrict__ x, int64_t B, float* __restrict__ y) {
  __shared__ float tiles[4][64 * FWH_LD];
  mlp_forward_h64_run(nd, p, x, B, y, tiles[threadIdx.x >> 6], (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), (int64_t)gridDim.x * 4);
}
__global__ __launch_bounds__(256) void k_mlp_forward_h64_multi(NetDesc nd, const crux_fwd_job* __restrict__ jobs, int64_t B) {
  __shared__ float tiles[4][64 * FWH_LD];
  const crux_fwd_job j = jobs[blockIdx.y];
  mlp_forward_h64_run(nd, j.p, j.x, B, j.y, tiles[threadIdx.x >> 6], (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), (int64_t)gridDim.x * 4);
}
static inline unsigned fwd_h64_blocks(int64_t B, int n_jobs) {      // 4 tiles (waves) per block; a few tiles per wave so that the register-resident weights are loaded once per ~8 tiles
  const int64_t tiles = (B + 15) >> 4; int64_t nb = (tiles + 31) / 32; const int64_t cap = n_jobs > 1 ? 64 : 1024; if (nb > cap) nb = cap; if (nb < 1) nb = 1; return (unsigned)nb;
}
int32_t crux_mlp_forward_multi_impl(crux_ctx* c, const NetDesc& nd, const crux_fwd_job* d_jobs, int n_jobs, int64_t B) {
  if (n_jobs < 1 || B < 1) return CRUX_OK;
  if (fwd_h64_ok(nd, B)) {
    hipLaunchKernelGGL(k_mlp_forward_h64_multi, dim3(fwd_h64_blocks(B, n_jobs), (unsigned)n_jobs), dim3(256), 0, c->stream, nd, d_jobs, B);
    return crux_launch_check(c, "k_mlp_forward_h64_multi"); }
  const size_t lds = sizeof(float) * 2 * (size_t)nd.maxdim * FWD_TS;
  if (lds > 65536) return crux_fail(c, CRUX_EUNSUP, "mlp_forward: layer width %d exceeds the generic kernel's LDS tile", nd.maxdim);
  int64_t nb = (B + FWD_TS - 1) / FWD_TS; if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(k_mlp_forward_multi, dim3((unsigned)nb, (unsigned)n_jobs), dim3(256), lds, c->stream, nd, d_jobs, B);
  return crux_launch_check(c, "k_mlp_forward_multi");
}

int32_t crux_mlp_forward_impl(crux_mlp* net, const float* d_x, int64_t B, float* d_y, const float* params_override);

__global__ void k_glorot(NetDesc nd, float* p, uint64_t seed, uint32_t stream, float extra_init) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= nd.n_params) return;
  if (gid >= nd.xoff) { p[gid] = extra_init; return; }
  int64_t ctr = 0;
  for (int l = 0; l < nd.L; ++l) {
    const int64_t nw = (int64_t)nd.dims[l] * nd.dims[l + 1];
    if (gid >= nd.woff[l] && gid < nd.boff[l]) {
      const float scale = sqrtf(24.0f / (float)(nd.dims[l] + nd.dims[l + 1]));
      crux_u32x4 x = crux_philox(seed, (uint64_t)(ctr + (gid - nd.woff[l])), stream, CRUX_RNG_INIT);
      p[gid] = (crux_u32_to_f32(x.v[0]) - 0.5f) * scale;
      return;
    }
    if (gid >= nd.boff[l] && gid < nd.boff[l] + nd.dims[l + 1]) { p[gid] = 0.f; return; }
    ctr += nw;
  }
}

__global__ void k_polyak(float* __restrict__ to, const float* __restrict__ from, float tau, int64_t n) { PolyakOp::run(blockIdx.x, gridDim.x, to, from, tau, n); }

// Flux.Optimise.Adam apply! with Float64 scalar fields: each broadcast evaluated in Float64 per element,
// rounded to Float32 on store.
__global__ void k_adam_apply(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             const double* __restrict__ bp, double eta, double b1, double b2, double eps, float gscale, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double bp1 = bp[0], bp2 = bp[1];
  const float gi = g[i] * gscale;
  const double gd = (double)gi;
  const float mi = (float)(b1 * (double)m[i] + (1.0 - b1) * gd);
  const float vi = (float)(b2 * (double)v[i] + ((1.0 - b2) * gd) * gd);
  const float d = (float)((double)mi / (1.0 - bp1) / (sqrt((double)vi / (1.0 - bp2)) + eps) * eta);
  m[i] = mi; v[i] = vi; p[i] = p[i] - d;
}
__global__ void k_adam_advance(double* bp, double b1, double b2) { bp[0] *= b1; bp[1] *= b2; }

extern "C" {

int32_t crux_mlp_create(crux_ctx* ctx, int32_t L, const int32_t* dims, const int32_t* acts, int32_t n_extra, crux_mlp** out) {
  if (!ctx || !out || (L > 0 && (!dims || !acts))) return CRUX_EINVAL;
  if (L < 0 || L > CRUX_MAXL || n_extra < 0 || (L == 0 && n_extra < 1)) return crux_fail(ctx, CRUX_EINVAL, "mlp: n_layers %d / n_extra %d out of range", L, n_extra);
  for (int i = 0; i <= L && L > 0; ++i) if (dims[i] < 1 || dims[i] > 1024) return crux_fail(ctx, CRUX_EINVAL, "mlp: dim %d = %d unsupported", i, dims[i]);
  crux_mlp* n = new crux_mlp(); n->ctx = ctx;
  const int32_t dim0[1] = {0};
  fill_desc(n->nd, L, L > 0 ? dims : dim0, acts, n_extra);
  const size_t bytes = sizeof(float) * (size_t)n->nd.n_params;
  if (hipMalloc(&n->p, bytes) != hipSuccess || hipMalloc(&n->g, bytes) != hipSuccess || hipMalloc(&n->m, bytes) != hipSuccess ||
      hipMalloc(&n->v, bytes) != hipSuccess || hipMalloc(&n->bp, 4 * sizeof(double)) != hipSuccess)   /* [beta1^t, beta2^t, ticket of k_adam_gated, pad] */ { delete n; return crux_fail(ctx, CRUX_ENOMEM, "mlp: hipMalloc failed"); }
  HIPCHK(ctx, hipMemsetAsync(n->p, 0, bytes, ctx->stream)); HIPCHK(ctx, hipMemsetAsync(n->g, 0, bytes, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(n->m, 0, bytes, ctx->stream)); HIPCHK(ctx, hipMemsetAsync(n->v, 0, bytes, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(n->bp, 0, 4 * sizeof(double), ctx->stream));
  *out = n; return CRUX_OK;
}

int32_t crux_mlp_destroy(crux_mlp* n) {
  if (!n) return CRUX_OK;
  crux_sync_before_free(n->ctx);
  (void)hipFree(n->p); (void)hipFree(n->g); (void)hipFree(n->m); (void)hipFree(n->v); (void)hipFree(n->bp); if (n->ws) (void)hipFree(n->ws);
  delete n; return CRUX_OK;
}

int64_t crux_mlp_n_params(const crux_mlp* n) { return n ? n->nd.n_params : -1; }
float* crux_mlp_params_ptr(crux_mlp* n) { return n ? n->p : nullptr; }
float* crux_mlp_grads_ptr(crux_mlp* n) { return n ? n->g : nullptr; }

int32_t crux_mlp_set_params(crux_mlp* n, const float* h, int64_t cnt) {
  if (!n || !h) return CRUX_EINVAL;
  if (cnt != n->nd.n_params) return crux_fail(n->ctx, CRUX_EINVAL, "set_params: got %lld values, network has %d", (long long)cnt, n->nd.n_params);
  HIPCHK(n->ctx, hipMemcpyAsync(n->p, h, sizeof(float) * (size_t)cnt, hipMemcpyHostToDevice, n->ctx->stream));
  HIPCHK(n->ctx, hipStreamSynchronize(n->ctx->stream));
  return CRUX_OK;
}
int32_t crux_mlp_get_params(crux_mlp* n, float* h, int64_t cnt) {
  if (!n || !h) return CRUX_EINVAL;
  if (cnt != n->nd.n_params) return crux_fail(n->ctx, CRUX_EINVAL, "get_params: asked %lld values, network has %d", (long long)cnt, n->nd.n_params);
  HIPCHK(n->ctx, hipMemcpyAsync(h, n->p, sizeof(float) * (size_t)cnt, hipMemcpyDeviceToHost, n->ctx->stream));
  HIPCHK(n->ctx, hipStreamSynchronize(n->ctx->stream));
  return CRUX_OK;
}

int32_t crux_mlp_init_glorot(crux_mlp* n, uint64_t seed, uint32_t stream, float extra_init) {
  if (!n) return CRUX_EINVAL;
  const int nb = (n->nd.n_params + 255) / 256;
  hipLaunchKernelGGL(k_glorot, dim3(nb), dim3(256), 0, n->ctx->stream, n->nd, n->p, seed, stream, extra_init);
  return crux_launch_check(n->ctx, "k_glorot");
}

int32_t crux_mlp_forward(crux_mlp* n, const float* d_x, int64_t B, float* d_y) {
  if (!n || !d_x || !d_y || B < 0) return CRUX_EINVAL;
  return crux_mlp_forward_impl(n, d_x, B, d_y, nullptr);
}

int32_t crux_mlp_forward_host(crux_mlp* n, const float* x, int64_t B, float* y) {
  if (!n || !x || !y || B < 0) return CRUX_EINVAL;
  if (B == 0) return CRUX_OK;
  crux_ctx* c = n->ctx;
  const size_t bi = sizeof(float) * (size_t)B * n->nd.dims[0], bo = sizeof(float) * (size_t)B * n->nd.dims[n->nd.L];
  char* sc = (char*)crux_scratch(c, bi + bo + 256);
  if (!sc) return crux_fail(c, CRUX_ENOMEM, "forward_host: scratch");
  float* dx = (float*)sc; float* dy = (float*)(sc + ((bi + 255) / 256) * 256);
  HIPCHK(c, hipMemcpyAsync(dx, x, bi, hipMemcpyHostToDevice, c->stream));
  int32_t rc = crux_mlp_forward_impl(n, dx, B, dy, nullptr); if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(y, dy, bo, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return CRUX_OK;
}

int32_t crux_mlp_copy(crux_mlp* to, const crux_mlp* from) {
  if (!to || !from) return CRUX_EINVAL;
  if (to->nd.n_params != from->nd.n_params) return crux_fail(to->ctx, CRUX_EINVAL, "copyto!: parameter counts differ");
  HIPCHK(to->ctx, hipMemcpyAsync(to->p, from->p, sizeof(float) * (size_t)to->nd.n_params, hipMemcpyDeviceToDevice, to->ctx->stream));
  return CRUX_OK;
}

int32_t crux_polyak(crux_mlp* to, const crux_mlp* from, float tau) {
  if (!to || !from) return CRUX_EINVAL;
  if (to->nd.n_params != from->nd.n_params) return crux_fail(to->ctx, CRUX_EINVAL, "polyak_average!: parameter counts differ");
  const int64_t n = to->nd.n_params;
  // inside a recorded chain the three polyak updates of an epoch share a phase with other ops: 64 grid-striding blocks each instead of n / 256 (a phase of > 768 blocks takes
  // two rounds over the chip)
  const unsigned nbk = (unsigned)((n + 255) / 256);
  CRUX_RUN(to->ctx, PolyakOp, OP_POLYAK, k_polyak, crux_exec_recording(to->ctx) ? (nbk < 64u ? nbk : 64u) : nbk, 256, to->ctx->stream, to->p, from->p, tau, n);
  return crux_launch_check(to->ctx, "k_polyak");
}

int32_t crux_adam_init(crux_mlp* n, double eta, double b1, double b2, double eps) {
  if (!n) return CRUX_EINVAL;
  n->eta = eta; n->b1 = b1; n->b2 = b2; n->eps = eps; n->has_adam = true;
  const size_t bytes = sizeof(float) * (size_t)n->nd.n_params;
  HIPCHK(n->ctx, hipMemsetAsync(n->m, 0, bytes, n->ctx->stream)); HIPCHK(n->ctx, hipMemsetAsync(n->v, 0, bytes, n->ctx->stream));
  double bp[2] = {b1, b2};
  HIPCHK(n->ctx, hipMemcpyAsync(n->bp, bp, sizeof bp, hipMemcpyHostToDevice, n->ctx->stream));
  HIPCHK(n->ctx, hipStreamSynchronize(n->ctx->stream));
  return CRUX_OK;
}
int32_t crux_adam_get_state(crux_mlp* n, float* m, float* v, double* bp) {
  if (!n) return CRUX_EINVAL;
  const size_t bytes = sizeof(float) * (size_t)n->nd.n_params;
  if (m) HIPCHK(n->ctx, hipMemcpyAsync(m, n->m, bytes, hipMemcpyDeviceToHost, n->ctx->stream));
  if (v) HIPCHK(n->ctx, hipMemcpyAsync(v, n->v, bytes, hipMemcpyDeviceToHost, n->ctx->stream));
  if (bp) HIPCHK(n->ctx, hipMemcpyAsync(bp, n->bp, 2 * sizeof(double), hipMemcpyDeviceToHost, n->ctx->stream));
  HIPCHK(n->ctx, hipStreamSynchronize(n->ctx->stream));
  return CRUX_OK;
}
int32_t crux_adam_set_state(crux_mlp* n, const float* m, const float* v, const double* bp) {
  if (!n) return CRUX_EINVAL;
  const size_t bytes = sizeof(float) * (size_t)n->nd.n_params;
  if (m) HIPCHK(n->ctx, hipMemcpyAsync(n->m, m, bytes, hipMemcpyHostToDevice, n->ctx->stream));
  if (v) HIPCHK(n->ctx, hipMemcpyAsync(n->v, v, bytes, hipMemcpyHostToDevice, n->ctx->stream));
  if (bp) HIPCHK(n->ctx, hipMemcpyAsync(n->bp, bp, 2 * sizeof(double), hipMemcpyHostToDevice, n->ctx->stream));
  HIPCHK(n->ctx, hipStreamSynchronize(n->ctx->stream));
  return CRUX_OK;
}

int32_t crux_mlp_set_squash(crux_mlp* n, float ascale) {
  if (!n) return CRUX_EINVAL;
  if (!(ascale >= 0.f)) return crux_fail(n->ctx, CRUX_EINVAL, "SquashedGaussianPolicy: ascale must be >= 0 (0 = plain GaussianPolicy)");
  n->squash = ascale; return CRUX_OK;
}
float crux_mlp_get_squash(const crux_mlp* n) { return n ? n->squash : 0.f; }

int32_t crux_adam_state_ptrs(crux_mlp* n, float** d_m, float** d_v) { if (!n) return CRUX_EINVAL; if (d_m) *d_m = n->m; if (d_v) *d_v = n->v; return CRUX_OK; }

int32_t crux_adam_apply(crux_mlp* n, float grad_scale) {
  if (!n) return CRUX_EINVAL;
  if (!n->has_adam) return crux_fail(n->ctx, CRUX_EINVAL, "adam_apply: crux_adam_init was not called");
  const int64_t cnt = n->nd.n_params;
  hipLaunchKernelGGL(k_adam_apply, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, n->ctx->stream, n->p, n->g, n->m, n->v, n->bp,
                     n->eta, n->b1, n->b2, n->eps, grad_scale, cnt);
  hipLaunchKernelGGL(k_adam_advance, dim3(1), dim3(1), 0, n->ctx->stream, n->bp, n->b1, n->b2);
  return crux_launch_check(n->ctx, "k_adam_apply");
}

}  // extern "C"

int32_t crux_mlp_forward_impl(crux_mlp* n, const float* d_x, int64_t B, float* d_y, const float* params_override) {
  if (B == 0) return CRUX_OK;
  crux_ctx* c = n->ctx;
  if (n->nd.L < 1) return crux_fail(c, CRUX_EINVAL, "mlp_forward: the handle is a bare parameter vector (n_layers = 0)");
  if (fwd_h64_ok(n->nd, B)) {      // the 64-wide family at large batches (fill_gae!'s critic evaluations): the matrix pipes, same bits (above)
    hipLaunchKernelGGL(k_mlp_forward_h64, dim3(fwd_h64_blocks(B, 1)), dim3(256), 0, c->stream, n->nd, params_override ? params_override : n->p, d_x, B, d_y);
    return crux_launch_check(c, "k_mlp_forward_h64"); }
  const size_t lds = sizeof(float) * 2 * (size_t)n->nd.maxdim * FWD_TS;
  if (lds > 65536) return crux_fail(c, CRUX_EUNSUP, "mlp_forward: layer width %d exceeds the generic kernel's LDS tile", n->nd.maxdim);
  int64_t nb = (B + FWD_TS - 1) / FWD_TS; if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(k_mlp_forward, dim3((unsigned)nb), dim3(256), lds, c->stream, n->nd, params_override ? params_override : n->p, d_x, B, d_y);
  return crux_launch_check(c, "k_mlp_forward");
}
