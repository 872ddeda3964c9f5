// advantage.hip -- fill_gae! / fill_returns! / whiten on device columns.
// Reference: src/sampler.jl:255-281 (GAE + returns), src/utils.jl:41-42 (whiten), episodes() src/experience_buffer.jl:194-212.
#include "common.h"
#include <vector>

int32_t crux_mlp_forward_impl(crux_mlp* net, const float* d_x, int64_t B, float* d_y, const float* params_override);

// One thread per episode: a thread sitting on an episode-end row (or the last row) walks its episode backwards.
// The recurrence is evaluated sequentially in Float32 with the reference's association
//   A = ((c*A + r) + ((1f0 - done)*gamma)*Vsp) - Vs        (sampler.jl:269)
//   R = r + gamma*R                                        (sampler.jl:278)
// so results are bit-identical to the scalar loop given the same V(s), V(sp).
// Row k of the scanned block lives at physical row (first + k) mod cap (a ring range; first = 0, cap = n for a whole buffer); Vs / Vsp are
// indexed by k. close_last: the block's last row terminates an episode even without :episode_end (episodes(d) closes a trailing episode at
// length(d), experience_buffer.jl:207-211; steps!(reset=true) sets the flag itself); otherwise the trailing open rows get 0 -- the values
// mdp_data initialised them with in the reference's fresh `data` block, which terminate_episode! never reached (sampler.jl:53-57,140-148).
__device__ __forceinline__ void gae_returns_body(const float* __restrict__ r, const uint8_t* __restrict__ done, const uint8_t* __restrict__ ee,
                              const float* __restrict__ Vs, const float* __restrict__ Vsp, float lambda, float gamma, int64_t n,
                              float* __restrict__ adv, float* __restrict__ ret, int32_t* __restrict__ nan_flag,
                              int64_t first = 0, int64_t cap = 0, int close_last = 1, int64_t seg = 0) {
  if (cap <= 0) cap = n;
  if (seg <= 0) seg = n;        // env-major blocks: rows [e*seg, (e+1)*seg) belong to environment e and no episode crosses that boundary
  auto phys = [&](int64_t k) -> int64_t { const int64_t q = first + k; return q >= cap ? q - cap : q; };
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const bool closed = ee[phys(i)];
    if (!(closed || i == n - 1 || (i + 1) % seg == 0)) continue;
    const int64_t k0 = (i / seg) * seg;     // first row of this environment's segment
    if (!closed && !close_last) {          // trailing rows of an episode that is still open at the end of the segment
      for (int64_t k = i; k >= k0 && !ee[phys(k)]; --k) { if (adv) adv[phys(k)] = 0.f; if (ret) ret[phys(k)] = 0.f; }
      continue;
    }
    float A = 0.f, R = 0.f; const float c = __fmul_rn(lambda, gamma);
    bool bad = false;
    for (int64_t k = i; k >= k0 && (k == i || !ee[phys(k)]); --k) {
      const int64_t q = phys(k);
      const float rk = r[q];
      if (adv) {
        const float t2 = __fadd_rn(__fmul_rn(c, A), rk);
        const float t3 = __fmul_rn(__fmul_rn(__fsub_rn(1.f, done[q] ? 1.f : 0.f), gamma), Vsp[k]);
        A = __fsub_rn(__fadd_rn(t2, t3), Vs[k]);
        adv[q] = A; bad |= isnan(A);
      }
      if (ret) { R = __fadd_rn(rk, __fmul_rn(gamma, R)); ret[q] = R; }
    }
    if (bad) atomicOr(nan_flag, 1);
  }
}
__global__ void k_gae_returns(const float* __restrict__ r, const uint8_t* __restrict__ done, const uint8_t* __restrict__ ee,
                              const float* __restrict__ Vs, const float* __restrict__ Vsp, float lambda, float gamma, int64_t n,
                              float* __restrict__ adv, float* __restrict__ ret, int32_t* __restrict__ nan_flag) {
  gae_returns_body(r, done, ee, Vs, Vsp, lambda, gamma, n, adv, ret, nan_flag);
}
__global__ void k_gae_returns_rows(const float* __restrict__ r, const uint8_t* __restrict__ done, const uint8_t* __restrict__ ee,
                                   const float* __restrict__ Vs, const float* __restrict__ Vsp, float lambda, float gamma, int64_t n,
                                   float* __restrict__ adv, float* __restrict__ ret, int32_t* __restrict__ nan_flag, int64_t first, int64_t cap, int close_last, int64_t seg) {
  gae_returns_body(r, done, ee, Vs, Vsp, lambda, gamma, n, adv, ret, nan_flag, first, cap, close_last, seg);
}
struct GaeJob { const float* r; const uint8_t* done; const uint8_t* ee; const float* Vs; const float* Vsp; float* adv; float* ret; int32_t* flag; };
__global__ void k_gae_returns_multi(const GaeJob* __restrict__ jobs, float lambda, float gamma, int64_t n) {     // grid.y = buffer
  const GaeJob j = jobs[blockIdx.y];
  gae_returns_body(j.r, j.done, j.ee, j.Vs, j.Vsp, lambda, gamma, n, j.adv, j.ret, j.flag);
}

// whiten(v) = (v .- mean(v)) ./ std(v) (utils.jl:41-42) with Julia's own Float32 reductions: Statistics.mean = sum(A) / length(A), Statistics.std = sqrt(centralize_sumabs2(A, m) / (n - 1)),
// both through Base.mapreduce_impl's pairwise scheme (reduce.jl; pairwise_blocksize = 1024): a range of at most 1024 elements is summed from the left, a longer one is split at
// ifirst + (ilast - ifirst) >> 1 and the halves are added. One block: thread 0 walks the recursion once to list the leaves, thread i sums leaf i from the left, thread 0 walks it
// again to add the leaf sums in the tree's order -- the oracle's orc_jl_mean_f32 / orc_jl_std_f32 bit for bit (round 5; the Float64 tree sums before agreed with the oracle, not with Julia).
#define JLW_MAXLEAF 4096        // leaves of one reduction: n up to ~2 M elements (an on-policy buffer); larger columns are refused by the host
struct JlFrame { int64_t a, b; float v1; int st; };
__device__ __forceinline__ float jlw_term(float x, bool centred, float m) { if (!centred) return x; const float d = __fsub_rn(x, m); return __fmul_rn(d, d); }
__device__ float jlw_reduce(const float* __restrict__ v, int64_t n, bool centred, float m, int32_t* leaf_lo, float* leaf_sum, int* n_leaf) {      // leaves are contiguous: leaf q = [leaf_lo[q], leaf_lo[q + 1])
  const int tid = threadIdx.x;
  __shared__ JlFrame stk[48];      // the recursion stack of thread 0 (its only user): LDS, not 1 152 bytes of private memory per thread of the launch
  if (tid == 0) {      // the leaves of mapreduce_impl(f, +, A, 1, n, 1024) in the order the recursion visits them
    int sp = 0, nl = 0; stk[sp++] = JlFrame{0, n - 1, 0.f, 0};
    while (sp) { JlFrame f = stk[sp - 1];
      if (f.b - f.a < 1024) { leaf_lo[nl] = (int32_t)f.a; ++nl; --sp; continue; }
      const int64_t mid = f.a + ((f.b - f.a) >> 1);
      if (f.st == 0) { stk[sp - 1].st = 1; stk[sp++] = JlFrame{f.a, mid, 0.f, 0}; continue; }
      if (f.st == 1) { stk[sp - 1].st = 2; stk[sp++] = JlFrame{mid + 1, f.b, 0.f, 0}; continue; }
      --sp; }
    leaf_lo[nl] = (int32_t)n; *n_leaf = nl; }
  __syncthreads();
  const int nl = *n_leaf;
  for (int q = tid; q < nl; q += 1024) { const int64_t lo = leaf_lo[q], hi = (int64_t)leaf_lo[q + 1] - 1;      // v = f(a1) + f(a2); v += f(a3); ... (reduce.jl, the sequential portion)
    float acc = jlw_term(v[lo], centred, m);
    int64_t i = lo + 1;
    for (; i + 15 <= hi; i += 16) { float x[16];      // 16 loads in flight, then the 16 additions in order (the sum itself is a serial chain; one load per addition made it a memory round trip each)
#pragma unroll
      for (int k = 0; k < 16; ++k) x[k] = v[i + k];
#pragma unroll
      for (int k = 0; k < 16; ++k) acc = __fadd_rn(acc, jlw_term(x[k], centred, m)); }
    for (; i <= hi; ++i) acc = __fadd_rn(acc, jlw_term(v[i], centred, m));
    leaf_sum[q] = acc; }
  __syncthreads();
  __shared__ float result;
  if (tid == 0) {      // op(v1, v2) up the same tree
    int sp = 0, nx = 0; float ret = 0.f; stk[sp++] = JlFrame{0, n - 1, 0.f, 0};
    while (sp) { JlFrame f = stk[sp - 1];
      if (f.b - f.a < 1024) { ret = leaf_sum[nx++]; --sp; continue; }
      const int64_t mid = f.a + ((f.b - f.a) >> 1);
      if (f.st == 0) { stk[sp - 1].st = 1; stk[sp++] = JlFrame{f.a, mid, 0.f, 0}; continue; }
      if (f.st == 1) { stk[sp - 1].v1 = ret; stk[sp - 1].st = 2; stk[sp++] = JlFrame{mid + 1, f.b, 0.f, 0}; continue; }
      ret = __fadd_rn(f.v1, ret); --sp; }
    result = ret; }
  __syncthreads();
  const float r = result;
  __syncthreads();
  return r;
}
__device__ __forceinline__ void whiten_block(float* __restrict__ v, int64_t n) {
  __shared__ int32_t leaf_lo[JLW_MAXLEAF + 1];
  __shared__ float leaf_sum[JLW_MAXLEAF];
  __shared__ int n_leaf;
  const float mean = __fdiv_rn(jlw_reduce(v, n, false, 0.f, leaf_lo, leaf_sum, &n_leaf), (float)n);                 // sum(A) / length(A)
  const float sd = sqrtf(__fdiv_rn(jlw_reduce(v, n, true, mean, leaf_lo, leaf_sum, &n_leaf), (float)(n - 1)));        // sqrt(centralize_sumabs2(A, m) / (n - 1))
  for (int64_t i = threadIdx.x; i < n; i += 1024) v[i] = __fdiv_rn(__fsub_rn(v[i], mean), sd);
}
__global__ __launch_bounds__(1024) void k_whiten(float* __restrict__ v, int64_t n) { whiten_block(v, n); }
__global__ __launch_bounds__(1024) void k_whiten_multi(float* const* __restrict__ vs, int64_t n) { whiten_block(vs[blockIdx.x], n); }

static int32_t values(crux_mlp* critic, const float* d_x, int64_t n, float* d_y) {
  return crux_mlp_forward_impl(critic, d_x, n, d_y, nullptr);   // 2 x 65536 critic evaluations = 0.14 ms per iteration with the generic forward kernel
}

// episodes!-style evaluation (sampler.jl:175-251) over an env-major rollout block: the FIRST episode of every environment.
// undiscounted_return = sum(r) (:203-213), discounted_return = reverse recursion r + gamma*R in Float32 (:223-229), length.
__global__ void k_first_episode_metrics(const float* __restrict__ r, const uint8_t* __restrict__ ee, int n_envs, int64_t T, float gamma,
                                        float* __restrict__ und, float* __restrict__ dis, int64_t* __restrict__ len, uint8_t* __restrict__ complete) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x; if (e >= n_envs) return;
  const int64_t base = (int64_t)e * T; int64_t stop = -1;
  for (int64_t t = 0; t < T; ++t) if (ee[base + t]) { stop = t; break; }
  complete[e] = stop >= 0 ? 1 : 0;
  if (stop < 0) stop = T - 1;
  float u = 0.f, d = 0.f;
  for (int64_t t = 0; t <= stop; ++t) u = __fadd_rn(u, r[base + t]);
  for (int64_t t = stop; t >= 0; --t) d = __fadd_rn(r[base + t], __fmul_rn(gamma, d));
  und[e] = u; dis[e] = d; len[e] = stop + 1;
}

// a Float32 one-row column that exists (fill_gae!/fill_returns! `source=` / `target=` keywords, sampler.jl:65-66,255,275)
static bool f32_scalar_col(const crux_buffer* b, int k) { return (k == CRUX_COL_R || has_col(b, k)) && col_rows(b, k) == 1 && col_elem(b, k) == 4; }

extern "C" {

int32_t crux_first_episode_metrics(crux_buffer* b, int32_t n_envs, int64_t T, float gamma, float* undisc, float* disc, int64_t* length, uint8_t* complete) {
  if (!b || n_envs < 1 || T < 1) return CRUX_EINVAL;
  crux_ctx* c = b->ctx;
  if ((int64_t)n_envs * T != b->elements || b->next_ind != (b->elements % b->capacity)) return crux_fail(c, CRUX_EINVAL, "episode metrics: buffer must hold exactly one %d x %lld env-major rollout block", n_envs, (long long)T);
  const size_t nb = (size_t)n_envs;
  char* sc = (char*)crux_scratch(c, nb * 17 + 1024); if (!sc) return crux_fail(c, CRUX_ENOMEM, "episode metrics: scratch");
  int64_t* d_len = (int64_t*)sc; float* d_u = (float*)(sc + 8 * nb); float* d_d = d_u + nb; uint8_t* d_c = (uint8_t*)(d_d + nb);
  hipLaunchKernelGGL(k_first_episode_metrics, dim3((unsigned)((n_envs + 63) / 64)), dim3(64), 0, c->stream, (const float*)b->col[CRUX_COL_R], (const uint8_t*)b->col[CRUX_COL_EPISODE_END],
                     n_envs, T, gamma, d_u, d_d, d_len, d_c);
  int32_t rc = crux_launch_check(c, "k_first_episode_metrics"); if (rc) return rc;
  if (undisc) HIPCHK(c, hipMemcpyAsync(undisc, d_u, 4 * nb, hipMemcpyDeviceToHost, c->stream));
  if (disc) HIPCHK(c, hipMemcpyAsync(disc, d_d, 4 * nb, hipMemcpyDeviceToHost, c->stream));
  if (length) HIPCHK(c, hipMemcpyAsync(length, d_len, 8 * nb, hipMemcpyDeviceToHost, c->stream));
  if (complete) HIPCHK(c, hipMemcpyAsync(complete, d_c, nb, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return CRUX_OK;
}

int32_t crux_fill_gae_keys(crux_buffer* b, crux_mlp* critic, float lambda, float gamma, int32_t source, int32_t target) {
  if (!b || !critic) return CRUX_EINVAL;
  crux_ctx* c = b->ctx;
  if (!f32_scalar_col(b, source) || !f32_scalar_col(b, target)) return crux_fail(c, CRUX_EINVAL, "fill_gae!: buffer lacks the source / target column (%d -> %d)", source, target);
  if (critic->nd.dims[critic->nd.L] != 1 || critic->nd.dims[0] != b->obs_dim) return crux_fail(c, CRUX_EINVAL, "fill_gae!: critic must map obs(%d) -> 1 (@assert length(Vs) == 1)", b->obs_dim);
  const int64_t n = b->elements; if (n == 0) return CRUX_OK;
  const size_t vb = ((4 * (size_t)n + 255) / 256) * 256;
  char* sc = (char*)crux_scratch(c, 2 * vb + 256);
  if (!sc) return crux_fail(c, CRUX_ENOMEM, "fill_gae!: scratch");
  float* Vs = (float*)sc; float* Vsp = (float*)(sc + vb); int32_t* flag = (int32_t*)(sc + 2 * vb);
  HIPCHK(c, hipMemsetAsync(flag, 0, 4, c->stream));
  crux_prof_begin(c, CRUX_PROF_VALUES);
  int32_t rc = values(critic, (const float*)b->col[CRUX_COL_S], n, Vs); if (rc) return rc;
  rc = values(critic, (const float*)b->col[CRUX_COL_SP], n, Vsp); if (rc) return rc;
  crux_prof_end(c, CRUX_PROF_VALUES);
  crux_prof_begin(c, CRUX_PROF_GAE);
  hipLaunchKernelGGL(k_gae_returns, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const float*)b->col[source], (const uint8_t*)b->col[CRUX_COL_DONE],
                     (const uint8_t*)b->col[CRUX_COL_EPISODE_END], Vs, Vsp, lambda, gamma, n, (float*)b->col[target], (float*)nullptr, flag);
  crux_prof_end(c, CRUX_PROF_GAE);
  rc = crux_launch_check(c, "k_gae_returns"); if (rc) return rc;
  int32_t h = 0;
  HIPCHK(c, hipMemcpyAsync(&h, flag, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (h) return crux_fail(c, CRUX_ENAN, "fill_gae!: NaN advantage (@assert !isnan(A))");
  return CRUX_OK;
}
int32_t crux_fill_gae(crux_buffer* b, crux_mlp* critic, float lambda, float gamma) { return crux_fill_gae_keys(b, critic, lambda, gamma, CRUX_COL_R, CRUX_COL_ADVANTAGE); }

// fill_gae! + fill_returns! for n buffers (the tail of a batched rollout): everything is enqueued, ONE host synchronisation reads the n NaN flags
int32_t crux_fill_gae_multi(int32_t n, crux_buffer* const* bufs, crux_mlp* const* critics, float lambda, float gamma, int32_t with_returns) {
  if (n < 1 || !bufs || !critics) return CRUX_EINVAL;
  crux_ctx* c = bufs[0]->ctx; size_t vb = 0;
  for (int i = 0; i < n; ++i) {
    crux_buffer* b = bufs[i]; crux_mlp* critic = critics[i]; if (!b || !critic) return CRUX_EINVAL;
    if (!has_col(b, CRUX_COL_ADVANTAGE)) return crux_fail(c, CRUX_EINVAL, "fill_gae!: buffer %d has no :advantage column", i);
    if (with_returns && !has_col(b, CRUX_COL_RETURN)) return crux_fail(c, CRUX_EINVAL, "fill_returns!: buffer %d has no :return column", i);
    if (critic->nd.dims[critic->nd.L] != 1 || critic->nd.dims[0] != b->obs_dim) return crux_fail(c, CRUX_EINVAL, "fill_gae!: critic must map obs(%d) -> 1 (@assert length(Vs) == 1)", b->obs_dim);
    const size_t v = ((4 * (size_t)b->elements + 255) / 256) * 256; if (v > vb) vb = v;
  }
  bool same = true;
  for (int i = 1; i < n; ++i) same = same && bufs[i]->elements == bufs[0]->elements && critics[i]->nd.n_params == critics[0]->nd.n_params && critics[i]->nd.L == critics[0]->nd.L &&
                                     memcmp(critics[i]->nd.dims, critics[0]->nd.dims, sizeof critics[0]->nd.dims) == 0 && memcmp(critics[i]->nd.acts, critics[0]->nd.acts, sizeof critics[0]->nd.acts) == 0;
  const size_t jb = (((sizeof(crux_fwd_job) * 2 + sizeof(GaeJob)) * (size_t)n) + 255) / 256 * 256, fb = (((size_t)n * 4 + 255) / 256) * 256;
  char* sc = (char*)crux_scratch(c, fb + jb + (size_t)n * 2 * vb + 256); if (!sc) return crux_fail(c, CRUX_ENOMEM, "fill_gae! (multi): scratch");
  int32_t* flags = (int32_t*)sc; char* vals = sc + fb + jb;
  HIPCHK(c, hipMemsetAsync(flags, 0, 4 * (size_t)n, c->stream));
  const int64_t m0 = bufs[0]->elements;
  if (same && m0 > 0) {     // equal shapes: 2 launches for the 2 n critic evaluations and 1 for the n scans (grid.y = buffer)
    std::vector<char> hj(jb); crux_fwd_job* fj = (crux_fwd_job*)hj.data(); GaeJob* gj = (GaeJob*)(hj.data() + sizeof(crux_fwd_job) * 2 * (size_t)n);
    for (int i = 0; i < n; ++i) { crux_buffer* b = bufs[i]; float* Vs = (float*)(vals + (size_t)i * 2 * vb); float* Vsp = (float*)(vals + (size_t)i * 2 * vb + vb);
      fj[2 * i] = {critics[i]->p, (const float*)b->col[CRUX_COL_S], Vs}; fj[2 * i + 1] = {critics[i]->p, (const float*)b->col[CRUX_COL_SP], Vsp};
      gj[i] = {(const float*)b->col[CRUX_COL_R], (const uint8_t*)b->col[CRUX_COL_DONE], (const uint8_t*)b->col[CRUX_COL_EPISODE_END], Vs, Vsp, (float*)b->col[CRUX_COL_ADVANTAGE],
               with_returns ? (float*)b->col[CRUX_COL_RETURN] : (float*)nullptr, flags + i}; }
    HIPCHK(c, hipMemcpyAsync(sc + fb, hj.data(), jb, hipMemcpyHostToDevice, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream));
    crux_prof_begin(c, CRUX_PROF_VALUES);
    int32_t rcf = crux_mlp_forward_multi_impl(c, critics[0]->nd, (const crux_fwd_job*)(sc + fb), 2 * n, m0); if (rcf) return rcf;
    crux_prof_end(c, CRUX_PROF_VALUES);
    crux_prof_begin(c, CRUX_PROF_GAE);
    hipLaunchKernelGGL(k_gae_returns_multi, dim3((unsigned)((m0 + 255) / 256), (unsigned)n), dim3(256), 0, c->stream, (const GaeJob*)(sc + fb + sizeof(crux_fwd_job) * 2 * (size_t)n), lambda, gamma, m0);
    crux_prof_end(c, CRUX_PROF_GAE);
  } else {
  crux_prof_begin(c, CRUX_PROF_VALUES);
  for (int i = 0; i < n; ++i) { crux_buffer* b = bufs[i]; const int64_t m = b->elements; if (m == 0) continue;
    float* Vs = (float*)(vals + (size_t)i * 2 * vb); float* Vsp = (float*)(vals + (size_t)i * 2 * vb + vb);
    int32_t rc = values(critics[i], (const float*)b->col[CRUX_COL_S], m, Vs); if (rc) return rc;
    rc = values(critics[i], (const float*)b->col[CRUX_COL_SP], m, Vsp); if (rc) return rc; }
  crux_prof_end(c, CRUX_PROF_VALUES);
  crux_prof_begin(c, CRUX_PROF_GAE);
  for (int i = 0; i < n; ++i) { crux_buffer* b = bufs[i]; const int64_t m = b->elements; if (m == 0) continue;
    float* Vs = (float*)(vals + (size_t)i * 2 * vb); float* Vsp = (float*)(vals + (size_t)i * 2 * vb + vb);
    hipLaunchKernelGGL(k_gae_returns, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream, (const float*)b->col[CRUX_COL_R], (const uint8_t*)b->col[CRUX_COL_DONE],
                       (const uint8_t*)b->col[CRUX_COL_EPISODE_END], (const float*)Vs, (const float*)Vsp, lambda, gamma, m, (float*)b->col[CRUX_COL_ADVANTAGE],
                       with_returns ? (float*)b->col[CRUX_COL_RETURN] : (float*)nullptr, flags + i); }
  crux_prof_end(c, CRUX_PROF_GAE);
  }
  int32_t rc = crux_launch_check(c, "k_gae_returns (multi)"); if (rc) return rc;
  std::vector<int32_t> h((size_t)n);
  HIPCHK(c, hipMemcpyAsync(h.data(), flags, 4 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i < n; ++i) if (h[(size_t)i]) return crux_fail(c, CRUX_ENAN, "fill_gae!: NaN advantage in buffer %d (@assert !isnan(A))", i);
  return CRUX_OK;
}

int32_t crux_whiten_multi(int32_t n, crux_buffer* const* bufs, int32_t key) {
  if (n < 1 || !bufs) return CRUX_EINVAL;
  crux_ctx* c = bufs[0]->ctx; const int64_t len = bufs[0]->elements;
  if (len > (int64_t)JLW_MAXLEAF * 512) return crux_fail(c, CRUX_EUNSUP, "whiten (multi): %lld elements per buffer (the pairwise reduction is laid out for at most %d leaves)", (long long)len, JLW_MAXLEAF);      // the same bound as crux_whiten: k_whiten_multi keeps the leaf tables in LDS
  std::vector<float*> hp((size_t)n);
  for (int i = 0; i < n; ++i) { crux_buffer* b = bufs[i];
    if (!b || !has_col(b, key) || col_elem(b, key) != 4 || col_rows(b, key) != 1) return crux_fail(c, CRUX_EINVAL, "whiten: column %d of buffer %d is not a 1 x N Float32 column", key, i);
    if (b->elements != len || len < 2) return crux_fail(c, CRUX_EINVAL, "whiten (multi): buffers must hold the same number (>= 2) of elements");
    hp[(size_t)i] = (float*)b->col[key]; }
  float** dp = (float**)crux_scratch(c, 8 * (size_t)n + 256); if (!dp) return crux_fail(c, CRUX_ENOMEM, "whiten (multi): scratch");
  HIPCHK(c, hipMemcpyAsync(dp, hp.data(), 8 * (size_t)n, hipMemcpyHostToDevice, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream));
  crux_prof_begin(c, CRUX_PROF_WHITEN);
  hipLaunchKernelGGL(k_whiten_multi, dim3((unsigned)n), dim3(1024), 0, c->stream, (float* const*)dp, len);
  crux_prof_end(c, CRUX_PROF_WHITEN);
  return crux_launch_check(c, "k_whiten_multi");
}

// fill_gae!(data, ep, V, lambda, gamma) / fill_returns!(data, ep, gamma) as terminate_episode! applies them to the block a steps! call just
// produced (sampler.jl:53-57,140-148), on the ring rows [first_row, first_row + n_rows) mod capacity that the block was pushed to.
int32_t crux_fill_gae_rows_keys(crux_buffer* b, crux_mlp* critic, float lambda, float gamma, int64_t first_row, int64_t n_rows, int64_t rows_per_env, int32_t close_last, int32_t source, int32_t target) {
  if (!b || !critic) return CRUX_EINVAL;
  crux_ctx* c = b->ctx;
  if (!f32_scalar_col(b, source) || !f32_scalar_col(b, target)) return crux_fail(c, CRUX_EINVAL, "fill_gae!: buffer lacks the source / target column (%d -> %d)", source, target);
  if (critic->nd.dims[critic->nd.L] != 1 || critic->nd.dims[0] != b->obs_dim) return crux_fail(c, CRUX_EINVAL, "fill_gae!: critic must map obs(%d) -> 1 (@assert length(Vs) == 1)", b->obs_dim);
  const int64_t C = b->capacity, n = n_rows;
  if (first_row < 0 || first_row >= C || n < 0 || n > C) return crux_fail(c, CRUX_EINVAL, "fill_gae!: rows [%lld, +%lld) outside the ring of %lld", (long long)first_row, (long long)n, (long long)C);
  if (n == 0) return CRUX_OK;
  const size_t vb = ((4 * (size_t)n + 255) / 256) * 256;
  char* sc = (char*)crux_scratch(c, 2 * vb + 256);
  if (!sc) return crux_fail(c, CRUX_ENOMEM, "fill_gae!: scratch");
  float* Vs = (float*)sc; float* Vsp = (float*)(sc + vb); int32_t* flag = (int32_t*)(sc + 2 * vb);
  HIPCHK(c, hipMemsetAsync(flag, 0, 4, c->stream));
  const int64_t n1 = first_row + n <= C ? n : C - first_row, n2 = n - n1; const int od = b->obs_dim;
  crux_prof_begin(c, CRUX_PROF_VALUES);
  int32_t rc = values(critic, (const float*)b->col[CRUX_COL_S] + first_row * od, n1, Vs); if (rc) return rc;
  rc = values(critic, (const float*)b->col[CRUX_COL_SP] + first_row * od, n1, Vsp); if (rc) return rc;
  if (n2 > 0) { rc = values(critic, (const float*)b->col[CRUX_COL_S], n2, Vs + n1); if (rc) return rc;
    rc = values(critic, (const float*)b->col[CRUX_COL_SP], n2, Vsp + n1); if (rc) return rc; }
  crux_prof_end(c, CRUX_PROF_VALUES);
  crux_prof_begin(c, CRUX_PROF_GAE);
  hipLaunchKernelGGL(k_gae_returns_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const float*)b->col[source], (const uint8_t*)b->col[CRUX_COL_DONE],
                     (const uint8_t*)b->col[CRUX_COL_EPISODE_END], (const float*)Vs, (const float*)Vsp, lambda, gamma, n, (float*)b->col[target], (float*)nullptr, flag,
                     first_row, C, close_last ? 1 : 0, rows_per_env);
  crux_prof_end(c, CRUX_PROF_GAE);
  rc = crux_launch_check(c, "k_gae_returns_rows"); if (rc) return rc;
  int32_t h = 0;
  HIPCHK(c, hipMemcpyAsync(&h, flag, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (h) return crux_fail(c, CRUX_ENAN, "fill_gae!: NaN advantage (@assert !isnan(A))");
  return CRUX_OK;
}
int32_t crux_fill_gae_rows(crux_buffer* b, crux_mlp* critic, float lambda, float gamma, int64_t first_row, int64_t n_rows, int64_t rows_per_env, int32_t close_last) { return crux_fill_gae_rows_keys(b, critic, lambda, gamma, first_row, n_rows, rows_per_env, close_last, CRUX_COL_R, CRUX_COL_ADVANTAGE); }
int32_t crux_fill_returns_rows_keys(crux_buffer* b, float gamma, int64_t first_row, int64_t n_rows, int64_t rows_per_env, int32_t close_last, int32_t source, int32_t target) {
  if (!b) return CRUX_EINVAL;
  crux_ctx* c = b->ctx;
  if (!f32_scalar_col(b, source) || !f32_scalar_col(b, target)) return crux_fail(c, CRUX_EINVAL, "fill_returns!: buffer lacks the source / target column (%d -> %d)", source, target);
  const int64_t C = b->capacity, n = n_rows;
  if (first_row < 0 || first_row >= C || n < 0 || n > C) return crux_fail(c, CRUX_EINVAL, "fill_returns!: rows [%lld, +%lld) outside the ring of %lld", (long long)first_row, (long long)n, (long long)C);
  if (n == 0) return CRUX_OK;
  int32_t* flag = (int32_t*)crux_scratch(c, 256);
  if (!flag) return crux_fail(c, CRUX_ENOMEM, "fill_returns!: scratch");
  crux_prof_begin(c, CRUX_PROF_GAE);
  hipLaunchKernelGGL(k_gae_returns_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const float*)b->col[source], (const uint8_t*)b->col[CRUX_COL_DONE],
                     (const uint8_t*)b->col[CRUX_COL_EPISODE_END], (const float*)nullptr, (const float*)nullptr, 0.f, gamma, n, (float*)nullptr, (float*)b->col[target], flag,
                     first_row, C, close_last ? 1 : 0, rows_per_env);
  crux_prof_end(c, CRUX_PROF_GAE);
  return crux_launch_check(c, "k_gae_returns_rows");
}
int32_t crux_fill_returns_rows(crux_buffer* b, float gamma, int64_t first_row, int64_t n_rows, int64_t rows_per_env, int32_t close_last) { return crux_fill_returns_rows_keys(b, gamma, first_row, n_rows, rows_per_env, close_last, CRUX_COL_R, CRUX_COL_RETURN); }

int32_t crux_fill_returns_keys(crux_buffer* b, float gamma, int32_t source, int32_t target) {
  if (!b) return CRUX_EINVAL;
  crux_ctx* c = b->ctx;
  if (!f32_scalar_col(b, source) || !f32_scalar_col(b, target)) return crux_fail(c, CRUX_EINVAL, "fill_returns!: buffer lacks the source / target column (%d -> %d)", source, target);
  const int64_t n = b->elements; if (n == 0) return CRUX_OK;
  int32_t* flag = (int32_t*)crux_scratch(c, 256);
  if (!flag) return crux_fail(c, CRUX_ENOMEM, "fill_returns!: scratch");
  crux_prof_begin(c, CRUX_PROF_GAE);
  hipLaunchKernelGGL(k_gae_returns, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const float*)b->col[source], (const uint8_t*)b->col[CRUX_COL_DONE],
                     (const uint8_t*)b->col[CRUX_COL_EPISODE_END], (const float*)nullptr, (const float*)nullptr, 0.f, gamma, n, (float*)nullptr, (float*)b->col[target], flag);
  crux_prof_end(c, CRUX_PROF_GAE);
  return crux_launch_check(c, "k_gae_returns");
}
int32_t crux_fill_returns(crux_buffer* b, float gamma) { return crux_fill_returns_keys(b, gamma, CRUX_COL_R, CRUX_COL_RETURN); }

int32_t crux_whiten(crux_buffer* b, int32_t key) {
  if (!b) return CRUX_EINVAL;
  crux_ctx* c = b->ctx;
  if (!has_col(b, key) || col_elem(b, key) != 4 || col_rows(b, key) != 1) return crux_fail(c, CRUX_EINVAL, "whiten: column %d is not a 1 x N Float32 column", key);
  const int64_t n = b->elements;
  if (n < 2) return crux_fail(c, CRUX_EINVAL, "whiten: need at least 2 elements");
  if (n > (int64_t)JLW_MAXLEAF * 512) return crux_fail(c, CRUX_EUNSUP, "whiten: %lld elements (the pairwise reduction is laid out for at most %d leaves)", (long long)n, JLW_MAXLEAF);
  crux_prof_begin(c, CRUX_PROF_WHITEN);
  hipLaunchKernelGGL(k_whiten, dim3(1), dim3(1024), 0, c->stream, (float*)b->col[key], n);
  crux_prof_end(c, CRUX_PROF_WHITEN);
  return crux_launch_check(c, "k_whiten");
}

}  // extern "C"

// ---- importance-weight columns (src/sampler.jl:58-62,108-111,283-308) ------------------------------------------------------------------------
// exp.(logpdf(pa, s, a) .- logprob): z = the nominal policy's outputs of the rows [nout x n]
__global__ void k_importance_weight(const float* __restrict__ z, int nout, const void* __restrict__ act, int act_kind, const float* __restrict__ ls, const float* __restrict__ lp, int64_t n, float* __restrict__ iw) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; if (j >= n) return;
  float nom;
  if (act_kind == CRUX_ACTION_DISCRETE) {      // categorical_logpdf (policies.jl:128-135): log(sum(softmax(z) .* onehot))
    const uint8_t* a = (const uint8_t*)act + j * nout; const float* zz = z + j * nout;
    float mx = zz[0]; for (int k = 1; k < nout; ++k) mx = zz[k] > mx ? zz[k] : mx;
    float sum = 0.f; for (int k = 0; k < nout; ++k) sum = __fadd_rn(sum, expf(__fsub_rn(zz[k], mx)));
    float q = 0.f; for (int k = 0; k < nout; ++k) q = __fadd_rn(q, __fmul_rn(__fdiv_rn(expf(__fsub_rn(zz[k], mx)), sum), a[k] ? 1.f : 0.f));
    nom = logf(q);
  } else {                                     // gaussian_logpdf (policies.jl:333-336)
    const float* a = (const float*)act + j * nout; const float* mu = z + j * nout; nom = 0.f;
    for (int d = 0; d < nout; ++d) { const float sg = expf(ls[d]), s2 = __fmul_rn(sg, sg), df = __fsub_rn(a[d], mu[d]);
      nom = __fadd_rn(nom, __fsub_rn(__fsub_rn(-(__fmul_rn(df, df)) / __fmul_rn(2.f, s2), 0.9189385332046727f), ls[d])); }
  }
  iw[j] = expf(__fsub_rn(nom, lp[j]));
}
// one thread per episode (the one sitting on its last row), as gae_returns_body: rev walks the episode backwards, fwd forwards, cum is the product of the whole episode
__global__ void k_importance_fills(const float* __restrict__ iw, const uint8_t* __restrict__ ee, int64_t n, float* __restrict__ fwd, float* __restrict__ cum, float* __restrict__ rev,
                                   int64_t first, int64_t cap, int close_last, int64_t seg) {
  if (cap <= 0) cap = n;
  if (seg <= 0) seg = n;
  auto phys = [&](int64_t k) -> int64_t { const int64_t q = first + k; return q >= cap ? q - cap : q; };
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const bool closed = ee[phys(i)];
    if (!(closed || i == n - 1 || (i + 1) % seg == 0)) continue;
    const int64_t k0 = (i / seg) * seg;
    int64_t start = i; while (start > k0 && !ee[phys(start - 1)]) --start;
    if (!closed && !close_last) {            // an episode still open at the end of the segment: terminate_episode! never reached it, the fresh data block holds ones
      for (int64_t k = start; k <= i; ++k) { const int64_t q = phys(k); if (fwd) fwd[q] = 1.f; if (cum) cum[q] = 1.f; if (rev) rev[q] = 1.f; }
      continue;
    }
    float w = 1.f;
    if (rev) for (int64_t k = i; k >= start; --k) { const int64_t q = phys(k); w = __fmul_rn(iw[q], w); rev[q] = w; }      // :301-308
    w = 1.f;
    for (int64_t k = start; k <= i; ++k) { const int64_t q = phys(k); w = __fmul_rn(iw[q], w); if (fwd) fwd[q] = w; }     // :283-290
    if (cum) for (int64_t k = start; k <= i; ++k) cum[phys(k)] = w;                                                          // :292-299
  }
}
extern "C" int32_t crux_importance_weight_rows(crux_buffer* b, crux_mlp* nominal, int32_t head, int64_t first_row, int64_t n_rows) {
  if (!b || !nominal) return CRUX_EINVAL;
  crux_ctx* c = b->ctx;
  if (!has_col(b, CRUX_COL_IMPORTANCE_WEIGHT) || !has_col(b, CRUX_COL_LOGPROB)) return crux_fail(c, CRUX_EINVAL, "importance_weight: the buffer needs :importance_weight and :logprob columns");
  if (n_rows == 0) return CRUX_OK;
  if (first_row < 0 || n_rows < 0 || first_row + n_rows > b->capacity) return crux_fail(c, CRUX_EINVAL, "importance_weight: rows [%lld, %lld) outside the buffer (a wrapped range takes two calls)", (long long)first_row, (long long)(first_row + n_rows));
  const int nout = nominal->nd.dims[nominal->nd.L];
  const bool disc = b->act_kind == CRUX_ACTION_DISCRETE;
  if (nominal->nd.dims[0] != b->obs_dim || nout != b->act_dim || (disc ? head != CRUX_HEAD_CATEGORICAL : (head != CRUX_HEAD_GAUSSIAN || nominal->nd.n_extra != nout)))
    return crux_fail(c, CRUX_EINVAL, "importance_weight: the nominal policy must map obs(%d) -> %d with a %s head", b->obs_dim, b->act_dim, disc ? "categorical" : "Gaussian (logSigma extras)");
  if (nominal->squash > 0.f) return crux_fail(c, CRUX_EUNSUP, "importance_weight: SquashedGaussianPolicy nominal policies are not wired up");
  float* z = (float*)crux_scratch(c, 4 * (size_t)n_rows * nout + 256); if (!z) return crux_fail(c, CRUX_ENOMEM, "importance_weight: scratch");
  int32_t rc = crux_mlp_forward_impl(nominal, (const float*)b->col[CRUX_COL_S] + (size_t)first_row * b->obs_dim, n_rows, z, nullptr); if (rc) return rc;
  const char* act = (const char*)b->col[CRUX_COL_A] + (size_t)first_row * col_stride(b, CRUX_COL_A);
  hipLaunchKernelGGL(k_importance_weight, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, c->stream, (const float*)z, nout, (const void*)act, b->act_kind, disc ? (const float*)nullptr : (const float*)(nominal->p + nominal->nd.xoff),
                     (const float*)b->col[CRUX_COL_LOGPROB] + first_row, n_rows, (float*)b->col[CRUX_COL_IMPORTANCE_WEIGHT] + first_row);
  return crux_launch_check(c, "k_importance_weight");
}
extern "C" int32_t crux_fill_importance_weights_rows(crux_buffer* b, int64_t first_row, int64_t n_rows, int64_t rows_per_env, int32_t close_last) {
  if (!b) return CRUX_EINVAL;
  crux_ctx* c = b->ctx;
  const bool hf = has_col(b, CRUX_COL_FWD_IMPORTANCE_WEIGHT), hc = has_col(b, CRUX_COL_CUM_IMPORTANCE_WEIGHT), hr = has_col(b, CRUX_COL_REV_IMPORTANCE_WEIGHT);
  if (!hf && !hc && !hr) return CRUX_OK;
  if (!has_col(b, CRUX_COL_IMPORTANCE_WEIGHT)) return crux_fail(c, CRUX_EINVAL, "fill_*_importance_weight!: the buffer has no :importance_weight column (@assert haskey(data, :importance_weight), sampler.jl:284)");
  if (n_rows == 0) return CRUX_OK;
  if (first_row < 0 || first_row >= b->capacity || n_rows < 0 || n_rows > b->capacity) return crux_fail(c, CRUX_EINVAL, "fill_*_importance_weight!: rows out of range");
  hipLaunchKernelGGL(k_importance_fills, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, c->stream, (const float*)b->col[CRUX_COL_IMPORTANCE_WEIGHT], (const uint8_t*)b->col[CRUX_COL_EPISODE_END], n_rows,
                     hf ? (float*)b->col[CRUX_COL_FWD_IMPORTANCE_WEIGHT] : (float*)nullptr, hc ? (float*)b->col[CRUX_COL_CUM_IMPORTANCE_WEIGHT] : (float*)nullptr, hr ? (float*)b->col[CRUX_COL_REV_IMPORTANCE_WEIGHT] : (float*)nullptr,
                     first_row, b->capacity, close_last, rows_per_env);
  return crux_launch_check(c, "k_importance_fills");
}
