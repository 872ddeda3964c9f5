/* crux_oracle.c -- TEST INFRASTRUCTURE ONLY (see crux_oracle.h for the pinning statement).
 *
 * Plain-C, single-threaded restatement of the sisl/Crux.jl hot path. Compile with
 * -ffp-contract=off: Julia does not contract a*b+c into FMA, and the sequential Float32 scans below
 * (GAE, returns, PER leaf sums) are bit-defined by that.
 * All `file:line` citations are into /root/reference (sisl/Crux.jl v0.1.4).
 */
#include "crux_oracle.h"
#include "../include/crux_rng.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#ifdef _OPENMP          /* libcruxoracle_omp.so: the all-cores CPU baseline of bench.py. The parity oracle is built WITHOUT -fopenmp. */
#include <omp.h>
#define ORC_STR_(x) #x
#define ORC_OMP(x) _Pragma(ORC_STR_(x))
int32_t orc_omp_threads(void) { return (int32_t)omp_get_max_threads(); }
#else
#define ORC_OMP(x)
int32_t orc_omp_threads(void) { return 1; }
#endif

#define MAXL 8
static const float EPS32 = 1.1920928955078125e-07f; /* eps(Float32) */

/* ============================================================================================
 * networks: Chain(Dense...) [3P: Flux.Dense = act.(W*x .+ b); accumulation order of W*x is BLAS-defined
 * in the reference; restated here as a k-ordered fp32 sum]                       src/policies.jl:68-157
 * ============================================================================================ */
struct orc_mlp {
  int32_t n_layers, dims[MAXL + 1], acts[MAXL], n_extra;
  int64_t n_params;
  float *p, *g, *m, *v;
  double eta, b1, b2, eps, bp[2];
  int has_adam;
  float squash;      /* > 0: SquashedGaussianPolicy with this ascale (policies.jl:353-400) instead of GaussianPolicy */
};

static int64_t woff(const orc_mlp* n, int l) { int64_t o = 0; for (int i = 0; i < l; ++i) o += (int64_t)n->dims[i + 1] * n->dims[i] + n->dims[i + 1]; return o; }
static int64_t boff(const orc_mlp* n, int l) { return woff(n, l) + (int64_t)n->dims[l + 1] * n->dims[l]; }
static int64_t xoff(const orc_mlp* n) { return woff(n, n->n_layers); }

orc_mlp* orc_mlp_create(int32_t n_layers, const int32_t* dims, const int32_t* acts, int32_t n_extra) {
  if (n_layers < 0 || n_layers > MAXL || (n_layers == 0 && n_extra < 1)) return NULL;   /* 0 layers = a bare trainable vector (ConstantLayer, utils.jl:31-36; SAC_log_alpha, sac.jl:96) */
  orc_mlp* n = (orc_mlp*)calloc(1, sizeof(orc_mlp));
  n->n_layers = n_layers; n->n_extra = n_extra;
  for (int i = 0; i <= n_layers; ++i) n->dims[i] = dims[i];
  for (int i = 0; i < n_layers; ++i) n->acts[i] = acts[i];
  n->n_params = xoff(n) + n_extra;
  n->p = (float*)calloc(n->n_params, 4); n->g = (float*)calloc(n->n_params, 4);
  n->m = (float*)calloc(n->n_params, 4); n->v = (float*)calloc(n->n_params, 4);
  return n;
}
int32_t orc_mlp_set_squash(orc_mlp* n, float ascale) { if (!n || !(ascale >= 0.f)) return CRUX_EINVAL; n->squash = ascale; return CRUX_OK; }
/* SquashedGaussianPolicy arithmetic (policies.jl:374-396): sigma = exp(clamp(logSigma, -5, 2)); for the un-tanh'd action u
 *   logprob = sum_d[-(u-mu)^2/(2 sigma^2) - 0.9189385 - logSigma - 2(log 2 - u - softplus(-2u))],  softplus(x) = log1p(exp(-|x|)) + relu(x) (NNlib)
 * logpdf(pi, s, a) uses u = atanh(clamp(a/ascale, -1+1f-5, 1-1f-5)) (:396). */
static float sq_softplus(float x) { return log1pf(expf(-fabsf(x))) + (x > 0.f ? x : 0.f); }
static float sq_corr(float u) { return 2.f * (logf(2.0f) - u - sq_softplus(-2.f * u)); }
static float sq_clampls(float ls) { return ls < -5.f ? -5.f : ls > 2.f ? 2.f : ls; }
static float sq_untanh(float a, float ascale) { float t = a / ascale; float lo = -1.0f + 1.0e-5f, hi = 1.0f - 1.0e-5f; t = t < lo ? lo : t > hi ? hi : t; return atanhf(t); }
void orc_mlp_destroy(orc_mlp* n) { if (!n) return; free(n->p); free(n->g); free(n->m); free(n->v); free(n); }
int64_t orc_mlp_n_params(const orc_mlp* n) { return n->n_params; }
float* orc_mlp_params(orc_mlp* n) { return n->p; }
float* orc_mlp_grads(orc_mlp* n) { return n->g; }

/* [3P] Flux.glorot_uniform: (rand(Float32,out,in) .- 0.5f0) .* sqrt(24f0/(in+out)); bias zeros. */
int32_t orc_mlp_init_glorot(orc_mlp* n, uint64_t seed, uint32_t stream, float extra_init) {
  uint64_t ctr = 0;
  for (int l = 0; l < n->n_layers; ++l) {
    int in = n->dims[l], out = n->dims[l + 1];
    float scale = sqrtf(24.0f / (float)(in + out));
    float* W = n->p + woff(n, l);
    for (int64_t e = 0; e < (int64_t)in * out; ++e, ++ctr) {
      crux_u32x4 x = crux_philox(seed, ctr, stream, CRUX_RNG_INIT);
      W[e] = (crux_u32_to_f32(x.v[0]) - 0.5f) * scale;
    }
    memset(n->p + boff(n, l), 0, 4 * (size_t)out);
  }
  for (int i = 0; i < n->n_extra; ++i) n->p[xoff(n) + i] = extra_init;
  return CRUX_OK;
}

static float act_f(int a, float z) { return a == CRUX_ACT_RELU ? (z > 0.f ? z : 0.f) : a == CRUX_ACT_TANH ? tanhf(z) : z; }

/* forward for one column; h[l] receives the post-activation of layer l (h[0] = input copy). */
static void fwd_col(const orc_mlp* n, const float* x, float** h) {
  memcpy(h[0], x, 4 * (size_t)n->dims[0]);
  for (int l = 0; l < n->n_layers; ++l) {
    int in = n->dims[l], out = n->dims[l + 1];
    const float* W = n->p + woff(n, l); const float* b = n->p + boff(n, l);
    for (int o = 0; o < out; ++o) {
      float acc = 0.f;
      for (int k = 0; k < in; ++k) acc = acc + W[o + (int64_t)out * k] * h[l][k];
      h[l + 1][o] = act_f(n->acts[l], acc + b[o]);
    }
  }
}

typedef struct { float* h[MAXL + 1]; } colcache;
static colcache cc_alloc(const orc_mlp* n) { colcache c; memset(&c, 0, sizeof c); for (int l = 0; l <= n->n_layers; ++l) c.h[l] = (float*)malloc(4 * (size_t)n->dims[l]); return c; }
static void cc_free(const orc_mlp* n, colcache* c) { for (int l = 0; l <= n->n_layers; ++l) free(c->h[l]); }

int32_t orc_mlp_forward(orc_mlp* n, const float* x, int64_t B, float* y) {
  int in = n->dims[0], out = n->dims[n->n_layers];
  ORC_OMP(omp parallel if (B >= 256))       /* the OpenMP build (bench.py's all-cores baseline) spreads the columns; the parity build runs this block once */
  { colcache c = cc_alloc(n);
    ORC_OMP(omp for schedule(static))
    for (int64_t s = 0; s < B; ++s) { fwd_col(n, x + s * in, c.h); memcpy(y + s * out, c.h[n->n_layers], 4 * (size_t)out); }
    cc_free(n, &c); }
  return CRUX_OK;
}

/* accumulate d(loss)/d(params) for one column given d(loss)/d(output) = dy (reverse-mode, the
 * pullback Zygote builds for Chain(Dense...) at src/training.jl:16-18). */
static void bwd_col(const orc_mlp* n, float** h, const float* dy, float* g) {
  float d0[1024], d1[1024];
  float *d = d0, *dn = d1;
  memcpy(d, dy, 4 * (size_t)n->dims[n->n_layers]);
  for (int l = n->n_layers - 1; l >= 0; --l) {
    int in = n->dims[l], out = n->dims[l + 1];
    const float* W = n->p + woff(n, l);
    float* gW = g + woff(n, l); float* gb = g + boff(n, l);
    for (int o = 0; o < out; ++o) {
      float y = h[l + 1][o];
      if (n->acts[l] == CRUX_ACT_RELU) d[o] = y > 0.f ? d[o] : 0.f;   /* relu'(0) = 0 */
      else if (n->acts[l] == CRUX_ACT_TANH) d[o] = d[o] * (1.f - y * y);
    }
    for (int o = 0; o < out; ++o) gb[o] += d[o];
    for (int k = 0; k < in; ++k) for (int o = 0; o < out; ++o) gW[o + (int64_t)out * k] += d[o] * h[l][k];
    if (l > 0) {
      for (int k = 0; k < in; ++k) { float acc = 0.f; for (int o = 0; o < out; ++o) acc = acc + W[o + (int64_t)out * k] * d[o]; dn[k] = acc; }
      float* t = d; d = dn; dn = t;
    }
  }
}

int32_t orc_mlp_copy(orc_mlp* to, const orc_mlp* from) { /* copyto!(to, from) src/policies.jl:61-65 */
  if (to->n_params != from->n_params) return CRUX_EINVAL;
  memcpy(to->p, from->p, 4 * (size_t)to->n_params); return CRUX_OK;
}
int32_t orc_polyak(orc_mlp* to, const orc_mlp* from, float tau) { /* src/policies.jl:48-59 */
  if (to->n_params != from->n_params) return CRUX_EINVAL;
  float omt = 1.0f - tau;
  for (int64_t i = 0; i < to->n_params; ++i) to->p[i] = tau * from->p[i] + omt * to->p[i];
  return CRUX_OK;
}

/* [3P] Flux.Optimise.Adam: eta/beta/eps are Float64 fields; every broadcast is evaluated in Float64
 * per element and rounded to Float32 on store. */
int32_t orc_adam_init(orc_mlp* n, double eta, double b1, double b2, double eps) {
  n->eta = eta; n->b1 = b1; n->b2 = b2; n->eps = eps; n->bp[0] = b1; n->bp[1] = b2; n->has_adam = 1;
  memset(n->m, 0, 4 * (size_t)n->n_params); memset(n->v, 0, 4 * (size_t)n->n_params);
  return CRUX_OK;
}
int32_t orc_adam_get_state(orc_mlp* n, float* m, float* v, double* bp) {
  if (m) memcpy(m, n->m, 4 * (size_t)n->n_params);
  if (v) memcpy(v, n->v, 4 * (size_t)n->n_params);
  if (bp) { bp[0] = n->bp[0]; bp[1] = n->bp[1]; }
  return CRUX_OK;
}
/* test infrastructure for the replica group's periodic form (local SGD twin): overwrite the Adam moments (the beta powers are kept when bp == NULL) */
int32_t orc_adam_set_state(orc_mlp* n, const float* m, const float* v, const double* bp) {
  if (!n->has_adam) return CRUX_EINVAL;
  if (m) memcpy(n->m, m, 4 * (size_t)n->n_params);
  if (v) memcpy(n->v, v, 4 * (size_t)n->n_params);
  if (bp) { n->bp[0] = bp[0]; n->bp[1] = bp[1]; }
  return CRUX_OK;
}
int32_t orc_adam_apply(orc_mlp* n, float grad_scale) {
  if (!n->has_adam) return CRUX_EINVAL;
  for (int64_t i = 0; i < n->n_params; ++i) {
    float gi = n->g[i] * grad_scale;
    double g = (double)gi;
    n->m[i] = (float)(n->b1 * (double)n->m[i] + (1.0 - n->b1) * g);
    n->v[i] = (float)(n->b2 * (double)n->v[i] + ((1.0 - n->b2) * g) * g);
    float d = (float)((double)n->m[i] / (1.0 - n->bp[0]) / (sqrt((double)n->v[i] / (1.0 - n->bp[1])) + n->eps) * n->eta);
    n->p[i] = n->p[i] - d;
  }
  n->bp[0] *= n->b1; n->bp[1] *= n->b2;
  return CRUX_OK;
}

/* ============================================================================================
 * ExperienceBuffer                                                src/experience_buffer.jl:4-35,53-80
 * ============================================================================================ */
struct orc_buffer {
  int32_t obs_dim, act_dim, act_kind;
  int64_t capacity, elements, next_ind, total_count;   /* next_ind 0-based */
  uint32_t mask;
  void* col[CRUX_NCOLS];
  int prioritized; float alpha;
  float* priorities; float* cumsum; int cumsum_valid; float max_priority, min_priority;
  int64_t* indices; int64_t n_indices;
  uint32_t sample_stream;   /* Philox stream of the draws that sample FROM this buffer (crux_buffer_set_sample_stream) */
};

static int col_elem(const orc_buffer* b, int k) {
  switch (k) { case CRUX_COL_A: return b->act_kind == CRUX_ACTION_DISCRETE ? 1 : 4;
    case CRUX_COL_DONE: case CRUX_COL_EPISODE_END: return 1; case CRUX_COL_T: case CRUX_COL_I: return 8; default: return 4; }
}
static int col_rows(const orc_buffer* b, int k) { return (k == CRUX_COL_S || k == CRUX_COL_SP) ? b->obs_dim : k == CRUX_COL_A ? b->act_dim : 1; }
static size_t col_stride(const orc_buffer* b, int k) { return (size_t)col_elem(b, k) * (size_t)col_rows(b, k); }

int32_t orc_per_update(orc_buffer* b, const int64_t* I, const void* v, int32_t v_is_f64, int64_t n);

orc_buffer* orc_buffer_create(int32_t obs_dim, int32_t act_dim, int32_t act_kind, int64_t capacity, uint32_t column_mask,
                              int32_t prioritized, float alpha) {
  orc_buffer* b = (orc_buffer*)calloc(1, sizeof(orc_buffer));
  b->obs_dim = obs_dim; b->act_dim = act_dim; b->act_kind = act_kind; b->capacity = capacity;
  b->mask = column_mask | 0x3Fu;
  if (prioritized) b->mask |= 1u << CRUX_COL_WEIGHT;            /* experience_buffer.jl:71 */
  for (int k = 0; k < CRUX_NCOLS; ++k) if (b->mask & (1u << k)) {
    b->col[k] = calloc((size_t)capacity, col_stride(b, k));
    if (CRUX_COL_INIT_ONE(k)) for (int64_t i = 0; i < capacity; ++i) ((float*)b->col[k])[i] = 1.0f;   /* :17-19 ones */
  }
  b->prioritized = prioritized; b->alpha = alpha;
  if (prioritized) { b->priorities = (float*)calloc((size_t)capacity, 4); b->cumsum = (float*)calloc((size_t)capacity, 4);
    b->max_priority = 1.0f; b->min_priority = INFINITY; }       /* PriorityParams :38-50 */
  b->indices = (int64_t*)calloc((size_t)capacity, 8);
  return b;
}
void orc_buffer_destroy(orc_buffer* b) { if (!b) return; for (int k = 0; k < CRUX_NCOLS; ++k) free(b->col[k]); free(b->priorities); free(b->cumsum); free(b->indices); free(b); }
int64_t orc_buffer_len(const orc_buffer* b) { return b->elements; }
int64_t orc_buffer_capacity(const orc_buffer* b) { return b->capacity; }
int64_t orc_buffer_next_ind(const orc_buffer* b) { return b->next_ind; }
int64_t orc_buffer_total_count(const orc_buffer* b) { return b->total_count; }
int32_t orc_buffer_has_column(const orc_buffer* b, int32_t k) { return k >= 0 && k < CRUX_NCOLS && (b->mask & (1u << k)) ? 1 : 0; }
int32_t orc_buffer_column_info(const orc_buffer* b, int32_t k, int32_t* eb, int32_t* rows) {
  if (!orc_buffer_has_column(b, k)) return CRUX_EINVAL; if (eb) *eb = col_elem(b, k); if (rows) *rows = col_rows(b, k); return CRUX_OK; }
void* orc_buffer_column(orc_buffer* b, int32_t k) { return orc_buffer_has_column(b, k) ? b->col[k] : NULL; }

int32_t orc_buffer_clear(orc_buffer* b) { /* clear! :97-104 */
  b->elements = 0; b->next_ind = 0; b->n_indices = 0; b->total_count = 0;
  if (b->prioritized) { memset(b->priorities, 0, 4 * (size_t)b->capacity); b->cumsum_valid = 0; b->min_priority = INFINITY; /* PriorityParams(N, pp) keeps alpha,beta,max_priority */ }
  return CRUX_OK;
}

/* mod1.(start:start+N-1, C) in 0-based form -- test/experience_buffer_tests.jl:23-28 */
void orc_circ_inds(int64_t start0, int64_t n, int64_t C, int64_t* out) { for (int64_t j = 0; j < n; ++j) out[j] = (start0 + j) % C; }

static void ring_advance(orc_buffer* b, int64_t N) { /* :256-257 */
  b->elements = b->elements + N < b->capacity ? b->elements + N : b->capacity;
  b->next_ind = (b->next_ind + N) % b->capacity;
}

static void per_on_push(orc_buffer* b, const int64_t* I, int64_t N) { /* :254 update_priorities!(b, I, max_priority*ones(N)) -> Float64 values */
  if (!b->prioritized) return;
  double* v = (double*)malloc(8 * (size_t)(N > 0 ? N : 1));
  for (int64_t j = 0; j < N; ++j) v[j] = (double)b->max_priority * 1.0;
  orc_per_update(b, I, v, 1, N); free(v);
}

/* push!(b, data) :232-259 */
int32_t orc_buffer_push_host(orc_buffer* b, int64_t N, const void* const* cols, int64_t* I_out) {
  if (N < 0) return CRUX_EINVAL;
  int64_t C = b->capacity;
  int64_t* I = (int64_t*)malloc(8 * (size_t)(N > 0 ? N : 1));
  b->total_count += N;
  orc_circ_inds(b->next_ind, N, C, I);
  for (int k = 0; k < CRUX_NCOLS; ++k) {
    if (!(b->mask & (1u << k)) || !cols || !cols[k]) continue;     /* :238-241 missing keys skipped */
    size_t st = col_stride(b, k);
    for (int64_t j = 0; j < N; ++j) memcpy((char*)b->col[k] + (size_t)I[j] * st, (const char*)cols[k] + (size_t)j * st, st);
  }
  per_on_push(b, I, N);
  ring_advance(b, N);
  if (I_out) memcpy(I_out, I, 8 * (size_t)N);
  free(I); return CRUX_OK;
}

/* push_reservoir!(buffer, data; weighted) (src/experience_buffer.jl:262-288), element by element:
 *   weighted && haskey(:weight) && rand() > weight[i]  -> skip
 *   total_count += 1; if length < capacity: push!(buffer, element)  (which adds 1 to total_count AGAIN, :235 -- reproduced)
 *   else j = rand(1:total_count); j <= capacity -> every column's slot j is overwritten (priorities are NOT touched on this branch)
 * Randomness (crux_rng.h): x = Philox(seed, counter + i, 0, RESERVOIR); rand() = f64(x0,x1); rand(1:n) = 1 + floor(u64(x2,x3) * n / 2^64). */
int32_t orc_buffer_push_reservoir(orc_buffer* b, int64_t N, const void* const* cols, int32_t weighted, uint64_t seed, uint64_t counter) {
  if (N < 0 || !cols) return CRUX_EINVAL;
  const float* W = (weighted && (b->mask & (1u << CRUX_COL_WEIGHT)) && cols[CRUX_COL_WEIGHT]) ? (const float*)cols[CRUX_COL_WEIGHT] : NULL;
  for (int64_t i = 0; i < N; ++i) {
    crux_u32x4 x = crux_philox(seed, counter + (uint64_t)i, 0, CRUX_RNG_RESERVOIR);
    if (W && crux_u32x2_to_f64(x.v[0], x.v[1]) > (double)W[i]) continue;
    b->total_count += 1;
    if (b->elements < b->capacity) {
      const void* one[CRUX_NCOLS];
      for (int k = 0; k < CRUX_NCOLS; ++k) one[k] = ((b->mask & (1u << k)) && cols[k]) ? (const char*)cols[k] + (size_t)i * col_stride(b, k) : NULL;
      int32_t rc = orc_buffer_push_host(b, 1, one, NULL); if (rc) return rc;
    } else {
      const uint64_t r = ((uint64_t)x.v[2] << 32) | (uint64_t)x.v[3];
      const int64_t j = 1 + (int64_t)(((unsigned __int128)r * (unsigned __int128)(uint64_t)b->total_count) >> 64);
      if (j <= b->capacity)
        for (int k = 0; k < CRUX_NCOLS; ++k) { if (!(b->mask & (1u << k)) || !cols[k]) continue; size_t st = col_stride(b, k);
          memcpy((char*)b->col[k] + (size_t)(j - 1) * st, (const char*)cols[k] + (size_t)i * st, st); }
    }
  }
  return CRUX_OK;
}

/* push!(target, source, ids=ids): v2 = collect(view(source, ids)) is materialised BEFORE the copy
 * (:249-252), which is what makes the self-push of test/experience_buffer_tests.jl:141-146 well defined. */
int32_t orc_buffer_push_buffer(orc_buffer* dst, const orc_buffer* src, const int64_t* ids, int64_t N, int64_t* I_out) {
  if (dst->obs_dim != src->obs_dim || dst->act_dim != src->act_dim || dst->act_kind != src->act_kind) return CRUX_EINVAL;
  int64_t C = dst->capacity;
  int64_t* I = (int64_t*)malloc(8 * (size_t)(N > 0 ? N : 1));
  dst->total_count += N;
  orc_circ_inds(dst->next_ind, N, C, I);
  for (int k = 0; k < CRUX_NCOLS; ++k) {
    if (!(dst->mask & (1u << k)) || !(src->mask & (1u << k))) continue;
    size_t st = col_stride(dst, k);
    char* tmp = (char*)malloc(st * (size_t)(N > 0 ? N : 1));
    for (int64_t j = 0; j < N; ++j) { int64_t id = ids ? ids[j] : j; if (id < 0 || id >= src->capacity) { free(tmp); free(I); return CRUX_EINVAL; }
      memcpy(tmp + (size_t)j * st, (const char*)src->col[k] + (size_t)id * st, st); }
    for (int64_t j = 0; j < N; ++j) memcpy((char*)dst->col[k] + (size_t)I[j] * st, tmp + (size_t)j * st, st);
    free(tmp);
  }
  per_on_push(dst, I, N);
  ring_advance(dst, N);
  if (I_out) memcpy(I_out, I, 8 * (size_t)N);
  free(I); return CRUX_OK;
}

/* shuffle!(b) with explicit permutation :118-124: b[k] .= bslice(b[k], new_i) */
int32_t orc_buffer_permute(orc_buffer* b, const int64_t* perm) {
  int64_t n = b->elements;
  for (int k = 0; k < CRUX_NCOLS; ++k) {
    if (!(b->mask & (1u << k))) continue;
    size_t st = col_stride(b, k);
    char* tmp = (char*)malloc(st * (size_t)(n > 0 ? n : 1));
    for (int64_t j = 0; j < n; ++j) { if (perm[j] < 0 || perm[j] >= n) { free(tmp); return CRUX_EINVAL; } memcpy(tmp + (size_t)j * st, (char*)b->col[k] + (size_t)perm[j] * st, st); }
    memcpy(b->col[k], tmp, st * (size_t)n); free(tmp);
  }
  return CRUX_OK;
}

/* get_last_N_indices :223-229 */
int64_t orc_buffer_last_n_indices(const orc_buffer* b, int64_t N, int64_t* out) {
  if (N > b->elements) N = b->elements;
  int64_t C = b->capacity;
  int64_t start = ((b->next_ind - N) % C + C) % C;   /* mod1(next_ind - N, C) in 0-based */
  for (int64_t j = 0; j < N; ++j) out[j] = (start + j) % C;
  return N;
}

int32_t orc_buffer_gather_host(orc_buffer* b, const int64_t* ids, int64_t n, void* const* outs) { /* minibatch_copy :171 */
  for (int k = 0; k < CRUX_NCOLS; ++k) {
    if (!(b->mask & (1u << k)) || !outs[k]) continue;
    size_t st = col_stride(b, k);
    for (int64_t j = 0; j < n; ++j) { if (ids[j] < 0 || ids[j] >= b->capacity) return CRUX_EINVAL; memcpy((char*)outs[k] + (size_t)j * st, (char*)b->col[k] + (size_t)ids[j] * st, st); }
  }
  return CRUX_OK;
}
int32_t orc_buffer_indices(const orc_buffer* b, int64_t* out, int64_t n) { if (n > b->n_indices) n = b->n_indices; memcpy(out, b->indices, 8 * (size_t)n); return CRUX_OK; }

/* episodes(b) via :episode_end :194-212 */
int64_t orc_buffer_episodes(const orc_buffer* b, int64_t* starts, int64_t* ends, int64_t max_eps) {
  const uint8_t* ee = (const uint8_t*)b->col[CRUX_COL_EPISODE_END];
  int64_t n = b->elements, ne = 0, st = 0;
  for (int64_t j = 0; j < n; ++j) if (ee[j]) { if (ne < max_eps) { starts[ne] = st; ends[ne] = j; } ++ne; st = j + 1; }
  if (n > 0 && st < n) { if (ne < max_eps) { starts[ne] = st; ends[ne] = n - 1; } ++ne; }   /* :207-211 */
  return ne;
}

/* split_batches :126-131 */
void orc_split_batches(int64_t N, const double* fracs, int32_t nf, int64_t* out) {
  int64_t sum = 0; for (int i = 0; i < nf; ++i) { out[i] = (int64_t)floor((double)N * fracs[i]); sum += out[i]; }
  out[0] += N - sum;
}

/* ============================================================================================
 * prioritized replay                                              src/experience_buffer.jl:290-349
 * ============================================================================================ */
/* [3P] Julia: Float64^Float32 promotes to Float64 pow; Float32^Float32 is computed through Float64
 * and rounded once. Both are (float)pow((double)val,(double)alpha) on the already-typed val. */
int32_t orc_per_update(orc_buffer* b, const int64_t* I, const void* v, int32_t v_is_f64, int64_t n) {
  if (!b->prioritized) return CRUX_EINVAL;
  for (int64_t i = 0; i < n; ++i) {
    if (I[i] < 0 || I[i] >= b->capacity) return CRUX_EINVAL;
    double val;
    if (v_is_f64) val = ((const double*)v)[i] + (double)EPS32;              /* :293 v[i] + eps(Float32) */
    else { float vf = ((const float*)v)[i] + EPS32; val = (double)vf; }
    b->priorities[I[i]] = (float)pow(val, (double)b->alpha);                  /* :294 */
    /* :297-298 max/min track the UN-powered val; the fields are Float32 */
    b->max_priority = (float)fmax(val, (double)b->max_priority);
    b->min_priority = (float)fmin(val, (double)b->min_priority);
    b->cumsum_valid = 0;                                                      /* :299 */
  }
  return CRUX_OK;
}

/* [3P] Base.cumsum(::Vector{Float32}) = accumulate_pairwise!(add_sum, ...). */
static float pw_rec(const float* v, float* c, float s, int64_t i1, int64_t n) {
  if (n < 128) {
    float s_ = v[i1]; c[i1] = s + s_;
    for (int64_t i = i1 + 1; i < i1 + n; ++i) { s_ = s_ + v[i]; c[i] = s + s_; }
    return s_;
  }
  int64_t n2 = n >> 1;
  float s_ = pw_rec(v, c, s, i1, n2);
  s_ = s_ + pw_rec(v, c, s + s_, i1 + n2, n - n2);
  return s_;
}
void orc_pairwise_cumsum_f32(const float* v, int64_t n, float* out) {
  if (n <= 0) return;
  out[0] = v[0];
  if (n > 1) {
    /* accumulate_pairwise!: s_ = v[1]; c[1] = s_; s_ += rec(c, v, s_, 2, n-1) */
    pw_rec(v, out, v[0], 1, n - 1);
  }
}

static int64_t searchsortedfirst_f32_f64(const float* c, int64_t n, double key) { /* first idx with c[idx] >= key, n if none */
  int64_t lo = 0, hi = n;
  while (lo < hi) { int64_t mid = lo + ((hi - lo) >> 1); if ((double)c[mid] < key) lo = mid + 1; else hi = mid; }
  return lo;
}

int32_t orc_per_sample(orc_buffer* target, orc_buffer* source, int64_t B, const double* rands, float beta, uint64_t i, uint64_t seed) {
  if (!source->prioritized || !(source->mask & (1u << CRUX_COL_WEIGHT))) return CRUX_EINVAL;   /* :325 */
  int64_t N = source->elements;
  if (N <= 0 || B <= 0 || B > target->capacity) return CRUX_EINVAL;
  if (!source->cumsum_valid) { orc_pairwise_cumsum_f32(source->priorities, N, source->cumsum); source->cumsum_valid = 1; }   /* :329-332 */
  float ptot = source->cumsum[N - 1];
  float dp = ptot / (float)B;                                                   /* :335 Float32 / Int */
  for (int64_t j = 0; j < B; ++j) {
    double u;
    if (rands) u = rands[j];
    else { crux_u32x4 x = crux_philox(seed, i * (uint64_t)B + (uint64_t)j, source->sample_stream, CRUX_RNG_SAMPLE); u = crux_u32x2_to_f64(x.v[0], x.v[1]); }
    double key = ((double)(j + 1) + u - 1.0) * (double)dp;                      /* :340 (j + rands[j] - 1) * dp */
    int64_t idx = searchsortedfirst_f32_f64(source->cumsum, N, key);
    if (idx >= N) idx = N - 1;   /* the reference would index out of bounds; clamp and document */
    target->indices[j] = idx;
  }
  target->n_indices = B;
  float pmin = source->min_priority / ptot;                                     /* :343 */
  float max_w = powf(pmin * (float)N, -beta);
  float* w = (float*)source->col[CRUX_COL_WEIGHT];
  for (int64_t j = 0; j < B; ++j) { int64_t id = target->indices[j];
    w[id] = powf(((float)N * source->priorities[id]) / ptot, beta) / max_w; }   /* :346 */
  return orc_buffer_push_buffer(target, source, target->indices, B, NULL);      /* :348 */
}

int32_t orc_uniform_sample(orc_buffer* target, orc_buffer* source, int64_t B, const int64_t* ids, uint64_t i, uint64_t seed) { /* :317-321 */
  int64_t N = source->elements;
  if (N <= 0 || B <= 0 || B > target->capacity) return CRUX_EINVAL;
  for (int64_t j = 0; j < B; ++j) {
    if (ids) target->indices[j] = ids[j];
    else { crux_u32x4 x = crux_philox(seed, i * (uint64_t)B + (uint64_t)j, source->sample_stream, CRUX_RNG_SAMPLE);
      target->indices[j] = (int64_t)(((uint64_t)x.v[0] * (uint64_t)N) >> 32); }
  }
  target->n_indices = B;
  return orc_buffer_push_buffer(target, source, target->indices, B, NULL);
}

int32_t orc_buffer_set_sample_stream(orc_buffer* b, uint32_t stream) { b->sample_stream = stream; return CRUX_OK; }

int32_t orc_per_get(orc_buffer* b, float* pr, float* maxp, float* minp, float* cs) {
  if (!b->prioritized) return CRUX_EINVAL;
  if (pr) memcpy(pr, b->priorities, 4 * (size_t)b->capacity);
  if (maxp) *maxp = b->max_priority; if (minp) *minp = b->min_priority;
  if (cs) { if (!b->cumsum_valid) { orc_pairwise_cumsum_f32(b->priorities, b->elements, b->cumsum); b->cumsum_valid = 1; } memcpy(cs, b->cumsum, 4 * (size_t)b->elements); }
  return CRUX_OK;
}

/* ============================================================================================
 * environments. The reference steps gymnasium through POMDPGym/PyCall (src/sampler.jl:89-97); the
 * dynamics are restated from the public gymnasium definitions and pinned by the recordings under
 * examples/il/expert_data/ (tests/golden/ npz files).
 * ============================================================================================ */
#define MAXSD 32
struct orc_env {
  int32_t kind, n_envs, max_steps, obs_dim, act_dim, state_dim;
  float gamma; float mu[32], sigma[32]; uint64_t seed;
  double* state; int64_t* ep_len; int64_t* n_resets; int64_t* steps_taken; float* svec;
};

static void env_dims(int kind, int so, int sa, int* obs, int* act, int* sd) {
  switch (kind) { case CRUX_ENV_CARTPOLE: *obs = 4; *act = 2; *sd = 4; break;
    case CRUX_ENV_PENDULUM: *obs = 3; *act = 1; *sd = 2; break;
    case CRUX_ENV_GRIDWORLD: *obs = 2; *act = 4; *sd = 2; break;
    default: *obs = so; *act = sa; *sd = so; break; }
}

/* gymnasium CartPole-v1 (classic_control/cartpole.py), float64 Euler step. one-hot index 1 (0-based)
 * = push right, confirmed by cartpole.bson. */
static void cartpole_step(const double* s, int action, double* sn, float* r, uint8_t* done) {
  const double gravity = 9.8, masscart = 1.0, masspole = 0.1, total_mass = masspole + masscart, length = 0.5,
               polemass_length = masspole * length, force_mag = 10.0, tau = 0.02;
  double x = s[0], x_dot = s[1], theta = s[2], theta_dot = s[3];
  double force = action == 1 ? force_mag : -force_mag;
  double costheta = cos(theta), sintheta = sin(theta);
  double temp = (force + polemass_length * (theta_dot * theta_dot) * sintheta) / total_mass;
  double thetaacc = (gravity * sintheta - costheta * temp) / (length * (4.0 / 3.0 - masspole * (costheta * costheta) / total_mass));
  double xacc = temp - polemass_length * thetaacc * costheta / total_mass;
  x = x + tau * x_dot; x_dot = x_dot + tau * xacc; theta = theta + tau * theta_dot; theta_dot = theta_dot + tau * thetaacc;
  sn[0] = x; sn[1] = x_dot; sn[2] = theta; sn[3] = theta_dot;
  const double xth = 2.4, thth = 12.0 * 2.0 * M_PI / 360.0;
  *done = (x < -xth || x > xth || theta < -thth || theta > thth) ? 1 : 0;
  *r = 1.0f;
}
static void cartpole_obs(const double* s, float* o) { for (int i = 0; i < 4; ++i) o[i] = (float)s[i]; }

/* gymnasium Pendulum-v1 (classic_control/pendulum.py): g=10, m=1, l=1, dt=0.05, max_speed 8, max_torque 2. */
static double angle_normalize(double x) { double t = fmod(x + M_PI, 2.0 * M_PI); if (t < 0) t += 2.0 * M_PI; return t - M_PI; }
static void pendulum_step(const double* s, float a, double* sn, float* r, uint8_t* done) {
  const double g = 10.0, m = 1.0, l = 1.0, dt = 0.05, max_speed = 8.0, max_torque = 2.0;
  double th = s[0], thdot = s[1];
  double u = (double)a; if (u < -max_torque) u = -max_torque; if (u > max_torque) u = max_torque;
  double an = angle_normalize(th);
  double costs = an * an + 0.1 * (thdot * thdot) + 0.001 * (u * u);
  double newthdot = thdot + (3.0 * g / (2.0 * l) * sin(th) + 3.0 / (m * l * l) * u) * dt;
  if (newthdot < -max_speed) newthdot = -max_speed; if (newthdot > max_speed) newthdot = max_speed;
  double newth = th + newthdot * dt;
  sn[0] = newth; sn[1] = newthdot; *r = (float)(-costs); *done = 0;
}
static void pendulum_obs(const double* s, float* o) { o[0] = (float)cos(s[0]); o[1] = (float)sin(s[0]); o[2] = (float)s[1]; }

/* [3P] POMDPModels.SimpleGridWorld (README example, configs[0]): 10x10 grid, rewards {(4,3):-10,(4,6):-5,(9,3):10,(8,8):3}, reward is
 * collected when acting FROM a reward cell, which then leads to the terminal state (-1,-1); otherwise the intended move
 * (0 up, 1 down, 2 left, 3 right) happens with probability tprob = 0.7 and one of the other three uniformly else; moves off the
 * grid leave the state unchanged. Restated from memory of the public package; dynamics parity is unpinned. */
static float gridworld_reward(double x, double y) {
  if (x == 4 && y == 3) return -10.f; if (x == 4 && y == 6) return -5.f; if (x == 9 && y == 3) return 10.f; if (x == 8 && y == 8) return 3.f; return 0.f;
}
static void gridworld_step(const double* s, int a, double u, double* sn, float* r, uint8_t* done) {
  const double tprob = 0.7; const int dx[4] = {0, 0, -1, 1}, dy[4] = {1, -1, 0, 0};
  double x = s[0], y = s[1];
  float rw = gridworld_reward(x, y);
  *r = rw;
  if (rw != 0.f) { sn[0] = -1; sn[1] = -1; *done = 1; return; }
  int dir = a;
  if (!(u < tprob)) { int k = (int)((u - tprob) / (1.0 - tprob) * 3.0); if (k > 2) k = 2; int cnt = 0; for (int d = 0; d < 4; ++d) { if (d == a) continue; if (cnt == k) { dir = d; break; } ++cnt; } }
  double nx = x + dx[dir], ny = y + dy[dir];
  if (nx < 1 || nx > 10 || ny < 1 || ny > 10) { nx = x; ny = y; }
  sn[0] = nx; sn[1] = ny; *done = 0;
}
/* SYNTH: the library's documented synthetic environment for configurations whose simulators cannot be restated (LunarLander 8 obs /
 * 4 discrete actions -> C3, HalfCheetah 17 obs / 6 continuous actions -> C5). State x in R^so (Float64), observation = Float32(x):
 *   u_i  = clamp(a[i mod sa], -1, 1)                       (continuous)      |  (i + k) mod sa == 0 ? 1 : -0.25   (discrete action k)
 *   x'_i = 0.9 x_i + 0.1 sin(x_{(i+1) mod so} + u_i);   r = -mean(x'^2) + 0.05 x'_0;   done = x'_0 > 0.9;   reset: x_i ~ U(-0.05, 0.05). */
static void synth_step(int so, int sa, int discrete, const double* s, int ai, const float* a, double* sn, float* r, uint8_t* done) {
  double ss = 0.0;
  for (int i = 0; i < so; ++i) {
    double u;
    if (discrete) u = ((i + ai) % sa == 0) ? 1.0 : -0.25;
    else { u = (double)a[i % sa]; if (u < -1.0) u = -1.0; if (u > 1.0) u = 1.0; }
    sn[i] = 0.9 * s[i] + 0.1 * sin(s[(i + 1) % so] + u);
    ss = ss + sn[i] * sn[i];
  }
  *r = (float)(-(ss / (double)so) + 0.05 * sn[0]);
  *done = sn[0] > 0.9 ? 1 : 0;
}
/* the cost channel of the restated environments (include/cruxhip.h): info["cost"] of the step that reached s' (sampler.jl:65-66,114) */
static float env_cost(int kind, int so, const double* sn) {
  if (kind == CRUX_ENV_SYNTH || kind == CRUX_ENV_SYNTH_DISCRETE) { double x = sn[1 % so]; return (float)(25.0 * (x * x)); }
  if (kind == CRUX_ENV_CARTPOLE) return fabs(sn[2]) > 0.05 ? 1.f : 0.f;
  if (kind == CRUX_ENV_PENDULUM) return fabs(sn[1]) > 4.0 ? 1.f : 0.f;
  return 0.f;
}
static void env_obs_n(int kind, int od, const double* s, float* o) {
  if (kind == CRUX_ENV_CARTPOLE) cartpole_obs(s, o); else if (kind == CRUX_ENV_PENDULUM) pendulum_obs(s, o);
  else if (kind == CRUX_ENV_SYNTH || kind == CRUX_ENV_SYNTH_DISCRETE) { for (int i = 0; i < od; ++i) o[i] = (float)s[i]; }
  else { o[0] = (float)s[0]; o[1] = (float)s[1]; }
}
static void env_obs(int kind, const double* s, float* o) { env_obs_n(kind, 0, s, o); }

static void env_reset_one(orc_env* e, int k) { /* reset_sampler! src/sampler.jl:31-43 */
  double* s = e->state + (size_t)k * e->state_dim;
  uint64_t c = (uint64_t)e->n_resets[k];
  crux_u32x4 a = crux_philox(e->seed, 2 * c, (uint32_t)k, CRUX_RNG_RESET), b = crux_philox(e->seed, 2 * c + 1, (uint32_t)k, CRUX_RNG_RESET);
  double u[4] = { crux_u32x2_to_f64(a.v[0], a.v[1]), crux_u32x2_to_f64(a.v[2], a.v[3]), crux_u32x2_to_f64(b.v[0], b.v[1]), crux_u32x2_to_f64(b.v[2], b.v[3]) };
  if (e->kind == CRUX_ENV_CARTPOLE) { for (int i = 0; i < 4; ++i) s[i] = -0.05 + 0.1 * u[i]; }          /* U(-0.05,0.05)^4 */
  else if (e->kind == CRUX_ENV_PENDULUM) { s[0] = -M_PI + 2.0 * M_PI * u[0]; s[1] = -1.0 + 2.0 * u[1]; } /* U(-pi,pi) x U(-1,1) */
  else if (e->kind == CRUX_ENV_SYNTH || e->kind == CRUX_ENV_SYNTH_DISCRETE) {                               /* U(-0.05,0.05)^so: two Float64 uniforms per Philox block */
    for (int i = 0; i < e->state_dim; ++i) { crux_u32x4 x = crux_philox(e->seed, 16 * c + (uint64_t)(i >> 1), (uint32_t)k, CRUX_RNG_RESET);
      double ui = (i & 1) ? crux_u32x2_to_f64(x.v[2], x.v[3]) : crux_u32x2_to_f64(x.v[0], x.v[1]); s[i] = -0.05 + 0.1 * ui; } }
  else { s[0] = 1.0 + floor(10.0 * u[0]); s[1] = 1.0 + floor(10.0 * u[1]); }                              /* uniform over the 100 cells */
  e->n_resets[k] += 1; e->ep_len[k] = 0;
  float o[32]; env_obs_n(e->kind, e->obs_dim, s, o);
  for (int i = 0; i < e->obs_dim; ++i) e->svec[(size_t)k * e->obs_dim + i] = (o[i] - e->mu[i]) / e->sigma[i];   /* tovec src/spaces.jl:25 */
}

orc_env* orc_env_create(int32_t kind, int32_t n_envs, int32_t max_steps, float gamma, const float* mu, const float* sigma, uint64_t seed, int32_t so, int32_t sa) {
  if (kind < CRUX_ENV_CARTPOLE || kind > CRUX_ENV_SYNTH_DISCRETE) return NULL;
  if ((kind == CRUX_ENV_SYNTH || kind == CRUX_ENV_SYNTH_DISCRETE) && (so < 1 || so > 32 || sa < 1 || sa > 32)) return NULL;
  orc_env* e = (orc_env*)calloc(1, sizeof(orc_env));
  e->kind = kind; e->n_envs = n_envs; e->max_steps = max_steps; e->gamma = gamma; e->seed = seed;
  env_dims(kind, so, sa, &e->obs_dim, &e->act_dim, &e->state_dim);
  for (int i = 0; i < e->obs_dim; ++i) { e->mu[i] = mu ? mu[i] : 0.f; e->sigma[i] = sigma ? sigma[i] : 1.f; }
  e->state = (double*)calloc((size_t)n_envs * e->state_dim, 8); e->ep_len = (int64_t*)calloc(n_envs, 8);
  e->n_resets = (int64_t*)calloc(n_envs, 8); e->steps_taken = (int64_t*)calloc(n_envs, 8); e->svec = (float*)calloc((size_t)n_envs * e->obs_dim, 4);
  for (int k = 0; k < n_envs; ++k) env_reset_one(e, k);
  return e;
}
void orc_env_destroy(orc_env* e) { if (!e) return; free(e->state); free(e->ep_len); free(e->n_resets); free(e->steps_taken); free(e->svec); free(e); }
int32_t orc_env_obs_dim(const orc_env* e) { return e->obs_dim; }
int32_t orc_env_act_dim(const orc_env* e) { return e->act_dim; }
int32_t orc_env_state_dim(const orc_env* e) { return e->state_dim; }
int32_t orc_env_reset(orc_env* e) { for (int k = 0; k < e->n_envs; ++k) env_reset_one(e, k); return CRUX_OK; }
int32_t orc_env_get_state(orc_env* e, double* st, int64_t* el, int64_t* nr) {
  if (st) memcpy(st, e->state, 8 * (size_t)e->n_envs * e->state_dim); if (el) memcpy(el, e->ep_len, 8 * (size_t)e->n_envs); if (nr) memcpy(nr, e->n_resets, 8 * (size_t)e->n_envs);
  return CRUX_OK;
}

int32_t orc_env_step_host(int32_t kind, int64_t n, const double* state, const void* action, const double* uniforms,
                          double* next_state, float* obs, float* r, uint8_t* done) {
  (void)uniforms;
  for (int64_t j = 0; j < n; ++j) {
    if (kind == CRUX_ENV_CARTPOLE) { const uint8_t* a = (const uint8_t*)action + 2 * j; int act = a[1] ? 1 : 0;
      cartpole_step(state + 4 * j, act, next_state + 4 * j, r + j, done + j); cartpole_obs(next_state + 4 * j, obs + 4 * j); }
    else if (kind == CRUX_ENV_PENDULUM) { pendulum_step(state + 2 * j, ((const float*)action)[j], next_state + 2 * j, r + j, done + j); pendulum_obs(next_state + 2 * j, obs + 3 * j); }
    else if (kind == CRUX_ENV_GRIDWORLD) { const uint8_t* a = (const uint8_t*)action + 4 * j; int act = 0; for (int q = 0; q < 4; ++q) if (a[q]) act = q;
      gridworld_step(state + 2 * j, act, uniforms ? uniforms[j] : 0.0, next_state + 2 * j, r + j, done + j); env_obs(kind, next_state + 2 * j, obs + 2 * j); }
    else return CRUX_EUNSUP;
  }
  return CRUX_OK;
}

/* [3P] NNlib.softmax over dims=1: subtract the column max, exp, divide by the column sum. */
static void softmax_col(const float* z, int n, float* p) {
  float mx = z[0]; for (int i = 1; i < n; ++i) if (z[i] > mx) mx = z[i];
  float sum = 0.f; for (int i = 0; i < n; ++i) { p[i] = expf(z[i] - mx); sum = sum + p[i]; }
  for (int i = 0; i < n; ++i) p[i] = p[i] / sum;
}

double orc_linear_decay(double start, double stop, int64_t steps, int64_t i) { /* utils.jl:122-126 */
  double rate = (start - stop) / (double)steps; double val = start - (double)i * rate; return val > stop ? val : stop;
}

/* standard normal from two uniforms (Box-Muller); the library's definition of randn(Float32). */
static float randn_f32(uint64_t seed, uint64_t ctr, uint32_t stream, int which) {
  crux_u32x4 x = crux_philox(seed, ctr, stream, CRUX_RNG_NOISE);
  double u1 = crux_u32x2_to_f64(x.v[0], x.v[1]), u2 = crux_u32x2_to_f64(x.v[2], x.v[3]);
  double rr = sqrt(-2.0 * log(1.0 - u1)), th = 2.0 * M_PI * u2;
  return (float)(which ? rr * sin(th) : rr * cos(th));
}

/* steps!(samplers, buffer; Nsteps=T, explore, reset, i) src/sampler.jl:139-173, env-major. */
/* exploration(pi_explore, svec; pi_on, i) / action(pi, svec) (sampler.jl:73) given the network outputs z of ONE observation: the action (aout: one-hot floats for the
 * discrete heads, the action vector otherwise; *ai_out the discrete index) and its log-probability. Draws: crux_philox(seed, f(ctr), stream, purpose), ctr = steps the sampler
 * has taken (crux_rng.h). Shared by orc_rollout and orc_policy_explore (the twin of crux_policy_explore, cruxhip.h). */
static void policy_head(const orc_mlp* pol, const crux_rollout_cfg* cfg, const float* z, int nout, int ad, uint64_t seed, uint64_t ctr, uint32_t k, uint64_t gi,
                        float* aout, int* ai_out, float* logprob_out) {
  float p[64]; float logprob = NAN; int ai = 0;
      if (cfg->head == CRUX_HEAD_CATEGORICAL || cfg->head == CRUX_HEAD_GREEDY_Q) {
        int greedy = 0; for (int q = 1; q < nout; ++q) if (z[q] > z[greedy]) greedy = q;      /* argmax, first on ties (policies.jl:124) */
        if (!cfg->explore) ai = greedy;
        else if (cfg->eps_steps > 0 || cfg->head == CRUX_HEAD_GREEDY_Q) {                       /* MixedPolicy policies.jl:474-494 */
          double eps = cfg->eps_steps > 0 ? orc_linear_decay(cfg->eps_start, cfg->eps_stop, cfg->eps_steps, (int64_t)gi) : 0.0;
          crux_u32x4 x = crux_philox(seed, ctr, k, CRUX_RNG_ACTION);
          double u = crux_u32x2_to_f64(x.v[0], x.v[1]);
          if (u < eps) { crux_u32x4 y = crux_philox(seed, ctr, k, CRUX_RNG_RANDACT); ai = (int)(((uint64_t)y.v[0] * (uint64_t)nout) >> 32); }
          else ai = greedy;
          /* p1 = eps*exp(logpdf(ObjectCategorical)) = eps/n ; p2 = 1-eps ; log(p1+p2) (policies.jl:485-493) */
          logprob = (float)log(eps * (1.0 / (double)nout) + (1.0 - eps));
        } else {                                                                                 /* DiscreteNetwork exploration policies.jl:137-142 */
          if (cfg->logit_div > 0.f) { float zs[64]; for (int q = 0; q < nout; ++q) zs[q] = z[q] / cfg->logit_div; softmax_col(zs, nout, p); }   /* softmax(value ./ alpha) softq.jl:53 */
          else softmax_col(z, nout, p);
          crux_u32x4 x = crux_philox(seed, ctr, k, CRUX_RNG_ACTION);
          float draw = crux_u32_to_f32(x.v[0]);
          /* [3P] Distributions rand(::DiscreteNonParametric): cp=p[1]; while cp <= draw && i<n: cp += p[++i] */
          float cp = p[0]; ai = 0; while (cp <= draw && ai < nout - 1) { ai += 1; cp = cp + p[ai]; }
          logprob = logf(p[ai]);                                                                  /* categorical_logpdf policies.jl:135 */
        }
        for (int q = 0; q < ad; ++q) aout[q] = (q == ai) ? 1.f : 0.f;
      } else if (cfg->head == CRUX_HEAD_GAUSSIAN) {                                              /* GaussianPolicy policies.jl:338-344 */
        const float* ls = pol->p + xoff(pol); float lp = 0.f;
        for (int q = 0; q < ad; ++q) {
          float mu = z[q];
          if (pol->squash > 0.f) {                                                              /* SquashedGaussianPolicy policies.jl:372,388-394 */
            if (cfg->explore) { float sg = expf(sq_clampls(ls[q])); float epsn = randn_f32(seed, ctr * (uint64_t)((ad + 1) / 2) + (uint64_t)(q / 2), k, q & 1);
              float u = epsn * sg + mu; float s2 = sg * sg; float dd = u - mu; aout[q] = pol->squash * tanhf(u);
              lp = lp + (((-(dd * dd) / (2.f * s2) - 0.9189385332046727f) - ls[q]) - sq_corr(u)); }
            else aout[q] = pol->squash * tanhf(mu);
          } else
          if (cfg->explore) { float sg = expf(ls[q]); float epsn = randn_f32(seed, ctr * (uint64_t)((ad + 1) / 2) + (uint64_t)(q / 2), k, q & 1);
            aout[q] = epsn * sg + mu; float s2 = sg * sg; float dd = aout[q] - mu; lp = lp + (-(dd * dd) / (2.f * s2) - 0.9189385332046727f - ls[q]); }
          else aout[q] = mu;
        }
        logprob = cfg->explore ? lp : NAN;
      } else {                                                                                   /* DETERMINISTIC (+ GaussianNoiseExplorationPolicy policies.jl:510-514) */
        for (int q = 0; q < ad; ++q) { float a = z[q];
          if (cfg->explore && cfg->noise_sigma >= 0.f) { float n0 = randn_f32(seed, ctr * (uint64_t)((ad + 1) / 2) + (uint64_t)(q / 2), k, q & 1) * cfg->noise_sigma;
            n0 = n0 < cfg->noise_eps_min ? cfg->noise_eps_min : n0 > cfg->noise_eps_max ? cfg->noise_eps_max : n0; a = a + n0;
            a = a < cfg->a_min ? cfg->a_min : a > cfg->a_max ? cfg->a_max : a; }
          aout[q] = a; }
      }
  if (cfg->explore == 2) logprob = NAN;   /* action(pi, s) of an always_stochastic policy = exploration(pi, s)[1] (policies.jl:124), logprob NaN (sampler.jl:73) */
  *ai_out = ai; *logprob_out = logprob;
}

int32_t orc_rollout(orc_env* e, orc_mlp* pol, const crux_rollout_cfg* cfg, orc_buffer* buf, int64_t T, double* sum_r, int64_t* n_ee) {
  int E = e->n_envs, od = e->obs_dim, ad = e->act_dim;
  int64_t N = (int64_t)E * T, C = buf->capacity;
  if (buf->obs_dim != od || buf->act_dim != ad || N > C) return CRUX_EINVAL;
  int nout = pol->dims[pol->n_layers];
  if (pol->dims[0] != od) return CRUX_EINVAL;
  int64_t* I = (int64_t*)malloc(8 * (size_t)N);
  orc_circ_inds(buf->next_ind, N, C, I);
  buf->total_count += N;
  float* S = (float*)buf->col[CRUX_COL_S]; float* SP = (float*)buf->col[CRUX_COL_SP]; float* R = (float*)buf->col[CRUX_COL_R];
  uint8_t* D = (uint8_t*)buf->col[CRUX_COL_DONE]; uint8_t* EE = (uint8_t*)buf->col[CRUX_COL_EPISODE_END];
  float* LP = (buf->mask & (1u << CRUX_COL_LOGPROB)) ? (float*)buf->col[CRUX_COL_LOGPROB] : NULL;
  int64_t* TT = (buf->mask & (1u << CRUX_COL_T)) ? (int64_t*)buf->col[CRUX_COL_T] : NULL;
  int64_t* II = (buf->mask & (1u << CRUX_COL_I)) ? (int64_t*)buf->col[CRUX_COL_I] : NULL;
  float* W = (buf->mask & (1u << CRUX_COL_WEIGHT)) ? (float*)buf->col[CRUX_COL_WEIGHT] : NULL;
  float* RET = (buf->mask & (1u << CRUX_COL_RETURN)) ? (float*)buf->col[CRUX_COL_RETURN] : NULL;
  float* ADV = (buf->mask & (1u << CRUX_COL_ADVANTAGE)) ? (float*)buf->col[CRUX_COL_ADVANTAGE] : NULL;
  float* COST = (buf->mask & (1u << CRUX_COL_COST)) ? (float*)buf->col[CRUX_COL_COST] : NULL;
  float* CADV = (buf->mask & (1u << CRUX_COL_COST_ADVANTAGE)) ? (float*)buf->col[CRUX_COL_COST_ADVANTAGE] : NULL;
  float* CRET = (buf->mask & (1u << CRUX_COL_COST_RETURN)) ? (float*)buf->col[CRUX_COL_COST_RETURN] : NULL;
  double sr = 0.0; int64_t nee = 0;
  /* environments are independent (own state, own Philox streams, own rows of the ring): the OpenMP build steps them on all cores */
  ORC_OMP(omp parallel for schedule(dynamic, 1) reduction(+ : sr, nee))
  for (int k = 0; k < E; ++k) {
    colcache c = cc_alloc(pol);
    float aout[64];
    double* st = e->state + (size_t)k * e->state_dim; float* sv = e->svec + (size_t)k * od;
    for (int64_t t = 0; t < T; ++t) {
      int64_t j = I[(int64_t)k * T + t];
      uint64_t gi = cfg->i0 + (uint64_t)t * (uint64_t)E + (uint64_t)k;     /* i + (j-1), j env-minor (sampler.jl:161-163) */
      uint64_t ctr = (uint64_t)e->steps_taken[k];
      /* ---- action + logprob: exploration(pi_explore, svec) / action(pi, svec)   sampler.jl:73 */
      fwd_col(pol, sv, c.h); const float* z = c.h[pol->n_layers];
      float logprob; int ai;
      policy_head(pol, cfg, z, nout, ad, e->seed, ctr, (uint32_t)k, gi, aout, &ai, &logprob);
      /* ---- env transition @gen(:sp,:r) + isterminal                            sampler.jl:93-97 */
      double sn[MAXSD]; float r; uint8_t done; float o[32], spv[32];
      if (e->kind == CRUX_ENV_CARTPOLE) cartpole_step(st, ai, sn, &r, &done);
      else if (e->kind == CRUX_ENV_PENDULUM) pendulum_step(st, aout[0], sn, &r, &done);
      else if (e->kind == CRUX_ENV_SYNTH || e->kind == CRUX_ENV_SYNTH_DISCRETE) synth_step(od, ad, e->kind == CRUX_ENV_SYNTH_DISCRETE, st, ai, aout, sn, &r, &done);
      else { crux_u32x4 xd = crux_philox(e->seed, ctr, (uint32_t)k, CRUX_RNG_ENVDYN); gridworld_step(st, ai, crux_u32x2_to_f64(xd.v[0], xd.v[1]), sn, &r, &done); }
      env_obs_n(e->kind, od, sn, o);
      for (int q = 0; q < od; ++q) spv[q] = (o[q] - e->mu[q]) / e->sigma[q];
      /* ---- column writes                                                       sampler.jl:100-107 */
      memcpy(S + (size_t)j * od, sv, 4 * (size_t)od);
      if (buf->act_kind == CRUX_ACTION_DISCRETE) { uint8_t* A = (uint8_t*)buf->col[CRUX_COL_A] + (size_t)j * ad; for (int q = 0; q < ad; ++q) A[q] = aout[q] != 0.f; }
      else memcpy((float*)buf->col[CRUX_COL_A] + (size_t)j * ad, aout, 4 * (size_t)ad);
      memcpy(SP + (size_t)j * od, spv, 4 * (size_t)od);
      R[j] = r; D[j] = done; EE[j] = 0;
      if (LP) LP[j] = logprob;
      if (TT) TT[j] = e->ep_len[k] + 1;
      if (II) II[j] = (int64_t)gi + 1;
      if (W) W[j] = 1.0f;             /* fresh mdp_data :weight is ones (experience_buffer.jl:17-19) */
      if (RET) RET[j] = 0.f; if (ADV) ADV[j] = 0.f;
      if (COST) COST[j] = env_cost(e->kind, od, sn);                                             /* data[:cost][1,j] = info["cost"] sampler.jl:114 */
      if (CADV) CADV[j] = 0.f; if (CRET) CRET[j] = 0.f;
      sr += (double)r;
      e->steps_taken[k] += 1;
      /* ---- episode bookkeeping                                                 sampler.jl:130-136 */
      e->ep_len[k] += 1;
      if (done || e->ep_len[k] >= e->max_steps) { EE[j] = 1; ++nee; env_reset_one(e, k); }     /* terminate_episode! :53-69 */
      else { memcpy(st, sn, 8 * (size_t)e->state_dim); memcpy(sv, spv, 4 * (size_t)od); }
    }
    if (cfg->reset_at_end && e->ep_len[k] > 0) {                                               /* sampler.jl:148 */
      int64_t j = I[(int64_t)k * T + T - 1]; if (!EE[j]) { EE[j] = 1; ++nee; } env_reset_one(e, k);
    }
    cc_free(pol, &c);
  }
  per_on_push(buf, I, N);
  ring_advance(buf, N);
  if (sum_r) *sum_r = sr; if (n_ee) *n_ee = nee;
  free(I);
  return CRUX_OK;
}

/* crux_policy_explore (cruxhip.h): the first line of step! (sampler.jl:73) for n_envs caller-stepped samplers -- the policy forward on each observation column, then the head. */
int32_t orc_policy_explore(orc_mlp* pol, const crux_rollout_cfg* cfg, int32_t n_envs, const float* obs, uint64_t seed, const int64_t* steps_taken, void* actions_out, float* logprob_out) {
  if (!pol || !cfg || !obs || !actions_out || n_envs < 1) return CRUX_EINVAL;
  int od = pol->dims[0], nout = pol->dims[pol->n_layers];
  int disc = cfg->head == CRUX_HEAD_CATEGORICAL || cfg->head == CRUX_HEAD_GREEDY_Q;
  colcache c = cc_alloc(pol);
  for (int k = 0; k < n_envs; ++k) {
    float aout[64]; int ai; float logprob;
    fwd_col(pol, obs + (size_t)k * od, c.h);
    policy_head(pol, cfg, c.h[pol->n_layers], nout, nout, seed, steps_taken ? (uint64_t)steps_taken[k] : 0, (uint32_t)k, cfg->i0 + (uint64_t)k, aout, &ai, &logprob);
    if (disc) { uint8_t* A = (uint8_t*)actions_out + (size_t)k * nout; for (int q = 0; q < nout; ++q) A[q] = aout[q] != 0.f; }
    else memcpy((float*)actions_out + (size_t)k * nout, aout, 4 * (size_t)nout);
    if (logprob_out) logprob_out[k] = logprob;
  }
  cc_free(pol, &c);
  return CRUX_OK;
}

/* ============================================================================================
 * advantage pipeline                                                    src/sampler.jl:255-281
 * ============================================================================================ */
/* fill_gae!(d, range, V, lambda, gamma) :262-273. A = c*A + r + (1f0-done)*gamma*Vsp - Vs evaluated
 * left to right in Float32: ((c*A + r) + ((1-done)*gamma)*Vsp) - Vs. */
void orc_gae_range(const float* r, const uint8_t* done, const float* Vs, const float* Vsp, int64_t start, int64_t stop,
                   float lambda, float gamma, float* adv) {
  float A = 0.f, c = lambda * gamma;
  for (int64_t i = stop; i >= start; --i) {
    float t1 = c * A; float t2 = t1 + r[i];
    float t3 = ((1.f - (done[i] ? 1.f : 0.f)) * gamma) * Vsp[i];
    float t4 = t2 + t3; A = t4 - Vs[i];
    adv[i] = A;
  }
}
void orc_returns_range(const float* r, int64_t start, int64_t stop, float gamma, float* ret) { /* :275-281 */
  float acc = 0.f; for (int64_t i = stop; i >= start; --i) { acc = r[i] + gamma * acc; ret[i] = acc; }
}

/* source= / target= keywords (sampler.jl:255): fill_gae!(data, ep, Vc, lambda, gamma, source=:cost, target=:cost_advantage) (:65) */
int32_t orc_fill_gae_keys(orc_buffer* b, orc_mlp* critic, float lambda, float gamma, int32_t source, int32_t target) { /* :255-260 over episodes(d) */
  if (!(b->mask & (1u << target)) || (source != CRUX_COL_R && !(b->mask & (1u << source)))) return CRUX_EINVAL;
  int64_t n = b->elements; if (n == 0) return CRUX_OK;
  float* Vs = (float*)malloc(4 * (size_t)n); float* Vsp = (float*)malloc(4 * (size_t)n);
  int vo = critic->dims[critic->n_layers]; if (vo != 1) { free(Vs); free(Vsp); return CRUX_EINVAL; }   /* @assert length(Vs)==1 :268 */
  orc_mlp_forward(critic, (float*)b->col[CRUX_COL_S], n, Vs); orc_mlp_forward(critic, (float*)b->col[CRUX_COL_SP], n, Vsp);
  int64_t* st = (int64_t*)malloc(8 * (size_t)n); int64_t* en = (int64_t*)malloc(8 * (size_t)n);
  int64_t ne = orc_buffer_episodes(b, st, en, n);
  float* adv = (float*)b->col[target]; int32_t rc = CRUX_OK;
  for (int64_t k = 0; k < ne; ++k) orc_gae_range((float*)b->col[source], (uint8_t*)b->col[CRUX_COL_DONE], Vs, Vsp, st[k], en[k], lambda, gamma, adv);
  for (int64_t i = 0; i < n; ++i) if (isnan(adv[i])) rc = CRUX_ENAN;                              /* @assert !isnan(A) :270 */
  free(Vs); free(Vsp); free(st); free(en); return rc;
}
int32_t orc_fill_gae(orc_buffer* b, orc_mlp* critic, float lambda, float gamma) { return orc_fill_gae_keys(b, critic, lambda, gamma, CRUX_COL_R, CRUX_COL_ADVANTAGE); }
/* episodes! metrics (sampler.jl:175-251) for the first episode of each env of an env-major n_envs x T block */
int32_t orc_first_episode_metrics(orc_buffer* b, int32_t n_envs, int64_t T, float gamma, float* und, float* dis, int64_t* len, uint8_t* complete) {
  if ((int64_t)n_envs * T != b->elements) return CRUX_EINVAL;
  const float* r = (const float*)b->col[CRUX_COL_R]; const uint8_t* ee = (const uint8_t*)b->col[CRUX_COL_EPISODE_END];
  for (int e = 0; e < n_envs; ++e) { int64_t base = (int64_t)e * T, stop = -1;
    for (int64_t t = 0; t < T; ++t) if (ee[base + t]) { stop = t; break; }
    if (complete) complete[e] = stop >= 0; if (stop < 0) stop = T - 1;
    float u = 0.f, d = 0.f;
    for (int64_t t = 0; t <= stop; ++t) u = u + r[base + t];                       /* sum(data[:r][1,start:stop]) :203 */
    for (int64_t t = stop; t >= 0; --t) d = r[base + t] + gamma * d;                /* discounted_return :223-229 */
    if (und) und[e] = u; if (dis) dis[e] = d; if (len) len[e] = stop + 1; }
  return CRUX_OK;
}
int32_t orc_fill_returns_keys(orc_buffer* b, float gamma, int32_t source, int32_t target) {   /* fill_returns!(...; source=:cost, target=:cost_return) (:66,275) */
  if (!(b->mask & (1u << target)) || (source != CRUX_COL_R && !(b->mask & (1u << source)))) return CRUX_EINVAL;
  int64_t n = b->elements; if (n == 0) return CRUX_OK;
  int64_t* st = (int64_t*)malloc(8 * (size_t)n); int64_t* en = (int64_t*)malloc(8 * (size_t)n);
  int64_t ne = orc_buffer_episodes(b, st, en, n);
  for (int64_t k = 0; k < ne; ++k) orc_returns_range((float*)b->col[source], st[k], en[k], gamma, (float*)b->col[target]);
  free(st); free(en); return CRUX_OK;
}
int32_t orc_fill_returns(orc_buffer* b, float gamma) { return orc_fill_returns_keys(b, gamma, CRUX_COL_R, CRUX_COL_RETURN); }
/* :importance_weight of every row: exp.(logpdf(pa, s, a) .- logprob) with the nominal action policy pa (step!, sampler.jl:108-111);
 * categorical_logpdf policies.jl:128-135, gaussian_logpdf :333-336 */
int32_t orc_importance_weight(orc_buffer* b, orc_mlp* nominal, int32_t head) {
  if (!(b->mask & (1u << CRUX_COL_IMPORTANCE_WEIGHT)) || !(b->mask & (1u << CRUX_COL_LOGPROB))) return CRUX_EINVAL;
  const int od = b->obs_dim, ad = b->act_dim, nout = nominal->dims[nominal->n_layers]; if (nout != ad || nominal->dims[0] != od) return CRUX_EINVAL;
  colcache c = cc_alloc(nominal); float p[64]; if (nout > 64) { cc_free(nominal, &c); return CRUX_EINVAL; }
  const float* S = (const float*)b->col[CRUX_COL_S]; const float* LP = (const float*)b->col[CRUX_COL_LOGPROB]; float* IW = (float*)b->col[CRUX_COL_IMPORTANCE_WEIGHT];
  for (int64_t j = 0; j < b->elements; ++j) {
    fwd_col(nominal, S + (size_t)j * od, c.h); const float* z = c.h[nominal->n_layers]; float nom;
    if (head == CRUX_HEAD_CATEGORICAL) { const uint8_t* a = (const uint8_t*)b->col[CRUX_COL_A] + (size_t)j * ad;
      softmax_col(z, nout, p); float q = 0.f; for (int k = 0; k < nout; ++k) q = q + p[k] * (a[k] ? 1.f : 0.f); nom = logf(q); }
    else { const float* a = (const float*)b->col[CRUX_COL_A] + (size_t)j * ad; const float* ls = nominal->p + xoff(nominal); nom = 0.f;
      for (int d = 0; d < ad; ++d) { float sg = expf(ls[d]), s2 = sg * sg, df = a[d] - z[d]; nom = nom + ((-(df * df) / (2.f * s2) - 0.9189385332046727f) - ls[d]); } }
    IW[j] = expf(nom - LP[j]);
  }
  cc_free(nominal, &c); return CRUX_OK;
}
/* fill_fwd_importance_weight! / fill_cum_importance_weight! / fill_rev_importance_weight! (sampler.jl:283-308) over episodes(b), as terminate_episode! calls them (:58-60) */
int32_t orc_fill_importance_weights(orc_buffer* b) {
  if (!(b->mask & (1u << CRUX_COL_IMPORTANCE_WEIGHT))) return CRUX_EINVAL;                       /* @assert haskey(data, :importance_weight) */
  int64_t n = b->elements; if (n == 0) return CRUX_OK;
  const float* iw = (const float*)b->col[CRUX_COL_IMPORTANCE_WEIGHT];
  float* fwd = (b->mask & (1u << CRUX_COL_FWD_IMPORTANCE_WEIGHT)) ? (float*)b->col[CRUX_COL_FWD_IMPORTANCE_WEIGHT] : NULL;
  float* cum = (b->mask & (1u << CRUX_COL_CUM_IMPORTANCE_WEIGHT)) ? (float*)b->col[CRUX_COL_CUM_IMPORTANCE_WEIGHT] : NULL;
  float* rev = (b->mask & (1u << CRUX_COL_REV_IMPORTANCE_WEIGHT)) ? (float*)b->col[CRUX_COL_REV_IMPORTANCE_WEIGHT] : NULL;
  int64_t* st = (int64_t*)malloc(8 * (size_t)n); int64_t* en = (int64_t*)malloc(8 * (size_t)n);
  int64_t ne = orc_buffer_episodes(b, st, en, n);
  for (int64_t k = 0; k < ne; ++k) { float w;
    if (fwd) { w = 1.f; for (int64_t i = st[k]; i <= en[k]; ++i) { w = iw[i] * w; fwd[i] = w; } }                    /* :283-290 */
    if (cum) { w = 1.f; for (int64_t i = st[k]; i <= en[k]; ++i) w = iw[i] * w; for (int64_t i = st[k]; i <= en[k]; ++i) cum[i] = w; }   /* :292-299 */
    if (rev) { w = 1.f; for (int64_t i = en[k]; i >= st[k]; --i) { w = iw[i] * w; rev[i] = w; } } }                  /* :301-308 */
  free(st); free(en); return CRUX_OK;
}
/* [3P] Julia's reductions over a Float32 array, restated from Base (reduce.jl: mapreduce_impl, pairwise_blocksize = 1024) and Statistics (mean = sum(A) / length(A);
 * var(A; corrected = true) = centralize_sumabs2(A, mean(A)) / (n - 1), the same pairwise scheme over abs2(x - m); std = sqrt(var)): a range of at most 1024 elements is
 * summed from the left, `v = a1 + a2; v += a3; ...`, a longer one is split at ifirst + (ilast - ifirst) >> 1 and the two halves are added -- all in Float32.
 * (Base marks the inner loop @simd, which lets LLVM re-associate it by vector lane on the machine Julia runs on; the left-to-right order is the scalar semantics.) */
static float jl_mapreduce_f32(const float* a, int64_t ifirst, int64_t ilast, int centred, float m) {      /* 0-based, inclusive */
#define JL_F(x) (centred ? ((x) - m) * ((x) - m) : (x))
  if (ifirst == ilast) return JL_F(a[ifirst]);
  if (ilast - ifirst < 1024) { float v = JL_F(a[ifirst]) + JL_F(a[ifirst + 1]); for (int64_t i = ifirst + 2; i <= ilast; ++i) v = v + JL_F(a[i]); return v; }
  int64_t imid = ifirst + ((ilast - ifirst) >> 1);
  float v1 = jl_mapreduce_f32(a, ifirst, imid, centred, m), v2 = jl_mapreduce_f32(a, imid + 1, ilast, centred, m);
  return v1 + v2;
#undef JL_F
}
float orc_jl_sum_f32(const float* a, int64_t n) { return n <= 0 ? 0.f : jl_mapreduce_f32(a, 0, n - 1, 0, 0.f); }
float orc_jl_mean_f32(const float* a, int64_t n) { return orc_jl_sum_f32(a, n) / (float)n; }                       /* Statistics.mean: sum(A) / length(A), Float32 / Int */
float orc_jl_std_f32(const float* a, int64_t n) {                                                                    /* Statistics.std, corrected */
  float m = orc_jl_mean_f32(a, n); return sqrtf(jl_mapreduce_f32(a, 0, n - 1, 1, m) / (float)(n - 1)); }
/* whiten(v) = (v .- mean(v)) ./ std(v)  utils.jl:41-42, with Julia's own reductions (above) */
int32_t orc_whiten(orc_buffer* b, int32_t key) {
  if (!orc_buffer_has_column(b, key) || col_elem(b, key) != 4 || col_rows(b, key) != 1) return CRUX_EINVAL;
  float* v = (float*)b->col[key]; int64_t n = b->elements; if (n < 2) return CRUX_EINVAL;
  float mean = orc_jl_mean_f32(v, n), sd = orc_jl_std_f32(v, n);
  for (int64_t i = 0; i < n; ++i) v[i] = (v[i] - mean) / sd;
  return CRUX_OK;
}

/* ============================================================================================
 * learner                                src/training.jl:13-55, src/model_free/rl/ppo.jl:4-21,59-60
 * ============================================================================================ */
/* loss + flat gradient on rows ids[0..n) ; info per training.jl:22-23 / ppo.jl:13-19. */
/* lagrange_ppo_loss's penalty controller (rl/ppo.jl:80-116), run inside the loss on every evaluation; state = the one-element arrays of P (:192-201) */
static crux_lagrange* g_lag = NULL;
static float clamp_jl(float x, float lo, float hi) { return x > hi ? hi : (x < lo ? lo : x); }     /* Base.clamp: NaN passes through */
static float lagrange_penalty(crux_lagrange* L, const orc_buffer* buf, const int64_t* ids, int64_t n) {
  const float* COST = (const float*)buf->col[CRUX_COL_COST]; const uint8_t* EE = (const uint8_t*)buf->col[CRUX_COL_EPISODE_END];
  double sc = 0.0; int64_t ne = 0; for (int64_t s = 0; s < n; ++s) { sc += (double)COST[ids[s]]; ne += EE[ids[s]] ? 1 : 0; }
  float Jc = (float)sc / (float)ne;                                                              /* :86 sum(D[:cost]) / sum(D[:episode_end]); Float32 sum restated through Float64 */
  float dl = Jc - L->target_cost;                                                                 /* :91 */
  L->I = clamp_jl(L->I + L->Ki * dl, 0.f, L->Ki_max);                                             /* :94 */
  L->smooth_delta = (float)(L->ema_alpha * (double)L->smooth_delta + (1.0 - L->ema_alpha) * (double)dl);   /* :98, ema_alpha is Float64 */
  L->smooth_Jc = (float)(L->ema_alpha * (double)L->smooth_Jc + (1.0 - L->ema_alpha) * (double)Jc);         /* :99 */
  { float x = L->smooth_Jc - L->Jc_prev; L->deriv_term = (x != x) ? x : (x > 0.f ? x : 0.f); }    /* :102 max(0, x): NaN if x is NaN */
  L->Jc_prev = L->smooth_Jc;                                                                      /* :105 */
  L->penalty = clamp_jl((L->Kp * L->smooth_delta + L->I) + L->Kd * L->deriv_term, 0.f, L->penalty_max);   /* :108 */
  L->cur_cost = Jc;
  return L->penalty;
}

static int32_t loss_grad(orc_mlp* net, orc_buffer* buf, const crux_train_cfg* cfg, const int64_t* ids, int64_t n, float* info) {
  int od = buf->obs_dim, ad = buf->act_dim, nout = net->dims[net->n_layers];
  if (net->dims[0] != od || n <= 0) return CRUX_EINVAL;
  memset(net->g, 0, 4 * (size_t)net->n_params);
  for (int q = 0; q < CRUX_INFO_N; ++q) info[q] = 0.f;
  const float* S = (const float*)buf->col[CRUX_COL_S];
  float invB = 1.0f / (float)n;
  /* the per-sample terms of every mean the loss reports: reduced at the end the way Julia's mean does (orc_jl_mean_f32) */
  const int64_t nsq = n * (cfg->loss == CRUX_LOSS_MSE_ACTION ? (int64_t)nout : 1);
  float* T_ = (float*)calloc((size_t)(6 * n + nsq), 4);
  float* T_lp = T_; float* T_H = T_ + n; float* T_kl = T_ + 2 * n; float* T_adv = T_ + 3 * n; float* T_ret = T_ + 4 * n; float* T_cost = T_ + 5 * n; float* T_sq = T_ + 6 * n; int64_t nclip = 0;
  /* The sample loops below run once, in order, into net->g in the parity build. The OpenMP build gives every thread a slice of the minibatch and a
   * private gradient that is added to net->g at the end (sample-parallel inside one step: the steps themselves are serially dependent). */
#ifdef _OPENMP
  const int nthr = n >= 32 ? (omp_get_max_threads() < (int)(n / 8) ? omp_get_max_threads() : (int)(n / 8)) : 1;
#define LG_BEGIN ORC_OMP(omp parallel num_threads(nthr) reduction(+ : nclip)) \
  { colcache c = cc_alloc(net); float dy[64], p[64]; (void)p; float* gl = nthr > 1 ? (float*)calloc((size_t)net->n_params, 4) : net->g; ORC_OMP(omp for schedule(static))
#define LG_END if (gl != net->g) { ORC_OMP(omp critical) { for (int64_t i_ = 0; i_ < net->n_params; ++i_) net->g[i_] += gl[i_]; } free(gl); } cc_free(net, &c); }
#else
#define LG_BEGIN { colcache c = cc_alloc(net); float dy[64], p[64]; (void)p; float* gl = net->g;
#define LG_END cc_free(net, &c); }
#endif
  if (cfg->loss == CRUX_LOSS_MSE_ACTION) {                                     /* mse_action_loss il/bc.jl:1: Flux.mse(action(pi, s), a) */
    if (buf->act_kind != CRUX_ACTION_CONTINUOUS || nout != ad) { free(T_); return CRUX_EINVAL; }
    const float* A = (const float*)buf->col[CRUX_COL_A]; float inv = invB / (float)nout;
    LG_BEGIN
    for (int64_t s = 0; s < n; ++s) { int64_t id = ids[s]; fwd_col(net, S + (size_t)id * od, c.h);
      for (int k = 0; k < nout; ++k) { float d = c.h[net->n_layers][k] - A[(size_t)id * ad + k]; T_sq[s * nout + k] = d * d; dy[k] = 2.f * d * inv; }      /* Flux.mse: mean(abs2.(yhat .- y)) over all nout x n elements, column-major */
      bwd_col(net, c.h, dy, gl); }
    LG_END
    info[CRUX_INFO_LOSS] = orc_jl_mean_f32(T_sq, nsq);
  } else if (cfg->loss == CRUX_LOSS_VALUE_MSE) {                               /* Flux.mse(value(pi, s), return) ppo.jl:60 */
    const int tk = cfg->target_col > 0 ? cfg->target_col : CRUX_COL_RETURN;      /* D[:return] (ppo.jl:60) or D[:cost_return] (:210) */
    if (nout != 1 || !(buf->mask & (1u << tk))) { free(T_); return CRUX_EINVAL; }
    const float* RET = (const float*)buf->col[tk];
    LG_BEGIN
    for (int64_t s = 0; s < n; ++s) { int64_t id = ids[s];
      fwd_col(net, S + (size_t)id * od, c.h); float d = c.h[net->n_layers][0] - RET[id];
      T_sq[s] = d * d; dy[0] = 2.f * d * invB; bwd_col(net, c.h, dy, gl); }
    LG_END
    info[CRUX_INFO_LOSS] = orc_jl_mean_f32(T_sq, n);
  } else {                                                                     /* ppo_loss ppo.jl:4-21 */
    const float* RET = (buf->mask & (1u << CRUX_COL_RETURN)) ? (const float*)buf->col[CRUX_COL_RETURN] : NULL;
    const int bc = cfg->loss == CRUX_LOSS_LOGPDF_BC;                            /* logpdf_bc_loss il/bc.jl:10-18: only :s and :a are read */
    if (!bc && (!(buf->mask & (1u << CRUX_COL_LOGPROB)) || (cfg->loss == CRUX_LOSS_REINFORCE ? !RET : !(buf->mask & (1u << CRUX_COL_ADVANTAGE))))) { free(T_); return CRUX_EINVAL; }
    if (nout != ad || (cfg->head != CRUX_HEAD_CATEGORICAL && net->n_extra != ad)) { free(T_); return CRUX_EINVAL; }
    const float* LP = bc ? NULL : (const float*)buf->col[CRUX_COL_LOGPROB]; const float* ADV = bc ? NULL : (cfg->loss == CRUX_LOSS_REINFORCE ? RET : (const float*)buf->col[CRUX_COL_ADVANTAGE]);
    float lo = 1.f - cfg->eps_clip, hi = 1.f + cfg->eps_clip;
    const float* ls = net->p + xoff(net);
    const int lagr = cfg->loss == CRUX_LOSS_LAGRANGE_PPO; float pen = 0.f; const float* CADV = NULL;
    if (lagr) { if (!g_lag || !(buf->mask & (1u << CRUX_COL_COST)) || !(buf->mask & (1u << CRUX_COL_COST_ADVANTAGE))) { free(T_); return CRUX_EINVAL; }
      CADV = (const float*)buf->col[CRUX_COL_COST_ADVANTAGE]; pen = lagrange_penalty(g_lag, buf, ids, n); }
    LG_BEGIN
    for (int64_t s = 0; s < n; ++s) { int64_t id = ids[s]; float* gx = gl + xoff(net);
      fwd_col(net, S + (size_t)id * od, c.h); const float* z = c.h[net->n_layers];
      float A = bc ? 1.f : ADV[id], oldlp = bc ? 0.f : LP[id], newlp, H = 0.f;
      if (cfg->head == CRUX_HEAD_CATEGORICAL) {
        const uint8_t* a = (const uint8_t*)buf->col[CRUX_COL_A] + (size_t)id * ad;
        softmax_col(z, nout, p);
        float q = 0.f; for (int k = 0; k < nout; ++k) q = q + p[k] * (a[k] ? 1.f : 0.f);        /* sum(probs .* a_oh) policies.jl:135 */
        newlp = logf(q);
        float hk[64], hp = 0.f;
        for (int k = 0; k < nout; ++k) { float l = logf(p[k] + EPS32); H = H - p[k] * l;        /* entropy policies.jl:152-155 */
          hk[k] = -l - p[k] / (p[k] + EPS32); hp = hp + hk[k] * p[k]; }
        float r = expf(newlp - oldlp), u = r * A, rc = r < lo ? lo : r > hi ? hi : r, cl = rc * A;
        float g = (u <= cl) ? A : 0.f;                       /* d min(u,c)/dr: ties -> first arg; clipped branch strictly smaller => clamp' = 0 */
        float coef = g * r, lterm = (u <= cl ? u : cl), lp_ = cfg->lambda_p, le_ = cfg->lambda_e;
        if (cfg->loss == CRUX_LOSS_A2C) { coef = A; lterm = newlp * A; }                                   /* a2c.jl:6 */
        else if (bc) { coef = 1.f; lterm = newlp; lp_ = 1.f; }                                             /* -mean(logpdf) bc.jl:12 */
        else if (cfg->loss == CRUX_LOSS_REINFORCE) { coef = RET[id]; lterm = newlp * RET[id]; lp_ = 1.f; le_ = 0.f; }   /* reinforce.jl:12 */
        T_lp[s] = lterm;
        float gcr = 0.f;                                           /* lagrange: d/dr max(r Ac, clamp(r) Ac) times r (ppo.jl:119) */
        if (lagr) { float Ac = CADV[id], uc = r * Ac, clc = rc * Ac; T_cost[s] = (uc >= clc ? uc : clc); gcr = (uc >= clc ? Ac : 0.f) * r; }
        for (int k = 0; k < nout; ++k) {
          float dlogpi = p[k] * ((a[k] ? 1.f : 0.f) / q) - p[k];   /* = y_k - p_k for one-hot y */
          float dH = p[k] * (hk[k] - hp);
          float base = -lp_ * coef * dlogpi - le_ * dH;
          dy[k] = lagr ? invB * ((base + pen * gcr * dlogpi) / (1.f + pen)) : invB * base;    /* (...) / (1 + penalty) (:131) */
        }
        if ((cfg->loss == CRUX_LOSS_PPO || lagr) && (r > hi || r < lo)) ++nclip;
      } else {                                                                   /* GaussianPolicy policies.jl:333-348 */
        const float* a = (const float*)buf->col[CRUX_COL_A] + (size_t)id * ad;
        newlp = 0.f; const float sq = net->squash; float ua[64];
        for (int k = 0; k < ad; ++k) { ua[k] = sq > 0.f ? sq_untanh(a[k], sq) : a[k];
          float sg = expf(sq > 0.f ? sq_clampls(ls[k]) : ls[k]); float s2 = sg * sg; float d = ua[k] - z[k];
          newlp = newlp + (-(d * d) / (2.f * s2) - 0.9189385332046727f - ls[k]);
          if (sq > 0.f) newlp = newlp - sq_corr(ua[k]); }
        float r = expf(newlp - oldlp), u = r * A, rc = r < lo ? lo : r > hi ? hi : r, cl = rc * A;
        float g = (u <= cl) ? A : 0.f;
        float coef = g * r, lterm = (u <= cl ? u : cl), lp_ = cfg->lambda_p;
        if (cfg->loss == CRUX_LOSS_A2C) { coef = A; lterm = newlp * A; }
        else if (bc) { coef = 1.f; lterm = newlp; lp_ = 1.f; }
        else if (cfg->loss == CRUX_LOSS_REINFORCE) { coef = RET[id]; lterm = newlp * RET[id]; lp_ = 1.f; }
        T_lp[s] = lterm;
        float cf = -lp_ * coef;                                                                  /* coefficient of d logpdf in d loss */
        if (lagr) { float Ac = CADV[id], uc = r * Ac, clc = rc * Ac; T_cost[s] = (uc >= clc ? uc : clc);
          cf = (cf + pen * ((uc >= clc ? Ac : 0.f) * r)) / (1.f + pen); }
        for (int k = 0; k < ad; ++k) { float sg = expf(sq > 0.f ? sq_clampls(ls[k]) : ls[k]); float s2 = sg * sg; float d = ua[k] - z[k];
          float inr = (sq > 0.f && !(ls[k] >= -5.f && ls[k] <= 2.f)) ? 0.f : 1.f;               /* d clamp(x, lo, hi)/dx = 1 inside [lo, hi], 0 outside (ChainRules) */
          dy[k] = invB * (cf * (d / s2));
          gx[k] += invB * (cf * (((d * d) / s2) * inr - 1.f)); }
        if ((cfg->loss == CRUX_LOSS_PPO || lagr) && (r > hi || r < lo)) ++nclip;
      }
      T_H[s] = H; T_kl[s] = oldlp - newlp; T_adv[s] = A; if (RET) T_ret[s] = RET[id];
      bwd_col(net, c.h, dy, gl);
    }
    LG_END
    float* gx = net->g + xoff(net);
    float p_loss = -orc_jl_mean_f32(T_lp, n), e_loss, entropy;
    if (cfg->head == CRUX_HEAD_CATEGORICAL) { entropy = orc_jl_mean_f32(T_H, n); e_loss = -entropy; }
    else { float Hs = 1.4189385332046727f; for (int k = 0; k < ad; ++k) Hs = Hs + ls[k]; entropy = Hs; e_loss = -Hs;   /* scalar entropy policies.jl:348 */
      if (cfg->loss != CRUX_LOSS_REINFORCE) for (int k = 0; k < ad; ++k) gx[k] += lagr ? -cfg->lambda_e / (1.f + pen) : -cfg->lambda_e; }
    info[CRUX_INFO_LOSS] = cfg->loss == CRUX_LOSS_REINFORCE ? p_loss : (bc ? 1.f : cfg->lambda_p) * p_loss + cfg->lambda_e * e_loss;   /* ppo.jl:20, a2c.jl:14, reinforce.jl:12 */
    info[CRUX_INFO_ENTROPY] = entropy; info[CRUX_INFO_KL] = orc_jl_mean_f32(T_kl, n);
    info[CRUX_INFO_CLIP_FRACTION] = (float)nclip / (float)n; info[CRUX_INFO_AVG_ADVANTAGE] = orc_jl_mean_f32(T_adv, n);
    info[CRUX_INFO_AVG_RETURN] = orc_jl_mean_f32(T_ret, n);
    if (lagr) { float cost_loss = pen * orc_jl_mean_f32(T_cost, n);                                              /* ppo.jl:119 */
      info[CRUX_INFO_LOSS] = ((cfg->lambda_p * p_loss + cfg->lambda_e * e_loss) + cost_loss) / (1.f + pen);          /* :131 */
      info[CRUX_INFO_PENALTY] = pen; info[CRUX_INFO_CUR_COST] = g_lag->cur_cost; info[CRUX_INFO_COST_LOSS] = cost_loss; info[CRUX_INFO_P_LOSS] = cfg->lambda_p * p_loss; }
  }
  /* norm(grad) utils.jl:49-55: 2-norm of the per-tensor 2-norms */
  double tot = 0;
  for (int l = 0; l < net->n_layers; ++l) {
    int64_t w0 = woff(net, l), b0 = boff(net, l), w1 = b0, b1 = b0 + net->dims[l + 1];
    double sw = 0, sb = 0; for (int64_t i = w0; i < w1; ++i) sw += (double)net->g[i] * net->g[i]; for (int64_t i = b0; i < b1; ++i) sb += (double)net->g[i] * net->g[i];
    float nw = (float)sqrt(sw), nb = (float)sqrt(sb); tot += (double)nw * nw + (double)nb * nb;
  }
  if (net->n_extra) { double sx = 0; for (int i = 0; i < net->n_extra; ++i) sx += (double)net->g[xoff(net) + i] * net->g[xoff(net) + i]; float nx = (float)sqrt(sx); tot += (double)nx * nx; }
  info[CRUX_INFO_GRAD_NORM] = (float)sqrt(tot);
  free(T_);
  return CRUX_OK;
#undef LG_BEGIN
#undef LG_END
}

int32_t orc_loss_grad(orc_mlp* net, orc_buffer* buf, const crux_train_cfg* cfg, const int64_t* ids, int64_t n, float* info) { return loss_grad(net, buf, cfg, ids, n, info); }

/* train!(pi, loss, p) training.jl:13-25 */
int32_t orc_train_step(orc_mlp* net, orc_buffer* buf, const crux_train_cfg* cfg, const int64_t* ids, int64_t n, float* info) {
  int32_t rc = loss_grad(net, buf, cfg, ids, n, info); if (rc) return rc;
  if (isnan(info[CRUX_INFO_GRAD_NORM])) return CRUX_ENAN;                      /* :20 */
  return orc_adam_apply(net, 1.0f);                                            /* :21 */
}

/* batch_train!(pi, p, P, D) training.jl:28-55 incl. the aliased-info early-stopping semantics
 * (SURVEY App. A-Q3): aggregate_info(minibatch_infos) == the latest minibatch's info. */
int32_t orc_batch_train(orc_mlp* net, orc_buffer* buf, const crux_train_cfg* cfg, const int64_t* perms, float* info_out, float* epoch_infos) {
  int64_t len = buf->elements, bs = cfg->batch_size; if (len <= 0 || bs <= 0) return CRUX_EINVAL;
  int64_t total = 0; int epochs_run = 0; int stop = 0;
  double agg[CRUX_INFO_N] = {0};
  int64_t* perm = (int64_t*)malloc(8 * (size_t)len); int64_t* ids = (int64_t*)malloc(8 * (size_t)bs);
  float info[CRUX_INFO_N]; memset(info, 0, sizeof info);
  for (int ep = 0; ep < cfg->epochs && !stop; ++ep) {
    if (perms) memcpy(perm, perms + (size_t)ep * len, 8 * (size_t)len);
    else { crux_perm pp = crux_perm_make(cfg->shuffle_seed, cfg->shuffle_counter + (uint64_t)ep, 0, (uint32_t)len); for (int64_t j = 0; j < len; ++j) perm[j] = crux_perm_at(&pp, (uint32_t)j); }
    int32_t rc = orc_buffer_permute(buf, perm); if (rc) { free(perm); free(ids); return rc; }            /* :36-38 */
    int brk = 0;
    for (int64_t st = 0; st < len; st += bs) {                                                          /* partition(1:len, bs) :40 */
      int64_t nb = len - st < bs ? len - st : bs; for (int64_t j = 0; j < nb; ++j) ids[j] = st + j;
      rc = orc_train_step(net, buf, cfg, ids, nb, info); if (rc) { free(perm); free(ids); return rc; }   /* :43 */
      total += 1;
      if (cfg->max_batches > 0 && total >= cfg->max_batches) { brk = 1; break; }                          /* :45 */
      if (cfg->target_kl >= 0.f && cfg->loss != CRUX_LOSS_VALUE_MSE && info[CRUX_INFO_KL] > cfg->target_kl) { brk = 1; break; }   /* :46 */
    }
    (void)brk;
    for (int q = 0; q < CRUX_INFO_N; ++q) { agg[q] += (double)info[q]; if (epoch_infos) epoch_infos[(size_t)ep * CRUX_INFO_N + q] = info[q]; }   /* :48 */
    epochs_run += 1;
    if (cfg->target_kl >= 0.f && cfg->loss != CRUX_LOSS_VALUE_MSE && info[CRUX_INFO_KL] > cfg->target_kl) stop = 1;               /* :49 */
    if (cfg->max_batches > 0 && total >= cfg->max_batches) stop = 1;                                      /* :50 */
  }
  for (int q = 0; q < CRUX_INFO_N; ++q) info_out[q] = epochs_run ? (float)(agg[q] / (double)epochs_run) : 0.f;   /* merge!(info, aggregate_info(infos)) :54 */
  info_out[CRUX_INFO_BATCHES_TRAINED] = (float)total; info_out[CRUX_INFO_EPOCHS_RUN] = (float)epochs_run;         /* :53 */
  free(perm); free(ids);
  return CRUX_OK;
}

/* batch_train!(actor, a_opt, P, D) with lagrange_ppo_loss (rl/ppo.jl:70-131,208): the PID state in *lag advances once per executed minibatch */
int32_t orc_batch_train_lagrange(orc_mlp* net, orc_buffer* buf, const crux_train_cfg* cfg, crux_lagrange* lag, const int64_t* perms, float* info_out, float* epoch_infos) {
  if (!lag || cfg->loss != CRUX_LOSS_LAGRANGE_PPO) return CRUX_EINVAL;
  g_lag = lag; int32_t rc = orc_batch_train(net, buf, cfg, perms, info_out, epoch_infos); g_lag = NULL; return rc;
}

/* ============================================================================================
 * off-policy pieces                        src/model_free/rl/dqn.jl:4-6, src/utils.jl:76-87,112
 * ============================================================================================ */
int32_t orc_dqn_target(orc_mlp* tn, orc_buffer* b, float gamma, float* y) {
  int64_t n = b->elements; int nout = tn->dims[tn->n_layers]; colcache c = cc_alloc(tn);
  const float* SP = (const float*)b->col[CRUX_COL_SP]; const float* R = (const float*)b->col[CRUX_COL_R]; const uint8_t* D = (const uint8_t*)b->col[CRUX_COL_DONE];
  for (int64_t s = 0; s < n; ++s) { fwd_col(tn, SP + (size_t)s * b->obs_dim, c.h); const float* q = c.h[tn->n_layers];
    float mx = q[0]; for (int k = 1; k < nout; ++k) if (q[k] > mx) mx = q[k];
    y[s] = R[s] + (gamma * (1.f - (D[s] ? 1.f : 0.f))) * mx; }                 /* r .+ gamma .* (1 .- done) .* max  dqn.jl:5 */
  cc_free(tn, &c); return CRUX_OK;
}
/* softq_target(alpha) softq.jl:4-13: r + gamma (1-done) alpha logsumexp(Q^-(sp) ./ alpha) [3P NNlib.logsumexp: max-shifted] */
int32_t orc_softq_target(orc_mlp* tn, orc_buffer* b, float gamma, float alpha, float* y) {
  int64_t n = b->elements; int nout = tn->dims[tn->n_layers]; colcache c = cc_alloc(tn);
  const float* SP = (const float*)b->col[CRUX_COL_SP]; const float* R = (const float*)b->col[CRUX_COL_R]; const uint8_t* D = (const uint8_t*)b->col[CRUX_COL_DONE];
  for (int64_t s = 0; s < n; ++s) { fwd_col(tn, SP + (size_t)s * b->obs_dim, c.h); const float* q = c.h[tn->n_layers];
    float mx = q[0] / alpha; for (int k = 1; k < nout; ++k) { float v = q[k] / alpha; if (v > mx) mx = v; }
    float sum = 0.f; for (int k = 0; k < nout; ++k) sum = sum + expf(q[k] / alpha - mx);
    float sv = alpha * (mx + logf(sum));
    y[s] = R[s] + (gamma * (1.f - (D[s] ? 1.f : 0.f))) * sv; }
  cc_free(tn, &c); return CRUX_OK;
}
static float q_sa(const orc_buffer* b, const float* q, int64_t s) { /* value(pi, s, a_oh) = sum(value .* a_oh) policies.jl:122 */
  const uint8_t* a = (const uint8_t*)b->col[CRUX_COL_A] + (size_t)s * b->act_dim; float acc = 0.f;
  for (int k = 0; k < b->act_dim; ++k) acc = acc + q[k] * (a[k] ? 1.f : 0.f); return acc;
}
int32_t orc_td_error(orc_mlp* net, orc_buffer* b, const float* y, float* err) { /* utils.jl:112 */
  int64_t n = b->elements; colcache c = cc_alloc(net); const float* S = (const float*)b->col[CRUX_COL_S];
  for (int64_t s = 0; s < n; ++s) { fwd_col(net, S + (size_t)s * b->obs_dim, c.h); err[s] = fabsf(q_sa(b, c.h[net->n_layers], s) - y[s]); }
  cc_free(net, &c); return CRUX_OK;
}
int32_t orc_td_step(orc_mlp* net, orc_buffer* b, const float* y, int32_t use_weight, float* info) { /* td_loss utils.jl:76-87 + train! */
  int64_t n = b->elements; int nout = net->dims[net->n_layers]; if (n <= 0 || nout != b->act_dim) return CRUX_EINVAL;
  colcache c = cc_alloc(net); const float* S = (const float*)b->col[CRUX_COL_S]; const float* W = use_weight ? (const float*)b->col[CRUX_COL_WEIGHT] : NULL;
  memset(net->g, 0, 4 * (size_t)net->n_params); for (int q = 0; q < CRUX_INFO_N; ++q) info[q] = 0.f;
  double sl = 0, sq = 0; float dy[64], invB = 1.f / (float)n;
  for (int64_t s = 0; s < n; ++s) { fwd_col(net, S + (size_t)s * b->obs_dim, c.h); float Q = q_sa(b, c.h[net->n_layers], s); float d = Q - y[s]; float w = W ? W[s] : 1.f;
    sl += (double)(d * d * w); sq += (double)Q; const uint8_t* a = (const uint8_t*)b->col[CRUX_COL_A] + (size_t)s * b->act_dim;
    for (int k = 0; k < nout; ++k) dy[k] = a[k] ? 2.f * d * w * invB : 0.f; bwd_col(net, c.h, dy, net->g); }
  info[CRUX_INFO_LOSS] = (float)(sl / (double)n); info[2] = (float)(sq / (double)n);
  double tot = 0; for (int l = 0; l < net->n_layers; ++l) { int64_t w0 = woff(net, l), b0 = boff(net, l), b1 = b0 + net->dims[l + 1]; double sw = 0, sb = 0;
    for (int64_t i = w0; i < b0; ++i) sw += (double)net->g[i] * net->g[i]; for (int64_t i = b0; i < b1; ++i) sb += (double)net->g[i] * net->g[i]; tot += sw + sb; }
  info[CRUX_INFO_GRAD_NORM] = (float)sqrt(tot);
  cc_free(net, &c);
  if (isnan(info[CRUX_INFO_GRAD_NORM])) return CRUX_ENAN;
  return orc_adam_apply(net, 1.0f);
}

/* ============================================================================================
 * SAC                     src/model_free/rl/sac.jl:4-9,34-52; double_Q_loss src/utils.jl:89-96;
 *                         GaussianPolicy exploration / gaussian_logpdf src/policies.jl:333-344
 * Noise of batch column j, action dim d: randn(Philox(seed, counter, stream = j*ad + d, NOISE)).
 * ============================================================================================ */
/* reverse pass for one column: parameter gradients into g (if non-NULL) and d(loss)/d(input) into dx (if non-NULL). */
static void bwd_col_dx(const orc_mlp* n, float** h, const float* dy, float* g, float* dx) {
  float d0[1024], d1[1024];
  float *d = d0, *dn = d1;
  memcpy(d, dy, 4 * (size_t)n->dims[n->n_layers]);
  for (int l = n->n_layers - 1; l >= 0; --l) {
    int in = n->dims[l], out = n->dims[l + 1];
    const float* W = n->p + woff(n, l);
    for (int o = 0; o < out; ++o) {
      float y = h[l + 1][o];
      if (n->acts[l] == CRUX_ACT_RELU) d[o] = y > 0.f ? d[o] : 0.f;
      else if (n->acts[l] == CRUX_ACT_TANH) d[o] = d[o] * (1.f - y * y);
    }
    if (g) { float* gW = g + woff(n, l); float* gb = g + boff(n, l);
      for (int o = 0; o < out; ++o) gb[o] += d[o];
      for (int k = 0; k < in; ++k) for (int o = 0; o < out; ++o) gW[o + (int64_t)out * k] += d[o] * h[l][k]; }
    if (l > 0 || dx) {
      float* dst = l > 0 ? dn : dx;
      for (int k = 0; k < in; ++k) { float acc = 0.f; for (int o = 0; o < out; ++o) acc = acc + W[o + (int64_t)out * k] * d[o]; dst[k] = acc; }
      float* t = d; d = dn; dn = t;
    }
  }
}
/* exploration(pi::GaussianPolicy, s) policies.jl:338-344 for one column; h = cache of the mean network (h[L] = mu). */
static float gauss_explore_col(const orc_mlp* actor, float** h, uint64_t seed, uint64_t counter, int64_t j, float* a, float* eps) {
  int ad = actor->dims[actor->n_layers]; const float* ls = actor->p + xoff(actor); const float* mu = h[actor->n_layers];
  float lp = 0.f;
  for (int d = 0; d < ad; ++d) {
    float sg = expf(ls[d]);                                                   /* sigma = exp.(logSigma) :340 */
    eps[d] = randn_f32(seed, counter, (uint32_t)(j * ad + d), 0);             /* :341 */
    a[d] = eps[d] * sg + mu[d];                                               /* :342 */
    float s2 = sg * sg, df = a[d] - mu[d];                                    /* gaussian_logpdf :333-336 */
    lp = lp + ((-(df * df) / (2.f * s2) - 0.9189385332046727f) - ls[d]);
  }
  return lp;
}
static double sumsq_tensors(const orc_mlp* n) {   /* norm(grads): per-tensor 2-norms, then the 2-norm of those (utils.jl:49-55) */
  double tot = 0;
  for (int l = 0; l < n->n_layers; ++l) { int64_t w0 = woff(n, l), b0 = boff(n, l), b1 = b0 + n->dims[l + 1]; double sw = 0, sb = 0;
    for (int64_t i = w0; i < b0; ++i) sw += (double)n->g[i] * n->g[i]; for (int64_t i = b0; i < b1; ++i) sb += (double)n->g[i] * n->g[i]; tot += sw + sb; }
  double sx = 0; for (int64_t i = xoff(n); i < n->n_params; ++i) sx += (double)n->g[i] * n->g[i];
  return tot + sx;
}
static int sac_shapes_ok(const orc_mlp* actor, const orc_mlp* q1, const orc_mlp* q2, const orc_buffer* b) {
  int ad = actor->dims[actor->n_layers];
  return b->act_kind == CRUX_ACTION_CONTINUOUS && actor->dims[0] == b->obs_dim && ad == b->act_dim && actor->n_extra == ad &&
         q1->dims[0] == b->obs_dim + ad && q2->dims[0] == b->obs_dim + ad && q1->dims[q1->n_layers] == 1 && q2->dims[q2->n_layers] == 1;
}

/* sac_target sac.jl:4-9: y = r + gamma*(1-done)*(min(Q1^-, Q2^-)(sp, a') - exp(log_alpha)*logprob), a' ~ actor(pi)(sp). */
int32_t orc_sac_target(orc_mlp* actor, orc_mlp* q1t, orc_mlp* q2t, orc_mlp* log_alpha, orc_buffer* b, float gamma, uint64_t seed, uint64_t counter, float* y) {
  if (!sac_shapes_ok(actor, q1t, q2t, b) || log_alpha->n_params < 1) return CRUX_EINVAL;
  int64_t n = b->elements; int od = b->obs_dim, ad = b->act_dim;
  colcache ca = cc_alloc(actor), c1 = cc_alloc(q1t), c2 = cc_alloc(q2t);
  const float* SP = (const float*)b->col[CRUX_COL_SP]; const float* R = (const float*)b->col[CRUX_COL_R]; const uint8_t* D = (const uint8_t*)b->col[CRUX_COL_DONE];
  float a[64], eps[64], sa[1024]; float alpha = expf(log_alpha->p[0]);
  for (int64_t j = 0; j < n; ++j) {
    fwd_col(actor, SP + (size_t)j * od, ca.h); float lp = gauss_explore_col(actor, ca.h, seed, counter, j, a, eps);
    memcpy(sa, SP + (size_t)j * od, 4 * (size_t)od); memcpy(sa + od, a, 4 * (size_t)ad);                 /* value(pi, s, a) = net(vcat(s, a)) policies.jl:96 */
    fwd_col(q1t, sa, c1.h); fwd_col(q2t, sa, c2.h);
    float qa = c1.h[q1t->n_layers][0], qb = c2.h[q2t->n_layers][0]; float mn = qb < qa ? qb : qa;
    y[j] = R[j] + (gamma * (1.f - (D[j] ? 1.f : 0.f))) * (mn - alpha * lp);
  }
  cc_free(actor, &ca); cc_free(q1t, &c1); cc_free(q2t, &c2); return CRUX_OK;
}

/* sac_temp_loss sac.jl:45-52 + train! on Flux.params(SAC_log_alpha): loss = -mean(exp(log_alpha) .* (logprob .+ H_target)). */
int32_t orc_sac_temp_step(orc_mlp* actor, orc_mlp* log_alpha, orc_buffer* b, float H_target, uint64_t seed, uint64_t counter, float* info) {
  if (actor->dims[0] != b->obs_dim || actor->n_extra != actor->dims[actor->n_layers] || log_alpha->n_params < 1) return CRUX_EINVAL;
  int64_t n = b->elements; if (n <= 0) return CRUX_EINVAL; int od = b->obs_dim;
  colcache ca = cc_alloc(actor); const float* S = (const float*)b->col[CRUX_COL_S];
  float a[64], eps[64]; float alpha = expf(log_alpha->p[0]); double st = 0;
  for (int q = 0; q < CRUX_INFO_N; ++q) info[q] = 0.f;
  for (int64_t j = 0; j < n; ++j) { fwd_col(actor, S + (size_t)j * od, ca.h); float lp = gauss_explore_col(actor, ca.h, seed, counter, j, a, eps); st += (double)(alpha * (lp + H_target)); }
  cc_free(actor, &ca);
  float mean_at = (float)(st / (double)n);
  memset(log_alpha->g, 0, 4 * (size_t)log_alpha->n_params);
  log_alpha->g[0] = -mean_at;                              /* d/dlog_alpha of -mean(exp(log_alpha) .* t) = -mean(exp(log_alpha) .* t) */
  info[CRUX_INFO_LOSS] = -mean_at; info[CRUX_INFO_GRAD_NORM] = fabsf(log_alpha->g[0]); info[CRUX_INFO_ALPHA] = alpha;
  if (isnan(info[CRUX_INFO_GRAD_NORM])) return CRUX_ENAN;
  return orc_adam_apply(log_alpha, 1.0f);
}

/* double_Q_loss utils.jl:89-96 + train!(critic(pi)) : 0.5*(mse(Q1(s,a), y) + mse(Q2(s,a), y)) [weighted_mean(:weight)], one gradient norm
 * over the parameters of both networks, one Adam step each (one optimiser object, per-array state). */
int32_t orc_double_q_step(orc_mlp* q1, orc_mlp* q2, orc_buffer* b, const float* y, int32_t use_weight, float* info) {
  int64_t n = b->elements; int od = b->obs_dim, ad = b->act_dim; if (n <= 0 || b->act_kind != CRUX_ACTION_CONTINUOUS) return CRUX_EINVAL;
  orc_mlp* qs[2] = {q1, q2}; for (int t = 0; t < 2; ++t) if (qs[t]->dims[0] != od + ad || qs[t]->dims[qs[t]->n_layers] != 1) return CRUX_EINVAL;
  const float* S = (const float*)b->col[CRUX_COL_S]; const float* A = (const float*)b->col[CRUX_COL_A]; const float* W = use_weight ? (const float*)b->col[CRUX_COL_WEIGHT] : NULL;
  for (int q = 0; q < CRUX_INFO_N; ++q) info[q] = 0.f;
  float sa[1024], invB = 1.f / (float)n; double loss = 0, tot = 0;
  for (int t = 0; t < 2; ++t) { orc_mlp* net = qs[t]; colcache c = cc_alloc(net); memset(net->g, 0, 4 * (size_t)net->n_params); double sl = 0, sq = 0;
    for (int64_t j = 0; j < n; ++j) { memcpy(sa, S + (size_t)j * od, 4 * (size_t)od); memcpy(sa + od, A + (size_t)j * ad, 4 * (size_t)ad);
      fwd_col(net, sa, c.h); float Q = c.h[net->n_layers][0], d = Q - y[j], w = W ? W[j] : 1.f; sl += (double)(d * d * w); sq += (double)Q;
      float dy = 0.5f * (2.f * d * w * invB); bwd_col_dx(net, c.h, &dy, net->g, NULL); }
    loss += 0.5 * (sl / (double)n); info[t == 0 ? CRUX_INFO_Q1AVG : CRUX_INFO_Q2AVG] = (float)(sq / (double)n); tot += sumsq_tensors(net); cc_free(net, &c); }
  info[CRUX_INFO_LOSS] = (float)loss; info[CRUX_INFO_GRAD_NORM] = (float)sqrt(tot);
  if (isnan(info[CRUX_INFO_GRAD_NORM])) return CRUX_ENAN;
  int32_t rc = orc_adam_apply(q1, 1.0f); if (rc) return rc; return orc_adam_apply(q2, 1.0f);
}

/* sac_actor_loss sac.jl:34-40 + train!(actor(pi)): mean(exp(log_alpha).*logprob .- min.(Q1(s,a), Q2(s,a))), a, logprob = exploration(pi.A, s);
 * reverse mode as Zygote accumulates it: abar = d/da(logprob term) + dQmin/da;  mubar = d/dmu(logprob term) + abar;
 * logSigmabar = d/dlogSigma(logprob term) + abar * eps * sigma. */
int32_t orc_sac_actor_step(orc_mlp* actor, orc_mlp* q1, orc_mlp* q2, orc_mlp* log_alpha, orc_buffer* b, uint64_t seed, uint64_t counter, float* info) {
  if (!sac_shapes_ok(actor, q1, q2, b) || log_alpha->n_params < 1) return CRUX_EINVAL;
  int64_t n = b->elements; if (n <= 0) return CRUX_EINVAL; int od = b->obs_dim, ad = b->act_dim;
  colcache ca = cc_alloc(actor), c1 = cc_alloc(q1), c2 = cc_alloc(q2);
  const float* S = (const float*)b->col[CRUX_COL_S]; const float* ls = actor->p + xoff(actor);
  float a[64], eps[64], sa[1024], dsa[1024], dmu[64]; float alpha = expf(log_alpha->p[0]), invB = 1.f / (float)n; double sl = 0, slp = 0;
  for (int q = 0; q < CRUX_INFO_N; ++q) info[q] = 0.f;
  memset(actor->g, 0, 4 * (size_t)actor->n_params);
  for (int64_t j = 0; j < n; ++j) {
    fwd_col(actor, S + (size_t)j * od, ca.h); float lp = gauss_explore_col(actor, ca.h, seed, counter, j, a, eps);
    memcpy(sa, S + (size_t)j * od, 4 * (size_t)od); memcpy(sa + od, a, 4 * (size_t)ad);
    fwd_col(q1, sa, c1.h); fwd_col(q2, sa, c2.h);
    float qa = c1.h[q1->n_layers][0], qb = c2.h[q2->n_layers][0]; int second = qb < qa; float mn = second ? qb : qa;
    sl += (double)(alpha * lp - mn); slp += (double)lp;
    float dyq = -invB; bwd_col_dx(second ? q2 : q1, second ? c2.h : c1.h, &dyq, NULL, dsa);            /* d(-mean(min Q))/d(vcat(s,a)) */
    const float* mu = ca.h[actor->n_layers]; float clp = alpha * invB;                                  /* d(loss)/d(logprob_j) */
    for (int d = 0; d < ad; ++d) { float sg = expf(ls[d]), s2 = sg * sg, df = a[d] - mu[d];
      float abar = clp * (-(df / s2)) + dsa[od + d];
      dmu[d] = clp * (df / s2) + abar;
      actor->g[xoff(actor) + d] += clp * ((df * df) / s2 - 1.f) + abar * (eps[d] * sg); }
    bwd_col_dx(actor, ca.h, dmu, actor->g, NULL);
  }
  cc_free(actor, &ca); cc_free(q1, &c1); cc_free(q2, &c2);
  info[CRUX_INFO_LOSS] = (float)(sl / (double)n); info[CRUX_INFO_ENTROPY] = (float)(-(slp / (double)n));
  info[CRUX_INFO_GRAD_NORM] = (float)sqrt(sumsq_tensors(actor));
  if (isnan(info[CRUX_INFO_GRAD_NORM])) return CRUX_ENAN;
  return orc_adam_apply(actor, 1.0f);
}

/* ============================================================================================
 * DDPG / TD3                       src/model_free/rl/ddpg.jl:6-26, td3.jl:4-12, policies.jl:510-514
 * ============================================================================================ */
static float clampf(float x, float lo, float hi) { return x < lo ? lo : x > hi ? hi : x; }
int32_t orc_dpg_target(orc_mlp* actor_t, orc_mlp* q1t, orc_mlp* q2t, orc_buffer* b, float gamma, float sigma, float eps_min, float eps_max, float a_min, float a_max,
                       uint64_t seed, uint64_t counter, float* y) {
  int64_t n = b->elements; int od = b->obs_dim, ad = b->act_dim;
  if (b->act_kind != CRUX_ACTION_CONTINUOUS || actor_t->dims[0] != od || actor_t->dims[actor_t->n_layers] != ad || q1t->dims[0] != od + ad || (q2t && q2t->dims[0] != od + ad)) return CRUX_EINVAL;
  colcache ca = cc_alloc(actor_t), c1 = cc_alloc(q1t), c2; if (q2t) c2 = cc_alloc(q2t);
  const float* SP = (const float*)b->col[CRUX_COL_SP]; const float* R = (const float*)b->col[CRUX_COL_R]; const uint8_t* D = (const uint8_t*)b->col[CRUX_COL_DONE];
  float sa[1024];
  for (int64_t j = 0; j < n; ++j) {
    fwd_col(actor_t, SP + (size_t)j * od, ca.h); memcpy(sa, SP + (size_t)j * od, 4 * (size_t)od);
    for (int d = 0; d < ad; ++d) { float a = ca.h[actor_t->n_layers][d];                               /* action(pi, sp) */
      if (sigma >= 0.f) { float e = randn_f32(seed, counter, (uint32_t)(j * ad + d), 0) * sigma; a = clampf(a + clampf(e, eps_min, eps_max), a_min, a_max); }   /* policies.jl:512-513 */
      sa[od + d] = a; }
    fwd_col(q1t, sa, c1.h); float q = c1.h[q1t->n_layers][0];
    if (q2t) { fwd_col(q2t, sa, c2.h); float qb = c2.h[q2t->n_layers][0]; q = qb < q ? qb : q; }       /* min.(value(pi, sp, ap)...) td3.jl:6 */
    y[j] = R[j] + (gamma * (1.f - (D[j] ? 1.f : 0.f))) * q;
  }
  cc_free(actor_t, &ca); cc_free(q1t, &c1); if (q2t) cc_free(q2t, &c2); return CRUX_OK;
}
/* train!(critic, td_loss) with value(pi, s, a) = net(vcat(s, a)) (utils.jl:76-87, policies.jl:96) */
int32_t orc_q_step(orc_mlp* q, orc_buffer* b, const float* y, int32_t use_weight, float* info) {
  int64_t n = b->elements; int od = b->obs_dim, ad = b->act_dim; if (n <= 0 || b->act_kind != CRUX_ACTION_CONTINUOUS || q->dims[0] != od + ad || q->dims[q->n_layers] != 1) return CRUX_EINVAL;
  const float* S = (const float*)b->col[CRUX_COL_S]; const float* A = (const float*)b->col[CRUX_COL_A]; const float* W = use_weight ? (const float*)b->col[CRUX_COL_WEIGHT] : NULL;
  for (int k = 0; k < CRUX_INFO_N; ++k) info[k] = 0.f;
  float sa[1024], invB = 1.f / (float)n; double sl = 0, sq = 0; colcache c = cc_alloc(q); memset(q->g, 0, 4 * (size_t)q->n_params);
  for (int64_t j = 0; j < n; ++j) { memcpy(sa, S + (size_t)j * od, 4 * (size_t)od); memcpy(sa + od, A + (size_t)j * ad, 4 * (size_t)ad);
    fwd_col(q, sa, c.h); float Q = c.h[q->n_layers][0], d = Q - y[j], w = W ? W[j] : 1.f; sl += (double)(d * d * w); sq += (double)Q;
    float dy = 2.f * d * w * invB; bwd_col_dx(q, c.h, &dy, q->g, NULL); }
  cc_free(q, &c);
  info[CRUX_INFO_LOSS] = (float)(sl / (double)n); info[CRUX_INFO_Q1AVG] = (float)(sq / (double)n); info[CRUX_INFO_GRAD_NORM] = (float)sqrt(sumsq_tensors(q));
  if (isnan(info[CRUX_INFO_GRAD_NORM])) return CRUX_ENAN;
  return orc_adam_apply(q, 1.0f);
}
/* train!(actor, ddpg_actor_loss) ddpg.jl:26 / td3_actor_loss td3.jl:12: -mean(value(Q, s, action(pi, s))) */
int32_t orc_dpg_actor_step(orc_mlp* actor, orc_mlp* q, orc_buffer* b, float* info) {
  int64_t n = b->elements; int od = b->obs_dim, ad = b->act_dim;
  if (n <= 0 || b->act_kind != CRUX_ACTION_CONTINUOUS || actor->dims[0] != od || actor->dims[actor->n_layers] != ad || q->dims[0] != od + ad || q->dims[q->n_layers] != 1) return CRUX_EINVAL;
  colcache ca = cc_alloc(actor), cq = cc_alloc(q); const float* S = (const float*)b->col[CRUX_COL_S];
  for (int k = 0; k < CRUX_INFO_N; ++k) info[k] = 0.f;
  float sa[1024], dsa[1024], invB = 1.f / (float)n; double sq = 0; memset(actor->g, 0, 4 * (size_t)actor->n_params);
  for (int64_t j = 0; j < n; ++j) { fwd_col(actor, S + (size_t)j * od, ca.h); memcpy(sa, S + (size_t)j * od, 4 * (size_t)od); memcpy(sa + od, ca.h[actor->n_layers], 4 * (size_t)ad);
    fwd_col(q, sa, cq.h); sq += (double)cq.h[q->n_layers][0];
    float dyq = -invB; bwd_col_dx(q, cq.h, &dyq, NULL, dsa); bwd_col_dx(actor, ca.h, dsa + od, actor->g, NULL); }
  cc_free(actor, &ca); cc_free(q, &cq);
  info[CRUX_INFO_LOSS] = (float)(-(sq / (double)n)); info[CRUX_INFO_GRAD_NORM] = (float)sqrt(sumsq_tensors(actor));
  if (isnan(info[CRUX_INFO_GRAD_NORM])) return CRUX_ENAN;
  return orc_adam_apply(actor, 1.0f);
}

/* ---- OnPolicyGAIL pieces (src/model_free/il/on_policy_gail.jl:1-5,49-54; src/extras/gans.jl:7-9) -----------------------------------------
 * Flux.Losses.logitbinarycrossentropy(z, y) = mean((1 - y) z - logsigmoid(z)) (Flux 0.13/0.14, third-party, restated from its docs);
 * NNlib.logsigmoid(x) = -softplus(-x), softplus(x) = log1p(exp(-|x|)) + relu(x).                                                        */
static float orc_logsigmoid(float x) { float nx = -x; float sp = log1pf(expf(-fabsf(nx))) + (nx > 0.f ? nx : 0.f); return -sp; }
static void gail_input(const orc_buffer* b, int64_t j, float* out) {        /* vcat(a, s): the action first (on_policy_gail.jl:3, :50) */
  int od = b->obs_dim, ad = b->act_dim;
  if (b->act_kind == CRUX_ACTION_CONTINUOUS) memcpy(out, (const float*)b->col[CRUX_COL_A] + (size_t)j * ad, 4 * (size_t)ad);
  else { const uint8_t* a = (const uint8_t*)b->col[CRUX_COL_A] + (size_t)j * ad; for (int k = 0; k < ad; ++k) out[k] = a[k] ? 1.f : 0.f; }
  memcpy(out + ad, (const float*)b->col[CRUX_COL_S] + (size_t)j * od, 4 * (size_t)od);
}
/* train!(D, gail_d_loss(GAN_BCELoss())) on rows [off_ex, off_ex + n_ex) of the expert buffer and [off_pi, off_pi + n_pi) of the policy buffer:
 * L = LBCE(D(vcat(a_ex, s_ex)), 1) + LBCE(D(vcat(a_pi, s_pi)), 0) */
int32_t orc_gail_d_step(orc_mlp* D, orc_buffer* ex, int64_t off_ex, int64_t n_ex, orc_buffer* pi, int64_t off_pi, int64_t n_pi, float* info) {
  int od = ex->obs_dim, ad = ex->act_dim;
  if (n_ex <= 0 || n_pi <= 0 || off_ex < 0 || off_pi < 0 || off_ex + n_ex > ex->elements || off_pi + n_pi > pi->elements || pi->obs_dim != od || pi->act_dim != ad ||
      pi->act_kind != ex->act_kind || D->dims[0] != od + ad || D->dims[D->n_layers] != 1) return CRUX_EINVAL;
  for (int k = 0; k < CRUX_INFO_N; ++k) info[k] = 0.f;
  float x[1024]; double l_ex = 0, l_pi = 0; colcache c = cc_alloc(D); memset(D->g, 0, 4 * (size_t)D->n_params);
  for (int64_t j = 0; j < n_ex; ++j) { gail_input(ex, off_ex + j, x); fwd_col(D, x, c.h); float z = c.h[D->n_layers][0];
    l_ex += (double)(-orc_logsigmoid(z)); float sg = 1.f / (1.f + expf(-z)); float dz = (sg - 1.f) / (float)n_ex; bwd_col_dx(D, c.h, &dz, D->g, NULL); }
  for (int64_t j = 0; j < n_pi; ++j) { gail_input(pi, off_pi + j, x); fwd_col(D, x, c.h); float z = c.h[D->n_layers][0];
    l_pi += (double)(z - orc_logsigmoid(z)); float sg = 1.f / (1.f + expf(-z)); float dz = sg / (float)n_pi; bwd_col_dx(D, c.h, &dz, D->g, NULL); }
  cc_free(D, &c);
  info[CRUX_INFO_LOSS] = (float)(l_ex / (double)n_ex) + (float)(l_pi / (double)n_pi); info[CRUX_INFO_GRAD_NORM] = (float)sqrt(sumsq_tensors(D));
  if (isnan(info[CRUX_INFO_GRAD_NORM])) return CRUX_ENAN;
  return orc_adam_apply(D, 1.0f);
}
/* GAIL_callback reward (on_policy_gail.jl:50-55): D_out = value(D, a, s); r = ar logsigmoid(D_out) - (1 - ar) logcompsigmoid(D_out); D[:r] .= r .* Rscale */
int32_t orc_gail_reward(orc_mlp* D, orc_buffer* b, float alpha_r, float rscale, float* mean_r) {
  int od = b->obs_dim, ad = b->act_dim; int64_t n = b->elements;
  if (n <= 0 || D->dims[0] != od + ad || D->dims[D->n_layers] != 1) return CRUX_EINVAL;
  float x[1024]; colcache c = cc_alloc(D); double sr = 0; float* R = (float*)b->col[CRUX_COL_R];
  for (int64_t j = 0; j < n; ++j) { gail_input(b, j, x); fwd_col(D, x, c.h); float z = c.h[D->n_layers][0];
    float ls = orc_logsigmoid(z), lc = ls - z; float r = alpha_r * ls - (1.f - alpha_r) * lc; sr += (double)r; R[j] = r * rscale; }
  cc_free(D, &c);
  if (mean_r) *mean_r = (float)(sr / (double)n);
  return CRUX_OK;
}

/* test hooks for the randomness spec in include/crux_rng.h */
void orc_perm(uint64_t seed, uint64_t counter, uint32_t n, int64_t* out) {
  crux_perm p = crux_perm_make(seed, counter, 0, n);
  for (uint32_t j = 0; j < n; ++j) out[j] = (int64_t)crux_perm_at(&p, j);
}
void orc_philox(uint64_t seed, uint64_t counter, uint32_t stream, uint32_t purpose, uint32_t* out4) {
  crux_u32x4 x = crux_philox(seed, counter, stream, purpose); for (int i = 0; i < 4; ++i) out4[i] = x.v[i];
}
