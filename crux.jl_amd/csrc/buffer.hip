// buffer.hip -- ExperienceBuffer: device SoA columns, ring push, gather, permutation, priorities.
// Reference: src/experience_buffer.jl (mdp_data :4-35, ExperienceBuffer :53-80, shuffle! :118-124,
// minibatch :170-171, get_last_N_indices :223-229, push! :232-259, update_priorities! :290-301).
#include "common.h"
#include <algorithm>
#include "exec.h"
#include "ops_small.h"

void crux_buffer_topo_free(crux_buffer* b);

// ---- kernels ------------------------------------------------------------------------------------
// row gather/scatter on a column: dst[dst_idx[j]] = src[src_idx[j]] (idx NULL => j). Rows are `stride`
// bytes; W = access width in bytes (4 when stride % 4 == 0, else 1). Consecutive threads touch
// consecutive bytes of a row, so a wave covers whole rows contiguously.
template <typename T>
__global__ void k_copy_rows(T* __restrict__ dst, const int64_t* __restrict__ dst_idx, const T* __restrict__ src,
                            const int64_t* __restrict__ src_idx, int64_t n, int32_t row_elems) {
  const int64_t total = n * row_elems;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = t / row_elems; const int32_t e = (int32_t)(t - j * row_elems);
    const int64_t sj = src_idx ? src_idx[j] : j, dj = dst_idx ? dst_idx[j] : j;
    dst[dj * row_elems + e] = src[sj * row_elems + e];
  }
}
__global__ void k_copy_rows_i32idx(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, const int32_t* __restrict__ src_idx,
                                   int64_t n, int32_t row_elems) {
  const int64_t total = n * row_elems;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = t / row_elems; const int32_t e = (int32_t)(t - j * row_elems);
    dst[j * row_elems + e] = src[(int64_t)src_idx[j] * row_elems + e];
  }
}
__global__ void k_copy_rows_i32idx_u8(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, const int32_t* __restrict__ src_idx,
                                      int64_t n, int32_t row_elems) {
  const int64_t total = n * row_elems;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = t / row_elems; const int32_t e = (int32_t)(t - j * row_elems);
    dst[j * row_elems + e] = src[(int64_t)src_idx[j] * row_elems + e];
  }
}
__global__ void k_fill_f32(float* p, float v, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void k_fill_f32_idx(float* p, const int64_t* idx, float v, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[idx[i]] = v;
}

// update_priorities! (:290-301): val = v + eps(Float32); priorities[I] = val^alpha; max/min track the
// un-powered val (Float32 fields). Positive floats order like their bit patterns, so atomicMax/Min on
// the int view reproduce the sequential max/min exactly.
__global__ void k_per_update(float* __restrict__ pr, float* pminmax, const int64_t* __restrict__ I,
                             const double* __restrict__ v64, const float* __restrict__ v32, const float* vconst_from_max,
                             float alpha, int64_t n) { PerUpdateOp::run(blockIdx.x, gridDim.x, pr, pminmax, I, v64, v32, vconst_from_max, alpha, n); }

static unsigned grid_for(int64_t total) { int64_t nb = (total + 255) / 256; if (nb < 1) nb = 1; if (nb > 8192) nb = 8192; return (unsigned)nb; }

// copy rows between columns with 64-bit index arrays resident on the device
static void launch_copy_rows(crux_ctx* c, void* dst, const int64_t* d_dst_idx, const void* src, const int64_t* d_src_idx, int64_t n, size_t stride) {
  if (n <= 0) return;
  if (stride % 4 == 0) { const int32_t re = (int32_t)(stride / 4);
    hipLaunchKernelGGL(k_copy_rows<uint32_t>, dim3(grid_for(n * re)), dim3(256), 0, c->stream, (uint32_t*)dst, d_dst_idx, (const uint32_t*)src, d_src_idx, n, re); }
  else { const int32_t re = (int32_t)stride;
    hipLaunchKernelGGL(k_copy_rows<uint8_t>, dim3(grid_for(n * re)), dim3(256), 0, c->stream, (uint8_t*)dst, d_dst_idx, (const uint8_t*)src, d_src_idx, n, re); }
}

extern "C" int32_t crux_buffer_indices(const crux_buffer* cb, int64_t* out, int64_t n);
int32_t crux_per_touched(crux_buffer* b, const int64_t* d_ids, int64_t n, bool from_push, unsigned* ticket = nullptr);   // per.hip: priorities of these elements changed
// exported to other translation units ------------------------------------------------------------------
int32_t crux_buffer_ring_indices(crux_buffer* b, int64_t N, std::vector<int64_t>& I) {
  I.resize((size_t)N);
  for (int64_t j = 0; j < N; ++j) I[(size_t)j] = (b->next_ind + j) % b->capacity;   // mod1.(next_ind:next_ind+N-1, C) (:236)
  return CRUX_OK;
}
void crux_buffer_ring_advance(crux_buffer* b, int64_t N) {   // :256-257
  b->total_count += N;
  b->elements = b->elements + N < b->capacity ? b->elements + N : b->capacity;
  b->next_ind = (b->next_ind + N) % b->capacity;
}
// priorities of freshly pushed rows (:254); d_I device index array
int32_t crux_buffer_per_on_push(crux_buffer* b, const int64_t* d_I, int64_t N) {
  if (!b->prioritized || N <= 0) return CRUX_OK;
  // all pushed rows get (max_priority + eps)^alpha with max_priority read ONCE before the update (push!: update_priorities!(b, I, max_priority*ones(N)),
  // experience_buffer.jl:254): the value is snapshotted into pminmax[2] by a stream-ordered copy, because the kernel's atomicMax raises
  // pminmax[0] to Float32(m + eps) while other waves are still reading it (for m in [1,2) that is the next float up)
  if (crux_exec_recording(b->ctx)) crux_exec_push<CopyF32Op, OP_COPY_F32>(b->ctx, 1u, b->pminmax + 2, (const float*)b->pminmax, (int64_t)1);
  else HIPCHK(b->ctx, hipMemcpyAsync(b->pminmax + 2, b->pminmax, 4, hipMemcpyDeviceToDevice, b->ctx->stream));
  CRUX_RUN(b->ctx, PerUpdateOp, OP_PER_UPDATE, k_per_update, grid_for(N), 256, b->ctx->stream, b->priorities, b->pminmax, d_I, (const double*)nullptr, (const float*)nullptr, (const float*)(b->pminmax + 2), b->alpha, N);
  { const int32_t rc = crux_launch_check(b->ctx, "k_per_update(push)"); if (rc) return rc; }
  return crux_per_touched(b, d_I, N, true);
}
// physical permutation of every column by a device int32 order: new[:,j] = old[:,order[j]]
int32_t crux_buffer_apply_order(crux_buffer* b, const int32_t* d_order, int64_t n) {
  crux_ctx* c = b->ctx;
  size_t maxst = 0; for (int k = 0; k < CRUX_NCOLS; ++k) if (has_col(b, k) && col_stride(b, k) > maxst) maxst = col_stride(b, k);
  void* tmp = crux_scratch(c, maxst * (size_t)n + 256);
  if (!tmp) return crux_fail(c, CRUX_ENOMEM, "apply_order: scratch");
  for (int k = 0; k < CRUX_NCOLS; ++k) {
    if (!has_col(b, k)) continue;
    const size_t st = col_stride(b, k);
    if (st % 4 == 0) { const int32_t re = (int32_t)(st / 4);
      hipLaunchKernelGGL(k_copy_rows_i32idx, dim3(grid_for(n * re)), dim3(256), 0, c->stream, (uint32_t*)tmp, (const uint32_t*)b->col[k], d_order, n, re); }
    else { const int32_t re = (int32_t)st;
      hipLaunchKernelGGL(k_copy_rows_i32idx_u8, dim3(grid_for(n * re)), dim3(256), 0, c->stream, (uint8_t*)tmp, (const uint8_t*)b->col[k], d_order, n, re); }
    HIPCHK(c, hipMemcpyAsync(b->col[k], tmp, st * (size_t)n, hipMemcpyDeviceToDevice, c->stream));
  }
  return crux_launch_check(c, "apply_order");
}

// the same for n buffers of equal shape: two launches per column (grid.y = buffer) instead of 2 n
struct ApplyPtrs { void* col[CRUX_NCOLS]; void* tmp; const int32_t* order; };
template <typename T>
__global__ void k_apply_order_multi(const ApplyPtrs* __restrict__ P, int k, int64_t n, int32_t re, int back) {
  const ApplyPtrs& p = P[blockIdx.y]; const int32_t* __restrict__ order = p.order; if (!order) return;      // (fields read in place: a private copy of the struct, indexed by k, lived in scratch)
  T* col = (T*)p.col[k]; T* tmp = (T*)p.tmp; const int64_t total = n * re;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    if (back) col[t] = tmp[t];
    else { const int64_t j = t / re; const int32_t e = (int32_t)(t - j * re); tmp[t] = col[(int64_t)order[j] * re + e]; }
  }
}
int32_t crux_buffer_apply_order_multi(int32_t n, crux_buffer* const* bufs, const int32_t* const* d_orders) {
  if (n < 1) return CRUX_OK;
  crux_ctx* c = bufs[0]->ctx; const int64_t len = bufs[0]->elements;
  bool same = true; for (int i = 1; i < n; ++i) same = same && bufs[i]->mask == bufs[0]->mask && bufs[i]->elements == len && bufs[i]->obs_dim == bufs[0]->obs_dim && bufs[i]->act_dim == bufs[0]->act_dim && bufs[i]->act_kind == bufs[0]->act_kind;
  if (!same || n == 1) { for (int i = 0; i < n; ++i) if (d_orders[i]) { const int32_t rc = crux_buffer_apply_order(bufs[i], d_orders[i], bufs[i]->elements); if (rc) return rc; } return CRUX_OK; }
  size_t maxst = 0; for (int k = 0; k < CRUX_NCOLS; ++k) if (has_col(bufs[0], k) && col_stride(bufs[0], k) > maxst) maxst = col_stride(bufs[0], k);
  const size_t tb = (maxst * (size_t)len + 255) / 256 * 256, pb = (sizeof(ApplyPtrs) * (size_t)n + 255) / 256 * 256;
  char* sc = (char*)crux_scratch(c, pb + tb * (size_t)n + 256); if (!sc) return crux_fail(c, CRUX_ENOMEM, "apply_order (multi): scratch");
  std::vector<ApplyPtrs> hp((size_t)n);
  for (int i = 0; i < n; ++i) { for (int k = 0; k < CRUX_NCOLS; ++k) hp[(size_t)i].col[k] = bufs[i]->col[k]; hp[(size_t)i].tmp = sc + pb + tb * (size_t)i; hp[(size_t)i].order = d_orders[i]; }
  HIPCHK(c, hipMemcpyAsync(sc, hp.data(), sizeof(ApplyPtrs) * (size_t)n, hipMemcpyHostToDevice, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int k = 0; k < CRUX_NCOLS; ++k) {
    if (!has_col(bufs[0], k)) continue;
    const size_t st = col_stride(bufs[0], k);
    for (int back = 0; back < 2; ++back) {
      if (st % 4 == 0) { const int32_t re = (int32_t)(st / 4); unsigned gx = grid_for(len * re); if (gx > 512) gx = 512;
        hipLaunchKernelGGL(k_apply_order_multi<uint32_t>, dim3(gx, (unsigned)n), dim3(256), 0, c->stream, (const ApplyPtrs*)sc, k, len, re, back); }
      else { const int32_t re = (int32_t)st; unsigned gx = grid_for(len * re); if (gx > 512) gx = 512;
        hipLaunchKernelGGL(k_apply_order_multi<uint8_t>, dim3(gx, (unsigned)n), dim3(256), 0, c->stream, (const ApplyPtrs*)sc, k, len, re, back); }
    }
  }
  return crux_launch_check(c, "apply_order (multi)");
}

extern "C" {

int32_t crux_buffer_create(crux_ctx* ctx, int32_t obs_dim, int32_t act_dim, int32_t act_kind, int64_t capacity, uint32_t column_mask,
                           int32_t prioritized, float alpha, crux_buffer** out) {
  if (!ctx || !out) return CRUX_EINVAL;
  if (obs_dim < 1 || act_dim < 1 || capacity < 1 || capacity > 0x7fffffffLL || (act_kind != CRUX_ACTION_DISCRETE && act_kind != CRUX_ACTION_CONTINUOUS))
    return crux_fail(ctx, CRUX_EINVAL, "buffer_create: bad shape obs=%d act=%d cap=%lld", obs_dim, act_dim, (long long)capacity);
  crux_buffer* b = new crux_buffer(); b->ctx = ctx;
  b->obs_dim = obs_dim; b->act_dim = act_dim; b->act_kind = act_kind; b->capacity = capacity;
  b->mask = (column_mask & ((1u << CRUX_NCOLS) - 1)) | 0x3Fu;
  if (prioritized) b->mask |= 1u << CRUX_COL_WEIGHT;                                   // :71
  for (int k = 0; k < CRUX_NCOLS; ++k) {
    if (!has_col(b, k)) continue;
    const size_t bytes = col_stride(b, k) * (size_t)capacity;
    if (hipMalloc(&b->col[k], bytes) != hipSuccess) { crux_buffer_destroy(b); return crux_fail(ctx, CRUX_ENOMEM, "buffer_create: hipMalloc(%zu) failed", bytes); }
    HIPCHK(ctx, hipMemsetAsync(b->col[k], 0, bytes, ctx->stream));
    if (CRUX_COL_INIT_ONE(k)) hipLaunchKernelGGL(k_fill_f32, dim3(grid_for(capacity)), dim3(256), 0, ctx->stream, (float*)b->col[k], 1.0f, capacity);   // :17-19
  }
  b->prioritized = prioritized != 0; b->alpha = alpha;
  if (hipMalloc(&b->d_indices, sizeof(int64_t) * (size_t)capacity) != hipSuccess || hipMalloc(&b->order_a, 4 * (size_t)capacity) != hipSuccess ||
      hipMalloc(&b->order_b, 4 * (size_t)capacity) != hipSuccess ||
      hipMalloc(&b->order_c, 4 * (size_t)capacity) != hipSuccess || hipMalloc(&b->order_d, 4 * (size_t)capacity) != hipSuccess) { crux_buffer_destroy(b); return crux_fail(ctx, CRUX_ENOMEM, "buffer_create: index arrays"); }
  if (b->prioritized) {
    if (hipMalloc(&b->priorities, 4 * (size_t)capacity) != hipSuccess || hipMalloc(&b->cumsum, 4 * (size_t)capacity) != hipSuccess ||
        hipMalloc(&b->pminmax, 16) != hipSuccess) { crux_buffer_destroy(b); return crux_fail(ctx, CRUX_ENOMEM, "buffer_create: priorities"); }
    HIPCHK(ctx, hipMemsetAsync(b->priorities, 0, 4 * (size_t)capacity, ctx->stream));
    const float mm[2] = {1.0f, INFINITY};                                              // PriorityParams :38-50
    HIPCHK(ctx, hipMemcpyAsync(b->pminmax, mm, 8, hipMemcpyHostToDevice, ctx->stream));
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  *out = b; return CRUX_OK;
}

int32_t crux_buffer_destroy(crux_buffer* b) {
  if (!b) return CRUX_OK;
  crux_sync_before_free(b->ctx);
  crux_buffer_topo_free(b);
  for (int k = 0; k < CRUX_NCOLS; ++k) if (b->col[k]) (void)hipFree(b->col[k]);
  if (b->priorities) (void)hipFree(b->priorities); if (b->cumsum) (void)hipFree(b->cumsum); if (b->pminmax) (void)hipFree(b->pminmax);
  if (b->d_indices) (void)hipFree(b->d_indices); if (b->order_a) (void)hipFree(b->order_a); if (b->order_b) (void)hipFree(b->order_b);
  if (b->order_c) (void)hipFree(b->order_c); if (b->order_d) (void)hipFree(b->order_d);
  if (b->pack) (void)hipFree(b->pack);
  for (int q = 0; q < 2; ++q) if (b->ord_all[q]) (void)hipFree(b->ord_all[q]);
  if (b->aux_ones) (void)hipFree(b->aux_ones); if (b->aux_zeros) (void)hipFree(b->aux_zeros);
  delete b; return CRUX_OK;
}

int64_t crux_buffer_len(const crux_buffer* b) { return b ? b->elements : -1; }
int64_t crux_buffer_capacity(const crux_buffer* b) { return b ? b->capacity : -1; }
int64_t crux_buffer_next_ind(const crux_buffer* b) { return b ? b->next_ind : -1; }
int64_t crux_buffer_total_count(const crux_buffer* b) { return b ? b->total_count : -1; }
int32_t crux_buffer_has_column(const crux_buffer* b, int32_t key) { return b && has_col(b, key) ? 1 : 0; }

int32_t crux_buffer_clear(crux_buffer* b) {                                            // clear! :97-104
  if (!b) return CRUX_EINVAL;
  b->elements = 0; b->next_ind = 0; b->total_count = 0; b->indices.clear(); b->indices_n = 0; b->indices_stale = false;
  if (b->prioritized) {
    HIPCHK(b->ctx, hipMemsetAsync(b->priorities, 0, 4 * (size_t)b->capacity, b->ctx->stream));
    const float inf = INFINITY;                                                         // PriorityParams(N, pp) keeps max_priority, resets min
    HIPCHK(b->ctx, hipMemcpyAsync(b->pminmax + 1, &inf, 4, hipMemcpyHostToDevice, b->ctx->stream));
    HIPCHK(b->ctx, hipStreamSynchronize(b->ctx->stream));
    b->cumsum_valid = false; b->per_full_dirty = true;
  }
  return CRUX_OK;
}

int32_t crux_buffer_column_info(const crux_buffer* b, int32_t key, int32_t* eb, int32_t* rows) {
  if (!b || !has_col(b, key)) return CRUX_EINVAL;
  if (eb) *eb = col_elem(b, key); if (rows) *rows = col_rows(b, key); return CRUX_OK;
}
int32_t crux_buffer_column_ptr(crux_buffer* b, int32_t key, void** d_ptr) {
  if (!b || !d_ptr) return CRUX_EINVAL;
  if (!has_col(b, key)) return crux_fail(b->ctx, CRUX_EINVAL, "buffer has no column %d", key);
  *d_ptr = b->col[key]; return CRUX_OK;
}

// push! of a SMALL host block in one launch (the caller-stepped environment seam pushes E x T rows of six to nine columns per call -- C3: four rows):
// the columns are gathered into the context's pinned, device-mapped staging block with plain memcpy and ONE kernel writes every column's rows to their ring positions,
// instead of one staged upload per column and ring segment (each a dependent stream operation of ~5-10 us from pageable memory).
struct PushCols { int n; unsigned src_off[CRUX_NCOLS]; unsigned stride[CRUX_NCOLS]; char* dst[CRUX_NCOLS]; };
__global__ __launch_bounds__(256) void k_push_cols(PushCols pc, const char* __restrict__ stage, int64_t N, int64_t first, int64_t C) {
  const int k = blockIdx.y; if (k >= pc.n) return;
  const unsigned st = pc.stride[k]; const char* src = stage + pc.src_off[k]; char* dst = pc.dst[k];
  if ((st & 3u) == 0u) { const unsigned w = st >> 2; const int64_t tot = N * (int64_t)w;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) { const int64_t j = i / w; const unsigned q = (unsigned)(i - j * w);
      ((uint32_t*)(dst + (size_t)((first + j) % C) * st))[q] = ((const uint32_t*)(src + (size_t)j * st))[q]; } }
  else { const int64_t tot = N * (int64_t)st;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) { const int64_t j = i / st; const unsigned q = (unsigned)(i - j * st);
      dst[(size_t)((first + j) % C) * st + q] = src[(size_t)j * st + q]; } }
}
int32_t crux_buffer_push_host(crux_buffer* b, int64_t N, const void* const* cols, int64_t* I_out) {   // push! :232-259
  if (!b || N < 0) return CRUX_EINVAL;
  crux_ctx* c = b->ctx;
  if (N == 0) return CRUX_OK;
  std::vector<int64_t> I; crux_buffer_ring_indices(b, N, I);
  const int64_t C = b->capacity;
  size_t total = 0; for (int k = 0; k < CRUX_NCOLS; ++k) if (has_col(b, k) && cols && cols[k]) total += (col_stride(b, k) * (size_t)N + 15) / 16 * 16;
  if (N <= C && total > 0 && total <= (1u << 20) && !b->prioritized && crux_sw().host_zerocopy) {      // (a prioritized ring goes on below: its priority bookkeeping needs the device index array anyway)
    char* hs = (char*)crux_pinned_mapped(c, total); if (!hs) return crux_fail(c, CRUX_ENOMEM, "push!: pinned staging");
    PushCols pc{}; size_t off = 0;
    for (int k = 0; k < CRUX_NCOLS; ++k) { if (!has_col(b, k) || !cols[k]) continue;
      const size_t st = col_stride(b, k); memcpy(hs + off, cols[k], st * (size_t)N);
      pc.src_off[pc.n] = (unsigned)off; pc.stride[pc.n] = (unsigned)st; pc.dst[pc.n] = (char*)b->col[k]; pc.n += 1; off += (st * (size_t)N + 15) / 16 * 16; }
    const unsigned gx = (unsigned)std::min<int64_t>(64, (N * 8 + 255) / 256 + 1);
    hipLaunchKernelGGL(k_push_cols, dim3(gx, (unsigned)pc.n), dim3(256), 0, c->stream, pc, (const char*)c->pinned_mapped_dev, N, b->next_ind, C);
    int32_t rc = crux_launch_check(c, "k_push_cols"); if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));      // the staging block is the context's: it must be free again when the call returns (and so must the caller's arrays, as before)
    crux_buffer_ring_advance(b, N);
    if (I_out) memcpy(I_out, I.data(), 8 * (size_t)N);
    return CRUX_OK;
  }
  for (int k = 0; k < CRUX_NCOLS; ++k) {
    if (!has_col(b, k) || !cols || !cols[k]) continue;                                   // :238-241
    const size_t st = col_stride(b, k);
    // the ring write is at most N/C+1 wraps of contiguous segments; later writes win like copyto! in index order
    int64_t j = 0;
    while (j < N) {
      const int64_t pos = I[(size_t)j]; int64_t run = C - pos; if (run > N - j) run = N - j;
      HIPCHK(c, hipMemcpyAsync((char*)b->col[k] + (size_t)pos * st, (const char*)cols[k] + (size_t)j * st, st * (size_t)run, hipMemcpyHostToDevice, c->stream));
      j += run;
    }
  }
  if (b->prioritized) {
    if (b->indices_stale) { int64_t dummy = 0; (void)crux_buffer_indices(b, &dummy, 0); }     // d_indices is about to be reused: bring the host mirror of the last sample up to date first
    HIPCHK(c, hipMemcpyAsync(b->d_indices, I.data(), 8 * (size_t)(N < C ? N : C), hipMemcpyHostToDevice, c->stream));
    int32_t rc = crux_buffer_per_on_push(b, b->d_indices, N < C ? N : C); if (rc) return rc;
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  crux_buffer_ring_advance(b, N);
  if (I_out) memcpy(I_out, I.data(), 8 * (size_t)N);
  return CRUX_OK;
}

// push_reservoir!(buffer, data; weighted) (src/experience_buffer.jl:262-288). The per-element decisions are a sequential recurrence on
// (length, total_count) and are taken on the host from the host-resident data (same rules and the same Philox draws as the oracle: see
// crux_rng.h CRUX_RNG_RESERVOIR); the rows then move in two steps: the run that fills the ring through the ordinary push!, the
// replacements (last write per slot wins, like the reference's sequential assignments) through one staged upload + one scatter launch per column.
// Quirks kept: total_count grows by 2 per element while the buffer fills (:272 and push! :235); replaced slots keep their priorities.
int32_t crux_buffer_push_reservoir(crux_buffer* b, int64_t N, const void* const* cols, int32_t weighted, uint64_t seed, uint64_t counter) {
  if (!b || N < 0 || !cols) return CRUX_EINVAL;
  crux_ctx* c = b->ctx; if (N == 0) return CRUX_OK;
  const float* W = (weighted && has_col(b, CRUX_COL_WEIGHT) && cols[CRUX_COL_WEIGHT]) ? (const float*)cols[CRUX_COL_WEIGHT] : nullptr;
  std::vector<int64_t> fill; std::vector<std::pair<int64_t, int64_t>> repl;   // source rows pushed in order; (slot, source row) assignments in order
  int64_t elements = b->elements, total = b->total_count;
  for (int64_t i = 0; i < N; ++i) {
    const crux_u32x4 x = crux_philox(seed, counter + (uint64_t)i, 0, CRUX_RNG_RESERVOIR);
    if (W && crux_u32x2_to_f64(x.v[0], x.v[1]) > (double)W[i]) continue;
    total += 1;
    if (elements < b->capacity) { fill.push_back(i); elements += 1; total += 1; }
    else { const uint64_t r = ((uint64_t)x.v[2] << 32) | (uint64_t)x.v[3];
      const int64_t j = 1 + (int64_t)(((unsigned __int128)r * (unsigned __int128)(uint64_t)total) >> 64);
      if (j <= b->capacity) repl.emplace_back(j - 1, i); }
  }
  if (!fill.empty()) {       // the fill phase is a run of 1-row push! calls = one push! of the accepted rows (the ring cannot wrap while it fills)
    size_t maxst = 0; for (int k = 0; k < CRUX_NCOLS; ++k) if (has_col(b, k) && col_stride(b, k) > maxst) maxst = col_stride(b, k);
    std::vector<std::vector<char>> stage(CRUX_NCOLS); const void* ptrs[CRUX_NCOLS];
    for (int k = 0; k < CRUX_NCOLS; ++k) { ptrs[k] = nullptr; if (!has_col(b, k) || !cols[k]) continue; const size_t st = col_stride(b, k);
      stage[k].resize(st * fill.size()); for (size_t q = 0; q < fill.size(); ++q) memcpy(stage[k].data() + q * st, (const char*)cols[k] + (size_t)fill[q] * st, st);
      ptrs[k] = stage[k].data(); }
    if (b->prioritized) {      // every 1-row push! raises max_priority by eps (update_priorities!: val = v + eps, :293-299), so the rows must go one at a time
      for (size_t q = 0; q < fill.size(); ++q) { const void* one[CRUX_NCOLS];
        for (int k = 0; k < CRUX_NCOLS; ++k) one[k] = ptrs[k] ? (const char*)ptrs[k] + q * col_stride(b, k) : nullptr;
        int32_t rc = crux_buffer_push_host(b, 1, one, nullptr); if (rc) return rc; }
    } else { int32_t rc = crux_buffer_push_host(b, (int64_t)fill.size(), ptrs, nullptr); if (rc) return rc; }
  }
  b->total_count = total;
  if (!repl.empty()) {
    std::vector<int64_t> slot_src((size_t)b->capacity, -1); for (auto& pr : repl) slot_src[(size_t)pr.first] = pr.second;     // last assignment per slot wins
    std::vector<int64_t> dst, src; for (int64_t sidx = 0; sidx < b->capacity; ++sidx) if (slot_src[(size_t)sidx] >= 0) { dst.push_back(sidx); src.push_back(slot_src[(size_t)sidx]); }
    const size_t n = dst.size(); size_t maxst = 0; for (int k = 0; k < CRUX_NCOLS; ++k) if (has_col(b, k) && col_stride(b, k) > maxst) maxst = col_stride(b, k);
    const size_t ib = ((8 * n + 255) / 256) * 256;
    char* sc = (char*)crux_scratch(c, ib + maxst * n + 256); if (!sc) return crux_fail(c, CRUX_ENOMEM, "push_reservoir!: scratch");
    HIPCHK(c, hipMemcpyAsync(sc, dst.data(), 8 * n, hipMemcpyHostToDevice, c->stream));
    std::vector<char> host(maxst * n);
    for (int k = 0; k < CRUX_NCOLS; ++k) { if (!has_col(b, k) || !cols[k]) continue; const size_t st = col_stride(b, k);
      for (size_t q = 0; q < n; ++q) memcpy(host.data() + q * st, (const char*)cols[k] + (size_t)src[q] * st, st);
      HIPCHK(c, hipMemcpyAsync(sc + ib, host.data(), st * n, hipMemcpyHostToDevice, c->stream));
      launch_copy_rows(c, b->col[k], (const int64_t*)sc, sc + ib, nullptr, (int64_t)n, st);
      HIPCHK(c, hipStreamSynchronize(c->stream)); }       // `host` is reused for the next column
    int32_t rc = crux_launch_check(c, "push_reservoir!"); if (rc) return rc;
  }
  return CRUX_OK;
}

int32_t crux_buffer_push_buffer(crux_buffer* dst, const crux_buffer* src, const int64_t* ids, int64_t N, int64_t* I_out) {
  if (!dst || !src || N < 0) return CRUX_EINVAL;
  crux_ctx* c = dst->ctx;
  if (dst->obs_dim != src->obs_dim || dst->act_dim != src->act_dim || dst->act_kind != src->act_kind)
    return crux_fail(c, CRUX_EINVAL, "push!: column shapes differ (@assert size(v1)[1:end-1] == size(v2)[1:end-1])");   // :251
  if (N == 0) return CRUX_OK;
  if (N > dst->capacity) return crux_fail(c, CRUX_EINVAL, "push!: %lld rows into capacity %lld is not supported by the device gather", (long long)N, (long long)dst->capacity);
  std::vector<int64_t> I; crux_buffer_ring_indices(dst, N, I);
  std::vector<int64_t> sid((size_t)N);
  for (int64_t j = 0; j < N; ++j) { const int64_t id = ids ? ids[j] : j; if (id < 0 || id >= src->capacity) return crux_fail(c, CRUX_EINVAL, "push!: id %lld out of range", (long long)id); sid[(size_t)j] = id; }
  size_t maxst = 0; for (int k = 0; k < CRUX_NCOLS; ++k) if (has_col(dst, k) && has_col(src, k) && col_stride(dst, k) > maxst) maxst = col_stride(dst, k);
  const size_t idx_bytes = ((8 * (size_t)N + 255) / 256) * 256;
  char* sc = (char*)crux_scratch(c, 2 * idx_bytes + maxst * (size_t)N + 256);
  if (!sc) return crux_fail(c, CRUX_ENOMEM, "push!: scratch");
  int64_t* dI = (int64_t*)sc; int64_t* dS = (int64_t*)(sc + idx_bytes); void* tmp = sc + 2 * idx_bytes;
  HIPCHK(c, hipMemcpyAsync(dI, I.data(), 8 * (size_t)N, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(dS, sid.data(), 8 * (size_t)N, hipMemcpyHostToDevice, c->stream));
  crux_prof_begin(c, CRUX_PROF_GATHER);
  for (int k = 0; k < CRUX_NCOLS; ++k) {
    if (!has_col(dst, k) || !has_col(src, k)) continue;
    const size_t st = col_stride(dst, k);
    launch_copy_rows(c, tmp, nullptr, src->col[k], dS, N, st);      // v2 = collect(view(src, ids))  (:250)
    launch_copy_rows(c, dst->col[k], dI, tmp, nullptr, N, st);      // copyto!(view(dst, I), v2)     (:252)
  }
  crux_prof_end(c, CRUX_PROF_GATHER);
  int32_t rc = crux_launch_check(c, "k_copy_rows"); if (rc) return rc;
  if (dst->prioritized) { rc = crux_buffer_per_on_push(dst, dI, N); if (rc) return rc; }
  HIPCHK(c, hipStreamSynchronize(c->stream));   // scratch and host index vectors are reused by the next call
  crux_buffer_ring_advance(dst, N);
  if (I_out) memcpy(I_out, I.data(), 8 * (size_t)N);
  return CRUX_OK;
}

int32_t crux_buffer_read_column(crux_buffer* b, int32_t key, void* host_out, int64_t n) {
  if (!b || !host_out) return CRUX_EINVAL;
  if (!has_col(b, key)) return crux_fail(b->ctx, CRUX_EINVAL, "buffer has no column %d", key);
  if (n < 0 || n > b->capacity) return crux_fail(b->ctx, CRUX_EINVAL, "read_column: n=%lld out of range", (long long)n);
  if (n == 0) return CRUX_OK;
  HIPCHK(b->ctx, hipMemcpyAsync(host_out, b->col[key], col_stride(b, key) * (size_t)n, hipMemcpyDeviceToHost, b->ctx->stream));
  HIPCHK(b->ctx, hipStreamSynchronize(b->ctx->stream));
  return CRUX_OK;
}
int32_t crux_buffer_write_column(crux_buffer* b, int32_t key, const void* host_in, int64_t n) {
  if (!b || !host_in) return CRUX_EINVAL;
  if (!has_col(b, key)) return crux_fail(b->ctx, CRUX_EINVAL, "buffer has no column %d", key);
  if (n < 0 || n > b->capacity) return crux_fail(b->ctx, CRUX_EINVAL, "write_column: n=%lld out of range", (long long)n);
  if (n == 0) return CRUX_OK;
  HIPCHK(b->ctx, hipMemcpyAsync(b->col[key], host_in, col_stride(b, key) * (size_t)n, hipMemcpyHostToDevice, b->ctx->stream));
  HIPCHK(b->ctx, hipStreamSynchronize(b->ctx->stream));
  return CRUX_OK;
}

int32_t crux_buffer_permute(crux_buffer* b, const int64_t* perm) {                      // shuffle! :118-124
  if (!b || !perm) return CRUX_EINVAL;
  crux_ctx* c = b->ctx; const int64_t n = b->elements;
  if (n == 0) return CRUX_OK;
  std::vector<int32_t> o((size_t)n);
  for (int64_t j = 0; j < n; ++j) { if (perm[j] < 0 || perm[j] >= n) return crux_fail(c, CRUX_EINVAL, "shuffle!: perm[%lld]=%lld out of range", (long long)j, (long long)perm[j]); o[(size_t)j] = (int32_t)perm[j]; }
  HIPCHK(c, hipMemcpyAsync(b->order_a, o.data(), 4 * (size_t)n, hipMemcpyHostToDevice, c->stream));
  int32_t rc = crux_buffer_apply_order(b, b->order_a, n); if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return CRUX_OK;
}

int64_t crux_buffer_last_n_indices(const crux_buffer* b, int64_t N, int64_t* out) {   // :223-229
  if (!b || !out) return -1;
  if (N > b->elements) N = b->elements;
  const int64_t C = b->capacity;
  const int64_t start = ((b->next_ind - N) % C + C) % C;
  for (int64_t j = 0; j < N; ++j) out[j] = (start + j) % C;
  return N;
}

int32_t crux_buffer_gather_host(crux_buffer* b, const int64_t* ids, int64_t n, void* const* outs) {   // minibatch_copy :171
  if (!b || !ids || !outs || n < 0) return CRUX_EINVAL;
  crux_ctx* c = b->ctx; if (n == 0) return CRUX_OK;
  for (int64_t j = 0; j < n; ++j) if (ids[j] < 0 || ids[j] >= b->capacity) return crux_fail(c, CRUX_EINVAL, "minibatch: index %lld out of range", (long long)ids[j]);
  size_t maxst = 0; for (int k = 0; k < CRUX_NCOLS; ++k) if (has_col(b, k) && outs[k] && col_stride(b, k) > maxst) maxst = col_stride(b, k);
  const size_t idx_bytes = ((8 * (size_t)n + 255) / 256) * 256;
  char* sc = (char*)crux_scratch(c, idx_bytes + maxst * (size_t)n + 256);
  if (!sc) return crux_fail(c, CRUX_ENOMEM, "minibatch: scratch");
  int64_t* dS = (int64_t*)sc; void* tmp = sc + idx_bytes;
  HIPCHK(c, hipMemcpyAsync(dS, ids, 8 * (size_t)n, hipMemcpyHostToDevice, c->stream));
  for (int k = 0; k < CRUX_NCOLS; ++k) {
    if (!has_col(b, k) || !outs[k]) continue;
    const size_t st = col_stride(b, k);
    launch_copy_rows(c, tmp, nullptr, b->col[k], dS, n, st);
    HIPCHK(c, hipMemcpyAsync(outs[k], tmp, st * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return CRUX_OK;
}

int64_t* crux_buffer_indices_ptr(crux_buffer* b) { return b ? b->d_indices : nullptr; }

int32_t crux_buffer_indices(const crux_buffer* cb, int64_t* out, int64_t n) {
  if (!cb || !out) return CRUX_EINVAL;
  crux_buffer* b = const_cast<crux_buffer*>(cb);
  if (b->indices_stale) { crux_ctx* c = b->ctx; b->indices.resize((size_t)b->indices_n);
    HIPCHK(c, hipMemcpyAsync(b->indices.data(), b->d_indices, 8 * (size_t)b->indices_n, hipMemcpyDeviceToHost, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream));
    b->indices_stale = false; }
  if (n > (int64_t)b->indices.size()) n = (int64_t)b->indices.size();
  memcpy(out, b->indices.data(), 8 * (size_t)n); return CRUX_OK;
}

int32_t crux_per_update(crux_buffer* b, const int64_t* I, const void* v, int32_t v_is_f64, int64_t n) {   // :290-301
  if (!b || !I || !v || n < 0) return CRUX_EINVAL;
  crux_ctx* c = b->ctx;
  if (!b->prioritized) return crux_fail(c, CRUX_EINVAL, "update_priorities!: buffer is not prioritized");
  if (n == 0) return CRUX_OK;
  for (int64_t j = 0; j < n; ++j) if (I[j] < 0 || I[j] >= b->capacity) return crux_fail(c, CRUX_EINVAL, "update_priorities!: index %lld out of range", (long long)I[j]);
  const size_t ib = ((8 * (size_t)n + 255) / 256) * 256, vb = (v_is_f64 ? 8 : 4) * (size_t)n;
  char* sc = (char*)crux_scratch(c, ib + vb + 256);
  if (!sc) return crux_fail(c, CRUX_ENOMEM, "update_priorities!: scratch");
  HIPCHK(c, hipMemcpyAsync(sc, I, 8 * (size_t)n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(sc + ib, v, vb, hipMemcpyHostToDevice, c->stream));
  // duplicate indices: the reference loop lets the last write win; values for duplicated rows are equal in every call site
  CRUX_RUN(c, PerUpdateOp, OP_PER_UPDATE, k_per_update, grid_for(n), 256, c->stream, b->priorities, b->pminmax, (const int64_t*)sc, v_is_f64 ? (const double*)(sc + ib) : (const double*)nullptr, v_is_f64 ? (const float*)nullptr : (const float*)(sc + ib), (const float*)nullptr, b->alpha, n);
  int32_t rc = crux_launch_check(c, "k_per_update"); if (rc) return rc;
  { bool in_tree = true; for (int64_t j = 0; j < n; ++j) in_tree = in_tree && I[j] < b->elements;      // rows beyond length(b) are outside the current tree
    if (!in_tree) { b->per_full_dirty = true; b->cumsum_valid = false; } else { rc = crux_per_touched(b, (const int64_t*)sc, n, false); if (rc) return rc; } }                     // :299 cumsum_valid = false; only the touched leaves are re-summed
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return CRUX_OK;
}

int32_t crux_per_update_device(crux_buffer* b, const int64_t* d_ids, const float* d_v, int64_t n) {
  if (!b || !d_ids || !d_v || n < 0) return CRUX_EINVAL;
  if (!b->prioritized) return crux_fail(b->ctx, CRUX_EINVAL, "update_priorities!: buffer is not prioritized");
  if (n == 0) return CRUX_OK;
  CRUX_RUN(b->ctx, PerUpdateOp, OP_PER_UPDATE, k_per_update, grid_for(n), 256, b->ctx->stream, b->priorities, b->pminmax, d_ids, (const double*)nullptr, d_v, (const float*)nullptr, b->alpha, n);
  { const int32_t rc = crux_launch_check(b->ctx, "k_per_update(device)"); if (rc) return rc; }
  return crux_per_touched(b, d_ids, n, false);
}

}  // extern "C"
