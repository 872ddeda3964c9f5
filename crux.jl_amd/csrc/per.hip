// per.hip -- replay sampling on device: prioritized_sample! / uniform_sample! / rand! pieces.
// Reference: src/experience_buffer.jl:303-349. Compiled with -ffp-contract=off (the scan is order-defined).
//
// prioritized_sample! recomputes `cumsum(priorities[1:N])` whenever priorities changed (:329-332), i.e. once per gradient
// step in DQN. Julia's Base.cumsum on a Float32 vector is accumulate_pairwise!: a binary recursion that
// halves segments down to leaves of < 128 elements, sums leaves sequentially and combines child totals in a fixed tree.
// Sample indices are bit-exact only if that summation tree is reproduced, so the scan here is that tree, parallelised:
//   k_leaf_totals   one thread per leaf: sequential Float32 sum of the leaf            (streams N floats)
//   k_tree          one block: child totals bottom-up, then prefixes top-down          (2 x leaves nodes)
//   k_leaf_scan     one thread per leaf: c[i] = prefix + running leaf sum              (streams N floats in, N out)
// followed by k_per_search (Float64 key against the Float32 cumsum, :335-340), the IS weights (:343-347) and the row gather.
// The tree shape depends only on N and is built on the host once per buffer length.
//
// Incremental form (the reference rescans all N priorities per gradient step, :329-332; its abandoned sum-tree is at :48,:334-336): the cumsum is
// never materialised. `cumsum[i]` holds the running sum INSIDE i's leaf and c[i] = prefix(leaf(i)) + cumsum[i] is formed when the search probes
// it -- the same Float32 additions in the same order as accumulate_pairwise! performs (c[i] = op(s, s_)). update_priorities! re-sums only the
// leaves it touched (k_leaf_refresh: <= 127 elements each) and the node totals on their root paths (k_tree_touch); prefix(leaf) = v[1] + the totals of the
// left siblings passed on the way down, added top-down, is formed per probe from those totals. searchsortedfirst probes exactly the elements the reference's
// binary search would. Per sampled step at N = 1 M: <= 128 x 127 x 8 B of leaf traffic + a few KB of node totals and probes instead of 8 MB.
#include "common.h"
#include "exec.h"
#include "ops_small.h"
#include "per_tree.h"
#include <algorithm>

int32_t crux_buffer_ring_indices(crux_buffer* b, int64_t N, std::vector<int64_t>& I);
void crux_buffer_ring_advance(crux_buffer* b, int64_t N);
int32_t crux_buffer_per_on_push(crux_buffer* b, const int64_t* d_I, int64_t N);

// ---- host: shape of accumulate_pairwise!(add_sum, c, v[2:n]) ----------------------------------------------------------
// Nodes carry HEAP numbers (root 1, children 2k and 2k + 1, level l in [2^l, 2^(l+1))). The recursion halves n -> (n >> 1, n - (n >> 1)) until n < 128, so sizes
// inside a level differ by at most one and leaves sit on the last two levels only: the numbering is sparse on the last level (holes have no links and are never
// referenced). What it buys: the leaf of an element, its root path and the left siblings along it follow from N by integer arithmetic (leaf_locate below), so
// the per-step kernels (search, leaf refresh, root paths) read no topology table at all -- each table lookup was a dependent ~1 us round trip to L2 / HBM.
struct TopoBuild { std::vector<int32_t> id, level, start, len; std::vector<char> leaf; };
static void topo_rec(TopoBuild& t, int64_t i1, int64_t n, int lvl, int32_t id) {
  t.id.push_back(id); t.level.push_back(lvl); t.start.push_back((int32_t)i1); t.len.push_back((int32_t)n); t.leaf.push_back(n < 128);
  if (n >= 128) { const int64_t n2 = n >> 1; topo_rec(t, i1, n2, lvl + 1, 2 * id); topo_rec(t, i1 + n2, n - n2, lvl + 1, 2 * id + 1); }
}

static void topo_free(crux_buffer* b) {
  int32_t** ps[] = {&b->topo_leaf_start, &b->topo_leaf_len, &b->topo_leaf_node, &b->topo_left, &b->topo_right, &b->topo_level_off};
  for (auto p : ps) if (*p) { (void)hipFree(*p); *p = nullptr; }
  if (b->topo_total) { (void)hipFree(b->topo_total); b->topo_total = nullptr; }
  if (b->topo_prefix) { (void)hipFree(b->topo_prefix); b->topo_prefix = nullptr; }
  b->topo_n = -1; b->per_run_n = -1; b->per_full_dirty = true;
}
void crux_buffer_topo_free(crux_buffer* b) { topo_free(b); }

static int32_t topo_ensure(crux_buffer* b, int64_t N) {
  crux_ctx* c = b->ctx;
  if (b->topo_n == N) return CRUX_OK;
  (void)hipStreamSynchronize(c->stream);
  topo_free(b);
  if (N < 2) { b->topo_n = N; b->topo_leaves = b->topo_nodes = b->topo_levels = 0; return CRUX_OK; }
  TopoBuild t; topo_rec(t, 1, N - 1, 0, 1);
  int maxlvl = 0; for (int v : t.level) if (v > maxlvl) maxlvl = v;
  if (maxlvl > 28) return crux_fail(c, CRUX_EINVAL, "cumsum tree: %lld elements are more than the heap numbering holds", (long long)N);
  const int nn = 1 << (maxlvl + 1);                       // heap slots 0 .. 2^(levels) - 1 (slot 0 unused)
  std::vector<int32_t> L(nn, -1), R(nn, -1), lvl_off(maxlvl + 2), ls, ll, ln;
  for (int l = 0; l <= maxlvl + 1; ++l) lvl_off[l] = 1 << l;
  for (size_t i = 0; i < t.id.size(); ++i) { const int k = t.id[i];
    if (t.leaf[i]) { ls.push_back(t.start[i]); ll.push_back(t.len[i]); ln.push_back(k); } else { L[k] = 2 * k; R[k] = 2 * k + 1; } }
  const int nl = (int)ls.size();            // the pre-order walk meets the leaves in memory order: a block of LEAF_BLK consecutive leaves covers one contiguous span of the vector
  auto up = [&](int32_t** d, const std::vector<int32_t>& h) -> bool {
    if (hipMalloc(d, 4 * h.size()) != hipSuccess) return false;
    return hipMemcpyAsync(*d, h.data(), 4 * h.size(), hipMemcpyHostToDevice, c->stream) == hipSuccess; };
  b->per_full_dirty = true;
  if (!up(&b->topo_leaf_start, ls) || !up(&b->topo_leaf_len, ll) || !up(&b->topo_leaf_node, ln) || !up(&b->topo_left, L) || !up(&b->topo_right, R) || !up(&b->topo_level_off, lvl_off) ||
      hipMalloc(&b->topo_total, 4 * (size_t)nn) != hipSuccess || hipMalloc(&b->topo_prefix, 4 * (size_t)nn) != hipSuccess ||
      hipMemsetAsync(b->topo_total, 0, 4 * (size_t)nn, c->stream) != hipSuccess || hipMemsetAsync(b->topo_prefix, 0, 4 * (size_t)nn, c->stream) != hipSuccess) { topo_free(b); return crux_fail(c, CRUX_ENOMEM, "cumsum tree"); }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  b->topo_n = N; b->topo_leaves = nl; b->topo_nodes = nn; b->topo_levels = maxlvl + 1;
  return CRUX_OK;
}

// ---- kernels ------------------------------------------------------------------------------------------------------------
// 64 consecutive leaves per block: their elements are contiguous in memory, so the block streams them coalesced into LDS and
// each thread then walks its own leaf from LDS (leaf length <= 127).
#define LEAF_BLK 64
#define LEAF_MAX 127
__global__ __launch_bounds__(LEAF_BLK) void k_leaf_totals(const float* __restrict__ v, const int32_t* __restrict__ lstart, const int32_t* __restrict__ llen,
                                                          const int32_t* __restrict__ lnode, int nl, float* __restrict__ total) {
  __shared__ float sm[LEAF_BLK * LEAF_MAX + 64];
  const int l0 = blockIdx.x * LEAF_BLK, t = threadIdx.x, l = l0 + t;
  const int lend = min(l0 + LEAF_BLK, nl) - 1;
  const int base = lstart[l0], cnt = lstart[lend] + llen[lend] - base;
  for (int i = t; i < cnt; i += LEAF_BLK) sm[i] = v[base + i];
  __syncthreads();
  if (l < nl) { const int o = lstart[l] - base, n = llen[l]; float s = sm[o]; for (int i = 1; i < n; ++i) s = s + sm[o + i]; total[lnode[l]] = s; }
}
__global__ __launch_bounds__(1024) void k_tree(const int32_t* __restrict__ left, const int32_t* __restrict__ right, const int32_t* __restrict__ lvl_off, int nlev,
                                               float* __restrict__ total, float* __restrict__ prefix, const float* __restrict__ v) {
  const int t = threadIdx.x;
  for (int lv = nlev - 1; lv >= 0; --lv) {           // s_ = rec(left); s_ += rec(right)
    for (int k = lvl_off[lv] + t; k < lvl_off[lv + 1]; k += 1024) if (left[k] >= 0) total[k] = total[left[k]] + total[right[k]];
    __syncthreads();
  }
  if (t == 0) prefix[1] = v[0];                        // accumulate_pairwise!: s_ = v[1]; rec(c, v, s_, 2, n-1)  (the root is heap slot 1)
  __syncthreads();
  for (int lv = 0; lv < nlev; ++lv) {                 // left gets s, right gets s + s_left
    for (int k = lvl_off[lv] + t; k < lvl_off[lv + 1]; k += 1024) if (left[k] >= 0) { const float s = prefix[k]; prefix[left[k]] = s; prefix[right[k]] = s + total[left[k]]; }
    __syncthreads();
  }
}
// the same two passes with every node's total and prefix held in LDS (2 x 4 B x nodes <= 150 KB, i.e. up to ~2.4 M elements): the 2 x levels
// dependent steps cost an LDS round trip each instead of an L2 round trip (20 us -> ~4 us at N = 1 M); the arithmetic and its order are unchanged
__global__ __launch_bounds__(1024) void k_tree_lds(const int32_t* __restrict__ left, const int32_t* __restrict__ right, const int32_t* __restrict__ lvl_off, int nlev, int nn,
                                                   float* __restrict__ total, float* __restrict__ prefix, const float* __restrict__ v) {
  extern __shared__ float tl[];
  float* tot = tl; float* pre = tl + nn;
  const int t = threadIdx.x;
  // thread t owns nodes t, t + 1024, ... in every pass: their child links are fetched ONCE into registers, so the 2 x levels dependent steps
  // below touch LDS only (with the links read from global memory inside the loops each level cost an L2 round trip: 17.6 us at N = 1 M)
  constexpr int NPT = 19;                                   // 19 x 1024 nodes >= 150 KB / 8 B
  int lft[NPT], rgt[NPT];
#pragma unroll
  for (int i = 0; i < NPT; ++i) { const int k = t + 1024 * i; lft[i] = k < nn ? left[k] : -1; rgt[i] = k < nn ? right[k] : -1; }
  for (int k = t; k < nn; k += 1024) tot[k] = total[k];          // leaf totals (k_leaf_totals); internal entries are overwritten below
  __syncthreads();
  for (int lv = nlev - 1; lv >= 0; --lv) {
    const int lo = lvl_off[lv], hi = lvl_off[lv + 1];
#pragma unroll
    for (int i = 0; i < NPT; ++i) { const int k = t + 1024 * i; if (k >= lo && k < hi && lft[i] >= 0) tot[k] = tot[lft[i]] + tot[rgt[i]]; }
    __syncthreads();
  }
  if (t == 0) pre[1] = v[0];
  __syncthreads();
  for (int lv = 0; lv < nlev; ++lv) {
    const int lo = lvl_off[lv], hi = lvl_off[lv + 1];
#pragma unroll
    for (int i = 0; i < NPT; ++i) { const int k = t + 1024 * i; if (k >= lo && k < hi && lft[i] >= 0) { const float s = pre[k]; pre[lft[i]] = s; pre[rgt[i]] = s + tot[lft[i]]; } }
    __syncthreads();
  }
  for (int k = t; k < nn; k += 1024) { prefix[k] = pre[k]; total[k] = tot[k]; }     // internal totals stay resident: k_tree_touch updates them along touched paths
}
__global__ __launch_bounds__(LEAF_BLK) void k_leaf_scan(const float* __restrict__ v, const int32_t* __restrict__ lstart, const int32_t* __restrict__ llen,
                                                        const int32_t* __restrict__ lnode, int nl, const float* __restrict__ prefix, float* __restrict__ c) {
  __shared__ float sm[LEAF_BLK * LEAF_MAX + 64];
  const int l0 = blockIdx.x * LEAF_BLK, t = threadIdx.x, l = l0 + t;
  const int lend = min(l0 + LEAF_BLK, nl) - 1;
  const int base = lstart[l0], cnt = lstart[lend] + llen[lend] - base;
  for (int i = t; i < cnt; i += LEAF_BLK) sm[i] = v[base + i];
  __syncthreads();
  if (l < nl) { const int o = lstart[l] - base, n = llen[l]; float s_ = sm[o];          // the running sum s_ of _accumulate_pairwise!'s leaf loop; c[i] = op(s, s_) is formed at probe time
    for (int i = 1; i < n; ++i) { s_ = s_ + sm[o + i]; sm[o + i] = s_; } }
  __syncthreads();
  for (int i = t; i < cnt; i += LEAF_BLK) c[base + i] = sm[i];
  if (blockIdx.x == 0 && t == 0) c[0] = v[0];
}
__global__ __launch_bounds__(256) void k_leaf_refresh(const float* __restrict__ v, const int64_t* __restrict__ ids, int64_t n, int64_t N, int nlev, float* __restrict__ run, float* __restrict__ total) { LeafRefreshOp::run(blockIdx.x, gridDim.x, v, ids, n, N, nlev, run, total); }
// LeafRefreshOp and TreeTouchOp in ONE launch (the chained C3 epochs, exec.hip dqn_epoch_tiles): every workgroup re-sums its leaves, publishes them (device-scope release) and takes
// a ticket; the workgroup that draws the last ticket -- all leaves are then visible to it -- walks the root paths. The root paths so leave the launch of the pullback they sit
// beside, and the next epoch's search can follow one launch earlier. `ticket` is a zeroed word; n <= 256 touched elements.
struct LeafTouchOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, const float* __restrict__ v, const int64_t* __restrict__ ids, int64_t n, int64_t N, int nlev,
                                                    float* __restrict__ run, float* __restrict__ total, unsigned* __restrict__ ticket) {
  __shared__ int last_;
  LeafRefreshOp::run(bid_, nb_, v, ids, n, N, nlev, run, total);
  __syncthreads();
  if (threadIdx.x == 0) { __threadfence(); last_ = atomicAdd(ticket, 1u) == nb_ - 1u ? 1 : 0; }
  __syncthreads();
  if (!last_) return;
  __threadfence();
  TreeTouchOp::run(0u, 1u, ids, n, N, nlev, total);
} };
__global__ __launch_bounds__(256) void k_leaf_touch(const float* __restrict__ v, const int64_t* __restrict__ ids, int64_t n, int64_t N, int nlev, float* __restrict__ run, float* __restrict__ total, unsigned* __restrict__ ticket) { LeafTouchOp::run(blockIdx.x, gridDim.x, v, ids, n, N, nlev, run, total, ticket); }
__global__ __launch_bounds__(1024) void k_tree_touch(const int64_t* __restrict__ ids, int64_t n, int64_t N, int nlev, float* __restrict__ total) { TreeTouchOp::run(blockIdx.x, gridDim.x, ids, n, N, nlev, total); }
__global__ __launch_bounds__(1024) void k_push_touch(int64_t* __restrict__ ids, int64_t n, int64_t base, int64_t C, float* pr, float* pminmax, float alpha, int64_t N, int nlev,
                                                     float* run, float* total, int touch) { push_touch_block(ids, n, base, C, pr, pminmax, alpha, N, nlev, run, total, touch); }
// host side of push_touch_block. crux_per_push_plan: is this push of that shape, and is the tree in its incremental state (the conditions of crux_per_touched(from_push = true),
// evaluated before the ring advances)? crux_per_push_done: the host flags after the launch that ran it (k_push_touch, or the rollout kernel that finished with it).
bool crux_per_push_plan(crux_buffer* b, int64_t n, int* touch) {
  if (!b->prioritized || n < 1 || n > 256 || !b->d_indices || crux_exec_recording(b->ctx) || !crux_sw().push_fused) return false;
  *touch = 1;
  if (b->elements < b->capacity) *touch = 0;
  else if (b->per_full_dirty || b->topo_n < 2 || b->per_run_n != b->topo_n || b->topo_n != b->elements || b->topo_levels > CRUX_PER_PMAX) *touch = 0;
  return true;
}
void crux_per_push_done(crux_buffer* b, int touch) { b->cumsum_valid = false; if (!touch) b->per_full_dirty = true; }
// (the one-round-trip form of a handful of rows keeps a root path's sibling totals in registers: its own launch bounds, so that it does not spill under the 1024-thread cap)
__global__ __launch_bounds__(256) void k_push_touch_small(int64_t* __restrict__ ids, int64_t n, int64_t base, int64_t C, float* pr, float* pminmax, float alpha, int64_t N, int nlev,
                                                          float* run, float* total) { push_touch_small(ids, (int)n, base, C, pr, pminmax, alpha, N, nlev, run, total); }
bool crux_per_push_fused(crux_buffer* b, int64_t n, int64_t* d_ids) {
  int touch = 0; if (!d_ids || !crux_per_push_plan(b, n, &touch)) return false;
  if (touch && n <= PUSH_SMALL_MAX && n <= b->capacity)
    hipLaunchKernelGGL(k_push_touch_small, dim3(1), dim3(256), 0, b->ctx->stream, d_ids, n, b->next_ind, b->capacity, b->priorities, b->pminmax, b->alpha, (int64_t)b->topo_n, (int)b->topo_levels, b->cumsum, b->topo_total);
  else
  hipLaunchKernelGGL(k_push_touch, dim3(1), dim3(1024), 0, b->ctx->stream, d_ids, n, b->next_ind, b->capacity, b->priorities, b->pminmax, b->alpha, (int64_t)b->topo_n, (int)b->topo_levels,
                     b->cumsum, b->topo_total, touch);
  crux_per_push_done(b, touch);
  return true;
}

// prefix of a leaf = _accumulate_pairwise!'s s at that leaf: v[1], then + total(left sibling) at every right turn of the root path, top-down. The node of the path
// at level q is id >> (depth - q); it is a right child when odd and its left sibling is the slot before it. Branch-free: a left turn (or a level below the leaf)
// reads slot 0 of the totals, which holds +0 and leaves the positive sum unchanged bit for bit. All loads are independent (one round trip).
__device__ __forceinline__ float leaf_prefix(const float* __restrict__ v, const float* __restrict__ total, const LeafLoc lf, int nlev) {
  float tv[CRUX_PER_PMAX];
#pragma unroll
  for (int q = 1; q <= CRUX_PER_PMAX; ++q) { tv[q - 1] = 0.f;
    if (q < nlev) { const int sh = lf.depth - q; const int node = sh >= 0 ? lf.id >> sh : 0; tv[q - 1] = total[(node & 1) ? node - 1 : 0]; } }
  float s = v[0];
#pragma unroll
  for (int q = 1; q <= CRUX_PER_PMAX; ++q) if (q < nlev) s = s + tv[q - 1];
  return s;
}
// cumsum(priorities) as the reference would hold it, materialised for crux_per_get: c[i] = prefix(leaf(i)) + run[i]
__global__ void k_materialize_cumsum(const float* __restrict__ run, const float* __restrict__ v, const float* __restrict__ total, int64_t N, int nlev, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = i == 0 ? run[0] : leaf_prefix(v, total, leaf_locate(N, i, nlev), nlev) + run[i];
}
__global__ void k_cumsum_tiny(const float* v, int64_t n, float* c) { if (threadIdx.x == 0 && blockIdx.x == 0) { if (n >= 1) c[0] = v[0]; } }

// stratified search + importance weights (:335-347)
// One WAVE per stratum. searchsortedfirst (:340) is a chain of ~log2(N) dependent probes, and each probe of the un-materialised cumsum costs a
// memory round trip (run[mid] and the sibling totals of mid's root path together -- the path itself is arithmetic); evaluated by one thread that is 20 serial
// round trips. The wave evaluates SIX levels of the binary search at once: lane L (1..63, heap order) assumes the outcomes spelled by the bits of L below its
// leading one, derives the (lo, hi) interval that path would have produced and probes its midpoint; a ballot then replays the real search over the 63 answers.
// The probes and comparisons are exactly those of the sequential search, so the result is the reference's index even where rounding makes the
// cumsum locally non-monotone; 20 levels cost 4 round trips instead of 20. Lane 0, idle in that scheme, fetches cumsum[N] (the total the stratum width is
// derived from) in the FIRST round, beside the probes, so the total costs no round trip of its own. The interval bookkeeping is predicated, not branched, and
// the real (lo, hi) live in scalar registers.
// the search of ONE stratum j by the calling wave (all 64 lanes); returns the sampled index (wave-uniform) and, through w_out, cumsum[N] (ptot)
// importance weight of the sampled index (:343-347), lane 0 only; writes ids[j] and the source's :weight entry
__device__ __forceinline__ float per_weight_lane0(const int64_t j, const int lo, const float ptot, const float* __restrict__ pr, const float* __restrict__ pminmax, int64_t N, float beta, int64_t* __restrict__ ids, float* __restrict__ weight) {
  ids[j] = lo;
  const float pmin = pminmax[1] / ptot;
  const float max_w = powf(pmin * (float)N, -beta);
  const float w = powf(((float)N * pr[lo]) / ptot, beta) / max_w;
  weight[lo] = w;
  return w;
}
__device__ __forceinline__ int per_search_wave(const int64_t j, const float* __restrict__ run, const float* __restrict__ total, const float* __restrict__ pr, const float* __restrict__ pminmax, int64_t N, int64_t B, int nlev,
                             const double* __restrict__ rands, uint64_t seed, uint32_t stream, uint64_t ictr, float beta, int64_t* __restrict__ ids, float* __restrict__ weight, float* w_out) {
  const int lane = threadIdx.x & 63;
  const int Ni = (int)N;
  // cumsum[i] exactly as accumulate_pairwise! forms it: the descent to i's leaf and the sibling totals of its root path in ONE pass (leaf_locate + leaf_prefix fused:
  // the left sibling of a right turn is the slot before the child just entered), all loads independent
  auto cs = [&](int i) -> float {
    const float ri = run[i];
    if (i == 0 || Ni < 2) return ri;
    int i1 = 1, n = Ni - 1, id = 1; float tv[CRUX_PER_PMAX];
#pragma unroll
    for (int it = 0; it < CRUX_PER_PMAX; ++it) { tv[it] = 0.f;
      if (it < nlev - 1) { const bool sp = n >= 128; const int n2 = n >> 1; const bool rt = sp && i >= i1 + n2;
        i1 += rt ? n2 : 0; n = sp ? (rt ? n - n2 : n2) : n; id = sp ? 2 * id + (rt ? 1 : 0) : id;
        tv[it] = total[rt ? id - 1 : 0]; } }
    float sacc = pr[0];
#pragma unroll
    for (int it = 0; it < CRUX_PER_PMAX; ++it) if (it < nlev - 1) sacc = sacc + tv[it];
    return sacc + ri; };
  double u;
  if (rands) u = rands[j];
  else { const crux_u32x4 x = crux_philox(seed, ictr * (uint64_t)B + (uint64_t)j, stream, CRUX_RNG_SAMPLE); u = crux_u32x2_to_f64(x.v[0], x.v[1]); }
  float ptot = 0.f; double key = 0.0; bool first = true;
  int lo = 0, hi = Ni;
  // Round 4: SEVEN levels per round -- every lane carries two nodes of the speculation tree (heap numbers lane and lane + 64; node 1 is the root) -- so that the 20 levels of
  // a search over 1 M elements take three dependent memory round trips (7 + 7 + 6) instead of four (6 + 6 + 6 + 2); the probes and comparisons are still exactly those of
  // the sequential search.
  while (lo < hi) {
    float cvs[2]; bool valids[2];
#pragma unroll
    for (int pnode = 0; pnode < 2; ++pnode) {
      const int nd = lane + 64 * pnode; const int dl = nd ? 31 - __builtin_clz((unsigned)nd) : 0;          // node 1 is the root (depth 0)
      int l = lo, h = hi; bool valid = nd != 0;
#pragma unroll
      for (int b = 5; b >= 0; --b) { const bool act = valid && b < dl, ok = l < h; const int m = l + ((h - l) >> 1); const bool bit = ((nd >> b) & 1) != 0;
        valid = valid && (!act || ok); const bool upd = act && ok; l = (upd && bit) ? m + 1 : l; h = (upd && !bit) ? m : h; }
      valid = valid && l < h;
      const bool want_tot = first && nd == 0;
      float cv = 0.f;
      if (valid || want_tot) cv = cs(want_tot ? Ni - 1 : l + ((h - l) >> 1));
      cvs[pnode] = cv; valids[pnode] = valid; }
    if (first) { ptot = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, cvs[0])));
      const float dp = ptot / (float)B; key = ((double)(j + 1) + u - 1.0) * (double)dp; first = false; }
    const unsigned long long lt0 = __ballot(valids[0] && (double)cvs[0] < key), lt1 = __ballot(valids[1] && (double)cvs[1] < key);
    int node = 1;
#pragma unroll
    for (int step = 0; step < 7; ++step) { if (!(lo < hi)) break; const int mid = lo + ((hi - lo) >> 1);
      const int bit = (int)(((node < 64 ? lt0 >> node : lt1 >> (node - 64))) & 1ull); if (bit) lo = mid + 1; else hi = mid; node = 2 * node + bit; }
    lo = __builtin_amdgcn_readfirstlane(lo); hi = __builtin_amdgcn_readfirstlane(hi);
  }
  if (lo >= Ni) lo = Ni - 1;       // the reference would index out of bounds here
  *w_out = ptot;      // (the total the stratum width came from: the caller forms the importance weight, per_weight_lane0 below)
  return lo;
}
struct PerSearchOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, const float* __restrict__ run, const float* __restrict__ total, const float* __restrict__ pr, const float* __restrict__ pminmax, int64_t N, int64_t B, int nlev,
                             const double* __restrict__ rands, uint64_t seed, uint32_t stream, uint64_t ictr, float beta, int64_t* __restrict__ ids, float* __restrict__ weight) {
  const int64_t j = (int64_t)bid_ * 4 + (threadIdx.x >> 6);
  if (j >= B) return;
  float ptot; const int lo = per_search_wave(j, run, total, pr, pminmax, N, B, nlev, rands, seed, stream, ictr, beta, ids, weight, &ptot);
  if ((threadIdx.x & 63) == 0) (void)per_weight_lane0(j, lo, ptot, pr, pminmax, N, beta, ids, weight);
} };
__global__ __launch_bounds__(256) void k_per_search(const float* __restrict__ run, const float* __restrict__ total, const float* __restrict__ pr, const float* __restrict__ pminmax, int64_t N, int64_t B, int nlev,
                             const double* __restrict__ rands, uint64_t seed, uint32_t stream, uint64_t ictr, float beta, int64_t* __restrict__ ids, float* __restrict__ weight) { PerSearchOp::run(blockIdx.x, gridDim.x, run, total, pr, pminmax, N, B, nlev, rands, seed, stream, ictr, beta, ids, weight); }
struct UniformIdsOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, int64_t N, int64_t B, uint64_t seed, uint32_t stream, uint64_t ictr, int64_t* ids) {
  const int64_t j = (int64_t)bid_ * blockDim.x + threadIdx.x;
  if (j >= B) return;
  const crux_u32x4 x = crux_philox(seed, ictr * (uint64_t)B + (uint64_t)j, stream, CRUX_RNG_SAMPLE);
  ids[j] = (int64_t)(((uint64_t)x.v[0] * (uint64_t)N) >> 32);
} };
__global__ void k_uniform_ids(int64_t N, int64_t B, uint64_t seed, uint32_t stream, uint64_t ictr, int64_t* ids) { UniformIdsOp::run(blockIdx.x, gridDim.x, N, B, seed, stream, ictr, ids); }
// gather rows src[ids[j]] into the ring of dst at (base + j) % C
template <typename T>
__global__ void k_gather_ring(T* __restrict__ dst, const T* __restrict__ src, const int64_t* __restrict__ ids, int64_t n, int32_t row_elems, int64_t base, int64_t C) {
  const int64_t total = n * row_elems;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = t / row_elems; const int32_t e = (int32_t)(t - j * row_elems);
    dst[((base + j) % C) * row_elems + e] = src[ids[j] * row_elems + e];
  }
}
// all columns of the sampled rows in ONE launch: a table of (dst, src, elements per row, element size) and the prefix of row widths
struct GatherCols { void* dst[CRUX_NCOLS]; const void* src[CRUX_NCOLS]; int32_t re[CRUX_NCOLS]; int32_t esz[CRUX_NCOLS]; int32_t pre[CRUX_NCOLS + 1]; int32_t n; };
struct GatherRingAllOp {
  // the column table is indexed per element: held in LDS (a by-value struct indexed at run time lives in scratch memory: 8 us for an 11-block gather). run_ptr takes
  // the table WHERE IT LIES (the executor's kernel-argument record): copied from there to LDS without a private copy in between -- passed by value through the
  // executor's argument pack it put 464 bytes of scratch into every phase launch.
  static __device__ __forceinline__ void run_ptr(const unsigned bid_, const unsigned nb_, const GatherCols* gp, const int64_t* __restrict__ ids, int64_t n, int64_t base, int64_t C) {
    __shared__ GatherCols gs;
    { const uint32_t* src = (const uint32_t*)gp; uint32_t* dst = (uint32_t*)&gs; for (int i = threadIdx.x; i < (int)(sizeof(GatherCols) / 4); i += blockDim.x) dst[i] = src[i]; }
    __syncthreads();
    const int32_t width = gs.pre[gs.n]; const int64_t total = n * width; const int ncol = gs.n;
    for (int64_t t = (int64_t)bid_ * blockDim.x + threadIdx.x; t < total; t += (int64_t)nb_ * blockDim.x) {
      const int64_t j = t / width; const int32_t w = (int32_t)(t - j * width);
      int k = 0; while (k + 1 < ncol && w >= gs.pre[k + 1]) ++k;
      const int32_t e = w - gs.pre[k]; const int64_t d = ((base + j) % C) * gs.re[k] + e, sidx = ids[j] * gs.re[k] + e;
      if (gs.esz[k] == 4) ((uint32_t*)gs.dst[k])[d] = ((const uint32_t*)gs.src[k])[sidx];
      else ((uint8_t*)gs.dst[k])[d] = ((const uint8_t*)gs.src[k])[sidx];
    }
  }
  static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, GatherCols g, const int64_t* __restrict__ ids, int64_t n, int64_t base, int64_t C) { run_ptr(bid_, nb_, &g, ids, n, base, C); }
};
__global__ void k_gather_ring_all(GatherCols g, const int64_t* __restrict__ ids, int64_t n, int64_t base, int64_t C) { GatherRingAllOp::run(blockIdx.x, gridDim.x, g, ids, n, base, C); }
// prioritized_sample! of one row per WAVE, search and gather in ONE launch (round 4): the wave that found stratum j's index copies that row into the batch ring right away --
// the sampled ids no longer cross a kernel boundary between `searchsortedfirst` and `push!(target, source, ids)` (experience_buffer.jl:340,348). The row's :weight entry is
// the value the wave has just computed (the source column is written too, as before; the copy takes it from the register instead of reading it back).
struct PerSampleArgs { const float* run; const float* total; const float* pr; const float* pminmax; int64_t N, B; const double* rands; uint64_t seed, ictr; int64_t* ids; float* weight; int64_t base, C;
                       int32_t nlev; uint32_t stream; float beta; int32_t pad; GatherCols g; };
struct PerSampleGatherOp {
  static __device__ __forceinline__ void run_ptr(const unsigned bid_, const unsigned nb_, const PerSampleArgs* a) {
    __shared__ GatherCols gs2;
    // the column table (456 bytes of the kernel arguments) is only needed by the gather: its loads are issued now, the LDS copy and the barrier come AFTER the search, so the
    // table's round trip hides behind the search's three instead of preceding them
    static_assert(sizeof(GatherCols) / 4 <= 256, "one table word per thread");
    const uint32_t tabw = threadIdx.x < sizeof(GatherCols) / 4 ? ((const uint32_t*)&a->g)[threadIdx.x] : 0u;
    const int lane = threadIdx.x & 63; const int64_t j = (int64_t)bid_ * 4 + (threadIdx.x >> 6); const int64_t B = a->B;
    float* const weight = a->weight; float ptot = 0.f; int lo = 0;
    if (j < B) lo = per_search_wave(j, a->run, a->total, a->pr, a->pminmax, a->N, B, a->nlev, a->rands, a->seed, a->stream, a->ictr, a->beta, a->ids, weight, &ptot);
    if (threadIdx.x < sizeof(GatherCols) / 4) ((uint32_t*)&gs2)[threadIdx.x] = tabw;
    __syncthreads();
    if (j >= B) return;
    const int32_t width = gs2.pre[gs2.n]; const int ncol = gs2.n; const int64_t drow = (a->base + j) % a->C;
    // the row's loads go out BEFORE lane 0 forms the importance weight (a dependent load of priorities[lo] and two powf): one memory round trip for both
    uint32_t v0 = 0; int k0 = -1; int64_t d0 = 0; bool byte0 = false;
    if (lane < width) { int k = 0; while (k + 1 < ncol && lane >= gs2.pre[k + 1]) ++k;
      const int32_t e = lane - gs2.pre[k]; d0 = drow * gs2.re[k] + e; const int64_t sidx = (int64_t)lo * gs2.re[k] + e; k0 = k; byte0 = gs2.esz[k] != 4;
      v0 = byte0 ? (uint32_t)((const uint8_t*)gs2.src[k])[sidx] : ((const uint32_t*)gs2.src[k])[sidx]; }
    float w = 0.f; if (lane == 0) w = per_weight_lane0(j, lo, ptot, a->pr, a->pminmax, a->N, a->beta, a->ids, weight);
    w = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, w)));
    if (k0 >= 0) { if (byte0) ((uint8_t*)gs2.dst[k0])[d0] = (uint8_t)v0; else ((uint32_t*)gs2.dst[k0])[d0] = gs2.src[k0] == (const void*)weight ? __builtin_bit_cast(uint32_t, w) : v0; }
    for (int32_t t = lane + 64; t < width; t += 64) {      // (rows wider than 64 elements)
      int k = 0; while (k + 1 < ncol && t >= gs2.pre[k + 1]) ++k;
      const int32_t e = t - gs2.pre[k]; const int64_t d = drow * gs2.re[k] + e, sidx = (int64_t)lo * gs2.re[k] + e;
      if (gs2.esz[k] == 4) { uint32_t v = ((const uint32_t*)gs2.src[k])[sidx]; if (gs2.src[k] == (const void*)weight) v = __builtin_bit_cast(uint32_t, w); ((uint32_t*)gs2.dst[k])[d] = v; }
      else ((uint8_t*)gs2.dst[k])[d] = ((const uint8_t*)gs2.src[k])[sidx];
    }
  }
  static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, PerSampleArgs a) { run_ptr(bid_, nb_, &a); }
};
__global__ __launch_bounds__(256) void k_per_sample_gather(PerSampleArgs a) { PerSampleGatherOp::run_ptr(blockIdx.x, gridDim.x, (const PerSampleArgs*)__builtin_amdgcn_kernarg_segment_ptr()); }      // (the table is read where it lies: no private copy)
struct RingIdsOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, int64_t* out, int64_t n, int64_t base, int64_t C) { const int64_t j = (int64_t)bid_ * blockDim.x + threadIdx.x; if (j < n) out[j] = (base + j) % C; } };
__global__ void k_ring_ids(int64_t* out, int64_t n, int64_t base, int64_t C) { RingIdsOp::run(blockIdx.x, gridDim.x, out, n, base, C); }

// the ring rows of the next N pushed transitions, mod1.(next_ind : next_ind + N - 1, C) (experience_buffer.jl:236), written on the device: no host vector, no copy to wait for
int32_t crux_buffer_ring_ids_device(crux_buffer* b, int64_t N, int64_t* d_out) {
  if (N <= 0) return CRUX_OK;
  hipLaunchKernelGGL(k_ring_ids, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, b->ctx->stream, d_out, N, b->next_ind, b->capacity);
  return crux_launch_check(b->ctx, "k_ring_ids");
}
static unsigned gridn(int64_t total) { int64_t nb = (total + 255) / 256; if (nb < 1) nb = 1; if (nb > 8192) nb = 8192; return (unsigned)nb; }

static int32_t ensure_cumsum(crux_buffer* s, int64_t N) {
  crux_ctx* c = s->ctx;
  if (s->cumsum_valid && s->topo_n == N) return CRUX_OK;
  int32_t rc = topo_ensure(s, N); if (rc) return rc;
  crux_prof_begin(c, CRUX_PROF_PER_SCAN);
  if (N < 2) hipLaunchKernelGGL(k_cumsum_tiny, dim3(1), dim3(1), 0, c->stream, s->priorities, N, s->cumsum);
  else {
    const int nb = (s->topo_leaves + LEAF_BLK - 1) / LEAF_BLK;
    const bool full = s->per_full_dirty || s->per_run_n != N || s->topo_levels > CRUX_PER_PMAX;      // otherwise leaves and root paths were re-summed when they were touched (crux_per_touched)
    if (full) {
      hipLaunchKernelGGL(k_leaf_totals, dim3(nb), dim3(LEAF_BLK), 0, c->stream, s->priorities, s->topo_leaf_start, s->topo_leaf_len, s->topo_leaf_node, s->topo_leaves, s->topo_total);
      const size_t tree_lds = 8 * (size_t)s->topo_nodes;
      if (tree_lds <= 150 * 1024) {
        static bool attr_dev[16] = {}; bool& attr = attr_dev[c->device & 15];      // (per device: a second device in the process sets the attribute for itself)
        if (!attr) { HIPCHK(c, hipFuncSetAttribute((const void*)k_tree_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); attr = true; }
        hipLaunchKernelGGL(k_tree_lds, dim3(1), dim3(1024), tree_lds, c->stream, s->topo_left, s->topo_right, s->topo_level_off, s->topo_levels, s->topo_nodes, s->topo_total, s->topo_prefix, s->priorities);
      } else
      hipLaunchKernelGGL(k_tree, dim3(1), dim3(1024), 0, c->stream, s->topo_left, s->topo_right, s->topo_level_off, s->topo_levels, s->topo_total, s->topo_prefix, s->priorities);
      hipLaunchKernelGGL(k_leaf_scan, dim3(nb), dim3(LEAF_BLK), 0, c->stream, s->priorities, s->topo_leaf_start, s->topo_leaf_len, s->topo_leaf_node, s->topo_leaves, s->topo_prefix, s->cumsum);
    }
    s->per_full_dirty = false; s->per_run_n = N;
  }
  crux_prof_end(c, CRUX_PROF_PER_SCAN);
  s->cumsum_valid = true;
  return crux_launch_check(c, "per scan");
}
int32_t crux_per_prepare(crux_buffer* source) {      // exec.hip: bring the tree up to date BEFORE a recording starts (a full rebuild launches plain kernels)
  if (!source->prioritized || source->elements < 1) return CRUX_OK;
  return ensure_cumsum(source, source->elements);
}
// called by every path that changes priorities (update_priorities!, push!'s max-priority rows): d_ids = the touched elements (device, int64)
int32_t crux_per_touched(crux_buffer* b, const int64_t* d_ids, int64_t n, bool from_push, unsigned* ticket) {      // ticket (a zeroed device word, recordings only): leaves and root paths as ONE op (LeafTouchOp)
  b->cumsum_valid = false;
  if (n <= 0) return CRUX_OK;
  if (from_push && b->elements < b->capacity) { b->per_full_dirty = true; return CRUX_OK; }   // the ring is still growing: the rows may lie beyond the current tree
  if (b->per_full_dirty || b->topo_n < 2 || b->per_run_n != b->topo_n || b->topo_n != b->elements || n > 4096 || !d_ids) { b->per_full_dirty = true; return CRUX_OK; }
  if (b->topo_levels > CRUX_PER_PMAX) { b->per_full_dirty = true; return CRUX_OK; }
  if (ticket && n <= 256 && crux_exec_recording(b->ctx)) { crux_exec_push<LeafTouchOp, OP_LEAF_TOUCH>(b->ctx, (unsigned)((n + 3) / 4), (const float*)b->priorities, d_ids, n, (int64_t)b->topo_n, (int)b->topo_levels, b->cumsum, b->topo_total, ticket); return CRUX_OK; }
  CRUX_RUN(b->ctx, LeafRefreshOp, OP_LEAF_REFRESH, k_leaf_refresh, (unsigned)((n + 3) / 4), 256, b->ctx->stream, b->priorities, d_ids, n, b->topo_n, b->topo_levels, b->cumsum, b->topo_total);
  CRUX_RUN(b->ctx, TreeTouchOp, OP_TREE_TOUCH, k_tree_touch, 1, 1024, b->ctx->stream, d_ids, n, b->topo_n, b->topo_levels, b->topo_total);
  return crux_launch_check(b->ctx, "k_leaf_refresh");
}

// push!(target, source, ids=device ids) (:232-259): gather B rows into target's ring; target.indices mirrors the ids
bool crux_per_fused_gather() { return crux_sw().per_fused_gather; }      // 0: search and gather as two launches (tests compare)
static void gather_table(crux_buffer* target, crux_buffer* source, GatherCols& g) {
  g = GatherCols{}; g.n = 0; g.pre[0] = 0;
  for (int k = 0; k < CRUX_NCOLS; ++k) {
    if (!has_col(target, k) || !has_col(source, k)) continue;
    const size_t st = col_stride(target, k); const int q = g.n++;
    g.dst[q] = target->col[k]; g.src[q] = source->col[k]; g.esz[q] = st % 4 == 0 ? 4 : 1; g.re[q] = (int32_t)(st % 4 == 0 ? st / 4 : st); g.pre[q + 1] = g.pre[q] + g.re[q];
  }
}
static int32_t gather_into(crux_buffer* target, crux_buffer* source, int64_t B, bool fetch_indices, bool rows_done = false) {
  crux_ctx* c = target->ctx;
  const int64_t base = target->next_ind, C = target->capacity;
  crux_prof_begin(c, CRUX_PROF_GATHER);
  GatherCols g; gather_table(target, source, g);
  if (g.n > 0 && !rows_done) CRUX_RUN(c, GatherRingAllOp, OP_GATHER_RING_ALL, k_gather_ring_all, gridn(B * g.pre[g.n]), 256, c->stream, g, (const int64_t*)target->d_indices, B, base, C);
  crux_prof_end(c, CRUX_PROF_GATHER);
  int32_t rc = crux_launch_check(c, "k_gather_ring"); if (rc) return rc;
  if (target->prioritized) {       // buffer_like of a prioritized buffer is prioritized too (:84): push! runs update_priorities! on it
    int64_t* ring = (int64_t*)crux_scratch(c, 8 * (size_t)B + 256); if (!ring) return crux_fail(c, CRUX_ENOMEM, "sample: scratch");
    CRUX_RUN(c, RingIdsOp, OP_RING_IDS, k_ring_ids, gridn(B), 256, c->stream, ring, B, base, C);
    rc = crux_buffer_per_on_push(target, ring, B); if (rc) return rc;
  }
  if (fetch_indices) { target->indices_n = B; target->indices_stale = true; }   // crux_buffer_indices copies them out when (if) the host asks: no synchronisation per sample
  crux_buffer_ring_advance(target, B);
  return CRUX_OK;
}

extern "C" {

int32_t crux_per_sample(crux_buffer* target, crux_buffer* source, int64_t B, const double* rands, float beta, uint64_t i) {
  if (!target || !source) return CRUX_EINVAL;
  crux_ctx* c = target->ctx;
  if (!source->prioritized || !has_col(source, CRUX_COL_WEIGHT)) return crux_fail(c, CRUX_EINVAL, "prioritized_sample!: source needs priorities and a :weight column (@assert haskey(source, :weight))");
  const int64_t N = source->elements;
  if (N <= 0 || B <= 0 || B > target->capacity) return crux_fail(c, CRUX_EINVAL, "prioritized_sample!: N=%lld B=%lld capacity=%lld", (long long)N, (long long)B, (long long)target->capacity);
  if (target->obs_dim != source->obs_dim || target->act_dim != source->act_dim || target->act_kind != source->act_kind) return crux_fail(c, CRUX_EINVAL, "prioritized_sample!: column shapes differ");
  int32_t rc = ensure_cumsum(source, N); if (rc) return rc;
  double* d_r = nullptr;
  if (rands) { d_r = (double*)crux_scratch(c, 8 * (size_t)B + 256); if (!d_r) return crux_fail(c, CRUX_ENOMEM, "prioritized_sample!: scratch");
    HIPCHK(c, hipMemcpyAsync(d_r, rands, 8 * (size_t)B, hipMemcpyHostToDevice, c->stream)); }
  const bool fuse_gather = crux_per_fused_gather() && !c->per_split_sample;
  crux_prof_begin(c, CRUX_PROF_PER_SEARCH);
  if (fuse_gather) {
    PerSampleArgs a{}; a.run = source->cumsum; a.total = source->topo_total; a.pr = source->priorities; a.pminmax = source->pminmax; a.N = N; a.B = B; a.nlev = source->topo_levels; a.rands = d_r;
    a.seed = source->sample_seed; a.stream = source->sample_stream; a.ictr = i; a.beta = beta; a.ids = target->d_indices; a.weight = (float*)source->col[CRUX_COL_WEIGHT]; a.base = target->next_ind; a.C = target->capacity;
    gather_table(target, source, a.g);
    CRUX_RUN(c, PerSampleGatherOp, OP_PER_SAMPLE, k_per_sample_gather, (unsigned)((B + 3) / 4), 256, c->stream, a);
    crux_prof_end(c, CRUX_PROF_PER_SEARCH);
    rc = crux_launch_check(c, "k_per_sample_gather"); if (rc) return rc;
    return gather_into(target, source, B, true, /*rows_done=*/true);
  }
  CRUX_RUN(c, PerSearchOp, OP_PER_SEARCH, k_per_search, (unsigned)((B + 3) / 4), 256, c->stream, source->cumsum, source->topo_total, source->priorities, source->pminmax, N, B, source->topo_levels, (const double*)d_r, source->sample_seed, source->sample_stream, i, beta, target->d_indices, (float*)source->col[CRUX_COL_WEIGHT]);
  crux_prof_end(c, CRUX_PROF_PER_SEARCH);
  rc = crux_launch_check(c, "k_per_search"); if (rc) return rc;
  return gather_into(target, source, B, true);
}

int32_t crux_uniform_sample(crux_buffer* target, crux_buffer* source, int64_t B, const int64_t* ids, uint64_t i) {
  if (!target || !source) return CRUX_EINVAL;
  crux_ctx* c = target->ctx;
  const int64_t N = source->elements;
  if (N <= 0 || B <= 0 || B > target->capacity) return crux_fail(c, CRUX_EINVAL, "uniform_sample!: N=%lld B=%lld capacity=%lld", (long long)N, (long long)B, (long long)target->capacity);
  if (target->obs_dim != source->obs_dim || target->act_dim != source->act_dim || target->act_kind != source->act_kind) return crux_fail(c, CRUX_EINVAL, "uniform_sample!: column shapes differ");
  if (ids) { for (int64_t j = 0; j < B; ++j) if (ids[j] < 0 || ids[j] >= N) return crux_fail(c, CRUX_EINVAL, "uniform_sample!: id %lld out of range", (long long)ids[j]);
    HIPCHK(c, hipMemcpyAsync(target->d_indices, ids, 8 * (size_t)B, hipMemcpyHostToDevice, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream)); }
  else CRUX_RUN(c, UniformIdsOp, OP_UNIFORM_IDS, k_uniform_ids, gridn(B), 256, c->stream, N, B, source->sample_seed, source->sample_stream, i, target->d_indices);
  return gather_into(target, source, B, true);
}

// Philox key and stream of the sampling draws taken FROM this buffer (rand!, experience_buffer.jl:303-315 draws each source independently)
int32_t crux_buffer_set_sample_stream(crux_buffer* source, uint64_t seed, uint32_t stream) {
  if (!source) return CRUX_EINVAL;
  source->sample_seed = seed; source->sample_stream = stream; return CRUX_OK;
}

int32_t crux_per_get(crux_buffer* b, float* priorities, float* max_priority, float* min_priority, float* cumsum) {
  if (!b) return CRUX_EINVAL;
  crux_ctx* c = b->ctx;
  if (!b->prioritized) return crux_fail(c, CRUX_EINVAL, "buffer is not prioritized");
  if (cumsum && b->elements > 0) { int32_t rc = ensure_cumsum(b, b->elements); if (rc) return rc;
    const int64_t N = b->elements; float* tmp = (float*)crux_scratch(c, 4 * (size_t)N + 256); if (!tmp) return crux_fail(c, CRUX_ENOMEM, "per_get: scratch");
    if (N < 2) HIPCHK(c, hipMemcpyAsync(tmp, b->cumsum, 4 * (size_t)N, hipMemcpyDeviceToDevice, c->stream));
    else hipLaunchKernelGGL(k_materialize_cumsum, dim3(gridn(N)), dim3(256), 0, c->stream, b->cumsum, b->priorities, b->topo_total, N, b->topo_levels, tmp);
    HIPCHK(c, hipMemcpyAsync(cumsum, tmp, 4 * (size_t)N, hipMemcpyDeviceToHost, c->stream)); }
  float mm[2];
  HIPCHK(c, hipMemcpyAsync(mm, b->pminmax, 8, hipMemcpyDeviceToHost, c->stream));
  if (priorities) HIPCHK(c, hipMemcpyAsync(priorities, b->priorities, 4 * (size_t)b->capacity, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (max_priority) *max_priority = mm[0]; if (min_priority) *min_priority = mm[1];
  return CRUX_OK;
}

}  // extern "C"
