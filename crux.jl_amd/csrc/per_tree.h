// per_tree.h -- device bodies of the prioritized buffer's incremental pairwise-cumsum tree that more than one translation unit runs (per.hip: the stand-alone kernels and the
// executor's ops; env.hip: the off-policy rollout kernel, which finishes push!'s bookkeeping in its own launch). The tree itself is described at the top of per.hip.
#pragma once
#include "common.h"
#include "ops_small.h"

// device twin of topo_rec's descent: the leaf holding element e (1 <= e < N): heap number, level, first element and length. `nlev` = levels of the tree
// (uniform): the loop runs nlev - 1 times for every lane with predicated updates -- branch-free (a data-dependent `while` costs an exec-mask branch per level).
struct LeafLoc { int id, depth, start, len; };
__device__ __forceinline__ LeafLoc leaf_locate(int64_t N, int64_t e, int nlev) {
  int i1 = 1, n = (int)(N - 1), id = 1, d = 0; const int ee = (int)e;
  for (int it = 0; it < nlev - 1; ++it) { const bool sp = n >= 128; const int n2 = n >> 1; const bool rt = sp && ee >= i1 + n2;
    i1 += rt ? n2 : 0; n = sp ? (rt ? n - n2 : n2) : n; id = sp ? 2 * id + (rt ? 1 : 0) : id; d += sp ? 1 : 0; }
  return LeafLoc{id, d, i1, n};
}

// update_priorities! touched element ids[j]: re-sum its leaf (running sums + total). One wave per touched element; duplicates write identical values.
struct LeafRefreshOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, const float* __restrict__ v, const int64_t* __restrict__ ids, int64_t n, int64_t N, int nlev,
                                                      float* __restrict__ run, float* __restrict__ total) {
  // one WAVE per touched element (4 per 256-thread block): lane l holds v[o + l] and v[o + 64 + l]; the running sum s_ = s_ + v[i] is inherently serial, so
  // it walks the lanes with v_readlane (constant lane numbers, fully unrolled: readlane + add + select per element, no branch) and lane i keeps the i-th
  // running sum. Lanes past the leaf's end hold +0, which leaves the (positive) sum unchanged bit for bit. The leaf follows from the element number by
  // arithmetic on wave-uniform values: ids -> v is the only dependent pair of memory round trips.
  const int lane = threadIdx.x & 63; const int64_t q = (int64_t)bid_ * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (q >= n) return;
  const int e = __builtin_amdgcn_readfirstlane((int)ids[q]);
  if (e == 0) { if (lane == 0) run[0] = v[0]; return; }       // element 1 of the reference is the seed s_ = v[1], outside the tree
  const LeafLoc lf = leaf_locate(N, e, nlev); const int node = lf.id, o = lf.start, len = lf.len;
  const float x0 = lane < len ? v[o + lane] : 0.f, x1 = 64 + lane < len ? v[o + 64 + lane] : 0.f;
  float s_ = 0.f, r0 = 0.f, r1 = 0.f;
#pragma unroll
  for (int i = 0; i < 64; ++i) { const float xi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x0), i)); s_ = i == 0 ? xi : s_ + xi; r0 = lane == i ? s_ : r0; }
  if (len > 64) {
#pragma unroll
    for (int i = 0; i < 64; ++i) { const float xi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x1), i)); s_ = s_ + xi; r1 = lane == i ? s_ : r1; }
  }
  if (lane < len) run[o + lane] = r0;
  if (64 + lane < len) run[o + 64 + lane] = r1;
  if (lane == 0) total[node] = s_;
} };
// After k_leaf_refresh: node totals along the touched leaves' root paths, bottom-up level by level (s_ = rec(left); s_ += rec(right)). One workgroup;
// thread q follows touched element q. Nodes shared by several paths are written by several threads with the same value. The ancestor of leaf L (level d) at
// level lv is L >> (d - lv) and its children are 2a and 2a + 1: each level costs ONE round trip (the two child totals), nothing is looked up.
struct TreeTouchOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, const int64_t* __restrict__ ids, int64_t n, int64_t N, int nlev, float* __restrict__ total) {
  if (n <= (int64_t)blockDim.x) {                   // the usual case (a minibatch of touched elements): the descent is done once, before the level loop
    const int64_t e = (int64_t)threadIdx.x < n ? ids[threadIdx.x] : 0;
    const LeafLoc lf = leaf_locate(N, e > 0 ? e : 1, nlev);
    for (int lv = nlev - 2; lv >= 0; --lv) {
      if (e != 0 && lf.depth - 1 - lv >= 0) { const int a = lf.id >> (lf.depth - lv); total[a] = total[2 * a] + total[2 * a + 1]; }
      __threadfence_block(); __syncthreads();
    }
    return;
  }
  for (int lv = nlev - 2; lv >= 0; --lv) {          // parents at level lv are complete once the level below is
    for (int64_t q = threadIdx.x; q < n; q += blockDim.x) { const int64_t e = ids[q]; if (e == 0) continue;
      const LeafLoc lf = leaf_locate(N, e, nlev);
      if (lf.depth - 1 - lv >= 0) { const int a = lf.id >> (lf.depth - lv); total[a] = total[2 * a] + total[2 * a + 1]; } }
    __threadfence_block(); __syncthreads();
  }
} };
// push!'s priority bookkeeping for n <= 256 freshly written ring rows as ONE launch of one workgroup (an off-policy solve pushes dN = 4..50 rows per iteration; as separate launches --
// ring rows, max-priority snapshot, update_priorities!, leaf re-sum, root paths -- it was five kernel boundaries of ~5 us for a few hundred bytes of work): ids[j] = (base + j) % C
// (experience_buffer.jl:236), priorities[ids] = (max_priority + eps)^alpha with max_priority read once before (:254, :290-301), and -- touch != 0: the tree is in its incremental
// state (crux_per_touched) -- the touched leaves' running sums and the totals along their root paths. The bodies are the stand-alone kernels' own, run back to back in one
// compute unit (write-through L1: fence + workgroup barrier order them).
// The same for a handful of rows (n <= 32: the dN rows of an off-policy iteration) in ONE memory round trip. Run back to back, the bodies above are a chain of dependent
// round trips -- max_priority, the stored priorities re-read by the leaf re-sum, then one per level of the root paths (14 at 1 M rows): ~10 us for four rows. Here every load
// goes out at once: max_priority, the touched leaves' elements (the rows being written are patched in registers: they are the ring rows base .. base + n - 1) and, per touched
// element, the totals of the SIBLINGS along its root path. The paths are then summed bottom-up out of registers; where two touched paths meet, the sibling's fresh value is
// taken from the other path through LDS. Operands and order of every addition are those of LeafRefreshOp / TreeTouchOp: the same bits.
#define PUSH_SMALL_MAX 32
__device__ __forceinline__ void pts_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }      // orders LDS traffic only: __syncthreads() would also wait for the acknowledgement of the global stores before it
__device__ __forceinline__ void push_touch_small(int64_t* __restrict__ ids, const int n, const int64_t base, const int64_t C, float* pr, float* pminmax, const float alpha, const int64_t N,
                                                 const int nlev, float* run, float* total) {
  __shared__ float pts_val[PUSH_SMALL_MAX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
#ifdef CRUX_RES_TIMING
  long long pt_[6]; int pti_ = 0;
#define PTS_T() pt_[pti_++] = wall_clock64();
#else
#define PTS_T()
#endif
  PTS_T()
  const float pmax = pminmax[0];
  // ---- thread q < n (all in wave 0): element q's leaf and the sibling totals of its root path (independent loads, issued before anything is waited for)
  const bool mine = tid < n;
  const int64_t eq = mine ? (base + tid) % C : 0;
  const bool in_tree = mine && eq != 0;
  const LeafLoc lq = leaf_locate(N, in_tree ? eq : 1, nlev);
  float sib[CRUX_PER_PMAX];
#pragma unroll
  for (int lv = 0; lv < CRUX_PER_PMAX; ++lv) { sib[lv] = 0.f;
    if (in_tree && lv < nlev - 1 && lq.depth - 1 - lv >= 0) { const int c = lq.id >> (lq.depth - lv - 1); sib[lv] = total[c ^ 1]; } }
  // ---- one wave per touched element: its leaf, with the rows being pushed patched in. The loads of the wave's first element go out before max_priority is waited for.
  auto pushed = [&](const int idx) -> bool { int64_t d = (int64_t)idx - base; d += d < 0 ? C : 0; return d < (int64_t)n; };
  int e = wave < n ? __builtin_amdgcn_readfirstlane((int)((base + wave) % C)) : 0;
  LeafLoc lf = leaf_locate(N, e > 0 ? e : 1, nlev);
  float x0 = (e > 0 && lane < lf.len) ? pr[lf.start + lane] : 0.f, x1 = (e > 0 && 64 + lane < lf.len) ? pr[lf.start + 64 + lane] : 0.f;
  // (pinned here: left to itself the compiler sinks the sibling loads to their first use, behind the leaf work -- a second, exposed round trip)
#pragma unroll
  for (int lv = 0; lv < CRUX_PER_PMAX; ++lv) asm volatile("" :: "v"(sib[lv]));
  asm volatile("" :: "v"(x0), "v"(x1), "v"(pmax));
  const double val = (double)pmax + (double)1.1920928955078125e-07f;      // max_priority*ones(N) .+ eps(Float32) (Float64, :254 / :292)
  PTS_T()
  const float pnew = (float)pow(val, (double)alpha), vf32 = (float)val;
  PTS_T()
  for (int q = wave; q < n; q += nw) {
    if (q != wave) { e = __builtin_amdgcn_readfirstlane((int)((base + q) % C)); lf = leaf_locate(N, e > 0 ? e : 1, nlev);
      x0 = (e > 0 && lane < lf.len) ? pr[lf.start + lane] : 0.f; x1 = (e > 0 && 64 + lane < lf.len) ? pr[lf.start + 64 + lane] : 0.f; }
    if (e == 0) { if (lane == 0) run[0] = pnew; continue; }       // element 1 of the reference is the seed s_ = v[1], outside the tree
    const int o = lf.start, len = lf.len;
    if (lane < len && pushed(o + lane)) x0 = pnew;
    if (64 + lane < len && pushed(o + 64 + lane)) x1 = pnew;
    float s_ = 0.f, r0 = 0.f, r1 = 0.f;
#pragma unroll
    for (int i = 0; i < 64; ++i) { const float xi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x0), i)); s_ = i == 0 ? xi : s_ + xi; r0 = lane == i ? s_ : r0; }
    if (len > 64) {
#pragma unroll
      for (int i = 0; i < 64; ++i) { const float xi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x1), i)); s_ = s_ + xi; r1 = lane == i ? s_ : r1; }
    }
    if (lane < len) run[o + lane] = r0;
    if (64 + lane < len) run[o + 64 + lane] = r1;
    if (lane == 0) { total[lf.id] = s_; pts_val[q] = s_; }
  }
  PTS_T()
  // ---- priorities[I] = val^alpha, the ring rows, max / min (PerUpdateOp)
  if (mine) { ids[tid] = eq; pr[eq] = pnew; }
  if (tid == 0) { pminmax[2] = pmax; atomicMax((int*)&pminmax[0], __float_as_int(vf32)); atomicMin((int*)&pminmax[1], __float_as_int(vf32)); }
  pts_lds_barrier();
  PTS_T()
  // ---- root paths, bottom-up, by wave 0 alone (n <= 32 lanes; no barrier): lane q carries the node of its path and its total; a sibling that lies on another touched path is
  // found by walking the other lanes' nodes with v_readlane
  if (wave == 0) {
    float cur = in_tree ? pts_val[lane < PUSH_SMALL_MAX ? lane : 0] : 0.f; int node = in_tree ? lq.id : -1;
#pragma unroll
    for (int lv = CRUX_PER_PMAX - 1; lv >= 0; --lv) {
      if (lv > nlev - 2) continue;                                       // (uniform)
      const bool act = in_tree && lq.depth - 1 - lv >= 0;
      const int c = act ? lq.id >> (lq.depth - lv - 1) : -2, sb = c ^ 1;
      float sv = sib[lv];
      for (int k = 0; k < n; ++k) {                                          // (uniform trip count; lane select in a scalar register)
        const int nk = __builtin_amdgcn_readlane(node, k); const float vk = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cur), k));
        sv = nk == sb ? vk : sv; }                                           // the sibling lies on another touched path: its fresh total
      const float nv = (c & 1) ? sv + cur : cur + sv;                        // total[a] = total[2a] + total[2a + 1]
      if (act) { total[c >> 1] = nv; node = c >> 1; cur = nv; }
    }
  }
  PTS_T()
#ifdef CRUX_RES_TIMING
  if (tid == 0) printf("[pts-timing] pmax %lld pow %lld leaf %lld upd %lld tree %lld\n", pt_[1] - pt_[0], pt_[2] - pt_[1], pt_[3] - pt_[2], pt_[4] - pt_[3], pt_[5] - pt_[4]);
#endif
#undef PTS_T
}
__device__ __forceinline__ void push_touch_block(int64_t* __restrict__ ids, int64_t n, int64_t base, int64_t C, float* pr, float* pminmax, float alpha, int64_t N, int nlev,
                                                 float* run, float* total, int touch) {
  // (a handful of rows with the tree in its incremental state take push_touch_small: the rollout kernel calls it directly, crux_per_push_fused launches k_push_touch_small)
  if (threadIdx.x == 0) pminmax[2] = pminmax[0];
  if ((int64_t)threadIdx.x < n) ids[threadIdx.x] = (base + (int64_t)threadIdx.x) % C;
  __threadfence_block(); __syncthreads();
  PerUpdateOp::run(0u, 1u, pr, pminmax, ids, (const double*)nullptr, (const float*)nullptr, (const float*)(pminmax + 2), alpha, n);
  if (!touch) return;
  __threadfence_block(); __syncthreads();
  for (unsigned b = 0; (int64_t)b * (blockDim.x >> 6) < n; ++b) LeafRefreshOp::run(b, 1u, pr, ids, n, N, nlev, run, total);
  __threadfence_block(); __syncthreads();
  TreeTouchOp::run(0u, 1u, ids, n, N, nlev, total);
}
