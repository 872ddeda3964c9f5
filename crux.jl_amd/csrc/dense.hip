// dense.hip -- shape-generic differentiable Chain(Dense...) engine on gfx950 f32 MFMA: forward with cached activations and the
// reverse pass (parameter gradients and input gradient). This is what Zygote's pullback of value(pi, x) does for the
// off-policy learners (src/training.jl:16-18 through src/utils.jl:76-96, src/model_free/rl/sac.jl:34-52), where the hidden
// layers are 256 wide (configs C3/C4) and a single-workgroup kernel no longer fits.
//
// Layout: activations are [feature][sample] column-major as in the reference (feature index fastest), weights out x in
// column-major (W[o + out*k]). Every layer is one launch of a tile GEMM: one wave per 16x16 output tile, the K loop in chunks
// of 16 with v_mfma_f32_16x16x4_f32 (exact f32 products, k-ordered fma chain, deterministic). Operands that are contiguous
// along K are fetched as one 16-byte load per lane and chunk (lane group g takes k = 16q+4g+r for MFMA r -- any k permutation
// is legal as long as both operands agree); the others as four coalesced dword loads. Operands live in L2/MALL: the whole
// working set of C3/C4 (a few hundred KB) is far below the 4 MB L2 of one XCD.
//   forward   Y[o,s]  = act(b[o] + sum_k W[o,k] X[k,s])                 M=out N=B K=in
//   backward  dX[k,s] = act'(X[k,s]) * sum_o W[o,k] dZ[o,s]             M=in  N=B K=out   (act' of the layer that produced X)
//   weights   dW[o,k] = scale * sum_s dZ[o,s] X[k,s];  db[o] = scale * sum_s dZ[o,s]      M=out N=in K=B   (db rides along in the first column tile)
#include "common.h"
#include "exec.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { EPI_FWD = 0, EPI_BWD_DATA = 1, EPI_WGRAD = 2 };

struct GemmArgs {
  const float* A; int64_t sAi, sAk;      // A(i,k) = A[i*sAi + k*sAk]
  const float* B; int64_t sBk, sBj;      // B(k,j) = B[k*sBk + j*sBj]
  int M, N, K;
  float* C; int64_t sCj;                 // C(i,j) = C[i + j*sCj]
  int epi; const float* bias; int act; const float* ysrc; float scale; float* gbias;
  int32_t* nf;                           // EPI_WGRAD: nf[0] |= 1 when a value this launch stores is NaN (the self-gated Adam of the fused epochs, sac.hip AdamSelfOp); may be NULL
};

// SPLITK: the four waves of a workgroup share ONE output tile and take a quarter of K each (combined through LDS in wave order, so the
// result is deterministic): a 256-deep reduction then costs one L2 round trip instead of four -- these launches are latency-bound.
template <bool AV, bool BV, bool SPLITK>
#define GEMM16_PART (3 * 64 * 5)      // floats of LDS the split-K combine needs (waves 1..3 hand 4 + 1 values per lane to wave 0); ONE buffer, owned by the caller
struct Gemm16 { static __device__ __forceinline__ void run(const unsigned bid_, const GemmArgs& q, float* part, const int quart = 0) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int tm = (q.M + 15) >> 4, tn = (q.N + 15) >> 4;
  const int tile = SPLITK ? (int)bid_ : (int)bid_ * 4 + wv;
  if (tile >= tm * tn) return;
  // The reduction over K is DEFINED in four quarters of kper = ceil16(ceil(K / 4)) whenever the split-K form applies (K >= 128): quarter sums are
  // formed separately and added in order ((q0 + q1) + q2) + q3. SPLITK gives one quarter to each wave of the workgroup (one L2 round trip for the
  // whole depth: the stand-alone launches); `quart` lets ONE wave walk the four quarters (four tiles per workgroup: what the fused executor prefers on
  // its 32 CUs). Both produce the same bits.
  const int kper = (SPLITK || quart) ? ((((q.K + 3) >> 2) + 15) & ~15) : q.K;
  const int i0 = (tile % tm) << 4, j0 = (tile / tm) << 4;
  const int ia = i0 + c, jb = j0 + c;
  const bool va = ia < q.M, vb = jb < q.N;
  const float* pa = q.A + (int64_t)(va ? ia : 0) * q.sAi;
  const float* pb = q.B + (int64_t)(vb ? jb : 0) * q.sBj;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float rowsum = 0.f;                      // EPI_WGRAD, first column tile: db[i] = sum_k A(i, k) rides along (A = dZ, k = sample)
  const bool want_rowsum = q.epi == EPI_WGRAD && j0 == 0 && q.gbias != nullptr;
  const int nparts = (!SPLITK && quart) ? 4 : 1;
  for (int part = 0; part < nparts; ++part) {
  const int kbeg = SPLITK ? wv * kper : (quart ? part * kper : 0), kend = (SPLITK || quart) ? (kbeg + kper < q.K ? kbeg + kper : q.K) : q.K;
  f32x4 pacc = {0.f, 0.f, 0.f, 0.f}; float prow = 0.f;
  // K loop in steps of 64: the loads of four 16-wide chunks are issued before the first MFMA, so one L2 round trip (~1 us under load) is paid per
  // 64 k instead of per 16 (the kernel is a wave-per-tile design with no LDS staging: latency, not bandwidth, is what has to be hidden). The loads of
  // step k0 + 64 are in flight while the MFMAs of step k0 run (two register sets, loop unrolled by two): a 256-deep reduction walked by one wave costs
  // about one round trip instead of four. The accumulation order (k ascending) is unchanged.
  auto load = [&](const int k0, float (&a)[4][4], float (&b)[4][4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int kb = k0 + 16 * u + 4 * g;
      if (AV) { f32x4 t = {0.f, 0.f, 0.f, 0.f}; if (va && kb < kend) t = *(const f32x4*)(pa + kb);
#pragma unroll
        for (int r = 0; r < 4; ++r) a[u][r] = t[r]; }
      else {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int kk = kb + r; a[u][r] = (va && kk < kend) ? pa[(int64_t)kk * q.sAk] : 0.f; } }
      if (BV) { f32x4 t = {0.f, 0.f, 0.f, 0.f}; if (vb && kb < kend) t = *(const f32x4*)(pb + kb);
#pragma unroll
        for (int r = 0; r < 4; ++r) b[u][r] = t[r]; }
      else {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int kk = kb + r; b[u][r] = (vb && kk < kend) ? pb[(int64_t)kk * q.sBk] : 0.f; } }
    } };
  auto compute = [&](const float (&a)[4][4], const float (&b)[4][4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) pacc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][r], b[u][r], pacc, 0, 0, 0);
    if (want_rowsum) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) prow += a[u][r];
    } };
  if (kbeg < kend) {
    float a0[4][4], b0[4][4], a1[4][4], b1[4][4];
    int k0 = kbeg; load(k0, a0, b0);
    for (;;) {
      bool more = k0 + 64 < kend; if (more) load(k0 + 64, a1, b1);
      compute(a0, b0);
      if (!more) break; k0 += 64;
      more = k0 + 64 < kend; if (more) load(k0 + 64, a0, b0);
      compute(a1, b1);
      if (!more) break; k0 += 64;
    }
  }
  if (part == 0) { acc = pacc; rowsum = prow; } else { acc[0] += pacc[0]; acc[1] += pacc[1]; acc[2] += pacc[2]; acc[3] += pacc[3]; rowsum += prow; }
  }
  if (SPLITK) {        // waves 1..3 hand their partial tile (and row sum) to wave 0, which adds them in wave order
    if (wv > 0) { float* p = part + ((wv - 1) * 64 + lane) * 5; p[0] = acc[0]; p[1] = acc[1]; p[2] = acc[2]; p[3] = acc[3]; p[4] = rowsum; }
    __syncthreads();
    if (wv > 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w) { const float* p = part + (w * 64 + lane) * 5; acc[0] += p[0]; acc[1] += p[1]; acc[2] += p[2]; acc[3] += p[3]; rowsum += p[4]; }
  }
  if (want_rowsum) {   // lanes c, c+16, c+32, c+48 hold the four k-groups of row i0+c: fixed-order combine
    rowsum += __shfl_xor(rowsum, 16, 64); rowsum += __shfl_xor(rowsum, 32, 64);
    if (g == 0 && va) { const float bv = q.scale * rowsum; q.gbias[ia] = bv; if (q.nf && bv != bv) atomicOr((int*)q.nf, 1); }
  }
  // D layout: reg r <-> row i0+4g+r, column j0+c
  const int j = j0 + c;
  if (j >= q.N) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) { const int i = i0 + 4 * g + r; if (i >= q.M) continue;
    const int64_t ci = (int64_t)i + (int64_t)j * q.sCj; float v = acc[r];
    if (q.epi == EPI_FWD) v = crux_act(q.act, v + q.bias[i]);
    else if (q.epi == EPI_BWD_DATA) { if (q.ysrc) v = crux_act_grad(q.act, q.ysrc[ci], v); }
    else { v *= q.scale; if (q.nf && v != v) atomicOr((int*)q.nf, 1); }
    q.C[ci] = v; }
} };
template <bool AV, bool BV, bool SPLITK>
__global__ __launch_bounds__(256) void k_gemm16(GemmArgs q) { __shared__ float part[SPLITK ? GEMM16_PART : 1]; Gemm16<AV, BV, SPLITK>::run(blockIdx.x, q, part); }
// the executor's form (exec.hip): the variant is data
struct GemmOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, GemmArgs q, int variant) {
  __shared__ float part[GEMM16_PART];         // one combine buffer for all variants (a static array per template instantiation cost the phase kernel 15 KB of LDS)
  const int quart = (variant >> 3) & 1;       // bit 0 AV, bit 1 BV, bit 2 split-K over the workgroup's waves, bit 3 the four K quarters walked by one wave
  switch (variant & 7) {
    case 0: Gemm16<false, false, false>::run(bid_, q, part, quart); break; case 1: Gemm16<true, false, false>::run(bid_, q, part, quart); break;
    case 2: Gemm16<false, true, false>::run(bid_, q, part, quart); break;  case 3: Gemm16<true, true, false>::run(bid_, q, part, quart); break;
    case 4: Gemm16<false, false, true>::run(bid_, q, part); break;  case 5: Gemm16<true, false, true>::run(bid_, q, part); break;
    case 6: Gemm16<false, true, true>::run(bid_, q, part); break;   default: Gemm16<true, true, true>::run(bid_, q, part); break; } } };

#include "dense_fused.h"
// CRUX_DENSE_FUSED=0: every layer through its own Gemm16 launch (the round-3 chains; tests compare the two forms bit for bit). Read once.
static bool dense_fused_on() { return crux_sw().dense_fused != 0; }
bool crux_dense_fwd_fused(const crux_mlp* n) { return dense_fused_on() && df_fwd12_ok(n->nd); }                      // layers 0 + 1 in one launch (exec.hip's phase plans ask)
bool crux_dense_bwd_fused(const crux_mlp* n, int64_t B) { return dense_fused_on() && df_bwd_ok(n->nd, B); }
bool crux_dense_bwd_fused3(const crux_mlp* n, int64_t B) {      // + the output layer's data gradient folded into the pair: the whole pullback of a three-layer network is ONE phase
  const bool on3 = crux_sw().dense_fused != 2;      // CRUX_DENSE_FUSED=2: the pair without the folded output layer (tests)
  return on3 && crux_dense_bwd_fused(n, B) && n->nd.L == 3 && (n->nd.dims[3] == 1 || n->nd.dims[3] == 4) && n->nd.acts[2] == CRUX_ACT_IDENTITY; }        // layer 1's dW beside (layer 1's dX -> layer 0's dW) in one phase

// dZ = act'(Y) .* dY for the output layer
struct ActGradOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, const float* __restrict__ dy, const float* __restrict__ y, int act, int64_t n, float* __restrict__ dz) {
  const int64_t i = (int64_t)bid_ * blockDim.x + threadIdx.x; if (i >= n) return;
  dz[i] = crux_act_grad(act, y[i], dy[i]);
} };
__global__ void k_act_grad(const float* __restrict__ dy, const float* __restrict__ y, int act, int64_t n, float* __restrict__ dz) { ActGradOp::run(blockIdx.x, gridDim.x, dy, y, act, n, dz); }
static inline bool vec_ok(const float* p, int64_t s_k, int64_t s_outer, int K) {
  return s_k == 1 && (K & 3) == 0 && (s_outer & 3) == 0 && (((uintptr_t)p) & 15) == 0;
}
static int32_t launch_gemm(crux_ctx* c, const GemmArgs& q, hipStream_t st) {
  const int tiles = ((q.M + 15) >> 4) * ((q.N + 15) >> 4);
  const dim3 block(256);
  const bool av = vec_ok(q.A, q.sAk, q.sAi, q.K), bv = vec_ok(q.B, q.sBk, q.sBj, q.K);
  constexpr bool no_split = false;
  if (crux_exec_recording(c)) {                       // fused sequence (exec.hip): the same tile bodies, run by the persistent executor
    // the stand-alone launch would split K over the four waves of a workgroup here; on the executor's 32 CUs one round of fat blocks beats several rounds of
    // thin ones, so up to 32 tiles keep the split form and larger GEMMs give each wave a whole tile with the K quarters walked in order (same bits)
    // (the persistent one-XCD executor prefers fat blocks; the default phase launches run over the whole chip like the stand-alone launches and split whenever those do)
    const bool persistent = crux_sw().exec_persistent;
    const bool deep = q.K >= 128 && tiles <= 4096 && !no_split, split = deep && (!persistent || tiles <= 32);
    crux_exec_push<GemmOp, OP_GEMM>(c, (unsigned)(split ? tiles : (tiles + 3) / 4), q, (int)((av ? 1 : 0) | (bv ? 2 : 0) | (split ? 4 : 0) | ((deep && !split) ? 8 : 0)));
    return CRUX_OK;
  }
  if (q.K >= 128 && tiles <= 4096 && !no_split) {     // deep reductions: split K over the workgroup's four waves
    const dim3 grid((unsigned)tiles);
    if (av && bv) hipLaunchKernelGGL((k_gemm16<true, true, true>), grid, block, 0, st, q);
    else if (av) hipLaunchKernelGGL((k_gemm16<true, false, true>), grid, block, 0, st, q);
    else if (bv) hipLaunchKernelGGL((k_gemm16<false, true, true>), grid, block, 0, st, q);
    else hipLaunchKernelGGL((k_gemm16<false, false, true>), grid, block, 0, st, q);
    return crux_launch_check(c, "k_gemm16");
  }
  const dim3 grid((unsigned)((tiles + 3) / 4));
  if (av && bv) hipLaunchKernelGGL((k_gemm16<true, true, false>), grid, block, 0, st, q);
  else if (av) hipLaunchKernelGGL((k_gemm16<true, false, false>), grid, block, 0, st, q);
  else if (bv) hipLaunchKernelGGL((k_gemm16<false, true, false>), grid, block, 0, st, q);
  else hipLaunchKernelGGL((k_gemm16<false, false, false>), grid, block, 0, st, q);
  return crux_launch_check(c, "k_gemm16");
}

// ---- per-network workspace: activations 1..L and two delta buffers, [maxdim x B] each --------------------------
static int32_t ensure_ws(crux_mlp* n, int64_t B) {
  if (n->ws && n->ws_B >= B) return CRUX_OK;
  crux_ctx* c = n->ctx;
  if (n->ws) { HIPCHK(c, hipStreamSynchronize(c->stream)); (void)hipFree(n->ws); n->ws = nullptr; n->ws_B = 0; }
  int64_t cap = 256; while (cap < B) cap *= 2;
  size_t tot = 0; for (int l = 1; l <= n->nd.L; ++l) tot += (size_t)n->nd.dims[l] * (size_t)cap;
  tot += 2 * (size_t)n->nd.maxdim * (size_t)cap;
  if (n->nd.L >= 2) tot += 4 * (size_t)n->nd.dims[1] * (size_t)(n->nd.dims[0] + 4);      // quarter partials of layer 0's gradient (Dgrad2W1Op)
  if (hipMalloc(&n->ws, sizeof(float) * tot + 64) != hipSuccess) return crux_fail(c, CRUX_ENOMEM, "dense workspace: hipMalloc(%zu) failed", sizeof(float) * tot);
  n->ws_B = cap; return CRUX_OK;
}
float* crux_dense_act(crux_mlp* n, int l) {   // l in 1..L
  size_t off = 0; for (int q = 1; q < l; ++q) off += (size_t)n->nd.dims[q] * (size_t)n->ws_B;
  return n->ws + off;
}
static float* ws_part(crux_mlp* n) {      // behind the two delta buffers
  size_t off = 0; for (int q = 1; q <= n->nd.L; ++q) off += (size_t)n->nd.dims[q] * (size_t)n->ws_B;
  return n->ws + off + 2 * (size_t)n->nd.maxdim * (size_t)n->ws_B;
}
static float* ws_delta(crux_mlp* n, int which) {
  size_t off = 0; for (int q = 1; q <= n->nd.L; ++q) off += (size_t)n->nd.dims[q] * (size_t)n->ws_B;
  return n->ws + off + (size_t)which * (size_t)n->nd.maxdim * (size_t)n->ws_B;
}

int32_t crux_dense_forward(crux_mlp* n, const float* d_x, int64_t B, hipStream_t st) {
  crux_ctx* c = n->ctx; const NetDesc& nd = n->nd;
  if (nd.L < 1) return crux_fail(c, CRUX_EINVAL, "forward: the handle has no layers");
  if (B < 1 || B > (1 << 20)) return crux_fail(c, CRUX_EINVAL, "forward: batch %lld out of range", (long long)B);
  int32_t rc = ensure_ws(n, B); if (rc) return rc;
  const float* x = d_x; int l0 = 0;
  if (crux_dense_fwd_fused(n)) {      // layers 0 and 1 as one launch (dense_fused.h)
    Fwd12Args a{}; a.W1 = n->p + nd.woff[0]; a.b1 = n->p + nd.boff[0]; a.W2 = n->p + nd.woff[1]; a.b2 = n->p + nd.boff[1]; a.x = d_x; a.H1 = crux_dense_act(n, 1); a.H2 = crux_dense_act(n, 2);
    a.in0 = nd.dims[0]; a.out1 = nd.dims[1]; a.out2 = nd.dims[2]; a.B = (int32_t)B; a.act1 = nd.acts[0]; a.act2 = nd.acts[1];
    CRUX_RUN(c, Fwd12Op, OP_FWD12, k_fwd12, df_fwd12_blocks(nd, B), 256, st, a);
    rc = crux_launch_check(c, "k_fwd12"); if (rc) return rc;
    x = a.H2; l0 = 2;
  }
  for (int l = l0; l < nd.L; ++l) {
    const int in = nd.dims[l], out = nd.dims[l + 1];
    GemmArgs q{}; q.A = n->p + nd.woff[l]; q.sAi = 1; q.sAk = out; q.B = x; q.sBk = 1; q.sBj = in; q.M = out; q.N = (int)B; q.K = in;
    q.C = crux_dense_act(n, l + 1); q.sCj = out; q.epi = EPI_FWD; q.bias = n->p + nd.boff[l]; q.act = nd.acts[l];
    rc = launch_gemm(c, q, st); if (rc) return rc;
    x = q.C;
  }
  return CRUX_OK;
}

// pieces of the passes for callers that fuse the output layer into an op of their own (exec.hip: the fused SAC epoch):
// layers 0 + 1 only (the caller's op evaluates the output layer from crux_dense_act(n, 2))
int32_t crux_dense_forward12(crux_mlp* n, const float* d_x, int64_t B, hipStream_t st) {
  crux_ctx* c = n->ctx; const NetDesc& nd = n->nd;
  if (!crux_dense_fwd_fused(n) || nd.L != 3) return crux_fail(c, CRUX_EUNSUP, "forward12: not a fused three-layer shape");
  int32_t rc = ensure_ws(n, B); if (rc) return rc;
  Fwd12Args a{}; a.W1 = n->p + nd.woff[0]; a.b1 = n->p + nd.boff[0]; a.W2 = n->p + nd.woff[1]; a.b2 = n->p + nd.boff[1]; a.x = d_x; a.H1 = crux_dense_act(n, 1); a.H2 = crux_dense_act(n, 2);
  a.in0 = nd.dims[0]; a.out1 = nd.dims[1]; a.out2 = nd.dims[2]; a.B = (int32_t)B; a.act1 = nd.acts[0]; a.act2 = nd.acts[1];
  CRUX_RUN(c, Fwd12Op, OP_FWD12, k_fwd12, df_fwd12_blocks(nd, B), 256, st, a);
  return crux_launch_check(c, "k_fwd12");
}
// the input-gradient chain down to layer 0's dZ (the caller's op applies layer 0's weights): returns the buffer [dims[1] x B] Dgrad2W1Op fills
int32_t crux_dense_dgrad_to_dz1(crux_mlp* n, const float* d_x, int64_t B, const float* d_dy, const float** d_dz1, hipStream_t st) {
  crux_ctx* c = n->ctx; const NetDesc& nd = n->nd;
  if (!crux_dense_bwd_fused3(n, B) || !n->ws || n->ws_B < B) return crux_fail(c, CRUX_EUNSUP, "dgrad_to_dz1: not a fused three-layer shape with a cached forward pass");
  DzSrc z{}; z.W3 = n->p + nd.woff[2]; z.dZ3 = d_dy; z.out3 = nd.dims[3]; z.act = nd.acts[1];
  Dgrad2Args a{}; a.z = z; a.W2 = n->p + nd.woff[1]; a.dZ2 = crux_dense_act(n, 2); a.H1 = crux_dense_act(n, 1); a.x = d_x; a.part = ws_part(n); a.dZ1 = ws_delta(n, 0);
  a.in0 = nd.dims[0]; a.out1 = nd.dims[1]; a.out2 = nd.dims[2]; a.B = (int32_t)B; a.act0 = nd.acts[0]; a.want_g = 0;
  CRUX_RUN(c, Dgrad2W1Op, OP_DGRAD2W1, k_dgrad2w1, (unsigned)((nd.dims[1] >> 4) * 4), 256, st, a);
  *d_dz1 = a.dZ1;
  return crux_launch_check(c, "k_dgrad2w1");
}

// Reverse pass after crux_dense_forward(n, d_x, B) with the same d_x. d_dy [out_L x B] is not modified.
int32_t crux_dense_backward(crux_mlp* n, const float* d_x, int64_t B, const float* d_dy, float gscale, bool want_g, float* d_dx, hipStream_t st, Sumsq2Fix* defer, int defer_slot, int32_t* nanflags) {
  crux_ctx* c = n->ctx; const NetDesc& nd = n->nd;
  if (nd.L < 1 || !n->ws || n->ws_B < B) return crux_fail(c, CRUX_EINVAL, "backward: no cached forward pass for this batch");
  const float* dcur = d_dy; float* dnxt = ws_delta(n, 0); float* dspare = ws_delta(n, 1);
  if (nd.acts[nd.L - 1] != CRUX_ACT_IDENTITY) {   // dZ_L = act'(Y_L) .* dY; an identity output layer (the usual critic / mean head) uses dY as it is
    const int64_t cnt = (int64_t)nd.dims[nd.L] * B;
    CRUX_RUN(c, ActGradOp, OP_ACT_GRAD, k_act_grad, (unsigned)((cnt + 255) / 256), 256, st, d_dy, crux_dense_act(n, nd.L), nd.acts[nd.L - 1], cnt, dnxt);
    dcur = dnxt; dnxt = dspare; dspare = const_cast<float*>(dcur);
  }
  // with parameter gradients the fused pair leaves layer 0's gradient as quarter partials: only where the caller runs Sumsq2Op next (defer); input-gradient chains need no such reader
  const bool fused = crux_dense_bwd_fused(n, B) && (!want_g || defer != nullptr);
  // three layers with a narrow identity output layer: its data gradient (K = out <= 4) is formed inside the consumers' panel staging (DzSrc) -- no launch, no dZ buffer
  const bool fused3 = fused && crux_dense_bwd_fused3(n, B);
  for (int l = nd.L - 1; l >= 0; --l) {
    const int in = nd.dims[l], out = nd.dims[l + 1];
    const float* x = l == 0 ? d_x : crux_dense_act(n, l);
    if (fused3 && l == 2) {      // the output layer: its weight gradient only (same phase as the fused pair below)
      if (want_g) { GemmArgs q{}; q.A = dcur; q.sAi = 1; q.sAk = out; q.B = x; q.sBk = in; q.sBj = 1; q.M = out; q.N = in; q.K = (int)B;
        q.C = n->g + nd.woff[l]; q.sCj = out; q.epi = EPI_WGRAD; q.scale = gscale; q.gbias = n->g + nd.boff[l]; q.nf = nanflags;
        int32_t rc = launch_gemm(c, q, st); if (rc) return rc; }
      continue;
    }
    if (fused && l == 1) {      // layers 1 and 0 together (dense_fused.h): dW1' = dcur X1' | dX1 = act0'(X1) .* (W1'' dcur) -> layer 0's dW, db inside the same workgroups
      DzSrc z{}; const float* dz1 = dcur;
      if (fused3) { z.W3 = n->p + nd.woff[2]; z.dZ3 = dcur; z.out3 = nd.dims[3]; z.act = nd.acts[1]; dz1 = crux_dense_act(n, 2); }      // dcur is still dZ of the output layer; the operand pointer becomes H2
      if (want_g) { Wgrad2Args w{}; w.z = z; w.dZ = dz1; w.X = x; w.dW = n->g + nd.woff[1]; w.db = n->g + nd.boff[1]; w.scale = gscale; w.out = out; w.in = in; w.B = (int32_t)B; w.nf = nanflags;
        CRUX_RUN(c, Wgrad2Op, OP_WGRAD2, k_wgrad2, (unsigned)((out >> 5) * (in >> 5)), 256, st, w); }
      Dgrad2Args a{}; a.z = z; a.W2 = n->p + nd.woff[1]; a.dZ2 = dz1; a.H1 = x; a.x = d_x; a.part = ws_part(n); a.dZ1 = d_dx ? dnxt : nullptr;
      a.in0 = nd.dims[0]; a.out1 = in; a.out2 = out; a.B = (int32_t)B; a.act0 = nd.acts[0]; a.want_g = want_g ? 1 : 0; a.nf = nanflags;
      CRUX_RUN(c, Dgrad2W1Op, OP_DGRAD2W1, k_dgrad2w1, (unsigned)((in >> 4) * 4), 256, st, a);
      if (want_g) { defer->part[defer_slot] = a.part; defer->out1[defer_slot] = in; defer->in0[defer_slot] = nd.dims[0]; defer->woff[defer_slot] = nd.woff[0]; defer->boff[defer_slot] = nd.boff[0]; defer->scale[defer_slot] = gscale; }
      if (d_dx) {                // the input gradient of layer 0 from the dZ of layer 0 the fused op left in the workspace
        GemmArgs q{}; q.A = n->p + nd.woff[0]; q.sAi = in; q.sAk = 1; q.B = dnxt; q.sBk = 1; q.sBj = in; q.M = nd.dims[0]; q.N = (int)B; q.K = in;
        q.C = d_dx; q.sCj = nd.dims[0]; q.epi = EPI_BWD_DATA; q.ysrc = nullptr; q.act = CRUX_ACT_IDENTITY;
        int32_t rc = launch_gemm(c, q, st); if (rc) return rc; }
      break;
    }
    if (want_g) {
      GemmArgs q{}; q.A = dcur; q.sAi = 1; q.sAk = out; q.B = x; q.sBk = in; q.sBj = 1; q.M = out; q.N = in; q.K = (int)B;
      q.C = n->g + nd.woff[l]; q.sCj = out; q.epi = EPI_WGRAD; q.scale = gscale; q.gbias = n->g + nd.boff[l]; q.nf = nanflags;   // db rides along in the first column tile
      int32_t rc = launch_gemm(c, q, st); if (rc) return rc;
    }
    if (l > 0 || d_dx) {
      GemmArgs q{}; q.A = n->p + nd.woff[l]; q.sAi = out; q.sAk = 1; q.B = dcur; q.sBk = 1; q.sBj = out; q.M = in; q.N = (int)B; q.K = out;
      q.C = l > 0 ? dnxt : d_dx; q.sCj = in; q.epi = EPI_BWD_DATA; q.ysrc = l > 0 ? x : nullptr; q.act = l > 0 ? nd.acts[l - 1] : CRUX_ACT_IDENTITY;
      int32_t rc = launch_gemm(c, q, st); if (rc) return rc;
      dcur = dnxt; float* t = dnxt; dnxt = dspare; dspare = t;      // ping-pong between the two workspace buffers; d_dy itself is never written
    }
  }
  return crux_launch_check(c, "dense backward");
}

extern "C" {

int32_t crux_mlp_forward_cached(crux_mlp* net, const float* d_x, int64_t B, float* d_y) {
  if (!net || !d_x) return CRUX_EINVAL;
  int32_t rc = crux_dense_forward(net, d_x, B, net->ctx->stream); if (rc) return rc;
  if (d_y) HIPCHK(net->ctx, hipMemcpyAsync(d_y, crux_dense_act(net, net->nd.L), sizeof(float) * (size_t)net->nd.dims[net->nd.L] * (size_t)B, hipMemcpyDeviceToDevice, net->ctx->stream));
  return CRUX_OK;
}
int32_t crux_mlp_backward(crux_mlp* net, const float* d_x, int64_t B, const float* d_dy, float grad_scale, int32_t want_param_grads, float* d_dx) {
  if (!net || !d_x || !d_dy) return CRUX_EINVAL;
  return crux_dense_backward(net, d_x, B, d_dy, grad_scale, want_param_grads != 0, d_dx, net->ctx->stream);
}

}  // extern "C"
