// sac_fused.h -- per-sample-tile ops of the fused SAC epoch (round 4): an output layer (out <= 4, K = 128 / 192 / 256) is a handful of MFMAs per 16 samples, and what
// follows it in sac_target / double_Q_loss / sac_actor_loss / exploration is arithmetic on those 16 samples alone. One workgroup per 16-sample tile therefore runs the
// output layers it needs (Gemm16's split-K form: wave w walks quarter w of K, wave 0 adds the quarters in wave order -- the bits of the per-layer launch) and the head
// right behind them, instead of one launch for the layers and one for the head:
//   ActorExploreTileOp     mu = output layer of the actor; a = mu + sigma * eps, logpdf                    (policies.jl:333-344)      was: GEMM | GaussExploreOp
//   SacCriticTileOp        Q1-, Q2-(s', a') and Q1, Q2(s, a); y = sac_target; double_Q_loss heads         (sac.jl:4-9, utils.jl:89-96) was: 2 GEMM phases | target + heads
//   SacActorTileOp         Q1, Q2(s, a~); sac_actor_loss head                                              (sac.jl:34-40)             was: GEMM | ActorHeadOp
//   CriticDxActorGradTileOp  d(min Q)/d(s, a~) through the critics' first layers; reverse of exploration   (Zygote pullback of the same) was: GEMM | ActorGradOp
// The loss statistics are block sums in the one-block heads; here every tile leaves its samples' terms and CriticInfo2Op / ActorInfo2Op add them in block_sum256's order.
// Results are bit-identical to the call-by-call chain (tools/fused_check.py, tests/test_gpu_round4.py).
#pragma once
#include "common.h"
#include <type_traits>

struct TileSet { const float* A; const float* bias; const float* H; float* Z; };      // A: the layer's weights; H: its input [K x B]; Z: its output [M x B] or NULL
#define TILE_PART_FLOATS(NS) (3 * (NS) * 64 * 4)
// M <= 16 output rows x the 16 samples j0 .. j0 + 15. AK: the weights are contiguous along K (a transposed use: dX = W' dZ with A(i, k) = W[k + K i]); otherwise A(i, k) = W[i + M k].
// On return wave 0 holds z[n][r] = row 4 g + r, sample j0 + c (bias added when given); the other waves hold garbage. Contains one __syncthreads.
struct TileNoHook { __device__ __forceinline__ void operator()() const {} };
// `between`: work of the caller that does not depend on the tile (random draws), placed between the issue of the loads and their first use
// (the loops over the NS tile sets are compile-time recursions: with `#pragma unroll` the epilogue loop of the NS = 4 instance stayed a loop over a switch, its `s[n]` a
//  run-time index, and the caller's whole argument struct went to private memory -- 176 bytes of scratch per thread in every phase kernel; round 6)
template <int I, int N, class Fn> __device__ __forceinline__ void tile_static_for(Fn&& fn) { if constexpr (I < N) { fn(std::integral_constant<int, I>{}); tile_static_for<I + 1, N>(fn); } }
template <int NS, bool AK, class F = TileNoHook> __device__ __forceinline__ void tile_splitk(const TileSet (&s)[NS], int M, int K, int B, int j0, float* part, f32x4 (&z)[NS], F between = F()) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int kper = K >> 2, ng = kper >> 4;                    // K in {128, 192, 256}: Gemm16's quarter = K / 4, 2..4 sixteen-groups
  const int jc = (j0 + c < B) ? j0 + c : B - 1; const int ic = c < M ? c : 0; const float mrow = c < M ? 1.f : 0.f;
  float a[NS][4][4]; f32x4 b[NS][4];
  tile_static_for<0, NS>([&](auto nc) { constexpr int n = decltype(nc)::value;
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int k16 = wv * kper + 16 * (u < ng ? u : 0) + 4 * g;      // (groups past the quarter re-read group 0 and are never used)
      b[n][u] = *(const f32x4*)(s[n].H + k16 + (int64_t)K * jc);
      if (AK) { const f32x4 t = *(const f32x4*)(s[n].A + k16 + (int64_t)K * ic);
#pragma unroll
        for (int r = 0; r < 4; ++r) a[n][u][r] = t[r] * mrow; }
      else {
#pragma unroll
        for (int r = 0; r < 4; ++r) a[n][u][r] = s[n].A[ic + (int64_t)M * (k16 + r)] * mrow; } } });
  between();
  tile_static_for<0, NS>([&](auto nc) { constexpr int n = decltype(nc)::value; f32x4 pa = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) { if (u < ng) {
#pragma unroll
      for (int r = 0; r < 4; ++r) pa = __builtin_amdgcn_mfma_f32_16x16x4f32(a[n][u][r], b[n][u][r], pa, 0, 0, 0); } }
    z[n] = pa; });
  if (wv > 0) {
    tile_static_for<0, NS>([&](auto nc) { constexpr int n = decltype(nc)::value; *(f32x4*)(part + (((wv - 1) * NS + n) * 64 + lane) * 4) = z[n]; }); }
  __syncthreads();
  if (wv > 0) return;
  tile_static_for<0, NS>([&](auto nc) { constexpr int n = decltype(nc)::value;
#pragma unroll
    for (int w = 0; w < 3; ++w) { const f32x4 p = *(const f32x4*)(part + ((w * NS + n) * 64 + lane) * 4); z[n][0] += p[0]; z[n][1] += p[1]; z[n][2] += p[2]; z[n][3] += p[3]; }
    if (s[n].bias) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int i = 4 * g + r; z[n][r] = z[n][r] + s[n].bias[i < M ? i : 0]; } }
    if (s[n].Z && j0 + c < B) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int i = 4 * g + r; if (i < M) s[n].Z[i + (int64_t)M * (j0 + c)] = z[n][r]; } } });
}

// ---- the actor's output layer + exploration(pi::GaussianPolicy, s) for up to two independent draws from the same means --------------------------------
struct ExploreCfg { uint64_t counter; float* sa; float* lp; float* eps; };      // sa: vcat(s, a) [od + ad x B] or NULL; eps: the draws or NULL
struct ActorExploreArgs { TileSet mu; const float* ls; const float* s; int32_t od, ad, K, B, n_cfg; uint64_t seed; ExploreCfg cfg[2]; };
struct ActorExploreTileOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, ActorExploreArgs q) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c = lane & 15, g = lane >> 4; const int j0 = (int)bid_ << 4;
  const TileSet sets[1] = {q.mu}; f32x4 z[1];
  const int64_t j = j0 + c; const bool mine = wv == 0 && g == 0 && j < q.B;      // ad <= 4: lane c of group 0 holds mu[0..ad) of sample j0 + c
  float epsv[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  // the draws (Philox + Box-Muller in Float64: ~1 us) do not depend on the means: they are taken while the tile's operands are on their way
  // (compile-time trip counts with predicates: run-time indices into q.cfg[], epsv[][] and z[0][] put the argument struct and the draws into private memory -- the 192 bytes of
  //  scratch every phase kernel carried; round 6. n_cfg <= 2, ad <= 4: sac_tile_case)
  tile_splitk<1, false>(sets, q.ad, q.K, q.B, j0, df_lds, z, [&]() { if (mine) {
#pragma unroll
    for (int e = 0; e < 2; ++e) { const uint64_t ctr = e == 0 ? q.cfg[0].counter : q.cfg[1].counter;
#pragma unroll
      for (int d = 0; d < 4; ++d) if (e < q.n_cfg && d < q.ad) epsv[e][d] = sac_randn(q.seed, ctr, (uint32_t)(j * q.ad + d)); } } });
  if (!mine) return;
#pragma unroll
  for (int e = 0; e < 2; ++e) { if (e >= q.n_cfg) continue;
    float* const x_sa = e == 0 ? q.cfg[0].sa : q.cfg[1].sa; float* const x_lp = e == 0 ? q.cfg[0].lp : q.cfg[1].lp; float* const x_eps = e == 0 ? q.cfg[0].eps : q.cfg[1].eps;
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < 4; ++d) { if (d >= q.ad) continue;             // GaussExploreOp (sac.hip), the same expressions
      const float sg = expf(q.ls[d]); const float ep = epsv[e][d];
      const float m = z[0][d]; const float a = __fadd_rn(__fmul_rn(ep, sg), m);
      const float s2 = __fmul_rn(sg, sg), df = __fsub_rn(a, m);
      acc = __fadd_rn(acc, __fsub_rn(__fsub_rn(-(__fmul_rn(df, df)) / __fmul_rn(2.f, s2), 0.9189385332046727f), q.ls[d]));
      if (x_sa) x_sa[j * (q.od + q.ad) + q.od + d] = a;
      if (x_eps) x_eps[j * q.ad + d] = ep; }
    if (x_sa) for (int k = 0; k < q.od; ++k) x_sa[j * (q.od + q.ad) + k] = q.s[j * q.od + k];
    x_lp[j] = acc; }
} };

// ---- sac_target + the two heads of double_Q_loss ---------------------------------------------------------------------------------------------------------------------
struct SacCriticArgs { TileSet q1t, q2t, q1, q2; const float* r; const uint8_t* done; const float* lp; const float* log_alpha; const float* w; float gamma, scale; int32_t K, B;
                       float* y; float* dy1; float* dy2; float* term1; float* term2; };
struct SacCriticTileOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, SacCriticArgs q) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c = lane & 15, g = lane >> 4; const int j0 = (int)bid_ << 4;
  const TileSet sets[4] = {q.q1t, q.q2t, q.q1, q.q2}; f32x4 z[4];
  tile_splitk<4, false>(sets, 1, q.K, q.B, j0, df_lds, z);
  if (wv > 0 || g != 0) return;
  const int64_t j = j0 + c; if (j >= q.B) return;
  const float alpha = expf(q.log_alpha[0]); const float q1t = z[0][0], q2t = z[1][0]; const float mn = q2t < q1t ? q2t : q1t;      // SacTargetOp
  const float yv = __fadd_rn(q.r[j], __fmul_rn(__fmul_rn(q.gamma, __fsub_rn(1.f, q.done[j] ? 1.f : 0.f)), __fsub_rn(mn, __fmul_rn(alpha, q.lp[j]))));
  q.y[j] = yv;
  const float invB = 1.f / (float)q.B, ww = q.w ? q.w[j] : 1.f;                                                                     // QHeadOp, twice
  { const float d = z[2][0] - yv; q.term1[j] = d * d * ww; q.dy1[j] = q.scale * (2.f * d * ww * invB); }
  { const float d = z[3][0] - yv; q.term2[j] = d * d * ww; q.dy2[j] = q.scale * (2.f * d * ww * invB); }
} };
struct CriticInfo2Op { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, const float* __restrict__ t1, const float* __restrict__ qv1, const float* __restrict__ t2, const float* __restrict__ qv2,
                                                                  const double* __restrict__ ssq, int64_t B, float* __restrict__ dinfo) {      // one block of 256: QHeadOp's block sums + CriticInfoOp
  __shared__ double red[4];
  double a1 = 0, b1 = 0, a2 = 0, b2 = 0;
  for (int64_t j = threadIdx.x; j < B; j += 256) { a1 += (double)t1[j]; b1 += (double)qv1[j]; a2 += (double)t2[j]; b2 += (double)qv2[j]; }
  a1 = block_sum256(a1, red); b1 = block_sum256(b1, red); a2 = block_sum256(a2, red); b2 = block_sum256(b2, red);
  if (threadIdx.x != 0) return;
  ssq_finalize(ssq);
  dinfo[CRUX_INFO_LOSS] = (float)(0.5 * (a1 / (double)B) + 0.5 * (a2 / (double)B));
  dinfo[CRUX_INFO_Q1AVG] = (float)(b1 / (double)B); dinfo[CRUX_INFO_Q2AVG] = (float)(b2 / (double)B);
  dinfo[CRUX_INFO_GRAD_NORM] = (float)sqrt(ssq[0]);
} };

// ---- sac_actor_loss head ------------------------------------------------------------------------------------------------------------------------------------------------
struct SacActorArgs { TileSet q1, q2; const float* lp; const float* log_alpha; int32_t K, B; float* dy1; float* dy2; float* term; };
struct SacActorTileOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, SacActorArgs q) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c = lane & 15, g = lane >> 4; const int j0 = (int)bid_ << 4;
  const TileSet sets[2] = {q.q1, q.q2}; f32x4 z[2];
  tile_splitk<2, false>(sets, 1, q.K, q.B, j0, df_lds, z);
  if (wv > 0 || g != 0) return;
  const int64_t j = j0 + c; if (j >= q.B) return;
  const float alpha = expf(q.log_alpha[0]), invB = 1.f / (float)q.B; const float q1 = z[0][0], q2 = z[1][0];      // ActorHeadOp
  const bool second = q2 < q1; const float mn = second ? q2 : q1;
  q.term[j] = alpha * q.lp[j] - mn; q.dy1[j] = second ? 0.f : -invB; q.dy2[j] = second ? -invB : 0.f;
} };
struct ActorInfo2Op { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, const float* __restrict__ term, const float* __restrict__ lp, const double* __restrict__ ssq, int64_t B, float* __restrict__ dinfo) {
  __shared__ double red[4];
  double sl = 0, slp = 0;
  for (int64_t j = threadIdx.x; j < B; j += 256) { sl += (double)term[j]; slp += (double)lp[j]; }
  sl = block_sum256(sl, red); slp = block_sum256(slp, red);
  if (threadIdx.x != 0) return;
  ssq_finalize(ssq);
  dinfo[CRUX_INFO_LOSS] = (float)(sl / (double)B); dinfo[CRUX_INFO_ENTROPY] = (float)(-(slp / (double)B)); dinfo[CRUX_INFO_GRAD_NORM] = (float)sqrt(ssq[0]);
} };

// ---- the critics' input gradients (first layer, dX = W1' dZ1) + the reverse pass through exploration() ----------------------------------------------------------------
struct CriticDxArgs { TileSet c1, c2; const float* sa; const float* mu; const float* eps; const float* ls; const float* log_alpha; int32_t od, ad, K, B; float* dmu; float* dls; };
struct CriticDxActorGradTileOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, CriticDxArgs q) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c = lane & 15, g = lane >> 4; const int j0 = (int)bid_ << 4;
  const TileSet sets[2] = {q.c1, q.c2}; f32x4 z[2];
  tile_splitk<2, true>(sets, q.od + q.ad, q.K, q.B, j0, df_lds, z);      // z[n][r] = d Qn / d sa[4 g + r] of sample j0 + c
  float* T = df_lds + TILE_PART_FLOATS(2);                                // [2][16 rows][16 samples]: the action rows may sit in another lane group than the sample's lane 0..15
  if (wv == 0) {
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) T[(n * 16 + 4 * g + r) * 16 + c] = z[n][r]; }
  __syncthreads();
  const int64_t i = (int64_t)j0 * q.ad + threadIdx.x;                     // ActorGradOp: one thread per (sample, action dimension) of the tile
  if ((int)threadIdx.x >= 16 * q.ad) return;
  const int64_t j = i / q.ad; const int d = (int)(i - j * q.ad); if (j >= q.B) return;
  const int cs = (int)(j - j0), row = q.od + d;
  const float alpha = expf(q.log_alpha[0]); const float clp = alpha * (1.f / (float)q.B);
  const float sg = expf(q.ls[d]), s2 = sg * sg, a = q.sa[j * (q.od + q.ad) + q.od + d], df = a - q.mu[i];
  const float dq = T[(0 * 16 + row) * 16 + cs] + T[(1 * 16 + row) * 16 + cs];      // only the selected network's entry is non-zero
  const float abar = clp * (-(df / s2)) + dq;
  q.dmu[i] = clp * (df / s2) + abar;
  q.dls[i] = clp * ((df * df) / s2 - 1.f) + abar * (q.eps[i] * sg);
} };

// ---- DQN family: both output layers + dqn_target | softq_target + td_loss head + td_error + update_priorities! of the tile's samples ------------------------------------------
// (rl/dqn.jl:4-6, rl/softq.jl:1-13, utils.jl:76-87,112, experience_buffer.jl:290-301). update_priorities! distributes over the tiles: a sample's new priority needs its own
// td error and the id list only (a repeated id keeps its LAST value: the id list decides, as in PerUpdateOp), max / min priority are atomics.
struct DqnTdArgs { TileSet qt, q; const float* r; const uint8_t* done; const uint8_t* a; const float* w; float gamma, softq_alpha; int32_t nout, K, B;
                   float* y; float* dy; float* err; float* term; float* qsel;
                   float* pr; float* pminmax; const int64_t* ids; float per_alpha; int32_t per; };
struct DqnTdTileOp { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, DqnTdArgs q) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c = lane & 15, g = lane >> 4; const int j0 = (int)bid_ << 4;
  int64_t* ids_s = (int64_t*)(df_lds + TILE_PART_FLOATS(2));
  if (q.per) for (int64_t j = threadIdx.x; j < q.B; j += 256) ids_s[j] = q.ids[j];      // (visible after the helper's workgroup barrier)
  const TileSet sets[2] = {q.qt, q.q}; f32x4 z[2];
  // the sample's reward, done flag, action row and weight are loaded BEFORE the output layers (one round trip together with their operands instead of one more after them)
  const int64_t jp = j0 + c < q.B ? j0 + c : q.B - 1;
  const float r_p = q.r[jp]; const uint8_t done_p = q.done[jp]; const float w_p = q.w ? q.w[jp] : 1.f;
  uint8_t a_p[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) a_p[k] = q.a[jp * q.nout + (k < q.nout ? k : 0)];
  tile_splitk<2, false>(sets, q.nout, q.K, q.B, j0, df_lds, z);
  if (wv > 0) return;
  bool sorted_ids = true;
  if (q.per) { bool ok = true; for (int64_t j = lane; j + 1 < q.B; j += 64) ok = ok && ids_s[j] <= ids_s[j + 1]; sorted_ids = __ballot(!ok) == 0ull; }      // stratified samples arrive in ascending order
  const int64_t j = j0 + c; const bool mine = g == 0 && j < q.B;      // nout <= 4: lane c of group 0 holds the nout outputs of both networks for sample j0 + c
  int wmax = (int)0x80000000, wmin = 0x7fffffff;
  if (mine) {
    float yv;
    if (q.softq_alpha > 0.f) { const float al = q.softq_alpha;      // SoftqTargetOp
      float mx = z[0][0] / al; for (int k = 1; k < q.nout; ++k) { const float v = z[0][k] / al; mx = v > mx ? v : mx; }
      float sum = 0.f; for (int k = 0; k < q.nout; ++k) sum = sum + expf(z[0][k] / al - mx);
      const float lse = mx + logf(sum); const float sv = al * lse;
      const float nd = 1.f - (done_p ? 1.f : 0.f); const float gn = q.gamma * nd; const float t = gn * sv; yv = r_p + t; }
    else {                                                             // DqnTargetOp
      float mx = z[0][0]; for (int k = 1; k < q.nout; ++k) mx = z[0][k] > mx ? z[0][k] : mx;
      const float nd = 1.f - (done_p ? 1.f : 0.f); const float gn = q.gamma * nd; const float t = gn * mx; yv = r_p + t; }
    q.y[j] = yv;
    const float invB = 1.f / (float)q.B; const uint8_t* a = a_p;      // TdHeadOp
    float Q = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (k < q.nout) Q += z[1][k] * (a[k] ? 1.f : 0.f);
    const float d = Q - yv; const float ww = w_p;
    if (q.err) q.err[j] = fabsf(d);
    q.term[j] = d * d * ww; q.qsel[j] = Q;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (k < q.nout) q.dy[j * q.nout + k] = a[k] ? 2.f * d * ww * invB : 0.f;
    if (q.per) {                                                        // PerUpdateOp on (ids[j], |d|)
      const float vf = __fadd_rn(fabsf(d), 1.1920928955078125e-07f); const double val = (double)vf; const int64_t me = ids_s[j];
      bool later = false;
      if (sorted_ids) later = j + 1 < q.B && ids_s[j + 1] == me;
      else for (int64_t t = j + 1; t < q.B; ++t) if (ids_s[t] == me) { later = true; break; }
      if (!later) q.pr[me] = (float)pow(val, (double)q.per_alpha);
      wmax = __float_as_int(vf); wmin = __float_as_int(vf); }
  }
  if (q.per) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { wmax = max(wmax, __shfl_xor(wmax, o, 64)); wmin = min(wmin, __shfl_xor(wmin, o, 64)); }
    if (lane == 0 && wmax != (int)0x80000000) { atomicMax((int*)&q.pminmax[0], wmax); atomicMin((int*)&q.pminmax[1], wmin); } }
} };
struct TdInfo2Op { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, const float* __restrict__ term, const float* __restrict__ qsel, const double* __restrict__ ssq, int64_t B, float* __restrict__ dinfo) {
  __shared__ double red[4];
  double sl = 0, sq = 0;
  for (int64_t j = threadIdx.x; j < B; j += 256) { sl += (double)term[j]; sq += (double)qsel[j]; }      // TdHeadOp's block sums
  sl = block_sum256(sl, red); sq = block_sum256(sq, red);
  if (threadIdx.x != 0) return;
  ssq_finalize(ssq);
  dinfo[CRUX_INFO_LOSS] = (float)(sl / (double)B); dinfo[2] = (float)(sq / (double)B); dinfo[CRUX_INFO_GRAD_NORM] = (float)sqrt(ssq[0]);
} };
