// dense_fused.h -- LDS-staged block kernels of the dense engine (round 4): fewer, fatter launches for the 256-wide off-policy learners (C3 / C4).
//
// The wave-per-tile GEMM of dense.hip (Gemm16) gives every Dense layer its own launch and re-reads both operands from L2 for every 16x16 tile; an epoch of
// DQN + PER / SAC was 9 / 26 dependent launches of >= 5.2 us each. The ops here fuse neighbouring layers so that a three-layer pullback needs fewer of them:
//   Fwd12Op      layers 0 and 1 of a Chain(Dense...) whose input is narrow (in <= 32: observations, vcat(s, a)): a workgroup owns a 32 x 32 block of layer 1's
//                output, re-evaluates the layer-0 panel it needs (K = in is tiny) straight into LDS and stages the weight panel through LDS.
//   Wgrad2Op     dW = scale * dZ X' and db for a wide layer: 32 x 32 output block per workgroup, both operand panels staged through LDS once.
//   Dgrad2W1Op   dX = act'(X) .* (W' dZ) of layer 1 for 16 features x ALL samples, then -- without leaving the workgroup -- layer 0's dW, db from it.
// Reference: Zygote's pullback of Chain(Dense...) (src/training.jl:16-18 through src/utils.jl:76-96, src/model_free/rl/sac.jl:34-52).
//
// ARITHMETIC IS THAT OF Gemm16: v_mfma_f32_16x16x4_f32 is a sequential fma chain over its four k slices (tools/mfma_chain_test.hip: 0 mismatches against
// fmaf in k-ascending order), and every reduction here feeds k to the chain in Gemm16's order -- 16-groups ascending, inside a group instruction r = 0..3 carries
// k = base + 4 g + r in lane group g -- with Gemm16's quarter rule for K >= 128 (four partial chains of kper = ceil16(ceil(K / 4)), combined ((q0 + q1) + q2) + q3).
// Results are therefore bit-identical to the per-layer launches (CRUX_DENSE_FUSED=0 selects those; tests/test_gpu_round4.py compares).
#pragma once
#include "common.h"

#define DF_LDS_FLOATS 10560                   // Dgrad2W1Op: (16 + 64) rows x 132 floats; Wgrad2Op: 128 x 36 x 2
__shared__ float df_lds[DF_LDS_FLOATS];       // ONE buffer for all fused ops (file scope: the phase kernel allocates it once, not once per op)

__device__ __forceinline__ int df_kper(int K) { return K >= 128 ? ((((K + 3) >> 2) + 15) & ~15) : K; }

// ---- layers 0 + 1 forward -------------------------------------------------------------------------------------------------------------------
struct Fwd12Args { const float* W1; const float* b1; const float* W2; const float* b2; const float* x; float* H1; float* H2; int32_t in0, out1, out2, B, act1, act2; };
struct Fwd12Op { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, Fwd12Args q) {
  // One 16 x 16 tile of layer 1's output per workgroup, Gemm16's split-K form: wave w walks quarter w of K = out1 and wave 0 adds the quarters in wave order. The
  // layer-0 values a wave needs are the out1 / 4 features of ITS quarter for the tile's 16 samples: it evaluates them itself (K = in0 <= 32: 4-8 MFMAs per 16 features),
  // and the D-layout result [feature 4 g + r][sample c] IS the B operand of the layer-1 MFMA for that 16-group -- no LDS panel, no workgroup barrier before the combine.
  // Latency, not bandwidth, bounds these blocks (a global load is a 1-2 us round trip here: the weights were just rewritten by Adam on other XCDs): every global
  // load of the wave -- observations, layer-0 weight fragments and biases, its layer-1 weight fragments -- is issued before the first use.
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int tI = q.out2 >> 4; const int bi = (int)bid_ % tI, bj = (int)bid_ / tI; const int i0 = bi << 4, j0 = bj << 4;
  const int K = q.out1, kper = K >> 2, ng = kper >> 4;          // df_fwd12_ok: K in {128, 192, 256} -> kper = df_kper(K) = K / 4, 2..4 sixteen-groups per quarter
  const int s = j0 + c; const bool vs = s < q.B; const bool u1 = q.in0 > 16; const int sc = vs ? s : q.B - 1;
  float xb[2][4];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int k0 = 16 * u + 4 * g + r; xb[u][r] = q.x[(k0 < q.in0 ? k0 : 0) + (int64_t)q.in0 * sc] * (k0 < q.in0 ? 1.f : 0.f); }
  // (every load is UNCONDITIONAL, from a clamped address; slots past the end are zeroed by a multiplication with 0 / 1 -- a select lets the compiler sink the load into a
  //  branch followed by s_waitcnt vmcnt(0), which serialised the loads of this kernel one round trip after the other. x * 1 is x; a zero on ONE operand of a product is enough.)
  float w1f[4][2][4], w2f[4][4]; f32x4 b1v[4];
#pragma unroll
  for (int gi = 0; gi < 4; ++gi) { const int fb = wv * kper + 16 * (gi < ng ? gi : 0);      // (groups past the quarter re-read group 0 and are never used)
    b1v[gi] = *(const f32x4*)(q.b1 + fb + 4 * g);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int k0 = 16 * u + 4 * g + r; w1f[gi][u][r] = q.W1[fb + c + (int64_t)q.out1 * (k0 < q.in0 ? k0 : 0)]; }      // (x carries the zero for k0 >= in0)
#pragma unroll
    for (int r = 0; r < 4; ++r) w2f[gi][r] = q.W2[i0 + c + (int64_t)q.out2 * (fb + 4 * g + r)]; }
  f32x4 b2v = {0.f, 0.f, 0.f, 0.f}; if (wv == 0) b2v = *(const f32x4*)(q.b2 + i0 + 4 * g);
  f32x4 pa = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int gi = 0; gi < 4; ++gi) { if (gi < ng) { const int fb = wv * kper + 16 * gi;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1f[gi][0][r], xb[0][r], a0, 0, 0, 0);
    if (u1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1f[gi][1][r], xb[1][r], a0, 0, 0, 0); }
    f32x4 hv;
#pragma unroll
    for (int r = 0; r < 4; ++r) hv[r] = crux_act(q.act1, a0[r] + b1v[gi][r]);
    if (bi == 0 && vs) *(f32x4*)(q.H1 + fb + 4 * g + (int64_t)q.out1 * s) = hv;      // the cached activation (one writer per sample tile)
#pragma unroll
    for (int r = 0; r < 4; ++r) pa = __builtin_amdgcn_mfma_f32_16x16x4f32(w2f[gi][r], hv[r], pa, 0, 0, 0); } }
  float* part = df_lds;
  if (wv > 0) { float* p = part + ((wv - 1) * 64 + lane) * 4; p[0] = pa[0]; p[1] = pa[1]; p[2] = pa[2]; p[3] = pa[3]; }
  __syncthreads();
  if (wv > 0) return;
#pragma unroll
  for (int w = 0; w < 3; ++w) { const float* p = part + (w * 64 + lane) * 4; pa[0] += p[0]; pa[1] += p[1]; pa[2] += p[2]; pa[3] += p[3]; }
  if (vs) { f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = crux_act(q.act2, pa[r] + b2v[r]);
    *(f32x4*)(q.H2 + i0 + 4 * g + (int64_t)q.out2 * s) = o; }
} };
__global__ __launch_bounds__(256) void k_fwd12(Fwd12Args q) { Fwd12Op::run(blockIdx.x, gridDim.x, q); }
static inline bool df_fwd12_ok(const NetDesc& nd) {
  return nd.L >= 2 && nd.dims[0] <= 32 && (nd.dims[1] == 128 || nd.dims[1] == 192 || nd.dims[1] == 256) && nd.dims[2] >= 16 && (nd.dims[2] & 15) == 0;
}
static inline unsigned df_fwd12_blocks(const NetDesc& nd, int64_t B) { return (unsigned)((nd.dims[2] >> 4) * ((B + 15) >> 4)); }

// ---- weight gradient of a wide layer: dW[i, k] = scale * sum_s dZ[i, s] X[k, s]; db[i] = scale * sum_s dZ[i, s] ----------------------------------
// dZ of the layer under a narrow output layer (out3 <= 4), formed where it is consumed instead of by a launch of its own: dZ[o, s] = act'(H[o, s]) * sum_q W3[q, o] dZ3[q, s].
// The sum is Gemm16's chain for K = out3 <= 4 (one MFMA slice per q, lane group 0 only): fmaf in q-ascending order from +0. W3 == nullptr: the operand is dZ itself.
struct DzSrc { const float* W3; const float* dZ3; int32_t out3, act; };
template <int O3> __device__ __forceinline__ float df_dz(const DzSrc& z, float h, const float (&w)[O3], const float (&d)[O3]) {
  float v = 0.f;
#pragma unroll
  for (int q = 0; q < O3; ++q) v = fmaf(w[q], d[q], v);      // (w is zero past out3: the term adds +-0)
  return crux_act_grad(z.act, h, v);
}
struct Wgrad2Args { const float* dZ; const float* X; float* dW; float* db; float scale; int32_t out, in, B; DzSrc z; int32_t* nf; };      // z.W3 != nullptr: dZ points at H (the layer's own output) and the gradient is formed on the fly
struct Wgrad2Op { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, Wgrad2Args q) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int tI = q.out >> 5; const int bi = (int)bid_ % tI, bk = (int)bid_ / tI; const int I0 = bi << 5, K0 = bk << 5;
  const int K = q.B; const bool quartered = K >= 128; const int kper = df_kper(K);
  const int K16 = (K + 15) & ~15;
  const int half_k = quartered ? 2 * kper : K16, nhalf = quartered ? 2 : 1;
  float* As = df_lds; float* Bs = df_lds + 128 * 36;
  const int mtw = wv & 1, ntw = wv >> 1;
  const bool want_rowsum = bk == 0 && ntw == 0 && q.db != nullptr;
  f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}; float rows[4] = {0.f, 0.f, 0.f, 0.f};
  // both halves of both panels are fetched before the first use (one memory round trip per block)
  const int i4 = (threadIdx.x & 7) << 2, sr = threadIdx.x >> 3;
  float w3[4][4], d3[2][4][4], mks[2][4];
  const float* W3 = q.z.W3 ? q.z.W3 : q.X; const float* D3 = q.z.W3 ? q.z.dZ3 : q.X; const int o3 = q.z.W3 ? q.z.out3 : 1;      // (no folding: harmless loads of valid addresses, results unused)
  const bool o4 = o3 == 4;      // crux_dense_bwd_fused3: out3 is 1 or 4
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) w3[e][qq] = 0.f;
  if (o4) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { const f32x4 t = *(const f32x4*)(W3 + 4 * (I0 + i4 + e));
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) w3[e][qq] = t[qq]; }
  } else { const f32x4 t = *(const f32x4*)(W3 + I0 + i4);
#pragma unroll
    for (int e = 0; e < 4; ++e) w3[e][0] = t[e]; }
  f32x4 ta[2][4], tb[2][4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int p = 0; p < 4; ++p) { const int sl = sr + 32 * p, s = h * half_k + sl; const bool v = h < nhalf && sl < half_k && s < K; const int sc = v ? s : 0;      // clamped address + select
      const float mk = v ? 1.f : 0.f;      // unconditional loads; rows past the end are zeroed on the dZ side by a multiplication (see Fwd12Op)
      ta[h][p] = *(const f32x4*)(q.dZ + (I0 + i4) + (int64_t)q.out * sc); tb[h][p] = *(const f32x4*)(q.X + (K0 + i4) + (int64_t)q.in * sc); mks[h][p] = mk;
      d3[h][p][1] = d3[h][p][2] = d3[h][p][3] = 0.f;
      if (o4) { const f32x4 t = *(const f32x4*)(D3 + 4 * sc);
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) d3[h][p][qq] = t[qq]; }
      else d3[h][p][0] = D3[sc]; }
  // (the loads above are all in flight before the first of them is used: the folded gradient is formed in a second pass)
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if (q.z.W3) {
#pragma unroll
        for (int e = 0; e < 4; ++e) ta[h][p][e] = df_dz<4>(q.z, ta[h][p][e], w3[e], d3[h][p]); }
      ta[h][p] = ta[h][p] * mks[h][p]; }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (h >= nhalf) break;
    const int sh0 = h * half_k; int sh1 = sh0 + half_k; if (sh1 > K16) sh1 = K16; const int HS = sh1 - sh0;
    if (HS <= 0) break;
    if (h) __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) { const int sl = sr + 32 * p; if (sl < HS) { *(f32x4*)(As + sl * 36 + i4) = ta[h][p]; *(f32x4*)(Bs + sl * 36 + i4) = tb[h][p]; } }
    __syncthreads();
    const int nq = quartered ? 2 : 1;
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      if (qq >= nq) break;
      const int qi = 2 * h + qq; const int kb = quartered ? qi * kper : 0; int ke = quartered ? kb + kper : K16; if (ke > K16) ke = K16;
      f32x4 pa = {0.f, 0.f, 0.f, 0.f}; float prow = 0.f;
      for (int k16 = kb; k16 < ke; k16 += 16) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int row = (k16 - sh0 + 4 * g + r) * 36; const float a = As[row + 16 * mtw + c], b = Bs[row + 16 * ntw + c];
          pa = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, pa, 0, 0, 0); if (want_rowsum) prow += a; }
      }
      acc[qi] = pa; rows[qi] = prow;
    }
  }
  f32x4 t = acc[0]; float rowsum = rows[0];
  if (quartered) {
#pragma unroll
    for (int p = 1; p < 4; ++p) { t[0] += acc[p][0]; t[1] += acc[p][1]; t[2] += acc[p][2]; t[3] += acc[p][3]; rowsum += rows[p]; } }
  bool bad = false;
  if (want_rowsum) { rowsum += __shfl_xor(rowsum, 16, 64); rowsum += __shfl_xor(rowsum, 32, 64); if (g == 0) { const float bv = q.scale * rowsum; q.db[I0 + 16 * mtw + c] = bv; bad = bv != bv; } }
  const int i = I0 + 16 * mtw + 4 * g, kc = K0 + 16 * ntw + c;
  f32x4 o;
#pragma unroll
  for (int r = 0; r < 4; ++r) { o[r] = t[r] * q.scale; bad = bad || o[r] != o[r]; }
  *(f32x4*)(q.dW + i + (int64_t)q.out * kc) = o;
  if (q.nf && bad) atomicOr((int*)q.nf, 1);
} };
__global__ __launch_bounds__(256) void k_wgrad2(Wgrad2Args q) { Wgrad2Op::run(blockIdx.x, gridDim.x, q); }

// ---- data gradient through layer 1, then layer 0's weight gradient, for 16 layer-0 features x ONE QUARTER of the samples -----------------------------------------
//   dH[f, s] = sum_o W2[o, f] dZ2[o, s]   (K = out2)      dZ1 = act0'(H1) .* dH      dW1[f, q] = scale * sum_s dZ1[f, s] x[q, s], db1[f] = scale * sum_s dZ1[f, s]   (K = B)
// Gemm16 defines layer 0's weight gradient (K = B >= 128) as four quarter chains over the samples, combined ((q0 + q1) + q2) + q3. A workgroup here owns 16 features x
// the samples of ONE quarter (kper(B) = 32 / 48 / 64 of them: 2-4 MFMA tiles, one per wave): 64 workgroups instead of 16, a quarter of the serial MFMA work each. It
// leaves its quarter's partial chain (and the per-lane-group partial row sums of db) in `part`; the combination -- the same additions in the same order -- is done by
// whoever reads the gradient next: Sumsq2Op (sac.hip), which every train! step runs right after the pullback, forms and stores dW1 / db1 from the four partials as it
// sums the squares. Both operand panels are staged through LDS with row-contiguous 16-byte loads (a lane-per-sample fragment load touches sixteen half-used cache
// lines per instruction: the texture path, not latency, bounded the first version of this kernel), all issued before the first use.
struct Dgrad2Args { const float* W2; const float* dZ2; const float* H1; const float* x; float* part; float* dZ1; int32_t in0, out1, out2, B, act0, want_g; DzSrc z; int32_t* nf; };      // z.W3 != nullptr: dZ2 points at H2 (see Wgrad2Args)
#define DF_PART_STRIDE(in0) ((in0) + 4)      // per feature: in0 partial dW entries + 4 partial row sums (lane groups g = 0..3)
// O3: the widest output layer whose data gradient the instantiation can fold in (DzSrc): 1 (critics, value heads; also "no folding") or 4. Two instantiations because the
// folded operands live in registers until the panels are written: 16 of them at O3 = 1, 64 at O3 = 4 -- and the register count decides how many workgroups share a CU.
template <int O3> struct Dgrad2W1OpT { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, Dgrad2Args q) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int nF = q.out1 >> 4; const int F0 = ((int)bid_ % nF) << 4, sb = (int)bid_ / nF;      // sb = the quarter of the samples
  const int K = q.out2, kper = K >> 2, half_k = K >> 1;                // df_bwd_ok: K in {128, 192, 256}
  const int kperb = df_kper(q.B); const int S0 = sb * kperb; int S1 = S0 + kperb; if (S1 > q.B) S1 = q.B; const int S = S1 - S0;      // B in [128, 256]: kperb in {32, 48, 64}
  const int HP = half_k + 4;                                           // panel row stride: (stride / 4) odd -> conflict-free ds_read_b128
  float* Ws = df_lds; float* Zt = df_lds + 16 * 132;                   // Ws[16][half_k + 4], Zt[<= 64][half_k + 4]
  // all global loads first: both halves of the weight rows (16 x K) and of the quarter's dZ2 columns (<= 64 x K), H1 of this wave's tile, x of the quarter (wave 0)
  const int col4 = (threadIdx.x & 31) << 2, row8 = threadIdx.x >> 5;   // 32 threads per half row of <= 128 floats, 8 rows per pass
  f32x4 wl[2][2], zl[2][8];
  float w3[2][4][O3], d3[8][O3];      // folded output layer (DzSrc): W3[q, o] of this thread's four features per half, dZ3[q, s] of its eight samples -- 16-byte loads
  { const float* W3 = q.z.W3 ? q.z.W3 : q.W2; const float* D3 = q.z.W3 ? q.z.dZ3 : q.W2;      // (no folding: harmless loads of valid addresses, results unused)
    const int ccz = col4 < half_k ? col4 : 0;
    if (O3 == 1) {
#pragma unroll
      for (int h = 0; h < 2; ++h) { const f32x4 t = *(const f32x4*)(W3 + h * half_k + ccz);
#pragma unroll
        for (int e = 0; e < 4; ++e) w3[h][e][0] = t[e]; }
#pragma unroll
      for (int p = 0; p < 8; ++p) { const int sl = row8 + 8 * p; d3[p][0] = D3[sl < S ? S0 + sl : 0]; }
    } else {      // out3 == 4 (crux_dense_bwd_fused3)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const f32x4 t = *(const f32x4*)(W3 + 4 * (h * half_k + ccz + e));
#pragma unroll
          for (int qq = 0; qq < O3; ++qq) w3[h][e][qq] = t[qq & 3]; }
#pragma unroll
      for (int p = 0; p < 8; ++p) { const int sl = row8 + 8 * p; const f32x4 t = *(const f32x4*)(D3 + 4 * (sl < S ? S0 + sl : 0));
#pragma unroll
        for (int qq = 0; qq < O3; ++qq) d3[p][qq] = t[qq & 3]; }
    } }
#pragma unroll
  for (int h = 0; h < 2; ++h) { const int cc = col4 < half_k ? col4 : 0;
#pragma unroll
    for (int p = 0; p < 2; ++p) wl[h][p] = *(const f32x4*)(q.W2 + h * half_k + cc + (int64_t)q.out2 * (F0 + row8 + 8 * p));
#pragma unroll
    for (int p = 0; p < 8; ++p) { const int sl = row8 + 8 * p; const int s = sl < S ? S0 + sl : 0; zl[h][p] = *(const f32x4*)(q.dZ2 + h * half_k + cc + (int64_t)q.out2 * s); } }      // (rows past the quarter re-read sample 0 and are never stored)
  const int st = S0 + 16 * wv + c; const bool vt = 16 * wv < S, vs = vt && st < S1;      // this wave's tile and sample
  const f32x4 y = *(const f32x4*)(q.H1 + F0 + 4 * g + (int64_t)q.out1 * (vs ? st : 0));
  float xf[1][4][4];      // (df_bwd_ok: in0 <= 16 -- one input tile)
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) xf[0][u][r] = 0.f;
  if (wv == 0 && q.want_g) {      // (one wave forms the quarter's partial chain; inside the branch the loads are still back to back)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int qc = c, s = S0 + 16 * u + 4 * g + r; const bool v = qc < q.in0 && s < S1;
        xf[0][u][r] = q.x[(v ? qc : 0) + (int64_t)q.in0 * (v ? s : 0)] * (v ? 1.f : 0.f); } }      // unconditional load, zeroed by a multiplication (see Fwd12Op)
  if (q.z.W3) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int e = 0; e < 4; ++e) zl[h][p][e] = df_dz<O3>(q.z, zl[h][p][e], w3[h][e], d3[p]); }
  f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (h) __syncthreads();
    if (col4 < half_k) {
#pragma unroll
      for (int p = 0; p < 2; ++p) *(f32x4*)(Ws + (row8 + 8 * p) * HP + col4) = wl[h][p];
#pragma unroll
      for (int p = 0; p < 8; ++p) { const int sl = row8 + 8 * p; if (sl < S) *(f32x4*)(Zt + sl * HP + col4) = zl[h][p]; } }
    __syncthreads();
    if (vt) {
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) { const int qi = 2 * h + qq; f32x4 pa = {0.f, 0.f, 0.f, 0.f};
        for (int k16 = qq * kper; k16 < (qq + 1) * kper; k16 += 16) {
          const f32x4 av = *(const f32x4*)(Ws + c * HP + k16 + 4 * g), bv = *(const f32x4*)(Zt + (16 * wv + c) * HP + k16 + 4 * g);
#pragma unroll
          for (int r = 0; r < 4; ++r) pa = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], bv[r], pa, 0, 0, 0); }
        acc[qi] = pa; } }
  }
  f32x4 t = acc[0];
#pragma unroll
  for (int p = 1; p < 4; ++p) { t[0] += acc[p][0]; t[1] += acc[p][1]; t[2] += acc[p][2]; t[3] += acc[p][3]; }
  f32x4 dz;
#pragma unroll
  for (int r = 0; r < 4; ++r) dz[r] = crux_act_grad(q.act0, y[r], t[r]);
  if (q.dZ1 && vs) *(f32x4*)(q.dZ1 + F0 + 4 * g + (int64_t)q.out1 * st) = dz;
  if (!q.want_g) return;
  __syncthreads();                                                     // the panels are consumed: Zs aliases them
  float* Zs = df_lds; const int ZP = 64 + 4;                           // Zs[16][64 + 4]: dZ1 of the quarter, zero past its end
  if (vt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) Zs[(4 * g + r) * ZP + 16 * wv + c] = vs ? dz[r] : 0.f; }
  else {
#pragma unroll
    for (int r = 0; r < 4; ++r) Zs[(4 * g + r) * ZP + 16 * wv + c] = 0.f; }
  __syncthreads();
  if (wv > 0) return;
  // the quarter's partial chain of layer 0's weight gradient: M = 16, N = in0, K = the quarter's samples (Gemm16: 16-groups ascending)
  float* prow_out = q.part + (int64_t)sb * q.out1 * DF_PART_STRIDE(q.in0);
#pragma unroll
  for (int tq = 0; tq < 1; ++tq) { if (16 * tq < q.in0) {
    f32x4 pa = {0.f, 0.f, 0.f, 0.f}; float prow = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) { if (16 * u < kperb) { const f32x4 av = *(const f32x4*)(Zs + c * ZP + 16 * u + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) { pa = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], xf[tq][u][r], pa, 0, 0, 0); if (tq == 0) prow += av[r]; } } }
    const int qc = 16 * tq + c;
    bool odd = !(fabsf(prow) <= 3.4028234664e38f);                     // a partial that is not finite: the final sum may be NaN (AdamSelfOp then looks at the sums themselves)
    if (qc < q.in0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { prow_out[(int64_t)(F0 + 4 * g + r) * DF_PART_STRIDE(q.in0) + qc] = pa[r]; odd = odd || !(fabsf(pa[r]) <= 3.4028234664e38f); } }
    if (q.nf && odd) atomicOr((int*)q.nf + 1, 1);
    if (tq == 0) prow_out[(int64_t)(F0 + c) * DF_PART_STRIDE(q.in0) + q.in0 + g] = prow;      // lane (c, g): the partial row sum of feature F0 + c over this quarter's k = 16 u + 4 g + r
  } }
} };
struct Dgrad2W1Op { static __device__ __forceinline__ void run(const unsigned bid_, const unsigned nb_, Dgrad2Args q) { if (q.z.W3 && q.z.out3 > 1) Dgrad2W1OpT<4>::run(bid_, nb_, q); else Dgrad2W1OpT<1>::run(bid_, nb_, q); } };      // (the op type the recorder packs; stand-alone launches)
__global__ __launch_bounds__(256) void k_dgrad2w1(Dgrad2Args q) { Dgrad2W1Op::run(blockIdx.x, gridDim.x, q); }
// the pair (Wgrad2Op on layer 1, Dgrad2W1Op through layer 1 into layer 0) applies to: a narrow input, layer widths in whole 32-blocks, K = out2 in {128, 192, 256}, 128 <= B <= 256
static inline bool df_bwd_ok(const NetDesc& nd, int64_t B) {
  return nd.L >= 2 && nd.dims[0] <= 16 && nd.dims[1] >= 32 && (nd.dims[1] & 31) == 0 && (nd.dims[2] == 128 || nd.dims[2] == 192 || nd.dims[2] == 256) && B >= 128 && B <= 256;
}
