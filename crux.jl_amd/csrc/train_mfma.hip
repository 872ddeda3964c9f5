// train_mfma.hip -- persistent batch_train! / train! kernel for the IN->64->64->OUT MLP family on gfx950 f32 MFMA.
//
// Reference semantics: src/training.jl:13-55 (train!, batch_train!), src/model_free/rl/ppo.jl:4-21,59-60, Flux Adam.
// One workgroup (4 waves, one per SIMD, up to 512 VGPRs each) runs the whole epochs x minibatches loop. HBM is touched
// only for the minibatch rows (gathered through the composed shuffle order and prefetched one step ahead) and for one
// info row per epoch; parameters, Adam moments, activations and gradients stay in registers / LDS.
//
// All GEMMs use v_mfma_f32_16x16x4_f32 (exact f32, k-ordered fma chain). Lane l: c = l&15, g = l>>4.
//   A operand: lane holds A[i=c][k=g];  B operand: B[k=g][j=c];  C/D reg r: D[row=4g+r][col=c].
//
// Forward/backward chain is data-parallel over waves: wave w owns samples [32w, 32w+32) as two 16-column tiles.
// Orientation trick (no LDS round trip between layers): a layer output computed as D[feature][sample] ("C orientation")
// holds, in reg r of tile m, the element [feature 16m+4g+r][sample c]. Used as a B operand this is B[k=g][j=c] with the
// contraction index PERMUTED to feature 16m+4g+r, which is legal because the weight fragment (A operand) is fetched from
// LDS in the same permuted order. Used as an A operand it is A[i=c -> sample][k=g -> feature] and produces the next
// result in "R orientation" D[sample][feature] (used for dH1).
//   forward (C):  H1 = act(W1 X + b1); H2 = act(W2 H1 + b2)                     [MFMA, B operand = previous D registers]
//   layer 3, loss head, dZ2 = act'(H2) .* (W3^T dz), dW3, db3                     [VALU: OUT <= 16 would waste an MFMA tile]
//   dH1 (R) = dZ2 (C regs as A) x W2 ; dZ1 (R) = act'(H1 R) .* dH1                 [MFMA]
//   dW1, db1, db2 partial over the wave's 32 samples                              [MFMA / VALU] -> small per-wave partials
// Weight gradient of the 64x64 layer is model-parallel over waves: H1 and dZ2 are exchanged as [feature][sample] tiles
// in LDS, then wave w computes rows [16w,16w+16) of dW2 over ALL 128 samples (A = dZ2 tile rows, B = H1 tiles) and
// applies Adam to those rows in registers -- theta, m, v of W2 (89 % of the parameters) never leave the owning wave's
// VGPRs; only the updated theta is republished to the LDS masters the next forward pass reads.
#include "train_args.h"

#include "mfma_helpers.h"

template <int IN, int OUT>
struct MfLayout {
  static constexpr int KS0 = (IN + 3) / 4;          // k-steps of layer 1
  static constexpr int IP = KS0 * 4;                // padded input width
  static constexpr int JT = (IN + 15) / 16;         // 16-column tiles of dW1
  static constexpr int XP = IP + 1;                 // staging row stride
  static constexpr int SCW = 4 + (OUT > 4 ? OUT : 4);
  // "small" parameters = everything except W2; index space s in [0, NS)
  static constexpr int sW1 = 0, sB1 = MF_HID * IN, sB2 = sB1 + MF_HID, sW3 = sB2 + MF_HID, sB3 = sW3 + MF_HID * OUT, sEX = sB3 + OUT, NS = sEX + 16;
  // canonical (Flux.params) offsets
  static constexpr int cW1 = 0, cB1 = MF_HID * IN, cW2 = cB1 + MF_HID, cB2 = cW2 + MF_HID * MF_HID, cW3 = cB2 + MF_HID, cB3 = cW3 + MF_HID * OUT, cEX = cB3 + OUT;
  // per-wave small partial gradients
  static constexpr int W1ROWS = IP < 16 * JT ? IP : 16 * JT;  // rows of the dW1 partial that are kept (inputs 0..IP-1)
  static constexpr int pW1 = 0;                               // [i < W1ROWS][o] stride MF_LD
  static constexpr int pB1 = pW1 + W1ROWS * MF_LD;
  static constexpr int pB2 = pB1 + MF_HID;
  static constexpr int pW3 = pB2 + MF_HID;                    // [o][i]
  static constexpr int pMISC = pW3 + OUT * MF_HID;            // [48]: 7 stat sums, then db3[OUT], then dlogSigma[OUT]
  static constexpr int pST = pMISC, pB3 = pMISC + 7, pEX = pB3 + OUT;
  static constexpr int PART = ((pMISC + 48 + 3) / 4) * 4;
  static constexpr int NSP = ((NS + 3) / 4) * 4;
  // everything except the optional row-major copy of W2
  static constexpr int BASE = MF_HID * MF_LD + MF_HID * IP + 2 * MF_HID + OUT * MF_HID + 32 + 4 + 2 * NSP + 8 * MF_HID * MF_TLD + 4 * PART + 4 * 32 * XP + 4 * 32 * SCW + 16;
  // W2 is kept in two LDS layouts when it fits in the CU's 160 KB: W2R[o][i] gives the forward A fragments with b128 reads; without
  // it they are gathered from W2C with b32 reads (same values, 4x the read instructions).
  static constexpr bool HAS_W2R = BASE + MF_HID * MF_LD <= 40960;
  // LDS masters
  static constexpr int oW2R = 0;                              // W2R[o][i]
  static constexpr int oW2C = oW2R + (HAS_W2R ? MF_HID * MF_LD : 0);   // W2C[i][o]
  static constexpr int oW1R = oW2C + MF_HID * MF_LD;          // W1R[o][i<IP]
  static constexpr int oB1 = oW1R + MF_HID * IP;
  static constexpr int oB2 = oB1 + MF_HID;
  static constexpr int oW3R = oB2 + MF_HID;                   // W3R[o][i]
  static constexpr int oB3 = oW3R + OUT * MF_HID;
  static constexpr int oEX = oB3 + 16;
  static constexpr int oMS = ((oEX + 16 + 3) / 4) * 4;        // Adam moments of the small parameters
  static constexpr int oVS = oMS + NSP;
  // exchange tiles, one pair per wave
  static constexpr int oT1 = oVS + NSP;                       // H1  [f][s]
  static constexpr int oT2 = oT1 + 4 * MF_HID * MF_TLD;       // dZ2 [f][s]
  static constexpr int oPART = oT2 + 4 * MF_HID * MF_TLD;
  static constexpr int oXS = oPART + 4 * PART;                // 4 x [32][XP] minibatch observations
  static constexpr int oSC = oXS + 4 * 32 * XP;               // 4 x [32][SCW] per-sample scalars
  static constexpr int oRED = oSC + 4 * 32 * SCW;
  static constexpr int TOTAL = oRED + 16;
  static_assert(TOTAL <= 40960, "LDS budget (160 KB) exceeded");
};

template <int IN, int OUT, int KIND, int ACT, bool TIMING = false>
__global__ __launch_bounds__(256, 1) void k_train_mfma(TrainArgs a) {
  using Lt = MfLayout<IN, OUT>;
  constexpr int KS0 = Lt::KS0, IP = Lt::IP, JT = Lt::JT, XP = Lt::XP, NS = Lt::NS;
  constexpr int NACT = (OUT > 4 ? OUT : 4);
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, g = lane >> 4;
  float* part = sm + Lt::oPART + w * Lt::PART;
  float* xs = sm + Lt::oXS + w * 32 * XP;
  float* sc = sm + Lt::oSC + w * 32 * Lt::SCW;
  float* T1 = sm + Lt::oT1 + w * MF_HID * MF_TLD;
  float* T2 = sm + Lt::oT2 + w * MF_HID * MF_TLD;
  const int n_extra = (KIND == MFK_GAUSSIAN) ? OUT : 0;
  // optional phase timing (s_memtime, shader cycles): per-wave totals in a.dbg[w*16 + phase]
  unsigned long long tacc[12]; unsigned long long tlast = 0;
  if (TIMING) { for (int k = 0; k < 12; ++k) tacc[k] = 0; tlast = __builtin_amdgcn_s_memtime(); }
#define MF_T(ph) do { if (TIMING) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); tacc[ph] += tn - tlast; tlast = tn; } } while (0)

  // small-parameter index s -> LDS master slot / canonical flat index / partial slot
  auto s_master = [&](int s) -> int {
    if (s < Lt::sB1) { const int o = s & 63, i = s >> 6; return Lt::oW1R + o * IP + i; }
    if (s < Lt::sB2) return Lt::oB1 + (s - Lt::sB1);
    if (s < Lt::sW3) return Lt::oB2 + (s - Lt::sB2);
    if (s < Lt::sB3) { const int t = s - Lt::sW3; const int o = t % OUT, i = t / OUT; return Lt::oW3R + o * MF_HID + i; }
    if (s < Lt::sEX) return Lt::oB3 + (s - Lt::sB3);
    return Lt::oEX + (s - Lt::sEX);
  };
  auto s_canon = [&](int s) -> int {
    if (s < Lt::sB1) return Lt::cW1 + s;
    if (s < Lt::sB2) return Lt::cB1 + (s - Lt::sB1);
    if (s < Lt::sW3) return Lt::cB2 + (s - Lt::sB2);
    if (s < Lt::sB3) return Lt::cW3 + (s - Lt::sW3);
    if (s < Lt::sEX) return Lt::cB3 + (s - Lt::sB3);
    return Lt::cEX + (s - Lt::sEX);
  };
  auto s_part = [&](int s) -> int {
    if (s < Lt::sB1) { const int o = s & 63, i = s >> 6; return Lt::pW1 + i * MF_LD + o; }
    if (s < Lt::sB2) return Lt::pB1 + (s - Lt::sB1);
    if (s < Lt::sW3) return Lt::pB2 + (s - Lt::sB2);
    if (s < Lt::sB3) { const int t = s - Lt::sW3; const int o = t % OUT, i = t / OUT; return Lt::pW3 + o * MF_HID + i; }
    if (s < Lt::sEX) return Lt::pB3 + (s - Lt::sB3);
    return Lt::pEX + (s - Lt::sEX);
  };
  const int ns_valid = Lt::sEX + n_extra;
  constexpr int NSI = (NS + 255) / 256;
  int so_part[NSI], so_master[NSI]; bool so_ok[NSI], so_ex[NSI];
#pragma unroll
  for (int k = 0; k < NSI; ++k) { const int s = tid + 256 * k; so_ok[k] = s < ns_valid; so_ex[k] = s >= Lt::sEX;
    so_part[k] = so_ok[k] ? s_part(s) : 0; so_master[k] = so_ok[k] ? s_master(s) : 0; }

  // ---- load parameters and Adam state --------------------------------------------------------------------------
  for (int q = tid; q < MF_HID * MF_HID; q += 256) { const int o = q & 63, i = q >> 6; const float v = a.p[Lt::cW2 + q];
    if (Lt::HAS_W2R) sm[Lt::oW2R + o * MF_LD + i] = v;
    sm[Lt::oW2C + i * MF_LD + o] = v; }
  for (int q = tid; q < MF_HID * IP; q += 256) sm[Lt::oW1R + q] = 0.f;
  if (tid < 16) { sm[Lt::oB3 + tid] = 0.f; sm[Lt::oEX + tid] = 0.f; }
  __syncthreads();
  for (int s = tid; s < NS; s += 256) { const bool in = s < ns_valid; const int pc = s_canon(s);
    if (in) sm[s_master(s)] = a.p[pc];
    sm[Lt::oMS + s] = in ? a.m[pc] : 0.f; sm[Lt::oVS + s] = in ? a.v[pc] : 0.f; }
  for (int q = tid; q < 4 * 32 * XP; q += 256) sm[Lt::oXS + q] = 0.f;
  // W2 rows owned by this wave, D layout of tile (w, m): reg r <-> W2[o = 16w+4g+r][i = 16m+c]
  f32x4 tW2[4], mW2[4], vW2[4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int pc = Lt::cW2 + (16 * w + 4 * g + r) + MF_HID * (16 * m + c);
      tW2[m][r] = a.p[pc]; mW2[m][r] = a.m[pc]; vW2[m][r] = a.v[pc]; }
  double bp1 = a.bp[0], bp2 = a.bp[1];
  const float lo = 1.f - a.eps_clip, hi = 1.f + a.eps_clip;
  const bool a2c = a.loss == CRUX_LOSS_A2C;
  AdamK ak; ak.b1 = (float)a.b1; ak.b2 = (float)a.b2; ak.omb1 = (float)(1.0 - a.b1); ak.omb2 = (float)(1.0 - a.b2); ak.eps = (float)a.eps; ak.eta = (float)a.eta;

  int32_t* order_cur = a.order_a; int32_t* order_nxt = a.order_b;
  long long total_batches = 0; int epochs_run = 0, err = 0; bool stop = false;
  float inf_loss = 0.f, inf_gn = 0.f, inf_ent = 0.f, inf_kl = 0.f, inf_clip = 0.f, inf_adv = 0.f, inf_ret = 0.f;
  const int n_epochs = a.ids ? 1 : a.epochs;
  if (!a.ids && !a.ord_all) { for (int64_t j = tid; j < a.len; j += 256) order_cur[j] = (int32_t)j; }
  if (!a.ids && !a.ord_all && a.pre_epochs > 0) {
    __syncthreads();
    for (int pe = 0; pe < a.pre_epochs; ++pe) {
      if (a.pre_perms) { for (int64_t j = tid; j < a.len; j += 256) order_nxt[j] = order_cur[a.pre_perms[(int64_t)pe * a.len + j]]; }
      else { const crux_perm pp = crux_perm_make(a.pre_seed, a.pre_counter + (uint64_t)pe, 0, (uint32_t)a.len);
        for (int64_t j = tid; j < a.len; j += 256) order_nxt[j] = order_cur[crux_perm_at(&pp, (uint32_t)j)]; }
      __syncthreads();
      int32_t* t = order_cur; order_cur = order_nxt; order_nxt = t;
    }
  }
  __syncthreads();
  const int64_t total_rows = a.ids ? a.n_ids : a.len;

  // ---- minibatch prefetch (HBM/L2 -> registers) and staging (registers -> this wave's LDS tiles) ----------------
  constexpr int NXL = (32 * IN + 63) / 64;
  float px[NXL]; float p_lp = 0.f, p_adv = 0.f, p_ret = 0.f; float p_act[NACT]; int p_valid = 0; uint8_t p_abyte[OUT];
#pragma unroll
  for (int k = 0; k < OUT; ++k) p_abyte[k] = 0;
#pragma unroll
  for (int k = 0; k < NACT; ++k) p_act[k] = 0.f;
  int n_row = 0, n_valid = 0;   // row index / validity of this lane's sample in the NEXT-to-be-fetched minibatch
  auto fetch_index = [&](const int32_t* ord, int64_t st, int nb) {   // issued one step before fetch_data uses it
    const int sidx = 32 * w + (lane & 31);
    n_valid = sidx < nb ? 1 : 0;
    n_row = n_valid ? (a.ids ? CRUX_GLOBAL_PTR(int32_t, a.ids)[st + sidx] : CRUX_GLOBAL_PTR(int32_t, ord)[st + sidx]) : 0;
  };
  auto fetch_data = [&]() {
    const int rowlo = n_row; p_valid = n_valid; const int64_t row = rowlo;
#pragma unroll
    for (int e = 0; e < NXL; ++e) {
      const int el = lane + 64 * e; const int s = el / IN, f = el - s * IN;
      const int rs = __shfl(rowlo, s & 31, 64); const int vs = __shfl(p_valid, s & 31, 64);
      px[e] = (el < 32 * IN && vs) ? CRUX_GLOBAL_PTR(float, a.S)[(int64_t)rs * IN + f] : 0.f;
    }
    p_lp = 0.f; p_adv = 0.f; p_ret = 0.f;
#pragma unroll
    for (int k = 0; k < NACT; ++k) p_act[k] = 0.f;
    if (lane < 32 && p_valid) {
      if (KIND != MFK_VALUE) { p_lp = CRUX_GLOBAL_PTR(float, a.LP)[row]; p_adv = CRUX_GLOBAL_PTR(float, a.ADV)[row]; }
      p_ret = a.RET ? CRUX_GLOBAL_PTR(float, a.RET)[row] : 0.f;
      if (KIND == MFK_CATEGORICAL) { const auto* av = CRUX_GLOBAL_PTR(uint8_t, a.A) + row * OUT;   // raw one-hot bytes; decoded in stage() so that
#pragma unroll                                                                              // nothing here waits on a load
        for (int k = 0; k < OUT; ++k) p_abyte[k] = av[k]; }
      if (KIND == MFK_GAUSSIAN) { const auto* av = CRUX_GLOBAL_PTR(float, a.A) + row * OUT;
#pragma unroll
        for (int k = 0; k < OUT; ++k) p_act[k] = av[k]; }
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int e = 0; e < NXL; ++e) { const int el = lane + 64 * e; const int s = el / IN, f = el - s * IN; if (el < 32 * IN) xs[s * XP + f] = px[e]; }
    if (KIND == MFK_CATEGORICAL) { int ai = 0;
#pragma unroll
      for (int k = 0; k < OUT; ++k) ai = p_abyte[k] ? k : ai;
      p_act[0] = (float)ai; }
    if (lane < 32) { float* q = sc + lane * Lt::SCW; q[0] = (float)p_valid; q[1] = p_lp; q[2] = p_adv; q[3] = p_ret;
#pragma unroll
      for (int k = 0; k < NACT; ++k) q[4 + k] = p_act[k]; }
    wave_sync();
  };

  for (int ep = 0; ep < n_epochs && !stop && !err; ++ep) {
    if (!a.ids && a.ord_all) order_cur = const_cast<int32_t*>(a.ord_all) + (size_t)ep * (size_t)a.len;   // shuffle orders composed ahead of time by k_compose_order
    else if (!a.ids) {   // shuffle!(D) as an index composition (experience_buffer.jl:118-124)
      if (a.perms) { for (int64_t j = tid; j < a.len; j += 256) order_nxt[j] = order_cur[a.perms[(int64_t)ep * a.len + j]]; }
      else { const crux_perm pp = crux_perm_make(a.shuffle_seed, a.shuffle_counter + (uint64_t)ep, 0, (uint32_t)a.len);
        for (int64_t j = tid; j < a.len; j += 256) order_nxt[j] = order_cur[crux_perm_at(&pp, (uint32_t)j)]; }
      __syncthreads();
      int32_t* t = order_cur; order_cur = order_nxt; order_nxt = t;
    }
    { const int nb0 = (int)(total_rows < a.bs ? total_rows : a.bs); fetch_index(order_cur, 0, nb0); fetch_data();
      const int64_t st1 = a.bs; const int nb1 = st1 < total_rows ? (int)((total_rows - st1) < a.bs ? (total_rows - st1) : a.bs) : 0; fetch_index(order_cur, st1 < total_rows ? st1 : 0, nb1); }
    for (int64_t st = 0; st < total_rows; st += a.bs) {
      const int nb = (int)((total_rows - st) < a.bs ? (total_rows - st) : a.bs);
      const float invB = 1.0f / (float)nb;
      // bias corrections of THIS step: 1 - beta^t in Float64 (one multiply per step), reciprocal on the f32 unit (1 ulp)
      ak.c1 = __builtin_amdgcn_rcpf((float)(1.0 - bp1)); ak.c2 = __builtin_amdgcn_rcpf((float)(1.0 - bp2));
      MF_T(0);
      stage();
      if (st + a.bs < total_rows) fetch_data();    // rows of minibatch t+1 (their indices were loaded during step t-1)
      { const int64_t st2 = st + 2 * (int64_t)a.bs; const int nb2 = st2 < total_rows ? (int)((total_rows - st2) < a.bs ? (total_rows - st2) : a.bs) : 0;
        fetch_index(order_cur, st2 < total_rows ? st2 : 0, nb2); }

      MF_T(1);
      // ======================= forward, C orientation =======================
      float xB[2][KS0], a1[4][KS0];
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int ks = 0; ks < KS0; ++ks) xB[n][ks] = xs[(16 * n + c) * XP + 4 * ks + g];
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int ks = 0; ks < KS0; ++ks) a1[m][ks] = sm[Lt::oW1R + (16 * m + c) * IP + 4 * ks + g];
      f32x4 h1[4][2];
#pragma unroll
      for (int m = 0; m < 4; ++m) { const f32x4 b = *(const f32x4*)&sm[Lt::oB1 + 16 * m + 4 * g];
#pragma unroll
        for (int n = 0; n < 2; ++n) { f32x4 acc = b;
#pragma unroll
          for (int ks = 0; ks < KS0; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[m][ks], xB[n][ks], acc, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[r] = actf<ACT>(acc[r]);
          h1[m][n] = acc; } }
      // publish H1 as a [feature][sample] tile (read by every wave for dW2, and by this wave as H1 in R orientation)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) T1[(16 * m + 4 * g + r) * MF_TLD + 16 * n + c] = h1[m][n][r];
      MF_T(2);
      f32x4 h2[4][2];
#pragma unroll
      for (int mp = 0; mp < 4; ++mp) { const f32x4 b = *(const f32x4*)&sm[Lt::oB2 + 16 * mp + 4 * g];
        f32x4 acc0 = b, acc1 = b;
#pragma unroll
        for (int m = 0; m < 4; ++m) { f32x4 wv;
          if (Lt::HAS_W2R) wv = *(const f32x4*)&sm[Lt::oW2R + (16 * mp + c) * MF_LD + 16 * m + 4 * g];
          else {
#pragma unroll
            for (int r = 0; r < 4; ++r) wv[r] = sm[Lt::oW2C + (16 * m + 4 * g + r) * MF_LD + 16 * mp + c]; }
#pragma unroll
          for (int r = 0; r < 4; ++r) { acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[r], h1[m][0][r], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[r], h1[m][1][r], acc1, 0, 0, 0); } }
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc0[r] = actf<ACT>(acc0[r]); acc1[r] = actf<ACT>(acc1[r]); }
        h2[mp][0] = acc0; h2[mp][1] = acc1; }

      MF_T(3);
      // ======================= layer 3 (VALU) + loss head =======================
      f32x4 w3[OUT][4];
#pragma unroll
      for (int o = 0; o < OUT; ++o)
#pragma unroll
        for (int m = 0; m < 4; ++m) w3[o][m] = *(const f32x4*)&sm[Lt::oW3R + o * MF_HID + 16 * m + 4 * g];
      float z[OUT][2];
#pragma unroll
      for (int o = 0; o < OUT; ++o)
#pragma unroll
        for (int n = 0; n < 2; ++n) { float acc = 0.f;
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = fmaf(w3[o][m][r], h2[m][n][r], acc);
          z[o][n] = g4_sum(acc) + sm[Lt::oB3 + o]; }
      float dz[OUT][2], dex[OUT][2];
      float s_lossp = 0.f, s_H = 0.f, s_kl = 0.f, s_adv = 0.f, s_ret = 0.f, s_clip = 0.f, s_sq = 0.f;
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const float* q = sc + (16 * n + c) * Lt::SCW;
        const bool valid = q[0] != 0.f; const float oldlp = q[1], A = q[2], R = q[3];
        const float cnt = (valid && g == 0) ? 1.f : 0.f;    // every sample is replicated in the 4 g-groups: count it once
#pragma unroll
        for (int k = 0; k < OUT; ++k) dex[k][n] = 0.f;
        if (KIND == MFK_VALUE) {
          const float d = z[0][n] - R; dz[0][n] = valid ? 2.f * d * invB : 0.f; s_sq += cnt * d * d; s_ret += cnt * R;
        } else if (KIND == MFK_CATEGORICAL) {
          // softmax / log / exp through the hardware transcendental units (v_exp_f32, v_log_f32, v_rcp_f32: ~1 ulp); arguments are
          // O(1) so the absolute error stays < 1e-6, inside the fp32 loss tolerance of the parity tests (1e-4 rel)
          const int ai = (int)q[4];
          float mx = z[0][n];
#pragma unroll
          for (int k = 1; k < OUT; ++k) mx = fmaxf(mx, z[k][n]);
          float pk[OUT], hk[OUT]; float sum = 0.f;
#pragma unroll
          for (int k = 0; k < OUT; ++k) { pk[k] = __expf(z[k][n] - mx); sum += pk[k]; }
          const float inv = __builtin_amdgcn_rcpf(sum); float pa = 0.f, H = 0.f, hp = 0.f;
#pragma unroll
          for (int k = 0; k < OUT; ++k) { pk[k] *= inv; pa = (k == ai) ? pk[k] : pa; const float pe = pk[k] + EPS32F; const float lg = __logf(pe); H -= pk[k] * lg;
            hk[k] = -lg - pk[k] * __builtin_amdgcn_rcpf(pe); hp += hk[k] * pk[k]; }
          const float newlp = __logf(pa); const float r = __expf(newlp - oldlp);
          const float u = r * A, rc = fminf(fmaxf(r, lo), hi), cl = rc * A; const float gsel = (u <= cl) ? A : 0.f;
          const float coef = a2c ? A : gsel * r, lterm = a2c ? newlp * A : fminf(u, cl), clipv = (!a2c && (r > hi || r < lo)) ? 1.f : 0.f;   // a2c_loss (a2c.jl:4-15): -mean(logpdf .* A)
#pragma unroll
          for (int k = 0; k < OUT; ++k) { const float dlogpi = ((k == ai) ? 1.f : 0.f) - pk[k];
            dz[k][n] = valid ? invB * (-a.lambda_p * coef * dlogpi - a.lambda_e * (pk[k] * (hk[k] - hp))) : 0.f; }
          s_lossp += cnt * lterm; s_H += cnt * H; s_kl += cnt * (oldlp - newlp); s_adv += cnt * A; s_ret += cnt * R;
          s_clip += cnt * clipv;
        } else {   // gaussian with constant log-std (policies.jl:333-348)
          float newlp = 0.f; float dd[OUT], s2[OUT];
#pragma unroll
          for (int k = 0; k < OUT; ++k) { const float ls = sm[Lt::oEX + k]; s2[k] = __expf(-2.f * ls); dd[k] = q[4 + k] - z[k][n];   // s2 = 1/sigma^2 through v_exp_f32 (1 ulp)
            newlp += (-(dd[k] * dd[k]) * (0.5f * s2[k]) - 0.9189385332046727f - ls); }
          const float r = __expf(newlp - oldlp); const float u = r * A, rc = fminf(fmaxf(r, lo), hi), cl = rc * A; const float gsel = (u <= cl) ? A : 0.f;
          const float coef = a2c ? A : gsel * r, lterm = a2c ? newlp * A : fminf(u, cl), clipv = (!a2c && (r > hi || r < lo)) ? 1.f : 0.f;   // a2c_loss (a2c.jl:4-15): -mean(logpdf .* A)
#pragma unroll
          for (int k = 0; k < OUT; ++k) { dz[k][n] = valid ? invB * (-a.lambda_p * coef * (dd[k] * s2[k])) : 0.f;
            dex[k][n] = valid ? invB * (-a.lambda_p * coef * ((dd[k] * dd[k]) * s2[k] - 1.f)) : 0.f; }
          s_lossp += cnt * lterm; s_kl += cnt * (oldlp - newlp); s_adv += cnt * A; s_ret += cnt * R; s_clip += cnt * clipv;
        }
      }

      MF_T(4);
      // ======================= backward, own samples =======================
      // dW3 partial over this wave's 32 samples and dZ2 (C) = act'(H2) .* (W3^T dz), overwriting h2
#pragma unroll
      for (int o = 0; o < OUT; ++o) { float pv[16];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) pv[4 * m + r] = fmaf(dz[o][0], h2[m][0][r], dz[o][1] * h2[m][1][r]);
        // lane c ends up with the row total of index c = (m = c>>2, r = c&3), i.e. feature 16(c>>2) + 4g + (c&3)
        part[Lt::pW3 + o * MF_HID + 16 * (c >> 2) + 4 * g + (c & 3)] = row16_reduce_scatter(pv, c); }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) { float d0 = 0.f, d1 = 0.f;
#pragma unroll
          for (int o = 0; o < OUT; ++o) { d0 = fmaf(w3[o][m][r], dz[o][0], d0); d1 = fmaf(w3[o][m][r], dz[o][1], d1); }
          h2[m][0][r] = actg<ACT>(h2[m][0][r], d0); h2[m][1][r] = actg<ACT>(h2[m][1][r], d1); }
      // stats, db3 and dlogSigma: one reduce-scatter per 16 values; only row g == 0 carries the stats (cnt), dz is the same in every row
      { constexpr int NV = 7 + OUT + (KIND == MFK_GAUSSIAN ? OUT : 0);
        float mv[((NV + 15) / 16) * 16];
#pragma unroll
        for (int k = 0; k < ((NV + 15) / 16) * 16; ++k) mv[k] = 0.f;
        mv[0] = s_lossp; mv[1] = s_H; mv[2] = s_kl; mv[3] = s_adv; mv[4] = s_ret; mv[5] = s_clip; mv[6] = s_sq;
#pragma unroll
        for (int o = 0; o < OUT; ++o) { mv[7 + o] = dz[o][0] + dz[o][1]; if (KIND == MFK_GAUSSIAN) mv[7 + OUT + o] = dex[o][0] + dex[o][1]; }
#pragma unroll
        for (int ch = 0; ch < (NV + 15) / 16; ++ch) { float cv[16];
#pragma unroll
          for (int k = 0; k < 16; ++k) cv[k] = mv[16 * ch + k];
          const float t = row16_reduce_scatter(cv, c);
          if (g == 0) part[Lt::pMISC + 16 * ch + c] = t; } }
      MF_T(5);
      // publish dZ2 tile
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) T2[(16 * m + 4 * g + r) * MF_TLD + 16 * n + c] = h2[m][n][r];
      // dH1 (R) = dZ2 (C regs as A: [i=c -> sample][k -> f' = 16m'+4g+r]) x W2 (B: W2[f'][f = 16m+c] = W2C[f][f'])
      f32x4 dz1r[2][4];
#pragma unroll
      for (int m = 0; m < 4; ++m) { f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mp = 0; mp < 4; ++mp) { const f32x4 wv = *(const f32x4*)&sm[Lt::oW2C + (16 * m + c) * MF_LD + 16 * mp + 4 * g];
#pragma unroll
          for (int r = 0; r < 4; ++r) { acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(h2[mp][0][r], wv[r], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(h2[mp][1][r], wv[r], acc1, 0, 0, 0); } }
        dz1r[0][m] = acc0; dz1r[1][m] = acc1; }
      MF_T(6);
      wave_sync();   // own T1/T2 tiles are complete for this wave's reads
      // dZ1 (R) = act'(H1 R) .* dH1 (R);  H1 (R)[sample 16n+4g+r'][f = 16m+c] comes from this wave's T1 tile
      float gb1[4], gb2[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) { float sb1 = 0.f, sb2 = 0.f;
#pragma unroll
        for (int n = 0; n < 2; ++n) { const f32x4 h1r = *(const f32x4*)&T1[(16 * m + c) * MF_TLD + 16 * n + 4 * g];
          const f32x4 d2 = *(const f32x4*)&T2[(16 * m + c) * MF_TLD + 16 * n + 4 * g];
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float d = actg<ACT>(h1r[r], dz1r[n][m][r]); dz1r[n][m][r] = d; sb1 += d; sb2 += d2[r]; } }
        gb1[m] = g4_sum(sb1); gb2[m] = g4_sum(sb2); }
      if (g == 0) {
#pragma unroll
        for (int m = 0; m < 4; ++m) { part[Lt::pB1 + 16 * m + c] = gb1[m]; part[Lt::pB2 + 16 * m + c] = gb2[m]; } }
      // dW1 partial over this wave's samples: A = dZ1 (R) [i=c -> o=16m+c][k -> sample], B = X (R) [k -> sample][j=c -> input 16jt+c]
#pragma unroll
      for (int jt = 0; jt < JT; ++jt) {
        float xR[2][4];
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) xR[n][r] = (16 * jt + c < IP) ? xs[(16 * n + 4 * g + r) * XP + 16 * jt + c] : 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) { f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dz1r[n][m][r], xR[n][r], acc, 0, 0, 0);
          if (16 * jt + c < Lt::W1ROWS) *(f32x4*)&part[Lt::pW1 + (16 * jt + c) * MF_LD + 16 * m + 4 * g] = acc; }   // D reg r <-> [o=16m+4g+r][i=16jt+c]
      }
      MF_T(7);
      __syncthreads();   // ---- B_a: all tiles and small partials are visible
      MF_T(8);

      // ======================= dW2 rows [16w,16w+16) over all samples =======================
      f32x4 gW2[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) gW2[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ws = 0; ws < 4; ++ws)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const float* t2 = sm + Lt::oT2 + ws * MF_HID * MF_TLD; const float* t1 = sm + Lt::oT1 + ws * MF_HID * MF_TLD;
          const f32x4 av = *(const f32x4*)&t2[(16 * w + c) * MF_TLD + 16 * n + 4 * g];       // A[i=c -> o=16w+c][k -> sample 16n+4g+r']
#pragma unroll
          for (int m = 0; m < 4; ++m) { const f32x4 bv = *(const f32x4*)&t1[(16 * m + c) * MF_TLD + 16 * n + 4 * g];   // B[k -> sample][j=c -> i=16m+c]
#pragma unroll
            for (int r = 0; r < 4; ++r) gW2[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], bv[r], gW2[m], 0, 0, 0); }
        }
      MF_T(9);
      // small parameters: reduce the 4 per-wave partials; collect sum of squares and the NaN flag
      float gs[NSI]; float ssq = 0.f; int bad = 0;
#pragma unroll
      for (int k = 0; k < NSI; ++k) { float gsum = 0.f;
        if (so_ok[k]) { const int po = so_part[k];
          gsum = ((sm[Lt::oPART + po] + sm[Lt::oPART + Lt::PART + po]) + sm[Lt::oPART + 2 * Lt::PART + po]) + sm[Lt::oPART + 3 * Lt::PART + po];
          if (KIND == MFK_GAUSSIAN && so_ex[k]) gsum += -a.lambda_e;         // d(-lambda_e * H)/dlogSigma, H = 1.4189 + sum(logSigma)
          ssq += gsum * gsum; bad |= isnan(gsum) ? 1 : 0; }
        gs[k] = gsum; }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) { ssq += gW2[m][r] * gW2[m][r]; bad |= isnan(gW2[m][r]) ? 1 : 0; }
      ssq = wave_sum(ssq);
      if (lane == 0) sm[Lt::oRED + w] = ssq;
      MF_T(10);
      const int any_bad = __syncthreads_or(bad);   // ---- B_or (also publishes RED)
      MF_T(8);
      // minibatch info (training.jl:22-23, ppo.jl:13-19); identical in every thread
      { float t[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) t[k] = ((sm[Lt::oPART + Lt::pST + k] + sm[Lt::oPART + Lt::PART + Lt::pST + k]) + sm[Lt::oPART + 2 * Lt::PART + Lt::pST + k]) + sm[Lt::oPART + 3 * Lt::PART + Lt::pST + k];
        const float fn = (float)nb;
        inf_gn = sqrtf(((sm[Lt::oRED] + sm[Lt::oRED + 1]) + sm[Lt::oRED + 2]) + sm[Lt::oRED + 3]);
        if (KIND == MFK_VALUE) { inf_loss = t[6] / fn; inf_ret = t[4] / fn; }
        else { const float p_loss = -(t[0] / fn); float entropy;
          if (KIND == MFK_CATEGORICAL) entropy = t[1] / fn;
          else { entropy = 1.4189385332046727f;
#pragma unroll
            for (int k = 0; k < OUT; ++k) entropy += sm[Lt::oEX + k]; }
          inf_ent = entropy; inf_loss = a.lambda_p * p_loss + a.lambda_e * (-entropy); inf_kl = t[2] / fn; inf_adv = t[3] / fn; inf_ret = t[4] / fn; inf_clip = t[5] / fn; }
      }
      if (any_bad) { inf_gn = NAN; err = CRUX_ENAN; break; }                   // training.jl:20: no update
      // ======================= Adam (Flux.update!, training.jl:21) =======================
      if (a.apply) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { float mm = mW2[m][r], vv = vW2[m][r]; const float d = adam1(gW2[m][r], mm, vv, ak);
            mW2[m][r] = mm; vW2[m][r] = vv; tW2[m][r] -= d;
            if (Lt::HAS_W2R) sm[Lt::oW2R + (16 * w + 4 * g + r) * MF_LD + 16 * m + c] = tW2[m][r]; }
          *(f32x4*)&sm[Lt::oW2C + (16 * m + c) * MF_LD + 16 * w + 4 * g] = tW2[m]; }
#pragma unroll
        for (int k = 0; k < NSI; ++k) { const int s = tid + 256 * k;
          if (so_ok[k]) { float mm = sm[Lt::oMS + s], vv = sm[Lt::oVS + s]; const float d = adam1(gs[k], mm, vv, ak);
            sm[Lt::oMS + s] = mm; sm[Lt::oVS + s] = vv; const int mo = so_master[k]; sm[mo] = sm[mo] - d; } }
        bp1 *= a.b1; bp2 *= a.b2;
      } else {   // gradient-only mode (crux_loss_grad): export the flat gradient
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) a.g[Lt::cW2 + (16 * w + 4 * g + r) + MF_HID * (16 * m + c)] = gW2[m][r];
#pragma unroll
        for (int k = 0; k < NSI; ++k) { const int s = tid + 256 * k; if (s < ns_valid) a.g[s_canon(s)] = gs[k]; }
      }
      MF_T(11);
      __syncthreads();   // ---- B_b: masters updated; tiles and partials may be overwritten
      MF_T(8);
      total_batches += 1;
      if (a.max_batches > 0 && total_batches >= a.max_batches) break;          // training.jl:45
      if (a.target_kl >= 0.f && KIND != MFK_VALUE && inf_kl > a.target_kl) break;   // :46
    }
    if (err) break;
    if (tid == 0 && a.epoch_infos) { float* e = a.epoch_infos + (size_t)ep * CRUX_INFO_N;   // aggregate_info(minibatch_infos) == last minibatch (Q3)
      for (int k = 0; k < CRUX_INFO_N; ++k) e[k] = 0.f;
      e[CRUX_INFO_LOSS] = inf_loss; e[CRUX_INFO_GRAD_NORM] = inf_gn;
      if (KIND != MFK_VALUE) { e[CRUX_INFO_ENTROPY] = inf_ent; e[CRUX_INFO_KL] = inf_kl; e[CRUX_INFO_CLIP_FRACTION] = inf_clip; e[CRUX_INFO_AVG_ADVANTAGE] = inf_adv; e[CRUX_INFO_AVG_RETURN] = inf_ret; } }
    epochs_run += 1;
    if (a.target_kl >= 0.f && KIND != MFK_VALUE && inf_kl > a.target_kl) stop = true;   // :49
    if (a.max_batches > 0 && total_batches >= a.max_batches) stop = true;               // :50
  }
  // ---- write back parameters and Adam state --------------------------------------------------------------------
  __syncthreads();
  if (a.apply) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int pc = Lt::cW2 + (16 * w + 4 * g + r) + MF_HID * (16 * m + c);
        a.p[pc] = tW2[m][r]; a.m[pc] = mW2[m][r]; a.v[pc] = vW2[m][r]; }
    for (int s = tid; s < ns_valid; s += 256) { const int pc = s_canon(s); a.p[pc] = sm[s_master(s)]; a.m[pc] = sm[Lt::oMS + s]; a.v[pc] = sm[Lt::oVS + s]; }
  }
  if (TIMING && lane == 0 && a.dbg) { for (int k = 0; k < 12; ++k) a.dbg[w * 16 + k] = tacc[k]; }
  if (tid == 0) {
    a.status[0] = err; a.status[1] = (int32_t)total_batches; a.status[2] = epochs_run; a.status[3] = (order_cur == a.order_a) ? 0 : 1;
    a.bp[0] = bp1; a.bp[1] = bp2;
    if (err && a.epoch_infos && epochs_run == 0) { a.epoch_infos[CRUX_INFO_LOSS] = inf_loss; a.epoch_infos[CRUX_INFO_GRAD_NORM] = NAN; }
  }
}

// ---- dispatch ---------------------------------------------------------------------------------------------------
template <int IN, int OUT, int KIND, int ACT, bool TIMING = false>
static int32_t launch_one(crux_ctx* c, const TrainArgs& a, hipStream_t stream) {
  using Lt = MfLayout<IN, OUT>;
  constexpr size_t lds = sizeof(float) * (size_t)Lt::TOTAL;
  static bool attr = false;
  if (!attr) { HIPCHK(c, hipFuncSetAttribute((const void*)k_train_mfma<IN, OUT, KIND, ACT, TIMING>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; }
  hipLaunchKernelGGL((k_train_mfma<IN, OUT, KIND, ACT, TIMING>), dim3(1), dim3(256), lds, stream, a);
  return crux_launch_check(c, "k_train_mfma");
}

int32_t crux_train_mfma8_launch(crux_ctx* c, const TrainArgs& a, int kind, bool* handled, hipStream_t stream);   // train_mfma8.hip
int32_t crux_train_mfma_x2_launch(crux_ctx* c, const TrainArgs& a, int kind, bool* handled, hipStream_t stream);   // train_mfma_x2.hip

int32_t crux_train_mfma_launch(crux_ctx* c, const TrainArgs& a, bool* handled, hipStream_t stream) {
  if (a.lag) { *handled = false; return CRUX_OK; }     // lagrange_ppo_loss: the penalty controller lives in the generic learner body only
  *handled = false;
  const NetDesc& nd = a.nd;
  if (getenv("CRUX_FORCE_GENERIC")) return CRUX_OK;
  if (nd.L != 3 || nd.dims[1] != MF_HID || nd.dims[2] != MF_HID || nd.acts[2] != CRUX_ACT_IDENTITY || nd.acts[0] != nd.acts[1]) return CRUX_OK;
  if (a.bs > 128 || a.loss == CRUX_LOSS_TD_INTERNAL || a.loss == CRUX_LOSS_MSE_ACTION) return CRUX_OK;   // those two heads exist in the generic kernel only
  if (a.ids && a.n_ids > 128) return CRUX_OK;
  const int in = nd.dims[0], out = nd.dims[3], act = nd.acts[0];
  int kind;
  if (a.loss == CRUX_LOSS_VALUE_MSE) kind = MFK_VALUE;
  else if (a.head == CRUX_HEAD_CATEGORICAL) kind = MFK_CATEGORICAL;
  else if (a.head == CRUX_HEAD_GAUSSIAN) kind = MFK_GAUSSIAN;
  else return CRUX_OK;
  if (getenv("CRUX_MFMA_TIMING") && getenv("CRUX_MFMA_WAVES4") && in == 4 && out == 2 && kind == MFK_CATEGORICAL && act == CRUX_ACT_RELU) {
    static unsigned long long* dbg = nullptr;
    if (!dbg) { if (hipMalloc(&dbg, 64 * 8) != hipSuccess) return crux_fail(c, CRUX_ENOMEM, "timing buffer"); }
    TrainArgs b = a; b.dbg = dbg; *handled = true;
    int32_t rc = launch_one<4, 2, MFK_CATEGORICAL, CRUX_ACT_RELU, true>(c, b, stream); if (rc) return rc;
    unsigned long long h[64]; HIPCHK(c, hipMemcpyAsync(h, dbg, sizeof h, hipMemcpyDeviceToHost, stream)); HIPCHK(c, hipStreamSynchronize(stream));
    static const char* nm[12] = {"loop+prefetch", "stage", "fwdL1+T1", "fwdL2", "L3+head", "dW3+dZ2+stats", "T2+dH1", "dZ1+db+dW1", "barrier-wait", "dW2", "small-reduce", "info+adam"};
    for (int w = 0; w < 4; ++w) { fprintf(stderr, "[mfma-timing] wave %d:", w); unsigned long long tot = 0; for (int k = 0; k < 12; ++k) tot += h[w * 16 + k];
      for (int k = 0; k < 12; ++k) fprintf(stderr, " %s=%.1f%%", nm[k], 100.0 * (double)h[w * 16 + k] / (double)tot); fprintf(stderr, " total=%llu cyc\n", tot); }
    return CRUX_OK;
  }
  { static const bool four = getenv("CRUX_MFMA_WAVES4") != nullptr;   // A/B switch: force the 4-wave kernel
    if (!four && (c->learner_cus != 1 || a.need_px)) { const int32_t rcx = crux_train_mfma_x2_launch(c, a, kind, handled, stream); if (rcx || *handled) return rcx; }
    if (a.need_px) return CRUX_OK;      // not covered by the two-CU kernel: the caller refuses (no un-synchronised training)
    if (!four) { const int32_t rc = crux_train_mfma8_launch(c, a, kind, handled, stream); if (rc || *handled) return rc; } }
  if (a.squash > 0.f) return CRUX_OK;     // SquashedGaussianPolicy: implemented in the 8-wave / two-CU kernels and the generic one, not in this 4-wave fallback
#define MF_CASE(I, O, K, A_) if (in == I && out == O && kind == K && act == A_) { *handled = true; return launch_one<I, O, K, A_>(c, a, stream); }
  MF_CASE(4, 2, MFK_CATEGORICAL, CRUX_ACT_RELU)     // C2 actor  (PPO CartPole)
  MF_CASE(4, 1, MFK_VALUE, CRUX_ACT_RELU)           // C2 critic
  MF_CASE(17, 6, MFK_GAUSSIAN, CRUX_ACT_RELU)       // C5 actor  (PPO HalfCheetah-shaped, 17 obs / 6 act)
  MF_CASE(17, 6, MFK_GAUSSIAN, CRUX_ACT_TANH)
  MF_CASE(17, 1, MFK_VALUE, CRUX_ACT_RELU)          // C5 critic
  MF_CASE(17, 1, MFK_VALUE, CRUX_ACT_TANH)
  MF_CASE(3, 1, MFK_GAUSSIAN, CRUX_ACT_RELU)        // Pendulum actor
  MF_CASE(3, 1, MFK_VALUE, CRUX_ACT_RELU)           // Pendulum critic
#undef MF_CASE
  return CRUX_OK;
}

