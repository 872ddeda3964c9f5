// train_mfma8.hip -- 8-wave variant of the persistent batch_train! / train! kernel (IN->64->64->OUT MLP family, gfx950).
//
// Same mathematics, data flow and reference semantics as train_mfma.hip (src/training.jl:13-55, ppo.jl:4-21,59-60, Flux Adam);
// what changes is the decomposition over waves. The 4-wave kernel runs one wave per SIMD, and the PMC counters of round 1
// showed ~23 % of its cycles waiting on LDS/L2 latency with nothing else to issue (an f32 MFMA and the FP32 VALU share a
// pipe, so a wave cannot cover its own waits). Here the workgroup has 8 waves = 2 per SIMD, each owning ONE 16-sample tile
// of the minibatch, so the SIMD's scheduler always has a second instruction stream:
//   forward / backward chain : wave w owns samples [16w, 16w+16)                                   (data parallel)
//   dW2 (64x64)              : wave w owns the two 16x16 tiles (mp = w>>1, m in {2(w&1), 2(w&1)+1}) (model parallel) and
//                              keeps theta/m/v of those 512 elements in registers for the whole launch
//   small parameters         : 8 per-wave partial gradients in LDS, reduced and Adam-updated by 512 threads
// VGPR budget is 256 per wave (2 waves/SIMD): weight fragments are fetched from LDS just in time instead of being held.
// Used when its LDS layout fits in 160 KB (IN <= 16); wider inputs use the 4-wave kernel.
#include <vector>
#include "train_args.h"

#include "mfma_helpers.h"

// LDS layouts are chosen against the gfx950 banking rules (ds_read_b128: 64 banks, four non-contiguous 16-lane groups; b32 accesses:
// 32 banks, two 32-lane halves): master rows are 72 floats, the per-wave exchange tiles are unpadded [64 features][16 samples]
// with the 16-byte sample slot XOR-swizzled by the feature's 4-row group (t8_h), W1/x rows are IP+2, scalar rows are odd --
// every b128 fragment read and every hot b32 read is conflict-free (round-1 PMC: stride 68/20 cost 2x on all 82 b128 reads/step).
#define MF8_LD 72
#define MF8_NW 8
__device__ __forceinline__ int t8_h(int q) { return (4 - q) & 3; }   // {0,3,2,1}

template <int IN, int OUT>
struct Mf8Layout {
  static constexpr int KS0 = (IN + 3) / 4, IP = KS0 * 4, JT = (IN + 15) / 16, XP = IP + 2, W1LD = IP + 2;
  static constexpr int SCW = (4 + (OUT > 4 ? OUT : 4)) | 1;
  static constexpr int sW1 = 0, sB1 = MF_HID * IN, sB2 = sB1 + MF_HID, sW3 = sB2 + MF_HID, sB3 = sW3 + MF_HID * OUT, sEX = sB3 + OUT, NS = sEX + 16;
  static constexpr int cW1 = 0, cB1 = MF_HID * IN, cW2 = cB1 + MF_HID, cB2 = cW2 + MF_HID * MF_HID, cW3 = cB2 + MF_HID, cB3 = cW3 + MF_HID * OUT, cEX = cB3 + OUT;
  static constexpr int W1ROWS = IP < 16 * JT ? IP : 16 * JT;
  static constexpr int pW1 = 0, pB1 = pW1 + W1ROWS * MF8_LD, pB2 = pB1 + MF_HID, pW3 = pB2 + MF_HID, pMISC = pW3 + OUT * MF_HID;
  static constexpr int pST = pMISC, pB3 = pMISC + 7, pEX = pB3 + OUT;
  static constexpr int PART = ((pMISC + 48 + 3) / 4) * 4;
  static constexpr int NSP = ((NS + 3) / 4) * 4;
  static constexpr int TILE = MF_HID * 16;
  static constexpr int BASE = MF_HID * MF8_LD + MF_HID * W1LD + 2 * MF_HID + OUT * MF_HID + 32 + 4 + 2 * NSP + 2 * MF8_NW * TILE + MF8_NW * PART + MF8_NW * 16 * XP + MF8_NW * 16 * SCW + 32;
  static constexpr bool FITS = BASE <= 40960;
  static constexpr bool HAS_W2R = BASE + MF_HID * MF8_LD <= 40960;
  static constexpr int oW2R = 0;
  static constexpr int oW2C = oW2R + (HAS_W2R ? MF_HID * MF8_LD : 0);
  static constexpr int oW1R = oW2C + MF_HID * MF8_LD;
  static constexpr int oB1 = oW1R + MF_HID * W1LD, oB2 = oB1 + MF_HID, oW3R = oB2 + MF_HID, oB3 = oW3R + OUT * MF_HID, oEX = oB3 + 16;
  static constexpr int oMS = ((oEX + 16 + 3) / 4) * 4, oVS = oMS + NSP;
  static constexpr int oT1 = oVS + NSP, oT2 = oT1 + MF8_NW * TILE;
  static constexpr int oPART = oT2 + MF8_NW * TILE;
  static constexpr int oXS = oPART + MF8_NW * PART;
  static constexpr int oSC = oXS + MF8_NW * 16 * XP;
  static constexpr int oRED = oSC + MF8_NW * 16 * SCW;        // [0,8): per-wave sum of squares; [8,15): reduced stat sums
  static constexpr int TOTAL = oRED + 32;
};

template <int IN, int OUT, int KIND, int ACT>
__global__ __launch_bounds__(512) void k_train_mfma8(TrainArgs a_single, const TrainArgs* __restrict__ multi) {
  // multi != NULL: a batch of independent learners, one workgroup (= one CU) each: the throughput form for multi-seed / population training
  const TrainArgs a = multi ? multi[blockIdx.x] : a_single;
  using Lt = Mf8Layout<IN, OUT>;
  static_assert(Lt::TOTAL <= 40960, "LDS budget (160 KB) exceeded");
  constexpr int KS0 = Lt::KS0, IP = Lt::IP, JT = Lt::JT, XP = Lt::XP, NS = Lt::NS;
  constexpr int NACT = (OUT > 4 ? OUT : 4);
  constexpr int NT = 64 * MF8_NW;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, g = lane >> 4;
  float* part = sm + Lt::oPART + w * Lt::PART;
  float* xs = sm + Lt::oXS + w * 16 * XP;
  float* sc = sm + Lt::oSC + w * 16 * Lt::SCW;
  float* T1 = sm + Lt::oT1 + w * Lt::TILE;
  float* T2 = sm + Lt::oT2 + w * Lt::TILE;
  const int n_extra = (KIND == MFK_GAUSSIAN) ? OUT : 0;
  const int t_wr = (4 * g) * 16 + 4 * ((c >> 2) ^ t8_h(g)) + (c & 3);   // tile element (feature 4g [+16m+r], sample c)
  const int t_rd = c * 16 + 4 * (g ^ t8_h(c >> 2));                   // tile b128 (feature c [+16m], samples 4g..4g+3)
  const int mp0 = w >> 1, m0 = 2 * (w & 1);        // dW2 / W2 ownership: tiles (mp0, m0) and (mp0, m0+1)

  auto s_master = [&](int s) -> int {
    if (s < Lt::sB1) { const int o = s & 63, i = s >> 6; return Lt::oW1R + o * Lt::W1LD + i; }
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
    if (s < Lt::sB1) { const int o = s & 63, i = s >> 6; return Lt::pW1 + i * MF8_LD + o; }
    if (s < Lt::sB2) return Lt::pB1 + (s - Lt::sB1);
    if (s < Lt::sW3) return Lt::pB2 + (s - Lt::sB2);
    if (s < Lt::sB3) { const int t = s - Lt::sW3; const int o = t % OUT, i = t / OUT; return Lt::pW3 + o * MF_HID + i; }
    if (s < Lt::sEX) return Lt::pB3 + (s - Lt::sB3);
    return Lt::pEX + (s - Lt::sEX);
  };
  const int ns_valid = Lt::sEX + n_extra;
  constexpr int NSI = (NS + NT - 1) / NT;
  int so_part[NSI], so_master[NSI]; bool so_ok[NSI], so_ex[NSI];
#pragma unroll
  for (int k = 0; k < NSI; ++k) { const int s = tid + NT * k; so_ok[k] = s < ns_valid; so_ex[k] = s >= Lt::sEX;
    so_part[k] = so_ok[k] ? s_part(s) : 0; so_master[k] = so_ok[k] ? s_master(s) : 0; }

  // ---- load parameters and Adam state --------------------------------------------------------------------------
  for (int q = tid; q < MF_HID * MF_HID; q += NT) { const int o = q & 63, i = q >> 6; const float v = a.p[Lt::cW2 + q];
    if (Lt::HAS_W2R) sm[Lt::oW2R + o * MF8_LD + i] = v;
    sm[Lt::oW2C + i * MF8_LD + o] = v; }
  for (int q = tid; q < MF_HID * Lt::W1LD; q += NT) sm[Lt::oW1R + q] = 0.f;
  if (tid < 16) { sm[Lt::oB3 + tid] = 0.f; sm[Lt::oEX + tid] = 0.f; }
  __syncthreads();
  for (int s = tid; s < NS; s += NT) { const bool in = s < ns_valid; const int pc = s_canon(s);
    if (in) sm[s_master(s)] = a.p[pc];
    sm[Lt::oMS + s] = in ? a.m[pc] : 0.f; sm[Lt::oVS + s] = in ? a.v[pc] : 0.f; }
  for (int q = tid; q < MF8_NW * 16 * XP; q += NT) sm[Lt::oXS + q] = 0.f;
  // owned W2 tiles, D layout: reg r of tile mm <-> W2[o = 16 mp0 + 4g + r][i = 16 (m0+mm) + c]
  f32x4 tW2[2], mW2[2], vW2[2];
#pragma unroll
  for (int mm = 0; mm < 2; ++mm)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int pc = Lt::cW2 + (16 * mp0 + 4 * g + r) + MF_HID * (16 * (m0 + mm) + c);
      tW2[mm][r] = a.p[pc]; mW2[mm][r] = a.m[pc]; vW2[mm][r] = a.v[pc]; }
  double bp1 = a.bp[0], bp2 = a.bp[1];
  const float lo = 1.f - a.eps_clip, hi = 1.f + a.eps_clip;
  const bool a2c = a.loss == CRUX_LOSS_A2C;
  AdamK ak; ak.b1 = (float)a.b1; ak.b2 = (float)a.b2; ak.omb1 = (float)(1.0 - a.b1); ak.omb2 = (float)(1.0 - a.b2); ak.eps = (float)a.eps; ak.eta = (float)a.eta;

  int32_t* order_cur = a.order_a; int32_t* order_nxt = a.order_b;
  long long total_batches = 0; int epochs_run = 0, err = 0; bool stop = false;
  float inf_loss = 0.f, inf_gn = 0.f, inf_ent = 0.f, inf_kl = 0.f, inf_clip = 0.f, inf_adv = 0.f, inf_ret = 0.f;
  const int n_epochs = a.ids ? 1 : a.epochs;
  if (!a.ids && !a.ord_all) { for (int64_t j = tid; j < a.len; j += NT) order_cur[j] = (int32_t)j; }
  if (!a.ids && !a.ord_all && a.pre_epochs > 0) {
    __syncthreads();
    for (int pe = 0; pe < a.pre_epochs; ++pe) {
      if (a.pre_perms) { for (int64_t j = tid; j < a.len; j += NT) order_nxt[j] = order_cur[a.pre_perms[(int64_t)pe * a.len + j]]; }
      else { const crux_perm pp = crux_perm_make(a.pre_seed, a.pre_counter + (uint64_t)pe, 0, (uint32_t)a.len);
        for (int64_t j = tid; j < a.len; j += NT) order_nxt[j] = order_cur[crux_perm_at(&pp, (uint32_t)j)]; }
      __syncthreads();
      int32_t* t = order_cur; order_cur = order_nxt; order_nxt = t;
    }
  }
  __syncthreads();
  const int64_t total_rows = a.ids ? a.n_ids : a.len;

  // ---- minibatch prefetch (HBM/L2 -> registers) and staging (registers -> this wave's LDS tiles) ----------------
  constexpr int NXL = (16 * IN + 63) / 64;
  float px[NXL]; float p_lp = 0.f, p_adv = 0.f, p_ret = 0.f; float p_act[NACT]; int p_valid = 0; uint8_t p_abyte[OUT];
#pragma unroll
  for (int k = 0; k < OUT; ++k) p_abyte[k] = 0;
#pragma unroll
  for (int k = 0; k < NACT; ++k) p_act[k] = 0.f;
  int n_row = 0, n_valid = 0;
  auto fetch_index = [&](const int32_t* ord, int64_t st, int nb) {
    const int sidx = 16 * w + c;
    n_valid = sidx < nb ? 1 : 0;
    n_row = n_valid ? (a.ids ? CRUX_GLOBAL_PTR(int32_t, a.ids)[st + sidx] : CRUX_GLOBAL_PTR(int32_t, ord)[st + sidx]) : 0;
  };
  auto fetch_data = [&]() {
    const int rowlo = n_row; p_valid = n_valid; const int64_t row = rowlo;
#pragma unroll
    for (int e = 0; e < NXL; ++e) {
      const int el = lane + 64 * e; const int s = el / IN, f = el - s * IN;
      const int rs = __shfl(rowlo, s & 15, 64); const int vs = __shfl(p_valid, s & 15, 64);
      px[e] = (el < 16 * IN && vs) ? CRUX_GLOBAL_PTR(float, a.S)[(int64_t)rs * IN + f] : 0.f;
    }
    p_lp = 0.f; p_adv = 0.f; p_ret = 0.f;
#pragma unroll
    for (int k = 0; k < NACT; ++k) p_act[k] = 0.f;
    if (lane < 16 && p_valid) {
      if (KIND != MFK_VALUE) { p_lp = CRUX_GLOBAL_PTR(float, a.LP)[row]; p_adv = CRUX_GLOBAL_PTR(float, a.ADV)[row]; }
      p_ret = a.RET ? CRUX_GLOBAL_PTR(float, a.RET)[row] : 0.f;
      if (KIND == MFK_CATEGORICAL) { const auto* av = CRUX_GLOBAL_PTR(uint8_t, a.A) + row * OUT;
#pragma unroll
        for (int k = 0; k < OUT; ++k) p_abyte[k] = av[k]; }
      if (KIND == MFK_GAUSSIAN) { const auto* av = CRUX_GLOBAL_PTR(float, a.A) + row * OUT;
#pragma unroll
        for (int k = 0; k < OUT; ++k) p_act[k] = av[k]; }
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int e = 0; e < NXL; ++e) { const int el = lane + 64 * e; const int s = el / IN, f = el - s * IN; if (el < 16 * IN) xs[s * XP + f] = px[e]; }
    if (KIND == MFK_CATEGORICAL) { int ai = 0;
#pragma unroll
      for (int k = 0; k < OUT; ++k) ai = p_abyte[k] ? k : ai;
      p_act[0] = (float)ai; }
    if (lane < 16) { float* q = sc + lane * Lt::SCW; q[0] = (float)p_valid; q[1] = p_lp; q[2] = p_adv; q[3] = p_ret;
      if (KIND == MFK_GAUSSIAN) {       // SquashedGaussianPolicy: the stored action is un-tanh'd once here and the tanh correction of logpdf rides in the spare slot
        static_assert(KIND != MFK_GAUSSIAN || ((4 + NACT) % 2 == 0), "the staging row needs its spare slot");
        float corr = 0.f;
        if (a.squash > 0.f) {
#pragma unroll
          for (int k = 0; k < OUT; ++k) { const float u = p_valid ? sq_untanh(p_act[k], a.squash) : 0.f; corr += p_valid ? sq_corr(u) : 0.f; p_act[k] = u; } }
        q[4 + NACT] = corr; }
#pragma unroll
      for (int k = 0; k < NACT; ++k) q[4 + k] = p_act[k]; }
    wave_sync();
  };

  for (int ep = 0; ep < n_epochs && !stop && !err; ++ep) {
    if (!a.ids && a.ord_all) order_cur = const_cast<int32_t*>(a.ord_all) + (size_t)ep * (size_t)a.len;   // shuffle orders composed ahead of time by k_compose_order
    else if (!a.ids) {   // shuffle!(D) as an index composition (experience_buffer.jl:118-124)
      if (a.perms) { for (int64_t j = tid; j < a.len; j += NT) order_nxt[j] = order_cur[a.perms[(int64_t)ep * a.len + j]]; }
      else { const crux_perm pp = crux_perm_make(a.shuffle_seed, a.shuffle_counter + (uint64_t)ep, 0, (uint32_t)a.len);
        for (int64_t j = tid; j < a.len; j += NT) order_nxt[j] = order_cur[crux_perm_at(&pp, (uint32_t)j)]; }
      __syncthreads();
      int32_t* t = order_cur; order_cur = order_nxt; order_nxt = t;
    }
    { const int nb0 = (int)(total_rows < a.bs ? total_rows : a.bs); fetch_index(order_cur, 0, nb0); fetch_data();
      const int64_t st1 = a.bs; const int nb1 = st1 < total_rows ? (int)((total_rows - st1) < a.bs ? (total_rows - st1) : a.bs) : 0; fetch_index(order_cur, st1 < total_rows ? st1 : 0, nb1); }
    for (int64_t st = 0; st < total_rows; st += a.bs) {
      const int nb = (int)((total_rows - st) < a.bs ? (total_rows - st) : a.bs);
      const float invB = 1.0f / (float)nb;
      ak.c1 = __builtin_amdgcn_rcpf((float)(1.0 - bp1)); ak.c2 = __builtin_amdgcn_rcpf((float)(1.0 - bp2));
      stage();
      if (st + a.bs < total_rows) fetch_data();
      { const int64_t st2 = st + 2 * (int64_t)a.bs; const int nb2 = st2 < total_rows ? (int)((total_rows - st2) < a.bs ? (total_rows - st2) : a.bs) : 0;
        fetch_index(order_cur, st2 < total_rows ? st2 : 0, nb2); }

      // ======================= forward, C orientation: D[feature 16m+4g+r][sample c] =======================
      float xB[KS0];
#pragma unroll
      for (int ks = 0; ks < KS0; ++ks) xB[ks] = xs[c * XP + 4 * ks + g];
      f32x4 h1[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) { f32x4 acc = *(const f32x4*)&sm[Lt::oB1 + 16 * m + 4 * g];
#pragma unroll
        for (int ks = 0; ks < KS0; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(sm[Lt::oW1R + (16 * m + c) * Lt::W1LD + 4 * ks + g], xB[ks], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = actf<ACT>(acc[r]);
        h1[m] = acc; }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) T1[t_wr + (16 * m + r) * 16] = h1[m][r];
      f32x4 h2[4];
#pragma unroll
      for (int mq = 0; mq < 2; ++mq) {     // two independent accumulator chains per pass
        f32x4 acc0 = *(const f32x4*)&sm[Lt::oB2 + 16 * (2 * mq) + 4 * g], acc1 = *(const f32x4*)&sm[Lt::oB2 + 16 * (2 * mq + 1) + 4 * g];
#pragma unroll
        for (int m = 0; m < 4; ++m) { f32x4 wv0, wv1;
          if (Lt::HAS_W2R) { wv0 = *(const f32x4*)&sm[Lt::oW2R + (32 * mq + c) * MF8_LD + 16 * m + 4 * g];
            wv1 = *(const f32x4*)&sm[Lt::oW2R + (32 * mq + 16 + c) * MF8_LD + 16 * m + 4 * g]; }
          else {
#pragma unroll
            for (int r = 0; r < 4; ++r) { wv0[r] = sm[Lt::oW2C + (16 * m + 4 * g + r) * MF8_LD + 32 * mq + c]; wv1[r] = sm[Lt::oW2C + (16 * m + 4 * g + r) * MF8_LD + 32 * mq + 16 + c]; } }
#pragma unroll
          for (int r = 0; r < 4; ++r) { acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv0[r], h1[m][r], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv1[r], h1[m][r], acc1, 0, 0, 0); } }
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc0[r] = actf<ACT>(acc0[r]); acc1[r] = actf<ACT>(acc1[r]); }
        h2[2 * mq] = acc0; h2[2 * mq + 1] = acc1; }

      // ======================= layer 3 (VALU) + loss head =======================
      f32x4 w3[OUT][4];
#pragma unroll
      for (int o = 0; o < OUT; ++o)
#pragma unroll
        for (int m = 0; m < 4; ++m) w3[o][m] = *(const f32x4*)&sm[Lt::oW3R + o * MF_HID + 16 * m + 4 * g];
      float z[OUT];
#pragma unroll
      for (int o = 0; o < OUT; ++o) { float acc = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc = fmaf(w3[o][m][r], h2[m][r], acc);
        z[o] = g4_sum(acc) + sm[Lt::oB3 + o]; }
      float dz[OUT], dex[OUT];
      float s_lossp = 0.f, s_H = 0.f, s_kl = 0.f, s_adv = 0.f, s_ret = 0.f, s_clip = 0.f, s_sq = 0.f;
      {
        const float* q = sc + c * Lt::SCW;
        const bool valid = q[0] != 0.f; const float oldlp = q[1], A = q[2], R = q[3];
        const float cnt = (valid && g == 0) ? 1.f : 0.f;    // every sample is replicated in the 4 g-groups: count it once
#pragma unroll
        for (int k = 0; k < OUT; ++k) dex[k] = 0.f;
        if (KIND == MFK_VALUE) {
          const float d = z[0] - R; dz[0] = valid ? 2.f * d * invB : 0.f; s_sq = cnt * d * d; s_ret = cnt * R;
        } else if (KIND == MFK_CATEGORICAL) {
          const int ai = (int)q[4];
          float mx = z[0];
#pragma unroll
          for (int k = 1; k < OUT; ++k) mx = fmaxf(mx, z[k]);
          float pk[OUT], hk[OUT]; float sum = 0.f;
#pragma unroll
          for (int k = 0; k < OUT; ++k) { pk[k] = __expf(z[k] - mx); sum += pk[k]; }
          const float inv = __builtin_amdgcn_rcpf(sum); float pa = 0.f, H = 0.f, hp = 0.f;
#pragma unroll
          for (int k = 0; k < OUT; ++k) { pk[k] *= inv; pa = (k == ai) ? pk[k] : pa; const float pe = pk[k] + EPS32F; const float lg = __logf(pe); H -= pk[k] * lg;
            hk[k] = -lg - pk[k] * __builtin_amdgcn_rcpf(pe); hp += hk[k] * pk[k]; }
          const float newlp = __logf(pa); const float r = __expf(newlp - oldlp);
          const float u = r * A, rc = fminf(fmaxf(r, lo), hi), cl = rc * A; const float gsel = (u <= cl) ? A : 0.f;
          const float coef = a2c ? A : gsel * r, lterm = a2c ? newlp * A : fminf(u, cl), clipv = (!a2c && (r > hi || r < lo)) ? 1.f : 0.f;   // a2c_loss (a2c.jl:4-15): -mean(logpdf .* A)
#pragma unroll
          for (int k = 0; k < OUT; ++k) { const float dlogpi = ((k == ai) ? 1.f : 0.f) - pk[k];
            dz[k] = valid ? invB * (-a.lambda_p * coef * dlogpi - a.lambda_e * (pk[k] * (hk[k] - hp))) : 0.f; }
          s_lossp = cnt * lterm; s_H = cnt * H; s_kl = cnt * (oldlp - newlp); s_adv = cnt * A; s_ret = cnt * R;
          s_clip = cnt * clipv;
        } else {   // gaussian with constant log-std (policies.jl:333-348)
          float newlp = 0.f; float dd[OUT], s2[OUT];
          float inr[OUT];
#pragma unroll
          for (int k = 0; k < OUT; ++k) { const float ls = sm[Lt::oEX + k]; const bool sq = a.squash > 0.f;
            s2[k] = __expf(-2.f * (sq ? sq_clampls(ls) : ls)); dd[k] = q[4 + k] - z[k];   // s2 = 1/sigma^2 through v_exp_f32 (1 ulp); squashed: sigma = exp(clamp(logSigma, -5, 2))
            inr[k] = (sq && !(ls >= -5.f && ls <= 2.f)) ? 0.f : 1.f;
            newlp += (-(dd[k] * dd[k]) * (0.5f * s2[k]) - 0.9189385332046727f - ls); }
          if (a.squash > 0.f) newlp -= q[4 + NACT];
          const float r = __expf(newlp - oldlp); const float u = r * A, rc = fminf(fmaxf(r, lo), hi), cl = rc * A; const float gsel = (u <= cl) ? A : 0.f;
          const float coef = a2c ? A : gsel * r, lterm = a2c ? newlp * A : fminf(u, cl), clipv = (!a2c && (r > hi || r < lo)) ? 1.f : 0.f;   // a2c_loss (a2c.jl:4-15): -mean(logpdf .* A)
#pragma unroll
          for (int k = 0; k < OUT; ++k) { dz[k] = valid ? invB * (-a.lambda_p * coef * (dd[k] * s2[k])) : 0.f;
            dex[k] = valid ? invB * (-a.lambda_p * coef * (((dd[k] * dd[k]) * s2[k]) * inr[k] - 1.f)) : 0.f; }
          s_lossp = cnt * lterm; s_kl = cnt * (oldlp - newlp); s_adv = cnt * A; s_ret = cnt * R; s_clip = cnt * clipv;
        }
      }

      // ======================= backward, own samples =======================
#pragma unroll
      for (int o = 0; o < OUT; ++o) { float pv[16];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) pv[4 * m + r] = dz[o] * h2[m][r];
        part[Lt::pW3 + o * MF_HID + 16 * (c >> 2) + 4 * g + (c & 3)] = row16_reduce_scatter(pv, c); }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) { float d0 = 0.f;
#pragma unroll
          for (int o = 0; o < OUT; ++o) d0 = fmaf(w3[o][m][r], dz[o], d0);
          h2[m][r] = actg<ACT>(h2[m][r], d0); }
      { constexpr int NV = 7 + OUT + (KIND == MFK_GAUSSIAN ? OUT : 0);
        float mv[((NV + 15) / 16) * 16];
#pragma unroll
        for (int k = 0; k < ((NV + 15) / 16) * 16; ++k) mv[k] = 0.f;
        mv[0] = s_lossp; mv[1] = s_H; mv[2] = s_kl; mv[3] = s_adv; mv[4] = s_ret; mv[5] = s_clip; mv[6] = s_sq;
#pragma unroll
        for (int o = 0; o < OUT; ++o) { mv[7 + o] = dz[o]; if (KIND == MFK_GAUSSIAN) mv[7 + OUT + o] = dex[o]; }
#pragma unroll
        for (int ch = 0; ch < (NV + 15) / 16; ++ch) { float cv[16];
#pragma unroll
          for (int k = 0; k < 16; ++k) cv[k] = mv[16 * ch + k];
          const float t = row16_reduce_scatter(cv, c);
          if (g == 0) part[Lt::pMISC + 16 * ch + c] = t; } }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) T2[t_wr + (16 * m + r) * 16] = h2[m][r];
      // dH1 (R) = dZ2 (C regs as A: [i=c -> sample][k -> f' = 16mp+4g+r]) x W2 (B: W2[f'][f = 16m+c] = W2C[f][f'])
      f32x4 dz1r[4];
#pragma unroll
      for (int mq = 0; mq < 2; ++mq) { f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mp = 0; mp < 4; ++mp) { const f32x4 wv0 = *(const f32x4*)&sm[Lt::oW2C + (32 * mq + c) * MF8_LD + 16 * mp + 4 * g];
          const f32x4 wv1 = *(const f32x4*)&sm[Lt::oW2C + (32 * mq + 16 + c) * MF8_LD + 16 * mp + 4 * g];
#pragma unroll
          for (int r = 0; r < 4; ++r) { acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(h2[mp][r], wv0[r], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(h2[mp][r], wv1[r], acc1, 0, 0, 0); } }
        dz1r[2 * mq] = acc0; dz1r[2 * mq + 1] = acc1; }
      wave_sync();   // own T1/T2 tiles are complete for this wave's reads
      // dZ1 (R)[sample 4g+r][f = 16m+c] = act'(H1 R) .* dH1 (R)
      float gb1[4], gb2[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) { float sb1 = 0.f, sb2 = 0.f;
        const f32x4 h1r = *(const f32x4*)&T1[t_rd + 256 * m];
        const f32x4 d2 = *(const f32x4*)&T2[t_rd + 256 * m];
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float d = actg<ACT>(h1r[r], dz1r[m][r]); dz1r[m][r] = d; sb1 += d; sb2 += d2[r]; }
        gb1[m] = g4_sum(sb1); gb2[m] = g4_sum(sb2); }
      if (g == 0) {
#pragma unroll
        for (int m = 0; m < 4; ++m) { part[Lt::pB1 + 16 * m + c] = gb1[m]; part[Lt::pB2 + 16 * m + c] = gb2[m]; } }
      // dW1 partial: A = dZ1 (R) [i=c -> o=16m+c][k -> sample 4g+r], B = X (R) [k -> sample][j=c -> input 16jt+c]
#pragma unroll
      for (int jt = 0; jt < JT; ++jt) {
        float xR[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) xR[r] = (16 * jt + c < IP) ? xs[(4 * g + r) * XP + 16 * jt + c] : 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) { f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dz1r[m][r], xR[r], acc, 0, 0, 0);
          if (16 * jt + c < Lt::W1ROWS) *(f32x4*)&part[Lt::pW1 + (16 * jt + c) * MF8_LD + 16 * m + 4 * g] = acc; }
      }
      __syncthreads();   // ---- B_a: all tiles and small partials are visible

      // ======================= dW2 tiles (mp0, m0) and (mp0, m0+1) over all 128 samples =======================
      f32x4 gW2[2];
      gW2[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; gW2[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ws = 0; ws < MF8_NW; ++ws) {
        const float* t2 = sm + Lt::oT2 + ws * Lt::TILE; const float* t1 = sm + Lt::oT1 + ws * Lt::TILE;
        const f32x4 av = *(const f32x4*)&t2[t_rd + 256 * mp0];          // A[i=c -> o=16mp0+c][k -> sample 4g+r]
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) { const f32x4 bv = *(const f32x4*)&t1[t_rd + 256 * (m0 + mm)];   // B[k -> sample][j=c -> i]
#pragma unroll
          for (int r = 0; r < 4; ++r) gW2[mm] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], bv[r], gW2[mm], 0, 0, 0); }
      }
      // small parameters: reduce the 8 per-wave partials; collect sum of squares and the NaN flag
      float gs[NSI]; float ssq = 0.f; int bad = 0;
#pragma unroll
      for (int k = 0; k < NSI; ++k) { float gsum = 0.f;
        if (so_ok[k]) { const int po = Lt::oPART + so_part[k];
          gsum = sm[po];
#pragma unroll
          for (int q = 1; q < MF8_NW; ++q) gsum += sm[po + q * Lt::PART];
          if (KIND == MFK_GAUSSIAN && so_ex[k]) gsum += -a.lambda_e;
          ssq += gsum * gsum; bad |= isnan(gsum) ? 1 : 0; }
        gs[k] = gsum; }
#pragma unroll
      for (int mm = 0; mm < 2; ++mm)
#pragma unroll
        for (int r = 0; r < 4; ++r) { ssq += gW2[mm][r] * gW2[mm][r]; bad |= isnan(gW2[mm][r]) ? 1 : 0; }
      ssq = wave_sum(ssq);
      if (lane == 0) sm[Lt::oRED + w] = ssq;
      if (tid >= NT - 8 && tid < NT - 1) { const int k = tid - (NT - 8); float t = sm[Lt::oPART + Lt::pST + k];   // stat sums, by 7 lanes of the last wave
#pragma unroll
        for (int q = 1; q < MF8_NW; ++q) t += sm[Lt::oPART + q * Lt::PART + Lt::pST + k];
        sm[Lt::oRED + 8 + k] = t; }
      const int any_bad = __syncthreads_or(bad);   // ---- B_or (also publishes RED)
      // minibatch info (training.jl:22-23, ppo.jl:13-19); identical in every thread. Only the epoch's last minibatch (or the one that stops the
      // loop) is ever reported (aggregate_info over aliased dicts), so the full row is built lazily; the KL statistic that
      // drives early stopping is the one value needed every step.
      { const float* t = sm + Lt::oRED + 8;
        if (KIND != MFK_VALUE && a.target_kl >= 0.f) inf_kl = t[2] * invB;
        const bool report = any_bad || st + a.bs >= total_rows || (a.max_batches > 0 && total_batches + 1 >= a.max_batches) ||
                            (KIND != MFK_VALUE && a.target_kl >= 0.f && inf_kl > a.target_kl);
        if (report) {
          float ss = sm[Lt::oRED];
#pragma unroll
          for (int q = 1; q < MF8_NW; ++q) ss += sm[Lt::oRED + q];
          inf_gn = sqrtf(ss);
          if (KIND == MFK_VALUE) { inf_loss = t[6] * invB; inf_ret = t[4] * invB; }
          else { const float p_loss = -(t[0] * invB); float entropy;
            if (KIND == MFK_CATEGORICAL) entropy = t[1] * invB;
            else { entropy = 1.4189385332046727f;
#pragma unroll
              for (int k = 0; k < OUT; ++k) entropy += sm[Lt::oEX + k]; }
            inf_ent = entropy; inf_loss = a.lambda_p * p_loss + a.lambda_e * (-entropy); inf_kl = t[2] * invB; inf_adv = t[3] * invB; inf_ret = t[4] * invB; inf_clip = t[5] * invB; }
        }
      }
      if (any_bad) { inf_gn = NAN; err = CRUX_ENAN; break; }                   // training.jl:20: no update
      // ======================= Adam (Flux.update!, training.jl:21) =======================
      if (a.apply) {
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { float m_ = mW2[mm][r], v_ = vW2[mm][r]; const float d = adam1(gW2[mm][r], m_, v_, ak);
            mW2[mm][r] = m_; vW2[mm][r] = v_; tW2[mm][r] -= d;
            if (Lt::HAS_W2R) sm[Lt::oW2R + (16 * mp0 + 4 * g + r) * MF8_LD + 16 * (m0 + mm) + c] = tW2[mm][r]; }
          *(f32x4*)&sm[Lt::oW2C + (16 * (m0 + mm) + c) * MF8_LD + 16 * mp0 + 4 * g] = tW2[mm]; }
#pragma unroll
        for (int k = 0; k < NSI; ++k) { const int s = tid + NT * k;
          if (so_ok[k]) { float m_ = sm[Lt::oMS + s], v_ = sm[Lt::oVS + s]; const float d = adam1(gs[k], m_, v_, ak);
            sm[Lt::oMS + s] = m_; sm[Lt::oVS + s] = v_; const int mo = so_master[k]; sm[mo] = sm[mo] - d; } }
        bp1 *= a.b1; bp2 *= a.b2;
      } else {   // gradient-only mode (crux_loss_grad): export the flat gradient
#pragma unroll
        for (int mm = 0; mm < 2; ++mm)
#pragma unroll
          for (int r = 0; r < 4; ++r) a.g[Lt::cW2 + (16 * mp0 + 4 * g + r) + MF_HID * (16 * (m0 + mm) + c)] = gW2[mm][r];
#pragma unroll
        for (int k = 0; k < NSI; ++k) { const int s = tid + NT * k; if (s < ns_valid) a.g[s_canon(s)] = gs[k]; }
      }
      __syncthreads();   // ---- B_b: masters updated; tiles and partials may be overwritten
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
    for (int mm = 0; mm < 2; ++mm)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int pc = Lt::cW2 + (16 * mp0 + 4 * g + r) + MF_HID * (16 * (m0 + mm) + c);
        a.p[pc] = tW2[mm][r]; a.m[pc] = mW2[mm][r]; a.v[pc] = vW2[mm][r]; }
    for (int s = tid; s < ns_valid; s += NT) { const int pc = s_canon(s); a.p[pc] = sm[s_master(s)]; a.m[pc] = sm[Lt::oMS + s]; a.v[pc] = sm[Lt::oVS + s]; }
  }
  if (tid == 0) {
    a.status[0] = err; a.status[1] = (int32_t)total_batches; a.status[2] = epochs_run; a.status[3] = (order_cur == a.order_a) ? 0 : 1;
    a.bp[0] = bp1; a.bp[1] = bp2;
    if (err && a.epoch_infos && epochs_run == 0) { a.epoch_infos[CRUX_INFO_LOSS] = inf_loss; a.epoch_infos[CRUX_INFO_GRAD_NORM] = NAN; }
  }
}

// ---- dispatch ---------------------------------------------------------------------------------------------------
template <int IN, int OUT, int KIND, int ACT>
static int32_t launch_one8(crux_ctx* c, const TrainArgs& a, hipStream_t stream) {
  using Lt = Mf8Layout<IN, OUT>;
  constexpr size_t lds = sizeof(float) * (size_t)Lt::TOTAL;
  static bool attr = false;
  if (!attr) { HIPCHK(c, hipFuncSetAttribute((const void*)k_train_mfma8<IN, OUT, KIND, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; }
  hipLaunchKernelGGL((k_train_mfma8<IN, OUT, KIND, ACT>), dim3(1), dim3(512), lds, stream, a, (const TrainArgs*)nullptr);
  return crux_launch_check(c, "k_train_mfma8");
}

// n independent learners, one CU each (grid n): argument blocks uploaded to a per-stream device array
template <int IN, int OUT, int KIND, int ACT>
static int32_t launch_multi8(crux_ctx* c, std::vector<TrainArgs>& as, hipStream_t stream) {
  using Lt = Mf8Layout<IN, OUT>;
  constexpr size_t lds = sizeof(float) * (size_t)Lt::TOTAL;
  static bool attr = false;
  if (!attr) { HIPCHK(c, hipFuncSetAttribute((const void*)k_train_mfma8<IN, OUT, KIND, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; }
  const int which = stream == c->stream ? 0 : 1; const size_t n = as.size(), need = n * sizeof(TrainArgs) + 256;
  if (c->amulti_bytes[which] < need) {
    if (c->amulti[which]) { HIPCHK(c, hipDeviceSynchronize()); (void)hipFree(c->amulti[which]); c->amulti[which] = nullptr; c->amulti_bytes[which] = 0; }
    if (hipMalloc(&c->amulti[which], need) != hipSuccess) return crux_fail(c, CRUX_ENOMEM, "multi-learner argument blocks (%zu bytes)", need);
    c->amulti_bytes[which] = need;
  }
  TrainArgs* d_args = (TrainArgs*)c->amulti[which];
  HIPCHK(c, hipMemcpyAsync(d_args, as.data(), n * sizeof(TrainArgs), hipMemcpyHostToDevice, stream));
  HIPCHK(c, hipStreamSynchronize(stream));      // `as` is pageable host memory: the copy must have left it before the caller's vector can change
  hipLaunchKernelGGL((k_train_mfma8<IN, OUT, KIND, ACT>), dim3((unsigned)n), dim3(512), lds, stream, as[0], (const TrainArgs*)d_args);
  return crux_launch_check(c, "k_train_mfma8 (multi)");
}

int32_t crux_train_mfma8_launch_multi(crux_ctx* c, std::vector<TrainArgs>& as, bool* handled, hipStream_t stream) {
  *handled = false;
  if (as.empty()) return CRUX_OK;
  const TrainArgs& a = as[0]; const NetDesc& nd = a.nd;
  if (nd.L != 3 || nd.dims[1] != MF_HID || nd.dims[2] != MF_HID || nd.acts[2] != CRUX_ACT_IDENTITY || nd.acts[0] != nd.acts[1] || a.ids || !a.apply || a.bs > 128 || a.len < a.bs) return CRUX_OK;
  const int in = nd.dims[0], out = nd.dims[3], act = nd.acts[0];
  const int kind = a.loss == CRUX_LOSS_VALUE_MSE ? MFK_VALUE : (a.head == CRUX_HEAD_CATEGORICAL ? MFK_CATEGORICAL : (a.head == CRUX_HEAD_GAUSSIAN ? MFK_GAUSSIAN : -1));
  if (!(a.loss == CRUX_LOSS_VALUE_MSE || CRUX_IS_PG(a.loss)) || kind < 0) return CRUX_OK;
#define MF8M_CASE(I, O, K, A_) if (in == I && out == O && kind == K && act == A_) { *handled = true; return launch_multi8<I, O, K, A_>(c, as, stream); }
  MF8M_CASE(4, 2, MFK_CATEGORICAL, CRUX_ACT_RELU)
  MF8M_CASE(4, 1, MFK_VALUE, CRUX_ACT_RELU)
  MF8M_CASE(3, 1, MFK_GAUSSIAN, CRUX_ACT_RELU)
  MF8M_CASE(3, 1, MFK_VALUE, CRUX_ACT_RELU)
#undef MF8M_CASE
  return CRUX_OK;
}

// Called by crux_train_mfma_launch after its shape checks; handles the narrow-input members of the family.
int32_t crux_train_mfma8_launch(crux_ctx* c, const TrainArgs& a, int kind, bool* handled, hipStream_t stream) {
  *handled = false;
  const int in = a.nd.dims[0], out = a.nd.dims[3], act = a.nd.acts[0];
#define MF8_CASE(I, O, K, A_) if (in == I && out == O && kind == K && act == A_) { *handled = true; return launch_one8<I, O, K, A_>(c, a, stream); }
  MF8_CASE(4, 2, MFK_CATEGORICAL, CRUX_ACT_RELU)     // C2 actor  (PPO CartPole)
  MF8_CASE(4, 1, MFK_VALUE, CRUX_ACT_RELU)           // C2 critic
  MF8_CASE(3, 1, MFK_GAUSSIAN, CRUX_ACT_RELU)        // Pendulum actor
  MF8_CASE(3, 1, MFK_VALUE, CRUX_ACT_RELU)           // Pendulum critic
#undef MF8_CASE
  return CRUX_OK;
}
