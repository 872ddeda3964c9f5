// train_mfma_kernel.h -- THE persistent batch_train! / train! kernel of the register-resident IN->64->64->OUT family (src/training.jl:13-55,
// ppo.jl:4-21,59-60, Flux Adam), one source for its two forms:
//   <NW = 8, NWG = 1>  one workgroup of 8 waves on ONE compute unit, a 16-sample tile per wave (train_mfma8.hip: single steps, small minibatches,
//                      and the population form -- one CU per learner keeps the most learners resident);
//   <NW = 4, NWG = 2>  the step split over TWO compute units of one XCD (train_mfma_x2.hip), described below.
// Whole-file include of both translation units; everything that differs between the forms hangs on the two template parameters.
//
// The 81 920 Adam steps of a PPO iteration are serially dependent, so one learner cannot use more than the CUs one STEP can be spread over.
// In the two-CU form workgroup p in {0,1} (blockIdx 0 and 8: consecutive workgroups go round-robin over the 8 XCDs, so these two share an L2)
// owns samples [64p, 64p+64) of every minibatch as four 16-sample tiles, one per wave (1 wave per SIMD):
//   * forward / backward chain and the partial weight gradients over its own 64 samples -- half of the MFMA and VALU work;
//   * the two partial gradients (4.6 k floats) are exchanged through the shared L2 once per step: plain stores (the vector L1 is
//     write-through) + s_waitcnt + one agent-scope atomic counter as the barrier + `sc1` loads that miss the L1. No cache
//     write-back/invalidate is involved because both workgroups sit behind the same L2 (tools/xcu_barrier_bench.hip: 1.1 us per
//     exchange, against 1.8 us with release/acquire fences and wrong data across XCDs);
//   * both workgroups add the partials in the same order (a+b == b+a bitwise), so they compute bit-identical gradients, norms,
//     early-stopping decisions and Adam updates and their parameter copies never diverge; workgroup 0 writes the results back.
// Every thread exchanges exactly the gradient elements it owns (same thread <-> element map in both workgroups), so the exchange
// needs no index translation: 4 b128 stores + 4 b128 loads per lane for W2, NSI dwords for the small parameters.
#pragma once
#include "train_args.h"

#include "mfma_helpers.h"
#include "peer_wait.h"

// LDS layouts are chosen against the gfx950 banking rules (ds_read_b128: 64 banks, four non-contiguous 16-lane groups; b32 accesses:
// 32 banks, two 32-lane halves): master rows are 72 floats, the per-wave exchange tiles are unpadded [64 features][16 samples]
// with the 16-byte sample slot XOR-swizzled by the feature's 4-row group (tx_h), W1/x rows are IP+2, scalar rows are odd --
// every b128 fragment read and every hot b32 read is conflict-free (round-1 PMC: stride 68/20 cost 2x on all 82 b128 reads/step).
#define MF8_LD 72
__device__ __forceinline__ int tx_h(int q) { return (4 - q) & 3; }   // {0,3,2,1}

template <int IN, int OUT, int NW>
struct MfLayout {
  static constexpr int MF8_NW = NW;
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
  static constexpr int oRED = oSC + MF8_NW * 16 * SCW;        // [0,8): per-wave sum of squares; [8,15): reduced stat sums; [16]: abort flag
  static constexpr int TOTAL = oRED + 32;
};

// PX: the replica-group form (per-minibatch gradient all-reduce over the peer slots). A separate instantiation, so that the single-GPU kernel carries
// neither the branch nor the live registers of the exchange (with a run-time test the C2 actor step was 3.6 % slower: 8.75 vs 8.44 us).
// LAG: lagrange_ppo_loss (ppo.jl:70-131) -- the PID penalty controller advanced once per minibatch inside the kernel and the cost-advantage term of the loss.
// A separate instantiation (two-CU form only), so that the plain kernels carry none of it.
template <int IN, int OUT, int KIND, int ACT, int NW, int NWG, bool TIMING = false, bool PX = false, bool LAG = false>
__global__ __launch_bounds__(64 * NW) void k_train_mfma(TrainArgs a_single, const TrainArgs* __restrict__ multi) {
  static_assert((NW == 4 && NWG == 2) || (NW == 8 && NWG == 1), "forms: two workgroups of four waves, or one of eight");
  static_assert(NWG == 2 || !PX, "the replica-group exchange lives in the two-CU form");
  static_assert(!LAG || (NWG == 2 && !PX && KIND != MFK_VALUE), "lagrange_ppo_loss: two-CU form, policy heads");
  constexpr int MF8_NW = NW;
  constexpr int WT = 16 / NW;                        // 16x16 tiles of W2 (and of its gradient, Adam state) owned by a wave
  // multi != NULL: a batch of independent learners (multi-seed / population training) in one launch. Two-CU form: replica r = blockIdx / 16 uses the
  // workgroups 16r and 16r + 8 (one XCD) and its own argument block, exchange area and status row; one-CU form: replica r = blockIdx
  const TrainArgs a = multi ? multi[NWG == 2 ? blockIdx.x >> 4 : blockIdx.x] : a_single;
  using Lt = MfLayout<IN, OUT, NW>;
  static_assert(Lt::TOTAL <= 40960, "LDS budget (160 KB) exceeded");
  constexpr int KS0 = Lt::KS0, IP = Lt::IP, JT = Lt::JT, XP = Lt::XP, NS = Lt::NS;
  constexpr int NACT = (OUT > 4 ? OUT : 4);
  constexpr int NT = 64 * MF8_NW;
  // learner r works in blocks 16r + x and 16r + x + 8 with x = (r/2)%8: same XCD for the pair (workgroups go round-robin over the 8 XCDs).
  // An XCD therefore hosts learners of both parities, whose workgroups are the (2r)-th/(2r+1)-th of that XCD's stream: measured, with
  // x = r%8 all busy workgroups of an XCD fell on HALF of its CUs (positions = 0,1 mod 4 or 2,3 mod 4: the in-order dispatcher deals workgroups
  // round-robin to the XCD's 4 shader arrays) and a population of 40 (20 busy workgroups per XCD on 16 CUs) ran actor and critic back to back.
  if (NWG == 2 && (blockIdx.x & 7) != ((blockIdx.x >> 5) & 7)) return;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, g = lane >> 4;
  float* part = sm + Lt::oPART + w * Lt::PART;
  float* xs = sm + Lt::oXS + w * 16 * XP;
  float* sc = sm + Lt::oSC + w * 16 * Lt::SCW;
  float* T1 = sm + Lt::oT1 + w * Lt::TILE;
  float* T2 = sm + Lt::oT2 + w * Lt::TILE;
  const int n_extra = (KIND == MFK_GAUSSIAN) ? OUT : 0;
  // optional phase timing (s_memtime): per-wave totals in a.dbg[(4p + w)*16 + phase]  (CRUX_MFMA_TIMING=1)
  unsigned long long tacc[16]; unsigned long long tlast = 0;
  if (TIMING) { for (int k = 0; k < 16; ++k) tacc[k] = 0; tlast = __builtin_amdgcn_s_memtime(); }
#define MX_T(ph) do { if (TIMING) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); tacc[ph] += tn - tlast; tlast = tn; } } while (0)
  const int t_wr = (4 * g) * 16 + 4 * ((c >> 2) ^ tx_h(g)) + (c & 3);   // tile element (feature 4g [+16m+r], sample c)
  const int t_rd = c * 16 + 4 * (g ^ tx_h(c >> 2));                   // tile b128 (feature c [+16m], samples 4g..4g+3)
  // dW2 / W2 ownership: four waves own the four tiles (mp0 = w, m = 0..3) = rows [16w, 16w+16); eight waves the tiles (mp0, m0) and (mp0, m0 + 1)
  const int mp0 = NW == 4 ? w : (w >> 1), m0 = NW == 4 ? 0 : 2 * (w & 1);
  const int p = NWG == 2 ? (int)((blockIdx.x >> 3) & 1) : 0;   // workgroup 0 or 1 of this learner (blockIdx 16r / 16r + 8)

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
  uint32_t my_xcc = 0;
  if (NWG == 2 && tid == 0) { asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my_xcc)); my_xcc &= 0xf;
    __hip_atomic_store(a.xctr + 8 + p, my_xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // checked against the peer's at the first exchange
  __syncthreads();
  for (int s = tid; s < NS; s += NT) { const bool in = s < ns_valid; const int pc = s_canon(s);
    if (in) sm[s_master(s)] = a.p[pc];
    sm[Lt::oMS + s] = in ? a.m[pc] : 0.f; sm[Lt::oVS + s] = in ? a.v[pc] : 0.f; }
  for (int q = tid; q < MF8_NW * 16 * XP; q += NT) sm[Lt::oXS + q] = 0.f;
  // owned W2 tiles, D layout: reg r of tile mm <-> W2[o = 16 mp0 + 4g + r][i = 16 (m0+mm) + c]
  f32x4 tW2[WT], mW2[WT], vW2[WT];
#pragma unroll
  for (int mm = 0; mm < WT; ++mm)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int pc = Lt::cW2 + (16 * mp0 + 4 * g + r) + MF_HID * (16 * (m0 + mm) + c);
      tW2[mm][r] = a.p[pc]; mW2[mm][r] = a.m[pc]; vW2[mm][r] = a.v[pc]; }
  double bp1 = a.bp[0], bp2 = a.bp[1];
  const float lo = 1.f - a.eps_clip, hi = 1.f + a.eps_clip;
  const bool a2c = a.loss == CRUX_LOSS_A2C;
  AdamK ak; ak.b1 = (float)a.b1; ak.b2 = (float)a.b2; ak.omb1 = (float)(1.0 - a.b1); ak.omb2 = (float)(1.0 - a.b2); ak.eps = (float)a.eps; ak.eta = (float)a.eta;
  // lagrange_ppo_loss: every thread carries an identical copy of the PID state (ppo.jl:192-201); the minibatch's :cost / :episode_end / :cost_advantage are staged
  // in LDS next to the observation tiles (all 128 rows of the minibatch in BOTH workgroups: each advances the controller on its own, bit for bit alike)
  __shared__ float lag_cost[LAG ? 128 : 1], lag_ee[LAG ? 128 : 1], lag_cadv[LAG ? NW * 16 : 1];
  crux_lagrange lg{}; float pen = 0.f;
  if constexpr (LAG) lg = *a.lag;

  int32_t* order_cur = a.order_a; int32_t* order_nxt = a.order_b;
  long long total_batches = 0; int epochs_run = 0, err = 0, why_failed = 0; bool stop = false;
  bool staged = false;                              // the next minibatch is already in this wave's LDS staging tiles
  long long xstep = 0;                              // exchanges done so far (the counter target and the slot parity)
  // replica group (comm.hip "peer"): exchanges done on this learner stream before this launch -- slot parity and flag values continue across launches
  float* const px_mine = PX ? a.px_tab[a.px_rank] : nullptr;
  const unsigned long long px0 = PX ? *(const unsigned long long*)(px_mine + CRUX_PX_COUNT) : 0ull;
  if (PX && tid == 0) px_launch_begin(px_mine, p);      // the launch's wait budget starts from zero (peer_wait.h, bound 2)
  const float px_inv = PX ? 1.0f / (float)a.px_n : 1.0f;
  constexpr int XSLOT = 4096 + NSI * NT + 16;
  // the reported minibatch's info (training.jl:22-23) is kept by thread 0 -- its only reader (epoch_infos) -- in free words of the reduction area, not in registers of every
  // thread that would stay live across the whole launch (k_train_fs2 has the measurement: ~30 VGPRs); the KL stays a register: the loop exits read it
  constexpr int iLOSS = Lt::oRED + 18, iGN = Lt::oRED + 19, iENT = Lt::oRED + 20, iCLIP = Lt::oRED + 21, iADV = Lt::oRED + 22, iRET = Lt::oRED + 23,
                iPEN = Lt::oRED + 24, iCUR = Lt::oRED + 25, iCLOSS = Lt::oRED + 26, iPLOSS = Lt::oRED + 27;
  if (threadIdx.x == 0) { for (int k = 18; k < 28; ++k) sm[Lt::oRED + k] = 0.f; }
  float inf_kl = 0.f;
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
  // observation rows: four lanes per sample (sample lane >> 2), lane part q = lane & 3 takes the NXL consecutive features q NXL .. -- one row-number shuffle per
  // fetch instead of one per element, no division by IN (with the element-major map the 17-wide family spent 7 % of its step here)
  constexpr int NXL = (IN + 3) / 4;
  float px[NXL]; float p_lp = 0.f, p_adv = 0.f, p_ret = 0.f; float p_act[NACT]; int p_valid = 0; uint8_t p_abyte[OUT];
#pragma unroll
  for (int k = 0; k < OUT; ++k) p_abyte[k] = 0;
#pragma unroll
  for (int k = 0; k < NACT; ++k) p_act[k] = 0.f;
  int n_row = 0, n_valid = 0;
  int n_row2 = -1; float p_cost2 = 0.f, p_ee2 = 0.f, p_cadv = 0.f;      // LAG: row tid of the whole minibatch (both halves), its :cost and :episode_end; :cost_advantage of the own sample
  auto fetch_index = [&](const int32_t* ord, int64_t st, int nb) {
    const int sidx = (NWG == 2 ? 64 * p : 0) + 16 * w + c;
    n_valid = sidx < nb ? 1 : 0;
    n_row = n_valid ? (a.ids ? CRUX_GLOBAL_PTR(int32_t, a.ids)[st + sidx] : CRUX_GLOBAL_PTR(int32_t, ord)[st + sidx]) : 0;
    if constexpr (LAG) n_row2 = tid < nb ? (a.ids ? CRUX_GLOBAL_PTR(int32_t, a.ids)[st + tid] : CRUX_GLOBAL_PTR(int32_t, ord)[st + tid]) : -1;
  };
  auto fetch_data = [&]() {
    const int rowlo = n_row; p_valid = n_valid; const int64_t row = rowlo;
    const int rs = __shfl(rowlo, lane >> 2, 64), vs = __shfl(p_valid, lane >> 2, 64);      // lanes 0..15 hold the rows of samples 0..15
    const float* xrow = CRUX_GLOBAL_PTR(float, a.S) + (int64_t)rs * IN + (lane & 3) * NXL;
#pragma unroll
    for (int e = 0; e < NXL; ++e) px[e] = ((lane & 3) * NXL + e < IN && vs) ? xrow[e] : 0.f;
    p_lp = 0.f; p_adv = 0.f; p_ret = 0.f;
    if constexpr (LAG) { p_cost2 = n_row2 >= 0 ? CRUX_GLOBAL_PTR(float, a.COST)[n_row2] : 0.f; p_ee2 = (n_row2 >= 0 && CRUX_GLOBAL_PTR(uint8_t, a.EE)[n_row2]) ? 1.f : 0.f;
      p_cadv = (lane < 16 && p_valid) ? CRUX_GLOBAL_PTR(float, a.CADV)[row] : 0.f; }
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
    for (int e = 0; e < NXL; ++e) { const int f = (lane & 3) * NXL + e; if (f < IN) xs[(lane >> 2) * XP + f] = px[e]; }
    if (KIND == MFK_CATEGORICAL) { int ai = 0;
#pragma unroll
      for (int k = 0; k < OUT; ++k) ai = p_abyte[k] ? k : ai;
      p_act[0] = (float)ai; }
    if constexpr (LAG) { if (tid < 128) { lag_cost[tid] = p_cost2; lag_ee[tid] = p_ee2; } if (lane < 16) lag_cadv[w * 16 + lane] = p_cadv; }
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
    staged = false;
    { const int nb0 = (int)(total_rows < a.bs ? total_rows : a.bs); fetch_index(order_cur, 0, nb0); fetch_data();
      const int64_t st1 = a.bs; const int nb1 = st1 < total_rows ? (int)((total_rows - st1) < a.bs ? (total_rows - st1) : a.bs) : 0; fetch_index(order_cur, st1 < total_rows ? st1 : 0, nb1); }
    for (int64_t st = 0; st < total_rows; st += a.bs) {
      const int nb = (int)((total_rows - st) < a.bs ? (total_rows - st) : a.bs);
      const float invB = 1.0f / (float)nb;
      ak.c1 = __builtin_amdgcn_rcpf((float)(1.0 - bp1)); ak.c2 = __builtin_amdgcn_rcpf((float)(1.0 - bp2));
      MX_T(0);
      const bool staged_now = !staged;
      if (!staged) stage();                        // normally done already, inside the previous step's exchange wait
      staged = false;
      if constexpr (LAG) {   // the penalty update inside the loss (ppo.jl:80-116), once per evaluation: sums of the minibatch's :cost and :episode_end, then the controller
        if (staged_now) __syncthreads();            // the first minibatch of an epoch is staged here, not under the previous step's barriers (uniform: staged is)
        double sc_ = (double)lag_cost[lane] + (double)lag_cost[lane + 64], ne_ = (double)lag_ee[lane] + (double)lag_ee[lane + 64];      // Float32 terms: any summation order gives the same Float64 sum
#pragma unroll
        for (int o_ = 32; o_ >= 1; o_ >>= 1) { sc_ += __shfl_xor(sc_, o_, 64); ne_ += __shfl_xor(ne_, o_, 64); }
        const float Jc = (float)sc_ / (float)ne_;                                      // :84-88
        const float dl = Jc - lg.target_cost;                                         // :91
        { const float x = lg.I + lg.Ki * dl; lg.I = x > lg.Ki_max ? lg.Ki_max : (x < 0.f ? 0.f : x); }             // :94 clamp(I + Ki*Delta, 0, Ki_max)
        lg.smooth_delta = (float)(lg.ema_alpha * (double)lg.smooth_delta + (1.0 - lg.ema_alpha) * (double)dl);      // :98 (Float64 arithmetic, Float32 store)
        lg.smooth_Jc = (float)(lg.ema_alpha * (double)lg.smooth_Jc + (1.0 - lg.ema_alpha) * (double)Jc);            // :99
        { const float x = lg.smooth_Jc - lg.Jc_prev; lg.deriv_term = (x != x) ? x : (x > 0.f ? x : 0.f); }           // :102 max(0, .) keeps NaN
        lg.Jc_prev = lg.smooth_Jc;                                                    // :105
        { const float x = (lg.Kp * lg.smooth_delta + lg.I) + lg.Kd * lg.deriv_term; pen = x > lg.penalty_max ? lg.penalty_max : (x < 0.f ? 0.f : x); }   // :108
        lg.penalty = pen; lg.cur_cost = Jc;
      }
      if (st + a.bs < total_rows) fetch_data();
      { const int64_t st2 = st + 2 * (int64_t)a.bs; const int nb2 = st2 < total_rows ? (int)((total_rows - st2) < a.bs ? (total_rows - st2) : a.bs) : 0;
        fetch_index(order_cur, st2 < total_rows ? st2 : 0, nb2); }

      MX_T(1);
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
      MX_T(2);
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

      MX_T(3);
      // ======================= layer 3 (VALU) + loss head =======================
      // W3 fragments: kept in registers across the head for narrow outputs (32 VGPRs at OUT = 2); for wider heads (OUT = 6: 96 VGPRs, which made the
      // 17->64->64->6 kernels spill) they are re-read from LDS in the backward pass instead (conflict-free b128 reads)
      constexpr bool W3_REG = OUT <= 2;
      f32x4 w3[W3_REG ? OUT : 1][4];
      if (W3_REG) {
#pragma unroll
        for (int o = 0; o < OUT; ++o)
#pragma unroll
          for (int m = 0; m < 4; ++m) w3[o][m] = *(const f32x4*)&sm[Lt::oW3R + o * MF_HID + 16 * m + 4 * g]; }
      float z[OUT];
#pragma unroll
      for (int o = 0; o < OUT; ++o) { float acc = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) { const f32x4 wv = W3_REG ? w3[W3_REG ? o : 0][m] : *(const f32x4*)&sm[Lt::oW3R + o * MF_HID + 16 * m + 4 * g];
#pragma unroll
          for (int r = 0; r < 4; ++r) acc = fmaf(wv[r], h2[m][r], acc); }
        z[o] = g4_sum(acc) + sm[Lt::oB3 + o]; }
      float dz[OUT], dex[OUT];
      float s_lossp = 0.f, s_H = 0.f, s_kl = 0.f, s_adv = 0.f, s_ret = 0.f, s_clip = 0.f, s_sq = 0.f, s_cost = 0.f;
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
          float gcr = 0.f;                                                          // lagrange: d/dr of max(r Ac, clamp(r) Ac) times r (ppo.jl:119)
          if constexpr (LAG) { const float Ac = lag_cadv[w * 16 + c]; const float uc = r * Ac, clc = rc * Ac; s_cost = cnt * (uc >= clc ? uc : clc); gcr = (uc >= clc ? Ac : 0.f) * r; }
#pragma unroll
          for (int k = 0; k < OUT; ++k) { const float dlogpi = ((k == ai) ? 1.f : 0.f) - pk[k];
            const float base = -a.lambda_p * coef * dlogpi - a.lambda_e * (pk[k] * (hk[k] - hp));
            dz[k] = !valid ? 0.f : (LAG ? invB * ((base + pen * gcr * dlogpi) / (1.f + pen)) : invB * base); }
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
          float cf = -a.lambda_p * coef;
          if constexpr (LAG) { const float Ac = lag_cadv[w * 16 + c]; const float uc = r * Ac, clc = rc * Ac; s_cost = cnt * (uc >= clc ? uc : clc);
            cf = (cf + pen * ((uc >= clc ? Ac : 0.f) * r)) / (1.f + pen); }
#pragma unroll
          for (int k = 0; k < OUT; ++k) { dz[k] = valid ? invB * (cf * (dd[k] * s2[k])) : 0.f;
            dex[k] = valid ? invB * (cf * (((dd[k] * dd[k]) * s2[k]) * inr[k] - 1.f)) : 0.f; }
          s_lossp = cnt * lterm; s_kl = cnt * (oldlp - newlp); s_adv = cnt * A; s_ret = cnt * R; s_clip = cnt * clipv;
        }
      }

      MX_T(4);
      // ======================= backward, own samples =======================
#pragma unroll
      for (int o = 0; o < OUT; ++o) { float pv[16];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) pv[4 * m + r] = dz[o] * h2[m][r];
        part[Lt::pW3 + o * MF_HID + 16 * (c >> 2) + 4 * g + (c & 3)] = row16_reduce_scatter(pv, c); }
      { f32x4 d2[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) d2[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < OUT; ++o)          // same fma order over o as before: d = fma(w3[o], dz[o], d)
#pragma unroll
          for (int m = 0; m < 4; ++m) { const f32x4 wv = W3_REG ? w3[W3_REG ? o : 0][m] : *(const f32x4*)&sm[Lt::oW3R + o * MF_HID + 16 * m + 4 * g];
#pragma unroll
            for (int r = 0; r < 4; ++r) d2[m][r] = fmaf(wv[r], dz[o], d2[m][r]); }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) h2[m][r] = actg<ACT>(h2[m][r], d2[m][r]); }
      { constexpr int NVB = 7 + OUT + (KIND == MFK_GAUSSIAN ? OUT : 0), NV = NVB + (LAG ? 1 : 0);
        float mv[((NV + 15) / 16) * 16];
#pragma unroll
        for (int k = 0; k < ((NV + 15) / 16) * 16; ++k) mv[k] = 0.f;
        mv[0] = s_lossp; mv[1] = s_H; mv[2] = s_kl; mv[3] = s_adv; mv[4] = s_ret; mv[5] = s_clip; mv[6] = s_sq;
#pragma unroll
        for (int o = 0; o < OUT; ++o) { mv[7 + o] = dz[o]; if (KIND == MFK_GAUSSIAN) mv[7 + OUT + o] = dex[o]; }
        if constexpr (LAG) mv[NVB] = s_cost;             // the cost term of the loss rides behind the head's gradient sums
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
      MX_T(5);
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
      MX_T(6);
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
      MX_T(7);
      __syncthreads();   // ---- B_a: all tiles and small partials are visible
      MX_T(8);

      // ======================= this wave's dW2 tiles over the workgroup's samples (all 128, or its 64 of them) =======================
      f32x4 gW2[WT];
#pragma unroll
      for (int mm = 0; mm < WT; ++mm) gW2[mm] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ws = 0; ws < MF8_NW; ++ws) {
        const float* t2 = sm + Lt::oT2 + ws * Lt::TILE; const float* t1 = sm + Lt::oT1 + ws * Lt::TILE;
        const f32x4 av = *(const f32x4*)&t2[t_rd + 256 * mp0];          // A[i=c -> o=16mp0+c][k -> sample 4g+r]
#pragma unroll
        for (int mm = 0; mm < WT; ++mm) { const f32x4 bv = *(const f32x4*)&t1[t_rd + 256 * (m0 + mm)];   // B[k -> sample][j=c -> i]
#pragma unroll
          for (int r = 0; r < 4; ++r) gW2[mm] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], bv[r], gW2[mm], 0, 0, 0); }
      }
      MX_T(9);
      // small parameters: reduce the per-wave partials of this workgroup
      float gs[NSI];
#pragma unroll
      for (int k = 0; k < NSI; ++k) { float gsum = 0.f;
        if (so_ok[k]) { const int po = Lt::oPART + so_part[k];
          gsum = sm[po];
#pragma unroll
          for (int q = 1; q < MF8_NW; ++q) gsum += sm[po + q * Lt::PART]; }
        gs[k] = gsum; }
      constexpr int stat_hi = LAG ? NT : NT - 1;          // lanes NT - 8 .. carry the statistics sums (7, and the cost term of lagrange_ppo_loss)
      float stat_loc = 0.f;
      if (tid >= NT - 8 && tid < NT - 1) { const int k = tid - (NT - 8); stat_loc = sm[Lt::oPART + Lt::pST + k];   // stat sums, by 7 lanes of the last wave
#pragma unroll
        for (int q = 1; q < MF8_NW; ++q) stat_loc += sm[Lt::oPART + q * Lt::PART + Lt::pST + k]; }
      if constexpr (LAG) { if (tid == NT - 1) { constexpr int pc_ = Lt::pMISC + 7 + OUT + (KIND == MFK_GAUSSIAN ? OUT : 0); stat_loc = sm[Lt::oPART + pc_];
#pragma unroll
          for (int q = 1; q < MF8_NW; ++q) stat_loc += sm[Lt::oPART + q * Lt::PART + pc_]; } }
      float stat_tot = stat_loc;
      // ---- exchange the partial gradients with the other workgroup through the shared L2 (see the header) ----
      if constexpr (NWG == 2) { float* mine = a.xbuf + (size_t)(((int)(xstep & 1) * 2 + p)) * XSLOT; const float* peer = a.xbuf + (size_t)(((int)(xstep & 1) * 2 + (1 - p))) * XSLOT;
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) *(f32x4*)&mine[tid * 16 + 4 * mm] = gW2[mm];
#pragma unroll
        for (int k = 0; k < NSI; ++k) mine[4096 + tid + NT * k] = gs[k];
        if (tid >= NT - 8 && tid < stat_hi) mine[4096 + NSI * NT + (tid - (NT - 8))] = stat_loc;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every store of this lane has reached the L2
        MX_T(10);
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(a.xctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // arrive (one atomic per workgroup: per-wave arrivals measured slower, 9.37 vs 9.19 us)
        // the ~1 us until the other workgroup arrives is spent staging the NEXT minibatch (rows prefetched a step ago; x/scalar tiles are free after B_a).
        // (Staging before the s_waitcnt instead, inside the store acknowledgement latency, measured slower: 8.85 vs 8.76 us per step.)
        if (st + a.bs < total_rows) { stage(); staged = true; }
        if (tid == 0) {
          const unsigned want = 2u * (unsigned)(xstep + 1); unsigned spins = 0; bool ok = true;
          while (__hip_atomic_load(a.xctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) { __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 24) || __hip_atomic_load(a.xctr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = false; break; } }   // never hang the GPU
          float why = ok ? 0.f : 1.f;                                   // 1: the other workgroup never arrived (or raised the abort word)
          if (ok && xstep == 0) {   // the unfenced exchange is only coherent inside one XCD's L2: refuse to train if the two workgroups were placed on different XCDs
            const unsigned peer_xcc = __hip_atomic_load(a.xctr + 8 + (1 - p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (peer_xcc != my_xcc + 1u) { ok = false; why = 2.f; } }       // 2: the two workgroups of this learner sit on different XCDs
          if (!ok) { __hip_atomic_store(a.xctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (PX) for (int r = 0; r < a.px_n; ++r) __hip_atomic_store((unsigned*)(a.px_tab[r] + CRUX_PX_ABORT), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }   // the replicas stop waiting for this one
          sm[Lt::oRED + 16] = why;
        }
        __syncthreads();
        if (sm[Lt::oRED + 16] != 0.f) { err = CRUX_EHIP; why_failed = (int)sm[Lt::oRED + 16]; break; }
        MX_T(11);
        f32x4 pw[4]; float pg[NSI]; float ps = 0.f;
        // all loads of the peer's slot are in flight together (one L2 round trip): the dword loads first, then the b128 block whose wait covers them
#pragma unroll
        for (int k = 0; k < NSI; ++k) pg[k] = __hip_atomic_load(peer + 4096 + tid + NT * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid >= NT - 8 && tid < stat_hi) ps = __hip_atomic_load(peer + 4096 + NSI * NT + (tid - (NT - 8)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:16 sc1\n\tglobal_load_dwordx4 %2, %4, off offset:32 sc1\n\t"
                     "global_load_dwordx4 %3, %4, off offset:48 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(pw[0]), "=&v"(pw[1]), "=&v"(pw[2]), "=&v"(pw[3]) : "v"(peer + tid * 16) : "memory");
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) gW2[mm] += pw[mm];      // a+b == b+a bitwise: both workgroups hold the same total
#pragma unroll
        for (int k = 0; k < NSI; ++k) gs[k] += pg[k];
        stat_tot = stat_loc + ps;
        if constexpr (PX) {
          // ---- SUM all-reduce of the local gradient over the replica group, between the pullback (training.jl:18) and Flux.update! (:21) ----
          // Both workgroups hold the same local total. They share the writes (peer i of the N-1 goes to workgroup i & 1): the total and the seven
          // statistics sums go into slot [parity][my rank] of the peer's region, a system-scope release makes them visible, then flag[my rank]
          // there is raised to the exchange number. Both workgroups then wait for the N-1 flags in the OWN region and add the N contributions in
          // rank order -- the own one from registers (nothing orders workgroup 1 after a store of workgroup 0, so it is never read back from a
          // slot), the others from the slots -- so every workgroup of every rank forms the same sum bit for bit.
          const unsigned long long xg = px0 + (unsigned long long)xstep;       // number of this exchange on this learner stream
          const int par = (int)(xg & 1ull);
          { int pi_ = 0;
            for (int r = 0; r < a.px_n; ++r) {
              if (r == a.px_rank || (pi_++ & 1) != p) continue;
              float* dst = a.px_tab[r] + (size_t)(par * CRUX_PX_MAXR + a.px_rank) * CRUX_PX_SLOT;
#pragma unroll
              for (int mm = 0; mm < 4; ++mm) *(f32x4*)&dst[tid * 16 + 4 * mm] = gW2[mm];
#pragma unroll
              for (int k = 0; k < NSI; ++k) dst[4096 + tid + NT * k] = gs[k];
              if (tid >= NT - 8 && tid < stat_hi) dst[4096 + NSI * NT + (tid - (NT - 8))] = stat_tot; } }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // every wave drains its own slot stores; the workgroup meets; ONE lane issues the system-scope
          __syncthreads();                                                      // release (the L2 write-back covers the lines of all waves) and drains it before the flags go out
          if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            int pi_ = 0;
            for (int r = 0; r < a.px_n; ++r) { if (r == a.px_rank) continue;
              if ((pi_++ & 1) != p) continue;
              // relaxed: the release above covers the slot stores of every wave (all drained before the barrier)
              __hip_atomic_store((unsigned long long*)(a.px_tab[r] + CRUX_PX_FLAGS) + 8 * a.px_rank, xg + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
            const long long t0 = wall_clock64();                // 100 MHz; the wait is bounded four ways (peer_wait.h)
            const unsigned gave_up = px_wait_peers(px_mine, a.px_n, a.px_rank, xg + 1ull, t0, a.px_timeout, p, true); const bool ok = gave_up == 0u;
            if (!ok) { px_raise_abort(a.px_tab, a.px_n, gave_up);
              __hip_atomic_store(a.xctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            if (a.px_hist) {      // how long this workgroup waited for the slowest peer's flag (10 ns ticks, log2 bins): the selftest's view of the xGMI hand-off
              const unsigned long long dtk = (unsigned long long)(wall_clock64() - t0) | 1ull;
              unsigned* hb = (unsigned*)(px_mine + CRUX_PX_HIST) + 32 * p + (63 - __builtin_clzll(dtk) > 31 ? 31 : 63 - __builtin_clzll(dtk));
              *hb = *hb + 1u; }
            sm[Lt::oRED + 16] = ok ? 0.f : (float)(16u + gave_up);      // 16 + bound (peer_wait.h): the replica group ended this launch
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");                       // system scope, one lane (the slot loads below are sc0 sc1 and pass the L1 anyway)
          }
          __syncthreads();
          if (sm[Lt::oRED + 16] != 0.f) { err = CRUX_EHIP; why_failed = (int)sm[Lt::oRED + 16]; break; }
          // the N - 1 slots are read two ranks at a time (all loads of a pair in flight together) and added in rank order
          f32x4 oW[4]; float oS[NSI]; const float oT = stat_tot;
#pragma unroll
          for (int mm = 0; mm < 4; ++mm) oW[mm] = gW2[mm];
#pragma unroll
          for (int k = 0; k < NSI; ++k) oS[k] = gs[k];
          auto px_load = [&](int r, f32x4 (&vW)[4], float (&vS)[NSI], float& vT) {
            if (r == a.px_rank) {
#pragma unroll
              for (int mm = 0; mm < 4; ++mm) vW[mm] = oW[mm];
#pragma unroll
              for (int k = 0; k < NSI; ++k) vS[k] = oS[k];
              vT = oT; return; }
            const float* src = px_mine + (size_t)(par * CRUX_PX_MAXR + r) * CRUX_PX_SLOT;
#pragma unroll
            for (int k = 0; k < NSI; ++k) vS[k] = __hip_atomic_load(src + 4096 + tid + NT * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            vT = 0.f;
            if (tid >= NT - 8 && tid < stat_hi) vT = __hip_atomic_load(src + 4096 + NSI * NT + (tid - (NT - 8)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n\tglobal_load_dwordx4 %1, %4, off offset:16 sc0 sc1\n\tglobal_load_dwordx4 %2, %4, off offset:32 sc0 sc1\n\t"
                         "global_load_dwordx4 %3, %4, off offset:48 sc0 sc1"
                         : "=&v"(vW[0]), "=&v"(vW[1]), "=&v"(vW[2]), "=&v"(vW[3]) : "v"(src + tid * 16) : "memory");
          };
          // The slots are read PXS ranks at a time -- all loads of a batch in flight together, one round trip to the fine-grained region per batch -- and added in
          // rank order. Heads of up to four outputs have the registers for four at a time (8 replicas = two round trips); the six-output heads sit at the
          // 512-register limit and take two (16 + NSI live registers per slot in flight; with four the step loop spills, with two only prologue / epilogue values do).
          constexpr int PXS = (OUT <= 4) ? 4 : 2;
          for (int r0 = 0; r0 < a.px_n; r0 += PXS) {
            f32x4 vW[PXS][4]; float vS[PXS][NSI]; float vT[PXS];
#pragma unroll
            for (int q = 0; q < PXS; ++q) {
              if (r0 + q < a.px_n) px_load(r0 + q, vW[q], vS[q], vT[q]);
              else { vT[q] = 0.f;
#pragma unroll
                for (int mm = 0; mm < 4; ++mm) vW[q][mm] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < NSI; ++k) vS[q][k] = 0.f; } }
            if constexpr (PXS == 4) {
              asm volatile("s_waitcnt vmcnt(0)" : "+v"(vW[0][0]), "+v"(vW[0][1]), "+v"(vW[0][2]), "+v"(vW[0][3]), "+v"(vW[1][0]), "+v"(vW[1][1]), "+v"(vW[1][2]), "+v"(vW[1][3]),
                                                  "+v"(vW[2][0]), "+v"(vW[2][1]), "+v"(vW[2][2]), "+v"(vW[2][3]), "+v"(vW[3][0]), "+v"(vW[3][1]), "+v"(vW[3][2]), "+v"(vW[3][3]) :: "memory");
            } else if constexpr (PXS == 2) {
              asm volatile("s_waitcnt vmcnt(0)" : "+v"(vW[0][0]), "+v"(vW[0][1]), "+v"(vW[0][2]), "+v"(vW[0][3]), "+v"(vW[1][0]), "+v"(vW[1][1]), "+v"(vW[1][2]), "+v"(vW[1][3]) :: "memory");
            } else {
              asm volatile("s_waitcnt vmcnt(0)" : "+v"(vW[0][0]), "+v"(vW[0][1]), "+v"(vW[0][2]), "+v"(vW[0][3]) :: "memory");
            }
#pragma unroll
            for (int q = 0; q < PXS; ++q) {
              if (r0 + q >= a.px_n) break;
              if (r0 + q == 0) {
#pragma unroll
                for (int mm = 0; mm < 4; ++mm) gW2[mm] = vW[q][mm];
#pragma unroll
                for (int k = 0; k < NSI; ++k) gs[k] = vS[q][k];
                stat_tot = vT[q];
              } else {
#pragma unroll
                for (int mm = 0; mm < 4; ++mm) gW2[mm] += vW[q][mm];
#pragma unroll
                for (int k = 0; k < NSI; ++k) gs[k] += vS[q][k];
                stat_tot += vT[q];
              }
            }
          }
          // mean over the group: global minibatch = px_n x nb samples, every rank's partial was already divided by nb
#pragma unroll
          for (int mm = 0; mm < 4; ++mm) gW2[mm] = gW2[mm] * px_inv;
#pragma unroll
          for (int k = 0; k < NSI; ++k) gs[k] = gs[k] * px_inv;
          stat_tot = stat_tot * px_inv;
        }
        xstep += 1;
      }
      if (tid >= NT - 8 && tid < stat_hi) sm[Lt::oRED + 8 + (tid - (NT - 8))] = stat_tot;
      float ssq = 0.f; int bad = 0;
#pragma unroll
      for (int k = 0; k < NSI; ++k) if (so_ok[k]) {
        if (KIND == MFK_GAUSSIAN && so_ex[k]) gs[k] += LAG ? -a.lambda_e / (1.f + pen) : -a.lambda_e;      // d(-lambda_e H)/dlogSigma, H = const + sum(logSigma); lagrange: the whole loss is divided by 1 + penalty
        ssq += gs[k] * gs[k]; bad |= isnan(gs[k]) ? 1 : 0; }
#pragma unroll
      for (int mm = 0; mm < WT; ++mm)
#pragma unroll
        for (int r = 0; r < 4; ++r) { ssq += gW2[mm][r] * gW2[mm][r]; bad |= isnan(gW2[mm][r]) ? 1 : 0; }
      ssq = wave_sum(ssq);
      if (lane == 0) sm[Lt::oRED + w] = ssq;
      MX_T(12);
      const int any_bad = __syncthreads_or(bad);   // ---- B_or (also publishes RED)
      MX_T(13);
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
          if (KIND != MFK_VALUE) inf_kl = t[2] * invB;
          if (tid == 0) {
          sm[iGN] = sqrtf(ss);
          if (KIND == MFK_VALUE) { sm[iLOSS] = t[6] * invB; sm[iRET] = t[4] * invB; }
          else { const float p_loss = -(t[0] * invB); float entropy;
            if (KIND == MFK_CATEGORICAL) entropy = t[1] * invB;
            else { entropy = 1.4189385332046727f;
#pragma unroll
              for (int k = 0; k < OUT; ++k) entropy += sm[Lt::oEX + k]; }
            sm[iENT] = entropy; sm[iLOSS] = a.lambda_p * p_loss + a.lambda_e * (-entropy); sm[iADV] = t[3] * invB; sm[iRET] = t[4] * invB; sm[iCLIP] = t[5] * invB;
            if constexpr (LAG) { const float cost_loss = pen * (t[7] * invB);                                        // ppo.jl:119
              sm[iLOSS] = ((a.lambda_p * p_loss + a.lambda_e * (-entropy)) + cost_loss) / (1.f + pen);                  // :131
              sm[iPEN] = pen; sm[iCUR] = lg.cur_cost; sm[iCLOSS] = cost_loss; sm[iPLOSS] = a.lambda_p * p_loss; } }
          }
        }
      }
      if (any_bad) { if (tid == 0) sm[iGN] = NAN; err = CRUX_ENAN; break; }      // training.jl:20: no update
      // ======================= Adam (Flux.update!, training.jl:21) =======================
      if (a.apply) {
#pragma unroll
        for (int mm = 0; mm < WT; ++mm) {
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
      } else if (p == 0) {   // gradient-only mode (crux_loss_grad): export the flat gradient
#pragma unroll
        for (int mm = 0; mm < WT; ++mm)
#pragma unroll
          for (int r = 0; r < 4; ++r) a.g[Lt::cW2 + (16 * mp0 + 4 * g + r) + MF_HID * (16 * (m0 + mm) + c)] = gW2[mm][r];
#pragma unroll
        for (int k = 0; k < NSI; ++k) { const int s = tid + NT * k; if (s < ns_valid) a.g[s_canon(s)] = gs[k]; }
      }
      MX_T(14);
      __syncthreads();   // ---- B_b: masters updated; tiles and partials may be overwritten
      MX_T(15);
      total_batches += 1;
      if (a.max_batches > 0 && total_batches >= a.max_batches) break;          // training.jl:45
      if (a.target_kl >= 0.f && KIND != MFK_VALUE && inf_kl > a.target_kl) break;   // :46
    }
    if (err) break;
    if (tid == 0 && p == 0 && a.epoch_infos) { float* e = a.epoch_infos + (size_t)ep * CRUX_INFO_N;   // aggregate_info(minibatch_infos) == last minibatch (Q3)
      for (int k = 0; k < CRUX_INFO_N; ++k) e[k] = 0.f;
      e[CRUX_INFO_LOSS] = sm[iLOSS]; e[CRUX_INFO_GRAD_NORM] = sm[iGN];
      if (KIND != MFK_VALUE) { e[CRUX_INFO_ENTROPY] = sm[iENT]; e[CRUX_INFO_KL] = inf_kl; e[CRUX_INFO_CLIP_FRACTION] = sm[iCLIP]; e[CRUX_INFO_AVG_ADVANTAGE] = sm[iADV]; e[CRUX_INFO_AVG_RETURN] = sm[iRET]; }
      if constexpr (LAG) { e[CRUX_INFO_PENALTY] = sm[iPEN]; e[CRUX_INFO_CUR_COST] = sm[iCUR]; e[CRUX_INFO_COST_LOSS] = sm[iCLOSS]; e[CRUX_INFO_P_LOSS] = sm[iPLOSS]; } }
    epochs_run += 1;
    if (a.target_kl >= 0.f && KIND != MFK_VALUE && inf_kl > a.target_kl) stop = true;   // :49
    if (a.max_batches > 0 && total_batches >= a.max_batches) stop = true;               // :50
  }
  // ---- write back parameters and Adam state --------------------------------------------------------------------
  __syncthreads();
  if (a.apply && p == 0) {
#pragma unroll
    for (int mm = 0; mm < WT; ++mm)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int pc = Lt::cW2 + (16 * mp0 + 4 * g + r) + MF_HID * (16 * (m0 + mm) + c);
        a.p[pc] = tW2[mm][r]; a.m[pc] = mW2[mm][r]; a.v[pc] = vW2[mm][r]; }
    for (int s = tid; s < ns_valid; s += NT) { const int pc = s_canon(s); a.p[pc] = sm[s_master(s)]; a.m[pc] = sm[Lt::oMS + s]; a.v[pc] = sm[Lt::oVS + s]; }
  }
  if (TIMING && lane == 0 && a.dbg) { for (int k = 0; k < 16; ++k) a.dbg[(4 * p + w) * 16 + k] = tacc[k]; }
  if (PX && tid == 0 && p == 0) *(unsigned long long*)(px_mine + CRUX_PX_COUNT) = px0 + (unsigned long long)xstep;
  if (tid == 0 && (p == 0 || err)) {
    a.status[0] = err; a.status[1] = (int32_t)total_batches; a.status[2] = epochs_run; a.status[3] = (order_cur == a.order_a) ? 0 : 1;
    if (err == CRUX_EHIP) a.status[4] = why_failed;      // 1 local workgroup missing, 2 workgroups on different XCDs, 3 replica group timeout / abort
    a.bp[0] = bp1; a.bp[1] = bp2;
    if constexpr (LAG) { if (p == 0) *a.lag = lg; }
    if (err && a.epoch_infos && epochs_run == 0) { a.epoch_infos[CRUX_INFO_LOSS] = sm[iLOSS]; a.epoch_infos[CRUX_INFO_GRAD_NORM] = NAN; }
  }
}

