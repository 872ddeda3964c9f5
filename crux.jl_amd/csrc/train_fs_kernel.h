// train_fs_kernel.h -- the persistent batch_train! kernel of the register-resident IN->64->64->OUT family in its FEATURE-SPLIT form (round 3; src/training.jl:13-55,
// ppo.jl:4-21,59-60, Flux Adam). Same arithmetic pieces as train_mfma_kernel.h, another decomposition of one minibatch step:
//
//   * a 16-sample tile of the minibatch belongs to a PAIR of waves (t, h), h in {0, 1}: wave h computes the hidden features [32h, 32h + 32) of the second layer, of its
//     gradient and of the first layer's gradient -- half of the 64x64 MFMA work (forward 32, dH1 32 instead of 64 each) and half of the per-feature VALU / LDS work of the
//     tile. The first layer (IN <= 17 inputs: 4..20 MFMAs) is evaluated by both waves, so the second layer needs no exchange; the pair meets twice per step through LDS:
//     the partial logits z (OUT values per sample) before the loss head, and the other half of dZ2 (A operand of dH1 = W2' dZ2) after it;
//   * NWG workgroups (compute units of ONE XCD) share the 128 samples: NWG = 2 -> 64 samples = 4 tiles = 8 waves per workgroup (two waves per SIMD),
//     NWG = 4 -> 32 samples = 2 tiles = 4 waves per workgroup (one wave per SIMD, four compute units per learner). Per wave and step: 4 (L1) + 32 (L2) + 32 (dW2)
//     + 32 (dH1) + 8 (dW1) = 108 MFMAs against 212 in the sample-split form -- the step loop is instruction-issue bound, and this halves the per-feature
//     instruction stream of a wave;
//   * the 64x64 weight gradient stays model-parallel over the waves of a workgroup (16 / NW tiles of W2 with theta, m, v in the owning wave's registers); it is formed
//     BEFORE dH1 and its partial sums leave for the other workgroups' L2 slots at once, so the store acknowledgement is covered by the rest of the backward pass;
//   * the NWG partial gradients are exchanged once per step through the shared L2 exactly like the two-CU form does (plain stores, s_waitcnt, one agent-scope arrival
//     counter, sc1 loads) and added in workgroup order by everyone: bit-identical totals, Adam updates and early-stopping decisions in all workgroups.
// Covers full minibatch loops (batch_train! with Adam, 65..128 rows per minibatch) of the plain policy-gradient / critic losses, of lagrange_ppo_loss (LAG) and of replica
// groups (PX); single steps and gradient-only calls stay on train_mfma_kernel.h.
#pragma once
#include <type_traits>
#include "train_args.h"

#include "mfma_helpers.h"
#include "peer_wait.h"

#define FS_LD 72
__device__ __forceinline__ int fs_tx(int q) { return (4 - q) & 3; }   // {0,3,2,1}

// H2: width of the second hidden layer, 64 or 32 (the first is MF_HID = 64): the reference's HalfCheetah PPO networks are 17-64-32-6 / 17-64-32-1
// (examples/rl/half_cheetah_mujoco.jl:33-38); a wave's half of the layer is then ONE 16-feature tile instead of two.
template <int IN, int OUT, int NWG, bool HELP = false, int H2 = 64, bool LAG = false>
struct FsLayout {
  static_assert(H2 == 64 || H2 == 32, "second hidden layer: 64 or 32 units");
  static constexpr int MH = H2 / 32, NT2 = H2 / 16, HH = H2 / 2, W2N = H2 * MF_HID;      // 16-feature tiles per half / per layer, features per half, elements of W2
  static constexpr int NWC = 16 / NWG, TILES = NWC / 2;                    // compute waves (a pair per 16-sample tile)
  static constexpr int NW = HELP ? 2 * NWC : NWC, NT = 64 * NW;            // + as many helper waves: owners of half the W2 tiles, and the minibatch staging
  static constexpr int NXB = HELP ? 2 : 1;                                 // staging rows are double-buffered when the helpers stage the next minibatch during the step
  static constexpr int KS0 = (IN + 3) / 4, IP = KS0 * 4, JT = (IN + 15) / 16, XP = IP + 2, W1LD = IP + 2;
  static constexpr int SCW = ((4 + (OUT > 4 ? OUT : 4)) | 1) + (LAG ? 2 : 0);      // lagrange_ppo_loss: :cost_advantage of the sample rides in the last slot
  static constexpr int ZW = OUT;                                                          // partial logits per (tile, half, g, sample)
  // flat index spaces of the small parameters (everything but W2): s = thread-owned index, c = canonical (Flux.params) index, p = index inside a tile's partial block
  static constexpr int sW1 = 0, sB1 = MF_HID * IN, sB2 = sB1 + MF_HID, sW3 = sB2 + H2, sB3 = sW3 + H2 * OUT, sEX = sB3 + OUT, NS = sEX + 16;
  static constexpr int cW1 = 0, cB1 = MF_HID * IN, cW2 = cB1 + MF_HID, cB2 = cW2 + W2N, cW3 = cB2 + H2, cB3 = cW3 + H2 * OUT, cEX = cB3 + OUT;
  static constexpr int W1ROWS = IP < 16 * JT ? IP : 16 * JT;
  static constexpr int pW1 = 0, pB1 = pW1 + W1ROWS * FS_LD, pB2 = pB1 + MF_HID, pW3 = pB2 + H2, pMISC = pW3 + OUT * H2;
  static constexpr int pST = pMISC, pB3 = pMISC + 7, pEX = pB3 + OUT;
  static constexpr int PART = ((pMISC + 48 + 3) / 4) * 4;
  static constexpr int NSP = ((NS + 3) / 4) * 4;
  static constexpr int TILE = MF_HID * 16, TILE2 = H2 * 16;
  static constexpr int oW2R = 0, oW2C = oW2R + H2 * FS_LD, oW1R = oW2C + MF_HID * FS_LD;      // W2R: [H2 rows o][i], W2C: [64 rows i][o]
  static constexpr int oB1 = oW1R + MF_HID * W1LD, oB2 = oB1 + MF_HID, oW3R = oB2 + H2, oB3 = oW3R + OUT * H2, oEX = oB3 + 16;
  static constexpr int oMS = ((oEX + 16 + 3) / 4) * 4, oVS = oMS + NSP;
  static constexpr int oT1 = oVS + NSP, oT2 = oT1 + TILES * TILE;
  static constexpr int oD2X = oT2 + TILES * TILE2;                        // [tile][half][MH][64 lanes] f32x4: the dZ2 half of a wave in A-operand layout, for its partner
  static constexpr int oZP = oD2X + TILES * 2 * MH * 256;                 // [tile][half][g 4][sample 16][ZW]
  static constexpr int oPART = ((oZP + TILES * 2 * 64 * ZW + 3) / 4) * 4; // [tile][PART]
  static constexpr int oXS = oPART + TILES * PART;
  static constexpr int oSC = oXS + NXB * TILES * 16 * XP;
  static constexpr int oRED = oSC + NXB * TILES * 16 * SCW;                     // [0,8): per-wave sum of squares; [8,15): reduced stat sums; [16]: abort flag
  static constexpr int oLAG = oRED + 32;                                        // lagrange_ppo_loss: [buffer][cost 128 | episode_end 128] of the WHOLE minibatch
  static constexpr int oLGS = oLAG + (LAG ? 2 * 256 : 0);                       // [wave][8]: the controller's state, one copy per wave (every wave advances its own, identically)
  static constexpr int TOTAL = oLGS + (LAG ? 8 * NW : 0);
  static constexpr int NSI = (NS + NT - 1) / NT;
  static constexpr int XSLOT = ((W2N + NSI * NT + 16 + 3) / 4) * 4;       // floats per exchange slot
  static_assert(TOTAL <= 40960, "LDS budget (160 KB) exceeded");
  static_assert(XSLOT <= 8192, "exchange slot");
};

// PX: the replica-group form (comm.hip "peer"): after the workgroups have formed the local total, the N replicas SUM-all-reduce it through peer-mapped slots inside the same step,
// between the pullback (training.jl:18) and Flux.update! (:21) -- the protocol of train_mfma_kernel.h with the peers shared among four workgroups instead of two.
// ACT / ACT2: activations of the first / second hidden layer (the reference's critic of that example has no activation on its second layer).
// LAG: lagrange_ppo_loss (ppo.jl:70-131) -- the PID penalty controller advanced once per minibatch inside the kernel (every thread, from the staged :cost / :episode_end
// columns of the whole minibatch) and the cost-advantage term of the loss; helper-wave form.
template <int IN, int OUT, int KIND, int ACT, int NWG, bool HELP = false, bool TIMING = false, bool PX = false, int H2 = 64, int ACT2 = ACT, bool LAG = false, bool PXK = false>
__global__ __launch_bounds__(64 * (HELP ? 2 : 1) * (16 / NWG)) void k_train_fs(TrainArgs a) {
  static_assert(NWG == 2 || NWG == 4, "two workgroups of eight waves, or four of four (+ four helper waves)");
  static_assert(!HELP || NWG == 4, "helper waves: the four-workgroup form");
  static_assert(!LAG || (HELP && !PX && KIND != MFK_VALUE), "lagrange_ppo_loss: helper-wave form, policy heads, one replica");
  static_assert(!PXK || PX, "PXK: the periodic form of the replica group");
  static_assert(!PX || FsLayout<IN, OUT, NWG, HELP, H2, LAG>::W2N + FsLayout<IN, OUT, NWG, HELP, H2, LAG>::NSI * FsLayout<IN, OUT, NWG, HELP, H2, LAG>::NT + 8 <= CRUX_PX_SEC, "a payload section must fit CRUX_PX_SEC");
  using Lt = FsLayout<IN, OUT, NWG, HELP, H2, LAG>;
  constexpr int NW = Lt::NW, NWC = Lt::NWC, TILES = Lt::TILES, NT = Lt::NT, MH = Lt::MH, HH = Lt::HH, W2N = Lt::W2N;
  constexpr int WT = (Lt::NT2 * 4) / NW;             // 16x16 tiles of W2 (H2/16 x 4 of them) owned by a wave
  static_assert(WT >= 1 && WT * NW == Lt::NT2 * 4, "W2 tiles must divide over the waves");
  constexpr int KS0 = Lt::KS0, IP = Lt::IP, JT = Lt::JT, XP = Lt::XP, NS = Lt::NS, NSI = Lt::NSI, XSLOT = Lt::XSLOT;
  constexpr int NACT = (OUT > 4 ? OUT : 4);
  constexpr int STAT_HI = LAG ? 64 * Lt::NW : 64 * Lt::NW - 1;
  if ((int)(blockIdx.x & 7) != a.xcd) return;        // the NWG workgroups of the learner: blocks x, x + 8, x + 16, x + 24 -> one XCD (consecutive workgroups go round-robin over the 8 XCDs)
  const int p = (int)(blockIdx.x >> 3);              // workgroup 0 .. NWG-1 of this learner
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, g = lane >> 4;
  // Helper waves (HELP): waves NWC .. 2 NWC - 1 have no tile. They own half of the W2 tiles (dW2, its exchange, Adam: the per-lane share of a wave halves and two waves
  // per SIMD interleave their instruction streams) and they prefetch and stage the NEXT minibatch while the compute waves run the forward pass.
  const bool cw = !HELP || w < NWC;                  // compute wave
  const int wt = HELP ? (w & (NWC - 1)) : w;         // (tile, half) role: of the tile work for a compute wave, of the staging work for a helper
  const int t = wt >> 1, h = wt & 1;                 // sample tile of the workgroup, feature half
  const bool sw = HELP ? !cw : true;                 // this wave prefetches / stages minibatch rows
  float* part = sm + Lt::oPART + t * Lt::PART;
  float* xs = sm + Lt::oXS + t * 16 * XP;            // (+ buffer offset when double-buffered)
  float* sc = sm + Lt::oSC + t * 16 * Lt::SCW;
  constexpr int XSB = TILES * 16 * XP, SCB = TILES * 16 * Lt::SCW;      // one staging buffer
  int xcur = 0;                                      // buffer the current minibatch sits in
  float* T1 = sm + Lt::oT1 + t * Lt::TILE;
  float* T2 = sm + Lt::oT2 + t * Lt::TILE2;
  const int n_extra = (KIND == MFK_GAUSSIAN) ? OUT : 0;
  unsigned long long tacc[16]; unsigned long long tlast = 0;
  if (TIMING) { for (int k = 0; k < 16; ++k) tacc[k] = 0; tlast = __builtin_amdgcn_s_memtime(); }
#define FS_T(ph) do { if (TIMING) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); tacc[ph] += tn - tlast; tlast = tn; } } while (0)
  const int t_wr = (4 * g) * 16 + 4 * ((c >> 2) ^ fs_tx(g)) + (c & 3);   // tile element (feature 4g [+16m+r], sample c)
  const int t_rd = c * 16 + 4 * (g ^ fs_tx(c >> 2));                    // tile b128 (feature c [+16m], samples 4g..4g+3)
  // dW2 / W2 ownership: tile (mp, m) = rows [16mp, 16mp+16) x columns [16m, 16m+16), numbered 4 mp + m; wave w owns the WT tiles from number w WT on (same mp)
  const int mp0 = (w * WT) >> 2, m0 = (w * WT) & 3;
  (void)sw; (void)xcur;

  auto s_master = [&](int s) -> int {
    if (s < Lt::sB1) { const int o = s & 63, i = s >> 6; return Lt::oW1R + o * Lt::W1LD + i; }
    if (s < Lt::sB2) return Lt::oB1 + (s - Lt::sB1);
    if (s < Lt::sW3) return Lt::oB2 + (s - Lt::sB2);
    if (s < Lt::sB3) { const int q = s - Lt::sW3; const int o = q % OUT, i = q / OUT; return Lt::oW3R + o * H2 + i; }
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
    if (s < Lt::sB1) { const int o = s & 63, i = s >> 6; return Lt::pW1 + i * FS_LD + o; }
    if (s < Lt::sB2) return Lt::pB1 + (s - Lt::sB1);
    if (s < Lt::sW3) return Lt::pB2 + (s - Lt::sB2);
    if (s < Lt::sB3) { const int q = s - Lt::sW3; const int o = q % OUT, i = q / OUT; return Lt::pW3 + o * H2 + i; }
    if (s < Lt::sEX) return Lt::pB3 + (s - Lt::sB3);
    return Lt::pEX + (s - Lt::sEX);
  };
  const int ns_valid = Lt::sEX + n_extra;
  int so_part[NSI], so_master[NSI]; bool so_ok[NSI], so_ex[NSI];
#pragma unroll
  for (int k = 0; k < NSI; ++k) { const int s = tid + NT * k; so_ok[k] = s < ns_valid; so_ex[k] = s >= Lt::sEX;
    so_part[k] = so_ok[k] ? s_part(s) : 0; so_master[k] = so_ok[k] ? s_master(s) : 0; }

  // ---- load parameters and Adam state --------------------------------------------------------------------------
  for (int q = tid; q < W2N; q += NT) { const int o = q % H2, i = q / H2; const float v = a.p[Lt::cW2 + q];
    sm[Lt::oW2R + o * FS_LD + i] = v; sm[Lt::oW2C + i * FS_LD + o] = v; }
  for (int q = tid; q < MF_HID * Lt::W1LD; q += NT) sm[Lt::oW1R + q] = 0.f;
  if (tid < 16) { sm[Lt::oB3 + tid] = 0.f; sm[Lt::oEX + tid] = 0.f; }
  for (int q = tid; q < TILES * Lt::PART; q += NT) sm[Lt::oPART + q] = 0.f;
  uint32_t my_xcc = 0;
  if (tid == 0) { asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my_xcc)); my_xcc &= 0xf;
    __hip_atomic_store(a.xctr + 8 + p, my_xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // checked against the peers' at the first exchange
  __syncthreads();
  for (int s = tid; s < NS; s += NT) { const bool in = s < ns_valid; const int pc = s_canon(s);
    if (in) sm[s_master(s)] = a.p[pc];
    sm[Lt::oMS + s] = in ? a.m[pc] : 0.f; sm[Lt::oVS + s] = in ? a.v[pc] : 0.f; }
  for (int q = tid; q < Lt::NXB * TILES * 16 * XP; q += NT) sm[Lt::oXS + q] = 0.f;
  // owned W2 tiles, D layout: reg r of tile mm <-> W2[o = 16 mp0 + 4g + r][i = 16 (m0+mm) + c]
  f32x4 tW2[WT], mW2[WT], vW2[WT];
#pragma unroll
  for (int mm = 0; mm < WT; ++mm)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int pc = Lt::cW2 + (16 * mp0 + 4 * g + r) + H2 * (16 * (m0 + mm) + c);
      tW2[mm][r] = a.p[pc]; mW2[mm][r] = a.m[pc]; vW2[mm][r] = a.v[pc]; }
  double bp1 = a.bp[0], bp2 = a.bp[1];
  const float lo = 1.f - a.eps_clip, hi = 1.f + a.eps_clip;
  const bool a2c = a.loss == CRUX_LOSS_A2C;
  AdamK ak; ak.b1 = (float)a.b1; ak.b2 = (float)a.b2; ak.omb1 = (float)(1.0 - a.b1); ak.omb2 = (float)(1.0 - a.b2); ak.eps = (float)a.eps; ak.eta = (float)a.eta;

  int32_t* order_cur = a.order_a; int32_t* order_nxt = a.order_b;
  long long total_batches = 0; int epochs_run = 0, err = 0, why_failed = 0; bool stop = false;
  bool staged = false;
  long long xstep = 0, pxc = 0;      // pxc: replica-group exchanges of this launch (one per step, or three per k-th step in the periodic form)
  // the reported minibatch's info (training.jl:22-23) is kept by thread 0 -- its only reader (epoch_infos) -- in free words of the reduction area, not in registers of every
  // thread that would stay live across the whole launch (k_train_fs2 has the measurement: ~30 VGPRs and every spill of the wide heads); the KL stays a register: the loop exits read it
  constexpr int iLOSS = Lt::oRED + 18, iGN = Lt::oRED + 19, iENT = Lt::oRED + 20, iCLIP = Lt::oRED + 21, iADV = Lt::oRED + 22, iRET = Lt::oRED + 23,
                iPEN = Lt::oRED + 24, iCUR = Lt::oRED + 25, iCLOSS = Lt::oRED + 26, iPLOSS = Lt::oRED + 27;
  if (threadIdx.x == 0) { for (int k = 18; k < 28; ++k) sm[Lt::oRED + k] = 0.f; }
  float inf_kl = 0.f;
  float pen = 0.f;                                     // lagrange_ppo_loss: the penalty of the current minibatch; the controller's state sits in LDS, one copy per wave
  float* lgs = sm + Lt::oLGS + 8 * w;                  // [I, smooth_delta, smooth_Jc, Jc_prev, deriv_term, penalty, cur_cost]
  if constexpr (LAG) { if (lane == 0) { lgs[0] = a.lag->I; lgs[1] = a.lag->smooth_delta; lgs[2] = a.lag->smooth_Jc; lgs[3] = a.lag->Jc_prev; lgs[4] = a.lag->deriv_term; lgs[5] = a.lag->penalty; lgs[6] = a.lag->cur_cost; } }
  // replica group: exchanges done on this learner stream before this launch -- slot parity and flag values continue across launches
  float* const px_mine = PX ? a.px_tab[a.px_rank] : nullptr;
  const unsigned long long px0 = PX ? *(const unsigned long long*)(px_mine + CRUX_PX_COUNT) : 0ull;
  if (PX && tid == 0) px_launch_begin(px_mine, p);      // the launch's wait budget starts from zero (peer_wait.h, bound 2)
  const float px_inv = PX ? 1.0f / (float)a.px_n : 1.0f;
  const int n_epochs = a.epochs;
  if (!a.ord_all) { for (int64_t j = tid; j < a.len; j += NT) order_cur[j] = (int32_t)j; }
  if (!a.ord_all && a.pre_epochs > 0) {
    __syncthreads();
    for (int pe = 0; pe < a.pre_epochs; ++pe) {
      if (a.pre_perms) { for (int64_t j = tid; j < a.len; j += NT) order_nxt[j] = order_cur[a.pre_perms[(int64_t)pe * a.len + j]]; }
      else { const crux_perm pp = crux_perm_make(a.pre_seed, a.pre_counter + (uint64_t)pe, 0, (uint32_t)a.len);
        for (int64_t j = tid; j < a.len; j += NT) order_nxt[j] = order_cur[crux_perm_at(&pp, (uint32_t)j)]; }
      __syncthreads();
      int32_t* tq = order_cur; order_cur = order_nxt; order_nxt = tq;
    }
  }
  __syncthreads();
  const int64_t total_rows = a.len;

  // ---- minibatch prefetch (HBM/L2 -> registers) and staging (registers -> the tile's LDS rows) ----------------
  // The pair shares the work: wave h = 0 fetches and stages the observation rows (four lanes per sample), wave h = 1 the scalars (logprob, advantage, return, action).
  constexpr int NXL = (IN + 3) / 4;
  float px[NXL]; float p_lp = 0.f, p_adv = 0.f, p_ret = 0.f; float p_act[NACT]; int p_valid = 0; uint8_t p_abyte[OUT];
#pragma unroll
  for (int k = 0; k < OUT; ++k) p_abyte[k] = 0;
#pragma unroll
  for (int k = 0; k < NACT; ++k) p_act[k] = 0.f;
#pragma unroll
  for (int e = 0; e < NXL; ++e) px[e] = 0.f;
  int n_row = 0, n_valid = 0;
  const int lt = tid - 64 * NWC;                       // LAG: helper thread lt < 128 carries row lt of the whole minibatch (its :cost and :episode_end)
  int n_row2 = -1; float p_cost2 = 0.f, p_ee2 = 0.f, p_cadv = 0.f;
  auto fetch_index = [&](const int32_t* ord, int64_t st, int nb) {
    const int sidx = 8 * NWC * p + 16 * t + c;
    n_valid = sidx < nb ? 1 : 0;
    n_row = n_valid ? CRUX_GLOBAL_PTR(int32_t, ord)[st + sidx] : 0;
    if constexpr (LAG) n_row2 = (lt >= 0 && lt < nb) ? CRUX_GLOBAL_PTR(int32_t, ord)[st + lt] : -1;
  };
  auto fetch_data = [&]() {
    const int rowlo = n_row; p_valid = n_valid; const int64_t row = rowlo;
    if constexpr (LAG) { p_cost2 = n_row2 >= 0 ? CRUX_GLOBAL_PTR(float, a.COST)[n_row2] : 0.f; p_ee2 = (n_row2 >= 0 && CRUX_GLOBAL_PTR(uint8_t, a.EE)[n_row2]) ? 1.f : 0.f; }
    if (h == 0) {
      const int rs = __shfl(rowlo, lane >> 2, 64), vs = __shfl(p_valid, lane >> 2, 64);      // lanes 0..15 hold the rows of samples 0..15
      const float* xrow = a.PACK ? CRUX_GLOBAL_PTR(float, a.PACK) + (int64_t)rs * a.pack_stride + (lane & 3) * NXL : CRUX_GLOBAL_PTR(float, a.S) + (int64_t)rs * IN + (lane & 3) * NXL;
#pragma unroll
      for (int e = 0; e < NXL; ++e) px[e] = ((lane & 3) * NXL + e < IN && vs) ? xrow[e] : 0.f;
    } else {
      p_lp = 0.f; p_adv = 0.f; p_ret = 0.f; p_cadv = 0.f;
#pragma unroll
      for (int k = 0; k < NACT; ++k) p_act[k] = 0.f;
      if (a.PACK && !LAG) {      // the sample's scalars and action from its packed line (the observation half of the same line goes to wave h = 0)
        if (lane < 16 && p_valid) { const float* q = CRUX_GLOBAL_PTR(float, a.PACK) + row * a.pack_stride;
          if (KIND != MFK_VALUE) { p_lp = q[a.pack_lp]; p_adv = q[a.pack_lp + 1]; }
          p_ret = q[a.pack_lp + 2];
          if (KIND == MFK_CATEGORICAL) { const int ai = (int)q[a.pack_act];
#pragma unroll
            for (int k = 0; k < OUT; ++k) p_abyte[k] = k == ai ? 1 : 0; }
          if (KIND == MFK_GAUSSIAN) {
#pragma unroll
            for (int k = 0; k < OUT; ++k) p_act[k] = q[a.pack_act + k]; } }
      } else
      if (lane < 16 && p_valid) {
        if constexpr (LAG) p_cadv = CRUX_GLOBAL_PTR(float, a.CADV)[row];
        if (KIND != MFK_VALUE) { p_lp = CRUX_GLOBAL_PTR(float, a.LP)[row]; p_adv = CRUX_GLOBAL_PTR(float, a.ADV)[row]; }
        p_ret = a.RET ? CRUX_GLOBAL_PTR(float, a.RET)[row] : 0.f;
        if (KIND == MFK_CATEGORICAL) { const auto* av = CRUX_GLOBAL_PTR(uint8_t, a.A) + row * OUT;
#pragma unroll
          for (int k = 0; k < OUT; ++k) p_abyte[k] = av[k]; }
        if (KIND == MFK_GAUSSIAN) { const auto* av = CRUX_GLOBAL_PTR(float, a.A) + row * OUT;
#pragma unroll
          for (int k = 0; k < OUT; ++k) p_act[k] = av[k]; }
      }
    }
  };
  auto stage = [&](int buf) {
    float* xs_ = xs + buf * XSB; float* sc_ = sc + buf * SCB;
    if constexpr (LAG) { if (lt >= 0 && lt < 128) { sm[Lt::oLAG + 256 * buf + lt] = p_cost2; sm[Lt::oLAG + 256 * buf + 128 + lt] = p_ee2; } }
    if (h == 0) {
#pragma unroll
      for (int e = 0; e < NXL; ++e) { const int f = (lane & 3) * NXL + e; if (f < IN) xs_[(lane >> 2) * XP + f] = px[e]; }
    } else {
      if (KIND == MFK_CATEGORICAL) { int ai = 0;
#pragma unroll
        for (int k = 0; k < OUT; ++k) ai = p_abyte[k] ? k : ai;
        p_act[0] = (float)ai; }
      if (lane < 16) { float* q = sc_ + lane * Lt::SCW; q[0] = (float)p_valid; q[1] = p_lp; q[2] = p_adv; q[3] = p_ret;
        if constexpr (LAG) q[Lt::SCW - 1] = p_cadv;
        if (KIND == MFK_GAUSSIAN) {       // SquashedGaussianPolicy: the stored action is un-tanh'd once here and the tanh correction of logpdf rides in the spare slot
          static_assert(KIND != MFK_GAUSSIAN || ((4 + NACT) % 2 == 0), "the staging row needs its spare slot");
          float corr = 0.f;
          if (a.squash > 0.f) {
#pragma unroll
            for (int k = 0; k < OUT; ++k) { const float u = p_valid ? sq_untanh(p_act[k], a.squash) : 0.f; corr += p_valid ? sq_corr(u) : 0.f; p_act[k] = u; } }
          q[4 + NACT] = corr; }
#pragma unroll
        for (int k = 0; k < NACT; ++k) q[4 + k] = p_act[k]; }
    }
  };

  if ((!PX || PXK) && tid == 0) {      // this workgroup's SUSPECT words (B_or elision below), both slot parities: a launch that ended on a suspect step must not slow the next one down
    a.xbuf[(size_t)(0 * NWG + p) * XSLOT + W2N + NSI * NT + 12] = 0.f; a.xbuf[(size_t)(1 * NWG + p) * XSLOT + W2N + NSI * NT + 12] = 0.f; }
  for (int ep = 0; ep < n_epochs && !stop && !err; ++ep) {
    if (a.spec_abort) {      // a speculative run (train_args.h): thread 0 of every workgroup ORs what it reads from the host's word into an L2 latch, all wait until all have (one
                             // arrival counter, NWG per epoch), then everyone reads the latch: the same decision in every workgroup, whenever the host's store lands
      if (tid == 0) {
        const unsigned r = __hip_atomic_load(a.spec_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        (void)__hip_atomic_fetch_or(a.xctr + 16, r ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        (void)__hip_atomic_fetch_add(a.xctr + 17, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned want = (unsigned)NWG * (unsigned)(ep + 1); unsigned spins = 0;
        bool late = false;
        while (__hip_atomic_load(a.xctr + 17, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) { __builtin_amdgcn_s_sleep(2); if (++spins > (1u << 24)) { late = true; break; } }
        // a workgroup that never arrived: the consensus cannot be formed. Leave with an error of its own (never "continue on whatever the latch says now" -- the others may read
        // something else) and raise the latch and the learner's abort word, so that every workgroup that still comes by leaves too instead of waiting at the next gradient exchange.
        if (late) { (void)__hip_atomic_fetch_or(a.xctr + 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(a.xctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        sm[Lt::oRED + 17] = late ? 2.f : (__hip_atomic_load(a.xctr + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1.f : 0.f); }
      __syncthreads();
      if (sm[Lt::oRED + 17] == 2.f) { err = CRUX_EHIP; why_failed = 4; break; }
      if (sm[Lt::oRED + 17] != 0.f) { err = CRUX_TRAIN_ABORTED; break; }      // (the critic's result is valid only when the latch was never set: the host discards or restores it otherwise)
    }
    if (a.ord_all) order_cur = const_cast<int32_t*>(a.ord_all) + (size_t)ep * (size_t)a.len;   // shuffle orders composed ahead of time by k_compose_order
    else {   // shuffle!(D) as an index composition (experience_buffer.jl:118-124)
      if (a.perms) { for (int64_t j = tid; j < a.len; j += NT) order_nxt[j] = order_cur[a.perms[(int64_t)ep * a.len + j]]; }
      else { const crux_perm pp = crux_perm_make(a.shuffle_seed, a.shuffle_counter + (uint64_t)ep, 0, (uint32_t)a.len);
        for (int64_t j = tid; j < a.len; j += NT) order_nxt[j] = order_cur[crux_perm_at(&pp, (uint32_t)j)]; }
      __syncthreads();
      int32_t* tq = order_cur; order_cur = order_nxt; order_nxt = tq;
    }
    staged = false;
    if (sw) { const int nb0 = (int)(total_rows < a.bs ? total_rows : a.bs); fetch_index(order_cur, 0, nb0); fetch_data();
      const int64_t st1 = a.bs; const int nb1 = st1 < total_rows ? (int)((total_rows - st1) < a.bs ? (total_rows - st1) : a.bs) : 0; fetch_index(order_cur, st1 < total_rows ? st1 : 0, nb1); }
    for (int64_t st = 0; st < total_rows; st += a.bs) {
      const int nb = (int)((total_rows - st) < a.bs ? (total_rows - st) : a.bs);
      const float invB = 1.0f / (float)nb;
      ak.c1 = __builtin_amdgcn_rcpf((float)(1.0 - bp1)); ak.c2 = __builtin_amdgcn_rcpf((float)(1.0 - bp2));
      FS_T(0);
      if (!staged) { if (sw) stage(HELP ? xcur : 0); __syncthreads(); }     // the first minibatch of an epoch; every other one was staged during the previous step (barriers follow it there)
      staged = false;
      if constexpr (LAG) {   // the penalty update inside the loss (ppo.jl:80-116), once per evaluation: sums of the minibatch's :cost and :episode_end, then the controller
        const float* lc = sm + Lt::oLAG + 256 * xcur;
        double sc_ = (double)lc[lane] + (double)lc[lane + 64], ne_ = (double)lc[128 + lane] + (double)lc[128 + lane + 64];      // Float32 terms: any summation order gives the same Float64 sum
#pragma unroll
        for (int o_ = 32; o_ >= 1; o_ >>= 1) { sc_ += __shfl_xor(sc_, o_, 64); ne_ += __shfl_xor(ne_, o_, 64); }
        const crux_lagrange* L = a.lag;              // the keywords: uniform (scalar) loads
        const float Jc = (float)sc_ / (float)ne_;                                      // :84-88
        const float dl = Jc - L->target_cost;                                         // :91
        float I_ = lgs[0], sd_ = lgs[1], sj_ = lgs[2]; const float jp_ = lgs[3];
        { const float x = I_ + L->Ki * dl; I_ = x > L->Ki_max ? L->Ki_max : (x < 0.f ? 0.f : x); }                   // :94 clamp(I + Ki*Delta, 0, Ki_max)
        sd_ = (float)(L->ema_alpha * (double)sd_ + (1.0 - L->ema_alpha) * (double)dl);                             // :98 (Float64 arithmetic, Float32 store)
        sj_ = (float)(L->ema_alpha * (double)sj_ + (1.0 - L->ema_alpha) * (double)Jc);                             // :99
        float dt_; { const float x = sj_ - jp_; dt_ = (x != x) ? x : (x > 0.f ? x : 0.f); }                          // :102 max(0, .) keeps NaN
        { const float x = (L->Kp * sd_ + I_) + L->Kd * dt_; pen = x > L->penalty_max ? L->penalty_max : (x < 0.f ? 0.f : x); }   // :108
        if (lane == 0) { lgs[0] = I_; lgs[1] = sd_; lgs[2] = sj_; lgs[3] = sj_ /* Jc_prev = smooth_Jc, :105 */; lgs[4] = dt_; lgs[5] = pen; lgs[6] = Jc; }
      }
      if (sw) {
        if (st + a.bs < total_rows) fetch_data();
        const int64_t st2 = st + 2 * (int64_t)a.bs; const int nb2 = st2 < total_rows ? (int)((total_rows - st2) < a.bs ? (total_rows - st2) : a.bs) : 0;
        fetch_index(order_cur, st2 < total_rows ? st2 : 0, nb2); }
      const float* xs_c = xs + (HELP ? xcur * XSB : 0); const float* sc_c = sc + (HELP ? xcur * SCB : 0);      // the current minibatch's rows

      FS_T(1);
      // ======================= forward, C orientation: D[feature 16m+4g+r][sample c] =======================
      f32x4 h1[4];                                   // the WHOLE first layer in both waves of the pair
      f32x4 h2[MH];                                  // second layer: output features [HH h, HH h + HH)
      constexpr bool W3_REG = OUT <= 2;
      f32x4 w3[OUT <= 2 ? OUT : 1][MH];
      float zp[OUT];
      if (cw) {
      float xB[KS0];
#pragma unroll
      for (int ks = 0; ks < KS0; ++ks) xB[ks] = xs_c[c * XP + 4 * ks + g];
#pragma unroll
      for (int m = 0; m < 4; ++m) { f32x4 acc = *(const f32x4*)&sm[Lt::oB1 + 16 * m + 4 * g];
#pragma unroll
        for (int ks = 0; ks < KS0; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(sm[Lt::oW1R + (16 * m + c) * Lt::W1LD + 4 * ks + g], xB[ks], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = actf<ACT>(acc[r]);
        h1[m] = acc; }
#pragma unroll
      for (int mm = 0; mm < 2; ++mm)                 // the tile's H1 rows of this wave's half (the partner writes the others): B operand of dW2, relu' mask of dZ1
#pragma unroll
        for (int r = 0; r < 4; ++r) T1[t_wr + (16 * (2 * h + mm) + r) * 16] = (h ? (mm ? h1[3][r] : h1[2][r]) : (mm ? h1[1][r] : h1[0][r]));
      FS_T(2);
      { f32x4 acc[MH];
#pragma unroll
        for (int mm = 0; mm < MH; ++mm) acc[mm] = *(const f32x4*)&sm[Lt::oB2 + HH * h + 16 * mm + 4 * g];
#pragma unroll
        for (int m = 0; m < 4; ++m) { f32x4 wv[MH];
#pragma unroll
          for (int mm = 0; mm < MH; ++mm) wv[mm] = *(const f32x4*)&sm[Lt::oW2R + (HH * h + 16 * mm + c) * FS_LD + 16 * m + 4 * g];
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mm = 0; mm < MH; ++mm) acc[mm] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[mm][r], h1[m][r], acc[mm], 0, 0, 0); }      // MH independent accumulator chains
#pragma unroll
        for (int mm = 0; mm < MH; ++mm) {
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[mm][r] = actf<ACT2>(acc[mm][r]);
          h2[mm] = acc[mm]; } }

      FS_T(3);
      // ======================= layer 3 (VALU): partial logits over this wave's 32 features, exchanged inside the pair =======================
      if (W3_REG) {
#pragma unroll
        for (int o = 0; o < OUT; ++o)
#pragma unroll
          for (int mm = 0; mm < MH; ++mm) w3[o][mm] = *(const f32x4*)&sm[Lt::oW3R + o * H2 + HH * h + 16 * mm + 4 * g]; }
#pragma unroll
      for (int o = 0; o < OUT; ++o) { float acc = 0.f;
#pragma unroll
        for (int mm = 0; mm < MH; ++mm) { const f32x4 wv = W3_REG ? w3[W3_REG ? o : 0][mm] : *(const f32x4*)&sm[Lt::oW3R + o * H2 + HH * h + 16 * mm + 4 * g];
#pragma unroll
          for (int r = 0; r < 4; ++r) acc = fmaf(wv[r], h2[mm][r], acc); }
        zp[o] = g4_sum(acc); }
      { float* zq = sm + Lt::oZP + ((t * 2 + h) * 64 + lane) * Lt::ZW;      // every lane its own slot (the four g rows hold the same value): no cross-lane read needed
#pragma unroll
        for (int o = 0; o < OUT; ++o) zq[o] = zp[o]; }
      }      // compute waves
      __syncthreads();   // ---- B_z: partial logits of both halves are visible
      if (HELP && !cw && st + a.bs < total_rows) stage(xcur ^ 1);      // helpers: the NEXT minibatch (rows requested at the top of this step) goes into the other staging buffer
      float dz[OUT], dex[OUT];
      float s_lossp = 0.f, s_H = 0.f, s_kl = 0.f, s_adv = 0.f, s_ret = 0.f, s_clip = 0.f, s_sq = 0.f, s_cost = 0.f;
      if (cw) {
      float z[OUT];
      { const float* zo = sm + Lt::oZP + ((t * 2 + (1 - h)) * 64 + lane) * Lt::ZW;
#pragma unroll
        for (int o = 0; o < OUT; ++o) { const float other = zo[o]; z[o] = (h ? other + zp[o] : zp[o] + other) + sm[Lt::oB3 + o]; } }      // (half 0 + half 1) + b: the same bits in both waves
      // ======================= loss head (identical in both waves of the pair) =======================
      {
        const float* q = sc_c + c * Lt::SCW;
        const bool valid = q[0] != 0.f; const float oldlp = q[1], A = q[2], R = q[3];
        const float cnt = (valid && g == 0) ? 1.f : 0.f;    // every sample is replicated in the 4 g-groups (and in both waves: only wave h = 0 reports statistics)
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
          if constexpr (LAG) { const float Ac = q[Lt::SCW - 1]; const float uc = r * Ac, clc = rc * Ac; s_cost = cnt * (uc >= clc ? uc : clc); gcr = (uc >= clc ? Ac : 0.f) * r; }
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
            s2[k] = __expf(-2.f * (sq ? sq_clampls(ls) : ls)); dd[k] = q[4 + k] - z[k];
            inr[k] = (sq && !(ls >= -5.f && ls <= 2.f)) ? 0.f : 1.f;
            newlp += (-(dd[k] * dd[k]) * (0.5f * s2[k]) - 0.9189385332046727f - ls); }
          if (a.squash > 0.f) newlp -= q[4 + NACT];
          const float r = __expf(newlp - oldlp); const float u = r * A, rc = fminf(fmaxf(r, lo), hi), cl = rc * A; const float gsel = (u <= cl) ? A : 0.f;
          const float coef = a2c ? A : gsel * r, lterm = a2c ? newlp * A : fminf(u, cl), clipv = (!a2c && (r > hi || r < lo)) ? 1.f : 0.f;
          float cf = -a.lambda_p * coef;
          if constexpr (LAG) { const float Ac = q[Lt::SCW - 1]; const float uc = r * Ac, clc = rc * Ac; s_cost = cnt * (uc >= clc ? uc : clc);
            cf = (cf + pen * ((uc >= clc ? Ac : 0.f) * r)) / (1.f + pen); }
#pragma unroll
          for (int k = 0; k < OUT; ++k) { dz[k] = valid ? invB * (cf * (dd[k] * s2[k])) : 0.f;
            dex[k] = valid ? invB * (cf * (((dd[k] * dd[k]) * s2[k]) * inr[k] - 1.f)) : 0.f; }
          s_lossp = cnt * lterm; s_kl = cnt * (oldlp - newlp); s_adv = cnt * A; s_ret = cnt * R; s_clip = cnt * clipv;
        }
      }

      FS_T(4);
      // ======================= backward, own samples, own feature half =======================
      // dW3 rows of this half: sum over the 16 samples of dz[o] * h2[feature]; PER outputs share one 16-value reduce-scatter (4 MH features each per lane row)
      constexpr int FPL = 4 * MH, PER = 16 / FPL;
#pragma unroll
      for (int o2 = 0; o2 < (OUT + PER - 1) / PER; ++o2) { float pv[16];
#pragma unroll
        for (int oo = 0; oo < PER; ++oo)
#pragma unroll
          for (int mm = 0; mm < MH; ++mm)
#pragma unroll
            for (int r = 0; r < 4; ++r) pv[FPL * oo + 4 * mm + r] = (PER * o2 + oo < OUT) ? dz[(PER * o2 + oo < OUT) ? PER * o2 + oo : 0] * h2[mm][r] : 0.f;
        const float sred = row16_reduce_scatter(pv, c);      // lane c: output PER o2 + c / FPL, feature HH h + 16 ((c >> 2) & (MH - 1)) + 4g + (c & 3)
        const int oo = c / FPL;
        if (PER * o2 + oo < OUT) part[Lt::pW3 + (PER * o2 + oo) * H2 + HH * h + 16 * ((c >> 2) & (MH - 1)) + 4 * g + (c & 3)] = sred; }
      { f32x4 d2[MH];
#pragma unroll
        for (int mm = 0; mm < MH; ++mm) d2[mm] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < OUT; ++o)
#pragma unroll
          for (int mm = 0; mm < MH; ++mm) { const f32x4 wv = W3_REG ? w3[W3_REG ? o : 0][mm] : *(const f32x4*)&sm[Lt::oW3R + o * H2 + HH * h + 16 * mm + 4 * g];
#pragma unroll
            for (int r = 0; r < 4; ++r) d2[mm][r] = fmaf(wv[r], dz[o], d2[mm][r]); }
#pragma unroll
        for (int mm = 0; mm < MH; ++mm)
#pragma unroll
          for (int r = 0; r < 4; ++r) h2[mm][r] = actg<ACT2>(h2[mm][r], d2[mm][r]); }       // h2 now holds dZ2 of this half
      if (h == 0) {      // statistics and the head's own gradient sums (db3, dlogSigma): once per tile
        constexpr int NVB = 7 + OUT + (KIND == MFK_GAUSSIAN ? OUT : 0), NV = NVB + (LAG ? 1 : 0);
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
          const float tq = row16_reduce_scatter(cv, c);
          if (g == 0) part[Lt::pMISC + 16 * ch + c] = tq; } }
#pragma unroll
      for (int mm = 0; mm < MH; ++mm)
#pragma unroll
        for (int r = 0; r < 4; ++r) T2[t_wr + (16 * (MH * h + mm) + r) * 16] = h2[mm][r];
      { float* dq = sm + Lt::oD2X + ((t * 2 + h) * MH) * 256 + lane * 4;     // the same values in A-operand (register) layout for the partner's dH1
#pragma unroll
        for (int mm = 0; mm < MH; ++mm) *(f32x4*)&dq[256 * mm] = h2[mm]; }
      }      // compute waves
      FS_T(5);
      __syncthreads();   // ---- B_1: T1 / T2 tiles of the workgroup and the dZ2 halves are visible
      FS_T(6);
      // ======================= this wave's dW2 tiles over the workgroup's samples, sent to the exchange slots at once =======================
      f32x4 gW2[WT];
#pragma unroll
      for (int mm = 0; mm < WT; ++mm) gW2[mm] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ws = 0; ws < TILES; ++ws) {
        const float* t2 = sm + Lt::oT2 + ws * Lt::TILE2; const float* t1 = sm + Lt::oT1 + ws * Lt::TILE;
        const f32x4 av = *(const f32x4*)&t2[t_rd + 256 * mp0];          // A[i=c -> o=16mp0+c][k -> sample 4g+r]
#pragma unroll
        for (int mm = 0; mm < WT; ++mm) { const f32x4 bv = *(const f32x4*)&t1[t_rd + 256 * (m0 + mm)];   // B[k -> sample][j=c -> i]
#pragma unroll
          for (int r = 0; r < 4; ++r) gW2[mm] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], bv[r], gW2[mm], 0, 0, 0); }
      }
      float* mine = a.xbuf + (size_t)(((int)(xstep & 1) * NWG + p)) * XSLOT;
#pragma unroll
      for (int mm = 0; mm < WT; ++mm) *(f32x4*)&mine[tid * (4 * WT) + 4 * mm] = gW2[mm];      // acknowledged long before the s_waitcnt below
      // B_or elision (round 4): the norm barrier below exists for three things -- the "gradient is NaN" consensus, the KL statistic for early stopping, the per-wave sums of
      // squares of a REPORTED step. A total can only be NaN when a workgroup's partial is NaN or huge (four partials of magnitude <= 1e30 cannot overflow): a thread that sees
      // such a partial raises its workgroup's SUSPECT word in the exchange slot, every thread reads the four words (and the four KL partial sums) beside the partials, and all
      // threads of all workgroups take the barrier only on a suspect, reporting or KL-stopping step -- the same decision everywhere. (Not in replica groups: there the
      // statistics come back from the group exchange in the stat lanes only.)
      constexpr bool ELIDE = !PX || PXK;      // (the periodic form of a replica group runs on LOCAL gradients and statistics between its exchanges, like a single learner)
      constexpr int SUS = W2N + NSI * NT + 12, KLW = W2N + NSI * NT + 2;
      bool odd = false;
      if constexpr (ELIDE) {
#pragma unroll
        for (int mm = 0; mm < WT; ++mm)
#pragma unroll
          for (int r = 0; r < 4; ++r) odd = odd || !(fabsf(gW2[mm][r]) <= 1e30f); }
      FS_T(7);
      // dH1 (R) for the h1 features [32h, 32h + 32) = dZ2 (C regs as A: [i=c -> sample][k -> f' = 16mp+4g+r]) x W2 (B: W2[f'][f = 16m+c] = W2C[f][f']); K = all 64 dZ2 features:
      // the own half from registers, the partner's from its A-layout copy
      if (cw) {
      f32x4 dz1r[2];
      { const float* dq = sm + Lt::oD2X + ((t * 2 + (1 - h)) * MH) * 256 + lane * 4;
        f32x4 av[2 * MH];      // own half (dZ2 features HH h + 16 mm ..), then the partner's (HH (1-h) + 16 mm ..)
#pragma unroll
        for (int mm = 0; mm < MH; ++mm) { av[mm] = h2[mm]; av[MH + mm] = *(const f32x4*)&dq[256 * mm]; }
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 2 * MH; ++q) { const int fo = HH * ((q >= MH) ? 1 - h : h) + 16 * (q % MH);      // dZ2 feature block of this k step
          const f32x4 wv0 = *(const f32x4*)&sm[Lt::oW2C + (32 * h + c) * FS_LD + fo + 4 * g];
          const f32x4 wv1 = *(const f32x4*)&sm[Lt::oW2C + (32 * h + 16 + c) * FS_LD + fo + 4 * g];
#pragma unroll
          for (int r = 0; r < 4; ++r) { acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q][r], wv0[r], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q][r], wv1[r], acc1, 0, 0, 0); } }
        dz1r[0] = acc0; dz1r[1] = acc1; }
      FS_T(8);
      // dZ1 (R)[sample 4g+r][f = 16m+c] = act'(H1 R) .* dH1 (R), m = 2h + mm; bias gradients of both hidden layers for this half
      float gb1[2], gb2[MH];
#pragma unroll
      for (int mm = 0; mm < 2; ++mm) { float sb1 = 0.f;
        const f32x4 h1r = *(const f32x4*)&T1[t_rd + 256 * (2 * h + mm)];
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float d = actg<ACT>(h1r[r], dz1r[mm][r]); dz1r[mm][r] = d; sb1 += d; }
        gb1[mm] = g4_sum(sb1); }
#pragma unroll
      for (int mm = 0; mm < MH; ++mm) { float sb2 = 0.f;
        const f32x4 d2 = *(const f32x4*)&T2[t_rd + 256 * (MH * h + mm)];
#pragma unroll
        for (int r = 0; r < 4; ++r) sb2 += d2[r];
        gb2[mm] = g4_sum(sb2); }
      if (g == 0) {
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) part[Lt::pB1 + 32 * h + 16 * mm + c] = gb1[mm];
#pragma unroll
        for (int mm = 0; mm < MH; ++mm) part[Lt::pB2 + HH * h + 16 * mm + c] = gb2[mm]; }
      // dW1 partial: A = dZ1 (R) [i=c -> o=16m+c][k -> sample 4g+r], B = X (R) [k -> sample][j=c -> input 16jt+c]
#pragma unroll
      for (int jt = 0; jt < JT; ++jt) {
        float xR[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) xR[r] = (16 * jt + c < IP) ? xs_c[(4 * g + r) * XP + 16 * jt + c] : 0.f;
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) { f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dz1r[mm][r], xR[r], acc, 0, 0, 0);
          if (16 * jt + c < Lt::W1ROWS) *(f32x4*)&part[Lt::pW1 + (16 * jt + c) * FS_LD + 32 * h + 16 * mm + 4 * g] = acc; }
      }
      }      // compute waves
      FS_T(9);
      __syncthreads();   // ---- B_2: the small partial gradients of every tile are visible
      // small parameters: add the tiles' partials of this workgroup
      float gs[NSI];
#pragma unroll
      for (int k = 0; k < NSI; ++k) { float gsum = 0.f;
        if (so_ok[k]) { const int po = Lt::oPART + so_part[k];
          gsum = sm[po];
#pragma unroll
          for (int q = 1; q < TILES; ++q) gsum += sm[po + q * Lt::PART]; }
        gs[k] = gsum; }
      float stat_loc = 0.f;
      if (tid >= NT - 8 && tid < STAT_HI) { const int k = tid - (NT - 8);      // stat sums, by 7 lanes of the last wave (8 with the cost term of lagrange_ppo_loss)
        const int ko = (LAG && k == 7) ? Lt::pMISC + 7 + OUT + (KIND == MFK_GAUSSIAN ? OUT : 0) : Lt::pST + k; stat_loc = sm[Lt::oPART + ko];
#pragma unroll
        for (int q = 1; q < TILES; ++q) stat_loc += sm[Lt::oPART + q * Lt::PART + ko]; }
      // ---- replica group: mean over the group of NSEC payload sections (W2-tile registers + the small parameters' registers) and one statistics word, the same bits on every
      // workgroup of every rank. Per-step form: ONE section, the minibatch gradient and its statistics. Periodic form (PXK): THREE sections -- theta, m, v after every k-th Adam
      // step -- in ONE exchange (one release, one flag round trip). Returns false when a replica did not answer (err / why_failed are set).
      // All NWG workgroups hold the same local values. They share the writes (peer i of the N-1 goes to workgroup i mod NWG): the sections go into slot [parity][my rank] of the
      // peer's region, a system-scope release makes them visible, then flag[my rank] there is raised to the exchange number. Every workgroup then waits for the N-1 flags in the
      // OWN region and adds the N contributions in rank order -- the own one from registers, the others from the slots -- so every workgroup of every rank forms the same sum.
      auto px_allreduce_mean = [&](auto nsec_c, f32x4* const (&Wp)[3], float* const (&Sp)[3], float& xT) -> bool {
          constexpr int NSEC = decltype(nsec_c)::value;
          const unsigned long long xg = px0 + (unsigned long long)pxc;       // number of this exchange on this learner stream
          const int par = (int)(xg & 1ull);
          { int pi_ = 0;
            for (int r = 0; r < a.px_n; ++r) {
              if (r == a.px_rank || (pi_++ % NWG) != p) continue;
              float* dst0 = a.px_tab[r] + (size_t)(par * CRUX_PX_MAXR + a.px_rank) * CRUX_PX_SLOT;
#pragma unroll
              for (int sec = 0; sec < NSEC; ++sec) { float* dst = dst0 + sec * CRUX_PX_SEC;
#pragma unroll
                for (int mm = 0; mm < WT; ++mm) *(f32x4*)&dst[tid * (4 * WT) + 4 * mm] = Wp[sec][mm];
#pragma unroll
                for (int k = 0; k < NSI; ++k) dst[W2N + tid + NT * k] = Sp[sec][k]; }
              if (tid >= NT - 8 && tid < STAT_HI) dst0[W2N + NSI * NT + (tid - (NT - 8))] = xT; } }
          // release, the hand-off recipe of the CDNA guides: every wave drains its own slot stores, the workgroup meets, ONE lane issues the system-scope release
          // (buffer_wbl2 sc0 sc1 covers the whole L2, whoever wrote the lines) and drains it before the flags go out -- one L2 write-back per workgroup and exchange
          // instead of one per wave
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            int pi_ = 0;
            for (int r = 0; r < a.px_n; ++r) { if (r == a.px_rank) continue;
              if ((pi_++ % NWG) != p) continue;
              __hip_atomic_store((unsigned long long*)(a.px_tab[r] + CRUX_PX_FLAGS) + 8 * a.px_rank, xg + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
            const long long t0 = wall_clock64();                // 100 MHz; the wait is bounded four ways (peer_wait.h)
            const unsigned gave_up = px_wait_peers(px_mine, a.px_n, a.px_rank, xg + 1ull, t0, a.px_timeout, p, true); const bool ok = gave_up == 0u;
            if (!ok) { px_raise_abort(a.px_tab, a.px_n, gave_up);
              __hip_atomic_store(a.xctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            if (a.px_hist) {      // how long this workgroup waited for the slowest peer's flag (10 ns ticks, log2 bins; workgroups 0 and 1 report)
              const unsigned long long dtk = (unsigned long long)(wall_clock64() - t0) | 1ull;
              if (p < 2) { unsigned* hb = (unsigned*)(px_mine + CRUX_PX_HIST) + 32 * p + (63 - __builtin_clzll(dtk) > 31 ? 31 : 63 - __builtin_clzll(dtk)); *hb = *hb + 1u; } }
            sm[Lt::oRED + 16] = ok ? 0.f : (float)(16u + gave_up);      // 16 + bound (peer_wait.h): the replica group ended this launch
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");                       // system scope, one lane: drops this compute unit's L1 (the slot loads below bypass it anyway: sc0 sc1)
          }
          __syncthreads();
          if (sm[Lt::oRED + 16] != 0.f) { err = CRUX_EHIP; why_failed = (int)sm[Lt::oRED + 16]; return false; }
          // the slots are read PXS ranks at a time (all loads of a batch in flight together: one round trip to the fine-grained region per batch) and added in rank order;
          // the six-output heads sit at the 256-register limit of two waves per SIMD and take one rank at a time (two spilled 14 registers)
          constexpr int PXS = (OUT <= 4 && NSEC == 1) ? 2 : 1;      // (the three-section exchange of the periodic form has all sections of ONE rank in flight together)
          f32x4 oW[NSEC][WT]; float oS[NSEC][NSI]; const float oT = xT;
#pragma unroll
          for (int sec = 0; sec < NSEC; ++sec) {
#pragma unroll
            for (int mm = 0; mm < WT; ++mm) oW[sec][mm] = Wp[sec][mm];
#pragma unroll
            for (int k = 0; k < NSI; ++k) oS[sec][k] = Sp[sec][k]; }
          for (int r0 = 0; r0 < a.px_n; r0 += PXS) {
            f32x4 vW[NSEC][PXS][WT]; float vS[NSEC][PXS][NSI]; float vT[PXS];
#pragma unroll
            for (int q = 0; q < PXS; ++q) {
              const int r = r0 + q;
              if (r < a.px_n && r != a.px_rank) {
                const float* src0 = px_mine + (size_t)(par * CRUX_PX_MAXR + r) * CRUX_PX_SLOT;
                vT[q] = 0.f;
                if (tid >= NT - 8 && tid < STAT_HI) vT[q] = __hip_atomic_load(src0 + W2N + NSI * NT + (tid - (NT - 8)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
                for (int sec = 0; sec < NSEC; ++sec) { const float* src = src0 + sec * CRUX_PX_SEC;
#pragma unroll
                  for (int k = 0; k < NSI; ++k) vS[sec][q][k] = __hip_atomic_load(src + W2N + tid + NT * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
                  for (int mm = 0; mm < WT; ++mm)
                    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=&v"(vW[sec][q][mm]) : "v"(src + tid * (4 * WT) + 4 * mm) : "memory"); }
              } else {      // the own contribution (registers: nothing orders another workgroup's read after a store of this one, so it is never read back), or past the last rank
                vT[q] = r < a.px_n ? oT : 0.f;
#pragma unroll
                for (int sec = 0; sec < NSEC; ++sec) {
#pragma unroll
                  for (int mm = 0; mm < WT; ++mm) vW[sec][q][mm] = r < a.px_n ? oW[sec][mm] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                  for (int k = 0; k < NSI; ++k) vS[sec][q][k] = r < a.px_n ? oS[sec][k] : 0.f; } } }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int sec = 0; sec < NSEC; ++sec)
#pragma unroll
              for (int q = 0; q < PXS; ++q)
#pragma unroll
                for (int mm = 0; mm < WT; ++mm) asm volatile("" : "+v"(vW[sec][q][mm]));
#pragma unroll
            for (int q = 0; q < PXS; ++q) {
              if (r0 + q >= a.px_n) break;
              if (r0 + q == 0) {
#pragma unroll
                for (int sec = 0; sec < NSEC; ++sec) {
#pragma unroll
                  for (int mm = 0; mm < WT; ++mm) Wp[sec][mm] = vW[sec][q][mm];
#pragma unroll
                  for (int k = 0; k < NSI; ++k) Sp[sec][k] = vS[sec][q][k]; }
                xT = vT[q];
              } else {
#pragma unroll
                for (int sec = 0; sec < NSEC; ++sec) {
#pragma unroll
                  for (int mm = 0; mm < WT; ++mm) Wp[sec][mm] += vW[sec][q][mm];
#pragma unroll
                  for (int k = 0; k < NSI; ++k) Sp[sec][k] += vS[sec][q][k]; }
                xT += vT[q];
              }
            }
          }
          // mean over the group (gradient: global minibatch = px_n x nb samples, every rank's partial was already divided by nb)
#pragma unroll
          for (int sec = 0; sec < NSEC; ++sec) {
#pragma unroll
            for (int mm = 0; mm < WT; ++mm) Wp[sec][mm] = Wp[sec][mm] * px_inv;
#pragma unroll
            for (int k = 0; k < NSI; ++k) Sp[sec][k] = Sp[sec][k] * px_inv; }
          xT = xT * px_inv;
          pxc += 1;
          return true;
      };
      float stat_tot = stat_loc; bool suspect = true; float kl_tot = 0.f;
      // ---- exchange the partial gradients with the other workgroups through the shared L2 ----
      {
#pragma unroll
        for (int k = 0; k < NSI; ++k) mine[W2N + tid + NT * k] = gs[k];
        if (tid >= NT - 8 && tid < STAT_HI) mine[W2N + NSI * NT + (tid - (NT - 8))] = stat_loc;
        if constexpr (ELIDE) {
#pragma unroll
          for (int k = 0; k < NSI; ++k) odd = odd || !(fabsf(gs[k]) <= 1e30f);
          if (odd) mine[SUS] = 1.f; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every store of this lane has reached the L2 (the dW2 partials left before dH1: long acknowledged)
        FS_T(10);
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(a.xctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // second arrival: the small partials are in the L2 too
        // (Arrival and polling per WAVE -- no workgroup barrier before the arrival, none to pass the news on -- was measured and dropped: 32 waves adding to and polling one L2
        //  word, 6.80 us per step against 6.18; C5 9.38 against 8.54.)
        // the wait for the slowest workgroup is spent staging the NEXT minibatch (rows prefetched a step ago; the tiles' x / scalar rows are free after B_2)
        if (st + a.bs < total_rows) { if (!HELP) stage(0); staged = true; }      // (with helper waves the next minibatch is staged already)
        if (tid == 0) {
          const unsigned want = (unsigned)NWG * (unsigned)(xstep + 1); unsigned spins = 0; bool ok = true;
          while (__hip_atomic_load(a.xctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) { __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 24) || __hip_atomic_load(a.xctr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = false; break; } }   // never hang the GPU
          float why = ok ? 0.f : 1.f;                                   // 1: a workgroup never arrived (or raised the abort word)
          if (ok && xstep == 0) {   // the unfenced exchange is only coherent inside one XCD's L2
            for (int q = 0; q < NWG; ++q) { const unsigned peer_xcc = __hip_atomic_load(a.xctr + 8 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (peer_xcc != my_xcc + 1u) { ok = false; why = 2.f; } } }   // 2: the workgroups of this learner sit on different XCDs
          if (!ok) __hip_atomic_store(a.xctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          sm[Lt::oRED + 16] = why;
        }
        __syncthreads();
        if (sm[Lt::oRED + 16] != 0.f) { err = CRUX_EHIP; why_failed = (int)sm[Lt::oRED + 16]; break; }
        FS_T(11);
        // The total is formed as (s0 + s1) + (s2 + s3): a workgroup adds its own contribution (registers) to its pair partner's slot -- a + b == b + a bitwise, so both
        // partners hold the same pair sum --, the other pair's two slots in index order, and then the two pair sums, again a commutative add: the same bits in all four
        // workgroups without reading the own slot back and without any selection by p. Two workgroups: own + peer.
        // (A two-phase form -- the dW2 partials, 89 % of a slot, announced by a counter of their own and loaded before the second arrival -- was measured and dropped: the
        //  exchange is a chain of L2 round trips, not a bandwidth problem, and every variant added a round trip: 7.19 / 7.59 us per step against 7.10 us.)
        constexpr int NLD = NWG - 1;
        f32x4 pw[NLD][WT]; float pg[NLD][NSI]; float ps[NLD]; float psus[NLD + 1], pkl[NLD + 1];
        if constexpr (ELIDE) { psus[NLD] = __hip_atomic_load(mine + SUS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); pkl[NLD] = __hip_atomic_load(mine + KLW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#pragma unroll
        for (int j = 0; j < NLD; ++j) { const int q = j == 0 ? (p ^ 1) : ((p ^ 2) & 2) + (j - 1);      // partner, then the other pair's first and second workgroup
          const float* peer = a.xbuf + (size_t)(((int)(xstep & 1) * NWG + q)) * XSLOT;
          if constexpr (ELIDE) { psus[j] = __hip_atomic_load(peer + SUS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); pkl[j] = __hip_atomic_load(peer + KLW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#pragma unroll
          for (int k = 0; k < NSI; ++k) pg[j][k] = __hip_atomic_load(peer + W2N + tid + NT * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ps[j] = 0.f;
          if (tid >= NT - 8 && tid < STAT_HI) ps[j] = __hip_atomic_load(peer + W2N + NSI * NT + (tid - (NT - 8)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
          for (int mm = 0; mm < WT; ++mm)
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(pw[j][mm]) : "v"(peer + tid * (4 * WT) + 4 * mm) : "memory"); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < NLD; ++j)
#pragma unroll
          for (int mm = 0; mm < WT; ++mm) asm volatile("" : "+v"(pw[j][mm]));      // (asm statements keep their order: every use of a loaded value follows the wait)
        if constexpr (NWG == 2) {
#pragma unroll
          for (int mm = 0; mm < WT; ++mm) gW2[mm] += pw[0][mm];
#pragma unroll
          for (int k = 0; k < NSI; ++k) gs[k] += pg[0][k];
          stat_tot = stat_loc + ps[0];
          if constexpr (ELIDE) { suspect = psus[0] != 0.f || psus[1] != 0.f; kl_tot = pkl[1] + pkl[0]; }
        } else {
#pragma unroll
          for (int mm = 0; mm < WT; ++mm) gW2[mm] = (gW2[mm] + pw[0][mm]) + (pw[1][mm] + pw[2][mm]);
#pragma unroll
          for (int k = 0; k < NSI; ++k) gs[k] = (gs[k] + pg[0][k]) + (pg[1][k] + pg[2][k]);
          stat_tot = (stat_loc + ps[0]) + (ps[1] + ps[2]);
          if constexpr (ELIDE) { suspect = psus[0] != 0.f || psus[1] != 0.f || psus[2] != 0.f || psus[3] != 0.f; kl_tot = (pkl[3] + pkl[0]) + (pkl[1] + pkl[2]); }      // (the stat lanes' own additions)
        }
        if constexpr (PX && !PXK) {
          f32x4* const Wp[3] = {gW2, nullptr, nullptr}; float* const Sp[3] = {gs, nullptr, nullptr};
          if (!px_allreduce_mean(std::integral_constant<int, 1>{}, Wp, Sp, stat_tot)) break;
        }
        xstep += 1;
      }
      if (tid >= NT - 8 && tid < STAT_HI) sm[Lt::oRED + 8 + (tid - (NT - 8))] = stat_tot;
      float ssq = 0.f; int bad = 0;
#pragma unroll
      for (int k = 0; k < NSI; ++k) if (so_ok[k]) {
        if (KIND == MFK_GAUSSIAN && so_ex[k]) gs[k] += LAG ? -a.lambda_e / (1.f + pen) : -a.lambda_e; }      // d(-lambda_e H)/dlogSigma, H = const + sum(logSigma); lagrange: the whole loss is divided by 1 + penalty
      bool need_bar = true; float kl_now = 0.f;
      if constexpr (ELIDE) { kl_now = kl_tot * invB;
        need_bar = suspect || st + a.bs >= total_rows || (a.max_batches > 0 && total_batches + 1 >= a.max_batches) || (KIND != MFK_VALUE && a.target_kl >= 0.f && kl_now > a.target_kl); }
      int any_bad = 0;
      if (need_bar) {      // (uniform over the learner's workgroups: every term is the same in all of them)
#pragma unroll
        for (int k = 0; k < NSI; ++k) if (so_ok[k]) { ssq += gs[k] * gs[k]; bad |= isnan(gs[k]) ? 1 : 0; }
#pragma unroll
        for (int mm = 0; mm < WT; ++mm)
#pragma unroll
          for (int r = 0; r < 4; ++r) { ssq += gW2[mm][r] * gW2[mm][r]; bad |= isnan(gW2[mm][r]) ? 1 : 0; }
        ssq = wave_sum(ssq);
        if (lane == 0) sm[Lt::oRED + w] = ssq;
        FS_T(12);
        any_bad = __syncthreads_or(bad);   // ---- B_or (also publishes RED)
      }
      FS_T(13);
      // minibatch info (training.jl:22-23, ppo.jl:13-19); identical in every thread; only the epoch's last minibatch (or the one that stops the loop) is ever reported
      { const float* tq = sm + Lt::oRED + 8;
        if (KIND != MFK_VALUE && a.target_kl >= 0.f) inf_kl = need_bar ? tq[2] * invB : kl_now;
        const bool report = need_bar && (any_bad || st + a.bs >= total_rows || (a.max_batches > 0 && total_batches + 1 >= a.max_batches) ||
                            (KIND != MFK_VALUE && a.target_kl >= 0.f && inf_kl > a.target_kl));
        if (report) {
          float ss = sm[Lt::oRED];
#pragma unroll
          for (int q = 1; q < NW; ++q) ss += sm[Lt::oRED + q];
          if (KIND != MFK_VALUE) inf_kl = tq[2] * invB;
          if (tid == 0) {
          sm[iGN] = sqrtf(ss);
          if (KIND == MFK_VALUE) { sm[iLOSS] = tq[6] * invB; sm[iRET] = tq[4] * invB; }
          else { const float p_loss = -(tq[0] * invB); float entropy;
            if (KIND == MFK_CATEGORICAL) entropy = tq[1] * invB;
            else { entropy = 1.4189385332046727f;
#pragma unroll
              for (int k = 0; k < OUT; ++k) entropy += sm[Lt::oEX + k]; }
            sm[iENT] = entropy; sm[iLOSS] = fmaf(a.lambda_p, p_loss, a.lambda_e * (-entropy));      /* (explicit fma: the same bits in every form of the kernel) */ sm[iADV] = tq[3] * invB; sm[iRET] = tq[4] * invB; sm[iCLIP] = tq[5] * invB;
            if constexpr (LAG) { const float cost_loss = pen * (tq[7] * invB);                                        // ppo.jl:119
              sm[iLOSS] = ((a.lambda_p * p_loss + a.lambda_e * (-entropy)) + cost_loss) / (1.f + pen);                  // :131
              sm[iPEN] = pen; sm[iCUR] = lgs[6]; sm[iCLOSS] = cost_loss; sm[iPLOSS] = a.lambda_p * p_loss; } }
          }
        }
      }
      if (any_bad) { if (tid == 0) sm[iGN] = NAN; err = CRUX_ENAN; break; }      // training.jl:20: no update
      // ======================= Adam (Flux.update!, training.jl:21) =======================
#pragma unroll
      for (int mm = 0; mm < WT; ++mm) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { float m_ = mW2[mm][r], v_ = vW2[mm][r]; const float d = adam1(gW2[mm][r], m_, v_, ak);
          mW2[mm][r] = m_; vW2[mm][r] = v_; tW2[mm][r] -= d;
          sm[Lt::oW2R + (16 * mp0 + 4 * g + r) * FS_LD + 16 * (m0 + mm) + c] = tW2[mm][r]; }
        *(f32x4*)&sm[Lt::oW2C + (16 * (m0 + mm) + c) * FS_LD + 16 * mp0 + 4 * g] = tW2[mm]; }
#pragma unroll
      for (int k = 0; k < NSI; ++k) { const int s = tid + NT * k;
        if (so_ok[k]) { float m_ = sm[Lt::oMS + s], v_ = sm[Lt::oVS + s]; const float d = adam1(gs[k], m_, v_, ak);
          sm[Lt::oMS + s] = m_; sm[Lt::oVS + s] = v_; const int mo = so_master[k]; sm[mo] = sm[mo] - d; } }
      bp1 *= a.b1; bp2 *= a.b2;
      if constexpr (PX && PXK) {
        // ---- periodic form (crux_peer_set_sync_every(k > 1)): between exchanges every replica takes LOCAL Adam steps on its own shard; after every k-th step the group averages
        // theta, m and v (sum in rank order x 1/N: the same bits everywhere, so the replicas leave the exchange identical). One exchange per k steps instead of k.
        if ((total_batches + 1) % a.px_every == 0) {
          float sT[NSI], sM[NSI], sV[NSI]; float dummy = 0.f;
#pragma unroll
          for (int k = 0; k < NSI; ++k) { const int s = tid + NT * k; const bool ok_ = so_ok[k];
            sT[k] = ok_ ? sm[so_master[k]] : 0.f; sM[k] = ok_ ? sm[Lt::oMS + s] : 0.f; sV[k] = ok_ ? sm[Lt::oVS + s] : 0.f; }
          f32x4* const Wp[3] = {tW2, mW2, vW2}; float* const Sp[3] = {sT, sM, sV};
          if (!px_allreduce_mean(std::integral_constant<int, 3>{}, Wp, Sp, dummy)) break;
#pragma unroll
          for (int k = 0; k < NSI; ++k) { const int s = tid + NT * k;
            if (so_ok[k]) { sm[so_master[k]] = sT[k]; sm[Lt::oMS + s] = sM[k]; sm[Lt::oVS + s] = sV[k]; } }
#pragma unroll
          for (int mm = 0; mm < WT; ++mm) {      // the LDS copies of W2 follow the averaged registers
#pragma unroll
            for (int r = 0; r < 4; ++r) sm[Lt::oW2R + (16 * mp0 + 4 * g + r) * FS_LD + 16 * (m0 + mm) + c] = tW2[mm][r];
            *(f32x4*)&sm[Lt::oW2C + (16 * (m0 + mm) + c) * FS_LD + 16 * mp0 + 4 * g] = tW2[mm]; }
        }
      }
      FS_T(14);
      __syncthreads();   // ---- B_b: masters updated; tiles and partials may be overwritten
      if (HELP) xcur ^= 1;
      FS_T(15);
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
  if (p == 0) {
#pragma unroll
    for (int mm = 0; mm < WT; ++mm)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int pc = Lt::cW2 + (16 * mp0 + 4 * g + r) + H2 * (16 * (m0 + mm) + c);
        a.p[pc] = tW2[mm][r]; a.m[pc] = mW2[mm][r]; a.v[pc] = vW2[mm][r]; }
    for (int s = tid; s < ns_valid; s += NT) { const int pc = s_canon(s); a.p[pc] = sm[s_master(s)]; a.m[pc] = sm[Lt::oMS + s]; a.v[pc] = sm[Lt::oVS + s]; }
  }
  if (TIMING && lane == 0 && a.dbg) { for (int k = 0; k < 16; ++k) a.dbg[(NW * p + w) * 16 + k] = tacc[k]; }
  if (PX && tid == 0 && p == 0) *(unsigned long long*)(px_mine + CRUX_PX_COUNT) = px0 + (unsigned long long)pxc;
  if constexpr (PX && PXK) {      // periodic form: a replica that leaves on a NaN step between two exchanges will not show up at the next one -- its peers must not wait for the timeout
    if (tid == 0 && p == 0 && err == CRUX_ENAN) { for (int r = 0; r < a.px_n; ++r) if (r != a.px_rank) __hip_atomic_store((unsigned*)(a.px_tab[r] + CRUX_PX_ABORT), 5u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } }
  if (tid == 0 && (p == 0 || err)) {
    a.status[0] = err; a.status[1] = (int32_t)total_batches; a.status[2] = epochs_run; a.status[3] = (order_cur == a.order_a) ? 0 : 1;
    if (err == CRUX_EHIP) a.status[4] = why_failed;      // 1 a workgroup of the learner is missing, 2 workgroups on different XCDs, 3 replica group timeout / abort, 4 a workgroup missed the abort-latch consensus of a speculative run
    a.bp[0] = bp1; a.bp[1] = bp2;
    if constexpr (LAG) { if (p == 0) { a.lag->I = lgs[0]; a.lag->smooth_delta = lgs[1]; a.lag->smooth_Jc = lgs[2]; a.lag->Jc_prev = lgs[3]; a.lag->deriv_term = lgs[4]; a.lag->penalty = lgs[5]; a.lag->cur_cost = lgs[6]; } }
    if (err && a.epoch_infos && epochs_run == 0) { a.epoch_infos[CRUX_INFO_LOSS] = sm[iLOSS]; a.epoch_infos[CRUX_INFO_GRAD_NORM] = NAN; }
  }
#undef FS_T
}
