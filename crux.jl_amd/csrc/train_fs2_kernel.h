// train_fs2_kernel.h -- the persistent batch_train! kernel of the register-resident IN->64->{64,32}->OUT family (src/training.jl:13-55, ppo.jl:4-21,59-60,70-131, Flux Adam):
// FEATURE-SPLIT decomposition of a minibatch step over four compute units of one XCD, ROLE-SPECIALISED waves.
//
// The decomposition (round 3):
//   * a 16-sample tile of the minibatch belongs to a PAIR of waves (t, h), h in {0, 1}: wave h computes the hidden features [32h, 32h + 32) of the second layer, of its
//     gradient and of the first layer's gradient -- half of the 64x64 MFMA work (forward 32, dH1 32 instead of 64 each) and half of the per-feature VALU / LDS work of the
//     tile. The first layer (IN <= 27 inputs: 4..28 MFMAs) is evaluated by both waves, so the second layer needs no exchange; the pair meets twice per step through LDS:
//     the partial logits z (OUT values per sample) before the loss head, and the other half of dZ2 (A operand of dH1 = W2' dZ2) after it;
//   * four workgroups (compute units of ONE XCD) share the 128 samples: 32 samples = 2 tiles = 4 compute waves per workgroup, one per SIMD. Per wave and step: 4 (L1) + 32 (L2)
//     + 32 (dW2) + 32 (dH1) + 8 (dW1) = 108 MFMAs against 212 in the sample-split form (train_mfma_kernel.h) -- the step loop is instruction-issue bound, and this
//     halves the per-feature instruction stream of a wave;
//   * the 64x64 weight gradient stays model-parallel over the waves of a workgroup (two 16x16 tiles of W2 with theta, m, v in the owning wave's registers); it is formed
//     BEFORE dH1 and its partial sums leave for the other workgroups' L2 slots at once, so the store acknowledgement is covered by the rest of the backward pass;
//   * the four partial gradients are exchanged once per step through the shared L2 (plain stores, sc1 loads) and added in a fixed order -- (own + partner) + (other pair) -- by
//     everyone: bit-identical totals, Adam updates and early-stopping decisions in all workgroups.
// The roles (round 5; until round 6 a second kernel, k_train_fs, ran the same arithmetic with every wave in every role -- 40 % of its step was the tail, a chain of L2 round trips):
//
//   COMPUTE waves (0..3)                                        HELPER waves (4..7)
//   forward L1, L2, partial logits                              prefetch minibatch k+2's row indices, k+1's rows; stage minibatch k+1
//   -- pair barrier (LDS) --
//   loss head, dW3 / db3 partials, dZ2 -> T2 tiles
//   ============================== B_1 (s_barrier): H1 / dZ2 tiles of the workgroup are visible ==============================
//   dW2 of the own two tiles -> exchange slot                   dW2 of the own two tiles -> exchange slot, drain the stores
//   dH1; the stores are acknowledged by now: tell the leader    leader: all eight waves' stores acknowledged -> ARRIVAL 1 -> wait for the four workgroups -> P1
//   dZ1, db1, db2, dW1 partials                                 P1: load the three peers' partials of the own tiles, total, Adam (theta, m, v in registers)
//   -- compute barrier (LDS) --
//   small partials -> 8-byte {value, step} GRANULES in the slot
//   P1: load the peers' partials of the own W2 tiles -> W2 total + Adam (the small granules are still on their way)
//   poll the peers' granules (the tag is the flag: ONE round trip) -> small total + Adam
//   (replica group, PX: the group's all-reduce of the totals sits between the totals and Adam; PXK: its theta / m / v average after every k-th Adam step)
//   ============================== B_b (s_barrier): masters updated ==========================================================
//
//   (1) the W2 partials (89 % of the bytes) leave right after dW2 as before, but their hand-shake (drain, arrival counter, wait) is now run by the helper leader WHILE the
//       compute waves are in dH1 / dZ1 / dW1: when the small partials are ready, the peers' W2 partials are known to be in the L2 already;
//   (2) the small partials, the last bytes of the step, travel as data-tagged granules -- store, then poll the peers' granules until the tag is the step's: one L2 round trip
//       where the flag protocol needs three (drain, arrival + wait, load);
//   (3) each role is its own code path: the register allocation is the maximum of the two roles, not their union -- no plain or lagrange instantiation spills;
//   (4) two workgroup barriers per step instead of five; pairs and roles meet through LDS counters.
// NaN semantics (training.jl:20: a NaN gradient norm is an error BEFORE the update): every thread forms the totals of its own elements in every workgroup; one that finds a
// NaN total marks the step SUSPECT in LDS; after B_b the whole workgroup puts the pre-step state back (W2 from registers, the small parameters from the values read for Adam)
// and leaves with CRUX_ENAN -- a NaN step leaves every parameter as it was. The same totals, hence the same decision, in all four workgroups.
// Covers the plain policy-gradient / critic losses of full minibatch loops (65..128 rows), alone or as a member of a replica group (PX: the group's all-reduce of the minibatch
// gradient between the totals and Adam; PXK: local steps, theta / m / v averaged every k-th), and lagrange_ppo_loss (LAG).
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

// CRUX_FS2_EXP (development builds only; tools/fs4_bound.sh): a TIMING experiment with WRONG results -- what would a step cost if every compute wave ran the MFMA chain of a
// four-waves-per-tile decomposition (half the second-layer, dH1 and dW1 MFMAs of the pair form)? 1: the chain is cut (no extra load anywhere: the upper bound of the gain);
// 2: the helper wave that shares the SIMD additionally issues the MFMAs and VALU work the second pair of compute waves would (the contention a real quad form has);
// 3: as 2 with the MFMAs only.
#ifndef CRUX_FS2_EXP
#define CRUX_FS2_EXP 0
#endif

template <int IN, int OUT, int H2, bool LAG = false>
struct Fs2Layout : FsLayout<IN, OUT, 4, true, H2, LAG> {
  using B = FsLayout<IN, OUT, 4, true, H2, LAG>;
  static constexpr int NTC = 256;                                   // compute threads = helper threads
  static constexpr int NSC = (B::NS + 511) / 512;                   // small parameters per thread (all 512 threads share them)
  static constexpr int WT2 = (B::NT2 * 4) / 8;                      // W2 tiles per wave (all eight waves own tiles)
  static constexpr int NGR = NSC * 512 + 8;                         // granules of a slot: the small partials, then the 7 statistics sums
  static constexpr int xW2 = 0, xGR = B::W2N;                       // exchange slot: W2 partials [thread][WT2][4] | granules {value, step} of the small partials and statistics
  static constexpr int XSLOT2 = ((xGR + 2 * NGR + 3) / 4) * 4;
  static_assert(2 * 4 * XSLOT2 <= CRUX_XBUF_FLOATS, "exchange area");
  // every thread's copy of its W2 theta / m / v from before the step's Adam (what a suspect step is put back to): in registers (plain learners: the LDS form measured 3 %
  // slower, 5.58 against 5.40 us per C2 step), in LDS in the replica-group forms, whose exchange needs the 24 registers (no spills in the C5 PX / PXK forms)
  static constexpr bool BK_FITS = B::TOTAL + 3 * B::W2N <= 40960;
  static constexpr int oBK = B::TOTAL, TOTAL_BK = B::TOTAL + 3 * B::W2N;
  // LDS words behind the reduction area (oRED .. oRED + 32): group-barrier counters and step flags
  // cACK: waves whose W2 stores are acknowledged (eight per step); cPAIR + t: the two waves of tile t, once per step; cCOMP: the four compute waves, once per step;
  // fP1: steps whose phase 1 is complete (helper leader -> everyone); fSUS: the last suspect step; fERR: why the exchange failed (sticky)
  static constexpr int cACK = B::oRED + 18, cPAIR = B::oRED + 24, cCOMP = B::oRED + 26, fP1 = B::oRED + 29, fSUS = B::oRED + 30, fERR = B::oRED + 31;
  // the reported minibatch's info (training.jl:22-23), kept by thread 0 -- its only reader (epoch_infos) -- in free words of the same area instead of six registers per thread
  // that would stay live across the whole launch: loss, grad norm, entropy, clip fraction, avg advantage, avg return (the KL stays in a register: every thread's loop exit reads it)
  static constexpr int iLOSS = B::oRED + 19, iGN = B::oRED + 20, iENT = B::oRED + 21, iCLIP = B::oRED + 22, iADV = B::oRED + 23, iRET = B::oRED + 27;
  // lagrange_ppo_loss: the reported penalty and current cost are words 5 and 6 of wave 0's copy of the controller state (B::oLGS); cost_loss and p_loss take the spare eighth words
  // of wave 0's and wave 1's copies
  static constexpr int iPEN = B::oLGS + 5, iCUR = B::oLGS + 6, iCLOSS = B::oLGS + 7, iPLOSS = B::oLGS + 15;
};

// meeting point of a subset of the workgroup's waves: one LDS counter, monotonic over the launch (target = members x number of uses so far). A wave's LDS operations execute in
// order, so everything it wrote before the add is in the LDS before the add is.
__device__ __forceinline__ void fs2_group_barrier(float* word, unsigned target, int lane) {
  unsigned* cnt = (unsigned*)word;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0) (void)__hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(0);
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void fs2_flag_set(float* word, unsigned v) { __hip_atomic_store((unsigned*)word, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ unsigned fs2_flag_get(const float* word) { return __hip_atomic_load((const unsigned*)word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void fs2_flag_wait(const float* word, unsigned target) {
  while (fs2_flag_get(word) < target) __builtin_amdgcn_s_sleep(0);
  asm volatile("" ::: "memory");
}

// LAG: lagrange_ppo_loss (ppo.jl:70-131) -- the PID penalty controller advanced once per minibatch inside the kernel (every wave, from the :cost / :episode_end columns of the
// WHOLE minibatch, which two helper waves stage beside the tiles) and the cost-advantage term of the loss head. A separate instantiation: the plain kernels carry none of it.
template <int IN, int OUT, int KIND, int ACT, int H2 = 64, int ACT2 = ACT, bool TIMING = false, bool PX = false, bool PXK = false, bool LAG = false>
__global__ __launch_bounds__(512) void k_train_fs2(TrainArgs a) {
  static_assert(!PXK || PX, "PXK: the periodic form of the replica group");
  static_assert(!LAG || (!PX && KIND != MFK_VALUE), "lagrange_ppo_loss: policy heads, one replica");
  static_assert(!PX || FsLayout<IN, OUT, 4, true, H2, false>::W2N + Fs2Layout<IN, OUT, H2>::NSC * 512 + 8 <= CRUX_PX_SEC, "a payload section must fit CRUX_PX_SEC");
  using Lt = Fs2Layout<IN, OUT, H2, LAG>;
  constexpr int NWG = 4, NWC = 4, TILES = 2, NT = 512, NTC = 256, MH = Lt::MH, HH = Lt::HH, W2N = Lt::W2N, NW = 8;
  constexpr int WT = Lt::WT2;                       // 16x16 tiles of W2 owned by a wave (all eight waves own tiles)
  constexpr int KS0 = Lt::KS0, IP = Lt::IP, JT = Lt::JT, XP = Lt::XP, NS = Lt::NS, NSC = Lt::NSC, XSLOT = Lt::XSLOT2;
  constexpr int NACT = (OUT > 4 ? OUT : 4);
  constexpr bool BK_LDS = PX && Lt::BK_FITS;
  if ((int)(blockIdx.x & 7) != a.xcd) return;        // the four workgroups of the learner: blocks x, x + 8, x + 16, x + 24 -> one XCD
  const int p = (int)(blockIdx.x >> 3);
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, g = lane >> 4;
  const bool cw = w < NWC;                           // compute wave
  const int wt = w & (NWC - 1);                      // (tile, half) role: of the tile work for a compute wave, of the staging work for a helper
  const int t = wt >> 1, h = wt & 1;
  const int ct = tid & (NTC - 1);                    // index inside the role's 256 threads
  float* part = sm + Lt::oPART + t * Lt::PART;
  float* xs = sm + Lt::oXS + t * 16 * XP;
  float* sc = sm + Lt::oSC + t * 16 * Lt::SCW;
  constexpr int XSB = TILES * 16 * XP, SCB = TILES * 16 * Lt::SCW;
  int xcur = 0;
  float* T1 = sm + Lt::oT1 + t * Lt::TILE;
  float* T2 = sm + Lt::oT2 + t * Lt::TILE2;
  const int n_extra = (KIND == MFK_GAUSSIAN) ? OUT : 0;
  unsigned long long tacc[16]; unsigned long long tlast = 0;
  if (TIMING) { for (int k = 0; k < 16; ++k) tacc[k] = 0; tlast = __builtin_amdgcn_s_memtime(); }
#define FS2_T(ph) do { if (TIMING) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); tacc[ph] += tn - tlast; tlast = tn; } } while (0)
  const int t_wr = (4 * g) * 16 + 4 * ((c >> 2) ^ fs_tx(g)) + (c & 3);
  const int t_rd = c * 16 + 4 * (g ^ fs_tx(c >> 2));
  // dW2 / W2 ownership: tile (mp, m) = rows [16mp, 16mp+16) x columns [16m, 16m+16), numbered 4 mp + m; wave w owns the WT tiles from number w WT on (same mp)
  const int mp0 = (w * WT) >> 2, m0 = (w * WT) & 3;

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

  // ---- load parameters and Adam state (all 512 threads) ------------------------------------------------------------
  for (int q = tid; q < W2N; q += NT) { const int o = q % H2, i = q / H2; const float v = a.p[Lt::cW2 + q];
    sm[Lt::oW2R + o * FS_LD + i] = v; sm[Lt::oW2C + i * FS_LD + o] = v; }
  for (int q = tid; q < MF_HID * Lt::W1LD; q += NT) sm[Lt::oW1R + q] = 0.f;
  if (tid < 16) { sm[Lt::oB3 + tid] = 0.f; sm[Lt::oEX + tid] = 0.f; }
  if (tid < 32) sm[Lt::oRED + tid] = 0.f;            // statistics, group-barrier counters and step flags start from zero
  for (int q = tid; q < TILES * Lt::PART; q += NT) sm[Lt::oPART + q] = 0.f;
  uint32_t my_xcc = 0;
  if (tid == 0) { asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my_xcc)); my_xcc &= 0xf;
    __hip_atomic_store(a.xctr + 8 + p, my_xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  __syncthreads();
  for (int s = tid; s < NS; s += NT) { const bool in = s < ns_valid; const int pc = s_canon(s);
    if (in) sm[s_master(s)] = a.p[pc];
    sm[Lt::oMS + s] = in ? a.m[pc] : 0.f; sm[Lt::oVS + s] = in ? a.v[pc] : 0.f; }
  for (int q = tid; q < Lt::NXB * TILES * 16 * XP; q += NT) sm[Lt::oXS + q] = 0.f;
  double bp1 = a.bp[0], bp2 = a.bp[1];
  const float lo = 1.f - a.eps_clip, hi = 1.f + a.eps_clip;
  const bool a2c = a.loss == CRUX_LOSS_A2C;
  AdamK ak; ak.b1 = (float)a.b1; ak.b2 = (float)a.b2; ak.omb1 = (float)(1.0 - a.b1); ak.omb2 = (float)(1.0 - a.b2); ak.eps = (float)a.eps; ak.eta = (float)a.eta;
  const double db1 = a.b1, db2 = a.b2;
  const float lambda_p = a.lambda_p, lambda_e = a.lambda_e, target_kl = a.target_kl, squash = a.squash;
  float pen = 0.f;                                     // lagrange_ppo_loss: the penalty of the current minibatch; the controller's state sits in LDS, one copy per wave
  float* lgs = sm + Lt::oLGS + 8 * w;                  // [I, smooth_delta, smooth_Jc, Jc_prev, deriv_term, penalty, cur_cost, -]
  if constexpr (LAG) { if (lane == 0) { lgs[0] = a.lag->I; lgs[1] = a.lag->smooth_delta; lgs[2] = a.lag->smooth_Jc; lgs[3] = a.lag->Jc_prev; lgs[4] = a.lag->deriv_term; lgs[5] = a.lag->penalty; lgs[6] = a.lag->cur_cost; lgs[7] = 0.f; } }
  // the penalty update inside the loss (ppo.jl:80-116), once per evaluation, by every wave of both roles at the top of its step: sums of the staged minibatch's :cost and
  // :episode_end (buffer `buf`, visible since the previous step's B_1 or the epoch's first barrier), then the controller -- the same bits in every wave of every workgroup
  auto lag_advance = [&](int buf) {
    const float* lc = sm + Lt::oLAG + 256 * buf;
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
  };
  // 32-bit loop control (the dispatcher sends only buffers below 2^30 rows and launches below 2^31 steps here): 64-bit counters cost SGPR pairs the wide heads do not have
  const int max_batches = a.max_batches > 0 && a.max_batches < 0x7fffffffll ? (int)a.max_batches : 0;
  const int bs = a.bs;

  int32_t* order_cur = a.order_a; int32_t* order_nxt = a.order_b;
  int total_batches = 0; int epochs_run = 0, err = 0, why_failed = 0; bool stop = false;
  bool staged = false;
  unsigned xstep = 0;                                // exchanges of this launch: every group-barrier target and both arrival counters are multiples of xstep + 1
  float inf_kl = 0.f;
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
  const int total_rows = (int)a.len;

  // what the minibatch loops of both roles share. Only the epoch's last minibatch, or the one that ends the loop, is ever reported (training.jl:22-23, 45-53).
  auto static_report = [&](int st) -> bool { return st + bs >= total_rows || (max_batches > 0 && total_batches + 1 >= max_batches); };
  // the end of a minibatch step after B_b, identical in every thread of the workgroup (its inputs are LDS words published before the barrier): the KL statistic and the loop exits.
  // false = the minibatch loop ends here.
  auto step_exit = [&](float invB, bool any_bad) -> bool {
    if (KIND != MFK_VALUE && target_kl >= 0.f) inf_kl = sm[Lt::oRED + 8 + 2] * invB;
    if (any_bad) { if (tid == 0) sm[Lt::iGN] = NAN; err = CRUX_ENAN; return false; }      // training.jl:20: no update
    total_batches += 1;
    if (max_batches > 0 && total_batches >= max_batches) return false;                // training.jl:45
    if (target_kl >= 0.f && KIND != MFK_VALUE && inf_kl > target_kl) return false;    // :46
    return true;
  };
  // both roles run the same epoch prologue (same barriers): the speculative-run consensus and the epoch's shuffle order. false = leave the epoch loop.
  auto epoch_prologue = [&](int ep) -> bool {
    if (a.spec_abort) {
      if (tid == 0) {
        const unsigned r = __hip_atomic_load(a.spec_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        (void)__hip_atomic_fetch_or(a.xctr + 16, r ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        (void)__hip_atomic_fetch_add(a.xctr + 17, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned want = (unsigned)NWG * (unsigned)(ep + 1); unsigned spins = 0; bool late = false;
        while (__hip_atomic_load(a.xctr + 17, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) { __builtin_amdgcn_s_sleep(2); if (++spins > (1u << 24)) { late = true; break; } }
        if (late) { (void)__hip_atomic_fetch_or(a.xctr + 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(a.xctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        sm[Lt::oRED + 17] = late ? 2.f : (__hip_atomic_load(a.xctr + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1.f : 0.f); }
      __syncthreads();
      if (sm[Lt::oRED + 17] == 2.f) { err = CRUX_EHIP; why_failed = 4; return false; }
      if (sm[Lt::oRED + 17] != 0.f) { err = CRUX_TRAIN_ABORTED; return false; }
    }
    if (a.ord_all) order_cur = const_cast<int32_t*>(a.ord_all) + (size_t)ep * (size_t)a.len;   // shuffle orders composed ahead of time by k_compose_order
    else {   // shuffle!(D) as an index composition (experience_buffer.jl:118-124)
      // (the thread index is laundered: from the plain `tid` the compiler hoists the 64-bit element addresses of this once-per-epoch loop out of the epoch loop and keeps them
      //  alive across the whole launch -- in scratch, in the forms that sit at the 256-register limit: the two spilled dwords of the C5 periodic form, round 6)
      int tidl = tid; asm volatile("" : "+v"(tidl));
      if (a.perms) { for (int64_t j = tidl; j < a.len; j += NT) order_nxt[j] = order_cur[a.perms[(int64_t)ep * a.len + j]]; }
      else { const crux_perm pp = crux_perm_make(a.shuffle_seed, a.shuffle_counter + (uint64_t)ep, 0, (uint32_t)a.len);
        for (int64_t j = tidl; j < a.len; j += NT) order_nxt[j] = order_cur[crux_perm_at(&pp, (uint32_t)j)]; }
      __syncthreads();
      int32_t* tq = order_cur; order_cur = order_nxt; order_nxt = tq;
    }
    return true;
  };
  auto epoch_epilogue = [&](int ep) {
    if (tid == 0 && p == 0 && a.epoch_infos) { float* e = a.epoch_infos + (size_t)ep * CRUX_INFO_N;   // aggregate_info(minibatch_infos) == last minibatch (Q3)
      for (int k = 0; k < CRUX_INFO_N; ++k) e[k] = 0.f;
      e[CRUX_INFO_LOSS] = sm[Lt::iLOSS]; e[CRUX_INFO_GRAD_NORM] = sm[Lt::iGN];
      if (KIND != MFK_VALUE) { e[CRUX_INFO_ENTROPY] = sm[Lt::iENT]; e[CRUX_INFO_KL] = inf_kl; e[CRUX_INFO_CLIP_FRACTION] = sm[Lt::iCLIP]; e[CRUX_INFO_AVG_ADVANTAGE] = sm[Lt::iADV]; e[CRUX_INFO_AVG_RETURN] = sm[Lt::iRET]; }
      if constexpr (LAG) { e[CRUX_INFO_PENALTY] = sm[Lt::iPEN]; e[CRUX_INFO_CUR_COST] = sm[Lt::iCUR]; e[CRUX_INFO_COST_LOSS] = sm[Lt::iCLOSS]; e[CRUX_INFO_P_LOSS] = sm[Lt::iPLOSS]; } }
    epochs_run += 1;
    if (target_kl >= 0.f && KIND != MFK_VALUE && inf_kl > target_kl) stop = true;   // training.jl:49
    if (max_batches > 0 && total_batches >= max_batches) stop = true;               // :50
  };

  // ---- the W2 path of a step, the same in both roles: every wave owns WT tiles (theta, m, v in registers; D layout: reg r of tile mm <-> W2[o = 16 mp0 + 4g + r][i = 16 (m0+mm) + c])
  f32x4 tW2[WT], mW2[WT], vW2[WT];
#pragma unroll
  for (int mm = 0; mm < WT; ++mm)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int pc = Lt::cW2 + (16 * mp0 + 4 * g + r) + H2 * (16 * (m0 + mm) + c);
      tW2[mm][r] = a.p[pc]; mW2[mm][r] = a.m[pc]; vW2[mm][r] = a.v[pc]; }
  // dW2 of the own tiles over the workgroup's samples, sent to the exchange slot at once (the acknowledgement hides under what follows)
  auto dw2_send = [&](f32x4 (&gW2)[WT]) {
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
    float* mine = a.xbuf + (size_t)(((int)(xstep & 1u) * NWG + p)) * XSLOT;
#pragma unroll
    for (int mm = 0; mm < WT; ++mm) *(f32x4*)&mine[Lt::xW2 + tid * (4 * WT) + 4 * mm] = gW2[mm];
  };
  // the three peers' partials of the own tiles: issue the loads (phase 1 is complete: they are in the L2) ...
  auto w2_loads = [&](f32x4 (&pw)[NWG - 1][WT]) {
#pragma unroll
    for (int j = 0; j < NWG - 1; ++j) { const int q = j == 0 ? (p ^ 1) : ((p ^ 2) & 2) + (j - 1);      // partner, then the other pair's first and second workgroup
      const float* peer = a.xbuf + (size_t)(((int)(xstep & 1u) * NWG + q)) * XSLOT;
#pragma unroll
      for (int mm = 0; mm < WT; ++mm)
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(pw[j][mm]) : "v"(peer + Lt::xW2 + tid * (4 * WT) + 4 * mm) : "memory"); }
  };
  // ... and, once they are in (the caller has waited), the total (s0 + s1) + (s2 + s3) -- own + partner, the other pair in index order, then the two pair sums: the same bits
  // in all four workgroups
  auto w2_total = [&](f32x4 (&gW2)[WT], f32x4 (&pw)[NWG - 1][WT]) {
#pragma unroll
    for (int j = 0; j < NWG - 1; ++j)
#pragma unroll
      for (int mm = 0; mm < WT; ++mm) asm volatile("" : "+v"(pw[j][mm]));      // (asm statements keep their order: every use of a loaded value follows the wait)
#pragma unroll
    for (int mm = 0; mm < WT; ++mm) gW2[mm] = (gW2[mm] + pw[0][mm]) + (pw[1][mm] + pw[2][mm]);
  };
  auto w2_check = [&](const f32x4 (&gW2)[WT], bool want_ssq, float& ssq, bool& bad) {      // the suspect test over the totals and this wave's share of the gradient norm
#pragma unroll
    for (int mm = 0; mm < WT; ++mm) {
#pragma unroll
      for (int r = 0; r < 4; ++r) bad = bad || isnan(gW2[mm][r]);
      if (want_ssq) {
#pragma unroll
        for (int r = 0; r < 4; ++r) ssq += gW2[mm][r] * gW2[mm][r]; } }
  };
  auto adam_w2 = [&](const f32x4 (&gT)[WT]) {      // Flux.update! on the owned tiles; the new weights go to both LDS layouts of W2
#pragma unroll
    for (int mm = 0; mm < WT; ++mm) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { float m_ = mW2[mm][r], v_ = vW2[mm][r]; const float d = adam1(gT[mm][r], m_, v_, ak);
        mW2[mm][r] = m_; vW2[mm][r] = v_; tW2[mm][r] -= d;
        sm[Lt::oW2R + (16 * mp0 + 4 * g + r) * FS_LD + 16 * (m0 + mm) + c] = tW2[mm][r]; }
      *(f32x4*)&sm[Lt::oW2C + (16 * (m0 + mm) + c) * FS_LD + 16 * mp0 + 4 * g] = tW2[mm]; } };
  // the small parameters (everything but W2), shared by all 512 threads
  int so_part[NSC], so_master[NSC]; bool so_ok[NSC], so_ex[NSC];
#pragma unroll
  for (int k = 0; k < NSC; ++k) { const int s = tid + NT * k; so_ok[k] = s < ns_valid; so_ex[k] = s >= Lt::sEX;
    so_part[k] = so_ok[k] ? s_part(s) : 0; so_master[k] = so_ok[k] ? s_master(s) : 0; }
  const bool stat_lane = tid >= NT - 8 && tid < (LAG ? NT : NT - 1);      // stat sums, by 7 lanes of the last wave (an eighth: the cost term of lagrange_ppo_loss)
  // ---- replica group (comm.hip "peer"): mean over the group of NSEC payload sections (the W2-tile registers + the small parameters'
  // registers of every thread) and one statistics word, the same bits on every workgroup of every rank. Per-step form: ONE section, the minibatch gradient and its statistics;
  // periodic form (PXK): THREE sections -- theta, m, v after every k-th Adam step -- in one exchange. All four workgroups hold the same local values and share the writes (peer i
  // of the N-1 goes to workgroup i mod 4): the sections go into slot [parity][my rank] of the peer's region, one lane issues the system-scope release and raises flag[my rank]
  // there; every workgroup then waits for the N-1 flags in the OWN region and adds the N contributions in rank order. Called by all 512 threads (two workgroup barriers).
  // 0 = done, 1 = a replica did not answer or this learner had already failed (fERR is set), 2 = skipped: the step is suspect (a NaN total: the launch ends with CRUX_ENAN).
  float* const px_mine = PX ? a.px_tab[a.px_rank] : nullptr;
  const unsigned long long px0 = PX ? *(const unsigned long long*)(px_mine + CRUX_PX_COUNT) : 0ull;
  if (PX && tid == 0) px_launch_begin(px_mine, p);      // the launch's wait budget starts from zero (peer_wait.h, bound 2)
  const float px_inv = PX ? 1.0f / (float)a.px_n : 1.0f;
  int pxc = 0;
  auto px_allreduce_mean = [&](auto nsec_c, f32x4* const (&Wp)[3], float* const (&Sp)[3], float& xT, const bool failed, const unsigned tag, const bool check_suspect) -> int {
    constexpr int NSEC = decltype(nsec_c)::value;
    const unsigned long long xg = px0 + (unsigned long long)pxc;       // number of this exchange on this learner stream
    const int par = (int)(xg & 1ull);
    if (!failed) { int pi_ = 0;
      for (int r = 0; r < a.px_n; ++r) {
        if (r == a.px_rank || (pi_++ % NWG) != p) continue;
        float* dst0 = a.px_tab[r] + (size_t)(par * CRUX_PX_MAXR + a.px_rank) * CRUX_PX_SLOT;
#pragma unroll
        for (int sec = 0; sec < NSEC; ++sec) { float* dst = dst0 + sec * CRUX_PX_SEC;
#pragma unroll
          for (int mm = 0; mm < WT; ++mm) *(f32x4*)&dst[tid * (4 * WT) + 4 * mm] = Wp[sec][mm];
#pragma unroll
          for (int k = 0; k < NSC; ++k) dst[W2N + tid + NT * k] = Sp[sec][k]; }
        if (stat_lane) dst0[W2N + NSC * NT + (tid - (NT - 8))] = xT; } }
    // release, the hand-off recipe of the CDNA guides: every wave drains its own slot stores, the workgroup meets, ONE lane issues the system-scope release and drains it
    // before the flags go out
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const bool dead = fs2_flag_get(sm + Lt::fERR) != 0u, sus = check_suspect && fs2_flag_get(sm + Lt::fSUS) == tag;
      bool ok = !dead && !sus;
      if (ok) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int pi_ = 0;
        for (int r = 0; r < a.px_n; ++r) { if (r == a.px_rank) continue;
          if ((pi_++ % NWG) != p) continue;
          __hip_atomic_store((unsigned long long*)(a.px_tab[r] + CRUX_PX_FLAGS) + 8 * a.px_rank, xg + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
        const long long t0 = wall_clock64();                // 100 MHz; the wait is bounded four ways (peer_wait.h): a peer that is absent, slow, gone, or a host that calls the launch off
        const unsigned gave_up = px_wait_peers(px_mine, a.px_n, a.px_rank, xg + 1ull, t0, a.px_timeout, p, true);
        if (gave_up) { ok = false; fs2_flag_set(sm + Lt::fERR, 16u + gave_up); }      // 16 + bound: the replica group ended this launch (train.hip names the bound)
        if (a.px_hist) {      // how long this workgroup waited for the slowest peer's flag (10 ns ticks, log2 bins; workgroups 0 and 1 report)
          const unsigned long long dtk = (unsigned long long)(wall_clock64() - t0) | 1ull;
          if (p < 2) { unsigned* hb = (unsigned*)(px_mine + CRUX_PX_HIST) + 32 * p + (63 - __builtin_clzll(dtk) > 31 ? 31 : 63 - __builtin_clzll(dtk)); *hb = *hb + 1u; } }
      }
      if (!ok) {      // this learner leaves the group (failure, or a NaN step): the peers must not wait for it
        { const unsigned e = fs2_flag_get(sm + Lt::fERR); px_raise_abort(a.px_tab, a.px_n, sus && !dead ? 5u : (e >= 16u ? e - 16u : 1u)); }      // the bound that fired (5: left on a NaN step)
        if (!sus || dead) __hip_atomic_store(a.xctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");                       // system scope, one lane: drops this compute unit's L1 (the slot loads below bypass it anyway: sc0 sc1)
    }
    __syncthreads();
    if (fs2_flag_get(sm + Lt::fERR) != 0u) return 1;
    if (check_suspect && fs2_flag_get(sm + Lt::fSUS) == tag) return 2;
    // the slots are read one rank at a time (all loads of a rank in flight together) and added in rank order; the own contribution comes from the registers
    f32x4 oW[NSEC][WT]; float oS[NSEC][NSC]; const float oT = xT;
#pragma unroll
    for (int sec = 0; sec < NSEC; ++sec) {
#pragma unroll
      for (int mm = 0; mm < WT; ++mm) oW[sec][mm] = Wp[sec][mm];
#pragma unroll
      for (int k = 0; k < NSC; ++k) oS[sec][k] = Sp[sec][k]; }
    for (int r = 0; r < a.px_n; ++r) {
      f32x4 vW[NSEC][WT]; float vS[NSEC][NSC]; float vT = 0.f;
      if (r != a.px_rank) {
        const float* src0 = px_mine + (size_t)(par * CRUX_PX_MAXR + r) * CRUX_PX_SLOT;
        if (stat_lane) vT = __hip_atomic_load(src0 + W2N + NSC * NT + (tid - (NT - 8)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
        for (int sec = 0; sec < NSEC; ++sec) { const float* src = src0 + sec * CRUX_PX_SEC;
#pragma unroll
          for (int k = 0; k < NSC; ++k) vS[sec][k] = __hip_atomic_load(src + W2N + tid + NT * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
          for (int mm = 0; mm < WT; ++mm)
            asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=&v"(vW[sec][mm]) : "v"(src + tid * (4 * WT) + 4 * mm) : "memory"); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int sec = 0; sec < NSEC; ++sec)
#pragma unroll
          for (int mm = 0; mm < WT; ++mm) asm volatile("" : "+v"(vW[sec][mm]));
      } else {      // the own contribution (registers: nothing orders another workgroup's read after a store of this one, so it is never read back)
        vT = oT;
#pragma unroll
        for (int sec = 0; sec < NSEC; ++sec) {
#pragma unroll
          for (int mm = 0; mm < WT; ++mm) vW[sec][mm] = oW[sec][mm];
#pragma unroll
          for (int k = 0; k < NSC; ++k) vS[sec][k] = oS[sec][k]; } }
      if (r == 0) {
#pragma unroll
        for (int sec = 0; sec < NSEC; ++sec) {
#pragma unroll
          for (int mm = 0; mm < WT; ++mm) Wp[sec][mm] = vW[sec][mm];
#pragma unroll
          for (int k = 0; k < NSC; ++k) Sp[sec][k] = vS[sec][k]; }
        xT = vT;
      } else {
#pragma unroll
        for (int sec = 0; sec < NSEC; ++sec) {
#pragma unroll
          for (int mm = 0; mm < WT; ++mm) Wp[sec][mm] += vW[sec][mm];
#pragma unroll
          for (int k = 0; k < NSC; ++k) Sp[sec][k] += vS[sec][k]; }
        xT += vT;
      }
    }
    // mean over the group (gradient: global minibatch = px_n x nb samples, every rank's partial was already divided by nb)
#pragma unroll
    for (int sec = 0; sec < NSEC; ++sec) {
#pragma unroll
      for (int mm = 0; mm < WT; ++mm) Wp[sec][mm] = Wp[sec][mm] * px_inv;
#pragma unroll
      for (int k = 0; k < NSC; ++k) Sp[sec][k] = Sp[sec][k] * px_inv; }
    xT = xT * px_inv;
    pxc += 1;
    return 0;
  };
  // ---- the end of a step in every wave, once the workgroup's small partials are in LDS (B_2): the small partials leave as granules; the peers' W2 partials (phase 1 complete)
  // and granules come in; totals, Adam, B_b; suspect steps. false = the launch ends here (err is set).
  int st_now = 0;      // the minibatch loops set it: first row of the current minibatch
  auto step_tail = [&](const unsigned tag, f32x4 (&gW2)[WT], int& any_bad) -> bool {
    float gs[NSC];
#pragma unroll
    for (int k = 0; k < NSC; ++k) { float gsum = 0.f;
      if (so_ok[k]) { const int po = Lt::oPART + so_part[k];
        gsum = sm[po];
#pragma unroll
        for (int q = 1; q < TILES; ++q) gsum += sm[po + q * Lt::PART]; }
      gs[k] = gsum; }
    float stat_loc = 0.f;
    if (stat_lane) { const int k_ = tid - (NT - 8); const int ko = (LAG && k_ == 7) ? Lt::pMISC + 7 + OUT + (KIND == MFK_GAUSSIAN ? OUT : 0) : Lt::pST + k_; stat_loc = sm[Lt::oPART + ko];
#pragma unroll
      for (int q = 1; q < TILES; ++q) stat_loc += sm[Lt::oPART + q * Lt::PART + ko]; }
    // the small partials travel as granules: {value, step} in ONE naturally aligned 8-byte store -- the tag is the flag
    float* mine = a.xbuf + (size_t)(((int)(xstep & 1u) * NWG + p)) * XSLOT;
    // (a PLAIN store: it writes through the L1 and KEEPS the line in this XCD's L2, where the peers' sc1 loads find it; sc1 / volatile stores would drop it to the fabric)
    auto gran_put = [&](int gi, float v) { typedef unsigned u32x2 __attribute__((ext_vector_type(2))); u32x2 gq; gq.x = __builtin_bit_cast(unsigned, v); gq.y = tag;
      *(u32x2*)(mine + Lt::xGR + 2 * gi) = gq; };
#pragma unroll
    for (int k = 0; k < NSC; ++k) gran_put(tid + NT * k, gs[k]);
    if (stat_lane) gran_put(NSC * NT + (tid - (NT - 8)), stat_loc);
    FS2_T(9);
    fs2_flag_wait(sm + Lt::fP1, tag);           // phase 1 is complete (or has failed): every workgroup's W2 partials are in the L2
    FS2_T(10);
    bool failed = fs2_flag_get(sm + Lt::fERR) != 0u;
    f32x4 pw[NWG - 1][WT];
    if (!failed) w2_loads(pw);
    float ssq = 0.f; bool bad_tot = false;
    const bool want_ssq = static_report(st_now) || (KIND != MFK_VALUE && target_kl >= 0.f);      // only a step that may report needs the gradient norm
    f32x4 tW2o[BK_LDS ? 1 : WT], mW2o[BK_LDS ? 1 : WT], vW2o[BK_LDS ? 1 : WT];      // the state before this step's update: a suspect step (known after B_b) is undone first
    float* const bk = sm + Lt::oBK + 4 * tid;      // (LDS form: [theta | m | v][tile][thread] x 16 B)
#pragma unroll
    for (int mm = 0; mm < WT; ++mm) {
      if constexpr (BK_LDS) { *(f32x4*)&bk[4 * NT * mm] = tW2[mm]; *(f32x4*)&bk[4 * NT * (WT + mm)] = mW2[mm]; *(f32x4*)&bk[4 * NT * (2 * WT + mm)] = vW2[mm]; }
      else { tW2o[mm] = tW2[mm]; mW2o[mm] = mW2[mm]; vW2o[mm] = vW2[mm]; } }
    // the W2 update BEFORE the peers' small granules are asked for: those left a moment ago and need an L2 hop (polls that come back stale only keep the load path busy),
    // the W2 partials (phase 1 complete) are in the L2 already -- same box: C2 5.40 -> 5.396, C5 7.61 -> 7.50 us per step. (The per-step replica-group form needs both totals
    // before its exchange.)
    constexpr bool W2_FIRST = !(PX && !PXK);
    if constexpr (W2_FIRST) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (!failed) { w2_total(gW2, pw); w2_check(gW2, want_ssq, ssq, bad_tot); adam_w2(gW2); }      // (every compute wave is past B_2: nobody reads the W2 masters any more)
    }
    // poll the peers' granules: the same loads again until every tag is this step's
    constexpr int NLD = NWG - 1;
    float pg[NLD][NSC]; float ps[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) { ps[j] = 0.f;
#pragma unroll
      for (int k = 0; k < NSC; ++k) pg[j][k] = 0.f; }
    if (!failed) {
      unsigned spins = 0;
      for (;;) {
        unsigned long long gv[NLD][NSC], gst[NLD];
#pragma unroll
        for (int j = 0; j < NLD; ++j) { const int q = j == 0 ? (p ^ 1) : ((p ^ 2) & 2) + (j - 1);
          const float* peer = a.xbuf + (size_t)(((int)(xstep & 1u) * NWG + q)) * XSLOT + Lt::xGR;
#pragma unroll
          for (int k = 0; k < NSC; ++k) gv[j][k] = __hip_atomic_load((const unsigned long long*)(peer + 2 * (tid + NT * k)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          gst[j] = ((unsigned long long)tag << 32);
          if (stat_lane) gst[j] = __hip_atomic_load((const unsigned long long*)(peer + 2 * (NSC * NT + (tid - (NT - 8)))), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        bool all_in = true;
#pragma unroll
        for (int j = 0; j < NLD; ++j) { all_in = all_in && (unsigned)(gst[j] >> 32) == tag; ps[j] = __builtin_bit_cast(float, (unsigned)gst[j]);
#pragma unroll
          for (int k = 0; k < NSC; ++k) { all_in = all_in && (unsigned)(gv[j][k] >> 32) == tag; pg[j][k] = __builtin_bit_cast(float, (unsigned)gv[j][k]); } }
        if (__all(all_in ? 1 : 0)) break;
        if ((++spins & 255u) == 0u && (spins > (1u << 22) || __hip_atomic_load(a.xctr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {      // never hang the GPU
          __hip_atomic_store(a.xctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (fs2_flag_get(sm + Lt::fERR) == 0u) fs2_flag_set(sm + Lt::fERR, 1u); failed = true; break; }
      }
    }
    FS2_T(11);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FS2_T(12);
    float th_o[NSC], m_o[NSC], v_o[NSC];
    float stat_tot = stat_loc;
    if constexpr (PX && !PXK) {
      if (!failed) {
        w2_total(gW2, pw);
#pragma unroll
        for (int k = 0; k < NSC; ++k) gs[k] = (gs[k] + pg[0][k]) + (pg[1][k] + pg[2][k]);
        stat_tot = (stat_loc + ps[0]) + (ps[1] + ps[2]);
      }
      // the group's all-reduce of the minibatch gradient and its statistics, between the pullback (training.jl:18) and Flux.update! (:21)
      f32x4* const Wp[3] = {gW2, nullptr, nullptr}; float* const Sp[3] = {gs, nullptr, nullptr};
      if (px_allreduce_mean(std::integral_constant<int, 1>{}, Wp, Sp, stat_tot, failed, tag, false) != 0) failed = true;
      if (!failed) {
        w2_check(gW2, want_ssq, ssq, bad_tot);
        adam_w2(gW2);                            // (every compute wave is past B_2: nobody reads the W2 masters any more)
#pragma unroll
        for (int k = 0; k < NSC; ++k) bad_tot = bad_tot || (so_ok[k] && isnan(gs[k]));
        if (bad_tot) fs2_flag_set(sm + Lt::fSUS, tag);
      }
    } else if (!failed) {
      if constexpr (!W2_FIRST) { w2_total(gW2, pw); w2_check(gW2, want_ssq, ssq, bad_tot);
        adam_w2(gW2); }                          // (every compute wave is past B_2: nobody reads the W2 masters any more)
#pragma unroll
      for (int k = 0; k < NSC; ++k) { gs[k] = (gs[k] + pg[0][k]) + (pg[1][k] + pg[2][k]); bad_tot = bad_tot || (so_ok[k] && isnan(gs[k])); }
      if (bad_tot) fs2_flag_set(sm + Lt::fSUS, tag);
      stat_tot = (stat_loc + ps[0]) + (ps[1] + ps[2]);
    }
    if (stat_lane) sm[Lt::oRED + 8 + (tid - (NT - 8))] = stat_tot;
#pragma unroll
    for (int k = 0; k < NSC; ++k) if (so_ok[k]) {
      if (KIND == MFK_GAUSSIAN && so_ex[k]) gs[k] += LAG ? -lambda_e / (1.f + pen) : -lambda_e;       // d(-lambda_e H)/dlogSigma, H = const + sum(logSigma); lagrange: the whole loss is divided by 1 + penalty
      if (want_ssq) ssq += gs[k] * gs[k]; }
    if (want_ssq) { ssq = wave_sum(ssq);
      if (lane == 0) sm[Lt::oRED + w] = ssq; }      // this wave's share of the gradient norm (the report is formed after B_b)
    FS2_T(13);
    // Adam on the small parameters (Flux.update!, training.jl:21); the values read here are what a suspect step is put back to
    auto adam_small = [&]() {
#pragma unroll
      for (int k = 0; k < NSC; ++k) { const int s = tid + NT * k;
        if (so_ok[k]) { float m_ = sm[Lt::oMS + s], v_ = sm[Lt::oVS + s]; const int mo = so_master[k]; const float th = sm[mo];
          m_o[k] = m_; v_o[k] = v_; th_o[k] = th;
          const float d = adam1(gs[k], m_, v_, ak);
          sm[Lt::oMS + s] = m_; sm[Lt::oVS + s] = v_; sm[mo] = th - d; } } };
    if (!failed) adam_small();
    if constexpr (PX && PXK) {
      // ---- periodic form (crux_peer_set_sync_every(k > 1)): between exchanges every replica takes LOCAL Adam steps on its own shard; after every k-th step the group averages
      // theta, m and v (sum in rank order x 1/N: the same bits everywhere, so the replicas leave the exchange identical). One exchange per k steps instead of k.
      if ((total_batches + 1) % a.px_every == 0) {
        float sT[NSC], sM[NSC], sV[NSC]; float dummy = 0.f;
#pragma unroll
        for (int k = 0; k < NSC; ++k) { const int s = tid + NT * k; const bool ok_ = so_ok[k] && !failed;
          sT[k] = ok_ ? sm[so_master[k]] : 0.f; sM[k] = ok_ ? sm[Lt::oMS + s] : 0.f; sV[k] = ok_ ? sm[Lt::oVS + s] : 0.f; }
        f32x4* const Wp[3] = {tW2, mW2, vW2}; float* const Sp[3] = {sT, sM, sV};
        const int rc = px_allreduce_mean(std::integral_constant<int, 3>{}, Wp, Sp, dummy, failed, tag, true);
        if (rc == 1) failed = true;
        if (rc == 0) {
#pragma unroll
          for (int k = 0; k < NSC; ++k) { const int s = tid + NT * k;
            if (so_ok[k]) { sm[so_master[k]] = sT[k]; sm[Lt::oMS + s] = sM[k]; sm[Lt::oVS + s] = sV[k]; } }
#pragma unroll
          for (int mm = 0; mm < WT; ++mm) {      // the LDS copies of W2 follow the averaged registers
#pragma unroll
            for (int r = 0; r < 4; ++r) sm[Lt::oW2R + (16 * mp0 + 4 * g + r) * FS_LD + 16 * (m0 + mm) + c] = tW2[mm][r];
            *(f32x4*)&sm[Lt::oW2C + (16 * (m0 + mm) + c) * FS_LD + 16 * mp0 + 4 * g] = tW2[mm]; } }
      }
    }
    FS2_T(14);
    __syncthreads();   // ---- B_b: masters updated; tiles and partials may be overwritten
    const unsigned why = fs2_flag_get(sm + Lt::fERR), sus = fs2_flag_get(sm + Lt::fSUS);      // (both reads in flight together: one LDS round trip)
    if (why != 0u) { err = CRUX_EHIP; why_failed = (int)why; return false; }
    any_bad = 0;
    if (sus == tag) {      // a thread of this workgroup found a NaN total: every thread back to its pre-step state (training.jl:20: error, no update)
#pragma unroll
      for (int k = 0; k < NSC; ++k) { const int s = tid + NT * k;
        if (so_ok[k]) { sm[Lt::oMS + s] = m_o[k]; sm[Lt::oVS + s] = v_o[k]; sm[so_master[k]] = th_o[k]; } }
#pragma unroll
      for (int mm = 0; mm < WT; ++mm) {
        if constexpr (BK_LDS) { tW2[mm] = *(const f32x4*)&bk[4 * NT * mm]; mW2[mm] = *(const f32x4*)&bk[4 * NT * (WT + mm)]; vW2[mm] = *(const f32x4*)&bk[4 * NT * (2 * WT + mm)]; }
        else { tW2[mm] = tW2o[mm]; mW2[mm] = mW2o[mm]; vW2[mm] = vW2o[mm]; } }
      any_bad = 1;
    }
    return true;
  };

  if (cw) {
    // =====================================================================================================================================================
    // COMPUTE WAVES
    // =====================================================================================================================================================
    for (int ep = 0; ep < n_epochs && !stop && !err; ++ep) {
      if (!epoch_prologue(ep)) break;
      staged = false;
      for (int st = 0; st < total_rows; st += bs) {
        const int nb = (int)((total_rows - st) < bs ? (total_rows - st) : bs);
        const float invB = 1.0f / (float)nb;
        ak.c1 = __builtin_amdgcn_rcpf((float)(1.0 - bp1)); ak.c2 = __builtin_amdgcn_rcpf((float)(1.0 - bp2));
        FS2_T(0);
        if (!staged) __syncthreads();      // the first minibatch of an epoch is staged by the helpers before this barrier; every other one during the previous step
        staged = false;
        if constexpr (LAG) lag_advance(xcur);
        const float* xs_c = xs + xcur * XSB; const float* sc_c = sc + xcur * SCB;
        FS2_T(1);
        // ======================= forward, C orientation: D[feature 16m+4g+r][sample c] =======================
        f32x4 h1[4]; f32x4 h2[MH];
        constexpr bool W3_REG = OUT <= 2;
        f32x4 w3[OUT <= 2 ? OUT : 1][MH];
        float zp[OUT];
        {
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
          for (int mm = 0; mm < 2; ++mm)
#pragma unroll
            for (int r = 0; r < 4; ++r) T1[t_wr + (16 * (2 * h + mm) + r) * 16] = (h ? (mm ? h1[3][r] : h1[2][r]) : (mm ? h1[1][r] : h1[0][r]));
          FS2_T(2);
          { f32x4 acc[MH];
#pragma unroll
            for (int mm = 0; mm < MH; ++mm) acc[mm] = *(const f32x4*)&sm[Lt::oB2 + HH * h + 16 * mm + 4 * g];
#pragma unroll
            for (int m = 0; m < (CRUX_FS2_EXP ? 2 : 4); ++m) { f32x4 wv[MH];
#pragma unroll
              for (int mm = 0; mm < MH; ++mm) wv[mm] = *(const f32x4*)&sm[Lt::oW2R + (HH * h + 16 * mm + c) * FS_LD + 16 * m + 4 * g];
#pragma unroll
              for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int mm = 0; mm < MH; ++mm) acc[mm] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[mm][r], h1[m][r], acc[mm], 0, 0, 0); }
#pragma unroll
            for (int mm = 0; mm < MH; ++mm) {
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[mm][r] = actf<ACT2>(acc[mm][r]);
              h2[mm] = acc[mm]; } }
          FS2_T(3);
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
          { float* zq = sm + Lt::oZP + ((t * 2 + h) * 64 + lane) * Lt::ZW;
#pragma unroll
            for (int o = 0; o < OUT; ++o) zq[o] = zp[o]; }
        }
        fs2_group_barrier(sm + Lt::cPAIR + t, 2u * (xstep + 1u), lane);      // ---- B_z: the pair's partial logits are visible (the two waves of the tile only)
        float dz[OUT], dex[OUT];
        float s_lossp = 0.f, s_H = 0.f, s_kl = 0.f, s_adv = 0.f, s_ret = 0.f, s_clip = 0.f, s_sq = 0.f, s_cost = 0.f;
        float ent_pre = 1.4189385332046727f;
        {
          float z[OUT];
          { const float* zo = sm + Lt::oZP + ((t * 2 + (1 - h)) * 64 + lane) * Lt::ZW;
#pragma unroll
            for (int o = 0; o < OUT; ++o) { const float other = zo[o]; z[o] = (h ? other + zp[o] : zp[o] + other) + sm[Lt::oB3 + o]; } }
          // ======================= loss head (identical in both waves of the pair) =======================
          {
            const float* q = sc_c + c * Lt::SCW;
            const bool valid = q[0] != 0.f; const float oldlp = q[1], A = q[2], R = q[3];
            const float cnt = (valid && g == 0) ? 1.f : 0.f;
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
              const float coef = a2c ? A : gsel * r, lterm = a2c ? newlp * A : fminf(u, cl), clipv = (!a2c && (r > hi || r < lo)) ? 1.f : 0.f;
              float gcr = 0.f;                                                          // lagrange: d/dr of max(r Ac, clamp(r) Ac) times r (ppo.jl:119)
              if constexpr (LAG) { const float Ac = q[Lt::SCW - 1]; const float uc = r * Ac, clc = rc * Ac; s_cost = cnt * (uc >= clc ? uc : clc); gcr = (uc >= clc ? Ac : 0.f) * r; }
#pragma unroll
              for (int k = 0; k < OUT; ++k) { const float dlogpi = ((k == ai) ? 1.f : 0.f) - pk[k];
                const float base = -lambda_p * coef * dlogpi - lambda_e * (pk[k] * (hk[k] - hp));
                dz[k] = !valid ? 0.f : (LAG ? invB * ((base + pen * gcr * dlogpi) / (1.f + pen)) : invB * base); }
              s_lossp = cnt * lterm; s_H = cnt * H; s_kl = cnt * (oldlp - newlp); s_adv = cnt * A; s_ret = cnt * R;
              s_clip = cnt * clipv;
            } else {   // gaussian with constant log-std (policies.jl:333-348)
              float newlp = 0.f; float dd[OUT], s2[OUT];
              float inr[OUT];
#pragma unroll
              for (int k = 0; k < OUT; ++k) { const float ls = sm[Lt::oEX + k]; const bool sq = squash > 0.f;
                s2[k] = __expf(-2.f * (sq ? sq_clampls(ls) : ls)); dd[k] = q[4 + k] - z[k];
                inr[k] = (sq && !(ls >= -5.f && ls <= 2.f)) ? 0.f : 1.f;
                newlp += (-(dd[k] * dd[k]) * (0.5f * s2[k]) - 0.9189385332046727f - ls); }
              if (squash > 0.f) newlp -= q[4 + NACT];
              const float r = __expf(newlp - oldlp); const float u = r * A, rc = fminf(fmaxf(r, lo), hi), cl = rc * A; const float gsel = (u <= cl) ? A : 0.f;
              const float coef = a2c ? A : gsel * r, lterm = a2c ? newlp * A : fminf(u, cl), clipv = (!a2c && (r > hi || r < lo)) ? 1.f : 0.f;
              float cf = -lambda_p * coef;
              if constexpr (LAG) { const float Ac = q[Lt::SCW - 1]; const float uc = r * Ac, clc = rc * Ac; s_cost = cnt * (uc >= clc ? uc : clc);
                cf = (cf + pen * ((uc >= clc ? Ac : 0.f) * r)) / (1.f + pen); }
#pragma unroll
              for (int k = 0; k < OUT; ++k) { dz[k] = valid ? invB * (cf * (dd[k] * s2[k])) : 0.f;
                dex[k] = valid ? invB * (cf * (((dd[k] * dd[k]) * s2[k]) * inr[k] - 1.f)) : 0.f; }
              s_lossp = cnt * lterm; s_kl = cnt * (oldlp - newlp); s_adv = cnt * A; s_ret = cnt * R; s_clip = cnt * clipv;
            }
          }
          FS2_T(4);
          if (KIND == MFK_GAUSSIAN) {      // the entropy of the report, 1.4189385 + sum(logSigma), as the loss saw it (this step's Adam on logSigma comes after B_1)
#pragma unroll
            for (int k = 0; k < OUT; ++k) ent_pre += sm[Lt::oEX + k]; }
          // ======================= backward, own samples, own feature half =======================
          constexpr int FPL = 4 * MH, PER = 16 / FPL;
#pragma unroll
          for (int o2 = 0; o2 < (OUT + PER - 1) / PER; ++o2) { float pv[16];
#pragma unroll
            for (int oo = 0; oo < PER; ++oo)
#pragma unroll
              for (int mm = 0; mm < MH; ++mm)
#pragma unroll
                for (int r = 0; r < 4; ++r) pv[FPL * oo + 4 * mm + r] = (PER * o2 + oo < OUT) ? dz[(PER * o2 + oo < OUT) ? PER * o2 + oo : 0] * h2[mm][r] : 0.f;
            const float sred = row16_reduce_scatter(pv, c);
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
          if (h == 0) {
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
          { float* dq = sm + Lt::oD2X + ((t * 2 + h) * MH) * 256 + lane * 4;
#pragma unroll
            for (int mm = 0; mm < MH; ++mm) *(f32x4*)&dq[256 * mm] = h2[mm]; }
        }
        FS2_T(5);
        __syncthreads();   // ---- B_1: T1 / T2 tiles of the workgroup and the dZ2 halves are visible
        FS2_T(6);
        f32x4 gW2[WT];
        dw2_send(gW2);
        // dH1 (R) for the h1 features [32h, 32h + 32)
        f32x4 dz1r[2];
        { const float* dq = sm + Lt::oD2X + ((t * 2 + (1 - h)) * MH) * 256 + lane * 4;
          f32x4 av[2 * MH];
#pragma unroll
          for (int mm = 0; mm < MH; ++mm) { av[mm] = h2[mm]; av[MH + mm] = *(const f32x4*)&dq[256 * mm]; }
          f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int q = 0; q < (CRUX_FS2_EXP ? MH : 2 * MH); ++q) { const int fo = HH * ((q >= MH) ? 1 - h : h) + 16 * (q % MH);
            const f32x4 wv0 = *(const f32x4*)&sm[Lt::oW2C + (32 * h + c) * FS_LD + fo + 4 * g];
            const f32x4 wv1 = *(const f32x4*)&sm[Lt::oW2C + (32 * h + 16 + c) * FS_LD + fo + 4 * g];
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q][r], wv0[r], acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q][r], wv1[r], acc1, 0, 0, 0); } }
          dz1r[0] = acc0; dz1r[1] = acc1; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the dW2 stores left before dH1: acknowledged by now -- tell the helper leader (phase 1)
        if (lane == 0) (void)__hip_atomic_fetch_add((unsigned*)(sm + Lt::cACK), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        FS2_T(7);
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
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
          float xR[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) xR[r] = (16 * jt + c < IP) ? xs_c[(4 * g + r) * XP + 16 * jt + c] : 0.f;
#pragma unroll
          for (int mm = 0; mm < (CRUX_FS2_EXP ? 1 : 2); ++mm) { f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dz1r[mm][r], xR[r], acc, 0, 0, 0);
            if (16 * jt + c < Lt::W1ROWS) *(f32x4*)&part[Lt::pW1 + (16 * jt + c) * FS_LD + 32 * h + 16 * mm + 4 * g] = acc; }
        }
        FS2_T(8);
        fs2_group_barrier(sm + Lt::cCOMP, (unsigned)NWC * (xstep + 1u), lane);   // ---- B_2: the small partial gradients of both tiles are visible (compute waves only)
        const unsigned tag = xstep + 1u;
        int any_bad = 0; st_now = st;
        if (!step_tail(tag, gW2, any_bad)) break;
        // minibatch info (training.jl:22-23, ppo.jl:13-19): identical in every compute thread; only the epoch's last minibatch (or the one that stops the loop) is ever reported
        if (any_bad || static_report(st) || (KIND != MFK_VALUE && target_kl >= 0.f)) {
          const float* tq = sm + Lt::oRED + 8;
          const bool report = any_bad || static_report(st) || (KIND != MFK_VALUE && target_kl >= 0.f && tq[2] * invB > target_kl);
          if (report) {
            float ss = sm[Lt::oRED];
#pragma unroll
            for (int q = 1; q < NW; ++q) ss += sm[Lt::oRED + q];
            if (KIND != MFK_VALUE) inf_kl = tq[2] * invB;
            if (tid == 0) { sm[Lt::iGN] = sqrtf(ss);
              if (KIND == MFK_VALUE) { sm[Lt::iLOSS] = tq[6] * invB; sm[Lt::iRET] = tq[4] * invB; }
              else { const float p_loss = -(tq[0] * invB); const float entropy = KIND == MFK_CATEGORICAL ? tq[1] * invB : ent_pre;
                sm[Lt::iENT] = entropy; sm[Lt::iLOSS] = fmaf(lambda_p, p_loss, lambda_e * (-entropy)); sm[Lt::iADV] = tq[3] * invB; sm[Lt::iRET] = tq[4] * invB; sm[Lt::iCLIP] = tq[5] * invB;
                if constexpr (LAG) { const float cost_loss = pen * (tq[7] * invB);                                        // ppo.jl:119
                  sm[Lt::iLOSS] = ((lambda_p * p_loss + lambda_e * (-entropy)) + cost_loss) / (1.f + pen);                  // :131   (iPEN / iCUR are wave 0's controller words: current)
                  sm[Lt::iCLOSS] = cost_loss; sm[Lt::iPLOSS] = lambda_p * p_loss; } } } } }
        const bool go = step_exit(invB, any_bad != 0);
        if (!any_bad) { bp1 *= db1; bp2 *= db2; }
        xcur ^= 1; xstep += 1u; staged = st + bs < total_rows;
        FS2_T(15);
        if (!go) break;
      }
      if (err) break;
      epoch_epilogue(ep);
    }
  } else {
    // =====================================================================================================================================================
    // HELPER WAVES: the whole W2 path of the step (dW2, its exchange, total, Adam) and the minibatch prefetch / staging
    // =====================================================================================================================================================
    // ---- minibatch prefetch (HBM/L2 -> registers) and staging (registers -> the tile's LDS rows): wave h = 0 of a tile role fetches and stages the observation rows (four
    // lanes per sample), wave h = 1 the scalars (logprob, advantage, return, action)
    constexpr int NXL = (IN + 3) / 4;
    float px[NXL]; float p_lp = 0.f, p_adv = 0.f, p_ret = 0.f; float p_act[NACT]; int p_valid = 0; uint8_t p_abyte[OUT];
#pragma unroll
    for (int k = 0; k < OUT; ++k) p_abyte[k] = 0;
#pragma unroll
    for (int k = 0; k < NACT; ++k) p_act[k] = 0.f;
#pragma unroll
    for (int e = 0; e < NXL; ++e) px[e] = 0.f;
    int n_row = 0, n_valid = 0;
    int n_row2 = -1; float p_cost2 = 0.f, p_ee2 = 0.f, p_cadv = 0.f;      // LAG: helper thread ct < 128 carries row ct of the WHOLE minibatch (its :cost and :episode_end); :cost_advantage of the own sample
    auto fetch_index = [&](const int32_t* ord, int st, int nb) {
      const int sidx = 8 * NWC * p + 16 * t + c;
      n_valid = sidx < nb ? 1 : 0;
      n_row = n_valid ? CRUX_GLOBAL_PTR(int32_t, ord)[st + sidx] : 0;
      if constexpr (LAG) n_row2 = ct < nb ? CRUX_GLOBAL_PTR(int32_t, ord)[st + ct] : -1;
    };
    auto fetch_data = [&]() {
      const int rowlo = n_row; p_valid = n_valid; const int64_t row = rowlo;
      if constexpr (LAG) { p_cost2 = n_row2 >= 0 ? CRUX_GLOBAL_PTR(float, a.COST)[n_row2] : 0.f; p_ee2 = (n_row2 >= 0 && CRUX_GLOBAL_PTR(uint8_t, a.EE)[n_row2]) ? 1.f : 0.f; }
      if (h == 0) {
        const int rs = __shfl(rowlo, lane >> 2, 64), vs = __shfl(p_valid, lane >> 2, 64);
        const float* xrow = a.PACK ? CRUX_GLOBAL_PTR(float, a.PACK) + (int64_t)rs * a.pack_stride + (lane & 3) * NXL : CRUX_GLOBAL_PTR(float, a.S) + (int64_t)rs * IN + (lane & 3) * NXL;
#pragma unroll
        for (int e = 0; e < NXL; ++e) px[e] = ((lane & 3) * NXL + e < IN && vs) ? xrow[e] : 0.f;
      } else {
        p_lp = 0.f; p_adv = 0.f; p_ret = 0.f; p_cadv = 0.f;
#pragma unroll
        for (int k = 0; k < NACT; ++k) p_act[k] = 0.f;
        if (a.PACK && !LAG) {
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
      if constexpr (LAG) { if (ct < 128) { sm[Lt::oLAG + 256 * buf + ct] = p_cost2; sm[Lt::oLAG + 256 * buf + 128 + ct] = p_ee2; } }
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
          if (KIND == MFK_GAUSSIAN) {
            static_assert(KIND != MFK_GAUSSIAN || ((4 + NACT) % 2 == 0), "the staging row needs its spare slot");
            float corr = 0.f;
            if (squash > 0.f) {
#pragma unroll
              for (int k = 0; k < OUT; ++k) { const float u = p_valid ? sq_untanh(p_act[k], squash) : 0.f; corr += p_valid ? sq_corr(u) : 0.f; p_act[k] = u; } }
            q[4 + NACT] = corr; }
#pragma unroll
          for (int k = 0; k < NACT; ++k) q[4 + k] = p_act[k]; }
      }
    };

    for (int ep = 0; ep < n_epochs && !stop && !err; ++ep) {
      if (!epoch_prologue(ep)) break;
      staged = false;
      { const int nb0 = (int)(total_rows < bs ? total_rows : bs); fetch_index(order_cur, 0, nb0); fetch_data();
        const int st1 = bs; const int nb1 = st1 < total_rows ? (int)((total_rows - st1) < bs ? (total_rows - st1) : bs) : 0; fetch_index(order_cur, st1 < total_rows ? st1 : 0, nb1); }
      for (int st = 0; st < total_rows; st += bs) {
        const int nb = (int)((total_rows - st) < bs ? (total_rows - st) : bs);
        const float invB = 1.0f / (float)nb;
        ak.c1 = __builtin_amdgcn_rcpf((float)(1.0 - bp1)); ak.c2 = __builtin_amdgcn_rcpf((float)(1.0 - bp2));
        FS2_T(0);
        if (!staged) { stage(xcur); __syncthreads(); }      // the first minibatch of an epoch
        staged = false;
        if constexpr (LAG) lag_advance(xcur);              // (the helpers' small-parameter Adam divides the entropy term by 1 + penalty)
        // rows of the NEXT minibatch (their indices came a step ago) and the indices of the one after it; the next minibatch goes into the other staging buffer, which the
        // compute waves left at the end of the previous step
        if (st + bs < total_rows) fetch_data();
        { const int st2 = st + 2 * bs; const int nb2 = st2 < total_rows ? (int)((total_rows - st2) < bs ? (total_rows - st2) : bs) : 0;
          fetch_index(order_cur, st2 < total_rows ? st2 : 0, nb2); }
        FS2_T(1);
        if (st + bs < total_rows) stage(xcur ^ 1);
        FS2_T(2);
#if CRUX_FS2_EXP >= 2
        { f32x4 dacc = {0.f, 0.f, 0.f, 0.f}; float da = (float)lane, db = (float)w;      // the forward of a second compute pair on this SIMD: L1 (4) + a quarter of L2 (16) MFMAs, logits + head VALU
#pragma unroll
          for (int q = 0; q < 20; ++q) { dacc = __builtin_amdgcn_mfma_f32_16x16x4f32(da, db, dacc, 0, 0, 0); asm volatile("" : "+v"(dacc)); }
#pragma unroll
          for (int q = 0; q < (CRUX_FS2_EXP == 2 ? 120 : 0); ++q) { da = fmaf(da, 1.0001f, db); asm volatile("" : "+v"(da)); }
          if (da == 123.456f) sm[Lt::oRED + 28] = dacc[0]; }
#endif
        __syncthreads();   // ---- B_1: T1 / T2 tiles of the workgroup are visible
        FS2_T(3);
        // ======================= dW2 of this wave's tiles, then phase 1 of the exchange: the hand-shake for the W2 partials of ALL eight waves =======================
        const unsigned tag = xstep + 1u;
        f32x4 gW2[WT];
        dw2_send(gW2);
#if CRUX_FS2_EXP >= 2
        { f32x4 dacc = {0.f, 0.f, 0.f, 0.f}; float da = (float)lane, db = (float)w;      // dH1 (16) + dW1 (4) MFMAs and the dZ1 / bias VALU of the second compute pair
#pragma unroll
          for (int q = 0; q < 20; ++q) { dacc = __builtin_amdgcn_mfma_f32_16x16x4f32(da, db, dacc, 0, 0, 0); asm volatile("" : "+v"(dacc)); }
#pragma unroll
          for (int q = 0; q < (CRUX_FS2_EXP == 2 ? 60 : 0); ++q) { da = fmaf(da, 1.0001f, db); asm volatile("" : "+v"(da)); }
          if (da == 123.456f) sm[Lt::oRED + 28] = dacc[0]; }
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every store of this lane has reached the L2 (and the prefetched rows are in)
        if (lane == 0) (void)__hip_atomic_fetch_add((unsigned*)(sm + Lt::cACK), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        FS2_T(4);
        if (ct == 0) {
          fs2_flag_wait(sm + Lt::cACK, (unsigned)NW * tag);        // all eight waves' W2 stores are acknowledged (the compute waves report after dH1)
          (void)__hip_atomic_fetch_add(a.xctr + 4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ARRIVAL 1
          const unsigned want = (unsigned)NWG * tag; unsigned spins = 0; bool ok = true;
          while (__hip_atomic_load(a.xctr + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) { __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 24) || __hip_atomic_load(a.xctr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = false; break; } }   // never hang the GPU
          unsigned why = ok ? 0u : 1u;                                // 1: a workgroup never arrived (or raised the abort word)
          if (ok && xstep == 0u) {   // the unfenced exchange is only coherent inside one XCD's L2
            uint32_t xcc = 0; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); xcc &= 0xf;
            for (int q = 0; q < NWG; ++q) { const unsigned peer_xcc = __hip_atomic_load(a.xctr + 8 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (peer_xcc != xcc + 1u) { ok = false; why = 2u; } } }   // 2: the workgroups of this learner sit on different XCDs
          if (!ok) { __hip_atomic_store(a.xctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (fs2_flag_get(sm + Lt::fERR) == 0u) fs2_flag_set(sm + Lt::fERR, why); }
          fs2_flag_set(sm + Lt::fP1, tag);
        }
        FS2_T(5);
        fs2_flag_wait(sm + Lt::cCOMP, (unsigned)NWC * tag);      // the compute waves have passed B_2: the small partials of both tiles are in LDS, nobody reads the W2 masters (dH1) any more
        int any_bad = 0; st_now = st;
        if (!step_tail(tag, gW2, any_bad)) break;
        const bool go = step_exit(invB, any_bad != 0);
        if (!any_bad) { bp1 *= db1; bp2 *= db2; }
        xcur ^= 1; xstep += 1u; staged = st + bs < total_rows;
        FS2_T(15);
        if (!go) break;
      }
      if (err) break;
      epoch_epilogue(ep);
    }
  }
  // ---- write back parameters, Adam state and the launch status ------------------------------------------------------
  __syncthreads();
  if (p == 0) {
    // (the element addresses are formed again from a laundered thread index: shared with the loads at the top of the kernel, the compiler kept the 64-bit addresses alive
    //  across the whole launch -- in scratch, at the 256-register limit of the wide heads)
    int tid2 = tid; asm volatile("" : "+v"(tid2));
    const int lane2 = tid2 & 63, w2_ = tid2 >> 6, c2 = lane2 & 15, g2 = lane2 >> 4, mp2 = (w2_ * WT) >> 2, m2 = (w2_ * WT) & 3;
#pragma unroll
    for (int mm = 0; mm < WT; ++mm)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int pc = Lt::cW2 + (16 * mp2 + 4 * g2 + r) + H2 * (16 * (m2 + mm) + c2);
        a.p[pc] = tW2[mm][r]; a.m[pc] = mW2[mm][r]; a.v[pc] = vW2[mm][r]; } }
  if (p == 0) { for (int s = tid; s < ns_valid; s += NT) { const int pc = s_canon(s); a.p[pc] = sm[s_master(s)]; a.m[pc] = sm[Lt::oMS + s]; a.v[pc] = sm[Lt::oVS + s]; } }
  if (TIMING && lane == 0 && a.dbg) { for (int k = 0; k < 16; ++k) a.dbg[(NW * p + w) * 16 + k] = tacc[k]; }
  if (PX && tid == 0 && p == 0) *(unsigned long long*)(px_mine + CRUX_PX_COUNT) = px0 + (unsigned long long)pxc;
  if constexpr (PX && PXK) {      // periodic form: a replica that leaves on a NaN step between two exchanges will not show up at the next one -- its peers must not wait for the timeout
    if (tid == 0 && p == 0 && err == CRUX_ENAN) { for (int r = 0; r < a.px_n; ++r) if (r != a.px_rank) __hip_atomic_store((unsigned*)(a.px_tab[r] + CRUX_PX_ABORT), 5u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } }
  if (tid == 0 && (p == 0 || err)) {
    a.status[0] = err; a.status[1] = (int32_t)total_batches; a.status[2] = epochs_run; a.status[3] = (order_cur == a.order_a) ? 0 : 1;
    if (err == CRUX_EHIP) a.status[4] = why_failed;      // 1 a workgroup of the learner is missing, 2 workgroups on different XCDs, 3 replica group timeout / abort, 4 a workgroup missed the abort-latch consensus
    a.bp[0] = bp1; a.bp[1] = bp2;
    if constexpr (LAG) { if (p == 0) { a.lag->I = lgs[0]; a.lag->smooth_delta = lgs[1]; a.lag->smooth_Jc = lgs[2]; a.lag->Jc_prev = lgs[3]; a.lag->deriv_term = lgs[4]; a.lag->penalty = lgs[5]; a.lag->cur_cost = lgs[6]; } }
    if (err && a.epoch_infos && epochs_run == 0) { a.epoch_infos[CRUX_INFO_LOSS] = sm[Lt::iLOSS]; a.epoch_infos[CRUX_INFO_GRAD_NORM] = NAN; }
  }
#undef FS2_T
}
