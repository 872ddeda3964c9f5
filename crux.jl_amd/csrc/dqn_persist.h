// dqn_persist.h -- value_training epochs of the DQN family on IN -> 256 -> 256 -> OUT networks (config C3: DQN + prioritized replay, 8-256-256-4, B = 128) as TWO
// persistent kernels behind ONE XCD's L2, one launch each per crux_dqn_epochs call (src/model_free/off_policy.jl:66-111 with dqn_target, rl/dqn.jl:4-6; td_loss
// utils.jl:76-87; Flux Adam; prioritized_sample! / update_priorities! experience_buffer.jl:291-349). Included by exec.hip (the off-policy translation unit).
//
//   k_dqn_learn   16 workgroups = 16 compute units. Workgroup p OWNS the second hidden layer's features F_p = [16 p, 16 p + 16): the rows W2[F_p, :] with their Adam
//                 state (gradient tiles in registers in the MFMA D layout, theta in LDS, m and v in global memory, touched once per epoch), b2[F_p], the columns W3[:, F_p]; the first layer (IN <= 16 inputs) is
//                 evaluated -- and updated -- redundantly by every workgroup, so the forward pass of BOTH networks (online on s, target on s') up to its slice of H2
//                 needs no exchange at all: every wave owns one 16-sample tile from the input to H2. Three exchanges per epoch go through the shared L2 (plain
//                 stores, s_waitcnt, flag barrier -- the protocol of the on-policy learner kernels):
//                   (b) the partial logits z_p = W3[:, F_p] H2[F_p, :] of both networks (OUT x B floats each), summed in workgroup order by everyone;
//                   (c) the partial input gradients of the second layer, P_p = W2[F_p, :]' dZ2[F_p, :] (B x 256), of which workgroup q sums the slice F_q;
//                   (d) the first layer's gradient rows dW1[F_p, :], db1[F_p] and the partial sums of squares (the norm gates Adam, training.jl:20).
//                 H1 is kept for 64 samples at a time (LDS) and recomputed for the weight gradient of the second layer (two MFMAs per tile).
//   k_dqn_replay  16 workgroups (DQP_R; they must leave 16 compute units of the XCD to the learner, see STATUS) running the RECORDED replay ops of the same epochs (exec.h records; the bodies are per.hip's and ops_small.h's): stratified
//                 search, row gather, push! of the batch buffer, and -- once the learner has published the td errors of epoch e -- update_priorities!, leaf re-sum
//                 and root paths, all of it beside the learner's backward pass and optimizer step of epoch e.
// The two kernels meet through two counters: batch_ready (replay -> learner: the minibatch of epoch e is in the batch buffer) and td_ready (learner -> replay: the
// batch of epoch e has been read and its td errors are written). Arithmetic: the same f32 MFMA products as the dense engine; the summation orders differ (per-tile
// sequential k, partial sums over workgroups in index order), Adam is evaluated in f32 like the on-policy kernels (mfma_helpers.h adam1) -- parity with the
// call-by-call chain is therefore to a tolerance, replay indices bit-exact (tests/test_gpu_round3.py).
//
// STATUS (round 3): correct, but SLOWER than one launch per phase -- 113-125 us against 73-79 us per C3 epoch -- and therefore opt-in (CRUX_DQN_PERSIST=1). What the
// in-kernel timeline (CRUX_DQP_DEBUG=1: s_memtime stamps of learner workgroup 0 and replay workgroup 0) showed, per epoch: learner batch load + forward of both networks
// 21 us, barrier 3, partial-logit loads 14 (64 `nt` loads per thread: each dependent batch of them takes ~3 us -- they are NOT served at L2-hit latency), head 1, dW3 / dZ2
// / P / dW2 16, barrier 2, P-slice loads + dW1 13, barrier 2, Adam 17 (moments read and written element by element: serialised round trips); replay chain from td_ready to
// batch_ready 79 us (six flag barriers of ~5 us with the agent-scope invalidate each, the search in two rounds on 16 workgroups 20 us, leaf re-sum 9, root paths 6,
// update_priorities! 4 + 2, gather 3). The epoch cycle is replay chain + learner front half (~112 us). What it taught: (1) where a grid's first workgroup lands differs
// from launch to launch, so two kernels only share an XCD if their workgroups select themselves by HW_REG_XCC_ID; (2) `buffer_inv sc0` is no acquire for another compute
// unit's stores (the second epoch read the first epoch's batch rows from the L1) -- exec.hip's barrier now uses the agent-scope form; (3) a learner workgroup at 512
// VGPRs owns its compute unit's register files, so the replay workgroups must leave 16 compute units free (16 of them, not 32), and at 256 VGPRs the kernel spills, which
// puts ~0.6 GB of device-wide scratch behind every launch (+250 us per launch); (4) an in-kernel hand-off with a correct acquire costs about what a launch does (3-5 us),
// so chains of small dependent ops gain nothing from being moved into a persistent kernel -- only the learner's nine launches shrink (to three barriers).
#pragma once
#include "mfma_helpers.h"

#define DQP_G 16
#define DQP_R 16
#define DQP_LD 260
#define NT(ptr) __builtin_nontemporal_load(ptr)      // `nt` loads are served by the L2, past this compute unit's L1

struct DqpArgs {
  float* p; float* m; float* v; double* bp; const float* pt;      // online network: parameters, Adam moments, beta powers; target network: parameters
  int32_t woff[3], boff[3];
  double eta, b1, b2, eps;
  const float* S; const float* SP; const uint8_t* A; const float* R; const uint8_t* DONE; const float* W;      // the batch buffer's columns (W: NULL = unweighted)
  float gamma; int32_t n_epochs;
  float* err;                                     // [B] |Q - y| of the epoch for update_priorities! (NULL: uniform replay)
  float* const* dinfo; int32_t* const* dstatus;   // per epoch: the info row and status word the recording reserved (read back by the host)
  float* zbuf;                                    // [2 networks][G][OUT][B]
  float* pbuf;                                    // [G][B][256]
  float* gbuf;                                    // [256 IN + 256]: dW1 | db1, rows written by their owners
  float* wtg;                                     // [G][16][64][4]: the target network's W2 rows in MFMA A-operand order, written by their workgroup at the start
  float* mv2;                                     // [G][2][256 IN + 256 + OUT]: every workgroup's copy of the Adam moments of the shared parameters
  double* ssq;                                    // [G]
  unsigned* ctrL;                                 // flag barrier of the learner workgroups (+256: abort word)
  unsigned* flags;                                // [0] batch_ready, [1] td_ready, [2] abort
  int32_t* status;                                // CRUX_EHIP when a wait timed out
  int32_t xcd;
  unsigned long long* dbg;                        // CRUX_DQP_DEBUG: s_memtime stamps of learner workgroup 0 ([0, 512)) and replay workgroup 0 ([512, 1024))
};

template <int IN, int OUT, int BT> struct DqpL {
  static_assert(IN >= 1 && IN <= 16 && OUT >= 1 && OUT <= 8 && (BT == 4 || BT == 8), "IN <= 16 inputs, OUT <= 8 actions, 64 or 128 samples");
  static constexpr int B = 16 * BT, NH = BT / 4, KS0 = (IN + 3) / 4, IP = 4 * KS0, XLD = IP + 1, H2LD = B + 4, W3LD = 20, LD = DQP_LD;
  static constexpr int oH1 = 0, oW2 = oH1 + 64 * LD, oX = oW2 + 16 * LD, oW1 = oX + ((B * XLD + 3) & ~3), oB1 = oW1 + ((256 * XLD + 3) & ~3), oH2 = oB1 + 256,
    oW3 = oH2 + 16 * H2LD, oB2 = oW3 + 16 * W3LD, oB3 = oB2 + 16, oMSK = oB3 + 16, oMISC = oMSK + B,
    oW1T = oMISC + 64, oB1T = oW1T + ((256 * XLD + 3) & ~3), oW3T = oB1T + 256, oB2T = oW3T + 16 * W3LD, oB3T = oB2T + 16, oU = oB3T + 16;
  static constexpr int oXP = oU, oH2T = oXP + ((B * XLD + 3) & ~3), endF = oH2T + 16 * H2LD;                        // forward view of the union
  static constexpr int oZ = oU, oDY = oZ + 2 * OUT * B, oDZ2 = oDY + ((OUT * H2LD + 3) & ~3), oDZ1 = oDZ2 + 16 * H2LD, oRED = oDZ1 + 16 * H2LD, oGS = oRED + 1024, endB = oGS + 256;      // backward view
  static constexpr int TOTAL = endF > endB ? endF : endB;
  static constexpr int NS = 256 * IN + 256 + 16 + 16 * OUT + OUT, NSI = (NS + 255) / 256;      // the small parameters: W1 | b1 | b2[F_p] | W3[:, F_p] | b3
  static_assert(TOTAL * 4 <= 150 * 1024, "LDS budget: a replay workgroup (5 KB) must fit beside a learner workgroup on one compute unit");
};

__device__ __forceinline__ bool dqp_wait(const unsigned* flag, unsigned want, const unsigned* abortw) {
  unsigned spins = 0;
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) { __builtin_amdgcn_s_sleep(1);
    if ((++spins & 255u) == 0u && (spins > (1u << 22) || __hip_atomic_load(abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) return false; }      // never hang the GPU
  return true;
}

template <int IN, int OUT, int BT>
__global__ __launch_bounds__(256) void k_dqn_learn(DqpArgs a) {
  using L = DqpL<IN, OUT, BT>;
  constexpr int B = L::B, NH = L::NH, KS0 = L::KS0, XLD = L::XLD, H2LD = L::H2LD, W3LD = L::W3LD, LD = L::LD, G = DQP_G, NS = L::NS, NSI = L::NSI;
  // Both kernels must sit behind the SAME L2. Consecutive workgroups of a grid go round-robin over the 8 XCDs, but where a grid STARTS differs from launch to launch
  // (measured: the learner on XCC 6 or 7, the replay kernel on 0 or 7), so the workgroups pick themselves by their hardware XCC id: a grid of 8 G workgroups has
  // G on every XCD; those on XCD a.xcd draw their index from a counter, the others leave.
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, g = lane >> 4;
  { const uint32_t x = (uint32_t)__builtin_amdgcn_s_getreg(20 | (31 << 11)) & 0xfu;      // hwreg(HW_REG_XCC_ID)
    if ((int)x != a.xcd) return; }
  if (tid == 0) ((unsigned*)sm)[0] = __hip_atomic_fetch_add(a.flags + 4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int p = (int)((unsigned*)sm)[0];
  __syncthreads();
  if (p >= DQP_G) return;
  const float* __restrict__ P = a.p; const float* __restrict__ PT = a.pt;

  // ---- the small parameters as a flat index space (thread-owned: s = tid + 256 k) ----------------------------------------------------------
  auto s_pidx = [&](int s) -> int {          // index into the flat parameter vector
    if (s < 256 * IN) return a.woff[0] + s;
    s -= 256 * IN; if (s < 256) return a.boff[0] + s;
    s -= 256; if (s < 16) return a.boff[1] + 16 * p + s;
    s -= 16; if (s < 16 * OUT) return a.woff[2] + (s % OUT) + OUT * (16 * p + s / OUT);
    return a.boff[2] + (s - 16 * OUT);
  };
  auto s_master = [&](int s) -> int {        // LDS word of the master copy
    if (s < 256 * IN) return L::oW1 + (s & 255) * XLD + (s >> 8);
    s -= 256 * IN; if (s < 256) return L::oB1 + s;
    s -= 256; if (s < 16) return L::oB2 + s;
    s -= 16; if (s < 16 * OUT) return L::oW3 + (s % OUT) * W3LD + s / OUT;
    return L::oB3 + (s - 16 * OUT);
  };
  constexpr int NSH = 256 * IN + 256 + OUT;      // W1 | b1 | b3: shared parameters, moments in this workgroup's copy
  auto s_mv = [&](int s, float*& mp, float*& vp) {
    const bool shared = s < 256 * IN + 256 || s >= 256 * IN + 256 + 16 + 16 * OUT;
    if (shared) { const int q = s < 256 * IN + 256 ? s : s - (16 + 16 * OUT); mp = a.mv2 + (size_t)(2 * p) * NSH + q; vp = a.mv2 + (size_t)(2 * p + 1) * NSH + q; }
    else { const int ix = s_pidx(s); mp = a.m + ix; vp = a.v + ix; }
  };
  // ---- prologue: parameters of both networks, Adam state ------------------------------------------------------------------------------------
  for (int q = tid; q < L::TOTAL; q += 256) sm[q] = 0.f;
  __syncthreads();
  for (int q = tid; q < 16 * 256; q += 256) { const int fl = q & 15, f = q >> 4; sm[L::oW2 + fl * LD + f] = P[a.woff[1] + 16 * p + fl + 256 * f]; }
  for (int q = tid; q < 256 * IN; q += 256) { const int o = q & 255, k = q >> 8; sm[L::oW1 + o * XLD + k] = P[a.woff[0] + q]; sm[L::oW1T + o * XLD + k] = PT[a.woff[0] + q]; }
  sm[L::oB1 + tid] = P[a.boff[0] + tid]; sm[L::oB1T + tid] = PT[a.boff[0] + tid];
  if (tid < 16) { sm[L::oB2 + tid] = P[a.boff[1] + 16 * p + tid]; sm[L::oB2T + tid] = PT[a.boff[1] + 16 * p + tid]; }
  if (tid < 16 * OUT) { const int aa = tid % OUT, fl = tid / OUT; sm[L::oW3 + aa * W3LD + fl] = P[a.woff[2] + aa + OUT * (16 * p + fl)]; sm[L::oW3T + aa * W3LD + fl] = PT[a.woff[2] + aa + OUT * (16 * p + fl)]; }
  if (tid < OUT) { sm[L::oB3 + tid] = P[a.boff[2] + tid]; sm[L::oB3T + tid] = PT[a.boff[2] + tid]; }
  // the target network's rows W2t[F_p, :] in the A-operand order of its second layer (row c, k = 16 u + 4 g + r): constant during the call -- written once into this
  // workgroup's slice of a.wtg ([G][16 u][64 lanes] x 4 floats) and read from there with one coalesced 16-byte load per k-chunk (plain loads: the L1 may keep them)
  f32x4* const wtg = (f32x4*)a.wtg + (size_t)p * 16 * 64;
  if (w == 0) {
#pragma unroll 4
    for (int u = 0; u < 16; ++u) { f32x4 x;
#pragma unroll
      for (int r = 0; r < 4; ++r) x[r] = PT[a.woff[1] + 16 * p + c + 256 * (16 * u + 4 * g + r)];
      wtg[u * 64 + lane] = x; } }
  // own tiles of W2 in the D layout of their gradient: tile ul of wave w = columns [16 (4 w + ul), +16), reg r <-> row 4 g + r, column c
  // Adam moments stay in global memory (their owner reads and writes them once per epoch, coalesced, L2-resident): W1, b1, b3 are updated by every workgroup
  // with identical values -- their moments live in a per-workgroup copy (a.mv2: [G][2][256 IN + 256 + OUT]) so that no workgroup reads what another one writes
#pragma unroll
  for (int k = 0; k < NSI; ++k) { const int s = tid + 256 * k;
    if (s < NS && (s < 256 * IN + 256 || s >= 256 * IN + 256 + 16 + 16 * OUT)) { float* mp; float* vp; s_mv(s, mp, vp); const int ix = s_pidx(s); *mp = a.m[ix]; *vp = a.v[ix]; } }
  double bp1 = a.bp[0], bp2 = a.bp[1];
  AdamK ak; ak.b1 = (float)a.b1; ak.b2 = (float)a.b2; ak.omb1 = (float)(1.0 - a.b1); ak.omb2 = (float)(1.0 - a.b2); ak.eps = (float)a.eps; ak.eta = (float)a.eta;
  const float invB = 1.f / (float)B;
  int nst = 0;
#define DQP_T() do { if (a.dbg && p == 0 && tid == 0 && nst < 500) a.dbg[nst++] = __builtin_amdgcn_s_memtime(); } while (0)
  DQP_T();
  unsigned phase = 0; int err = 0, why = 0;      // why: epoch and place of a failed wait (status[1])
  __syncthreads();

  // first layer of one network for the 64 samples [64 h, 64 h + 64): wave w = sample tile w; H1 rows (sample-major, feature fastest) into LDS. With `mask` the relu
  // pattern of the OWN features F_p is kept as 16 bits per sample (the backward pass needs H1[F_p] only as that mask).
  auto layer1 = [&](const float* sW1n, const float* sB1n, const float* sXn, int h, bool mask) {
    float xb[KS0];
#pragma unroll
    for (int ks = 0; ks < KS0; ++ks) xb[ks] = sXn[(64 * h + 16 * w + c) * XLD + 4 * ks + g];
#pragma unroll 4
    for (int mt = 0; mt < 16; ++mt) { f32x4 acc = *(const f32x4*)&sB1n[16 * mt + 4 * g];
#pragma unroll
      for (int ks = 0; ks < KS0; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(sW1n[(16 * mt + c) * XLD + 4 * ks + g], xb[ks], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = relu1(acc[r]);
      *(f32x4*)&sm[L::oH1 + (16 * w + c) * LD + 16 * mt + 4 * g] = acc;
      if (mask && mt == p) { unsigned bits = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) bits |= (acc[r] > 0.f ? 1u : 0u) << (4 * g + r);
        bits |= __shfl_xor(bits, 16, 64); bits |= __shfl_xor(bits, 32, 64);
        if (g == 0) ((unsigned*)sm)[L::oMSK + 64 * h + 16 * w + c] = bits; } }
  };

  for (int e = 0; e < a.n_epochs && !err; ++e) {
    // ---- the minibatch of this epoch -------------------------------------------------------------------------------------------------------
    if (tid == 0) { const bool ok = dqp_wait(a.flags + 0, (unsigned)(e + 1), a.flags + 2); if (!ok) __hip_atomic_store(a.flags + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); sm[L::oMISC] = ok ? 1.f : 0.f; }
    __syncthreads();
    if (sm[L::oMISC] == 0.f) { err = CRUX_EHIP; why = 100 * e + 1; break; }
    DQP_T();      // 1: batch ready
    // (everything another compute unit wrote -- batch rows, the exchange areas -- is read with L1-bypassing loads below: a CU's vector L1 is never refreshed by other
    //  CUs' stores, and the agent-scope invalidate costs ~1.5 us per use)
    for (int q = tid; q < B * IN; q += 256) { const int s = q / IN, k = q - s * IN; sm[L::oX + s * XLD + k] = NT(&a.S[q]); sm[L::oXP + s * XLD + k] = NT(&a.SP[q]); }
    if constexpr (IN < L::IP) { for (int q = tid; q < B * (L::IP - IN); q += 256) { const int s = q / (L::IP - IN), k = IN + q % (L::IP - IN); sm[L::oXP + s * XLD + k] = 0.f; } }      // (the union is reused: the pad columns of s' again)
    float rr = 0.f, ww = 1.f; unsigned amask = 0, dn = 0;
    if (tid < B) { rr = NT(&a.R[tid]); dn = NT(&a.DONE[tid]); ww = a.W ? NT(&a.W[tid]) : 1.f;
#pragma unroll
      for (int k = 0; k < OUT; ++k) amask |= (NT(&a.A[tid * OUT + k]) ? 1u : 0u) << k; }
    __syncthreads();
    // ---- forward: both networks, every wave its own sample tile from the input to its slice of H2 (no workgroup barrier) ------------------------
#pragma unroll 1
    for (int net = 0; net < 2; ++net) {        // 0: target network on s', 1: online network on s
      const float* sW1n = sm + (net ? L::oW1 : L::oW1T); const float* sB1n = sm + (net ? L::oB1 : L::oB1T); const float* sXn = sm + (net ? L::oX : L::oXP);
      const float* sB2n = sm + (net ? L::oB2 : L::oB2T); float* sH2n = sm + (net ? L::oH2 : L::oH2T);
#pragma unroll 1
      for (int h = 0; h < NH; ++h) {
        layer1(sW1n, sB1n, sXn, h, net == 1);
        f32x4 acc = *(const f32x4*)&sB2n[4 * g];
        if (net) {
#pragma unroll 4
          for (int u = 0; u < 16; ++u) { const f32x4 av = *(const f32x4*)&sm[L::oW2 + c * LD + 16 * u + 4 * g];
            const f32x4 bv = *(const f32x4*)&sm[L::oH1 + (16 * w + c) * LD + 16 * u + 4 * g];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], bv[r], acc, 0, 0, 0); }
        } else {
#pragma unroll 4
          for (int u = 0; u < 16; ++u) { const f32x4 av = wtg[u * 64 + lane]; const f32x4 bv = *(const f32x4*)&sm[L::oH1 + (16 * w + c) * LD + 16 * u + 4 * g];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], bv[r], acc, 0, 0, 0); }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) sH2n[(4 * g + r) * H2LD + 64 * h + 16 * w + c] = relu1(acc[r]);
      }
      // partial logits over the own 16 features: D[a][s], tiles t = w, w + 4 (the columns this wave wrote itself)
      { const f32x4 aw = *(const f32x4*)&sm[(net ? L::oW3 : L::oW3T) + c * W3LD + 4 * g];      // rows a >= OUT are zero
#pragma unroll
        for (int t = w; t < BT; t += 4) { f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[r], sH2n[(4 * g + r) * H2LD + 16 * t + c], acc, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) if (4 * g + r < OUT) a.zbuf[((size_t)(net * G + p) * OUT + 4 * g + r) * B + 16 * t + c] = acc[r]; } }
    }
    DQP_T();      // 2: forward done
    if (!exec_barrier(a.ctrL, (unsigned)p, G, ++phase, 1)) { err = CRUX_EHIP; why = 100 * e + 2; break; }      // ---- (b)
    DQP_T();      // 3: barrier b
    // ---- logits of both networks: partials in workgroup order, then the bias ---------------------------------------------------------------
    if (tid < 2 * B) { const int n = tid / B, s = tid - n * B;
      float zz[OUT];
#pragma unroll
      for (int k = 0; k < OUT; ++k) zz[k] = 0.f;
#pragma unroll 4
      for (int q = 0; q < G; ++q)
#pragma unroll
        for (int k = 0; k < OUT; ++k) { const float x = NT(&a.zbuf[((size_t)(n * G + q) * OUT + k) * B + s]); zz[k] = q == 0 ? x : zz[k] + x; }
#pragma unroll
      for (int k = 0; k < OUT; ++k) sm[L::oZ + (n * OUT + k) * B + s] = zz[k] + sm[(n ? L::oB3 : L::oB3T) + k]; }
    DQP_T();      // 3a: z loaded
    __syncthreads();
    DQP_T();      // 3b: sync
    // ---- dqn_target (rl/dqn.jl:5) and the td_loss head (utils.jl:76-87), every workgroup for itself ---------------------------------------------
    double sl = 0.0, sq = 0.0;
    if (tid < B) {
      float mx = sm[L::oZ + tid];
#pragma unroll
      for (int k = 1; k < OUT; ++k) { const float x = sm[L::oZ + k * B + tid]; mx = x > mx ? x : mx; }
      const float nd = 1.f - (dn ? 1.f : 0.f); const float gn = a.gamma * nd; const float tq = gn * mx; const float y = rr + tq;      // r .+ gamma .* (1 .- done) .* max, un-fused
      float Q = 0.f;
#pragma unroll
      for (int k = 0; k < OUT; ++k) { const float t_ = sm[L::oZ + (OUT + k) * B + tid] * (((amask >> k) & 1u) ? 1.f : 0.f); Q = Q + t_; }
      const float d = Q - y;
      if (p == 0 && a.err) a.err[tid] = fabsf(d);
      sl = (double)(d * d * ww); sq = (double)Q;
#pragma unroll
      for (int k = 0; k < OUT; ++k) sm[L::oDY + k * H2LD + tid] = ((amask >> k) & 1u) ? 2.f * d * ww * invB : 0.f;
    }
    DQP_T();      // 3c: head
    if (p == 0) { sl = wave_sum_d(sl); sq = wave_sum_d(sq); if (lane == 0) { ((double*)(sm + L::oMISC + 8))[2 * w] = sl; ((double*)(sm + L::oMISC + 8))[2 * w + 1] = sq; }
      DQP_T();      // 3d: stats
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); 
      DQP_T(); }    // 3e: stores acknowledged
    __syncthreads();
    DQP_T();      // 4: z totals + head
    if (p == 0 && tid == 0) __hip_atomic_store(a.flags + 1, (unsigned)(e + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // td_ready: the batch is read, its td errors are in the L2
    // ---- backward, own features: dW3[:, F_p] (K = samples, split over the waves), dZ2[F_p, :], db2[F_p] ----------------------------------------------
    { f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = w; t < BT; t += 4) { f32x4 av = {0.f, 0.f, 0.f, 0.f}; if (c < OUT) av = *(const f32x4*)&sm[L::oDY + c * H2LD + 16 * t + 4 * g];
        const f32x4 bv = *(const f32x4*)&sm[L::oH2 + c * H2LD + 16 * t + 4 * g];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], bv[r], acc, 0, 0, 0); }
      *(f32x4*)&sm[L::oRED + (w * 64 + lane) * 4] = acc; }
    { const int fl = tid >> 4, sc = tid & 15; float sb = 0.f;
      float w3[OUT];
#pragma unroll
      for (int k = 0; k < OUT; ++k) w3[k] = sm[L::oW3 + k * W3LD + fl];
#pragma unroll
      for (int j = 0; j < BT; ++j) { const int s = sc + 16 * j; float dh = 0.f;
#pragma unroll
        for (int k = 0; k < OUT; ++k) dh = fmaf(w3[k], sm[L::oDY + k * H2LD + s], dh);
        const float dz = sm[L::oH2 + fl * H2LD + s] > 0.f ? dh : 0.f; sm[L::oDZ2 + fl * H2LD + s] = dz; sb += dz; }
      sb += __shfl_xor(sb, 8, 64); sb += __shfl_xor(sb, 4, 64); sb += __shfl_xor(sb, 2, 64); sb += __shfl_xor(sb, 1, 64);
      if (sc == 0) sm[L::oGS + fl] = sb; }                                   // db2[F_p]
    __syncthreads();
    double ssx = 0.0;                                                        // sums of squares of the gradients this thread produces
    if (w == 0) { f32x4 acc = *(const f32x4*)&sm[L::oRED + lane * 4];
#pragma unroll
      for (int q = 1; q < 4; ++q) { const f32x4 o = *(const f32x4*)&sm[L::oRED + (q * 64 + lane) * 4]; acc[0] += o[0]; acc[1] += o[1]; acc[2] += o[2]; acc[3] += o[3]; }
#pragma unroll
      for (int r = 0; r < 4; ++r) if (4 * g + r < OUT) { sm[L::oGS + 16 + c * OUT + 4 * g + r] = acc[r]; ssx += (double)acc[r] * (double)acc[r]; } }      // dW3[a][F_p], flat (feature, action)
    if (w == 1) {
#pragma unroll
      for (int k = 0; k < OUT; ++k) { float x = 0.f;
#pragma unroll
        for (int j = 0; j < B / 64; ++j) x += sm[L::oDY + k * H2LD + lane + 64 * j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
        if (lane == 0) { sm[L::oGS + 16 + 16 * OUT + k] = x; if (p == 0) ssx += (double)x * (double)x; } } }                                                  // db3 (identical in every workgroup; counted once)
    if (tid < 16) { const float x = sm[L::oGS + tid]; ssx += (double)x * (double)x; }
    // ---- the second layer's input gradient, partial over the own rows: P[s][f] = sum_{f' in F_p} W2[f'][f] dZ2[f'][s], straight to the L2 ------------------
    { float bz[BT][4];
#pragma unroll
      for (int t = 0; t < BT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) bz[t][r] = sm[L::oDZ2 + (4 * g + r) * H2LD + 16 * t + c];
#pragma unroll
      for (int ml = 0; ml < 4; ++ml) { const int mt = 4 * w + ml; float ar[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) ar[r] = sm[L::oW2 + (4 * g + r) * LD + 16 * mt + c];
#pragma unroll
        for (int t = 0; t < BT; ++t) { f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[r], bz[t][r], acc, 0, 0, 0);
          *(f32x4*)&a.pbuf[((size_t)p * B + 16 * t + c) * 256 + 16 * mt + 4 * g] = acc; } } }
    // ---- dW2[F_p, :] = dZ2[F_p, :] H1': H1 recomputed 64 samples at a time -----------------------------------------------------------------
    f32x4 gw2[4];
#pragma unroll
    for (int ul = 0; ul < 4; ++ul) gw2[ul] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int h = 0; h < NH; ++h) {
      layer1(sm + L::oW1, sm + L::oB1, sm + L::oX, h, false);
      __syncthreads();
#pragma unroll
      for (int t = 0; t < 4; ++t) { const f32x4 av = *(const f32x4*)&sm[L::oDZ2 + c * H2LD + 64 * h + 16 * t + 4 * g];
#pragma unroll
        for (int ul = 0; ul < 4; ++ul)
#pragma unroll
          for (int r = 0; r < 4; ++r) gw2[ul] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], sm[L::oH1 + (16 * t + 4 * g + r) * LD + 16 * (4 * w + ul) + c], gw2[ul], 0, 0, 0); }
      __syncthreads();
    }
#pragma unroll
    for (int ul = 0; ul < 4; ++ul)
#pragma unroll
      for (int r = 0; r < 4; ++r) ssx += (double)gw2[ul][r] * (double)gw2[ul][r];
    DQP_T();      // 5: dW3, dZ2, P, dW2
    if (!exec_barrier(a.ctrL, (unsigned)p, G, ++phase, 1)) { err = CRUX_EHIP; why = 100 * e + 3; break; }      // ---- (c)
    DQP_T();      // 6: barrier c
    // ---- dH1[:, F_p]: the partials of all workgroups in index order; relu mask; dZ1[F_p, :] ---------------------------------------------------
    if (tid < 2 * B) { const int s = tid >> 1, jj = tid & 1;
      f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int q = 0; q < G; ++q) { const f32x4 x0 = NT((const f32x4*)&a.pbuf[((size_t)q * B + s) * 256 + 16 * p + 8 * jj]), x1 = NT((const f32x4*)&a.pbuf[((size_t)q * B + s) * 256 + 16 * p + 8 * jj + 4]);
        if (q == 0) { s0 = x0; s1 = x1; } else { s0[0] += x0[0]; s0[1] += x0[1]; s0[2] += x0[2]; s0[3] += x0[3]; s1[0] += x1[0]; s1[1] += x1[1]; s1[2] += x1[2]; s1[3] += x1[3]; } }
      const unsigned bits = ((const unsigned*)sm)[L::oMSK + s];
#pragma unroll
      for (int r = 0; r < 4; ++r) { sm[L::oDZ1 + (8 * jj + r) * H2LD + s] = ((bits >> (8 * jj + r)) & 1u) ? s0[r] : 0.f; sm[L::oDZ1 + (8 * jj + 4 + r) * H2LD + s] = ((bits >> (8 * jj + 4 + r)) & 1u) ? s1[r] : 0.f; } }
    __syncthreads();
    // ---- dW1[F_p, :] (K = samples, split over the waves), db1[F_p]: rows of the gathered first-layer gradient ---------------------------------------
    { f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = w; t < BT; t += 4) { const f32x4 av = *(const f32x4*)&sm[L::oDZ1 + c * H2LD + 16 * t + 4 * g];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], c < L::IP ? sm[L::oX + (16 * t + 4 * g + r) * XLD + c] : 0.f, acc, 0, 0, 0); }
      *(f32x4*)&sm[L::oRED + (w * 64 + lane) * 4] = acc; }
    { const int fl = tid >> 4, sc = tid & 15; float sb = 0.f;
#pragma unroll
      for (int j = 0; j < BT; ++j) sb += sm[L::oDZ1 + fl * H2LD + sc + 16 * j];
      sb += __shfl_xor(sb, 8, 64); sb += __shfl_xor(sb, 4, 64); sb += __shfl_xor(sb, 2, 64); sb += __shfl_xor(sb, 1, 64);
      if (sc == 0) { a.gbuf[256 * IN + 16 * p + fl] = sb; ssx += (double)sb * (double)sb; } }
    __syncthreads();
    if (w == 0) { f32x4 acc = *(const f32x4*)&sm[L::oRED + lane * 4];
#pragma unroll
      for (int q = 1; q < 4; ++q) { const f32x4 o = *(const f32x4*)&sm[L::oRED + (q * 64 + lane) * 4]; acc[0] += o[0]; acc[1] += o[1]; acc[2] += o[2]; acc[3] += o[3]; }
      if (c < IN) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { a.gbuf[16 * p + 4 * g + r + 256 * c] = acc[r]; ssx += (double)acc[r] * (double)acc[r]; } } }
    ssx = wave_sum_d(ssx);
    if (lane == 0) ((double*)(sm + L::oMISC + 24))[w] = ssx;
    __syncthreads();
    if (tid == 0) { const double* q = (const double*)(sm + L::oMISC + 24); a.ssq[p] = ((q[0] + q[1]) + q[2]) + q[3]; }
    DQP_T();      // 7: dH1, dW1
    if (!exec_barrier(a.ctrL, (unsigned)p, G, ++phase, 1)) { err = CRUX_EHIP; why = 100 * e + 4; break; }      // ---- (d)
    DQP_T();      // 8: barrier d
    // ---- norm(grads) (utils.jl:49-55), info, NaN gate (training.jl:20), Adam (Flux.update!, :21) ----------------------------------------------------
    double tot = 0.0;
#pragma unroll 1
    for (int q = 0; q < G; ++q) tot += NT(&a.ssq[q]);
    const bool bad = tot != tot;
    if (p == 0 && tid == 0) { const double* st = (const double*)(sm + L::oMISC + 8); double l_ = 0.0, q_ = 0.0;
      for (int k = 0; k < (B + 63) / 64; ++k) { l_ += st[2 * k]; q_ += st[2 * k + 1]; }
      float* di = a.dinfo[e]; di[CRUX_INFO_LOSS] = (float)(l_ / (double)B); di[2] = (float)(q_ / (double)B); di[CRUX_INFO_GRAD_NORM] = (float)sqrt(tot);
      if (bad) a.dstatus[e][0] = CRUX_ENAN; }
    if (!bad) {      // NaN: no update, no beta-power advance (the host reports "NaN detected!")
      ak.c1 = __builtin_amdgcn_rcpf((float)(1.0 - bp1)); ak.c2 = __builtin_amdgcn_rcpf((float)(1.0 - bp2));
#pragma unroll
      for (int ul = 0; ul < 4; ++ul)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int o = L::oW2 + (4 * g + r) * LD + 16 * (4 * w + ul) + c; const int idx = a.woff[1] + 16 * p + 4 * g + r + 256 * (16 * (4 * w + ul) + c);
          float m_ = a.m[idx], v_ = a.v[idx]; const float d = adam1(gw2[ul][r], m_, v_, ak); a.m[idx] = m_; a.v[idx] = v_; sm[o] = sm[o] - d; }
#pragma unroll
      for (int k = 0; k < NSI; ++k) { const int s = tid + 256 * k;
        if (s < NS) { const float gsm = s < 256 * IN + 256 ? NT(&a.gbuf[s]) : sm[L::oGS + (s - (256 * IN + 256))];
          float* mp; float* vp; s_mv(s, mp, vp);
          float m_ = *mp, v_ = *vp; const float d = adam1(gsm, m_, v_, ak); *mp = m_; *vp = v_; const int mo = s_master(s); sm[mo] = sm[mo] - d; } }
      bp1 *= a.b1; bp2 *= a.b2;
    }
    __syncthreads();
    DQP_T();      // 9: Adam
  }
  // ---- write back ------------------------------------------------------------------------------------------------------------------------------
  __syncthreads();
#pragma unroll
  for (int ul = 0; ul < 4; ++ul)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int idx = a.woff[1] + 16 * p + 4 * g + r + 256 * (16 * (4 * w + ul) + c);
      a.p[idx] = sm[L::oW2 + (4 * g + r) * LD + 16 * (4 * w + ul) + c]; }
#pragma unroll
  for (int k = 0; k < NSI; ++k) { const int s = tid + 256 * k;
    if (s < NS) { const bool shared = s < 256 * IN + 256 || s >= 256 * IN + 256 + 16 + 16 * OUT;      // W1, b1, b3: the same values in every workgroup
      if (!shared || p == 0) { const int ix = s_pidx(s); a.p[ix] = sm[s_master(s)];
        if (shared) { float* mp; float* vp; s_mv(s, mp, vp); a.m[ix] = *mp; a.v[ix] = *vp; } } } }      // (workgroup 0's copy of the shared moments goes back to the network)
  DQP_T();
  if (tid == 0 && p == 0) { a.bp[0] = bp1; a.bp[1] = bp2; }
  if (err && tid == 0) { a.status[0] = err; a.status[1] = why; __hip_atomic_store(a.flags + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}

// The replay side: stage A of epoch e (search | gather, ring ids, snapshot, zero fills | push! priorities of the batch) after stage C of epoch e - 1
// (update_priorities! | leaf re-sum | root paths), which waits for the learner's td errors. tab: per epoch [offA, nA, offC, nC], then entries (op index, barrier after it).
__global__ __launch_bounds__(256) void k_dqn_replay(const ExecOp* __restrict__ ops, const int32_t* __restrict__ tab, int n_epochs, unsigned* ctr, unsigned* flags, int xcd, int32_t* status, unsigned long long* dbg) {
  { const uint32_t x = (uint32_t)__builtin_amdgcn_s_getreg(20 | (31 << 11)) & 0xfu;      // the workgroups on XCD `xcd` (see k_dqn_learn)
    if ((int)x != xcd) return; }
  __shared__ int ok_w; __shared__ unsigned wg_s;
  if (threadIdx.x == 0) wg_s = __hip_atomic_fetch_add(flags + 5, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const unsigned wg = wg_s, R = gridDim.x >> 3;
  if (wg >= R) return;
  unsigned phase = 0, off = 0; bool fail = false; int why = 0; int nst = 0;
#define DQR_T() do { if (dbg && wg == 0 && threadIdx.x == 0 && nst < 500) dbg[512 + nst++] = __builtin_amdgcn_s_memtime(); } while (0)
  DQR_T();

  auto wait_td = [&](int upto) -> bool {
    if (threadIdx.x == 0) { const bool ok = dqp_wait(flags + 1, (unsigned)upto, flags + 2); ok_w = ok ? 1 : 0; }
    __syncthreads();
    const bool ok = ok_w != 0;
    asm volatile("buffer_inv sc1\n\ts_dcache_inv\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // agent-scope acquire: the op bodies read with plain loads
    __syncthreads();
    return ok;
  };
  auto stage = [&](int o0, int n) -> bool {
    for (int i = 0; i < n; ++i) {
      const int oi = tab[o0 + 2 * i], bar = tab[o0 + 2 * i + 1];
      const ExecOp* op = ops + oi; const int kid = op->kid; const unsigned nb = op->nblocks;
      const unsigned b0 = (wg + R - off) % R;
      for (unsigned b = b0; b < nb; b += R) { EXEC_SWITCH_REPLAY(exec_dispatch_g) __syncthreads(); }
      off = bar ? 0u : (off + nb) % R;
      DQR_T();
      if (bar) { if (!exec_barrier(ctr, wg, R, ++phase, 4)) return false; DQR_T(); }      // 4: buffer_inv sc1 (buffer_inv sc0 does not reach the L1)
    }
    return true;
  };
  for (int e = 0; e <= n_epochs && !fail; ++e) {
    if (e > 0) {        // stage C of epoch e - 1 needs its td errors; stage A of epoch e overwrites the batch rows the learner has read by then
      DQR_T();
      if (!wait_td(e)) { fail = true; why = 100 * e + 11; break; }
      DQR_T();
      if (!stage(tab[4 * (e - 1) + 2], tab[4 * (e - 1) + 3])) { fail = true; why = 100 * e + 12; break; }
    }
    if (e == n_epochs) break;
    if (!stage(tab[4 * e + 0], tab[4 * e + 1])) { fail = true; why = 100 * e + 13; break; }
    if (wg == 0 && threadIdx.x == 0) __hip_atomic_store(flags + 0, (unsigned)(e + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // batch_ready (every stage ends with a barrier: all rows are in the L2)
  }
  if (fail && threadIdx.x == 0) { status[0] = CRUX_EHIP; status[2] = why; __hip_atomic_store(flags + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}
