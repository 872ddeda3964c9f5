// train_fs2.hip -- dispatch of the feature-split, role-specialised learner kernel (train_fs2_kernel.h: k_train_fs2<IN, OUT, KIND, ACT, H2, ACT2, TIMING, PX, PXK, LAG>): full
// batch_train! loops (src/training.jl:28-55) of the IN->64->{64,32}->OUT family with minibatches of 65..128 rows, on four compute units of one XCD with four compute + four
// helper waves each -- the plain policy-gradient / critic losses, lagrange_ppo_loss (LAG) and the replica-group forms (PX / PXK) of every shape in the list below.
// CRUX_FS=0 switches the kernel off (the sample-split two-CU kernel, or the dense engine for the 32-wide second layer, then run).
#include "train_fs2_kernel.h"

template <int IN, int OUT, int KIND, int ACT, int H2, int ACT2, bool TIMING, bool PX = false, bool PXK = false, bool LAG = false>
static int32_t launch_fs2_form(crux_ctx* c, TrainArgs& a, hipStream_t stream) {
  using Lt = Fs2Layout<IN, OUT, H2, LAG>;
  constexpr size_t lds = sizeof(float) * (size_t)(PX && Lt::BK_FITS ? Lt::TOTAL_BK : Lt::TOTAL);
  static bool attr_dev[16] = {}; bool& attr = attr_dev[c->device & 15];
  if (!attr) { HIPCHK(c, hipFuncSetAttribute((const void*)k_train_fs2<IN, OUT, KIND, ACT, H2, ACT2, TIMING, PX, PXK, LAG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; }
  hipLaunchKernelGGL((k_train_fs2<IN, OUT, KIND, ACT, H2, ACT2, TIMING, PX, PXK, LAG>), dim3(32), dim3(512), lds, stream, a);
  return crux_launch_check(c, PXK ? "k_train_fs2 (replica group, periodic form)" : PX ? "k_train_fs2 (replica group)" : LAG ? "k_train_fs2 (lagrange_ppo_loss)" : "k_train_fs2");
}
// the policy shapes lagrange_ppo_loss (ppo.jl:70-131) is instantiated for: C2's actor, the LunarLander-shaped one, Pendulum's and C5's
template <int IN, int OUT, int KIND, int ACT, int H2, int ACT2> constexpr bool FS2_HAS_LAG = KIND != MFK_VALUE && H2 == 64 && ACT2 == ACT &&
  ((IN == 4 && OUT == 2 && ACT == CRUX_ACT_RELU) || (IN == 8 && OUT == 4 && ACT == CRUX_ACT_RELU) || (IN == 3 && OUT == 1 && ACT == CRUX_ACT_RELU) || (IN == 17 && OUT == 6 && ACT == CRUX_ACT_TANH));
// replica-group forms (PX / PXK): every shape. Where the W2 backups fit into LDS beside the kernel's own 113..132 KB they live there; the 24- and 27-input shapes keep them in
// registers (their periodic forms spill 2..61 registers, the per-step forms 0..4)
template <int IN, int OUT, int KIND, int ACT, int H2, int ACT2> constexpr bool FS2_HAS_PX = true;
template <int IN, int OUT, int KIND, int ACT, int H2, int ACT2>
static int32_t launch_fs2(crux_ctx* c, TrainArgs a, bool timing, hipStream_t stream) {
  const int which = stream == c->stream ? 0 : 1;
  constexpr size_t xfloats = (size_t)CRUX_XBUF_FLOATS;
  if (!c->xbuf[which]) { if (hipMalloc(&c->xbuf[which], sizeof(float) * xfloats + 256) != hipSuccess) return crux_fail(c, CRUX_ENOMEM, "learner exchange buffer"); }
  a.xbuf = (float*)c->xbuf[which]; a.xctr = (unsigned*)((char*)c->xbuf[which] + sizeof(float) * xfloats);
  HIPCHK(c, hipMemsetAsync(c->xbuf[which], 0, sizeof(float) * xfloats + 256, stream));      // counters AND slots: the granules' step tags start from zero (a stale tag must never look like this launch's)
  a.xcd = which;      // actor / critic (the context's two learner streams) behind different L2s
  if constexpr (FS2_HAS_LAG<IN, OUT, KIND, ACT, H2, ACT2>) { if (a.lag) return launch_fs2_form<IN, OUT, KIND, ACT, H2, ACT2, false, false, false, true>(c, a, stream); }
  if (a.lag) return crux_fail(c, CRUX_EINVAL, "k_train_fs2: no lagrange instantiation for this shape");
  if constexpr (FS2_HAS_PX<IN, OUT, KIND, ACT, H2, ACT2>) if (crux_grouped(c) && a.need_px) {      // replica group: the in-kernel all-reduce over the peer slots (comm.hip)
    a.px_hist = c->peer_hist ? 1 : 0; a.px_n = c->peer_n; a.px_rank = c->peer_rank; a.px_tab = c->peer_tab + which * CRUX_PX_MAXR;
    if (a.px_every > 1) return launch_fs2_form<IN, OUT, KIND, ACT, H2, ACT2, false, true, true>(c, a, stream);      // periodic form: local Adam steps, theta / m / v averaged every k-th
    return launch_fs2_form<IN, OUT, KIND, ACT, H2, ACT2, false, true, false>(c, a, stream);
  }
  constexpr bool HAS_TIMING = H2 == 64 && ACT2 == ACT && ((IN == 4 && (OUT == 2 || OUT == 1)) || (IN == 17 && ACT == CRUX_ACT_TANH));      // the in-kernel phase timers are instantiated for the C2 / C5 learners only
  if constexpr (HAS_TIMING) if (timing) {
    static unsigned long long* dbg = nullptr;
    if (!dbg) { if (hipMalloc(&dbg, 512 * 8) != hipSuccess) return crux_fail(c, CRUX_ENOMEM, "timing buffer"); }
    a.dbg = dbg;
    int32_t rc = launch_fs2_form<IN, OUT, KIND, ACT, H2, ACT2, true>(c, a, stream); if (rc) return rc;
    unsigned long long h[512]; HIPCHK(c, hipMemcpyAsync(h, dbg, sizeof h, hipMemcpyDeviceToHost, stream)); HIPCHK(c, hipStreamSynchronize(stream));
    static const char* nc[16] = {"loop", "wait staged", "fwdL1+T1", "fwdL2", "L3+pair barrier+head", "dW3+dZ2+stats+T2", "wait B_1", "dW2+send+dH1", "dZ1+db+dW1", "B_2+reduce+granules", "wait P1",
                                 "W2 loads+granule poll", "drain loads", "totals+adam W2+ssq", "adam small", "B_b+report+exit"};
    static const char* nh[16] = {"loop", "fetch", "stage next", "wait B_1", "dW2+send+drain", "arrival 1+wait(leader)", "-", "-", "-", "wait B_2+reduce+granules", "wait P1",
                                 "W2 loads+granule poll", "drain loads", "totals+adam W2+ssq", "adam small", "B_b+exit"};
    for (int wg = 0; wg < 4; ++wg) for (int w : {0, 4}) { const char** nm = w == 0 ? nc : nh;
      fprintf(stderr, "[fs2-timing] %d-%d wg %d %s wave %d:", IN, OUT, wg, w == 0 ? "compute" : "helper", w); unsigned long long tot = 0; for (int k = 0; k < 16; ++k) tot += h[(8 * wg + w) * 16 + k];
      for (int k = 0; k < 16; ++k) if (nm[k][0] != '-') fprintf(stderr, " %s=%.1f%%", nm[k], 100.0 * (double)h[(8 * wg + w) * 16 + k] / (double)tot);
      fprintf(stderr, " total=%llu\n", tot); }
    return CRUX_OK;
  }
  return launch_fs2_form<IN, OUT, KIND, ACT, H2, ACT2, false>(c, a, stream);
}

extern "C" int crux_x2_placement_ok(crux_ctx* c);      // train_mfma_x2.hip: workgroups i and i + 8 of a grid share an XCD (probed once per process)

// Called first by crux_train_mfma_launch (train_mfma.hip): IN -> 64 -> {64, 32} -> OUT, identity output layer, full minibatch loops with Adam, the plain policy-gradient / critic
// losses or lagrange_ppo_loss; replica groups take the PX / PXK instantiations.
// probe: only answer whether this call would be taken (policy_gradient_training asks before it commits a pair of learners to the two learner streams)
int32_t crux_train_fs_launch(crux_ctx* c, const TrainArgs& a, bool* handled, hipStream_t stream, bool probe) {
  *handled = false;
  if (crux_sw().fs == 0) return CRUX_OK;            // read per call: tests switch the form inside one process
  if (c->learner_cus != 0 && !a.need_px) return CRUX_OK;      // crux_ctx_set_learner_cus(1 | 2): the caller asked for the one- / two-CU kernels (population runs)
  const NetDesc& nd = a.nd;
  if (nd.L != 3 || nd.dims[1] != MF_HID || (nd.dims[2] != 64 && nd.dims[2] != 32) || nd.acts[2] != CRUX_ACT_IDENTITY) return CRUX_OK;
  if (a.ids || !a.apply || a.bs <= 64 || a.bs > 128 || a.len < a.bs) return CRUX_OK;
  int kind;
  if (a.loss == CRUX_LOSS_VALUE_MSE) kind = MFK_VALUE;
  else if (!CRUX_IS_PG(a.loss)) return CRUX_OK;
  else if (a.head == CRUX_HEAD_CATEGORICAL) kind = MFK_CATEGORICAL;
  else if (a.head == CRUX_HEAD_GAUSSIAN) kind = MFK_GAUSSIAN;
  else return CRUX_OK;
  if (!crux_x2_placement_ok(c)) return CRUX_OK;
  // 32-bit loop control inside the kernel: buffers below 2^30 rows, launches below 2^31 steps (anything larger stays with the two-CU kernel / the dense-engine learner)
  if (a.len >= (1ll << 30) || (long long)a.epochs * ((a.len + a.bs - 1) / a.bs) >= 0x7fffffffll) return CRUX_OK;
  const int in = nd.dims[0], h2 = nd.dims[2], out = nd.dims[3], act = nd.acts[0], act2 = nd.acts[1];
  if (a.lag) {     // lagrange_ppo_loss (crux_batch_train_lagrange passes the PPO head with the controller attached): one replica, the shapes of FS2_HAS_LAG;
                   // everything else stays with the two-CU kernel / the dense-engine learner
    if (kind == MFK_VALUE || a.loss != CRUX_LOSS_PPO || (crux_grouped(c) && a.need_px) || h2 != 64 || act2 != act) return CRUX_OK;
    const bool shape = (in == 4 && out == 2 && kind == MFK_CATEGORICAL && act == CRUX_ACT_RELU) || (in == 8 && out == 4 && kind == MFK_CATEGORICAL && act == CRUX_ACT_RELU) ||
                       (in == 3 && out == 1 && kind == MFK_GAUSSIAN && act == CRUX_ACT_RELU) || (in == 17 && out == 6 && kind == MFK_GAUSSIAN && act == CRUX_ACT_TANH);
    if (!shape) return CRUX_OK;
  }
  const bool timing = crux_sw().mfma_timing;
  const bool grouped = crux_grouped(c) && a.need_px;
#define FS2_CASE2(I, O, K, A1, H, A2) if (in == I && out == O && kind == K && act == A1 && h2 == H && act2 == A2) { if (grouped && !FS2_HAS_PX<I, O, K, A1, H, A2>) return CRUX_OK; \
    *handled = true; if (probe) return CRUX_OK; return launch_fs2<I, O, K, A1, H, A2>(c, a, timing, stream); }
#define FS2_CASE(I, O, K, A_) FS2_CASE2(I, O, K, A_, 64, A_)
  FS2_CASE(4, 2, MFK_CATEGORICAL, CRUX_ACT_RELU)     // C2 actor  (PPO CartPole)
  FS2_CASE(4, 1, MFK_VALUE, CRUX_ACT_RELU)           // C2 critic
  FS2_CASE(17, 6, MFK_GAUSSIAN, CRUX_ACT_TANH)       // C5 actor  (PPO HalfCheetah-shaped, 17 obs / 6 act)
  FS2_CASE(17, 1, MFK_VALUE, CRUX_ACT_TANH)          // C5 critic
#ifndef CRUX_FS2_MINIMAL
  FS2_CASE(3, 1, MFK_GAUSSIAN, CRUX_ACT_RELU)        // Pendulum actor
  FS2_CASE(3, 1, MFK_VALUE, CRUX_ACT_RELU)           // Pendulum critic
  FS2_CASE(17, 6, MFK_GAUSSIAN, CRUX_ACT_RELU)
  FS2_CASE(17, 1, MFK_VALUE, CRUX_ACT_RELU)
  FS2_CASE(8, 4, MFK_CATEGORICAL, CRUX_ACT_RELU)     // 8 observations / 4 discrete actions (LunarLander-shaped)
  FS2_CASE(8, 1, MFK_VALUE, CRUX_ACT_RELU)
  FS2_CASE(2, 1, MFK_GAUSSIAN, CRUX_ACT_RELU)        // the reference's own Pendulum examples observe (theta, theta_dot): 2 inputs (examples/rl/pendulum.jl)
  FS2_CASE(2, 1, MFK_VALUE, CRUX_ACT_RELU)
  // standard Gym shapes: Acrobot 6 / 3, MountainCar 2 / 3, LunarLanderContinuous 8 / 2, Hopper 11 / 3, BipedalWalker 24 / 4, Ant 27 / 8
  FS2_CASE(6, 3, MFK_CATEGORICAL, CRUX_ACT_RELU)  FS2_CASE(6, 1, MFK_VALUE, CRUX_ACT_RELU)
  FS2_CASE(2, 3, MFK_CATEGORICAL, CRUX_ACT_RELU)
  FS2_CASE(8, 2, MFK_GAUSSIAN, CRUX_ACT_RELU)     FS2_CASE(8, 2, MFK_GAUSSIAN, CRUX_ACT_TANH)     FS2_CASE(8, 1, MFK_VALUE, CRUX_ACT_TANH)
  FS2_CASE(11, 3, MFK_GAUSSIAN, CRUX_ACT_RELU)    FS2_CASE(11, 3, MFK_GAUSSIAN, CRUX_ACT_TANH)    FS2_CASE(11, 1, MFK_VALUE, CRUX_ACT_RELU)   FS2_CASE(11, 1, MFK_VALUE, CRUX_ACT_TANH)
  FS2_CASE(24, 4, MFK_GAUSSIAN, CRUX_ACT_RELU)    FS2_CASE(24, 4, MFK_GAUSSIAN, CRUX_ACT_TANH)    FS2_CASE(24, 1, MFK_VALUE, CRUX_ACT_RELU)   FS2_CASE(24, 1, MFK_VALUE, CRUX_ACT_TANH)
  FS2_CASE(27, 8, MFK_GAUSSIAN, CRUX_ACT_RELU)    FS2_CASE(27, 8, MFK_GAUSSIAN, CRUX_ACT_TANH)    FS2_CASE(27, 1, MFK_VALUE, CRUX_ACT_RELU)   FS2_CASE(27, 1, MFK_VALUE, CRUX_ACT_TANH)
  // the reference's HalfCheetah PPO networks (examples/rl/half_cheetah_mujoco.jl:33-38): mu = 17 -tanh-> 64 -tanh-> 32 -> 6, V = 17 -tanh-> 64 -> 32 -> 1
  FS2_CASE2(17, 6, MFK_GAUSSIAN, CRUX_ACT_TANH, 32, CRUX_ACT_TANH)
  FS2_CASE2(17, 1, MFK_VALUE, CRUX_ACT_TANH, 32, CRUX_ACT_IDENTITY)
  FS2_CASE2(17, 1, MFK_VALUE, CRUX_ACT_TANH, 32, CRUX_ACT_TANH)
#endif
#undef FS2_CASE
#undef FS2_CASE2
  return CRUX_OK;
}
