// train_fs.hip -- dispatch of the feature-split form of the register-resident learner kernel (train_fs_kernel.h: k_train_fs<IN, OUT, KIND, ACT, NWG, HELP, ..., H2, ACT2>):
// full batch_train! loops (src/training.jl:28-55) of the IN->64->H2->OUT family (H2 = 64 or 32) with minibatches of 65..128 rows, on four compute units of one XCD with
// four compute + four helper waves each (default), on four with four waves, or on two with eight.
#include "train_fs_kernel.h"

extern "C" int crux_x2_placement_ok(crux_ctx* c);      // train_mfma_x2.hip: workgroups i and i + 8 of a grid share an XCD (probed once per process)
int32_t crux_train_fs2_launch(crux_ctx* c, const TrainArgs& a, int kind, bool* handled, hipStream_t stream, bool probe);      // train_fs2.hip: the role-specialised form (round 5)

template <int IN, int OUT, int KIND, int ACT, int H2, int ACT2, int NWG, bool HELP, bool TIMING, bool PX, bool LAG = false, bool PXK = false>
static int32_t launch_fs_form(crux_ctx* c, TrainArgs& a, hipStream_t stream) {
  using Lt = FsLayout<IN, OUT, NWG, HELP, H2, LAG>;
  constexpr size_t lds = sizeof(float) * (size_t)Lt::TOTAL;
  static bool attr_dev[16] = {}; bool& attr = attr_dev[c->device & 15];      // (per device: a second device in the process sets the attribute for itself)
  if (!attr) { HIPCHK(c, hipFuncSetAttribute((const void*)k_train_fs<IN, OUT, KIND, ACT, NWG, HELP, TIMING, PX, H2, ACT2, LAG, PXK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; }
  hipLaunchKernelGGL((k_train_fs<IN, OUT, KIND, ACT, NWG, HELP, TIMING, PX, H2, ACT2, LAG, PXK>), dim3(8 * NWG), dim3(Lt::NT), lds, stream, a);
  return crux_launch_check(c, PXK ? "k_train_fs (replica group, periodic form)" : PX ? "k_train_fs (replica group)" : LAG ? "k_train_fs (lagrange_ppo_loss)" : "k_train_fs");
}
// form: 2 = two workgroups of eight waves; 4 = four workgroups of four waves; 8 = four workgroups of four compute + four helper waves (the only form of the 32-wide second layer)
// (the shapes added in round 3 for the standard Gym tasks -- IN > 17 or not one of the benchmark shapes -- are instantiated in the default form only: FS_LITE)
template <int IN, int OUT> constexpr bool FS_LITE = !((IN == 4 && (OUT == 2 || OUT == 1)) || (IN == 3 && OUT == 1) || (IN == 17 && (OUT == 6 || OUT == 1)) || (IN == 8 && (OUT == 4 || OUT == 1)) || (IN == 2 && OUT == 1));
template <int IN, int OUT> constexpr bool FS_WIDE_HEAD = IN == 27 && OUT == 8;
// replica-group forms (PX / PXK) of THIS kernel: the shapes k_train_fs2 does not take as a group (24 / 27 inputs) and the C2 / C5 learners, on which the two kernels are
// compared (tests/test_gpu_fs2.py) and CRUX_FS2=0 still runs a group; every other shape of a group is k_train_fs2's
template <int IN, int OUT, int KIND, int ACT, int H2, int ACT2> constexpr bool FS_HAS_PX = IN >= 24 || (H2 == 64 && ACT2 == ACT &&
  ((IN == 4 && ACT == CRUX_ACT_RELU && ((OUT == 2 && KIND == MFK_CATEGORICAL) || (OUT == 1 && KIND == MFK_VALUE))) ||
   (IN == 17 && ACT == CRUX_ACT_TANH && ((OUT == 6 && KIND == MFK_GAUSSIAN) || (OUT == 1 && KIND == MFK_VALUE)))));
template <int IN, int OUT, int KIND, int ACT, int H2, int ACT2, bool TIMING>
static int32_t launch_fs_pick(crux_ctx* c, TrainArgs& a, int form, hipStream_t stream) {
  if constexpr (H2 == 64 && ACT2 == ACT && !FS_LITE<IN, OUT>) {
    if (form == 2) return launch_fs_form<IN, OUT, KIND, ACT, H2, ACT2, 2, false, TIMING, false>(c, a, stream);
    if (form == 4) return launch_fs_form<IN, OUT, KIND, ACT, H2, ACT2, 4, false, TIMING, false>(c, a, stream);
  }
  if constexpr (H2 == 64 && ACT2 == ACT && FS_WIDE_HEAD<IN, OUT> && !TIMING) {      // wide Gaussian heads (Ant 27 / 8): four waves per workgroup have 512 registers each, the helper-wave form spilled 52
    if (form == 4 || form == 0) return launch_fs_form<IN, OUT, KIND, ACT, H2, ACT2, 4, false, false, false>(c, a, stream);
  }
  return launch_fs_form<IN, OUT, KIND, ACT, H2, ACT2, 4, true, TIMING, false>(c, a, stream);
}
template <int IN, int OUT, int KIND, int ACT, int H2, int ACT2>
static int32_t launch_fs(crux_ctx* c, TrainArgs a, int form, bool timing, hipStream_t stream) {
  const int which = stream == c->stream ? 0 : 1;
  constexpr size_t xfloats = (size_t)CRUX_XBUF_FLOATS;
  if (!c->xbuf[which]) { if (hipMalloc(&c->xbuf[which], sizeof(float) * xfloats + 256) != hipSuccess) return crux_fail(c, CRUX_ENOMEM, "learner exchange buffer"); }
  a.xbuf = (float*)c->xbuf[which]; a.xctr = (unsigned*)((char*)c->xbuf[which] + sizeof(float) * xfloats);
  HIPCHK(c, hipMemsetAsync(a.xctr, 0, 256, stream));
  a.xcd = which;      // actor / critic (the context's two learner streams) behind different L2s. (Replicas sharing a device stay on the same XCD pair: spreading them over XCDs
                      // sent their flag / slot traffic across L2s and measured 15.5 against 12.9 us per step on one GPU.)
  if constexpr (FS_HAS_PX<IN, OUT, KIND, ACT, H2, ACT2>) if (crux_grouped(c) && a.need_px) {      // replica group: four workgroups with helper waves, the in-kernel all-reduce over the peer slots
    a.px_hist = c->peer_hist ? 1 : 0; a.px_n = c->peer_n; a.px_rank = c->peer_rank; a.px_tab = c->peer_tab + which * CRUX_PX_MAXR;
    if (a.px_every > 1) return launch_fs_form<IN, OUT, KIND, ACT, H2, ACT2, 4, true, false, true, false, true>(c, a, stream);      // periodic form: local Adam steps, theta / m / v averaged every k-th
    return launch_fs_form<IN, OUT, KIND, ACT, H2, ACT2, 4, true, false, true>(c, a, stream);
  }
  if constexpr (KIND != MFK_VALUE && H2 == 64 && ACT2 == ACT && ((IN == 4 && OUT == 2) || (IN == 8 && OUT == 4) || (IN == 3 && OUT == 1) || (IN == 17 && OUT == 6 && ACT == CRUX_ACT_TANH))) {
    if (a.lag) return launch_fs_form<IN, OUT, KIND, ACT, H2, ACT2, 4, true, false, false, true>(c, a, stream);      // lagrange_ppo_loss (ppo.jl:70-131): the helper-wave form with the penalty controller
  }
  if (a.lag) return crux_fail(c, CRUX_EINVAL, "k_train_fs: no lagrange instantiation for this shape");
  constexpr bool HAS_TIMING = H2 == 64 && ACT2 == ACT && ((IN == 4 && (OUT == 2 || OUT == 1)) || (IN == 17 && ACT == CRUX_ACT_TANH));      // the in-kernel phase timers are instantiated for the C2 / C5 learners only
  if constexpr (HAS_TIMING) if (timing) {
    static unsigned long long* dbg = nullptr;
    if (!dbg) { if (hipMalloc(&dbg, 512 * 8) != hipSuccess) return crux_fail(c, CRUX_ENOMEM, "timing buffer"); }
    a.dbg = dbg;
    int32_t rc = launch_fs_pick<IN, OUT, KIND, ACT, H2, ACT2, true>(c, a, form, stream); if (rc) return rc;
    unsigned long long h[512]; HIPCHK(c, hipMemcpyAsync(h, dbg, sizeof h, hipMemcpyDeviceToHost, stream)); HIPCHK(c, hipStreamSynchronize(stream));
    static const char* nm[16] = {"loop+prefetch", "stage", "fwdL1+T1", "fwdL2", "L3+z-exchange+head", "dW3+dZ2+stats+T2", "wait B_1", "dW2+send", "dH1", "dZ1+db+dW1", "B_2+reduce+store", "exchange wait",
                                 "load slots+total+ssq", "wait B_or", "info+adam", "wait B_b"};
    const int nw = form == 8 ? 8 : 16 / form, ntot = form == 8 ? 32 : 16;
    if (form == 8) { fprintf(stderr, "[fs-timing] %d-%d wg 0 helper wave 4:", IN, OUT); unsigned long long tot = 0; for (int k = 0; k < 16; ++k) tot += h[4 * 16 + k];
      for (int k = 0; k < 16; ++k) fprintf(stderr, " %s=%.1f%%", nm[k], 100.0 * (double)h[4 * 16 + k] / (double)tot); fprintf(stderr, " total=%llu\n", tot); }
    for (int w = 0; w < ntot; w += nw) { fprintf(stderr, "[fs-timing] %d-%d wg %d wave 0:", IN, OUT, w / nw); unsigned long long tot = 0; for (int k = 0; k < 16; ++k) tot += h[w * 16 + k];
      for (int k = 0; k < 16; ++k) fprintf(stderr, " %s=%.1f%%", nm[k], 100.0 * (double)h[w * 16 + k] / (double)tot); fprintf(stderr, " total=%llu\n", tot); }
    return CRUX_OK;
  }
  return launch_fs_pick<IN, OUT, KIND, ACT, H2, ACT2, false>(c, a, form, stream);
}

// Called first by crux_train_mfma_launch (train_mfma.hip) with its own shape test: IN -> 64 -> {64, 32} -> OUT, identity output layer, full minibatch loops with Adam, the plain
// policy-gradient / critic losses; replica groups take the PX instantiation. CRUX_FS=0 switches the form off (the sample-split two-CU kernel, or the dense engine for the 32-wide
// second layer, then run), CRUX_FS_WG=2|4|8 picks the form.
// probe: only answer whether this call would be taken (policy_gradient_training asks before it commits a pair of learners to the two learner streams)
int32_t crux_train_fs_launch(crux_ctx* c, const TrainArgs& a, bool* handled, hipStream_t stream, bool probe) {
  *handled = false;
  const int mode = crux_sw().fs;            // read per call: tests switch the form inside one process
  const int form_env = crux_sw().fs_wg;
  if (mode == 0) return CRUX_OK;
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
  const int in = nd.dims[0], h2 = nd.dims[2], out = nd.dims[3], act = nd.acts[0], act2 = nd.acts[1];
  if (a.lag) {     // lagrange_ppo_loss (crux_batch_train_lagrange passes the PPO head with the controller attached): the helper-wave form, one replica, the shapes instantiated in launch_fs;
                   // everything else stays with the two-CU kernel / the dense-engine learner
    if (kind == MFK_VALUE || a.loss != CRUX_LOSS_PPO || (crux_grouped(c) && a.need_px) || (form_env != 0 && form_env != 8) || h2 != 64 || act2 != act) return CRUX_OK;
    const bool shape = (in == 4 && out == 2 && kind == MFK_CATEGORICAL && act == CRUX_ACT_RELU) || (in == 8 && out == 4 && kind == MFK_CATEGORICAL && act == CRUX_ACT_RELU) ||
                       (in == 3 && out == 1 && kind == MFK_GAUSSIAN && act == CRUX_ACT_RELU) || (in == 17 && out == 6 && kind == MFK_GAUSSIAN && act == CRUX_ACT_TANH);
    if (!shape) return CRUX_OK;
  }
  // round 5: the plain learners (no lagrange_ppo_loss, no explicitly requested older form) take the role-specialised kernel k_train_fs2 -- the same arithmetic, bit-identical
  // parameters, the W2 hand-over in the helper waves beside the compute waves' backward pass --, replica groups too on the shapes train_fs2.hip instantiates. CRUX_FS2=0 keeps k_train_fs.
  if (crux_sw().fs2 && form_env == 0 && !a.lag && a.len < (1ll << 30) && (long long)a.epochs * ((a.len + a.bs - 1) / a.bs) < 0x7fffffffll) {
    const int32_t rc2 = crux_train_fs2_launch(c, a, kind, handled, stream, probe);
    if (rc2 || *handled) return rc2; }
  const int form = form_env == 2 || form_env == 4 || form_env == 8 ? form_env : CRUX_FS_DEFAULT_WG;      // 8 = four workgroups with helper waves
  const bool timing = crux_sw().mfma_timing;
  const bool grouped = crux_grouped(c) && a.need_px;
#define FS_CASE2(I, O, K, A1, H, A2) if (in == I && out == O && kind == K && act == A1 && h2 == H && act2 == A2) { if (grouped && !FS_HAS_PX<I, O, K, A1, H, A2>) return CRUX_OK; \
    *handled = true; if (probe) return CRUX_OK; return launch_fs<I, O, K, A1, H, A2>(c, a, form, timing, stream); }
#define FS_CASE(I, O, K, A_) FS_CASE2(I, O, K, A_, 64, A_)
  FS_CASE(4, 2, MFK_CATEGORICAL, CRUX_ACT_RELU)     // C2 actor  (PPO CartPole)
  FS_CASE(4, 1, MFK_VALUE, CRUX_ACT_RELU)           // C2 critic
  FS_CASE(3, 1, MFK_GAUSSIAN, CRUX_ACT_RELU)        // Pendulum actor
  FS_CASE(3, 1, MFK_VALUE, CRUX_ACT_RELU)           // Pendulum critic
  FS_CASE(17, 6, MFK_GAUSSIAN, CRUX_ACT_RELU)       // C5 actor  (PPO HalfCheetah-shaped, 17 obs / 6 act)
  FS_CASE(17, 6, MFK_GAUSSIAN, CRUX_ACT_TANH)
  FS_CASE(17, 1, MFK_VALUE, CRUX_ACT_RELU)          // C5 critic
  FS_CASE(17, 1, MFK_VALUE, CRUX_ACT_TANH)
  FS_CASE(8, 4, MFK_CATEGORICAL, CRUX_ACT_RELU)     // 8 observations / 4 discrete actions (LunarLander-shaped)
  FS_CASE(8, 1, MFK_VALUE, CRUX_ACT_RELU)
  FS_CASE(2, 1, MFK_GAUSSIAN, CRUX_ACT_RELU)        // the reference's own Pendulum examples observe (theta, theta_dot): 2 inputs (examples/rl/pendulum.jl)
  FS_CASE(2, 1, MFK_VALUE, CRUX_ACT_RELU)
  // standard Gym shapes (default form + replica-group form only): Acrobot 6 / 3, MountainCar 2 / 3, LunarLanderContinuous 8 / 2, Hopper 11 / 3, BipedalWalker 24 / 4, Ant 27 / 8
  FS_CASE(6, 3, MFK_CATEGORICAL, CRUX_ACT_RELU)  FS_CASE(6, 1, MFK_VALUE, CRUX_ACT_RELU)
  FS_CASE(2, 3, MFK_CATEGORICAL, CRUX_ACT_RELU)
  FS_CASE(8, 2, MFK_GAUSSIAN, CRUX_ACT_RELU)     FS_CASE(8, 2, MFK_GAUSSIAN, CRUX_ACT_TANH)     FS_CASE(8, 1, MFK_VALUE, CRUX_ACT_TANH)
  FS_CASE(11, 3, MFK_GAUSSIAN, CRUX_ACT_RELU)    FS_CASE(11, 3, MFK_GAUSSIAN, CRUX_ACT_TANH)    FS_CASE(11, 1, MFK_VALUE, CRUX_ACT_RELU)   FS_CASE(11, 1, MFK_VALUE, CRUX_ACT_TANH)
  FS_CASE(24, 4, MFK_GAUSSIAN, CRUX_ACT_RELU)    FS_CASE(24, 4, MFK_GAUSSIAN, CRUX_ACT_TANH)    FS_CASE(24, 1, MFK_VALUE, CRUX_ACT_RELU)   FS_CASE(24, 1, MFK_VALUE, CRUX_ACT_TANH)
  FS_CASE(27, 8, MFK_GAUSSIAN, CRUX_ACT_RELU)    FS_CASE(27, 8, MFK_GAUSSIAN, CRUX_ACT_TANH)    FS_CASE(27, 1, MFK_VALUE, CRUX_ACT_RELU)   FS_CASE(27, 1, MFK_VALUE, CRUX_ACT_TANH)
  // the reference's HalfCheetah PPO networks (examples/rl/half_cheetah_mujoco.jl:33-38): mu = 17 -tanh-> 64 -tanh-> 32 -> 6, V = 17 -tanh-> 64 -> 32 -> 1 (no activation on V's second layer)
  FS_CASE2(17, 6, MFK_GAUSSIAN, CRUX_ACT_TANH, 32, CRUX_ACT_TANH)
  FS_CASE2(17, 1, MFK_VALUE, CRUX_ACT_TANH, 32, CRUX_ACT_IDENTITY)
  FS_CASE2(17, 1, MFK_VALUE, CRUX_ACT_TANH, 32, CRUX_ACT_TANH)
#undef FS_CASE
#undef FS_CASE2
  return CRUX_OK;
}
