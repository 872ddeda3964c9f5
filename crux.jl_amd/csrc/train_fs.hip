// train_fs.hip -- dispatch of the feature-split form of the register-resident learner kernel (train_fs_kernel.h: k_train_fs<IN, OUT, KIND, ACT, NWG>): full batch_train!
// loops (src/training.jl:28-55) of the IN->64->64->OUT family with minibatches of 65..128 rows, on two compute units with eight waves each or on four with four.
#include "train_fs_kernel.h"

extern "C" int crux_x2_placement_ok(crux_ctx* c);      // train_mfma_x2.hip: workgroups i and i + 8 of a grid share an XCD (probed once per process)

template <int IN, int OUT, int KIND, int ACT, int NWG, bool HELP, bool TIMING>
static int32_t launch_fs_form(crux_ctx* c, TrainArgs& a, hipStream_t stream) {
  using Lt = FsLayout<IN, OUT, NWG, HELP>;
  constexpr size_t lds = sizeof(float) * (size_t)Lt::TOTAL;
  static bool attr = false;
  if (!attr) { HIPCHK(c, hipFuncSetAttribute((const void*)k_train_fs<IN, OUT, KIND, ACT, NWG, HELP, TIMING>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; }
  hipLaunchKernelGGL((k_train_fs<IN, OUT, KIND, ACT, NWG, HELP, TIMING>), dim3(8 * NWG), dim3(Lt::NT), lds, stream, a);
  return crux_launch_check(c, "k_train_fs");
}
// form: 2 = two workgroups of eight waves; 4 = four workgroups of four waves; 8 = four workgroups of four compute + four helper waves
template <int IN, int OUT, int KIND, int ACT, bool TIMING>
static int32_t launch_fs_pick(crux_ctx* c, TrainArgs& a, int form, hipStream_t stream) {
  if (form == 2) return launch_fs_form<IN, OUT, KIND, ACT, 2, false, TIMING>(c, a, stream);
  if (form == 4) return launch_fs_form<IN, OUT, KIND, ACT, 4, false, TIMING>(c, a, stream);
  return launch_fs_form<IN, OUT, KIND, ACT, 4, true, TIMING>(c, a, stream);
}
template <int IN, int OUT, int KIND, int ACT>
static int32_t launch_fs_px(crux_ctx* c, TrainArgs& a, hipStream_t stream) {      // replica group: four workgroups with helper waves, the in-kernel all-reduce over the peer slots
  using Lt = FsLayout<IN, OUT, 4, true>;
  constexpr size_t lds = sizeof(float) * (size_t)Lt::TOTAL;
  static bool attr = false;
  if (!attr) { HIPCHK(c, hipFuncSetAttribute((const void*)k_train_fs<IN, OUT, KIND, ACT, 4, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; }
  hipLaunchKernelGGL((k_train_fs<IN, OUT, KIND, ACT, 4, true, false, true>), dim3(32), dim3(Lt::NT), lds, stream, a);
  return crux_launch_check(c, "k_train_fs (replica group)");
}
template <int IN, int OUT, int KIND, int ACT>
static int32_t launch_fs(crux_ctx* c, TrainArgs a, int nwg, bool timing, hipStream_t stream) {
  const int which = stream == c->stream ? 0 : 1;
  constexpr size_t xfloats = (size_t)CRUX_XBUF_FLOATS;
  if (!c->xbuf[which]) { if (hipMalloc(&c->xbuf[which], sizeof(float) * xfloats + 256) != hipSuccess) return crux_fail(c, CRUX_ENOMEM, "learner exchange buffer"); }
  a.xbuf = (float*)c->xbuf[which]; a.xctr = (unsigned*)((char*)c->xbuf[which] + sizeof(float) * xfloats);
  HIPCHK(c, hipMemsetAsync(a.xctr, 0, 256, stream));
  a.xcd = which;      // actor / critic (the context's two learner streams) behind different L2s. (Replicas sharing a device stay on the same XCD pair: spreading them over XCDs
                      // sent their flag / slot traffic across L2s and measured 15.5 against 12.9 us per step on one GPU.)
  if (c->peer_n > 1 && a.need_px) {
    a.px_hist = c->peer_hist ? 1 : 0; a.px_n = c->peer_n; a.px_rank = c->peer_rank; a.px_tab = c->peer_tab + which * CRUX_PX_MAXR;
    return launch_fs_px<IN, OUT, KIND, ACT>(c, a, stream);
  }
  constexpr bool HAS_TIMING = (IN == 4 && (OUT == 2 || OUT == 1)) || (IN == 17 && ACT == CRUX_ACT_TANH);      // the in-kernel phase timers are instantiated for the C2 / C5 learners only
  if constexpr (HAS_TIMING) if (timing) {
    static unsigned long long* dbg = nullptr;
    if (!dbg) { if (hipMalloc(&dbg, 512 * 8) != hipSuccess) return crux_fail(c, CRUX_ENOMEM, "timing buffer"); }
    a.dbg = dbg;
    int32_t rc = launch_fs_pick<IN, OUT, KIND, ACT, true>(c, a, nwg, stream); if (rc) return rc;
    unsigned long long h[512]; HIPCHK(c, hipMemcpyAsync(h, dbg, sizeof h, hipMemcpyDeviceToHost, stream)); HIPCHK(c, hipStreamSynchronize(stream));
    static const char* nm[16] = {"loop+prefetch", "stage", "fwdL1+T1", "fwdL2", "L3+z-exchange+head", "dW3+dZ2+stats+T2", "wait B_1", "dW2+send", "dH1", "dZ1+db+dW1", "B_2+reduce+store", "exchange wait",
                                 "load slots+total+ssq", "wait B_or", "info+adam", "wait B_b"};
    const int nw = nwg == 8 ? 8 : 16 / nwg, ntot = nwg == 2 ? 16 : (nwg == 4 ? 16 : 32);
    if (nwg == 8) { fprintf(stderr, "[fs-timing] %d-%d wg 0 helper wave 4:", IN, OUT); unsigned long long tot = 0; for (int k = 0; k < 16; ++k) tot += h[4 * 16 + k];
      for (int k = 0; k < 16; ++k) fprintf(stderr, " %s=%.1f%%", nm[k], 100.0 * (double)h[4 * 16 + k] / (double)tot); fprintf(stderr, " total=%llu\n", tot); }
    for (int w = 0; w < ntot; w += nw) { fprintf(stderr, "[fs-timing] %d-%d wg %d wave 0:", IN, OUT, w / nw); unsigned long long tot = 0; for (int k = 0; k < 16; ++k) tot += h[w * 16 + k];
      for (int k = 0; k < 16; ++k) fprintf(stderr, " %s=%.1f%%", nm[k], 100.0 * (double)h[w * 16 + k] / (double)tot); fprintf(stderr, " total=%llu\n", tot); }
    return CRUX_OK;
  }
  return launch_fs_pick<IN, OUT, KIND, ACT, false>(c, a, nwg, stream);
}

// Called by crux_train_mfma_x2_launch for the plain learners (no replica group, no lagrange loss, no explicit ids). CRUX_FS=0 switches the form off (the sample-split
// two-CU kernel then runs), CRUX_FS_WG=2|4 picks the number of compute units per learner.
int32_t crux_train_fs_launch(crux_ctx* c, const TrainArgs& a, int kind, bool* handled, hipStream_t stream) {
  *handled = false;
  const int mode = [] { const char* e = getenv("CRUX_FS"); return e ? atoi(e) : 1; }();            // read per call: tests switch the form inside one process
  const int nwg_env = [] { const char* e = getenv("CRUX_FS_WG"); return e ? atoi(e) : 0; }();
  if (mode == 0) return CRUX_OK;
  if (a.ids || !a.apply || a.bs <= 64 || a.bs > 128 || a.len < a.bs || a.lag) return CRUX_OK;
  if (!crux_x2_placement_ok(c)) return CRUX_OK;
  const int nwg = nwg_env == 2 || nwg_env == 4 || nwg_env == 8 ? nwg_env : CRUX_FS_DEFAULT_WG;      // 8 = four workgroups with helper waves
  const bool timing = getenv("CRUX_MFMA_TIMING") != nullptr;
  const int in = a.nd.dims[0], out = a.nd.dims[3], act = a.nd.acts[0];
#define FS_CASE(I, O, K, A_) if (in == I && out == O && kind == K && act == A_) { *handled = true; return launch_fs<I, O, K, A_>(c, a, nwg, timing, stream); }
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
#undef FS_CASE
  return CRUX_OK;
}
