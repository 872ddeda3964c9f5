// train_mfma8.hip -- dispatch of the one-CU form of the register-resident learner kernel (train_mfma_kernel.h: <NW = 8, NWG = 1>): one workgroup of eight
// waves, a 16-sample tile per wave. Single steps, gradient-only calls and small minibatches of the narrow-input shapes, and the population form (one CU per
// learner keeps the most learners resident: bench.py's multi_seed_one_cu_per_learner line).
#include "train_mfma_kernel.h"

// ---- dispatch ---------------------------------------------------------------------------------------------------
template <int IN, int OUT, int KIND, int ACT>
static int32_t launch_one8(crux_ctx* c, const TrainArgs& a, hipStream_t stream) {
  using Lt = MfLayout<IN, OUT, 8>;
  constexpr size_t lds = sizeof(float) * (size_t)Lt::TOTAL;
  static bool attr_dev[16] = {}; bool& attr = attr_dev[c->device & 15];      // (per device: a second device in the process sets the attribute for itself)
  if (!attr) { HIPCHK(c, hipFuncSetAttribute((const void*)k_train_mfma<IN, OUT, KIND, ACT, 8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; }
  hipLaunchKernelGGL((k_train_mfma<IN, OUT, KIND, ACT, 8, 1>), dim3(1), dim3(512), lds, stream, a, (const TrainArgs*)nullptr);
  return crux_launch_check(c, "k_train_mfma<8,1>");
}

// n independent learners, one CU each (grid n): argument blocks uploaded to a per-stream device array
template <int IN, int OUT, int KIND, int ACT>
static int32_t launch_multi8(crux_ctx* c, std::vector<TrainArgs>& as, hipStream_t stream) {
  using Lt = MfLayout<IN, OUT, 8>;
  constexpr size_t lds = sizeof(float) * (size_t)Lt::TOTAL;
  static bool attr_dev[16] = {}; bool& attr = attr_dev[c->device & 15];      // (per device: a second device in the process sets the attribute for itself)
  if (!attr) { HIPCHK(c, hipFuncSetAttribute((const void*)k_train_mfma<IN, OUT, KIND, ACT, 8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; }
  const int which = stream == c->stream ? 0 : 1; const size_t n = as.size(), need = n * sizeof(TrainArgs) + 256;
  if (c->amulti_bytes[which] < need) {
    if (c->amulti[which]) { HIPCHK(c, hipDeviceSynchronize()); (void)hipFree(c->amulti[which]); c->amulti[which] = nullptr; c->amulti_bytes[which] = 0; }
    if (hipMalloc(&c->amulti[which], need) != hipSuccess) return crux_fail(c, CRUX_ENOMEM, "multi-learner argument blocks (%zu bytes)", need);
    c->amulti_bytes[which] = need;
  }
  TrainArgs* d_args = (TrainArgs*)c->amulti[which];
  HIPCHK(c, hipMemcpyAsync(d_args, as.data(), n * sizeof(TrainArgs), hipMemcpyHostToDevice, stream));
  HIPCHK(c, hipStreamSynchronize(stream));      // `as` is pageable host memory: the copy must have left it before the caller's vector can change
  hipLaunchKernelGGL((k_train_mfma<IN, OUT, KIND, ACT, 8, 1>), dim3((unsigned)n), dim3(512), lds, stream, as[0], (const TrainArgs*)d_args);
  return crux_launch_check(c, "k_train_mfma<8,1> (multi)");
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
  if (crux_sw().mfma_timing && in == 4 && out == 2 && kind == MFK_CATEGORICAL && act == CRUX_ACT_RELU) {      // CRUX_MFMA_TIMING=1: per-phase s_memtime totals of the C2 actor (development)
    using Lt = MfLayout<4, 2, 8>; constexpr size_t lds = sizeof(float) * (size_t)Lt::TOTAL;
    static unsigned long long* dbg = nullptr;
    if (!dbg) { if (hipMalloc(&dbg, 128 * 8) != hipSuccess) return crux_fail(c, CRUX_ENOMEM, "timing buffer"); }
    TrainArgs b = a; b.dbg = dbg; *handled = true;
    HIPCHK(c, hipFuncSetAttribute((const void*)k_train_mfma<4, 2, MFK_CATEGORICAL, CRUX_ACT_RELU, 8, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_train_mfma<4, 2, MFK_CATEGORICAL, CRUX_ACT_RELU, 8, 1, true>), dim3(1), dim3(512), lds, stream, b, (const TrainArgs*)nullptr);
    int32_t rc = crux_launch_check(c, "k_train_mfma<8,1,timing>"); if (rc) return rc;
    unsigned long long h[128]; HIPCHK(c, hipMemcpyAsync(h, dbg, sizeof h, hipMemcpyDeviceToHost, stream)); HIPCHK(c, hipStreamSynchronize(stream));
    static const char* nm[16] = {"loop+prefetch", "stage", "fwdL1+T1", "fwdL2", "L3+head", "dW3+dZ2+stats+T2", "dH1", "dZ1+db+dW1", "wait B_a", "dW2", "reduce+store", "exchange wait",
                                 "load peer+total+ssq", "wait B_or", "info+adam", "wait B_b"};
    for (int w = 0; w < 8; w += 3) { fprintf(stderr, "[one-cu-timing] wave %d:", w); unsigned long long tot = 0; for (int k = 0; k < 16; ++k) tot += h[w * 16 + k];
      for (int k = 0; k < 16; ++k) fprintf(stderr, " %s=%.1f%%", nm[k], 100.0 * (double)h[w * 16 + k] / (double)tot); fprintf(stderr, " total=%llu\n", tot); }
    return CRUX_OK;
  }
#define MF8_CASE(I, O, K, A_) if (in == I && out == O && kind == K && act == A_) { *handled = true; return launch_one8<I, O, K, A_>(c, a, stream); }
  MF8_CASE(4, 2, MFK_CATEGORICAL, CRUX_ACT_RELU)     // C2 actor  (PPO CartPole)
  MF8_CASE(4, 1, MFK_VALUE, CRUX_ACT_RELU)           // C2 critic
  MF8_CASE(3, 1, MFK_GAUSSIAN, CRUX_ACT_RELU)        // Pendulum actor
  MF8_CASE(3, 1, MFK_VALUE, CRUX_ACT_RELU)           // Pendulum critic
#undef MF8_CASE
  return CRUX_OK;
}
