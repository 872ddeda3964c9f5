// train_mfma_x2.hip -- dispatch of the two-CU form of the register-resident learner kernel (train_mfma_kernel.h: <NW = 4, NWG = 2>).
#include "train_mfma_kernel.h"

// ---- dispatch ---------------------------------------------------------------------------------------------------
// Placement probe: consecutive workgroups of a grid must land on XCDs round-robin (blockIdx i and i+8 on the same XCD). Checked
// once per context with the hardware XCC_ID register; if the assumption does not hold the two-CU kernel is never used.
__global__ void k_xcc_probe(uint32_t* out) {
  uint32_t id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  if (threadIdx.x == 0) out[blockIdx.x] = id & 0xf;
}
extern "C" int crux_x2_placement_ok(crux_ctx* c);
static int x2_placement_ok(crux_ctx* c) { return crux_x2_placement_ok(c); }
// also called when a replica group is attached: the probe ends in a hipFree, which waits for the whole device -- it must not run for the first time
// while a peer replica on the same device already spins in its learner kernel
extern "C" int crux_x2_placement_ok(crux_ctx* c) {
  static int cached = -1;
  if (cached >= 0) return cached;
  cached = 0;
  uint32_t* d = nullptr; uint32_t h[32];
  if (hipMalloc(&d, sizeof h) != hipSuccess) return 0;
  bool ok = true;
  for (int rep = 0; rep < 3 && ok; ++rep) {
    hipLaunchKernelGGL(k_xcc_probe, dim3(32), dim3(64), 0, c->stream, d);
    if (hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { ok = false; break; }
    for (int i = 0; i + 8 < 32; ++i) ok = ok && h[i] == h[i + 8];
    ok = ok && h[0] != h[1];
  }
  (void)hipFree(d);
  cached = ok ? 1 : 0;
  if (!ok && crux_sw().verbose) fprintf(stderr, "[cruxhip] workgroups are not placed round-robin over XCDs: two-CU learner kernel disabled\n");
  return cached;
}

template <int IN, int OUT, int KIND, int ACT, bool TIMING, bool PX, bool LAG = false>
static int32_t launch_x2_form(crux_ctx* c, const TrainArgs& a, size_t lds, hipStream_t stream) {
  static bool attr_dev[16] = {}; bool& attr = attr_dev[c->device & 15];      // (per device: a second device in the process sets the attribute for itself)
  if (!attr) { HIPCHK(c, hipFuncSetAttribute((const void*)k_train_mfma<IN, OUT, KIND, ACT, 4, 2, TIMING, PX, LAG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; }
  hipLaunchKernelGGL((k_train_mfma<IN, OUT, KIND, ACT, 4, 2, TIMING, PX, LAG>), dim3(16), dim3(256), lds, stream, a, (const TrainArgs*)nullptr);
  return crux_launch_check(c, "k_train_mfma<4,2>");
}
template <int IN, int OUT, int KIND, int ACT, bool TIMING = false>
static int32_t launch_x2(crux_ctx* c, TrainArgs a, hipStream_t stream) {
  using Lt = MfLayout<IN, OUT, 4>;
  static_assert(Lt::FITS, "x2 layout must fit");
  constexpr size_t lds = sizeof(float) * (size_t)Lt::TOTAL;
  // exchange area: one per stream the learners run on (actor || critic use the context's two streams concurrently)
  const int which = stream == c->stream ? 0 : 1;
  constexpr size_t xbytes = sizeof(float) * CRUX_XBUF_FLOATS + 256;
  if (!c->xbuf[which]) { if (hipMalloc(&c->xbuf[which], xbytes) != hipSuccess) return crux_fail(c, CRUX_ENOMEM, "learner exchange buffer"); }
  a.xbuf = (float*)c->xbuf[which]; a.xctr = (unsigned*)((char*)c->xbuf[which] + sizeof(float) * CRUX_XBUF_FLOATS);
  HIPCHK(c, hipMemsetAsync(a.xctr, 0, 256, stream));
  if (crux_grouped(c) && a.need_px) {     // local calls (single steps, gradients) never exchange
    a.px_hist = c->peer_hist ? 1 : 0; a.px_n = c->peer_n; a.px_rank = c->peer_rank; a.px_tab = c->peer_tab + which * CRUX_PX_MAXR;
    return launch_x2_form<IN, OUT, KIND, ACT, false, true>(c, a, lds, stream);
  }
  if constexpr (KIND != MFK_VALUE && !TIMING) { if (a.lag) return launch_x2_form<IN, OUT, KIND, ACT, false, false, true>(c, a, lds, stream); }      // lagrange_ppo_loss (ppo.jl:70-131)
  return launch_x2_form<IN, OUT, KIND, ACT, TIMING, false>(c, a, lds, stream);
}

// n independent learners in one launch (grid 16 n): argument blocks uploaded to a per-stream device array, one exchange area each
template <int IN, int OUT, int KIND, int ACT>
static int32_t launch_x2_multi(crux_ctx* c, std::vector<TrainArgs>& as, hipStream_t stream) {
  using Lt = MfLayout<IN, OUT, 4>;
  constexpr size_t lds = sizeof(float) * (size_t)Lt::TOTAL;
  static bool attr_dev[16] = {}; bool& attr = attr_dev[c->device & 15];      // (per device: a second device in the process sets the attribute for itself)
  if (!attr) { HIPCHK(c, hipFuncSetAttribute((const void*)k_train_mfma<IN, OUT, KIND, ACT, 4, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; }
  const int which = stream == c->stream ? 0 : 1; const size_t n = as.size();
  constexpr size_t xbytes = sizeof(float) * 4 * 8192 + 256;
  const size_t need = n * xbytes + n * sizeof(TrainArgs) + 256;
  if (c->xmulti_bytes[which] < need) {
    if (c->xmulti[which]) { HIPCHK(c, hipDeviceSynchronize()); (void)hipFree(c->xmulti[which]); c->xmulti[which] = nullptr; c->xmulti_bytes[which] = 0; }
    if (hipMalloc(&c->xmulti[which], need) != hipSuccess) return crux_fail(c, CRUX_ENOMEM, "multi-learner exchange area (%zu bytes)", need);
    c->xmulti_bytes[which] = need;
  }
  char* base = (char*)c->xmulti[which];
  for (size_t i = 0; i < n; ++i) { as[i].xbuf = (float*)(base + i * xbytes); as[i].xctr = (unsigned*)(base + i * xbytes + sizeof(float) * 4 * 8192);
    HIPCHK(c, hipMemsetAsync(as[i].xctr, 0, 256, stream)); }
  TrainArgs* d_args = (TrainArgs*)(base + n * xbytes);
  HIPCHK(c, hipMemcpyAsync(d_args, as.data(), n * sizeof(TrainArgs), hipMemcpyHostToDevice, stream));
  HIPCHK(c, hipStreamSynchronize(stream));      // `as` is pageable host memory: the copy must have left it before the caller's vector can change
  hipLaunchKernelGGL((k_train_mfma<IN, OUT, KIND, ACT, 4, 2, false>), dim3((unsigned)(16 * n)), dim3(256), lds, stream, as[0], (const TrainArgs*)d_args);
  return crux_launch_check(c, "k_train_mfma<4,2> (multi)");
}

int32_t crux_train_mfma_x2_launch_multi(crux_ctx* c, std::vector<TrainArgs>& as, bool* handled, hipStream_t stream) {
  *handled = false;
  if (as.empty() || !x2_placement_ok(c)) return CRUX_OK;
  const TrainArgs& a = as[0];
  const NetDesc& nd = a.nd;
  if (nd.L != 3 || nd.dims[1] != MF_HID || nd.dims[2] != MF_HID || nd.acts[2] != CRUX_ACT_IDENTITY || nd.acts[0] != nd.acts[1] || a.ids || !a.apply || a.bs <= 64 || a.bs > 128 || a.len < a.bs) return CRUX_OK;
  const int in = nd.dims[0], out = nd.dims[3], act = nd.acts[0];
  const int kind = a.loss == CRUX_LOSS_VALUE_MSE ? MFK_VALUE : (a.head == CRUX_HEAD_CATEGORICAL ? MFK_CATEGORICAL : (a.head == CRUX_HEAD_GAUSSIAN ? MFK_GAUSSIAN : -1));
  if (!(a.loss == CRUX_LOSS_VALUE_MSE || CRUX_IS_PG(a.loss)) || kind < 0) return CRUX_OK;
#define MFXM_CASE(I, O, K, A_) if (in == I && out == O && kind == K && act == A_) { *handled = true; return launch_x2_multi<I, O, K, A_>(c, as, stream); }
  MFXM_CASE(4, 2, MFK_CATEGORICAL, CRUX_ACT_RELU)
  MFXM_CASE(4, 1, MFK_VALUE, CRUX_ACT_RELU)
  MFXM_CASE(17, 6, MFK_GAUSSIAN, CRUX_ACT_TANH)
  MFXM_CASE(17, 1, MFK_VALUE, CRUX_ACT_TANH)
#undef MFXM_CASE
  return CRUX_OK;
}

// Called by crux_train_mfma_launch after its shape checks, before the single-CU kernels.
// any_mode = false: only full minibatch loops of 65..128 rows (where two CUs pay); true: also single steps, gradient-only calls and small minibatches
// (workgroup 1 then idles on empty tiles) -- used for the shapes that have no one-CU instantiation.
int32_t crux_train_mfma_x2_launch(crux_ctx* c, const TrainArgs& a, int kind, bool* handled, hipStream_t stream, bool any_mode) {
  *handled = false;
  const bool off = !crux_sw().mfma_x2;
  if (off) return CRUX_OK;
  if (!any_mode && (a.ids || !a.apply || a.bs <= 64 || a.len < a.bs)) return CRUX_OK;     // single steps and small batches stay on one CU when it has the shape
  if (!x2_placement_ok(c)) return CRUX_OK;
  const int in = a.nd.dims[0], out = a.nd.dims[3], act = a.nd.acts[0];
  if (a.lag) {     // lagrange_ppo_loss: its own instantiations, for the shapes below; replica groups and every other shape stay on the dense-engine learner
    if (crux_grouped(c) && a.need_px) return CRUX_OK;
#define MFXL_CASE(I, O, K, A_) if (in == I && out == O && kind == K && act == A_) { *handled = true; return launch_x2<I, O, K, A_>(c, a, stream); }
    MFXL_CASE(4, 2, MFK_CATEGORICAL, CRUX_ACT_RELU)
    MFXL_CASE(8, 4, MFK_CATEGORICAL, CRUX_ACT_RELU)
    MFXL_CASE(3, 1, MFK_GAUSSIAN, CRUX_ACT_RELU)
    MFXL_CASE(17, 6, MFK_GAUSSIAN, CRUX_ACT_TANH)
#undef MFXL_CASE
    return CRUX_OK;
  }
  if (crux_sw().mfma_timing && ((in == 4 && out == 2 && kind == MFK_CATEGORICAL && act == CRUX_ACT_RELU) || (in == 17 && out == 6 && kind == MFK_GAUSSIAN && act == CRUX_ACT_TANH))) {
    static unsigned long long* dbg = nullptr;
    if (!dbg) { if (hipMalloc(&dbg, 128 * 8) != hipSuccess) return crux_fail(c, CRUX_ENOMEM, "timing buffer"); }
    TrainArgs b = a; b.dbg = dbg; *handled = true;
    int32_t rc = in == 4 ? launch_x2<4, 2, MFK_CATEGORICAL, CRUX_ACT_RELU, true>(c, b, stream) : launch_x2<17, 6, MFK_GAUSSIAN, CRUX_ACT_TANH, true>(c, b, stream); if (rc) return rc;
    unsigned long long h[128]; HIPCHK(c, hipMemcpyAsync(h, dbg, sizeof h, hipMemcpyDeviceToHost, stream)); HIPCHK(c, hipStreamSynchronize(stream));
    static const char* nm[16] = {"loop+prefetch", "stage", "fwdL1+T1", "fwdL2", "L3+head", "dW3+dZ2+stats+T2", "dH1", "dZ1+db+dW1", "wait B_a", "dW2", "reduce+store", "exchange wait",
                                 "load peer+total+ssq", "wait B_or", "info+adam", "wait B_b"};
    for (int w = 0; w < 8; w += 4) { fprintf(stderr, "[x2-timing] %d-%d wg %d wave %d:", in, out, w >> 2, w & 3); unsigned long long tot = 0; for (int k = 0; k < 16; ++k) tot += h[w * 16 + k];
      for (int k = 0; k < 16; ++k) fprintf(stderr, " %s=%.1f%%", nm[k], 100.0 * (double)h[w * 16 + k] / (double)tot); fprintf(stderr, " total=%llu\n", tot); }
    return CRUX_OK;
  }
#define MFX_CASE(I, O, K, A_) if (in == I && out == O && kind == K && act == A_) { *handled = true; return launch_x2<I, O, K, A_>(c, a, stream); }
  MFX_CASE(4, 2, MFK_CATEGORICAL, CRUX_ACT_RELU)     // C2 actor  (PPO CartPole)
  MFX_CASE(4, 1, MFK_VALUE, CRUX_ACT_RELU)           // C2 critic
  MFX_CASE(3, 1, MFK_GAUSSIAN, CRUX_ACT_RELU)        // Pendulum actor
  MFX_CASE(3, 1, MFK_VALUE, CRUX_ACT_RELU)           // Pendulum critic
  MFX_CASE(17, 6, MFK_GAUSSIAN, CRUX_ACT_RELU)       // C5 actor  (PPO HalfCheetah-shaped, 17 obs / 6 act): 4 waves per workgroup keep the 17-wide layout inside 160 KB
  MFX_CASE(17, 6, MFK_GAUSSIAN, CRUX_ACT_TANH)
  MFX_CASE(17, 1, MFK_VALUE, CRUX_ACT_RELU)          // C5 critic
  MFX_CASE(17, 1, MFK_VALUE, CRUX_ACT_TANH)
  MFX_CASE(8, 4, MFK_CATEGORICAL, CRUX_ACT_RELU)     // 8 observations / 4 discrete actions (LunarLander-shaped, the C3 environment under an on-policy learner)
  MFX_CASE(8, 1, MFK_VALUE, CRUX_ACT_RELU)
#undef MFX_CASE
  return CRUX_OK;
}
