// train_mfma.hip -- dispatch of batch_train! / train! (src/training.jl:13-55) onto the MFMA learner kernels of the IN->64->64->OUT family.
//
// ONE kernel template (train_mfma_kernel.h: k_train_mfma<IN, OUT, KIND, ACT, NW, NWG, ...>) in two forms: exact-f32 v_mfma_f32_16x16x4_f32 GEMMs for the
// 64-wide layers, the last layer and the loss head on the VALU, the 64x64 weight gradient model-parallel over waves with theta / m / v of W2 living
// in the owning wave's registers for the whole launch.
//   * <NW = 4, NWG = 2> (train_mfma_x2.hip): one learner on TWO compute units of an XCD; minibatches of 65..128 rows (the batch_train! fast path),
//     and -- for the shapes the one-CU form does not instantiate -- also single steps, gradient-only calls and small minibatches.
//   * <NW = 8, NWG = 1> (train_mfma8.hip): one learner on ONE compute unit with 8 waves; narrow inputs (IN <= 4), any minibatch up to 128 rows,
//     single steps (train!, crux_loss_grad), and the population launches where compute units are the scarce resource.
// (Round 1 had three separate copies of the step body -- a 4-wave one-CU kernel, the 8-wave one and the two-CU one; they are this one template now.)
#include "train_args.h"

#include "mfma_helpers.h"

int32_t crux_train_mfma8_launch(crux_ctx* c, const TrainArgs& a, int kind, bool* handled, hipStream_t stream);   // train_mfma8.hip
int32_t crux_train_mfma_x2_launch(crux_ctx* c, const TrainArgs& a, int kind, bool* handled, hipStream_t stream, bool any_mode);   // train_mfma_x2.hip

int32_t crux_train_fs_launch(crux_ctx* c, const TrainArgs& a, bool* handled, hipStream_t stream, bool probe = false);      // train_fs2.hip: the feature-split form (its own shape test: IN -> 64 -> {64, 32} -> OUT)
int32_t crux_train_mfma_launch(crux_ctx* c, const TrainArgs& a, bool* handled, hipStream_t stream) {
  *handled = false;
  const NetDesc& nd = a.nd;
  if (crux_sw().force_generic) return CRUX_OK;      // parity tests run the same cases through the generic learner
  // 0. full minibatch loops (with or without a replica group): the feature-split kernel on four compute units
  { const int32_t rc = crux_train_fs_launch(c, a, handled, stream); if (rc || *handled) return rc; }
  if (nd.L != 3 || nd.dims[1] != MF_HID || nd.dims[2] != MF_HID || nd.acts[2] != CRUX_ACT_IDENTITY || nd.acts[0] != nd.acts[1]) return CRUX_OK;
  if (a.bs > 128 || a.loss == CRUX_LOSS_TD_INTERNAL || a.loss == CRUX_LOSS_MSE_ACTION) return CRUX_OK;   // those two heads exist in the generic kernel only
  if (a.ids && a.n_ids > 128) return CRUX_OK;
  int kind;
  if (a.loss == CRUX_LOSS_VALUE_MSE) kind = MFK_VALUE;
  else if (a.head == CRUX_HEAD_CATEGORICAL) kind = MFK_CATEGORICAL;
  else if (a.head == CRUX_HEAD_GAUSSIAN) kind = MFK_GAUSSIAN;
  else return CRUX_OK;
  // 0. lagrange_ppo_loss (ppo.jl:70-131): the penalty controller and the cost term exist in the two-CU form only (its own instantiations, any call shape)
  if (a.lag) { if (kind == MFK_VALUE || a.loss != CRUX_LOSS_PPO) return CRUX_OK;      // (crux_batch_train_lagrange passes the PPO head with the controller attached)
    return crux_train_mfma_x2_launch(c, a, kind, handled, stream, /*any_mode=*/true); }
  // 1. the two-CU kernel for full minibatch loops (and whenever a replica group needs its in-kernel exchange)
  if (c->learner_cus != 1 || a.need_px) { const int32_t rc = crux_train_mfma_x2_launch(c, a, kind, handled, stream, /*any_mode=*/false); if (rc || *handled) return rc; }
  if (a.need_px) return CRUX_OK;      // not covered by the two-CU kernel: the caller refuses (no un-synchronised training)
  // 2. the one-CU kernel: narrow inputs, single steps, small minibatches
  { const int32_t rc = crux_train_mfma8_launch(c, a, kind, handled, stream); if (rc || *handled) return rc; }
  // 3. shapes only the two-CU kernel instantiates (17-wide, 8-wide): it also takes their single steps / gradient-only calls / small minibatches
  return crux_train_mfma_x2_launch(c, a, kind, handled, stream, /*any_mode=*/true);
}
