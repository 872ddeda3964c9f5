// train_mfma.hip -- MFMA learner / value kernels for the 64-wide MLP family (placeholder until the kernel lands).
#include "train_args.h"

int32_t crux_train_mfma_launch(crux_ctx* c, const TrainArgs& a, bool* handled) { (void)c; (void)a; *handled = false; return CRUX_OK; }
int32_t crux_values_fast(crux_mlp* net, const float* d_x, int64_t B, float* d_y) { (void)net; (void)d_x; (void)B; (void)d_y; return CRUX_EUNSUP; }
