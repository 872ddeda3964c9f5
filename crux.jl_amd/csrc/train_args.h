// train_args.h -- argument block shared by the learner kernels (generic and MFMA).
#pragma once
#include "common.h"

#define CRUX_LOSS_TD_INTERNAL 2
#define CRUX_TRAIN_ABORTED (-99)   // status[0] of a learner launch that left at an epoch boundary because its speculative start order turned out wrong (never crosses the ABI)
#define CRUX_IS_PG(l) ((l) == CRUX_LOSS_PPO || (l) == CRUX_LOSS_A2C)   // policy-gradient losses share the head, statistics and KL early stopping   // td_loss (src/utils.jl:76-87); reached through crux_td_step

struct TrainArgs {
  NetDesc nd;
  float* p; float* g; float* m; float* v; double* bp;
  double eta, b1, b2, eps;
  const float* S; const void* A; const float* LP; const float* ADV; const float* RET; const float* Y; const float* Wt;
  int32_t od, ad, act_kind;
  int32_t loss, head, bs, epochs;
  long long max_batches;
  float eps_clip, lambda_p, lambda_e, target_kl;
  uint64_t shuffle_seed, shuffle_counter;
  int64_t len;
  int32_t* order_a; int32_t* order_b;
  const int64_t* perms;      // device [epochs x len] or NULL
  const int32_t* ord_all;    // device [epochs x len]: the composed shuffle order of every epoch, built before the launch (k_compose_order); NULL = compose in the kernel
  // shuffles that a PRECEDING batch_train! applied to the same buffer (actor before critic, on_policy.jl:65-69): composed into
  // the starting order so this learner can run concurrently with the preceding one and still see the reference's row order
  int32_t pre_epochs; uint64_t pre_seed, pre_counter; const int64_t* pre_perms;
  const int32_t* ids;        // device explicit rows (single-step mode) or NULL
  int64_t n_ids;
  int32_t apply;             // 1: Adam update after each minibatch
  float* epoch_infos;        // device [epochs x CRUX_INFO_N]
  int32_t* status;           // device [4]: err, batches_trained, epochs_run, final order selector
  unsigned long long* dbg;   // optional phase-timing output (CRUX_MFMA_TIMING)
  float squash;              // SquashedGaussianPolicy ascale (0 = GaussianPolicy): actions are un-tanh'd, sigma uses clamp(logSigma, -5, 2), logpdf carries the tanh correction
  float* xbuf; unsigned* xctr;   // two-CU kernel (train_mfma_x2.hip): gradient exchange slots [parity][workgroup] and {arrival counter, abort flag}
  // replica group (comm.hip "peer"): SUM all-reduce of the minibatch gradient over px_n GPUs inside the persistent kernel, between the pullback
  // (training.jl:18) and Flux.update! (:21). px_tab[r] = rank r's slot region for THIS learner stream as mapped in this process.
  int32_t need_px;           // host-side: a replica group is attached and this launch updates parameters -> only the two-CU kernel may take it
  int32_t xcd;               // feature-split kernel: the XCD (blockIdx mod 8) whose compute units this learner takes -- actor and critic (and same-device replicas) sit behind different L2s
  int32_t px_hist;           // 1: lane 0 of each workgroup bins its flag wait of every exchange into the own region's histogram (CRUX_PX_HIST)
  int32_t px_every;          // <= 1: the gradient is exchanged every minibatch; k > 1: local Adam steps, theta / m / v are averaged after every k-th (crux_peer_set_sync_every)
  long long px_timeout;      // flag-wait timeout of one exchange in 10 ns wall-clock ticks (crux_peer_set_timeout_ms): a missing peer becomes CRUX_EHIP instead of a hung GPU
  int32_t px_n, px_rank; float* const* px_tab;   // px_tab: device table [px_n] of the ranks' region bases for this learner stream (own region at px_rank)
  // lagrange_ppo_loss (rl/ppo.jl:70-131): device copy of crux_lagrange (hyper-parameters + PID state), the cost columns; NULL = plain ppo_loss
  crux_lagrange* lag; const float* COST; const float* CADV; const uint8_t* EE;
  // packed learner rows (train.hip: ensure_pack): [s (od) | action (index, or ad floats) | logprob | advantage | return] of every buffer row in ONE line of
  // pack_stride floats (a power of two), written once per batch_train! call. NULL = gather the pieces from the SoA columns (68 / 24 / 4 / 4-byte pieces cost a sector each:
  // 60.7 KB of memory-side traffic per C5 minibatch step against 13.3 KB algorithmic).
  const float* PACK; int32_t pack_stride, pack_act, pack_lp;
  // speculative launch (train.hip: policy_gradient_training with KL early stopping): host-pinned word the kernel looks at once per epoch; non-zero = leave at this epoch
  // boundary with status CRUX_TRAIN_ABORTED (decided once for all workgroups of the learner through a latch in xctr). NULL = never.
  const unsigned* spec_abort;
  void* host_net;            // host-side: the crux_mlp this block was filled from (the dense-engine learner drives its GEMM workspace); never read on the device
};

// The minibatch rows are device global memory. Typing the per-step loads as address_space(1) makes them global_load instead of the flat_load a
// generic pointer compiles to: flat accesses also count on lgkmcnt, so every LDS wait after the prefetch of the next minibatch was issued
// waited for that prefetch as well (an L2 round trip per step), and their 64-bit addresses cost extra VALU adds.
#if defined(__HIP_DEVICE_COMPILE__)
#define CRUX_GLOBAL_PTR(T, ptr) ((const __attribute__((address_space(1))) T*)(ptr))
#else
#define CRUX_GLOBAL_PTR(T, ptr) ((const T*)(ptr))
#endif
