// common.h -- internal structures of libcruxhip (gfx950 only; no portability layers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <cstring>
#include "switches.h"
#include <cstdio>
#include <cmath>
#include "../../include/cruxhip.h"
#include "../../include/crux_rng.h"

#define CRUX_MAXL 8

struct crux_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  bool prof_on = false;
  double prof_ms[CRUX_PROF_NSLOTS] = {0};
  int64_t prof_n[CRUX_PROF_NSLOTS] = {0};
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> pending;   // slot -> (start, stop) not yet resolved
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  void* scratch = nullptr; size_t scratch_bytes = 0;   // reusable device scratch
  void* pinned = nullptr; size_t pinned_bytes = 0;     // reusable pinned host staging
  void* pinned_mapped = nullptr; void* pinned_mapped_dev = nullptr; size_t pinned_mapped_bytes = 0;   // reusable pinned, DEVICE-MAPPED staging (kernels read / write it in place: crux_policy_explore, the small-block push!)
  hipStream_t aux_stream = nullptr; hipEvent_t aux_ev0 = nullptr, aux_ev1 = nullptr;   // second learner stream (actor || critic)
  hipStream_t aux_rejected[8] = {}; int aux_n_rejected = 0; float aux_probe_ms = 0.f;   // candidates that shared the main stream's hardware queue (kept alive until destroy)
  void* comm = nullptr; int comm_rank = 0, comm_n = 0;   // RCCL communicator of the replica group (comm.hip)
  void* amulti[2] = {nullptr, nullptr}; size_t amulti_bytes[2] = {0, 0};   // argument blocks of the one-CU batched learner launch
  int learner_cus = 0;                 // 0 = automatic, 1 = one CU per learner (k_train_mfma<...,8,1>), 2 = two CUs (k_train_mfma<...,4,2>)
  void* xmulti[2] = {nullptr, nullptr}; size_t xmulti_bytes[2] = {0, 0};   // exchange areas + argument blocks of the batched multi-learner launch
#define CRUX_XBUF_FLOATS (8 * 10240)    // exchange area of one learner stream: [parity 2][workgroups <= 4][slot <= 10240 floats; x2: <= 8192], followed by 256 bytes of counters
  void* xbuf[2] = {nullptr, nullptr};   // gradient exchange areas of the two-CU learner kernel, one per learner stream
  // replica group with direct peer slots (comm.hip "peer"): every rank owns one fine-grained region that its peers write their minibatch
  // gradients into over xGMI; peer_ptr[r] is rank r's region as mapped here (own region for r == peer_rank)
  int peer_every = 1;                  // crux_peer_set_sync_every: 1 = gradient exchange every minibatch; k > 1 = local Adam steps, theta / m / v averaged after every k-th
  bool peer_hist = false;              // record the per-step flag waits of the replica-group exchange (crux_peer_hist_enable)
  bool peer_solo = false;              // crux_peer_attach(ctx, 0, 1, ...): a group of ONE -- the replica-group instantiations of the learner kernels with no peer (the reference form bit-exact group results are compared with)
  long long peer_timeout_ticks = 3000000000ll;   // in-kernel flag-wait timeout of the replica-group exchange in 10 ns ticks (crux_peer_set_timeout_ms; default 30 s)
  long long peer_budget_ticks = 6000000000ll;    // what the flag waits of ONE learner launch may add up to (crux_peer_set_budget_ms; default 60 s): bounds a group whose replicas answer, but slowly
  unsigned* peer_host = nullptr;                 // host-pinned, device-mapped block: word 0 = the abort word the kernels' slow path polls (crux_peer_abort), words 16.. = results of crux_peer_probe
  void* peer_host_dev = nullptr;                 // its device address
  unsigned long long peer_probe_base = 0;        // rendezvous rounds announced so far (every rank of a group calls crux_peer_probe the same number of times)
  bool peer_probe_bad = false;                   // a probe failed: the ranks' round counters may differ until the next attach
  int peer_n = 0, peer_rank = 0; void* peer_local = nullptr; void* peer_ptr[8] = {}; bool peer_ipc[8] = {}; bool peer_fine = false;
  void* rec = nullptr;                 // ExecRec* (exec.h): the fused-step executor's recording state
  // replica group with another context of THIS process on the same device (crux_peer_attach_local): hipFree waits for the whole device, i.e. for the peer's
  // spinning learner kernel, which waits for this replica -- blocks that must be re-allocated while the group is attached are parked instead and freed at detach
  bool peer_same_device = false;
  void* lag_dev = nullptr;        // device copy of crux_lagrange for crux_batch_train_lagrange
  void* dense_tmp2 = nullptr; size_t dense_tmp2_bytes = 0; void* dense_pinned2 = nullptr;   // the same for a chain on the second learner stream
  void* dense_tmp = nullptr; size_t dense_tmp_bytes = 0;   // minibatch staging of the dense-engine on-policy learner (train_dense.hip)
  void* epoch_tmp = nullptr; size_t epoch_tmp_bytes = 0;   // targets / td errors of the un-fused epoch path
  void* epoch_rows = nullptr; size_t epoch_rows_bytes = 0;      // device info rows of a multi-chain epochs call (exec.hip: one synchronisation per call)
  unsigned* spec_abort = nullptr;   // host-pinned word a speculatively started critic learner polls once per epoch (train.hip: policy_gradient_training under KL early stopping)
  bool per_split_sample = false;   // prioritized_sample! as two launches (search | gather) while set: the two persistent kernels of dqn_persist.h replay them as separate stages
  bool dqp_broken = false;      // the two persistent kernels of dqn_persist.h did not run side by side once: the phase launches from then on
  float** peer_tab = nullptr;   // device [2 learner streams][8]: region base of every rank for that stream (what the kernel indexes)
};

// Layout of a peer region (floats unless noted), per learner stream w in {0,1} at w * CRUX_PX_STREAM_FLOATS:
//   slots   [parity 2][source rank 8][CRUX_PX_SLOT]   the source's local gradient sum + statistics of one minibatch step
//   flags   uint64 [8] at CRUX_PX_FLAGS, 64-byte stride   flag[src] = number of exchanges whose data src has completely written here
//   abort   uint32 at CRUX_PX_ABORT                        set by any rank that gave up waiting: the bound that ended its wait (peer_wait.h: 1 timeout, 2 launch budget,
//                                                          3 passed on, 4 its host called the launch off) or 5 = it left on a NaN step. TERMINAL for the group: cleared by
//                                                          the next crux_peer_attach* only (a rank that left holds other parameters than its peers)
//   count   uint64 at CRUX_PX_COUNT                        exchanges done so far on this stream (local bookkeeping: slot parity and flag values
//                                                          continue across launches, so the double buffering argument holds across them too)
//   waited / habort / budget / probe                       the bounds of peer_wait.h and the rendezvous probe (below)
#define CRUX_PER_PMAX 24   // deepest pairwise-cumsum tree handled incrementally (N < 128 * 2^24)
#define CRUX_PX_MAXR 8
#define CRUX_PX_SEC 8192                        // one payload section (a gradient, or one of theta / m / v in the periodic form)
#define CRUX_PX_SLOT (3 * CRUX_PX_SEC)
#define CRUX_PX_FLAGS (2 * CRUX_PX_MAXR * CRUX_PX_SLOT)
#define CRUX_PX_ABORT (CRUX_PX_FLAGS + 16 * CRUX_PX_MAXR)
#define CRUX_PX_COUNT (CRUX_PX_ABORT + 16)
#define CRUX_PX_HIST (CRUX_PX_COUNT + 16)       // uint32 [2 workgroups][32]: log2 histogram of the flag waits in 10 ns ticks (crux_peer_wait_hist; filled only while enabled)
#define CRUX_PX_WAITED (CRUX_PX_HIST + 64)     // uint32 [8] (+ 8 spare): per workgroup, what is left of the current launch's wait budget in 1.28 us units (peer_wait.h, bound 2)
#define CRUX_PX_HABORT (CRUX_PX_WAITED + 16)   // device pointer (8 bytes) of the owning context's HOST-pinned abort word (crux_peer_abort), written at attach
#define CRUX_PX_BUDGET (CRUX_PX_HABORT + 2)    // int64: the per-launch wait budget in ticks (crux_peer_set_budget_ms), written at attach
#define CRUX_PX_PROBE (CRUX_PX_BUDGET + 14)    // uint64 [8] at 64-byte stride: probe[src] = rendezvous rounds src has announced here (crux_peer_probe)
#define CRUX_PX_STREAM_FLOATS (CRUX_PX_PROBE + 16 * CRUX_PX_MAXR)
#define CRUX_PX_BYTES (2 * CRUX_PX_STREAM_FLOATS * sizeof(float))

static inline bool crux_grouped(const crux_ctx* c) { return c->peer_n > 1 || c->peer_solo; }      // a replica group is attached (a group of one included)
int32_t crux_fail(crux_ctx* ctx, int32_t code, const char* fmt, ...);
void* crux_scratch(crux_ctx* ctx, size_t bytes);       // grows; contents undefined
void* crux_pinned(crux_ctx* ctx, size_t bytes);
void* crux_pinned_mapped(crux_ctx* ctx, size_t bytes);   // grows (synchronising the stream first); host address returned, ctx->pinned_mapped_dev is the device's view
// hipFree waits for the whole device. While contexts of this process form a replica group on ONE device (crux_peer_attach_local: tests, single-GPU
// development), a learner kernel of one replica spins until the others answer -- a device-wide wait issued by another host thread (a growing scratch block,
// or a finalizer of the host language's garbage collector destroying an unrelated handle) would wait for that kernel and stop the thread the kernel is
// waiting for. Every hipFree of the library therefore goes through crux_hip_free: a plain hipFree normally, parked until the last same-device group detaches otherwise.
hipError_t crux_hip_free(void* p);
void crux_same_device_group_enter();
void crux_same_device_group_leave();
void crux_exec_destroy(crux_ctx* ctx);                 // exec.hip: frees the fused-step executor's device / pinned blocks
#define hipFree(p) crux_hip_free((void*)(p))
void crux_sync_before_free(crux_ctx* ctx);           // hipStreamSynchronize(ctx->stream) unless a same-device group exists (then the frees are parked anyway and the stream may hold a spinning learner)
void crux_free_device(crux_ctx* ctx, void* p);      // stream-synchronised free (parked like every free while a same-device group exists)
void crux_prof_begin(crux_ctx* ctx, int slot);
void crux_prof_end(crux_ctx* ctx, int slot);

#define HIPCHK(ctx, expr)                                                                       \
  do {                                                                                          \
    hipError_t e__ = (expr);                                                                    \
    if (e__ != hipSuccess) (void)hipGetLastError();   /* reported here: must not resurface as the "launch error" of the next kernel (crux_launch_check) */ \
    if (e__ != hipSuccess) return crux_fail((ctx), CRUX_EHIP, "%s failed: %s (%s:%d)", #expr,    \
                                            hipGetErrorString(e__), __FILE__, __LINE__);        \
  } while (0)

// Device-side description of a Chain(Dense...) with the Flux.params flat layout.
struct NetDesc {
  int32_t L;
  int32_t dims[CRUX_MAXL + 1];
  int32_t acts[CRUX_MAXL];
  int32_t woff[CRUX_MAXL];
  int32_t boff[CRUX_MAXL];
  int32_t xoff;        // offset of the trailing extras (logSigma)
  int32_t n_extra;
  int32_t n_params;
  int32_t maxdim;
};

struct crux_mlp {
  crux_ctx* ctx = nullptr;
  NetDesc nd{};
  float* p = nullptr;   // flat params (device)
  float* g = nullptr;   // flat grads  (device)
  float* m = nullptr;   // Adam first moment
  float* v = nullptr;   // Adam second moment
  double* bp = nullptr; // device: [beta1^t, beta2^t]
  double eta = 0, b1 = 0, b2 = 0, eps = 0;
  bool has_adam = false;
  float squash = 0.f;   // > 0: SquashedGaussianPolicy with this ascale (policies.jl:353-400) wherever the Gaussian head is used
  float* ws = nullptr;  // dense-engine workspace (dense.hip): cached activations 1..L + two delta buffers, capacity ws_B samples
  int64_t ws_B = 0;
};

struct crux_buffer {
  crux_ctx* ctx = nullptr;
  int32_t obs_dim = 0, act_dim = 0, act_kind = 0;
  int64_t capacity = 0, elements = 0, next_ind = 0, total_count = 0;
  uint32_t mask = 0;
  void* col[CRUX_NCOLS] = {nullptr};
  bool prioritized = false;
  float alpha = 0.6f;
  float* priorities = nullptr;   // device [capacity]
  float* cumsum = nullptr;       // device [capacity]
  bool cumsum_valid = false;
  float* pminmax = nullptr;      // device [2]: max_priority, min_priority (un-powered, Float32 fields)
  std::vector<int64_t> indices;  // host copy of the last sample's ids (target.indices), fetched from d_indices on demand
  int64_t indices_n = 0; bool indices_stale = false;
  int64_t* d_indices = nullptr;  // device copy [capacity]
  uint64_t sample_seed = 0x5EED5A3Full; uint32_t sample_stream = 0;   // Philox key / stream of the draws that sample FROM this buffer
  // pairwise-cumsum tree of Base.cumsum for the current length (built on the host once per length, see per.hip)
  int64_t topo_n = -1; int32_t topo_leaves = 0, topo_nodes = 0, topo_levels = 0;
  int32_t* topo_leaf_start = nullptr; int32_t* topo_leaf_len = nullptr; int32_t* topo_leaf_node = nullptr;
  int32_t* topo_left = nullptr; int32_t* topo_right = nullptr; int32_t* topo_level_off = nullptr;   // heap-numbered nodes (root 1, children 2k / 2k + 1); level l = [2^l, 2^(l+1))
  float* topo_total = nullptr; float* topo_prefix = nullptr;
  // incremental maintenance (per.hip): `cumsum` holds the leaf-LOCAL running sums; c[i] = prefix[leaf(i)] + cumsum[i]. After update_priorities! only the
  // touched leaves are re-summed (k_leaf_refresh); the node totals / prefixes are re-derived by the LDS tree pass at the next sample.
  int64_t per_run_n = -1; bool per_full_dirty = true;
  int32_t* order_a = nullptr;    // device [capacity] logical->physical order scratch for batch_train
  int32_t* order_b = nullptr;
  float* aux_ones = nullptr; float* aux_zeros = nullptr;   // [capacity] constant columns (logpdf_bc_loss = a2c_loss with advantage 1, old logprob 0)
  int32_t* ord_all[2] = {nullptr, nullptr}; size_t ord_all_cap[2] = {0, 0};   // per-epoch composed orders (ints) for the actor / critic learner
  float* pack = nullptr; size_t pack_floats = 0;   // packed learner rows (train.hip: ensure_pack), capacity x stride floats
  int32_t* order_c = nullptr;    // second pair for the concurrent critic learner
  int32_t* order_d = nullptr;
};

static inline int col_elem(const crux_buffer* b, int k) {
  switch (k) {
    case CRUX_COL_A: return b->act_kind == CRUX_ACTION_DISCRETE ? 1 : 4;
    case CRUX_COL_DONE: case CRUX_COL_EPISODE_END: return 1;
    case CRUX_COL_T: case CRUX_COL_I: return 8;
    default: return 4;
  }
}
static inline int col_rows(const crux_buffer* b, int k) {
  return (k == CRUX_COL_S || k == CRUX_COL_SP) ? b->obs_dim : (k == CRUX_COL_A ? b->act_dim : 1);
}
static inline size_t col_stride(const crux_buffer* b, int k) { return (size_t)col_elem(b, k) * (size_t)col_rows(b, k); }
static inline bool has_col(const crux_buffer* b, int k) { return k >= 0 && k < CRUX_NCOLS && (b->mask & (1u << k)); }

struct crux_env {
  crux_ctx* ctx = nullptr;
  int32_t kind = 0, n_envs = 0, max_steps = 0, obs_dim = 0, act_dim = 0, state_dim = 0;
  float gamma = 0.99f;
  uint64_t seed = 0;
  float* mu = nullptr;       // device [obs_dim]
  float* sigma = nullptr;    // device [obs_dim]
  double* state = nullptr;   // device [state_dim x n_envs]
  int64_t* ep_len = nullptr; // device [n_envs]
  int64_t* n_resets = nullptr;
  int64_t* steps_taken = nullptr;
  float* svec = nullptr;     // device [obs_dim x n_envs] current (whitened) observation
  double* acc = nullptr;     // device [2*n_envs]: per-env sum_r, n_episode_end of the last rollout
};

// device helpers shared by kernels -------------------------------------------------------------
__device__ __forceinline__ float crux_act(int a, float z) {
  return a == CRUX_ACT_RELU ? (z > 0.f ? z : 0.f) : (a == CRUX_ACT_TANH ? tanhf(z) : z);
}
__device__ __forceinline__ float crux_act_grad(int a, float y, float d) {   // y = post-activation
  return a == CRUX_ACT_RELU ? (y > 0.f ? d : 0.f) : (a == CRUX_ACT_TANH ? d * (1.f - y * y) : d);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

int32_t crux_launch_check(crux_ctx* ctx, const char* what);
// SquashedGaussianPolicy arithmetic shared by the rollout and learner kernels (policies.jl:374-396): sigma = exp(clamp(logSigma, -5, 2));
// correction term of the un-tanh'd action u: 2(log 2 - u - softplus(-2u)), softplus(x) = log1p(exp(-|x|)) + relu(x) (NNlib);
// logpdf(pi, s, a) un-tanh's the stored action as atanh(clamp(a / ascale, -1 + 1f-5, 1 - 1f-5)).
__device__ __forceinline__ float sq_softplus(float x) { return log1pf(expf(-fabsf(x))) + (x > 0.f ? x : 0.f); }
__device__ __forceinline__ float sq_corr(float u) { return 2.f * (logf(2.0f) - u - sq_softplus(-2.f * u)); }
__device__ __forceinline__ float sq_clampls(float ls) { return ls < -5.f ? -5.f : ls > 2.f ? 2.f : ls; }
__device__ __forceinline__ float sq_untanh(float a, float ascale) { float t = a / ascale; const float lo = -1.0f + 1.0e-5f, hi = 1.0f - 1.0e-5f; t = t < lo ? lo : t > hi ? hi : t; return atanhf(t); }
struct crux_fwd_job { const float* p; const float* x; float* y; };
int32_t crux_mlp_forward_multi_impl(crux_ctx* c, const NetDesc& nd, const crux_fwd_job* d_jobs, int n_jobs, int64_t B);   // mlp.hip
int32_t crux_comm_allreduce_mean_impl(crux_ctx* c, crux_mlp* const* nets, int n_nets);   // comm.hip
// dense.hip: differentiable Chain(Dense...) on the tile-GEMM engine
int32_t crux_dense_forward(crux_mlp* n, const float* d_x, int64_t B, hipStream_t st);
// Layer 0's weight / bias gradient left as four quarter partials by Dgrad2W1Op (dense_fused.h) for up to two networks: Sumsq2Op, the reader that follows every pullback
// of a train! step, forms and stores the final values while it sums the squares. part == nullptr: nothing deferred for that network.
struct Sumsq2Fix { const float* part[2]; int32_t out1[2], in0[2], woff[2], boff[2]; float scale[2]; };
// defer (slot of a Sumsq2Fix the caller passes to the Sumsq2Op that follows, or NULL): permits the fused pullback of layers 1 / 0, whose layer-0 gradient is completed by that op
// nanflags (or NULL): [0] |= 1 when a gradient value stored by the pullback is NaN, [1] |= 1 when a deferred layer-0 partial is not finite (its final sum may be NaN): what
// AdamSelfOp (sac.hip) gates on, so that Adam shares the phase of the norm instead of following it
int32_t crux_dense_backward(crux_mlp* n, const float* d_x, int64_t B, const float* d_dy, float gscale, bool want_g, float* d_dx, hipStream_t st, Sumsq2Fix* defer = nullptr, int defer_slot = 0, int32_t* nanflags = nullptr);
float* crux_dense_act(crux_mlp* n, int l);
int32_t crux_td_step_dense(crux_mlp* net, crux_buffer* b, const float* d_y, int32_t use_weight, float* info_out, float* d_err);   // sac.hip
#define CRUX_DENSE_MIN_WIDTH 128   // networks at least this wide go through the multi-CU dense engine
