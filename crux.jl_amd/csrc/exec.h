// exec.h -- the fused-step executor: a recorded sequence of kernel bodies run by ONE persistent launch.
//
// The off-policy learners on wide networks (value_training, src/model_free/off_policy.jl:66-111: rand! -> target -> td_error / update_priorities! ->
// train!(critic) -> train!(actor) -> target update) are chains of 20-70 small dependent kernels; launched one by one each costs ~5 us whatever its
// size, so an epoch was launch-bound. In record mode the launch sites push an ExecOp {kernel body id, block count, packed arguments}
// instead of launching; crux_exec_run uploads the list and starts k_exec: G workgroups on ONE XCD walk the list, each taking the blocks
// b = wg, wg + G, ... of every op, with a counter barrier through the shared L2 between dependent ops (1.3 us instead of a launch).
// The bodies are the SAME device functions the stand-alone kernels call (struct XOp::run), so both forms compute identical results.
#pragma once
#include "common.h"
#include <type_traits>
#include <vector>

enum {
  OP_GEMM = 1, OP_ACT_GRAD, OP_GAUSS_EXPLORE, OP_CONCAT_SA, OP_SAC_TARGET, OP_DPG_ACTION, OP_DPG_TARGET, OP_FILL, OP_SLICE_ROWS, OP_MEAN_INFO, OP_TEMP_HEAD,
  OP_Q_HEAD, OP_TD_HEAD, OP_TD_INFO, OP_SUMSQ2, OP_CRITIC_INFO, OP_ACTOR_HEAD, OP_ACTOR_GRAD, OP_ROWSUM, OP_ACTOR_INFO, OP_ADAM_GATED,
  OP_PER_SEARCH, OP_UNIFORM_IDS, OP_GATHER_RING_ALL, OP_RING_IDS, OP_LEAF_REFRESH, OP_TREE_TOUCH, OP_PER_UPDATE, OP_DQN_TARGET, OP_TD_ERROR, OP_POLYAK, OP_COPY_F32, OP_ADAM_ADVANCE, OP_SOFTQ_TARGET,
  OP_FWD12, OP_WGRAD2, OP_DGRAD2W1,     // dense_fused.h (round 4)
  OP_PER_SAMPLE,                       // per.hip: search + gather of one row per wave (round 4)
  OP_ACTOR_EXPLORE_TILE, OP_SAC_CRITIC_TILE, OP_CRITIC_INFO2, OP_SAC_ACTOR_TILE, OP_ACTOR_INFO2, OP_CRITIC_DX_TILE, OP_DQN_TD_TILE, OP_TD_INFO2,     // sac_fused.h (round 4)
  OP_LEAF_TOUCH,                       // per.hip: leaf re-sum + root paths in one launch (round 4)
  OP_ADAM_SELF, OP_ADAM_ADVANCE_SELF   // sac.hip: Adam gated on the producers' NaN flags, in the phase of the norm (round 4)
};

#define CRUX_EXEC_ARG_BYTES 768
struct ExecOp { int32_t kid; uint32_t nblocks; int32_t barrier; int32_t abytes; alignas(8) unsigned char args[CRUX_EXEC_ARG_BYTES]; };   // abytes: size of the packed arguments actually used

// ---- argument packs: the parameters of XOp::run after (bid, nblocks), stored by value in declaration order --------------------------------
template <class... T> struct ArgPack;
template <> struct ArgPack<> { __host__ __device__ ArgPack() {} };
template <class H, class... T> struct ArgPack<H, T...> {
  H head; ArgPack<T...> tail;
  __host__ __device__ ArgPack() {}
  __host__ __device__ ArgPack(H h, T... t) : head(h), tail(t...) {}
};
template <class F> struct OpSig;
template <class... A> struct OpSig<void (*)(const unsigned, const unsigned, A...)> { using pack = ArgPack<std::remove_cv_t<A>...>; };
template <class Op> using OpPack = typename OpSig<decltype(&Op::run)>::pack;

template <class Op, class... Done> __device__ __forceinline__ void exec_apply(unsigned bid, unsigned nb, const ArgPack<>&, Done... d) { Op::run(bid, nb, d...); }
template <class Op, class H, class... T, class... Done> __device__ __forceinline__ void exec_apply(unsigned bid, unsigned nb, const ArgPack<H, T...>& p, Done... d) {
  exec_apply<Op>(bid, nb, p.tail, d..., p.head);
}

// ---- host: recording --------------------------------------------------------------------------------------------------------------------
struct ExecReadback { float* host_info; const float* d_info; const int32_t* d_status; const char* who; };
struct ExecRec {
  std::vector<ExecOp> ops;
  bool active = false;
  char* small = nullptr; size_t small_cap = 0, small_off = 0;     // device: per-piece info rows / statistics / status words that outlive the piece until the read-back
  std::vector<ExecReadback> readbacks;
  void* d_ops = nullptr; size_t d_ops_cap = 0;                     // device copy of the op list
  unsigned* d_ctr = nullptr;                                       // device: barrier counter, abort flag
  void* h_stage = nullptr; size_t h_stage_cap = 0;                 // pinned staging of the op list and of the read-backs
  size_t scratch_floor = 0, scratch_off = 0;                       // scratch requests made while recording are carved one after the other from the pre-sized block
  // chained epochs (crux_dqn_epochs / crux_sac_epochs): several value_training epochs recorded into ONE list, scheduled and run once -- no host round trip between
  // the epochs of an iteration. While `chain` is set the per-epoch entry points append their phase tags (offset by chain_base) instead of scheduling and running.
  bool chain = false, chain_ok = true; int chain_base = 0; std::vector<int> chain_tags;
  // the two-kernel persistent form of the DQN-family epochs (dqn_persist.h): first op of every recorded epoch; what crux_exec_run launches instead of the phases
  std::vector<size_t> epoch_marks;
  struct Dqp { bool on = false; int in = 0, out = 0, bt = 0, n_epochs = 0; void* net = nullptr; void* tnet = nullptr; void* batch = nullptr; float gamma = 0.f; bool use_weight = false; float* d_err = nullptr;
               std::vector<int32_t> tab; } dqp;
  void* dqp_buf = nullptr;                                          // device: learner barrier words, flags, tables, exchange areas
  // asynchronous runs (crux_dqn_epochs_async): no read-back, no host synchronisation -- the info rows are copied to a caller-owned device array by ops of the list itself.
  // The op list is uploaded from a ring of pinned staging buffers, so the host may record and enqueue up to three chains ahead of the one the device is running.
  bool async = false;
  float* info_row_override = nullptr;      // asynchronous tile-plan epochs: the epoch's info op writes the caller's device row itself (no copy op, no extra phase behind the chain's last epoch)
  void* h_ring[4] = {nullptr, nullptr, nullptr, nullptr}; size_t h_ring_cap[4] = {0, 0, 0, 0}; void* h_ring_ev[4] = {nullptr, nullptr, nullptr, nullptr}; unsigned h_ring_next = 0;
};
bool crux_exec_recording(const crux_ctx* c);
int32_t crux_exec_begin(crux_ctx* c);                 // start recording on this context (the launch sites below push ops instead of launching)
int32_t crux_exec_run(crux_ctx* c);                   // upload, run the persistent kernel, fulfil the recorded read-backs; ends the recording
void crux_exec_abort(crux_ctx* c);                    // drop a recording after an error
ExecOp* crux_exec_new_op(crux_ctx* c, int kid, unsigned nblocks);
void* crux_exec_small(crux_ctx* c, size_t bytes);     // 256-byte aligned block of the small region (valid until the next crux_exec_begin)
void crux_exec_add_readback(crux_ctx* c, float* host_info, const float* d_info, const int32_t* d_status, const char* who);

template <class Op, int KID, class... A> inline void crux_exec_push(crux_ctx* c, unsigned nblocks, A... a) {
  using P = OpPack<Op>;
  static_assert(sizeof(P) <= CRUX_EXEC_ARG_BYTES, "op arguments exceed the ExecOp slot");
  static_assert(std::is_trivially_copyable<P>::value, "op arguments must be plain data");
  ExecOp* op = crux_exec_new_op(c, KID, nblocks);
  P p(a...);
  memcpy(op->args, &p, sizeof p); op->abytes = (int32_t)sizeof p;
}
// a launch site: record when the context is recording, launch otherwise. NT = threads per block of the stand-alone launch (the executor always runs 256).
#define CRUX_RUN(c, OpT, KID, kernel, nblocks, NT, stream, ...)                                                            \
  do { if (crux_exec_recording(c)) crux_exec_push<OpT, KID>((c), (unsigned)(nblocks), __VA_ARGS__);                          \
       else hipLaunchKernelGGL(kernel, dim3((unsigned)(nblocks)), dim3(NT), 0, (stream), __VA_ARGS__); } while (0)
// hipMemsetAsync(p, 0, bytes) of Float32 data inside a recordable chain
int32_t crux_exec_zero(crux_ctx* c, void* d_ptr, size_t bytes, hipStream_t st);
