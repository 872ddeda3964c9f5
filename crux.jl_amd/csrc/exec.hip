// exec.hip -- the fused-step executor (see exec.h) and the fused value_training epochs built on it. Part of the off-policy unit
// (offpolicy_unit.hip includes dense.hip, sac.hip, per.hip and this file, so that one translation unit sees every op body).
#include "exec.h"
#include "ops_small.h"

#define EXEC_G 64                 // workgroups of the persistent launch, all on one XCD (32 CUs x 2)
#define EXEC_SMALL_BYTES (256 * 1024)

int32_t crux_x2_placement_ok_c(crux_ctx* c);

// ---- device: the interpreter ------------------------------------------------------------------------------------------------------------
template <class Op> __device__ __forceinline__ void exec_dispatch(const ExecOp* op, unsigned bid) {
  const OpPack<Op> p = *(const OpPack<Op>*)op->args;        // uniform address: scalar loads
  exec_apply<Op>(bid, op->nblocks, p);
}
// Counter barrier between workgroups that sit behind ONE L2 (the learner kernels' exchange, tools/xcu_barrier_bench.hip): stores are write-through
// to the L2, so s_waitcnt + one relaxed agent-scope atomic is the release; the acquire side drops this CU's L1 and scalar cache.
__device__ __forceinline__ bool exec_barrier(unsigned* ctr, unsigned target) {
  __shared__ int ok_s;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0; int ok = 1;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) { __builtin_amdgcn_s_sleep(1);
      if ((++spins & 255u) == 0u && (spins > (1u << 23) || __hip_atomic_load(ctr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) { ok = 0; break; } }   // never hang the GPU
    if (!ok) __hip_atomic_store(ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ok_s = ok;
  }
  __syncthreads();
  asm volatile("buffer_inv sc1\n\ts_dcache_inv\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  return ok_s != 0;
}
__global__ __launch_bounds__(256) void k_exec(const ExecOp* __restrict__ ops, int nops, unsigned* ctr, int xcd, int32_t* status) {
  if (xcd >= 0 && (int)(blockIdx.x & 7) != xcd) return;
  const unsigned wg = xcd >= 0 ? blockIdx.x >> 3 : blockIdx.x, G = xcd >= 0 ? gridDim.x >> 3 : gridDim.x;
  unsigned phase = 0;
  for (int o = 0; o < nops; ++o) {
    const ExecOp* op = ops + o;
    const int kid = op->kid; const unsigned nb = op->nblocks;
    for (unsigned b = wg; b < nb; b += G) {
      switch (kid) {
        case OP_GEMM: exec_dispatch<GemmOp>(op, b); break;
        case OP_ACT_GRAD: exec_dispatch<ActGradOp>(op, b); break;
        case OP_GAUSS_EXPLORE: exec_dispatch<GaussExploreOp>(op, b); break;
        case OP_CONCAT_SA: exec_dispatch<ConcatSaOp>(op, b); break;
        case OP_SAC_TARGET: exec_dispatch<SacTargetOp>(op, b); break;
        case OP_DPG_ACTION: exec_dispatch<DpgActionOp>(op, b); break;
        case OP_DPG_TARGET: exec_dispatch<DpgTargetOp>(op, b); break;
        case OP_FILL: exec_dispatch<FillOp>(op, b); break;
        case OP_SLICE_ROWS: exec_dispatch<SliceRowsOp>(op, b); break;
        case OP_MEAN_INFO: exec_dispatch<MeanInfoOp>(op, b); break;
        case OP_TEMP_HEAD: exec_dispatch<TempHeadOp>(op, b); break;
        case OP_Q_HEAD: exec_dispatch<QHeadOp>(op, b); break;
        case OP_TD_HEAD: exec_dispatch<TdHeadOp>(op, b); break;
        case OP_TD_INFO: exec_dispatch<TdInfoOp>(op, b); break;
        case OP_SUMSQ2: exec_dispatch<Sumsq2Op>(op, b); break;
        case OP_CRITIC_INFO: exec_dispatch<CriticInfoOp>(op, b); break;
        case OP_ACTOR_HEAD: exec_dispatch<ActorHeadOp>(op, b); break;
        case OP_ACTOR_GRAD: exec_dispatch<ActorGradOp>(op, b); break;
        case OP_ROWSUM: exec_dispatch<RowsumOp>(op, b); break;
        case OP_ACTOR_INFO: exec_dispatch<ActorInfoOp>(op, b); break;
        case OP_ADAM_GATED: exec_dispatch<AdamGatedOp>(op, b); break;
        case OP_PER_SEARCH: exec_dispatch<PerSearchOp>(op, b); break;
        case OP_UNIFORM_IDS: exec_dispatch<UniformIdsOp>(op, b); break;
        case OP_GATHER_RING_ALL: exec_dispatch<GatherRingAllOp>(op, b); break;
        case OP_RING_IDS: exec_dispatch<RingIdsOp>(op, b); break;
        case OP_LEAF_REFRESH: exec_dispatch<LeafRefreshOp>(op, b); break;
        case OP_TREE_TOUCH: exec_dispatch<TreeTouchOp>(op, b); break;
        case OP_PER_UPDATE: exec_dispatch<PerUpdateOp>(op, b); break;
        case OP_DQN_TARGET: exec_dispatch<DqnTargetOp>(op, b); break;
        case OP_TD_ERROR: exec_dispatch<TdErrorOp>(op, b); break;
        case OP_POLYAK: exec_dispatch<PolyakOp>(op, b); break;
        case OP_COPY_F32: exec_dispatch<CopyF32Op>(op, b); break;
        default: break;
      }
      __syncthreads();                                   // the bodies' static LDS is reused by the next block / op of this workgroup
    }
    if (op->barrier) { phase += 1; if (!exec_barrier(ctr, phase * G)) { if (threadIdx.x == 0 && wg == 0) status[0] = CRUX_EHIP; return; } }
  }
}

// ---- host: recording -------------------------------------------------------------------------------------------------------------------
static ExecRec* rec_of(crux_ctx* c) { return (ExecRec*)c->rec; }
bool crux_exec_recording(const crux_ctx* c) { return c && c->rec && ((const ExecRec*)c->rec)->active; }
int32_t crux_exec_begin(crux_ctx* c) {
  if (!c->rec) c->rec = new ExecRec();
  ExecRec* r = rec_of(c);
  if (r->active) return crux_fail(c, CRUX_EINVAL, "executor: a recording is already open on this context");
  if (!r->small) { if (hipMalloc(&r->small, EXEC_SMALL_BYTES) != hipSuccess) return crux_fail(c, CRUX_ENOMEM, "executor: small region"); r->small_cap = EXEC_SMALL_BYTES; }
  if (!r->d_ctr) { if (hipMalloc(&r->d_ctr, 256) != hipSuccess) return crux_fail(c, CRUX_ENOMEM, "executor: barrier counter"); }
  if (!crux_scratch(c, (size_t)32 << 20)) return crux_fail(c, CRUX_ENOMEM, "executor: scratch");     // pre-sized: the scratch block must not move while pointers into it are recorded
  r->scratch_floor = c->scratch_bytes;
  r->ops.clear(); r->readbacks.clear(); r->small_off = 0; r->active = true;
  return CRUX_OK;
}
void crux_exec_abort(crux_ctx* c) { if (c->rec) { ExecRec* r = rec_of(c); r->active = false; r->ops.clear(); r->readbacks.clear(); } }
ExecOp* crux_exec_new_op(crux_ctx* c, int kid, unsigned nblocks) {
  ExecRec* r = rec_of(c); r->ops.emplace_back(); ExecOp* op = &r->ops.back();
  op->kid = kid; op->nblocks = nblocks; op->barrier = 1; op->pad = 0; return op;
}
void* crux_exec_small(crux_ctx* c, size_t bytes) {
  ExecRec* r = rec_of(c); bytes = (bytes + 255) / 256 * 256;
  if (!r || r->small_off + bytes > r->small_cap) return nullptr;
  void* p = r->small + r->small_off; r->small_off += bytes; return p;
}
void crux_exec_add_readback(crux_ctx* c, float* host_info, const float* d_info, const int32_t* d_status, const char* who) { rec_of(c)->readbacks.push_back({host_info, d_info, d_status, who}); }
int32_t crux_exec_zero(crux_ctx* c, void* d_ptr, size_t bytes, hipStream_t st) {
  if (!crux_exec_recording(c)) { HIPCHK(c, hipMemsetAsync(d_ptr, 0, bytes, st)); return CRUX_OK; }
  if (bytes % 4) return crux_fail(c, CRUX_EINVAL, "executor: zero-fill of %zu bytes", bytes);
  const int64_t n = (int64_t)(bytes / 4);
  crux_exec_push<FillOp, OP_FILL>(c, (unsigned)((n + 255) / 256), (float*)d_ptr, 0.f, n);
  return CRUX_OK;
}
// ops i and i+1 .. may run in the same phase (no barrier between them) when the caller knows they are independent
static void exec_no_barrier_before_last(crux_ctx* c, int count) { ExecRec* r = rec_of(c); const size_t n = r->ops.size(); for (int k = 0; k < count && (size_t)k + 2 <= n; ++k) r->ops[n - 2 - k].barrier = 0; }

int32_t crux_exec_run(crux_ctx* c) {
  ExecRec* r = rec_of(c);
  if (!r || !r->active) return crux_fail(c, CRUX_EINVAL, "executor: no open recording");
  r->active = false;
  if (c->scratch_bytes != r->scratch_floor) return crux_fail(c, CRUX_EHIP, "executor: the scratch block moved while it was being recorded");
  const size_t nops = r->ops.size();
  int32_t rc = CRUX_OK;
  if (nops) {
    const size_t ob = nops * sizeof(ExecOp), rb = r->readbacks.size() * (sizeof(float) * CRUX_INFO_N + 16), need_h = ob + rb + 64;
    if (r->d_ops_cap < ob) { if (r->d_ops) { HIPCHK(c, hipStreamSynchronize(c->stream)); (void)hipFree(r->d_ops); } r->d_ops_cap = ob * 2 + 4096; if (hipMalloc(&r->d_ops, r->d_ops_cap) != hipSuccess) { r->d_ops = nullptr; r->d_ops_cap = 0; return crux_fail(c, CRUX_ENOMEM, "executor: op list"); } }
    if (r->h_stage_cap < need_h) { if (r->h_stage) { HIPCHK(c, hipStreamSynchronize(c->stream)); (void)hipHostFree(r->h_stage); } r->h_stage_cap = need_h * 2 + 4096; if (hipHostMalloc(&r->h_stage, r->h_stage_cap, hipHostMallocDefault) != hipSuccess) { r->h_stage = nullptr; r->h_stage_cap = 0; return crux_fail(c, CRUX_ENOMEM, "executor: staging"); } }
    r->ops.back().barrier = 0;
    memcpy(r->h_stage, r->ops.data(), ob);
    HIPCHK(c, hipMemcpyAsync(r->d_ops, r->h_stage, ob, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(r->d_ctr, 0, 256, c->stream));
    static const int one_xcd = getenv("CRUX_EXEC_CHIP") ? 0 : 1;
    // the counter barrier relies on one shared L2: all workgroups on XCD 0 (workgroup i of a grid lands on XCD i mod 8, verified by the placement probe)
    const int xcd = (one_xcd && crux_x2_placement_ok_c(c)) ? 0 : -2;
    if (xcd == -2) return crux_fail(c, CRUX_EUNSUP, "executor: workgroups are not placed round-robin over the XCDs on this device");
    static const int g_env = getenv("CRUX_EXEC_G") ? atoi(getenv("CRUX_EXEC_G")) : 0;
    const int G = (g_env >= 1 && g_env <= 128) ? g_env : EXEC_G;
    hipLaunchKernelGGL(k_exec, dim3(G * 8), dim3(256), 0, c->stream, (const ExecOp*)r->d_ops, (int)nops, r->d_ctr, xcd, (int32_t*)(r->d_ctr + 8));
    rc = crux_launch_check(c, "k_exec"); if (rc) return rc;
    char* hb = (char*)r->h_stage + ob;
    for (size_t k = 0; k < r->readbacks.size(); ++k) { char* h = hb + k * (sizeof(float) * CRUX_INFO_N + 16);
      HIPCHK(c, hipMemcpyAsync(h, r->readbacks[k].d_info, sizeof(float) * CRUX_INFO_N, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipMemcpyAsync(h + sizeof(float) * CRUX_INFO_N, r->readbacks[k].d_status, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream)); }
    int32_t* hst = (int32_t*)(hb + rb);
    HIPCHK(c, hipMemcpyAsync(hst, r->d_ctr + 8, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (*hst) return crux_fail(c, *hst, "executor: the fused launch reported status %d (a workgroup did not reach a barrier)", *hst);
    for (size_t k = 0; k < r->readbacks.size(); ++k) { const char* h = hb + k * (sizeof(float) * CRUX_INFO_N + 16);
      if (r->readbacks[k].host_info) memcpy(r->readbacks[k].host_info, h, sizeof(float) * CRUX_INFO_N);
      int32_t st; memcpy(&st, h + sizeof(float) * CRUX_INFO_N, sizeof st);
      if (st == CRUX_ENAN && !rc) rc = crux_fail(c, CRUX_ENAN, "NaN detected! (grad norm is NaN, src/training.jl:20) in %s", r->readbacks[k].who); }
  }
  r->ops.clear(); r->readbacks.clear();
  return rc;
}
extern "C" int crux_x2_placement_ok(crux_ctx* c);
int32_t crux_x2_placement_ok_c(crux_ctx* c) { return crux_x2_placement_ok(c); }

// ---- fused value_training epochs ------------------------------------------------------------------------------------------------------------
int32_t crux_per_prepare(crux_buffer* source);      // per.hip: any full rebuild of the cumsum tree happens before the recording starts
extern "C" {
int32_t crux_per_sample(crux_buffer* target, crux_buffer* source, int64_t B, const double* rands, float beta, uint64_t i);
int32_t crux_uniform_sample(crux_buffer* target, crux_buffer* source, int64_t B, const int64_t* ids, uint64_t i);
int32_t crux_dqn_target(crux_mlp* tn, crux_buffer* batch, float gamma, float* d_y);
int32_t crux_td_step(crux_mlp* net, crux_buffer* batch, const float* d_y, int32_t use_weight, float* info_out);
int32_t crux_td_step_with_error(crux_mlp* net, crux_buffer* batch, const float* d_y, int32_t use_weight, float* d_err, float* info_out);
int32_t crux_per_update_device(crux_buffer* b, const int64_t* d_ids, const float* d_v, int64_t n);
int32_t crux_polyak(crux_mlp* to, const crux_mlp* from, float tau);

// One epoch of value_training for the DQN family (src/model_free/off_policy.jl:69-93 with dqn_target, rl/dqn.jl:4-6): rand!(batch, source; i) ->
// y = target(pi_minus, batch) -> [td_error -> update_priorities!(source, batch.indices, .)] -> train!(pi, td_loss). Networks at least
// CRUX_DENSE_MIN_WIDTH wide run the whole epoch as ONE fused launch; narrower ones take the same steps one call at a time.
int32_t crux_dqn_epoch(crux_mlp* net, crux_mlp* target_net, crux_buffer* source, crux_buffer* batch, float gamma, int32_t use_weight, float beta,
                       uint64_t sample_counter, float* info_out) {
  if (!net || !target_net || !source || !batch) return CRUX_EINVAL;
  crux_ctx* c = net->ctx; const int64_t B = batch->capacity;
  const bool per = source->prioritized;
  const bool fuse = net->nd.maxdim >= CRUX_DENSE_MIN_WIDTH && target_net->nd.maxdim >= CRUX_DENSE_MIN_WIDTH && !getenv("CRUX_NO_FUSED_EPOCH");
  int32_t rc;
  if (per) { rc = crux_per_prepare(source); if (rc) return rc; }
  float* d_y = nullptr; float* d_err = nullptr;
  { char* sc2 = (char*)c->epoch_tmp;       // targets and td errors: a block of the context that no piece carves
    if (c->epoch_tmp_bytes < 8 * (size_t)B + 512) { if (sc2) { HIPCHK(c, hipStreamSynchronize(c->stream)); (void)hipFree(sc2); }
      c->epoch_tmp_bytes = 16 * (size_t)B + 4096; if (hipMalloc(&c->epoch_tmp, c->epoch_tmp_bytes) != hipSuccess) { c->epoch_tmp = nullptr; c->epoch_tmp_bytes = 0; return crux_fail(c, CRUX_ENOMEM, "dqn_epoch: targets"); } sc2 = (char*)c->epoch_tmp; }
    d_y = (float*)sc2; d_err = (float*)(sc2 + ((4 * (size_t)B + 255) / 256) * 256); }
  static const int eager_mask = getenv("CRUX_EXEC_EAGER_MASK") ? atoi(getenv("CRUX_EXEC_EAGER_MASK")) : 0;      // debugging: run piece k outside the fused launch
  auto piece = [&](int bit) -> int32_t { if (!fuse) return CRUX_OK;
    if (eager_mask & bit) { if (crux_exec_recording(c)) return crux_exec_run(c); return CRUX_OK; }
    if (!crux_exec_recording(c)) return crux_exec_begin(c); return CRUX_OK; };
  auto bail = [&](int32_t e) { if (fuse) crux_exec_abort(c); return e; };
  rc = piece(1); if (rc) return bail(rc);
  rc = per ? crux_per_sample(batch, source, B, nullptr, beta, sample_counter) : crux_uniform_sample(batch, source, B, nullptr, sample_counter); if (rc) return bail(rc);
  rc = piece(2); if (rc) return bail(rc);
  rc = crux_dqn_target(target_net, batch, gamma, d_y); if (rc) return bail(rc);
  rc = piece(4); if (rc) return bail(rc);
  if (per) { rc = crux_td_step_with_error(net, batch, d_y, use_weight, d_err, info_out); if (rc) return bail(rc);
    rc = piece(8); if (rc) return bail(rc);
    rc = crux_per_update_device(source, batch->d_indices, d_err, B); if (rc) return bail(rc); }
  else { rc = crux_td_step(net, batch, d_y, use_weight, info_out); if (rc) return bail(rc); }
  return (fuse && crux_exec_recording(c)) ? crux_exec_run(c) : CRUX_OK;
}

int32_t crux_sac_target(crux_mlp* actor, crux_mlp* q1t, crux_mlp* q2t, crux_mlp* la, crux_buffer* b, float gamma, uint64_t seed, uint64_t counter, float* d_y);
int32_t crux_sac_temp_step(crux_mlp* actor, crux_mlp* la, crux_buffer* b, float H_target, uint64_t seed, uint64_t counter, float* info_out);
int32_t crux_double_q_step(crux_mlp* q1, crux_mlp* q2, crux_buffer* b, const float* d_y, int32_t use_weight, float* info_out);
int32_t crux_sac_actor_step(crux_mlp* actor, crux_mlp* q1, crux_mlp* q2, crux_mlp* la, crux_buffer* b, uint64_t seed, uint64_t counter, float* info_out);

// One epoch of value_training with SAC's pieces (off_policy.jl:69-104, rl/sac.jl): rand! -> sac_target -> train!(log_alpha, sac_temp_loss) ->
// [train!(critic, double_Q_loss)] -> [train!(actor, sac_actor_loss) -> polyak_average!(pi_minus, pi, tau)] as ONE fused launch.
int32_t crux_sac_epoch(crux_mlp* actor, crux_mlp* q1, crux_mlp* q2, crux_mlp* actor_targ, crux_mlp* q1_targ, crux_mlp* q2_targ, crux_mlp* log_alpha,
                       crux_buffer* source, crux_buffer* batch, float gamma, float H_target, float tau, int32_t use_weight, int32_t update_critic, int32_t update_actor,
                       uint64_t sample_counter, uint64_t noise_seed, uint64_t noise_counter0, float* info_temp, float* info_critic, float* info_actor) {
  if (!actor || !q1 || !q2 || !q1_targ || !q2_targ || !log_alpha || !source || !batch) return CRUX_EINVAL;
  crux_ctx* c = actor->ctx; const int64_t B = batch->capacity;
  if (source->prioritized) return crux_fail(c, CRUX_EUNSUP, "sac_epoch: prioritized replay over a DoubleNetwork critic is not defined (td_error, src/utils.jl:112)");
  const bool fuse = !getenv("CRUX_NO_FUSED_EPOCH");
  int32_t rc; float* d_y = nullptr;
  if (fuse) { rc = crux_exec_begin(c); if (rc) return rc; d_y = (float*)crux_exec_small(c, 4 * (size_t)B);
    if (!d_y) { crux_exec_abort(c); return crux_fail(c, CRUX_EUNSUP, "sac_epoch: batch of %lld rows exceeds the executor's region", (long long)B); } }
  else { if (c->epoch_tmp_bytes < 4 * (size_t)B + 256) { if (c->epoch_tmp) { HIPCHK(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->epoch_tmp); }
      c->epoch_tmp_bytes = 16 * (size_t)B + 4096; if (hipMalloc(&c->epoch_tmp, c->epoch_tmp_bytes) != hipSuccess) { c->epoch_tmp = nullptr; c->epoch_tmp_bytes = 0; return crux_fail(c, CRUX_ENOMEM, "sac_epoch: targets"); } }
    d_y = (float*)c->epoch_tmp; }
  auto bail = [&](int32_t e) { if (fuse) crux_exec_abort(c); return e; };
  rc = crux_uniform_sample(batch, source, B, nullptr, sample_counter); if (rc) return bail(rc);
  rc = crux_sac_target(actor, q1_targ, q2_targ, log_alpha, batch, gamma, noise_seed, noise_counter0, d_y); if (rc) return bail(rc);
  rc = crux_sac_temp_step(actor, log_alpha, batch, H_target, noise_seed, noise_counter0 + 1, info_temp); if (rc) return bail(rc);
  if (update_critic) { rc = crux_double_q_step(q1, q2, batch, d_y, use_weight, info_critic); if (rc) return bail(rc); }
  if (update_actor) {
    rc = crux_sac_actor_step(actor, q1, q2, log_alpha, batch, noise_seed, noise_counter0 + 2, info_actor); if (rc) return bail(rc);
    if (actor_targ) { rc = crux_polyak(actor_targ, actor, tau); if (rc) return bail(rc); }
    rc = crux_polyak(q1_targ, q1, tau); if (rc) return bail(rc);
    rc = crux_polyak(q2_targ, q2, tau); if (rc) return bail(rc);
  }
  return fuse ? crux_exec_run(c) : CRUX_OK;
}
}  // extern "C"

// test hook: value(pi, x) through the executor (the same tile bodies as crux_mlp_forward_cached, run by k_exec) -- tests compare the two bit for bit
extern "C" int32_t crux_debug_exec_forward(crux_mlp* net, const float* d_x, int64_t B, float* d_y, int32_t with_backward, const float* d_dy) {
  if (!net || !d_x || !d_y) return CRUX_EINVAL;
  crux_ctx* c = net->ctx; int32_t rc = crux_exec_begin(c); if (rc) return rc;
  rc = crux_dense_forward(net, d_x, B, c->stream); if (rc) { crux_exec_abort(c); return rc; }
  if (with_backward) { rc = crux_dense_backward(net, d_x, B, d_dy, 1.0f, true, nullptr, c->stream); if (rc) { crux_exec_abort(c); return rc; } }
  rc = crux_exec_run(c); if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(d_y, crux_dense_act(net, net->nd.L), sizeof(float) * (size_t)net->nd.dims[net->nd.L] * (size_t)B, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return CRUX_OK;
}
