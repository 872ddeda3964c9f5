// exec.hip -- the fused-step executor (see exec.h) and the fused value_training epochs built on it. Part of the off-policy unit
// (offpolicy_unit.hip includes dense.hip, sac.hip, per.hip and this file, so that one translation unit sees every op body).
#include "exec.h"
#include "ops_small.h"
#include <algorithm>

#define EXEC_G 64                 // workgroups of the persistent launch, all on one XCD (32 CUs x 2)
#define EXEC_SMALL_BYTES (256 * 1024)

int32_t crux_x2_placement_ok_c(crux_ctx* c);

// ---- device: the interpreter ------------------------------------------------------------------------------------------------------------
template <class Op> __device__ __forceinline__ void exec_dispatch(const ExecOp* op, unsigned bid) {
  // the record sits in LDS (staged one op ahead by k_exec). Its words are moved to SCALAR registers (v_readfirstlane): arguments that arrive in vector
  // registers turn every pointer computation and every uniform branch of the body into per-lane work (the tile GEMM ran 3x slower that way)
  constexpr int NW = (int)((sizeof(OpPack<Op>) + 3) / 4);
  uint32_t w[NW]; const uint32_t* src = (const uint32_t*)op->args;
#pragma unroll
  for (int i = 0; i < NW; ++i) w[i] = __builtin_amdgcn_readfirstlane(src[i]);
  OpPack<Op> p; __builtin_memcpy(&p, w, sizeof p);
  exec_apply<Op>(bid, __builtin_amdgcn_readfirstlane(op->nblocks), p);
}
// Counter barrier between workgroups that sit behind ONE L2 (the learner kernels' exchange, tools/xcu_barrier_bench.hip): stores are write-through
// to the L2, so s_waitcnt + one relaxed agent-scope atomic is the release; the acquire side drops this CU's L1 and scalar cache.
// Flag barrier between workgroups that sit behind ONE L2. Workgroup w publishes the phase number in ITS OWN word (plain store: the vector L1 is
// write-through); one wave then polls all G words with a single coalesced L1-bypassing load per try. No atomics: G arrivals on one address are
// serialised by the L2 (2.4 us per barrier with 64 workgroups, tools/dbg_nops.py). ctr[0..255] = arrival words, ctr[256] = abort flag.
__device__ __forceinline__ bool exec_barrier(unsigned* ctr, unsigned wg, unsigned G, unsigned phase, int flags) {
  __shared__ int ok_s;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // this wave's stores are in the L2
  __syncthreads();
  if (threadIdx.x < 64) {
    if (threadIdx.x == 0) { __hip_atomic_store(ctr + wg, phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    unsigned spins = 0; int ok = 1;
    for (;;) {
      bool here = true;
      for (unsigned q = threadIdx.x; q < G; q += 64) here = here && __hip_atomic_load(ctr + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= phase;
      if (__ballot(!here) == 0ull) break;
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 255u) == 0u && (spins > (1u << 22) || __hip_atomic_load(ctr + 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) { ok = 0; break; }      // never hang the GPU
    }
    if (threadIdx.x == 0) { if (!ok) __hip_atomic_store(ctr + 256, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok_s = ok; }
  }
  __syncthreads();
  // acquire side: drop THIS CU's vector L1 (buffer_inv sc0, workgroup scope in the ISA's terms: the L2 behind it is shared by the whole XCD and needs nothing)
  // and its scalar cache. The agent-scope form (sc1) also walks the L2 and cost 17 us per barrier with 64 workgroups (tools/dbg_nops.py)
  // (round 3: the agent-scope form. `buffer_inv sc0` is a workgroup-scope invalidate and leaves the L1 as it is -- a re-read of a line this CU had cached before another
  //  CU rewrote it returned the OLD data (found with dqn_persist.h's kernels, whose second epoch re-reads the batch rows); flag 1 = the caller reads with L1-bypassing loads)
  if (!(flags & 1)) asm volatile("buffer_inv sc1" ::: "memory");
  if (!(flags & 2)) asm volatile("s_dcache_inv" ::: "memory");
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  return ok_s != 0;
}
#define EXEC_SWITCH(DISPATCH) \
      switch (kid) { \
        case OP_GEMM: DISPATCH<GemmOp>(op, b); break; \
        case OP_ACT_GRAD: DISPATCH<ActGradOp>(op, b); break; \
        case OP_GAUSS_EXPLORE: DISPATCH<GaussExploreOp>(op, b); break; \
        case OP_CONCAT_SA: DISPATCH<ConcatSaOp>(op, b); break; \
        case OP_SAC_TARGET: DISPATCH<SacTargetOp>(op, b); break; \
        case OP_DPG_ACTION: DISPATCH<DpgActionOp>(op, b); break; \
        case OP_DPG_TARGET: DISPATCH<DpgTargetOp>(op, b); break; \
        case OP_FILL: DISPATCH<FillOp>(op, b); break; \
        case OP_SLICE_ROWS: DISPATCH<SliceRowsOp>(op, b); break; \
        case OP_MEAN_INFO: DISPATCH<MeanInfoOp>(op, b); break; \
        case OP_TEMP_HEAD: DISPATCH<TempHeadOp>(op, b); break; \
        case OP_Q_HEAD: DISPATCH<QHeadOp>(op, b); break; \
        case OP_TD_HEAD: DISPATCH<TdHeadOp>(op, b); break; \
        case OP_TD_INFO: DISPATCH<TdInfoOp>(op, b); break; \
        case OP_SUMSQ2: DISPATCH<Sumsq2Op>(op, b); break; \
        case OP_CRITIC_INFO: DISPATCH<CriticInfoOp>(op, b); break; \
        case OP_ACTOR_HEAD: DISPATCH<ActorHeadOp>(op, b); break; \
        case OP_ACTOR_GRAD: DISPATCH<ActorGradOp>(op, b); break; \
        case OP_ROWSUM: DISPATCH<RowsumOp>(op, b); break; \
        case OP_ACTOR_INFO: DISPATCH<ActorInfoOp>(op, b); break; \
        case OP_ADAM_GATED: DISPATCH<AdamGatedOp>(op, b); break; \
        case OP_PER_SEARCH: DISPATCH<PerSearchOp>(op, b); break; \
        case OP_UNIFORM_IDS: DISPATCH<UniformIdsOp>(op, b); break; \
        case OP_GATHER_RING_ALL: DISPATCH<GatherRingAllOp>(op, b); break; \
        case OP_RING_IDS: DISPATCH<RingIdsOp>(op, b); break; \
        case OP_LEAF_REFRESH: DISPATCH<LeafRefreshOp>(op, b); break; \
        case OP_TREE_TOUCH: DISPATCH<TreeTouchOp>(op, b); break; \
        case OP_LEAF_TOUCH: DISPATCH<LeafTouchOp>(op, b); break; \
        case OP_PER_UPDATE: DISPATCH<PerUpdateOp>(op, b); break; \
        case OP_DQN_TARGET: DISPATCH<DqnTargetOp>(op, b); break; \
        case OP_TD_ERROR: DISPATCH<TdErrorOp>(op, b); break; \
        case OP_POLYAK: DISPATCH<PolyakOp>(op, b); break; \
        case OP_COPY_F32: DISPATCH<CopyF32Op>(op, b); break; \
        case OP_ADAM_ADVANCE: DISPATCH<AdamAdvanceOp>(op, b); break; \
        case OP_ADAM_SELF: DISPATCH<AdamSelfOp>(op, b); break; \
        case OP_ADAM_ADVANCE_SELF: DISPATCH<AdamAdvanceSelfOp>(op, b); break; \
        case OP_SOFTQ_TARGET: DISPATCH<SoftqTargetOp>(op, b); break; \
        case OP_PER_SAMPLE: PerSampleGatherOp::run_ptr(b, op->nblocks, (const PerSampleArgs*)op->args); break; \
        case OP_ACTOR_EXPLORE_TILE: DISPATCH<ActorExploreTileOp>(op, b); break; \
        case OP_SAC_CRITIC_TILE: DISPATCH<SacCriticTileOp>(op, b); break; \
        case OP_CRITIC_INFO2: DISPATCH<CriticInfo2Op>(op, b); break; \
        case OP_SAC_ACTOR_TILE: DISPATCH<SacActorTileOp>(op, b); break; \
        case OP_ACTOR_INFO2: DISPATCH<ActorInfo2Op>(op, b); break; \
        case OP_CRITIC_DX_TILE: DISPATCH<CriticDxActorGradTileOp>(op, b); break; \
        case OP_DQN_TD_TILE: DISPATCH<DqnTdTileOp>(op, b); break; \
        case OP_TD_INFO2: DISPATCH<TdInfo2Op>(op, b); break; \
        case OP_FWD12: DISPATCH<Fwd12Op>(op, b); break; \
        case OP_WGRAD2: DISPATCH<Wgrad2Op>(op, b); break; \
        case OP_DGRAD2W1: if constexpr (EXEC_HEAVY == 1) { DISPATCH<Dgrad2W1OpT<1>>(op, b); } else if constexpr (EXEC_HEAVY == 2) { DISPATCH<Dgrad2W1Op>(op, b); } break; \
        default: break; \
      }

// the ops that may run as the sequential tail of a one-block op (the loss heads that consume a target): a small switch of its own, so that the main dispatch stays
// straight-line (with the full switch inside a loop every phase ran ~1 us longer)
#define EXEC_SWITCH_TAIL(DISPATCH) \
      switch (kid) { \
        case OP_TD_HEAD: DISPATCH<TdHeadOp>(op, b); break; \
        case OP_Q_HEAD: DISPATCH<QHeadOp>(op, b); break; \
        case OP_PER_UPDATE: DISPATCH<PerUpdateOp>(op, b); break; \
        default: break; \
      }


// the same dispatch with the record read straight from global memory at a uniform address (scalar loads): the one-launch-per-phase form below
template <class Op> __device__ __forceinline__ void exec_dispatch_g(const ExecOp* op, unsigned bid) {
  const OpPack<Op> p = *(const OpPack<Op>*)op->args;
  exec_apply<Op>(bid, op->nblocks, p);
}
// the replay ops alone (dqn_persist.h: k_dqn_replay); the gather's column table is read where it lies
#define EXEC_SWITCH_REPLAY(DISPATCH) \
      switch (kid) { \
        case OP_FILL: DISPATCH<FillOp>(op, b); break; \
        case OP_PER_SEARCH: DISPATCH<PerSearchOp>(op, b); break; \
        case OP_UNIFORM_IDS: DISPATCH<UniformIdsOp>(op, b); break; \
        case OP_GATHER_RING_ALL: { using P_ = OpPack<GatherRingAllOp>; const P_* pp = (const P_*)op->args; \
          GatherRingAllOp::run_ptr(b, op->nblocks, &pp->head, pp->tail.head, pp->tail.tail.head, pp->tail.tail.tail.head, pp->tail.tail.tail.tail.head); } break; \
        case OP_RING_IDS: DISPATCH<RingIdsOp>(op, b); break; \
        case OP_LEAF_REFRESH: DISPATCH<LeafRefreshOp>(op, b); break; \
        case OP_TREE_TOUCH: DISPATCH<TreeTouchOp>(op, b); break; \
        case OP_PER_UPDATE: DISPATCH<PerUpdateOp>(op, b); break; \
        case OP_COPY_F32: DISPATCH<CopyF32Op>(op, b); break; \
        default: break; \
      }
// One PHASE of a recorded sequence as one launch over the whole chip: block x of the grid belongs to the op whose block range contains x. The ops of a
// phase do not depend on each other, the dependency between phases is the kernel boundary -- no in-kernel barrier, no coherence question, all 256 CUs.
// A fused epoch then costs (number of phases) launches instead of (number of kernels): 13 instead of 25 for a DQN epoch, 30 instead of ~75 for SAC (10 / 27 per epoch inside a chain).
// EXEC_HEAVY: the register-hungry op bodies (Dgrad2W1Op: ~300 VGPRs) are compiled into the HEAVY instantiations only; a phase without such an op runs the light
// kernel (~100 VGPRs: four workgroups per CU instead of one -- a phase of 500-800 light blocks then takes one round over the chip instead of three)
__global__ __launch_bounds__(256) void k_phase(const ExecOp* __restrict__ ops, int n) {
  constexpr int EXEC_HEAVY = 2;
  // ops flagged sequential (barrier bit 1) own no blocks: they run in the block of the op before them, after it (see k_phase_k)
  unsigned b = blockIdx.x; int o = 0;
  for (;;) { const unsigned nb = (ops[o].barrier & 2) ? 0u : ops[o].nblocks; if (o + 1 < n && b >= nb) { b -= nb; ++o; } else break; }
  { const ExecOp* op = ops + o; const int kid = op->kid;
    EXEC_SWITCH(exec_dispatch_g) }
  while (o + 1 < n && (ops[o + 1].barrier & 2)) { __threadfence(); __syncthreads(); ++o; b = 0;
    const ExecOp* op = ops + o; const int kid = op->kid;
    EXEC_SWITCH_TAIL(exec_dispatch_g) }
}
// The same phase with its op records INSIDE the kernel arguments (<= 8 ops, <= 3.8 KB of packed arguments: every phase of the DQN / SAC epochs). With the records in
// global memory a workgroup walks block counts -> body id -> arguments -> the op's own data: three dependent scalar loads from memory the host copy has just written
// (cold in every cache) before the first useful load, ~4 us of a one-block phase's 6. The kernel-argument segment is read once, by independent scalar loads, so the
// chain is arguments -> data, as in a stand-alone launch.
// The argument struct comes in three sizes (the host copies it on every launch: with 4 KB per launch the enqueue, not the GPU, paced a SAC epoch).
#define PHASEK_MAXOPS 14
#define PHASEK_SEQ 0x10000
template <int BYTES> struct PhaseK { int32_t n; int32_t pad; int32_t kid[PHASEK_MAXOPS]; uint32_t nblocks[PHASEK_MAXOPS]; uint32_t off[PHASEK_MAXOPS]; alignas(16) unsigned char args[BYTES]; };
static_assert(sizeof(PhaseK<3840>) <= 4096, "HIP kernel arguments are limited to 4 KB");
struct KOp { const unsigned char* args; unsigned nblocks; };
template <class Op> __device__ __forceinline__ void exec_dispatch_k(const KOp* op, unsigned bid) {
  const OpPack<Op> p = *(const OpPack<Op>*)op->args;
  exec_apply<Op>(bid, op->nblocks, p);
}
// the gather's column table (456 bytes) is read where it lies in the kernel arguments instead of travelling by value (see GatherRingAllOp::run_ptr)
template <> __device__ __forceinline__ void exec_dispatch_k<GatherRingAllOp>(const KOp* op, unsigned bid) {
  using P = OpPack<GatherRingAllOp>; const P* pp = (const P*)op->args;
  GatherRingAllOp::run_ptr(bid, op->nblocks, &pp->head, pp->tail.head, pp->tail.tail.head, pp->tail.tail.tail.head, pp->tail.tail.tail.tail.head);
}
// the same for the fused prioritized search + gather (its arguments carry the same table: taking the address of a by-value copy put 176 bytes of every thread into scratch,
// on the one phase of a C3 epoch that is a chain of dependent memory round trips; round 6)
template <> __device__ __forceinline__ void exec_dispatch_k<PerSampleGatherOp>(const KOp* op, unsigned bid) {
  PerSampleGatherOp::run_ptr(bid, op->nblocks, &((const OpPack<PerSampleGatherOp>*)op->args)->head);
}
template <int BYTES, int EXEC_HEAVY>
__global__ __launch_bounds__(256) void k_phase_k(PhaseK<BYTES> by_value) {
  // read through the kernel-argument segment pointer, not through the by-value parameter: indexing the parameter at a run-time offset would make the compiler
  // copy the aggregate to private memory first
  const PhaseK<BYTES>* pk = (const PhaseK<BYTES>*)__builtin_amdgcn_kernarg_segment_ptr();
  unsigned b = blockIdx.x; int o = 0; const int n = pk->n;
#pragma unroll
  for (int q = 0; q < PHASEK_MAXOPS - 1; ++q) { const unsigned nb = pk->nblocks[q]; const bool adv = o == q && q + 1 < n && b >= nb; b -= adv ? nb : 0u; o += adv ? 1 : 0; }
  // Sequential ops (kid flagged PHASEK_SEQ; block count 0 in the table above, so no block starts on them) run in the block of the op before them, after it: a one-block
  // op whose only consumer is another one-block op (target -> loss head) shares its launch instead of paying a kernel boundary (~5 us) for a 128-float hand-over.
  // Same compute unit, write-through L1: the fence + workgroup barrier make the first op's global stores visible to the second.
  { const int kid = pk->kid[o]; const KOp kop{pk->args + pk->off[o], pk->nblocks[o]}; const KOp* op = &kop;
    EXEC_SWITCH(exec_dispatch_k) }
  while (o + 1 < n && (pk->kid[o + 1] & PHASEK_SEQ)) { __threadfence(); __syncthreads(); ++o; b = 0;
    const int kid = pk->kid[o] & (PHASEK_SEQ - 1); const KOp kop{pk->args + pk->off[o], 1u}; const KOp* op = &kop;
    EXEC_SWITCH_TAIL(exec_dispatch_k) }
}
// host: pack ops [i0, i1] of a recording into a PhaseK<BYTES>; false when they do not fit
template <int BYTES> static bool phasek_launch(const std::vector<ExecOp>& ops, size_t i0, size_t i1, unsigned blocks, hipStream_t st) {
  PhaseK<BYTES> pk; pk.n = 0; pk.pad = 0; size_t used = 0;
  for (size_t i = i0; i <= i1; ++i) { const ExecOp& e = ops[i]; if (!e.nblocks) continue;
    const size_t raw = (size_t)(e.abytes > 0 ? e.abytes : CRUX_EXEC_ARG_BYTES), ab = (raw + 15) & ~(size_t)15;
    if (pk.n >= PHASEK_MAXOPS || used + ab > (size_t)BYTES) return false;
    const bool seq = (e.barrier & 2) != 0;
    pk.kid[pk.n] = e.kid | (seq ? PHASEK_SEQ : 0); pk.nblocks[pk.n] = seq ? 0u : e.nblocks; pk.off[pk.n] = (uint32_t)used; memcpy(pk.args + used, e.args, raw); used += ab; pk.n++; }
  if (pk.n == 0) return false;
  for (int q = pk.n; q < PHASEK_MAXOPS; ++q) { pk.kid[q] = 0; pk.nblocks[q] = 0; pk.off[q] = 0; }
  int heavy = 0;      // 0: no Dgrad2W1Op in the phase; 1: its O3 = 1 form (critics / no folded output layer); 2: the O3 = 4 form
  for (int q = 0; q < pk.n; ++q) if ((pk.kid[q] & (PHASEK_SEQ - 1)) == OP_DGRAD2W1) { Dgrad2Args a; memcpy(&a, pk.args + pk.off[q], sizeof a); heavy = std::max(heavy, (a.z.W3 && a.z.out3 > 1) ? 2 : 1); }
  if (heavy == 2) hipLaunchKernelGGL((k_phase_k<BYTES, 2>), dim3(blocks), dim3(256), 0, st, pk);
  else if (heavy == 1) hipLaunchKernelGGL((k_phase_k<BYTES, 1>), dim3(blocks), dim3(256), 0, st, pk);
  else hipLaunchKernelGGL((k_phase_k<BYTES, 0>), dim3(blocks), dim3(256), 0, st, pk);
  return true;
}
// would phasek_launch<3840> take ops [i0, i1]? (the same packing rules, nothing launched)
static bool phasek_fits(const std::vector<ExecOp>& ops, size_t i0, size_t i1) {
  int n = 0; size_t used = 0;
  for (size_t i = i0; i <= i1; ++i) { const ExecOp& e = ops[i]; if (!e.nblocks) continue;
    const size_t raw = (size_t)(e.abytes > 0 ? e.abytes : CRUX_EXEC_ARG_BYTES), ab = (raw + 15) & ~(size_t)15;
    if (n >= PHASEK_MAXOPS || used + ab > (size_t)3840) return false;
    used += ab; ++n; }
  return n > 0;
}
__global__ __launch_bounds__(256) void k_exec(const ExecOp* __restrict__ ops, int nops, unsigned* ctr, int xcd, int32_t* status, int flags) {
  constexpr int EXEC_HEAVY = 2;
  if (xcd >= 0 && (int)(blockIdx.x & 7) != xcd) return;
  const unsigned wg = xcd >= 0 ? blockIdx.x >> 3 : blockIdx.x, G = xcd >= 0 ? gridDim.x >> 3 : gridDim.x;
  // op records are staged through LDS one op ahead: the 512-byte record of op o+1 is fetched while op o runs and its barrier is waited for, so
  // dispatch reads its arguments from LDS instead of paying two dependent L2 round trips per op (1.5 us measured, tools/dbg_nops.py)
  __shared__ ExecOp op_s[2];
  constexpr int OPW = (int)(sizeof(ExecOp) / 4);
  if ((int)threadIdx.x < OPW && nops > 0) ((uint32_t*)&op_s[0])[threadIdx.x] = ((const uint32_t*)&ops[0])[threadIdx.x];
  __syncthreads();
  unsigned phase = 0, off = 0;
  for (int o = 0; o < nops; ++o) {
    const ExecOp* op = &op_s[o & 1];
    uint32_t nxt = 0;
    if ((int)threadIdx.x < OPW && o + 1 < nops) nxt = ((const uint32_t*)&ops[o + 1])[threadIdx.x];
    const int kid = __builtin_amdgcn_readfirstlane(op->kid); const unsigned nb = __builtin_amdgcn_readfirstlane(op->nblocks);
    unsigned long long* tdbg = (unsigned long long*)(ctr + 1024);          // CRUX_EXEC_FLAGS & 8: per-op timestamps of workgroups 0 and 1 (s_memtime, 100 MHz)
    if ((flags & 8) && wg < 2 && threadIdx.x == 0 && o < 512) tdbg[(wg * 512 + o) * 3 + 0] = __builtin_amdgcn_s_memtime();
    // blocks are dealt to the workgroups round-robin over the whole PHASE (ops without a barrier between them), not per op: a phase made of a
    // one-block op and two GEMMs then keeps 1 + 16 + 32 different workgroups busy instead of giving workgroup 0 a block of each
    const unsigned b0 = (wg + G - off) % G;
    for (unsigned b = b0; b < nb; b += G) {
      EXEC_SWITCH(exec_dispatch)
      __syncthreads();                                   // the bodies' static LDS is reused by the next block / op of this workgroup
    }
    const int bar = __builtin_amdgcn_readfirstlane(op->barrier) & 1;
    if ((flags & 8) && wg < 2 && threadIdx.x == 0 && o < 512) tdbg[(wg * 512 + o) * 3 + 1] = __builtin_amdgcn_s_memtime();
    if ((int)threadIdx.x < OPW && o + 1 < nops) ((uint32_t*)&op_s[(o + 1) & 1])[threadIdx.x] = nxt;
    off = bar ? 0u : (off + nb) % G;
    if (bar) { phase += 1; if (!exec_barrier(ctr, wg, G, phase, flags)) { if (threadIdx.x == 0 && wg == 0) status[0] = CRUX_EHIP; return; } }
    else __syncthreads();
    if ((flags & 8) && wg < 2 && threadIdx.x == 0 && o < 512) tdbg[(wg * 512 + o) * 3 + 2] = __builtin_amdgcn_s_memtime();
  }
}

#include "dqn_persist.h"

// ---- host: recording -------------------------------------------------------------------------------------------------------------------
static ExecRec* rec_of(crux_ctx* c) { return (ExecRec*)c->rec; }
// CRUX_EXEC_PERSISTENT (development): the one-XCD persistent executor k_exec, for recordings that are run one at a time and waited for. Chained epochs ignore the switch: their
// op tags assume the phase plan (sequential one-block groups, tile ops), and an asynchronous chain could not read k_exec's status word back.
static bool exec_persistent_on(crux_ctx* c) { const ExecRec* r = (const ExecRec*)c->rec; return crux_sw().exec_persistent && !(r && r->chain); }
extern "C" int32_t crux_ensure_aux_stream(crux_ctx* c);      // train.hip: a second stream on its own hardware queue (probed)

// The persistent two-kernel form of a recorded chain of DQN-family epochs (dqn_persist.h). The recording holds every op of every epoch; the learner's ops (tile GEMMs,
// target, head, norm, info, Adam) are replaced by k_dqn_learn, the replay ops are handed to k_dqn_replay grouped into the stages the two kernels synchronise on.
#define DQP_BUF_BYTES ((size_t)6 << 20)
struct DqpBuf { static constexpr size_t oCtrL = 0, oFlags = 2048, oSsq = 2304, oTab = 4096, oPtr = 4096 + 32768, oG = 65536, oZ = oG + 32768, oP = oZ + 262144, oMV = oP + ((size_t)DQP_G * 128 * 256 * 4), oDbg = oMV + (1 << 20), oWT = oDbg + 16384; };      // oWT: 16 x 16 KB      // oMV: 16 x 2 x <= 4360 floats = 558 KB
template <int IN, int OUT, int BT> static int32_t dqp_launch_learn(crux_ctx* c, const DqpArgs& a, hipStream_t st) {
  constexpr size_t lds = sizeof(float) * (size_t)DqpL<IN, OUT, BT>::TOTAL;
  static bool attr_dev[16] = {}; bool& attr = attr_dev[c->device & 15];      // (per device: a second device in the process sets the attribute for itself)
  if (!attr) { HIPCHK(c, hipFuncSetAttribute((const void*)k_dqn_learn<IN, OUT, BT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; }
  hipLaunchKernelGGL((k_dqn_learn<IN, OUT, BT>), dim3(8 * DQP_G), dim3(256), lds, st, a);
  return crux_launch_check(c, "k_dqn_learn");
}
static bool dqp_shape(int in, int out, int64_t B) { return ((in == 8 && out == 4) || (in == 4 && out == 2)) && (B == 128 || B == 64); }
static int32_t dqp_build(ExecRec* r) {        // the replay table; CRUX_EUNSUP when the recording holds something the two kernels do not know
  ExecRec::Dqp& d = r->dqp; const int n = d.n_epochs; if (n < 1 || (int)r->epoch_marks.size() != n || (int)r->readbacks.size() != n) return CRUX_EUNSUP;
  d.tab.assign(4 * (size_t)n, 0);
  for (int e = 0; e < n; ++e) {
    const size_t i0 = r->epoch_marks[e], i1 = e + 1 < n ? r->epoch_marks[e + 1] : r->ops.size();
    std::vector<std::pair<int, int>> A, Cs; bool head = false;      // (phase, op index)
    for (size_t i = i0; i < i1; ++i) { const int kid = r->ops[i].kid;
      switch (kid) {
        case OP_PER_SEARCH: case OP_UNIFORM_IDS: case OP_PER_SAMPLE: A.push_back({0, (int)i}); break;
        case OP_GATHER_RING_ALL: case OP_RING_IDS: case OP_COPY_F32: case OP_FILL: if (head) return CRUX_EUNSUP; A.push_back({1, (int)i}); break;
        case OP_PER_UPDATE: if (head) Cs.push_back({0, (int)i}); else A.push_back({2, (int)i}); break;
        case OP_LEAF_REFRESH: if (!head) return CRUX_EUNSUP; Cs.push_back({1, (int)i}); break;
        case OP_TREE_TOUCH: if (!head) return CRUX_EUNSUP; Cs.push_back({2, (int)i}); break;
        case OP_TD_HEAD: head = true; break;
        case OP_GEMM: case OP_FWD12: case OP_WGRAD2: case OP_DGRAD2W1: case OP_DQN_TARGET: case OP_SUMSQ2: case OP_TD_INFO: case OP_ADAM_GATED: case OP_ADAM_ADVANCE: case OP_ADAM_SELF: case OP_ADAM_ADVANCE_SELF: break;      // the learner kernel's work
        default: return CRUX_EUNSUP; } }
    if (!head || A.empty()) return CRUX_EUNSUP;
    auto emit = [&](std::vector<std::pair<int, int>>& v, int slot) {
      std::stable_sort(v.begin(), v.end(), [](const std::pair<int, int>& x, const std::pair<int, int>& y) { return x.first < y.first; });
      const int32_t at = (int32_t)d.tab.size(); d.tab[4 * e + slot] = at; d.tab[4 * e + slot + 1] = (int32_t)v.size();
      for (size_t k = 0; k < v.size(); ++k) { d.tab.push_back(v[k].second); d.tab.push_back((k + 1 == v.size() || v[k + 1].first != v[k].first) ? 1 : 0); } };
    emit(A, 0); emit(Cs, 2);
  }
  return d.tab.size() * 4 <= 32768 ? CRUX_OK : CRUX_EUNSUP;
}
static int32_t dqp_launch(crux_ctx* c, ExecRec* r) {      // inside crux_exec_run: the op list is uploaded, d_ctr is zeroed
  ExecRec::Dqp& d = r->dqp; crux_mlp* net = (crux_mlp*)d.net; crux_mlp* tn = (crux_mlp*)d.tnet; crux_buffer* b = (crux_buffer*)d.batch;
  const int n = d.n_epochs; char* buf = (char*)r->dqp_buf;
  std::vector<void*> ptrs(2 * (size_t)n);
  for (int e = 0; e < n; ++e) { ptrs[e] = (void*)r->readbacks[e].d_info; ptrs[n + e] = (void*)r->readbacks[e].d_status; }
  if (ptrs.size() * sizeof(void*) > 16384) return crux_fail(c, CRUX_EINVAL, "dqn epochs: chain of %d epochs", n);
  HIPCHK(c, hipMemsetAsync(buf, 0, 4096, c->stream));
  HIPCHK(c, hipMemcpyAsync(buf + DqpBuf::oTab, d.tab.data(), d.tab.size() * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(buf + DqpBuf::oPtr, ptrs.data(), ptrs.size() * sizeof(void*), hipMemcpyHostToDevice, c->stream));
  DqpArgs a{}; a.p = net->p; a.m = net->m; a.v = net->v; a.bp = net->bp; a.pt = tn->p;
  for (int l = 0; l < 3; ++l) { a.woff[l] = net->nd.woff[l]; a.boff[l] = net->nd.boff[l]; }
  a.eta = net->eta; a.b1 = net->b1; a.b2 = net->b2; a.eps = net->eps;
  a.S = (const float*)b->col[CRUX_COL_S]; a.SP = (const float*)b->col[CRUX_COL_SP]; a.A = (const uint8_t*)b->col[CRUX_COL_A]; a.R = (const float*)b->col[CRUX_COL_R];
  a.DONE = (const uint8_t*)b->col[CRUX_COL_DONE]; a.W = d.use_weight ? (const float*)b->col[CRUX_COL_WEIGHT] : nullptr;
  a.gamma = d.gamma; a.n_epochs = n; a.err = d.d_err;
  a.dinfo = (float* const*)(buf + DqpBuf::oPtr); a.dstatus = (int32_t* const*)(buf + DqpBuf::oPtr + n * sizeof(void*));
  a.zbuf = (float*)(buf + DqpBuf::oZ); a.pbuf = (float*)(buf + DqpBuf::oP); a.gbuf = (float*)(buf + DqpBuf::oG); a.mv2 = (float*)(buf + DqpBuf::oMV); a.wtg = (float*)(buf + DqpBuf::oWT); a.ssq = (double*)(buf + DqpBuf::oSsq);
  a.ctrL = (unsigned*)(buf + DqpBuf::oCtrL); a.flags = (unsigned*)(buf + DqpBuf::oFlags); a.status = (int32_t*)(r->d_ctr + 264); a.xcd = 0;
  a.dbg = crux_sw().dqp_debug ? (unsigned long long*)(buf + DqpBuf::oDbg) : nullptr; if (a.dbg) HIPCHK(c, hipMemsetAsync(a.dbg, 0, 8192, c->stream));
  // the replay kernel first, on the second stream (its own hardware queue): it samples epoch 0 while the learner loads its parameters
  HIPCHK(c, hipEventRecord(c->aux_ev0, c->stream)); HIPCHK(c, hipStreamWaitEvent(c->aux_stream, c->aux_ev0, 0));
  hipLaunchKernelGGL(k_dqn_replay, dim3(8 * DQP_R), dim3(256), 0, c->aux_stream, (const ExecOp*)r->d_ops, (const int32_t*)(buf + DqpBuf::oTab), n, r->d_ctr, a.flags, a.xcd, a.status, a.dbg);
  HIPCHK(c, hipEventRecord(c->aux_ev1, c->aux_stream));
  int32_t rc = CRUX_EUNSUP;
  if (d.in == 8 && d.out == 4 && d.bt == 8) rc = dqp_launch_learn<8, 4, 8>(c, a, c->stream);
  else if (d.in == 8 && d.out == 4 && d.bt == 4) rc = dqp_launch_learn<8, 4, 4>(c, a, c->stream);
  else if (d.in == 4 && d.out == 2 && d.bt == 8) rc = dqp_launch_learn<4, 2, 8>(c, a, c->stream);
  else if (d.in == 4 && d.out == 2 && d.bt == 4) rc = dqp_launch_learn<4, 2, 4>(c, a, c->stream);
  HIPCHK(c, hipStreamWaitEvent(c->stream, c->aux_ev1, 0));
  if (crux_sw().dqp_debug) { HIPCHK(c, hipStreamSynchronize(c->stream)); unsigned h[1024]; HIPCHK(c, hipMemcpy(h, buf, 4096, hipMemcpyDeviceToHost));
    fprintf(stderr, "[dqp] rc %d n %d flags %u %u %u  ctrL", rc, n, h[512], h[513], h[514]); for (int q = 0; q < 16; ++q) fprintf(stderr, " %u", h[q]); fprintf(stderr, " abort %u  ssq", h[256]);
    const double* sq = (const double*)(h + 576); for (int q = 0; q < 16; ++q) fprintf(stderr, " %.3g", sq[q]); { unsigned hc[300]; (void)hipMemcpy(hc, r->d_ctr, 1200, hipMemcpyDeviceToHost); fprintf(stderr, "  ctrR"); for (int q = 0; q < 32; ++q) fprintf(stderr, " %u", hc[q]); fprintf(stderr, " abortR %u st %d whyL %d whyR %d claims %u %u", hc[256], (int)hc[264], (int)hc[265], (int)hc[266], h[516], h[517]); }
    fprintf(stderr, "  xcc L"); for (int q = 0; q < 16; ++q) fprintf(stderr, " %u", h[512 + 16 + q]); fprintf(stderr, " R"); for (int q = 0; q < 32; ++q) fprintf(stderr, " %u", h[512 + 32 + q]); fprintf(stderr, "  tab"); for (size_t q = 0; q < d.tab.size() && q < 40; ++q) fprintf(stderr, " %d", d.tab[q]); fprintf(stderr, "\n");
    static int dumps = 0; if (dumps++ == 2) { unsigned long long t[1024]; HIPCHK(c, hipMemcpy(t, buf + DqpBuf::oDbg, 8192, hipMemcpyDeviceToHost));
      fprintf(stderr, "[dqp] learner wg 0 (us since its start; ~2.04 GHz ticks; after the start stamp, per epoch: batch ready | forward | barrier b | z loaded | sync | head | stats | acked | sync | dW3 dZ2 P dW2 | barrier c | dH1 dW1 | barrier d | Adam):"); for (int q = 0; q < 512 && t[q]; ++q) fprintf(stderr, " %.1f", (double)(t[q] - t[0]) / 2040.0);
      fprintf(stderr, "\n[dqp] replay wg 0 (us since the LEARNER's start):"); for (int q = 512; q < 1024 && t[q]; ++q) fprintf(stderr, " %.1f", ((double)t[q] - (double)t[0]) / 2040.0); fprintf(stderr, "\n"); } }
  return rc;
}
bool crux_exec_recording(const crux_ctx* c) { return c && c->rec && ((const ExecRec*)c->rec)->active; }
int32_t crux_exec_begin(crux_ctx* c) {
  if (!c->rec) c->rec = new ExecRec();
  ExecRec* r = rec_of(c);
  if (r->active) return crux_fail(c, CRUX_EINVAL, "executor: a recording is already open on this context");
  if (!r->small) { if (hipMalloc(&r->small, EXEC_SMALL_BYTES) != hipSuccess) return crux_fail(c, CRUX_ENOMEM, "executor: small region"); r->small_cap = EXEC_SMALL_BYTES; }
  if (!r->d_ctr) { if (hipMalloc(&r->d_ctr, 65536) != hipSuccess) return crux_fail(c, CRUX_ENOMEM, "executor: barrier counter"); }
  if (!crux_scratch(c, (size_t)32 << 20)) return crux_fail(c, CRUX_ENOMEM, "executor: scratch");     // pre-sized: the scratch block must not move while pointers into it are recorded
  r->scratch_floor = c->scratch_bytes; r->scratch_off = 0;
  r->ops.clear(); r->readbacks.clear(); r->small_off = 0; r->active = true;
  r->chain_tags.clear(); r->chain_base = 0; r->chain_ok = true; r->epoch_marks.clear(); r->dqp.on = false;
  return CRUX_OK;
}
// frees everything a context's executor ever allocated (called by crux_ctx_destroy after the stream has drained)
void crux_exec_destroy(crux_ctx* c) {
  if (!c || !c->rec) return;
  ExecRec* r = rec_of(c);
  if (r->small) (void)hipFree(r->small);
  if (r->d_ctr) (void)hipFree(r->d_ctr);
  if (r->d_ops) (void)hipFree(r->d_ops);
  if (r->h_stage) (void)hipHostFree(r->h_stage);
  if (r->dqp_buf) (void)hipFree(r->dqp_buf);
  for (int k = 0; k < 4; ++k) { if (r->h_ring[k]) (void)hipHostFree(r->h_ring[k]); if (r->h_ring_ev[k]) (void)hipEventDestroy((hipEvent_t)r->h_ring_ev[k]); }
  delete r; c->rec = nullptr;
}
void crux_exec_abort(crux_ctx* c) { if (c->rec) { ExecRec* r = rec_of(c); r->active = false; r->ops.clear(); r->readbacks.clear(); } }
ExecOp* crux_exec_new_op(crux_ctx* c, int kid, unsigned nblocks) {
  ExecRec* r = rec_of(c); r->ops.emplace_back(); ExecOp* op = &r->ops.back();
  op->kid = kid; op->nblocks = nblocks; op->barrier = 1; op->abytes = 0; return op;
}
void* crux_exec_scratch(crux_ctx* c, size_t bytes) {
  ExecRec* r = rec_of(c); bytes = (bytes + 255) / 256 * 256;
  if (r->scratch_off + bytes > r->scratch_floor) return nullptr;
  void* p = (char*)c->scratch + r->scratch_off; r->scratch_off += bytes; return p;
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
// Phases: ops that do not depend on each other share a barrier. The caller assigns every recorded op a phase number (non-decreasing along every
// dependency chain); the list is stably sorted by phase and only the last op of a phase keeps its barrier.
static size_t exec_mark(crux_ctx* c) { return rec_of(c)->ops.size(); }
// Tags are 4 * phase + sub: sub 0 = an ordinary op of the phase; sub 1 = head of a sequential group, sorted behind the ordinary ops; sub 2 = runs in the block of
// the op sorted before it, after it (both must be one-block ops: the caller only tags such pairs -- PH_SEQ_HEAD / PH_SEQ_TAIL below).
static int32_t exec_schedule(crux_ctx* c, const std::vector<int>& phase) {
  ExecRec* r = rec_of(c); const size_t n = r->ops.size();
  if (phase.size() != n) return crux_fail(c, CRUX_EHIP, "executor: %zu phase tags for %zu ops", phase.size(), n);
  std::vector<size_t> idx(n); for (size_t i = 0; i < n; ++i) idx[i] = i;
  std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return phase[a] < phase[b]; });
  std::vector<ExecOp> out(n);
  for (size_t k = 0; k < n; ++k) { out[k] = r->ops[idx[k]]; out[k].barrier = (k + 1 == n || (phase[idx[k + 1]] >> 2) != (phase[idx[k]] >> 2)) ? 1 : 0;
    if ((phase[idx[k]] & 3) == 2) {
      // the in-block tail switch (EXEC_SWITCH_TAIL) knows these bodies only: anything else would be skipped silently
      if (out[k].kid != OP_TD_HEAD && out[k].kid != OP_Q_HEAD && out[k].kid != OP_PER_UPDATE) return crux_fail(c, CRUX_EHIP, "executor: op %d cannot run as a sequential tail", out[k].kid);
      if (k == 0 || (phase[idx[k - 1]] >> 2) != (phase[idx[k]] >> 2) || out[k].nblocks != 1 || out[k - 1].nblocks != 1) return crux_fail(c, CRUX_EHIP, "executor: a sequential op without a one-block predecessor in its phase");
      out[k].barrier |= 2; } }
  r->ops.swap(out); return CRUX_OK;
}
// launches of the dense engine inside a recorded chain: a tile GEMM, or one of the fused block kernels of dense_fused.h (round 4). The phase plans below count THESE
// in recording order; a network's forward pass is nf of them (nf = L, or L - 1 when layers 0 + 1 are one Fwd12Op), its pullback with parameter gradients 2 L - 1
// (weight + data gradient per layer, no data gradient for layer 0) in L phases -- or 2 (L - 1) in L - 1 phases when Wgrad2Op || Dgrad2W1Op take layers 1 and 0 together.
static inline bool is_mm(int kid) { return kid == OP_GEMM || kid == OP_FWD12 || kid == OP_WGRAD2 || kid == OP_DGRAD2W1; }
// With the output layer's data gradient folded into the pair (three layers, out <= 4: crux_dense_bwd_fused3) the whole pullback is ONE phase of three launches
// (output-layer dW, Wgrad2Op, Dgrad2W1Op), and an input-gradient chain is two (Dgrad2W1Op, layer 0's data gradient).
struct NetPlan { int nf, nbops, nb, dq; bool one; int stage(int k) const { return one ? 0 : k / 2; } };
static inline NetPlan net_plan(const crux_mlp* n, int64_t B) { const int L = n->nd.L; const bool ff = crux_dense_fwd_fused(n), fb = crux_dense_bwd_fused(n, B), f3 = crux_dense_bwd_fused3(n, B);
  return NetPlan{ff ? L - 1 : L, f3 ? 3 : fb ? 2 * (L - 1) : 2 * L - 1, f3 ? 1 : fb ? L - 1 : L, f3 ? 2 : L, f3}; }
#define PH_SEQ_HEAD (1 << 12)
#define PH_SEQ_TAIL (2 << 12)
static inline int ph_tag(int p_mapped, int sub) { return 4 * p_mapped + sub; }

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
    const bool async = r->async; r->async = false;
    void* stage = nullptr;
    r->ops.back().barrier &= 2;
    // an asynchronous chain whose phases all travel in kernel arguments needs neither the device copy of the list nor the zeroed counters (no persistent form, no status
    // read-back): two stream operations less between chains (they sit IN the stream there, ~15 us per chain)
    bool lean = false;
    if (async && !r->dqp.on && !crux_sw().exec_no_kernarg) { lean = true;
      size_t i0 = 0;
      while (i0 < nops && lean) { size_t i1 = i0; unsigned blocks = 0; for (;;) { blocks += (r->ops[i1].barrier & 2) ? 0u : r->ops[i1].nblocks; if ((r->ops[i1].barrier & 1) || i1 + 1 == nops) break; ++i1; }
        if (blocks && !phasek_fits(r->ops, i0, i1)) lean = false;
        i0 = i1 + 1; } }
    if (lean) { /* nothing to upload */ }
    else if (async) {      // a staging buffer of its own for this chain: the previous chains' uploads may not have executed yet
      const unsigned k = r->h_ring_next++ & 3u;
      if (!r->h_ring_ev[k]) { hipEvent_t ev; HIPCHK(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming)); r->h_ring_ev[k] = ev; }
      else HIPCHK(c, hipEventSynchronize((hipEvent_t)r->h_ring_ev[k]));      // the upload that last used this buffer (four chains ago) has run
      if (r->h_ring_cap[k] < ob) { if (r->h_ring[k]) (void)hipHostFree(r->h_ring[k]); r->h_ring_cap[k] = ob * 2 + 4096;
        if (hipHostMalloc(&r->h_ring[k], r->h_ring_cap[k], hipHostMallocDefault) != hipSuccess) { r->h_ring[k] = nullptr; r->h_ring_cap[k] = 0; return crux_fail(c, CRUX_ENOMEM, "executor: staging ring"); } }
      stage = r->h_ring[k];
      r->ops.back().barrier &= 2;
      memcpy(stage, r->ops.data(), ob);
      HIPCHK(c, hipMemcpyAsync(r->d_ops, stage, ob, hipMemcpyHostToDevice, c->stream));
      HIPCHK(c, hipEventRecord((hipEvent_t)r->h_ring_ev[k], c->stream));
    } else {
    if (r->h_stage_cap < need_h) { if (r->h_stage) { HIPCHK(c, hipStreamSynchronize(c->stream)); (void)hipHostFree(r->h_stage); } r->h_stage_cap = need_h * 2 + 4096; if (hipHostMalloc(&r->h_stage, r->h_stage_cap, hipHostMallocDefault) != hipSuccess) { r->h_stage = nullptr; r->h_stage_cap = 0; return crux_fail(c, CRUX_ENOMEM, "executor: staging"); } }
    r->ops.back().barrier &= 2;
    memcpy(r->h_stage, r->ops.data(), ob);
    HIPCHK(c, hipMemcpyAsync(r->d_ops, r->h_stage, ob, hipMemcpyHostToDevice, c->stream));
    }
    if (!lean) HIPCHK(c, hipMemsetAsync(r->d_ctr, 0, 2048, c->stream));
    // CRUX_EXEC_PERSISTENT (development): the one-XCD persistent executor, for synchronous calls only -- its status word (a workgroup that does not reach a barrier: the grid was
    // not co-resident, which the register-heavy block kernels of round 4 cause) is read back by the synchronous path; an asynchronous chain would go on with garbage
    const bool persistent = crux_sw().exec_persistent && !r->async && r->chain_tags.empty();      // (a chained recording carries op tags)
    if (r->dqp.on) { r->dqp.on = false; rc = dqp_launch(c, r); if (rc) return rc; }
    else if (!persistent) {
      // default: one launch per phase over the whole chip (see k_phase). Measured against the persistent one-XCD form (CRUX_EXEC_PERSISTENT=1): the latter
      // saves the launches but runs every op on 32 CUs behind one L2 and pays ~2 us per barrier; DESIGN 4.3 has the numbers.
      size_t i0 = 0;
      while (i0 < nops) { size_t i1 = i0; unsigned blocks = 0; for (;;) { blocks += (r->ops[i1].barrier & 2) ? 0u : r->ops[i1].nblocks; if ((r->ops[i1].barrier & 1) || i1 + 1 == nops) break; ++i1; }
        if (blocks) {
          // the phase's records travel in the kernel arguments when they fit (see k_phase_k); zero-block ops are dropped there
          const bool no_kernarg = crux_sw().exec_no_kernarg;      // tests: every phase through the global-record form
          if (no_kernarg || (!phasek_launch<384>(r->ops, i0, i1, blocks, c->stream) && !phasek_launch<1024>(r->ops, i0, i1, blocks, c->stream) && !phasek_launch<3840>(r->ops, i0, i1, blocks, c->stream)))
            hipLaunchKernelGGL(k_phase, dim3(blocks), dim3(256), 0, c->stream, (const ExecOp*)r->d_ops + i0, (int)(i1 - i0 + 1));
        }
        i0 = i1 + 1; }
    } else {
    // the counter barrier relies on one shared L2: all workgroups on XCD 0 (workgroup i of a grid lands on XCD i mod 8, verified by the placement probe)
    const int xcd = crux_x2_placement_ok_c(c) ? 0 : -2;
    if (xcd == -2) return crux_fail(c, CRUX_EUNSUP, "executor: workgroups are not placed round-robin over the XCDs on this device");
    const int G = EXEC_G;
    const int xflags = crux_sw().exec_flags;      // 8: per-op timestamps of workgroups 0 and 1, printed below
    hipLaunchKernelGGL(k_exec, dim3(G * 8), dim3(256), 0, c->stream, (const ExecOp*)r->d_ops, (int)nops, r->d_ctr, xcd, (int32_t*)(r->d_ctr + 264), xflags);
    if (xflags & 8) { static int dumps = 0;
      std::vector<unsigned long long> tb(2 * 512 * 3); HIPCHK(c, hipStreamSynchronize(c->stream));
      HIPCHK(c, hipMemcpy(tb.data(), r->d_ctr + 1024, tb.size() * 8, hipMemcpyDeviceToHost));
      if (dumps++ == 3) { double tw = 0, tbar = 0; const size_t nn = nops < 512 ? nops : 512;
        for (size_t o = 0; o < nn; ++o) { const double w0 = (tb[o * 3 + 1] - tb[o * 3]) * 1e-3, b0 = (tb[o * 3 + 2] - tb[o * 3 + 1]) * 1e-3; tw += w0; tbar += b0;      // s_memtime ticks are shader cycles (~2 GHz)
          fprintf(stderr, "[k_exec] op %3zu kid %2d blocks %4u barrier %d  work %7.2f kcycles  wait %7.2f kcycles\n", o, r->ops[o].kid, r->ops[o].nblocks, r->ops[o].barrier, w0, b0); }
        fprintf(stderr, "[k_exec] %zu ops: work %.1f kcycles, barrier / wait %.1f kcycles (workgroup 0; shader cycles, ~0.5 ns each)\n", nn, tw, tbar); } }
    }
    rc = crux_launch_check(c, "k_exec"); if (rc) return rc;
    if (async) { r->ops.clear(); r->readbacks.clear(); return CRUX_OK; }      // nothing is read back: the list itself copied the info rows where the caller wants them
    char* hb = (char*)r->h_stage + ob;
    for (size_t k = 0; k < r->readbacks.size(); ++k) { char* h = hb + k * (sizeof(float) * CRUX_INFO_N + 16);
      HIPCHK(c, hipMemcpyAsync(h, r->readbacks[k].d_info, sizeof(float) * CRUX_INFO_N, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipMemcpyAsync(h + sizeof(float) * CRUX_INFO_N, r->readbacks[k].d_status, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream)); }
    int32_t* hst = (int32_t*)(hb + rb);
    HIPCHK(c, hipMemcpyAsync(hst, r->d_ctr + 264, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
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
// softq_target(alpha) (rl/softq.jl:4-13). Lives in this translation unit so that the stand-alone kernel and the executor's phase kernel are compiled under the same
// floating-point contraction setting (exp / log are inlined library code: the two forms must agree bit for bit).
int32_t crux_mlp_forward_impl(crux_mlp* net, const float* d_x, int64_t B, float* d_y, const float* params_override);
__global__ void k_softq_target(const float* __restrict__ q, int nout, const float* __restrict__ r, const uint8_t* __restrict__ done, float gamma, float alpha, int64_t n, float* __restrict__ y) { SoftqTargetOp::run(blockIdx.x, gridDim.x, q, nout, r, done, gamma, alpha, n, y); }
extern "C" int32_t crux_softq_target(crux_mlp* tn, crux_buffer* batch, float gamma, float alpha, float* d_y) {
  if (!tn || !batch || !d_y) return CRUX_EINVAL;
  crux_ctx* c = tn->ctx; const int64_t n = batch->elements; if (n == 0) return CRUX_OK;
  if (!(alpha > 0.f)) return crux_fail(c, CRUX_EINVAL, "softq_target: alpha must be positive");
  const int nout = tn->nd.dims[tn->nd.L];
  float* q = (float*)crux_scratch(c, 4 * (size_t)n * nout + 256); if (!q) return crux_fail(c, CRUX_ENOMEM, "softq_target: scratch");
  int32_t rc;
  if (tn->nd.maxdim >= CRUX_DENSE_MIN_WIDTH) { rc = crux_dense_forward(tn, (const float*)batch->col[CRUX_COL_SP], n, c->stream); if (rc) return rc; q = crux_dense_act(tn, tn->nd.L); }
  else { rc = crux_mlp_forward_impl(tn, (const float*)batch->col[CRUX_COL_SP], n, q, nullptr); if (rc) return rc; }
  CRUX_RUN(c, SoftqTargetOp, OP_SOFTQ_TARGET, k_softq_target, (unsigned)((n + 255) / 256), 256, c->stream, q, nout, (const float*)batch->col[CRUX_COL_R], (const uint8_t*)batch->col[CRUX_COL_DONE], gamma, alpha, n, d_y);
  return crux_launch_check(c, "k_softq_target");
}

extern "C" {
int32_t crux_per_sample(crux_buffer* target, crux_buffer* source, int64_t B, const double* rands, float beta, uint64_t i);
int32_t crux_uniform_sample(crux_buffer* target, crux_buffer* source, int64_t B, const int64_t* ids, uint64_t i);
int32_t crux_dqn_target(crux_mlp* tn, crux_buffer* batch, float gamma, float* d_y);
int32_t crux_softq_target(crux_mlp* tn, crux_buffer* batch, float gamma, float alpha, float* d_y);
int32_t crux_td_step(crux_mlp* net, crux_buffer* batch, const float* d_y, int32_t use_weight, float* info_out);
int32_t crux_td_step_with_error(crux_mlp* net, crux_buffer* batch, const float* d_y, int32_t use_weight, float* d_err, float* info_out);
int32_t crux_per_update_device(crux_buffer* b, const int64_t* d_ids, const float* d_v, int64_t n);
int32_t crux_polyak(crux_mlp* to, const crux_mlp* from, float tau);

// The DQN-family epoch on the fused block kernels and DqnTdTileOp (sac_fused.h, round 4): 8 phases, FIVE launches per epoch inside a chain (round 3: 13 / 9).
//   0 [uniform ids] | 1 prioritized search + gather in one launch (or the gather), zero-fills | 2 layers 0+1 of Q(s) and Q-(s') | 3 both output layers + target + td head +
//   update_priorities! per 16-sample tile | 4 the whole pullback ; leaf re-sums | 5 norm ; root paths | 6 info, Adam | 7 beta-power advance
// In a chain the sampling of epoch e + 1 (phases 0, 1) sits beside the norm and Adam of epoch e -- after the root paths of phase 5 --, its phase 2 beside the advance.
static bool dqn_tile_case(crux_mlp* net, crux_mlp* tnet, crux_buffer* source, crux_buffer* batch) {
  const bool on = crux_sw().sac_tile_ops && !exec_persistent_on(net->ctx) && !crux_sw().no_fused_epoch && !crux_sw().no_chained_epochs;
  const int64_t B = batch->capacity; crux_ctx* c = net->ctx;
  if (!on || c->per_split_sample || !net->has_adam) return false;
  if (source->prioritized && !crux_per_fused_gather()) return false;
  crux_mlp* both[2] = {net, tnet};
  for (crux_mlp* n : both) if (n->nd.L != 3 || !crux_dense_fwd_fused(n) || n->nd.acts[2] != CRUX_ACT_IDENTITY || n->nd.dims[2] != net->nd.dims[2] || n->nd.dims[3] != net->nd.dims[3] || n->nd.dims[0] != batch->obs_dim) return false;
  if (!crux_dense_bwd_fused3(net, B)) return false;
  return batch->act_kind == CRUX_ACTION_DISCRETE && batch->act_dim == net->nd.dims[3] && net->nd.dims[3] <= 4;
}
static int32_t dqn_epoch_tiles(crux_mlp* net, crux_mlp* tnet, crux_buffer* source, crux_buffer* batch, float gamma, float softq_alpha, int32_t use_weight, float beta,
                               uint64_t sample_counter, float* info_out, float* d_y, float* d_err) {
  crux_ctx* c = net->ctx; const int64_t B = batch->capacity; const bool per = source->prioritized; const int nout = net->nd.dims[3], K = net->nd.dims[2];
  ExecRec* r = rec_of(c); const int base = r->chain_base; std::vector<int> ph; bool plan_ok = true; int32_t rc;
  // Phases: 0 uniform ids | 1 search + gather / gather | 2 forward (both nets) | 3 td tile (+ update_priorities!) | 4 pullback || leaf re-sum -> root paths (LeafTouchOp) |
  // 5 norm || Adam (AdamSelfOp) | 6 info, beta-power advance. Chained: the sampling of epoch e + 1 runs beside the tail of epoch e -- ids beside 4, search + gather beside 5 (the
  // replay tree is complete after 4; the gather rewrites the batch rows the pullback read: not before 5), forward beside 6 (after Adam): FOUR launches per epoch.
  const bool touch_split = per && (B > 256 || source->per_full_dirty);      // (root paths as an op of their own in 5: the search then waits one launch longer)
  const int ov = touch_split ? 2 : 3;
  auto bail = [&](int32_t e) { crux_exec_abort(c); return e; };
  size_t m = exec_mark(c); const size_t ops0 = m; r->epoch_marks.push_back(ops0);
  auto sect = [&](auto&& rule) { for (size_t i = m; i < r->ops.size(); ++i) { int p = rule(r->ops[i].kid); if (p < 0) { plan_ok = false; p = 0; }
      ph.push_back(ph_tag(base > 0 ? base + p - ov : p, 0)); } m = r->ops.size(); };
  auto only = [&](int p) { sect([p](int) { return p; }); };
  if (use_weight && !has_col(batch, CRUX_COL_WEIGHT)) return bail(crux_fail(c, CRUX_EINVAL, "td_loss(weight=:weight): batch has no :weight column"));
  rc = per ? crux_per_sample(batch, source, B, nullptr, beta, sample_counter) : crux_uniform_sample(batch, source, B, nullptr, sample_counter); if (rc) return bail(rc);
  sect([](int kid) { return kid == OP_UNIFORM_IDS ? 0 : (kid == OP_PER_SAMPLE || kid == OP_GATHER_RING_ALL || kid == OP_RING_IDS || kid == OP_COPY_F32) ? 1 : kid == OP_PER_UPDATE ? 2 : -1; });
  Carve cv{(char*)crux_scratch(c, 4 * (size_t)B * (nout + 2) + 8192), 0}; if (!cv.p) return bail(crux_fail(c, CRUX_ENOMEM, "dqn_epoch: scratch"));
  float* dy = cv.take<float>((size_t)B * nout); float* term = cv.take<float>((size_t)B); float* qsel = cv.take<float>((size_t)B);
  Carve sv{(char*)crux_exec_small(c, 256 * 6), 0}; if (!sv.p) return bail(crux_fail(c, CRUX_ENOMEM, "dqn_epoch: executor region"));
  float* dinfo = sv.take<float>(CRUX_INFO_N); double* ssq = sv.take<double>(2 + SUMSQ_BLOCKS); int32_t* status = sv.take<int32_t>(1); int32_t* nanf = sv.take<int32_t>(2);
  rc = crux_exec_zero(c, dinfo, 256 * 6, c->stream); if (rc) return bail(rc);
  float* const info_dst = r->info_row_override ? r->info_row_override : dinfo;      // asynchronous chains: the info op writes the caller's device row itself (round 6: one launch less behind the chain's last epoch)
  if (r->info_row_override) { rc = crux_exec_zero(c, r->info_row_override, sizeof(float) * CRUX_INFO_N, c->stream); if (rc) return bail(rc); }
  only(1);
  const float* S = (const float*)batch->col[CRUX_COL_S]; const float* SP = (const float*)batch->col[CRUX_COL_SP];
  rc = crux_dense_forward12(net, S, B, c->stream); if (!rc) rc = crux_dense_forward12(tnet, SP, B, c->stream); if (rc) return bail(rc);
  only(2);
  { DqnTdArgs a{}; auto l3 = [&](crux_mlp* n) { const NetDesc& nd = n->nd; return TileSet{n->p + nd.woff[2], n->p + nd.boff[2], crux_dense_act(n, 2), crux_dense_act(n, 3)}; };
    a.qt = l3(tnet); a.q = l3(net); a.r = (const float*)batch->col[CRUX_COL_R]; a.done = (const uint8_t*)batch->col[CRUX_COL_DONE]; a.a = (const uint8_t*)batch->col[CRUX_COL_A];
    a.w = use_weight ? (const float*)batch->col[CRUX_COL_WEIGHT] : nullptr; a.gamma = gamma; a.softq_alpha = softq_alpha; a.nout = nout; a.K = K; a.B = (int32_t)B;
    a.y = d_y; a.dy = dy; a.err = per ? d_err : nullptr; a.term = term; a.qsel = qsel;
    a.per = per ? 1 : 0; a.pr = source->priorities; a.pminmax = source->pminmax; a.ids = batch->d_indices; a.per_alpha = source->alpha;
    crux_exec_push<DqnTdTileOp, OP_DQN_TD_TILE>(c, (unsigned)((B + 15) / 16), a); }
  only(3);
  Sumsq2Fix fx{};
  rc = crux_dense_backward(net, S, B, dy, 1.0f, true, nullptr, c->stream, &fx, 0, nanf); if (rc) return bail(rc);
  only(4);
  if (per) { rc = crux_per_touched(source, batch->d_indices, B, false, (unsigned*)(nanf + 2)); if (rc) return bail(rc);      // leaves + root paths as one op beside the pullback (LeafTouchOp)
    sect([](int kid) { return (kid == OP_LEAF_TOUCH || kid == OP_LEAF_REFRESH) ? 4 : kid == OP_TREE_TOUCH ? 5 : -1; }); }
  // 5: the norm (for the info row) and, beside it, Adam gated on the producers' NaN flags (AdamSelfOp, sac.hip) | 6: info, beta-power advance
  CRUX_RUN(c, Sumsq2Op, OP_SUMSQ2, k_sumsq2, SUMSQ_BLOCKS, 256, c->stream, net->g, (int64_t)net->nd.n_params, (float*)nullptr, (int64_t)0, ssq, fx);
  rc = adam_self(net, nanf, status, fx, 0); if (rc) return bail(rc);
  sect([](int kid) { return kid == OP_ADAM_ADVANCE_SELF ? 6 : 5; });
  crux_exec_push<TdInfo2Op, OP_TD_INFO2>(c, 1u, (const float*)term, (const float*)qsel, (const double*)ssq, B, info_dst);
  only(6);
  crux_exec_add_readback(c, info_out, dinfo, status, "td_loss");
  if (!(plan_ok && ph.size() == r->ops.size() - ops0)) r->chain_ok = false;
  r->chain_tags.insert(r->chain_tags.end(), ph.begin(), ph.end()); r->chain_base += 7 - (r->chain_base > 0 ? ov : 0);
  return CRUX_OK;
}

// One epoch of value_training for the DQN family (src/model_free/off_policy.jl:69-93 with dqn_target, rl/dqn.jl:4-6): rand!(batch, source; i) ->
// y = target(pi_minus, batch) -> [td_error -> update_priorities!(source, batch.indices, .)] -> train!(pi, td_loss). Networks at least
// CRUX_DENSE_MIN_WIDTH wide run the whole epoch as ONE fused launch; narrower ones take the same steps one call at a time.
static int32_t dqn_epoch_impl(crux_mlp* net, crux_mlp* target_net, crux_buffer* source, crux_buffer* batch, float gamma, float softq_alpha, int32_t use_weight, float beta,
                       uint64_t sample_counter, float* info_out) {
  if (!net || !target_net || !source || !batch) return CRUX_EINVAL;
  crux_ctx* c = net->ctx; const int64_t B = batch->capacity;
  const bool per = source->prioritized;
  const bool fuse = net->nd.maxdim >= CRUX_DENSE_MIN_WIDTH && target_net->nd.maxdim >= CRUX_DENSE_MIN_WIDTH && !crux_sw().no_fused_epoch;
  int32_t rc;
  if (per) { rc = crux_per_prepare(source); if (rc) return rc; }
  float* d_y = nullptr; float* d_err = nullptr;
  { char* sc2 = (char*)c->epoch_tmp;       // targets and td errors: a block of the context that no piece carves
    if (c->epoch_tmp_bytes < 8 * (size_t)B + 512) { if (sc2) { HIPCHK(c, hipStreamSynchronize(c->stream)); (void)hipFree(sc2); }
      c->epoch_tmp_bytes = 16 * (size_t)B + 4096; if (hipMalloc(&c->epoch_tmp, c->epoch_tmp_bytes) != hipSuccess) { c->epoch_tmp = nullptr; c->epoch_tmp_bytes = 0; return crux_fail(c, CRUX_ENOMEM, "dqn_epoch: targets"); } sc2 = (char*)c->epoch_tmp; }
    d_y = (float*)sc2; d_err = (float*)(sc2 + ((4 * (size_t)B + 255) / 256) * 256); }
  if (fuse && c->rec && rec_of(c)->chain && crux_exec_recording(c) && dqn_tile_case(net, target_net, source, batch))      // chained epochs of the C3 family: the tile op (5 launches per epoch)
    return dqn_epoch_tiles(net, target_net, source, batch, gamma, softq_alpha, use_weight, beta, sample_counter, info_out, d_y, d_err);
  constexpr int eager_mask = 0;
  auto piece = [&](int bit) -> int32_t { if (!fuse) return CRUX_OK;
    if (eager_mask & bit) { if (crux_exec_recording(c)) return crux_exec_run(c); return CRUX_OK; }
    if (!crux_exec_recording(c)) return crux_exec_begin(c); return CRUX_OK; };
  auto bail = [&](int32_t e) { if (fuse) crux_exec_abort(c); return e; };
  // Phase plan of the fused epoch (L = number of Dense layers): the target network's forward chain runs beside the online network's, the replay
  // bookkeeping beside the backward chain.    0 search | 1 gather, ring ids, zero-fill | 2+k forward layer k (both nets; + push! priorities of the batch)
  //   2+L dqn_target | 3+L td head | 4+L+j backward of layer L-1-j (weight + data gradient; + update_priorities!, leaf re-sum, root paths) | then norm, info, Adam
  std::vector<int> ph; bool plan_ok = true; const int Ld = net->nd.L;
  // chained epochs (crux_dqn_epochs): the sampling of epoch e + 1 (phase 0: search / uniform ids, phase 1: gather, ring ids, fills) touches nothing that the last three
  // phases of epoch e (norm | info + Adam | beta-power advance) read or write -- the batch rows and the sampled ids were last read by the first-layer weight gradient and
  // the tree refresh one phase earlier -- so it runs BESIDE them: phases 0 and 1 of a later epoch are tagged as the previous epoch's last-but-two and last-but-one, and
  // its remaining phases close up by THREE: the first forward layer of epoch e + 1 shares a launch with the beta-power advance of epoch e (the last phase, which writes
  // the two powers only; Adam of epoch e + 1 reads them nine phases later). The replay chain (leaf refresh -> root paths -> search -> gather) then hides the optimizer
  // tail instead of following it: 13 phases, 10 launches per chained epoch.
  // The overlap is bounded by the replay tree: update_priorities! -> leaf re-sum -> root paths of epoch e sit at phases 4+L-sq .. 6+L-sq and read batch->d_indices / write
  // the tree total, which the search of epoch e + 1 (phase 4+2L-sq with the full overlap of three) rewrites and probes. With fewer than three Dense layers the backward
  // chain is too short to cover them, so the overlap shrinks to L phases there (search(e + 1) strictly after the root paths of e).
  // Round 4 (dense_fused.h): a network whose first two layers run as ONE forward launch (Fwd12Op) has nf = Ld - 1 forward launches, and one whose layer-1 / layer-0
  // pullback runs as one phase (Wgrad2Op || Dgrad2W1Op) nb = Ld - 1 backward phases; the plan below is written in nf, nb. With the sequential group (sq) the replay
  // chain starts INSIDE the head's block: update_priorities! is a second sequential tail behind the td head (one block, reads the head's td errors), so the leaf
  // re-sum and the root paths sit at 3 + nf and 4 + nf, and the search of epoch e + 1 (phase 7 + nf + nb - sq - ov of this epoch) stays behind the root paths as long
  // as ov <= nb + 1; without the group the chain is update | leaf | paths at 4 + nf - sq .., and ov <= nb as before.
  const NetPlan pn = net_plan(net, B); const bool ffw = crux_dense_fwd_fused(net);
  const int nf = pn.nf, nb = pn.nb;
  const int sq0 = (B <= 256 && !exec_persistent_on(c)) ? 1 : 0;
  const bool tailp = per && sq0 == 1;
  // sph: with the search and the gather in one launch (PerSampleGatherOp) the sampling of an epoch is phase 1 alone, so the search of epoch e + 1 sits one phase later and the
  // overlap may be one deeper
  const int sph = (per && crux_per_fused_gather() && !c->per_split_sample) ? 1 : 0;
  const int ovmax = (tailp ? nb + 1 : nb) + sph;
  const int ov = per ? (ovmax < 1 ? 1 : (ovmax < 3 ? ovmax : 3)) : 3;
  auto tag = [&](size_t from, auto&& rule) { if (!fuse || !crux_exec_recording(c)) return; ExecRec* r = rec_of(c); int g = 0; const int base = r->chain ? r->chain_base : 0;
    for (size_t i = from; i < r->ops.size(); ++i) { int p = rule(r->ops[i].kid, g); if (p < 0) { plan_ok = false; p = 0; }
      const int sub = p >> 12; p &= 4095;
      ph.push_back(ph_tag(base > 0 ? (p < 2 ? base - ov + p : base + p - ov) : p, sub)); } };
  // the target (one block at B <= 256) and the loss head (one block) are a sequential pair: the head runs in the target's block, right after it, and every later
  // phase moves up by one (sq). Not in the persistent one-XCD form, whose workgroups walk the ops of a phase in lockstep.
  const int sq = sq0;
  if (crux_dense_fwd_fused(target_net) != ffw) plan_ok = false;
  rc = piece(1); if (rc) return bail(rc);
  const size_t ops0 = fuse && crux_exec_recording(c) ? exec_mark(c) : 0;      // first op of THIS epoch (a chained recording already holds the earlier epochs)
  if (fuse && crux_exec_recording(c) && rec_of(c)->chain) rec_of(c)->epoch_marks.push_back(ops0);
  size_t m = ops0;
  rc = per ? crux_per_sample(batch, source, B, nullptr, beta, sample_counter) : crux_uniform_sample(batch, source, B, nullptr, sample_counter); if (rc) return bail(rc);
  tag(m, [&](int kid, int&) { return (kid == OP_PER_SEARCH || kid == OP_UNIFORM_IDS) ? 0 : (kid == OP_PER_SAMPLE || kid == OP_GATHER_RING_ALL || kid == OP_RING_IDS || kid == OP_COPY_F32) ? 1 : kid == OP_PER_UPDATE ? 2 : -1; });
  rc = piece(2); if (rc) return bail(rc);
  m = fuse && crux_exec_recording(c) ? exec_mark(c) : 0;
  rc = softq_alpha > 0.f ? crux_softq_target(target_net, batch, gamma, softq_alpha, d_y) : crux_dqn_target(target_net, batch, gamma, d_y); if (rc) return bail(rc);      // softq_target(alpha) (rl/softq.jl:4-13) | dqn_target (rl/dqn.jl:4-6)
  tag(m, [&](int kid, int& g) { if (kid == OP_FWD12) { g = 1; return 2; }
    return kid == OP_GEMM ? (g < nf ? 2 + g++ : -1) : (kid == OP_DQN_TARGET || kid == OP_SOFTQ_TARGET) ? (2 + nf) | (sq ? PH_SEQ_HEAD : 0) : -1; });
  rc = piece(4); if (rc) return bail(rc);
  m = fuse && crux_exec_recording(c) ? exec_mark(c) : 0;
  auto td_rule = [&](int kid, int& g) {      // g counts the GEMMs: Ld forward, then (weight, data) pairs from the last layer down, the first layer has no data gradient
    if (kid == OP_FILL) return 1;
    if (kid == OP_FWD12) { g = 1; return 2; }
    if (kid == OP_GEMM) { const int k = g++; if (k < nf) return 2 + k; const int j = pn.stage(k - nf); return j < nb ? 4 + nf + j - sq : -1; }
    if (kid == OP_WGRAD2 || kid == OP_DGRAD2W1) return 3 + nf + nb - sq;      // the last backward phase: layer 1's dW beside layer 1's dX -> layer 0's dW
    if (kid == OP_TD_HEAD) return sq ? ((2 + nf) | PH_SEQ_TAIL) : 3 + nf;
    if (kid == OP_SUMSQ2) return 4 + nf + nb - sq; if (kid == OP_TD_INFO || kid == OP_ADAM_GATED) return 5 + nf + nb - sq; if (kid == OP_ADAM_ADVANCE) return 6 + nf + nb - sq;
    return -1; };
  if (per) { rc = crux_td_step_with_error(net, batch, d_y, use_weight, d_err, info_out); if (rc) return bail(rc);
    tag(m, td_rule);
    rc = piece(8); if (rc) return bail(rc);
    m = fuse && crux_exec_recording(c) ? exec_mark(c) : 0;
    rc = crux_per_update_device(source, batch->d_indices, d_err, B); if (rc) return bail(rc);
    tag(m, [&](int kid, int&) { if (tailp) return kid == OP_PER_UPDATE ? ((2 + nf) | PH_SEQ_TAIL) : kid == OP_LEAF_REFRESH ? 3 + nf : kid == OP_TREE_TOUCH ? 4 + nf : -1;
      return kid == OP_PER_UPDATE ? 4 + nf - sq : kid == OP_LEAF_REFRESH ? 5 + nf - sq : kid == OP_TREE_TOUCH ? 6 + nf - sq : -1; }); }
  else { rc = crux_td_step(net, batch, d_y, use_weight, info_out); if (rc) return bail(rc); tag(m, td_rule); }
  if (fuse && crux_exec_recording(c) && rec_of(c)->chain) {     // chained: the caller (crux_dqn_epochs) schedules and runs the whole list
    ExecRec* r = rec_of(c);
    if (!(plan_ok && !eager_mask && target_net->nd.L == Ld && ph.size() == r->ops.size() - ops0)) r->chain_ok = false;
    r->chain_tags.insert(r->chain_tags.end(), ph.begin(), ph.end()); r->chain_base += (r->chain_base > 0 ? 7 - ov : 7) + nf + nb - sq;
    return CRUX_OK;
  }
  if (fuse && crux_exec_recording(c) && plan_ok && !eager_mask && target_net->nd.L == Ld && ph.size() == rec_of(c)->ops.size()) { rc = exec_schedule(c, ph); if (rc) return bail(rc); }
  return (fuse && crux_exec_recording(c)) ? crux_exec_run(c) : CRUX_OK;
}

int32_t crux_dqn_epoch(crux_mlp* net, crux_mlp* target_net, crux_buffer* source, crux_buffer* batch, float gamma, int32_t use_weight, float beta,
                       uint64_t sample_counter, float* info_out) { return dqn_epoch_impl(net, target_net, source, batch, gamma, 0.f, use_weight, beta, sample_counter, info_out); }

// value_training's epoch loop (off_policy.jl:69: `for epoch in 1:c_opt.epochs`) for the DQN family as ONE recorded list: the n epochs are recorded back to back, the
// phases of epoch e follow those of epoch e - 1, and the host uploads, launches and reads back once. Epoch e draws with sample counter sample_counter0 + e; beta is
// the iteration's (rand!(…, i = S.i)). infos: host [n x CRUX_INFO_N]. Falls back to n single-epoch calls wherever an epoch cannot be chained (narrow networks, a
// priority tree that needs a full rebuild between epochs).
static int32_t dqn_epochs_impl(crux_mlp* net, crux_mlp* target_net, crux_buffer* source, crux_buffer* batch, float gamma, float softq_alpha, int32_t use_weight, float beta,
                               uint64_t sample_counter0, int32_t n_epochs, float* infos, float* d_infos_async = nullptr, float polyak_tau = -1.f) {
  if (!net || !target_net || !source || !batch || n_epochs < 1) return CRUX_EINVAL;
  crux_ctx* c = net->ctx;
  const bool fuse = net->nd.maxdim >= CRUX_DENSE_MIN_WIDTH && target_net->nd.maxdim >= CRUX_DENSE_MIN_WIDTH && !crux_sw().no_fused_epoch &&
                    !crux_sw().no_chained_epochs;
  // 256-wide networks of the shapes dqn_persist.h instantiates: the whole chain as two persistent kernels. OPT-IN (CRUX_DQN_PERSIST=1): correct (tests/test_gpu_round3.py)
  // but measured SLOWER than the phase launches in round 3 -- 113-125 us against 73-79 us per C3 epoch (DESIGN 4.3 has the in-kernel timeline) -- so the default stays
  // with the phases.
  const int64_t Bc = batch->capacity;
  const bool persist = fuse && softq_alpha == 0.f && net->nd.L == 3 && target_net->nd.L == 3 && net->nd.dims[1] == 256 && net->nd.dims[2] == 256 && target_net->nd.dims[1] == 256 && target_net->nd.dims[2] == 256 &&
                       net->nd.dims[0] == target_net->nd.dims[0] && net->nd.dims[3] == target_net->nd.dims[3] && dqp_shape(net->nd.dims[0], net->nd.dims[3], Bc) &&
                       net->nd.acts[0] == CRUX_ACT_RELU && net->nd.acts[1] == CRUX_ACT_RELU && net->nd.acts[2] == CRUX_ACT_IDENTITY &&
                       target_net->nd.acts[0] == CRUX_ACT_RELU && target_net->nd.acts[1] == CRUX_ACT_RELU && target_net->nd.acts[2] == CRUX_ACT_IDENTITY &&
                       net->has_adam && batch->obs_dim == net->nd.dims[0] && batch->act_kind == CRUX_ACTION_DISCRETE && batch->act_dim == net->nd.dims[3] && (!use_weight || has_col(batch, CRUX_COL_WEIGHT)) &&
                       crux_sw().dqn_persist && !c->dqp_broken && crux_x2_placement_ok_c(c);
  auto flush = [&]() -> int32_t {
    if (!crux_exec_recording(c)) return CRUX_OK;
    ExecRec* r = rec_of(c); r->chain = false;
    if (persist) {
      ExecRec::Dqp& d = r->dqp; d.in = net->nd.dims[0]; d.out = net->nd.dims[3]; d.bt = (int)(Bc / 16); d.n_epochs = (int)r->epoch_marks.size(); d.net = net; d.tnet = target_net; d.batch = batch;
      d.gamma = gamma; d.use_weight = use_weight != 0; d.d_err = source->prioritized ? (float*)((char*)c->epoch_tmp + ((4 * (size_t)Bc + 255) / 256) * 256) : nullptr;
      int32_t rp = dqp_build(r);
      if (!rp) rp = crux_ensure_aux_stream(c);
      if (!rp && !r->dqp_buf && hipMalloc(&r->dqp_buf, DQP_BUF_BYTES) != hipSuccess) { r->dqp_buf = nullptr; rp = CRUX_ENOMEM; }
      if (!rp) { d.on = true; const int32_t rr = crux_exec_run(c);
        if (rr == CRUX_EHIP) c->dqp_broken = true;      // a wait timed out (the two kernels did not run side by side): later chains take the phase launches
        return rr; }
    }
    if (r->chain_ok && r->chain_tags.size() == r->ops.size()) { const int32_t rs = exec_schedule(c, r->chain_tags); if (rs) { crux_exec_abort(c); return rs; } }
    r->async = d_infos_async != nullptr;
    return crux_exec_run(c);
  };
  int32_t rc = CRUX_OK; int in_chain = 0;
  struct SplitGuard { crux_ctx* c; bool old; ~SplitGuard() { c->per_split_sample = old; } } split_guard{c, c->per_split_sample};
  if (persist) c->per_split_sample = true;
  for (int e = 0; e < n_epochs; ++e) {
    float* info_e = infos ? infos + (size_t)e * CRUX_INFO_N : nullptr;
    if (!fuse) { if (d_infos_async) return CRUX_EUNSUP;      // narrow networks run call by call with a read-back per epoch: the caller takes the synchronous entry point
      rc = dqn_epoch_impl(net, target_net, source, batch, gamma, softq_alpha, use_weight, beta, sample_counter0 + (uint64_t)e, info_e); if (rc) return rc; continue; }
    // a priority tree that needs a plain rebuild (the ring is still filling, or a bulk change) cannot be refreshed inside a recording: run what is recorded first
    if (in_chain && ((source->prioritized && source->per_full_dirty) || in_chain >= 8)) { rc = flush(); in_chain = 0; if (rc) return rc; }
    if (!in_chain) { if (source->prioritized) { rc = crux_per_prepare(source); if (rc) return rc; }
      rc = crux_exec_begin(c); if (rc) return rc; }
    rec_of(c)->chain = true;
    const bool tiles = dqn_tile_case(net, target_net, source, batch) && !crux_sw().no_fused_epoch;
    rec_of(c)->info_row_override = (d_infos_async && tiles) ? d_infos_async + (size_t)e * CRUX_INFO_N : nullptr;      // tile plan: the info op writes the caller's row itself
    rc = dqn_epoch_impl(net, target_net, source, batch, gamma, softq_alpha, use_weight, beta, sample_counter0 + (uint64_t)e, info_e);
    if (c->rec) rec_of(c)->info_row_override = nullptr;
    if (rc) { if (c->rec) rec_of(c)->chain = false; crux_exec_abort(c); return rc; }
    if (d_infos_async && !tiles) {      // the epoch's info row goes to the caller's device array, copied in the epoch's last phase (one phase after the info op wrote it)
      ExecRec* r = rec_of(c);
      crux_exec_push<CopyF32Op, OP_COPY_F32>(c, 1u, d_infos_async + (size_t)e * CRUX_INFO_N, (const float*)r->readbacks.back().d_info, (int64_t)CRUX_INFO_N);
      int tmax = 0; for (size_t k = r->epoch_marks.empty() ? 0 : r->epoch_marks.back(); k < r->chain_tags.size(); ++k) tmax = std::max(tmax, r->chain_tags[k] & ~3);      // the epoch's last phase (the beta-power advance)
      r->chain_tags.push_back(tmax + (tiles ? 4 : 0)); }      // (tile plan: the info op sits IN the epoch's last phase -- the copy joins the launch after it)
    ++in_chain;
  }
  // polyak_average!(pi_minus, pi, tau) after the epoch loop (off_policy.jl:108: the DQN family updates its target once per value_training call) as an op of the chain: it needs
  // the last epoch's Adam (phase 5) and shares the launch of that epoch's info / beta-power phase -- the stand-alone k_polyak launch (~5 us of a 230 us C3 iteration) is gone.
  if (fuse && polyak_tau >= 0.f && in_chain > 0) {
    ExecRec* r = rec_of(c); const size_t n0 = r->ops.size();
    rc = crux_polyak(target_net, net, polyak_tau); if (rc) { r->chain = false; crux_exec_abort(c); return rc; }
    int tmax = 0; for (size_t k = r->epoch_marks.empty() ? 0 : r->epoch_marks.back(); k < r->chain_tags.size(); ++k) tmax = std::max(tmax, r->chain_tags[k] & ~3);
    for (size_t k = n0; k < r->ops.size(); ++k) r->chain_tags.push_back(tmax);
  }
  return fuse ? flush() : rc;
}
int32_t crux_dqn_epochs(crux_mlp* net, crux_mlp* target_net, crux_buffer* source, crux_buffer* batch, float gamma, int32_t use_weight, float beta,
                        uint64_t sample_counter0, int32_t n_epochs, float* infos) {
  return dqn_epochs_impl(net, target_net, source, batch, gamma, 0.f, use_weight, beta, sample_counter0, n_epochs, infos);
}
// the same loop with softq_target(alpha) (rl/softq.jl:4-13) in place of dqn_target: SoftQ's value_training
// The same chain WITHOUT the host: nothing is read back and nothing is waited for; the info row of epoch e lands in d_infos[e] (device, [n_epochs x CRUX_INFO_N]) when
// the device gets there. For the iteration loop of solve(::OffPolicySolver) (off_policy.jl:133-147): the host records and enqueues the next iteration's steps! and
// value_training while the device runs this one. A NaN gradient norm shows as a NaN in the info row (the update is skipped on the device as always, training.jl:20);
// CRUX_EUNSUP for networks that do not take the recorded form (the caller uses crux_dqn_epochs).
int32_t crux_dqn_epochs_async(crux_mlp* net, crux_mlp* target_net, crux_buffer* source, crux_buffer* batch, float gamma, int32_t use_weight, float beta,
                              uint64_t sample_counter0, int32_t n_epochs, float* d_infos) {
  if (!d_infos) return CRUX_EINVAL;
  if (!net || !target_net || net->nd.maxdim < CRUX_DENSE_MIN_WIDTH || target_net->nd.maxdim < CRUX_DENSE_MIN_WIDTH || crux_sw().no_fused_epoch || crux_sw().no_chained_epochs) return CRUX_EUNSUP;
  return dqn_epochs_impl(net, target_net, source, batch, gamma, 0.f, use_weight, beta, sample_counter0, n_epochs, nullptr, d_infos);
}
// value_training of the DQN family INCLUDING its target update (off_policy.jl:66-111 with :108), without the host: the chain of crux_dqn_epochs_async (softq_alpha > 0:
// crux_softq_epochs_async) with polyak_average!(target_net, net, tau) riding in the last epoch's final phase. tau < 0: no target update (== the plain async entries).
int32_t crux_dqn_value_training_async(crux_mlp* net, crux_mlp* target_net, crux_buffer* source, crux_buffer* batch, float gamma, float softq_alpha, int32_t use_weight, float beta,
                                      uint64_t sample_counter0, int32_t n_epochs, float tau, float* d_infos) {
  if (!d_infos || softq_alpha < 0.f || tau > 1.f) return CRUX_EINVAL;
  if (!net || !target_net || net->nd.maxdim < CRUX_DENSE_MIN_WIDTH || target_net->nd.maxdim < CRUX_DENSE_MIN_WIDTH || crux_sw().no_fused_epoch || crux_sw().no_chained_epochs) return CRUX_EUNSUP;
  if (tau >= 0.f && net->nd.n_params != target_net->nd.n_params) return crux_fail(net->ctx, CRUX_EINVAL, "value_training: polyak_average! needs equal parameter counts");
  return dqn_epochs_impl(net, target_net, source, batch, gamma, softq_alpha, use_weight, beta, sample_counter0, n_epochs, nullptr, d_infos, tau);
}
int32_t crux_softq_epochs_async(crux_mlp* net, crux_mlp* target_net, crux_buffer* source, crux_buffer* batch, float gamma, float alpha, int32_t use_weight, float beta,
                                uint64_t sample_counter0, int32_t n_epochs, float* d_infos) {      // crux_softq_epochs without the host in the loop (see crux_dqn_epochs_async)
  if (!d_infos || !(alpha > 0.f)) return CRUX_EINVAL;
  if (!net || !target_net || net->nd.maxdim < CRUX_DENSE_MIN_WIDTH || target_net->nd.maxdim < CRUX_DENSE_MIN_WIDTH || crux_sw().no_fused_epoch || crux_sw().no_chained_epochs) return CRUX_EUNSUP;
  return dqn_epochs_impl(net, target_net, source, batch, gamma, alpha, use_weight, beta, sample_counter0, n_epochs, nullptr, d_infos);
}
int32_t crux_softq_epochs(crux_mlp* net, crux_mlp* target_net, crux_buffer* source, crux_buffer* batch, float gamma, float alpha, int32_t use_weight, float beta,
                          uint64_t sample_counter0, int32_t n_epochs, float* infos) {
  if (!(alpha > 0.f)) return CRUX_EINVAL;
  return dqn_epochs_impl(net, target_net, source, batch, gamma, alpha, use_weight, beta, sample_counter0, n_epochs, infos);
}

int32_t crux_sac_target(crux_mlp* actor, crux_mlp* q1t, crux_mlp* q2t, crux_mlp* la, crux_buffer* b, float gamma, uint64_t seed, uint64_t counter, float* d_y);
int32_t crux_sac_temp_step(crux_mlp* actor, crux_mlp* la, crux_buffer* b, float H_target, uint64_t seed, uint64_t counter, float* info_out);
int32_t crux_double_q_step(crux_mlp* q1, crux_mlp* q2, crux_buffer* b, const float* d_y, int32_t use_weight, float* info_out);
int32_t crux_sac_actor_step(crux_mlp* actor, crux_mlp* q1, crux_mlp* q2, crux_mlp* la, crux_buffer* b, uint64_t seed, uint64_t counter, float* info_out);
int32_t crux_q_step(crux_mlp* q, crux_buffer* b, const float* d_y, int32_t use_weight, float* info_out);
int32_t crux_dpg_target(crux_mlp* actor_t, crux_mlp* q1t, crux_mlp* q2t, crux_buffer* b, float gamma, float sigma, float eps_min, float eps_max, float a_min, float a_max,
                        uint64_t seed, uint64_t counter, float* d_y);
int32_t crux_dpg_actor_step(crux_mlp* actor, crux_mlp* q, crux_buffer* b, float* info_out);

// The SAC epoch on the fused block kernels and the per-sample-tile ops of sac_fused.h (round 4): 17 phases, 14 launches per epoch inside a chain (round 3: 30 / 26).
//   0 ids | 1 gather, zero-fills | 2 actor(s') layers 0+1 ; vcat(s, a) | 3 actor(s') output layer + exploration -> (s', a'), log pi ; Q1, Q2 (s, a) layers 0+1
//   4 target Q1, Q2 (s', a') layers 0+1 ; actor(s) layers 0+1 | 5 the four critic output layers + sac_target + both double_Q_loss heads ; actor(s) output layer + the TWO
//   exploration draws of the temperature and the actor step (same means: the temperature step leaves the actor untouched) | 6 critic pullbacks (one phase each network) ;
//   temperature head | 7 norm ; Adam(log alpha) | 8 info, Adam(Q1), Adam(Q2) | 9 Q1, Q2 (s, a~) layers 0+1 | 10 their output layers + sac_actor_loss head
//   11 critic input gradients down to layer 0's dZ | 12 layer 0's W' dZ + reverse of exploration | 13 actor pullback ; logSigma row sums | 14 norm | 15 info, Adam | 16 advance, polyak
// The order inside every chain is the reference's (temperature, critic and actor each see what the previous step left; sac_target reads log alpha BEFORE the temperature
// update lands, the actor head after). Same arithmetic as the generic recording below (tools/fused_check.py compares the two bit for bit); false = not this case.
static bool sac_tile_case(crux_mlp* actor, crux_mlp* q1, crux_mlp* q2, crux_mlp* q1t, crux_mlp* q2t, crux_buffer* source, crux_buffer* batch, int32_t uc, int32_t ua) {
  const bool on = crux_sw().sac_tile_ops && !exec_persistent_on(actor->ctx) && !crux_sw().no_fused_epoch;      // (read per call: tests switch forms inside one process)
  if (!on || !uc || !ua || source->prioritized) return false;
  const int64_t B = batch->capacity; crux_mlp* all[5] = {actor, q1, q2, q1t, q2t};
  for (crux_mlp* n : all) { if (n->nd.L != 3 || !crux_dense_fwd_fused(n) || n->nd.acts[2] != CRUX_ACT_IDENTITY || n->nd.dims[2] != actor->nd.dims[2]) return false; }
  if (!crux_dense_bwd_fused3(actor, B) || !crux_dense_bwd_fused3(q1, B) || !crux_dense_bwd_fused3(q2, B)) return false;
  if (batch->act_dim > 4 || batch->obs_dim + batch->act_dim > 16 || actor->nd.dims[3] != batch->act_dim || q1->nd.dims[3] != 1 || q2->nd.dims[3] != 1) return false;
  if (q1t->nd.dims[3] != 1 || q2t->nd.dims[3] != 1 || q1->nd.dims[0] != batch->obs_dim + batch->act_dim) return false;
  return true;
}
static int32_t sac_epoch_tiles(crux_mlp* actor, crux_mlp* q1, crux_mlp* q2, crux_mlp* actor_targ, crux_mlp* q1t, crux_mlp* q2t, crux_mlp* la, crux_buffer* source, crux_buffer* batch,
                               float gamma, float H_target, float tau, int32_t use_weight, uint64_t sample_counter, uint64_t noise_seed, uint64_t noise_counter0,
                               float* info_temp, float* info_critic, float* info_actor) {
  crux_ctx* c = actor->ctx; const int64_t B = batch->capacity; const int od = batch->obs_dim, ad = batch->act_dim, sd = od + ad, K = actor->nd.dims[2];
  int32_t rc = CRUX_OK;
  if (use_weight && !has_col(batch, CRUX_COL_WEIGHT)) return crux_fail(c, CRUX_EINVAL, "double_Q_loss(weight=:weight): batch has no :weight column");
  if (!crux_exec_recording(c)) { rc = crux_exec_begin(c); if (rc) return rc; }
  auto bail = [&](int32_t e) { crux_exec_abort(c); return e; };
  ExecRec* r = rec_of(c); const int base = r->chain ? r->chain_base : 0; std::vector<int> ph; bool plan_ok = true;
  size_t m = exec_mark(c); const size_t ops0 = m;
  auto sect = [&](auto&& rule) { for (size_t i = m; i < r->ops.size(); ++i) { int p = rule(r->ops[i].kid); if (p < 0) { plan_ok = false; p = 0; }
      ph.push_back(ph_tag(base > 0 ? base + p - 3 : p, 0)); } m = r->ops.size(); };
  auto only = [&](int p) { sect([p](int) { return p; }); };
  // buffers of this epoch (scratch of the recording: live until the list has run)
  float* y = (float*)crux_exec_small(c, 4 * (size_t)B); if (!y) return bail(crux_fail(c, CRUX_EUNSUP, "sac_epoch: batch of %lld rows exceeds the executor's region", (long long)B));
  Carve cv{(char*)crux_scratch(c, 4 * (size_t)B * (3 * sd + 3 * ad + 12) + 65536), 0}; if (!cv.p) return bail(crux_fail(c, CRUX_ENOMEM, "sac_epoch: scratch"));
  float* sa_c = cv.take<float>((size_t)B * sd); float* sa_t = cv.take<float>((size_t)B * sd); float* sa_a = cv.take<float>((size_t)B * sd);
  float* lp_t = cv.take<float>((size_t)B); float* lp_temp = cv.take<float>((size_t)B); float* lp_a = cv.take<float>((size_t)B); float* eps = cv.take<float>((size_t)B * ad);
  float* dy1 = cv.take<float>((size_t)B); float* dy2 = cv.take<float>((size_t)B); float* t1 = cv.take<float>((size_t)B); float* t2 = cv.take<float>((size_t)B);
  float* da1 = cv.take<float>((size_t)B); float* da2 = cv.take<float>((size_t)B); float* ta = cv.take<float>((size_t)B); float* dmu = cv.take<float>((size_t)B * ad); float* dls = cv.take<float>((size_t)B * ad);
  // info rows / statistics / status words of the three steps
  Carve st_{(char*)crux_exec_small(c, 256 * 5), 0}, sc_{(char*)crux_exec_small(c, 256 * 6), 0}, sa_{(char*)crux_exec_small(c, 256 * 6), 0};
  if (!st_.p || !sc_.p || !sa_.p) return bail(crux_fail(c, CRUX_ENOMEM, "sac_epoch: executor region"));
  float* it = st_.take<float>(CRUX_INFO_N); double* ssq_t = st_.take<double>(2 + SUMSQ_BLOCKS); int32_t* stt = st_.take<int32_t>(1);
  float* ic = sc_.take<float>(CRUX_INFO_N); double* ssq_c = sc_.take<double>(2 + SUMSQ_BLOCKS); int32_t* stc = sc_.take<int32_t>(1); int32_t* nfc = sc_.take<int32_t>(2);
  float* ia = sa_.take<float>(CRUX_INFO_N); double* ssq_a = sa_.take<double>(2 + SUMSQ_BLOCKS); int32_t* sta = sa_.take<int32_t>(1); int32_t* nfa = sa_.take<int32_t>(2);
  const float* S = (const float*)batch->col[CRUX_COL_S]; const float* SP = (const float*)batch->col[CRUX_COL_SP];
  const float* ls = actor->p + actor->nd.xoff; const float* w = use_weight ? (const float*)batch->col[CRUX_COL_WEIGHT] : nullptr;
  auto l3 = [&](crux_mlp* n, bool store) { const NetDesc& nd = n->nd; return TileSet{n->p + nd.woff[2], n->p + nd.boff[2], crux_dense_act(n, 2), store ? crux_dense_act(n, 3) : nullptr}; };
  const unsigned nt = (unsigned)((B + 15) / 16);
  // 0, 1: rand!
  rc = crux_uniform_sample(batch, source, B, nullptr, sample_counter); if (rc) return bail(rc);
  sect([](int kid) { return kid == OP_UNIFORM_IDS ? 0 : kid == OP_GATHER_RING_ALL ? 1 : -1; });
  rc = check_sac(c, actor, q1, q2, la, batch, "sac_epoch"); if (rc) return bail(rc);
  rc = crux_exec_zero(c, it, 256 * 5, c->stream); if (!rc) rc = crux_exec_zero(c, ic, 256 * 6, c->stream); if (!rc) rc = crux_exec_zero(c, ia, 256 * 6, c->stream);
  if (!rc) rc = crux_exec_zero(c, la->g, sizeof(float) * (size_t)la->nd.n_params, c->stream); if (rc) return bail(rc);
  only(1);
  // 2
  rc = crux_dense_forward12(actor, SP, B, c->stream); if (rc) return bail(rc);
  CRUX_RUN(c, ConcatSaOp, OP_CONCAT_SA, k_concat_sa, nblk(B * sd), 256, c->stream, S, (const float*)batch->col[CRUX_COL_A], od, ad, B, sa_c);
  only(2);
  // 3
  { ActorExploreArgs a{}; a.mu = l3(actor, true); a.ls = ls; a.s = SP; a.od = od; a.ad = ad; a.K = K; a.B = (int32_t)B; a.n_cfg = 1; a.seed = noise_seed; a.cfg[0] = ExploreCfg{noise_counter0, sa_t, lp_t, nullptr};
    crux_exec_push<ActorExploreTileOp, OP_ACTOR_EXPLORE_TILE>(c, nt, a); }
  rc = crux_dense_forward12(q1, sa_c, B, c->stream); if (!rc) rc = crux_dense_forward12(q2, sa_c, B, c->stream); if (rc) return bail(rc);
  only(3);
  // 4
  rc = crux_dense_forward12(q1t, sa_t, B, c->stream); if (!rc) rc = crux_dense_forward12(q2t, sa_t, B, c->stream); if (!rc) rc = crux_dense_forward12(actor, S, B, c->stream); if (rc) return bail(rc);
  only(4);
  // 5
  { SacCriticArgs a{}; a.q1t = l3(q1t, true); a.q2t = l3(q2t, true); a.q1 = l3(q1, true); a.q2 = l3(q2, true); a.r = (const float*)batch->col[CRUX_COL_R]; a.done = (const uint8_t*)batch->col[CRUX_COL_DONE];
    a.lp = lp_t; a.log_alpha = la->p; a.w = w; a.gamma = gamma; a.scale = 0.5f; a.K = K; a.B = (int32_t)B; a.y = y; a.dy1 = dy1; a.dy2 = dy2; a.term1 = t1; a.term2 = t2;
    crux_exec_push<SacCriticTileOp, OP_SAC_CRITIC_TILE>(c, nt, a); }
  { ActorExploreArgs a{}; a.mu = l3(actor, true); a.ls = ls; a.s = S; a.od = od; a.ad = ad; a.K = K; a.B = (int32_t)B; a.n_cfg = 1; a.seed = noise_seed;      // the two draws side by side (each re-evaluates the output layer: 8 MFMAs)
    a.cfg[0] = ExploreCfg{noise_counter0 + 1, nullptr, lp_temp, nullptr}; crux_exec_push<ActorExploreTileOp, OP_ACTOR_EXPLORE_TILE>(c, nt, a);
    a.cfg[0] = ExploreCfg{noise_counter0 + 2, sa_a, lp_a, eps}; crux_exec_push<ActorExploreTileOp, OP_ACTOR_EXPLORE_TILE>(c, nt, a); }
  only(5);
  // 6
  Sumsq2Fix fxc{}, fxa{};
  rc = crux_dense_backward(q1, sa_c, B, dy1, 1.0f, true, nullptr, c->stream, &fxc, 0, nfc); if (!rc) rc = crux_dense_backward(q2, sa_c, B, dy2, 1.0f, true, nullptr, c->stream, &fxc, 1, nfc); if (rc) return bail(rc);
  CRUX_RUN(c, TempHeadOp, OP_TEMP_HEAD, k_temp_head, 1, 256, c->stream, (const float*)lp_temp, B, H_target, (const float*)la->p, la->g, it, ssq_t);
  only(6);
  // 7: the critics' norm (for the info row) and, beside it, their Adam steps gated on the pullback's NaN flags (AdamSelfOp, sac.hip); log alpha's step (+ 8: the advances)
  CRUX_RUN(c, Sumsq2Op, OP_SUMSQ2, k_sumsq2, SUMSQ_BLOCKS, 256, c->stream, q1->g, (int64_t)q1->nd.n_params, q2->g, (int64_t)q2->nd.n_params, ssq_c, fxc);
  rc = adam_self(q1, nfc, stc, fxc, 0); if (!rc) rc = adam_self(q2, nfc, stc, fxc, 1); if (!rc) rc = adam_gated(la, ssq_t, stt, false); if (rc) return bail(rc);
  sect([](int kid) { return (kid == OP_ADAM_ADVANCE || kid == OP_ADAM_ADVANCE_SELF) ? 8 : 7; });
  // 8: critic info | the updated critics' first two layers on (s, a ~ pi)
  crux_exec_push<CriticInfo2Op, OP_CRITIC_INFO2>(c, 1u, (const float*)t1, (const float*)crux_dense_act(q1, 3), (const float*)t2, (const float*)crux_dense_act(q2, 3), (const double*)ssq_c, B, ic);
  rc = crux_dense_forward12(q1, sa_a, B, c->stream); if (!rc) rc = crux_dense_forward12(q2, sa_a, B, c->stream); if (rc) return bail(rc);
  only(8);
  // 9
  { SacActorArgs a{}; a.q1 = l3(q1, true); a.q2 = l3(q2, true); a.lp = lp_a; a.log_alpha = la->p; a.K = K; a.B = (int32_t)B; a.dy1 = da1; a.dy2 = da2; a.term = ta;
    crux_exec_push<SacActorTileOp, OP_SAC_ACTOR_TILE>(c, nt, a); }
  only(9);
  // 10
  const float* dz1a = nullptr; const float* dz1b = nullptr;
  rc = crux_dense_dgrad_to_dz1(q1, sa_a, B, da1, &dz1a, c->stream); if (!rc) rc = crux_dense_dgrad_to_dz1(q2, sa_a, B, da2, &dz1b, c->stream); if (rc) return bail(rc);
  only(10);
  // 11
  { CriticDxArgs a{}; a.c1 = TileSet{q1->p + q1->nd.woff[0], nullptr, dz1a, nullptr}; a.c2 = TileSet{q2->p + q2->nd.woff[0], nullptr, dz1b, nullptr};
    a.sa = sa_a; a.mu = crux_dense_act(actor, 3); a.eps = eps; a.ls = ls; a.log_alpha = la->p; a.od = od; a.ad = ad; a.K = q1->nd.dims[1]; a.B = (int32_t)B; a.dmu = dmu; a.dls = dls;
    crux_exec_push<CriticDxActorGradTileOp, OP_CRITIC_DX_TILE>(c, nt, a); }
  only(11);
  // 12
  rc = crux_dense_backward(actor, S, B, dmu, 1.0f, true, nullptr, c->stream, &fxa, 0, nfa); if (rc) return bail(rc);
  CRUX_RUN(c, RowsumOp, OP_ROWSUM, k_rowsum, ad, 256, c->stream, (const float*)dls, ad, B, actor->g + actor->nd.xoff, nfa);
  only(12);
  // 13: the actor's norm | its Adam step (self-gated)
  CRUX_RUN(c, Sumsq2Op, OP_SUMSQ2, k_sumsq2, SUMSQ_BLOCKS, 256, c->stream, actor->g, (int64_t)actor->nd.n_params, (float*)nullptr, (int64_t)0, ssq_a, fxa);
  rc = adam_self(actor, nfa, sta, fxa, 0); if (rc) return bail(rc);
  sect([](int kid) { return kid == OP_ADAM_ADVANCE_SELF ? 14 : 13; });
  // 14: actor info, polyak
  crux_exec_push<ActorInfo2Op, OP_ACTOR_INFO2>(c, 1u, (const float*)ta, (const float*)lp_a, (const double*)ssq_a, B, ia);
  if (actor_targ) { rc = crux_polyak(actor_targ, actor, tau); if (rc) return bail(rc); }
  rc = crux_polyak(q1t, q1, tau); if (!rc) rc = crux_polyak(q2t, q2, tau); if (rc) return bail(rc);
  only(14);
  // the steps' read-backs in the order they ran (temperature, critics, actor), as the generic recording registers them
  crux_exec_add_readback(c, info_temp, it, stt, "sac_temp_loss"); crux_exec_add_readback(c, info_critic, ic, stc, "double_Q_loss"); crux_exec_add_readback(c, info_actor, ia, sta, "sac_actor_loss");
  // phases 0 / 1 of a chained epoch (ids | gather, fills) overlap the previous epoch's actor pullback (12: the gather rewrites the batch rows it read -- not before 13) and
  // norm + Adam (13), phase 2 (the new actor's forward on s') its info + advance + polyak (14): base + p - 3 for every p. 15 phases, 12 launches per chained epoch.
  if (r->chain) {
    if (!(plan_ok && ph.size() == r->ops.size() - ops0)) r->chain_ok = false;
    r->chain_tags.insert(r->chain_tags.end(), ph.begin(), ph.end()); r->chain_base += 15 - (r->chain_base > 0 ? 3 : 0);
    return CRUX_OK;
  }
  if (plan_ok && ph.size() == r->ops.size()) { rc = exec_schedule(c, ph); if (rc) return bail(rc); }
  return crux_exec_run(c);
}

// One epoch of value_training with SAC's pieces (off_policy.jl:69-104, rl/sac.jl): rand! -> sac_target -> train!(log_alpha, sac_temp_loss) ->
// [train!(critic, double_Q_loss)] -> [train!(actor, sac_actor_loss) -> polyak_average!(pi_minus, pi, tau)] as ONE fused launch.
int32_t crux_sac_epoch(crux_mlp* actor, crux_mlp* q1, crux_mlp* q2, crux_mlp* actor_targ, crux_mlp* q1_targ, crux_mlp* q2_targ, crux_mlp* log_alpha,
                       crux_buffer* source, crux_buffer* batch, float gamma, float H_target, float tau, int32_t use_weight, int32_t update_critic, int32_t update_actor,
                       uint64_t sample_counter, uint64_t noise_seed, uint64_t noise_counter0, float* info_temp, float* info_critic, float* info_actor) {
  if (!actor || !q1 || !q2 || !q1_targ || !q2_targ || !log_alpha || !source || !batch) return CRUX_EINVAL;
  crux_ctx* c = actor->ctx; const int64_t B = batch->capacity;
  if (source->prioritized) return crux_fail(c, CRUX_EUNSUP, "sac_epoch: prioritized replay over a DoubleNetwork critic is not defined (td_error, src/utils.jl:112)");
  if (c->rec && rec_of(c)->chain && sac_tile_case(actor, q1, q2, q1_targ, q2_targ, source, batch, update_critic, update_actor))      // chained epochs of the C4 family: the tile ops (14 launches per epoch)
    return sac_epoch_tiles(actor, q1, q2, actor_targ, q1_targ, q2_targ, log_alpha, source, batch, gamma, H_target, tau, use_weight, sample_counter, noise_seed, noise_counter0, info_temp, info_critic, info_actor);
  const bool fuse = !crux_sw().no_fused_epoch;
  int32_t rc; float* d_y = nullptr;
  if (fuse) { if (!crux_exec_recording(c)) { rc = crux_exec_begin(c); if (rc) return rc; }       // a chained recording (crux_sac_epochs) is already open
    d_y = (float*)crux_exec_small(c, 4 * (size_t)B);
    if (!d_y) { crux_exec_abort(c); return crux_fail(c, CRUX_EUNSUP, "sac_epoch: batch of %lld rows exceeds the executor's region", (long long)B); } }
  else { if (c->epoch_tmp_bytes < 4 * (size_t)B + 256) { if (c->epoch_tmp) { HIPCHK(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->epoch_tmp); }
      c->epoch_tmp_bytes = 16 * (size_t)B + 4096; if (hipMalloc(&c->epoch_tmp, c->epoch_tmp_bytes) != hipSuccess) { c->epoch_tmp = nullptr; c->epoch_tmp_bytes = 0; return crux_fail(c, CRUX_ENOMEM, "sac_epoch: targets"); } }
    d_y = (float*)c->epoch_tmp; }
  auto bail = [&](int32_t e) { if (fuse) crux_exec_abort(c); return e; };
  // Phase plan (LA = layers of the actor, LQ = of a critic; X = 3 + LA + LQ, Y = X + 4 + LQ). Chains that touch different networks run side by side:
  //   0 ids | 1 gather, zero-fills | 2.. actor(sp) forward ; vcat(s, a) ; Q1 || Q2 forward on (s, a) | 2+LA exploration at sp | 3+LA.. target Q1 || Q2 ; actor(s) forward
  //   X sac_target ; exploration at s | X+1 critic heads ; temperature head | X+2.. critic backward (weight || data gradient, both critics) ; Adam(log_alpha) | .. norm | info, Adam(Q1), Adam(Q2)
  //   X+1.. actor(s) forward of the ACTOR step, exploration (beside the critic's backward) | Y.. Q1 || Q2 forward | actor head | Q1 || Q2 input gradients | reverse of exploration | actor backward ; logSigma row sums | norm | info, Adam | polyak
  // The order inside every chain is the reference's (temperature before critic before actor: each sees the parameters the previous step left).
  // sq: sac_target (one block at B <= 256) and the critic heads (one block each) form a sequential group in ONE block of phase X when the critics train; the critic
  // chain and everything behind it (Y ..) then sit one phase earlier. The temperature chain (X .. X + 3) is independent of it.
  const int sq = (update_critic && B <= 256 && !exec_persistent_on(c)) ? 1 : 0;
  // Round 4: written in launches -- FA / FQ forward launches of the actor / a critic, BQ / BA phases of a pullback with parameter gradients (BQo ops per critic), LQ phases
  // of a critic's input-gradient chain; with the fused block kernels FA = FQ = L - 1 and BQ = BA = L - 1 (net_plan above), otherwise all equal L as in round 3.
  const NetPlan pa = net_plan(actor, B), pq = net_plan(q1, B);
  std::vector<int> ph; bool plan_ok = true; const int LA = actor->nd.L, LQ = q1->nd.L, FA = pa.nf, FQ = pq.nf, BQ = pq.nb, BQo = pq.nbops, BA = pa.nb, DQ = pq.dq, X = 3 + FA + FQ, Y = X + 4 + BQ - sq;
  if (LA != LQ || q2->nd.L != LQ || q1_targ->nd.L != LQ || q2_targ->nd.L != LQ) plan_ok = false;       // (the early actor forward needs X + 1 + FA < Y)
  { const NetPlan p2 = net_plan(q2, B), t1 = net_plan(q1_targ, B), t2 = net_plan(q2_targ, B); if (p2.nf != FQ || p2.nb != BQ || p2.one != pq.one || t1.nf != FQ || t2.nf != FQ || FA != FQ) plan_ok = false; }
  // chained epochs: phases 0 (ids) and 1 (gather, fills) of a later epoch run beside the previous epoch's actor norm and info + Adam (neither reads the batch), and its
  // phase 2 (actor(sp) forward, vcat(s, a): online networks only) beside the previous epoch's advance + polyak, which writes beta powers and TARGET networks: the rest
  // closes up by three -- see crux_dqn_epoch
  auto tag = [&](size_t from, auto&& rule) { if (!fuse) return; ExecRec* r = rec_of(c); int g = 0; const int base = r->chain ? r->chain_base : 0;
    for (size_t i = from; i < r->ops.size(); ++i) { int p = rule(r->ops[i].kid, g); if (p < 0) { plan_ok = false; p = 0; }
      const int sub = p >> 12; p &= 4095;
      ph.push_back(ph_tag(base > 0 ? (p < 2 ? base - 3 + p : base + p - 3) : p, sub)); } };
  const size_t ops0 = fuse ? exec_mark(c) : 0;
  size_t m = ops0;
  rc = crux_uniform_sample(batch, source, B, nullptr, sample_counter); if (rc) return bail(rc);
  tag(m, [&](int kid, int&) { return kid == OP_UNIFORM_IDS ? 0 : kid == OP_GATHER_RING_ALL ? 1 : -1; });
  m = fuse ? exec_mark(c) : 0;
  rc = crux_sac_target(actor, q1_targ, q2_targ, log_alpha, batch, gamma, noise_seed, noise_counter0, d_y); if (rc) return bail(rc);
  tag(m, [&](int kid, int& g) { if (is_mm(kid)) { const int k = g++; return k < FA ? 2 + k : 3 + FA + (k - FA) % FQ; }
    return kid == OP_GAUSS_EXPLORE ? 2 + FA : kid == OP_SAC_TARGET ? X | (sq ? PH_SEQ_HEAD : 0) : -1; });
  m = fuse ? exec_mark(c) : 0;
  rc = crux_sac_temp_step(actor, log_alpha, batch, H_target, noise_seed, noise_counter0 + 1, info_temp); if (rc) return bail(rc);
  tag(m, [&](int kid, int& g) { if (kid == OP_FILL) return 1; if (is_mm(kid)) { const int k = g++; return k < FA ? 3 + FA + k : -1; }
    return kid == OP_GAUSS_EXPLORE ? X : kid == OP_TEMP_HEAD ? X + 1 : kid == OP_ADAM_GATED ? X + 2 : kid == OP_ADAM_ADVANCE ? X + 3 : -1; });
  if (update_critic) {
    m = fuse ? exec_mark(c) : 0;
    rc = crux_double_q_step(q1, q2, batch, d_y, use_weight, info_critic); if (rc) return bail(rc);
    tag(m, [&](int kid, int& g) {      // per critic: LQ forward GEMMs, head, then (weight, data) pairs from the last layer down (the first layer has no data gradient)
      if (kid == OP_FILL) return 1; if (kid == OP_CONCAT_SA) return 2;
      if (is_mm(kid)) { const int k = (g++) % (FQ + BQo); return k < FQ ? 3 + k : X + 2 - sq + pq.stage(k - FQ); }
      return kid == OP_Q_HEAD ? (sq ? (X | PH_SEQ_TAIL) : X + 1) : kid == OP_SUMSQ2 ? X + 2 - sq + BQ : (kid == OP_CRITIC_INFO || kid == OP_ADAM_GATED) ? X + 3 - sq + BQ : kid == OP_ADAM_ADVANCE ? X + 4 - sq + BQ : -1; });
  }
  if (update_actor) {
    m = fuse ? exec_mark(c) : 0;
    rc = crux_sac_actor_step(actor, q1, q2, log_alpha, batch, noise_seed, noise_counter0 + 2, info_actor); if (rc) return bail(rc);
    tag(m, [&](int kid, int& g) {      // GEMMs: LA actor forward, LQ + LQ critic forwards, LQ + LQ critic input gradients, then the actor's (weight, data) pairs
      if (kid == OP_FILL) return 1;
      // the actor's own forward pass and its exploration draw depend on neither the critic update nor the temperature: they run beside the critic's head / backward
      // phases (X + 1 ..), after the temperature step's last read of the actor's activations (its exploration at X)
      if (is_mm(kid)) { const int k = g++; if (k < FA) return X + 1 + k; if (k < FA + 2 * FQ) return Y + (k - FA) % FQ;
        if (k < FA + 2 * FQ + 2 * DQ) return Y + 1 + FQ + (k - FA - 2 * FQ) % DQ; return Y + 2 + FQ + DQ + pa.stage(k - FA - 2 * FQ - 2 * DQ); }
      return kid == OP_GAUSS_EXPLORE ? X + 1 + FA : kid == OP_ACTOR_HEAD ? Y + FQ : kid == OP_ACTOR_GRAD ? Y + 1 + FQ + DQ : kid == OP_ROWSUM ? Y + 2 + FQ + DQ :
             kid == OP_SUMSQ2 ? Y + BA + 2 + FQ + DQ : (kid == OP_ACTOR_INFO || kid == OP_ADAM_GATED) ? Y + BA + 3 + FQ + DQ : kid == OP_ADAM_ADVANCE ? Y + BA + 4 + FQ + DQ : -1; });
    m = fuse ? exec_mark(c) : 0;
    if (actor_targ) { rc = crux_polyak(actor_targ, actor, tau); if (rc) return bail(rc); }
    rc = crux_polyak(q1_targ, q1, tau); if (rc) return bail(rc);
    rc = crux_polyak(q2_targ, q2, tau); if (rc) return bail(rc);
    tag(m, [&](int kid, int&) { return kid == OP_POLYAK ? Y + BA + 4 + FQ + DQ : -1; });
  }
  if (fuse && rec_of(c)->chain) {      // chained: crux_sac_epochs schedules and runs the whole list
    ExecRec* r = rec_of(c);
    if (!(plan_ok && ph.size() == r->ops.size() - ops0)) r->chain_ok = false;
    r->chain_tags.insert(r->chain_tags.end(), ph.begin(), ph.end()); r->chain_base += Y + BA + 5 + FQ + DQ - (r->chain_base > 0 ? 3 : 0);
    return CRUX_OK;
  }
  if (fuse && plan_ok && ph.size() == rec_of(c)->ops.size()) { rc = exec_schedule(c, ph); if (rc) return bail(rc); }
  return fuse ? crux_exec_run(c) : CRUX_OK;
}

// value_training's epoch loop with SAC's pieces (off_policy.jl:69-104; SAC's c_opt.epochs = dN = 50, rl/sac.jl) in chains of up to 8 epochs per recorded list.
// Epoch e (global index epoch0 + e within the iteration) trains the critic when (epoch0 + e) % critic_every == 0 and the actor (then the target update) when
// (epoch0 + e) % actor_every == 0 (:91,96); it draws with sample counter sample_counter0 + e and noise counters noise_counter0 + 3 e .. + 2. infos_*: host [n x CRUX_INFO_N].
static int32_t sac_epochs_impl(crux_mlp* actor, crux_mlp* q1, crux_mlp* q2, crux_mlp* actor_targ, crux_mlp* q1_targ, crux_mlp* q2_targ, crux_mlp* log_alpha,
                        crux_buffer* source, crux_buffer* batch, float gamma, float H_target, float tau, int32_t use_weight, int32_t epoch0, int32_t n_epochs,
                        int32_t critic_every, int32_t actor_every, uint64_t sample_counter0, uint64_t noise_seed, uint64_t noise_counter0,
                        float* infos_temp, float* infos_critic, float* infos_actor, float* d_infos_async) {
  if (!actor || n_epochs < 1 || critic_every < 1 || actor_every < 1) return CRUX_EINVAL;
  crux_ctx* c = actor->ctx;
  const bool fuse = !crux_sw().no_fused_epoch && !crux_sw().no_chained_epochs;
  if (d_infos_async && !fuse) return CRUX_EUNSUP;
  auto flush = [&]() -> int32_t {
    if (!crux_exec_recording(c)) return CRUX_OK;
    ExecRec* r = rec_of(c); r->chain = false;
    if (r->chain_ok && r->chain_tags.size() == r->ops.size()) { const int32_t rs = exec_schedule(c, r->chain_tags); if (rs) { crux_exec_abort(c); return rs; } }
    r->async = d_infos_async != nullptr;
    return crux_exec_run(c);
  };
  int32_t rc = CRUX_OK; int in_chain = 0;
  for (int e = 0; e < n_epochs; ++e) {
    const int ge = epoch0 + e; const int32_t uc = ge % critic_every == 0, ua = ge % actor_every == 0;
    float* it = infos_temp ? infos_temp + (size_t)e * CRUX_INFO_N : nullptr; float* ic = infos_critic ? infos_critic + (size_t)e * CRUX_INFO_N : nullptr; float* ia = infos_actor ? infos_actor + (size_t)e * CRUX_INFO_N : nullptr;
    if (fuse) {
      if (in_chain >= 8) { rc = flush(); in_chain = 0; if (rc) return rc; }
      if (!in_chain) { rc = crux_exec_begin(c); if (rc) return rc; }
      rec_of(c)->chain = true;
    }
    const size_t rb0 = fuse ? rec_of(c)->readbacks.size() : 0, tg0 = fuse ? rec_of(c)->chain_tags.size() : 0;
    rc = crux_sac_epoch(actor, q1, q2, actor_targ, q1_targ, q2_targ, log_alpha, source, batch, gamma, H_target, tau, use_weight, uc, ua,
                        sample_counter0 + (uint64_t)e, noise_seed, noise_counter0 + 3ull * (uint64_t)e, it, ic, ia);
    if (rc) { if (fuse && c->rec) { rec_of(c)->chain = false; crux_exec_abort(c); } return rc; }
    const bool tiles = fuse && sac_tile_case(actor, q1, q2, q1_targ, q2_targ, source, batch, uc, ua);
    if (d_infos_async) {      // the epoch's info rows (temperature, [critics], [actor] -- the order the steps ran in) go to rows 3 e .. 3 e + 2 of the caller's device array, copied in the epoch's last phase
      ExecRec* r = rec_of(c);
      if (r->readbacks.size() != rb0 + 1 + (uc ? 1 : 0) + (ua ? 1 : 0) || r->chain_tags.size() != r->ops.size()) { r->chain = false; crux_exec_abort(c); return crux_fail(c, CRUX_EHIP, "sac epochs (async): unexpected recording"); }
      int tmax = 0; for (size_t k = tg0; k < r->chain_tags.size(); ++k) tmax = std::max(tmax, r->chain_tags[k] & ~3);
      size_t q = rb0; const int slot_of[3] = {0, uc ? 1 : -1, ua ? 2 : -1};
      for (int sl = 0; sl < 3; ++sl) { if (slot_of[sl] < 0) continue;
        crux_exec_push<CopyF32Op, OP_COPY_F32>(c, 1u, d_infos_async + ((size_t)e * 3 + sl) * CRUX_INFO_N, (const float*)r->readbacks[q++].d_info, (int64_t)CRUX_INFO_N);
        r->chain_tags.push_back(tmax + (tiles ? 4 : 0)); } }      // (tile plans: the actor's info op sits IN the epoch's last phase -- the copy joins the launch after it, the next epoch's third)
    if (fuse) ++in_chain;
  }
  return fuse ? flush() : rc;
}
int32_t crux_sac_epochs(crux_mlp* actor, crux_mlp* q1, crux_mlp* q2, crux_mlp* actor_targ, crux_mlp* q1_targ, crux_mlp* q2_targ, crux_mlp* log_alpha,
                        crux_buffer* source, crux_buffer* batch, float gamma, float H_target, float tau, int32_t use_weight, int32_t epoch0, int32_t n_epochs,
                        int32_t critic_every, int32_t actor_every, uint64_t sample_counter0, uint64_t noise_seed, uint64_t noise_counter0,
                        float* infos_temp, float* infos_critic, float* infos_actor) {
  // More than one chain (n_epochs > 8): the chains run back to back without a read-back between them -- the asynchronous form with the info rows in a device block of the
  // context -- and the host synchronises ONCE, at the end of the call: the device no longer idles while the host reads back, records and uploads the next chain
  // (C4, 50 epochs per call: 180 -> ~155 us per epoch). Same results; a NaN gradient norm is reported from the rows.
  if (actor && n_epochs > 8 && critic_every >= 1 && actor_every >= 1 && !crux_sw().no_fused_epoch && !crux_sw().no_chained_epochs && !crux_sw().sync_chains) {
    crux_ctx* c = actor->ctx; const size_t need = sizeof(float) * 3 * CRUX_INFO_N * (size_t)n_epochs;
    if (c->epoch_rows_bytes < need) { if (c->epoch_rows) { HIPCHK(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->epoch_rows); c->epoch_rows = nullptr; c->epoch_rows_bytes = 0; }
      if (hipMalloc(&c->epoch_rows, 2 * need) != hipSuccess) return crux_fail(c, CRUX_ENOMEM, "sac epochs: info rows"); c->epoch_rows_bytes = 2 * need; }
    HIPCHK(c, hipMemsetAsync(c->epoch_rows, 0, need, c->stream));
    int32_t rc = sac_epochs_impl(actor, q1, q2, actor_targ, q1_targ, q2_targ, log_alpha, source, batch, gamma, H_target, tau, use_weight, epoch0, n_epochs, critic_every, actor_every,
                                 sample_counter0, noise_seed, noise_counter0, nullptr, nullptr, nullptr, (float*)c->epoch_rows);
    if (rc) return rc;
    std::vector<float> rows(3 * CRUX_INFO_N * (size_t)n_epochs);
    HIPCHK(c, hipMemcpyAsync(rows.data(), c->epoch_rows, need, hipMemcpyDeviceToHost, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream));
    bool nan = false;
    for (int e = 0; e < n_epochs; ++e) { const int ge = epoch0 + e; const bool has[3] = {true, ge % critic_every == 0, ge % actor_every == 0}; float* dst[3] = {infos_temp, infos_critic, infos_actor};
      for (int sl = 0; sl < 3; ++sl) { const float* row = rows.data() + ((size_t)e * 3 + sl) * CRUX_INFO_N;
        if (has[sl] && row[CRUX_INFO_GRAD_NORM] != row[CRUX_INFO_GRAD_NORM]) nan = true;
        if (dst[sl] && has[sl]) memcpy(dst[sl] + (size_t)e * CRUX_INFO_N, row, sizeof(float) * CRUX_INFO_N); } }
    if (nan) return crux_fail(c, CRUX_ENAN, "NaN detected! (grad norm is NaN, src/training.jl:20) in the SAC epochs");
    return CRUX_OK;
  }
  return sac_epochs_impl(actor, q1, q2, actor_targ, q1_targ, q2_targ, log_alpha, source, batch, gamma, H_target, tau, use_weight, epoch0, n_epochs, critic_every, actor_every,
                         sample_counter0, noise_seed, noise_counter0, infos_temp, infos_critic, infos_actor, nullptr);
}
// crux_sac_epochs without the host in the loop (see crux_dqn_epochs_async): d_infos is DEVICE memory, [n_epochs][3][CRUX_INFO_N] = temperature | critics | actor rows of every
// epoch (rows of steps an epoch skipped -- critic_every / actor_every -- are left as they were).
int32_t crux_sac_epochs_async(crux_mlp* actor, crux_mlp* q1, crux_mlp* q2, crux_mlp* actor_targ, crux_mlp* q1_targ, crux_mlp* q2_targ, crux_mlp* log_alpha,
                              crux_buffer* source, crux_buffer* batch, float gamma, float H_target, float tau, int32_t use_weight, int32_t epoch0, int32_t n_epochs,
                              int32_t critic_every, int32_t actor_every, uint64_t sample_counter0, uint64_t noise_seed, uint64_t noise_counter0, float* d_infos) {
  if (!d_infos) return CRUX_EINVAL;
  return sac_epochs_impl(actor, q1, q2, actor_targ, q1_targ, q2_targ, log_alpha, source, batch, gamma, H_target, tau, use_weight, epoch0, n_epochs, critic_every, actor_every,
                         sample_counter0, noise_seed, noise_counter0, nullptr, nullptr, nullptr, d_infos);
}

// One epoch of value_training with DDPG's / TD3's pieces (off_policy.jl:69-104; rl/ddpg.jl, rl/td3.jl): rand! -> ddpg_target | td3_target -> [train!(critic, td_loss |
// double_Q_loss)] -> [train!(actor, -mean(Q(s, mu(s)))) -> polyak_average!] recorded as one list. q2 / q2_targ = NULL: single critic (DDPG); q2 with q2_targ = NULL:
// twin critics trained against a single-target (not used by the reference's solvers).
static int32_t dpg_epoch(crux_mlp* actor, crux_mlp* q1, crux_mlp* q2, crux_mlp* actor_targ, crux_mlp* q1_targ, crux_mlp* q2_targ, crux_buffer* source, crux_buffer* batch,
                         float gamma, float tau, float sigma, float eps_min, float eps_max, float a_min, float a_max, int32_t use_weight, int32_t update_critic, int32_t update_actor,
                         uint64_t sample_counter, uint64_t noise_seed, uint64_t noise_counter, float* info_critic, float* info_actor) {
  crux_ctx* c = actor->ctx; const int64_t B = batch->capacity;
  if (source->prioritized) return crux_fail(c, CRUX_EUNSUP, "dpg_epoch: DDPG / TD3 with a prioritized buffer is not wired up");
  int32_t rc;
  if (!crux_exec_recording(c)) { rc = crux_exec_begin(c); if (rc) return rc; }
  float* d_y = (float*)crux_exec_small(c, 4 * (size_t)B);
  if (!d_y) { crux_exec_abort(c); return crux_fail(c, CRUX_EUNSUP, "dpg_epoch: batch of %lld rows exceeds the executor's region", (long long)B); }
  auto bail = [&](int32_t e) { crux_exec_abort(c); return e; };
  // Phase plan (LA = layers of the actor, LQ = of a critic; X = 3 + LA + LQ, Y = X + 4 + LQ), chains that touch different networks side by side:
  //   0 ids | 1 gather, fills | 2.. target actor(sp) forward ; vcat(s, a) ; actor(s) forward of the ACTOR step | 2+LA target action (+ smoothing noise) ; mu(s) -> vcat(s, mu(s))
  //   3+LA.. target Q1 || Q2 forward (and, from 3: Q1 || Q2 forward on (s, a)) | X target | X+1 critic heads | X+2.. critic backward | norm | info, Adam | advance
  //   Y.. Q(s, mu(s)) forward | its input gradient | slice | actor backward | norm | info, Adam | advance, polyak
  const int sq = (update_critic && B <= 256 && !exec_persistent_on(c)) ? 1 : 0;      // target + critic head(s) as a sequential one-block group (see crux_sac_epoch)
  const NetPlan pa = net_plan(actor, B), pq = net_plan(q1, B);      // launches per pass (see crux_sac_epoch)
  std::vector<int> ph; bool plan_ok = true; const int LA = actor->nd.L, LQ = q1->nd.L, FA = pa.nf, FQ = pq.nf, BQ = pq.nb, BQo = pq.nbops, BA = pa.nb, DQ = pq.dq, X = 3 + FA + FQ, Y = X + 4 + BQ - sq;
  const int ag = actor->nd.acts[LA - 1] != CRUX_ACT_IDENTITY ? 1 : 0;
  if (LA != LQ || actor_targ->nd.L != LA || q1_targ->nd.L != LQ || (q2 && q2->nd.L != LQ) || (q2_targ && q2_targ->nd.L != LQ)) plan_ok = false;
  if (FA != FQ || net_plan(actor_targ, B).nf != FA || net_plan(q1_targ, B).nf != FQ || (q2 && (net_plan(q2, B).nf != FQ || net_plan(q2, B).nb != BQ)) || (q2_targ && net_plan(q2_targ, B).nf != FQ)) plan_ok = false;
  // chained epochs: the sampling of epoch e + 1 beside the actor's norm and info + Adam of epoch e; its phase 2 reads the TARGET actor, which polyak (last phase) writes,
  // so the rest closes up by two only
  auto tag = [&](size_t from, auto&& rule) { ExecRec* r = rec_of(c); int g = 0; const int base = r->chain ? r->chain_base : 0;
    for (size_t i = from; i < r->ops.size(); ++i) { int p = rule(r->ops[i].kid, g); if (p < 0) { if (crux_sw().verbose) fprintf(stderr, "[cruxhip] dpg_epoch: op kind %d has no phase in the plan\n", r->ops[i].kid); plan_ok = false; p = 0; }
      const int sub = p >> 12; p &= 4095;
      ph.push_back(ph_tag(base > 0 ? (p < 2 ? base - 3 + p : base + p - 2) : p, sub)); } };
  const size_t ops0 = exec_mark(c);
  size_t m = ops0;
  rc = crux_uniform_sample(batch, source, B, nullptr, sample_counter); if (rc) return bail(rc);
  tag(m, [&](int kid, int&) { return kid == OP_UNIFORM_IDS ? 0 : kid == OP_GATHER_RING_ALL ? 1 : -1; });
  m = exec_mark(c);
  rc = crux_dpg_target(actor_targ, q1_targ, q2_targ, batch, gamma, sigma, eps_min, eps_max, a_min, a_max, noise_seed, noise_counter, d_y); if (rc) return bail(rc);
  tag(m, [&](int kid, int& g) { if (is_mm(kid)) { const int k = g++; return k < FA ? 2 + k : 3 + FA + (k - FA) % FQ; }
    return kid == OP_DPG_ACTION ? 2 + FA : kid == OP_DPG_TARGET ? X | (sq ? PH_SEQ_HEAD : 0) : -1; });
  if (update_critic) {
    m = exec_mark(c);
    rc = q2 ? crux_double_q_step(q1, q2, batch, d_y, use_weight, info_critic) : crux_q_step(q1, batch, d_y, use_weight, info_critic); if (rc) return bail(rc);
    tag(m, [&](int kid, int& g) {      // per critic: LQ forward GEMMs, head, then (weight, data) pairs from the last layer down (the first layer has no data gradient)
      if (kid == OP_FILL) return 1; if (kid == OP_CONCAT_SA) return 2;
      if (is_mm(kid)) { const int k = (g++) % (FQ + BQo); return k < FQ ? 3 + k : X + 2 - sq + pq.stage(k - FQ); }
      return kid == OP_Q_HEAD ? (sq ? (X | PH_SEQ_TAIL) : X + 1) : kid == OP_SUMSQ2 ? X + 2 - sq + BQ : (kid == OP_CRITIC_INFO || kid == OP_ADAM_GATED) ? X + 3 - sq + BQ : kid == OP_ADAM_ADVANCE ? X + 4 - sq + BQ : -1; });
  }
  if (update_actor) {
    m = exec_mark(c);
    rc = crux_dpg_actor_step(actor, q1, batch, info_actor); if (rc) return bail(rc);
    tag(m, [&](int kid, int& g) {      // GEMMs: LA actor forward, LQ critic forward, LQ critic input gradients, then the actor's (weight, data) pairs
      if (kid == OP_FILL) return 1;      // the info / status rows and the constant dQ = -1 / B
      if (is_mm(kid)) { const int k = g++; if (k < FA) return 2 + k; if (k < FA + FQ) return Y + (k - FA); if (k < FA + FQ + DQ) return Y + FQ + (k - FA - FQ);
        return Y + FQ + DQ + 1 + ag + pa.stage(k - FA - FQ - DQ); }
      if (kid == OP_ACT_GRAD) return ag ? Y + FQ + DQ + 1 : -1;      // dZ = act'(mu) .* dmu of a bounded (tanh) action head, between the slice and the actor's backward GEMMs
      return kid == OP_DPG_ACTION ? 2 + FA : kid == OP_SLICE_ROWS ? Y + FQ + DQ : kid == OP_SUMSQ2 ? Y + FQ + DQ + 1 + ag + BA : (kid == OP_MEAN_INFO || kid == OP_ADAM_GATED) ? Y + FQ + DQ + 2 + ag + BA :
             kid == OP_ADAM_ADVANCE ? Y + FQ + DQ + 3 + ag + BA : -1; });
    m = exec_mark(c);
    rc = crux_polyak(actor_targ, actor, tau); if (rc) return bail(rc);
    rc = crux_polyak(q1_targ, q1, tau); if (rc) return bail(rc);
    if (q2 && q2_targ) { rc = crux_polyak(q2_targ, q2, tau); if (rc) return bail(rc); }
    tag(m, [&](int kid, int&) { return kid == OP_POLYAK ? Y + FQ + DQ + 3 + ag + BA : -1; });
  }
  ExecRec* r = rec_of(c);
  if (r->chain) {
    if (!(plan_ok && ph.size() == r->ops.size() - ops0)) r->chain_ok = false;
    r->chain_tags.insert(r->chain_tags.end(), ph.begin(), ph.end()); r->chain_base += Y + FQ + DQ + 4 + ag + BA - (r->chain_base > 0 ? 2 : 0);
    return CRUX_OK;
  }
  if (plan_ok && ph.size() == r->ops.size()) { rc = exec_schedule(c, ph); if (rc) return bail(rc); }
  return crux_exec_run(c);
}

// value_training's epoch loop with DDPG's / TD3's pieces in chains of up to 8 epochs per recorded list (see crux_sac_epochs). sigma < 0: no target-policy smoothing
// (DDPG); otherwise TD3's clamp(a' + clamp(sigma randn, eps_min, eps_max), a_min, a_max) with noise counters noise_counter0 + e. infos_*: host [n x CRUX_INFO_N].
static int32_t dpg_epochs_impl(crux_mlp* actor, crux_mlp* q1, crux_mlp* q2, crux_mlp* actor_targ, crux_mlp* q1_targ, crux_mlp* q2_targ, crux_buffer* source, crux_buffer* batch,
                        float gamma, float tau, float sigma, float eps_min, float eps_max, float a_min, float a_max, int32_t use_weight, int32_t epoch0, int32_t n_epochs,
                        int32_t critic_every, int32_t actor_every, uint64_t sample_counter0, uint64_t noise_seed, uint64_t noise_counter0, float* infos_critic, float* infos_actor,
                        float* d_infos_async) {
  if (!actor || !q1 || !actor_targ || !q1_targ || !source || !batch || n_epochs < 1 || critic_every < 1 || actor_every < 1) return CRUX_EINVAL;
  crux_ctx* c = actor->ctx;
  const bool chain = !crux_sw().no_chained_epochs;
  if (d_infos_async && !chain) return CRUX_EUNSUP;
  auto flush = [&]() -> int32_t {
    if (!crux_exec_recording(c)) return CRUX_OK;
    ExecRec* r = rec_of(c); r->chain = false;
    if (r->chain_ok && r->chain_tags.size() == r->ops.size()) { const int32_t rs = exec_schedule(c, r->chain_tags); if (rs) { crux_exec_abort(c); return rs; } }
    r->async = d_infos_async != nullptr;
    return crux_exec_run(c);
  };
  int32_t rc = CRUX_OK; int in_chain = 0;
  for (int e = 0; e < n_epochs; ++e) {
    const int ge = epoch0 + e; const int32_t uc = ge % critic_every == 0, ua = ge % actor_every == 0;
    float* ic = infos_critic ? infos_critic + (size_t)e * CRUX_INFO_N : nullptr; float* ia = infos_actor ? infos_actor + (size_t)e * CRUX_INFO_N : nullptr;
    if (chain) {
      if (in_chain >= 8) { rc = flush(); in_chain = 0; if (rc) return rc; }
      if (!in_chain) { rc = crux_exec_begin(c); if (rc) return rc; }
      rec_of(c)->chain = true;
    }
    const size_t rb0 = chain && c->rec ? rec_of(c)->readbacks.size() : 0, tg0 = chain && c->rec ? rec_of(c)->chain_tags.size() : 0;
    rc = dpg_epoch(actor, q1, q2, actor_targ, q1_targ, q2_targ, source, batch, gamma, tau, sigma, eps_min, eps_max, a_min, a_max, use_weight, uc, ua,
                   sample_counter0 + (uint64_t)e, noise_seed, noise_counter0 + (uint64_t)e, ic, ia);
    if (rc) { if (chain && c->rec) { rec_of(c)->chain = false; crux_exec_abort(c); } return rc; }
    if (d_infos_async) {      // rows 2 e (critic) and 2 e + 1 (actor) of the caller's device array, copied in the epoch's last phase
      ExecRec* r = rec_of(c);
      if (!crux_exec_recording(c) || r->readbacks.size() != rb0 + (uc ? 1 : 0) + (ua ? 1 : 0) || r->chain_tags.size() != r->ops.size()) { if (c->rec) { r->chain = false; crux_exec_abort(c); } return crux_fail(c, CRUX_EHIP, "dpg epochs (async): unexpected recording"); }
      int tmax = 0; for (size_t k = tg0; k < r->chain_tags.size(); ++k) tmax = std::max(tmax, r->chain_tags[k] & ~3);
      size_t q = rb0; const int has[2] = {uc, ua};
      for (int sl = 0; sl < 2; ++sl) { if (!has[sl]) continue;
        crux_exec_push<CopyF32Op, OP_COPY_F32>(c, 1u, d_infos_async + ((size_t)e * 2 + sl) * CRUX_INFO_N, (const float*)r->readbacks[q++].d_info, (int64_t)CRUX_INFO_N);
        r->chain_tags.push_back(tmax); } }
    if (chain) ++in_chain;
  }
  return chain ? flush() : rc;
}
// crux_dpg_epochs without the host in the loop (see crux_dqn_epochs_async): d_infos is DEVICE memory, [n_epochs][2][CRUX_INFO_N] = critic | actor rows of every epoch
int32_t crux_dpg_epochs_async(crux_mlp* actor, crux_mlp* q1, crux_mlp* q2, crux_mlp* actor_targ, crux_mlp* q1_targ, crux_mlp* q2_targ, crux_buffer* source, crux_buffer* batch,
                              float gamma, float tau, float sigma, float eps_min, float eps_max, float a_min, float a_max, int32_t use_weight, int32_t epoch0, int32_t n_epochs,
                              int32_t critic_every, int32_t actor_every, uint64_t sample_counter0, uint64_t noise_seed, uint64_t noise_counter0, float* d_infos) {
  if (!d_infos) return CRUX_EINVAL;
  return dpg_epochs_impl(actor, q1, q2, actor_targ, q1_targ, q2_targ, source, batch, gamma, tau, sigma, eps_min, eps_max, a_min, a_max, use_weight, epoch0, n_epochs, critic_every, actor_every,
                         sample_counter0, noise_seed, noise_counter0, nullptr, nullptr, d_infos);
}
int32_t crux_dpg_epochs(crux_mlp* actor, crux_mlp* q1, crux_mlp* q2, crux_mlp* actor_targ, crux_mlp* q1_targ, crux_mlp* q2_targ, crux_buffer* source, crux_buffer* batch,
                        float gamma, float tau, float sigma, float eps_min, float eps_max, float a_min, float a_max, int32_t use_weight, int32_t epoch0, int32_t n_epochs,
                        int32_t critic_every, int32_t actor_every, uint64_t sample_counter0, uint64_t noise_seed, uint64_t noise_counter0, float* infos_critic, float* infos_actor) {
  // several chains per call: run back to back, one synchronisation at the end of the call (see crux_sac_epochs)
  if (actor && n_epochs > 8 && critic_every >= 1 && actor_every >= 1 && !crux_sw().no_chained_epochs && !crux_sw().sync_chains) {
    crux_ctx* c = actor->ctx; const size_t need = sizeof(float) * 2 * CRUX_INFO_N * (size_t)n_epochs;
    if (c->epoch_rows_bytes < need) { if (c->epoch_rows) { HIPCHK(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->epoch_rows); c->epoch_rows = nullptr; c->epoch_rows_bytes = 0; }
      if (hipMalloc(&c->epoch_rows, 2 * need) != hipSuccess) return crux_fail(c, CRUX_ENOMEM, "dpg epochs: info rows"); c->epoch_rows_bytes = 2 * need; }
    HIPCHK(c, hipMemsetAsync(c->epoch_rows, 0, need, c->stream));
    int32_t rc = dpg_epochs_impl(actor, q1, q2, actor_targ, q1_targ, q2_targ, source, batch, gamma, tau, sigma, eps_min, eps_max, a_min, a_max, use_weight, epoch0, n_epochs, critic_every, actor_every,
                                 sample_counter0, noise_seed, noise_counter0, nullptr, nullptr, (float*)c->epoch_rows);
    if (rc) return rc;
    std::vector<float> rows(2 * CRUX_INFO_N * (size_t)n_epochs);
    HIPCHK(c, hipMemcpyAsync(rows.data(), c->epoch_rows, need, hipMemcpyDeviceToHost, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream));
    bool nan = false;
    for (int e = 0; e < n_epochs; ++e) { const int ge = epoch0 + e; const bool has[2] = {ge % critic_every == 0, ge % actor_every == 0}; float* dst[2] = {infos_critic, infos_actor};
      for (int sl = 0; sl < 2; ++sl) { const float* row = rows.data() + ((size_t)e * 2 + sl) * CRUX_INFO_N;
        if (has[sl] && row[CRUX_INFO_GRAD_NORM] != row[CRUX_INFO_GRAD_NORM]) nan = true;
        if (dst[sl] && has[sl]) memcpy(dst[sl] + (size_t)e * CRUX_INFO_N, row, sizeof(float) * CRUX_INFO_N); } }
    if (nan) return crux_fail(c, CRUX_ENAN, "NaN detected! (grad norm is NaN, src/training.jl:20) in the DDPG / TD3 epochs");
    return CRUX_OK;
  }
  return dpg_epochs_impl(actor, q1, q2, actor_targ, q1_targ, q2_targ, source, batch, gamma, tau, sigma, eps_min, eps_max, a_min, a_max, use_weight, epoch0, n_epochs, critic_every, actor_every,
                         sample_counter0, noise_seed, noise_counter0, infos_critic, infos_actor, nullptr);
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

// test hook: n dependent trivial ops (a 256-float fill each): the executor's per-op overhead (dispatch + barrier)
extern "C" int32_t crux_debug_exec_nops(crux_ctx* c, int32_t n, int32_t blocks) {
  int32_t rc = crux_exec_begin(c); if (rc) return rc;
  float* p = (float*)crux_exec_small(c, 4 * 256 * (size_t)(blocks > 0 ? blocks : 1));
  for (int k = 0; k < n; ++k) crux_exec_push<FillOp, OP_FILL>(c, (unsigned)(blocks > 0 ? blocks : 1), p, (float)k, (int64_t)256 * (blocks > 0 ? blocks : 1));
  return crux_exec_run(c);
}
// test hook: the forward pass of `net` recorded `reps` times (dependent ops): per-op cost of the tile GEMM inside the executor
extern "C" int32_t crux_debug_exec_forward_reps(crux_mlp* net, const float* d_x, int64_t B, int32_t reps) {
  crux_ctx* c = net->ctx; int32_t rc = crux_exec_begin(c); if (rc) return rc;
  for (int k = 0; k < reps; ++k) { rc = crux_dense_forward(net, d_x, B, c->stream); if (rc) { crux_exec_abort(c); return rc; } }
  return crux_exec_run(c);
}
