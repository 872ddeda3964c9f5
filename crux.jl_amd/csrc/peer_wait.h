// peer_wait.h -- the bounded flag wait of the replica-group exchange (comm.hip "peer"), shared by every kernel that takes part in a group
// (train_fs2_kernel.h, train_mfma_kernel.h, train_dense.hip). SURVEY 8(e): the reference is single-process (src/model_free/on_policy.jl:80-109);
// the exchange is this build's addition, so its failure modes are this build's to bound. A wait ends, besides on the flag it waits for, when
//   1  the ONE exchange has waited longer than crux_peer_set_timeout_ms (a peer that is ABSENT);
//   2  the waits of this LAUNCH add up to more than crux_peer_set_budget_ms (a peer that is SLOW: replicas whose hardware queues are time-sliced answer every exchange
//      after a scheduling quantum -- 335 ms measured with four processes on one GPU, profiles/r06_same_device_oversubscription.txt -- and never trip bound 1);
//   3  a peer raised the abort word of this rank's region (it left: failure, NaN step);
//   4  the HOST raised this context's pinned abort word (crux_peer_abort: a watchdog, a launcher tearing a job down) -- the one bound that needs no GPU work to deliver.
// Only the slow path (every 128th poll, ~50-100 us) reads the clock and the abort words; the accounting of bound 2 only runs for waits that reached the slow path,
// so an exchange among healthy replicas costs what it did before.
#pragma once
#include "common.h"

#define CRUX_PX_POLL_MASK 127u
// Inside the waits time is a 32-bit count of 1.28 us units (wall clock >> 7; wraps after 91 min, the longest bound is 60 min): the slow path then needs one register per
// quantity -- the C5 periodic forms of k_train_fs2 sit at 255 of 256 VGPRs, and everything live in this cold path counts against the hot loop's allocation.
#define CRUX_PX_UNIT_SHIFT 7
__device__ __forceinline__ unsigned px_units(long long ticks) { return (unsigned)((unsigned long long)ticks >> CRUX_PX_UNIT_SHIFT); }

// what is left of the launch's wait budget, in units, per workgroup slot p in [0, 8): written at launch begin from the region's budget word (0 = no budget)
__device__ __forceinline__ unsigned* px_left_word(float* mine, int p) { return (unsigned*)(mine + CRUX_PX_WAITED) + p; }
__device__ __forceinline__ void px_launch_begin(float* mine, int p) {
  const long long b = *(const long long*)(mine + CRUX_PX_BUDGET);
  *px_left_word(mine, p) = b > 0 ? px_units(b) + 1u : 0xffffffffu;
}

// slow path of a flag wait: 0 = keep waiting, else the bound that ended it (1 .. 4 as above). One quantity live at a time.
__device__ __forceinline__ unsigned px_give_up(float* mine, int p, long long t0, unsigned timeout_u, bool use_budget) {
  const unsigned waited = px_units(wall_clock64() - t0);
  if (waited > timeout_u) return 1u;
  if (use_budget && waited > *px_left_word(mine, p)) return 2u;
  if (__hip_atomic_load((const unsigned*)(mine + CRUX_PX_ABORT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return 3u;
  const unsigned* const hab = *(const unsigned* const*)(mine + CRUX_PX_HABORT);
  if (hab && __hip_atomic_load(hab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return 4u;
  return 0u;
}

// wait for flag[r] >= want of every peer r. Returns 0 when all arrived, else the bound that ended the wait. `mine`: this rank's region for this learner stream.
__device__ __forceinline__ unsigned px_wait_peers(float* mine, int n, int rank, unsigned long long want, long long t0, long long timeout, int p, bool use_budget) {
  unsigned why = 0u, spins = 0u;
  for (int r = 0; r < n && why == 0u; ++r) { if (r == rank) continue;
    const unsigned long long* fl = (const unsigned long long*)(mine + CRUX_PX_FLAGS) + 8 * r;
    while (__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want) { __builtin_amdgcn_s_sleep(1);
      if ((++spins & CRUX_PX_POLL_MASK) == 0u) { why = px_give_up(mine, p, t0, px_units(timeout), use_budget); if (why) break; } } }
  if (spins > CRUX_PX_POLL_MASK && use_budget) {      // this exchange reached the slow path: charge it to the launch's budget
    const unsigned waited = px_units(wall_clock64() - t0); unsigned* lw = px_left_word(mine, p); const unsigned left = *lw;
    if (left != 0xffffffffu) *lw = left > waited ? left - waited : 0u; }
  return why;
}
// a rank that gives up tells every rank's region (its own included) so that nobody waits for it. A rank that only passes an abort on (bound 3) writes nothing: whoever
// raised the word told every rank, and its reason -- which the host reports (crux_peer_abort_reason) -- must not be overwritten by "passed on".
__device__ __forceinline__ void px_raise_abort(float* const* tab, int n, unsigned why) {
  if (why == 3u) return;
  for (int r = 0; r < n; ++r) __hip_atomic_store((unsigned*)(tab[r] + CRUX_PX_ABORT), why ? why : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
