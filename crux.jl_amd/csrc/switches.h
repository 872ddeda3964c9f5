// switches.h -- the CRUX_* environment switches (development / test knobs; DESIGN.md lists them). They are read ONCE, on first use (normally when the first context is created; also by
// crux_reload_switches), into one process-wide snapshot that every launch site consults: no getenv on any launch path. A test that changes a switch inside one process
// calls crux_reload_switches() (cruxhip.h) afterwards.
#pragma once
#include <cstdlib>
struct CruxSwitches {
  bool no_fused_epoch, no_chained_epochs, exec_persistent, exec_no_kernarg, force_generic, quiet, verbose, sync_chains, dqp_debug, mfma_timing, small_solve_generic, small_solve_timing;
  bool sac_tile_ops, per_fused_gather, pack_rows, spec_pair, dense_pair, mfma_x2, dqn_persist;      // default on, "0" switches off (dqn_persist: default off, "1" switches on)
  int dense_fused;      // 1 (default) | 0: one Gemm16 launch per layer | 2: the fused pair without the folded output layer
  int fs;               // CRUX_FS (default 1; 0: feature-split learner off)
  bool push_fused;      // CRUX_PUSH_FUSED (default on; "0": push!'s priority bookkeeping of a few rows as its separate launches -- tests compare)
  bool host_zerocopy;   // CRUX_HOST_ZEROCOPY (default on; "0": crux_policy_explore / the small-block push! through staged uploads and read-backs, the round-5 form -- the bench's A/B and a bit-identity test)
  int exec_flags;       // CRUX_EXEC_FLAGS (8: per-op timestamps of the persistent executor)
};
inline CruxSwitches crux_switches_read() {
  auto set = [](const char* k) { return getenv(k) != nullptr; };
  auto not0 = [](const char* k) { const char* e = getenv(k); return !(e && e[0] == '0'); };
  auto num = [](const char* k, int d) { const char* e = getenv(k); return e ? atoi(e) : d; };
  CruxSwitches s{};
  s.no_fused_epoch = set("CRUX_NO_FUSED_EPOCH"); s.no_chained_epochs = set("CRUX_NO_CHAINED_EPOCHS"); s.exec_persistent = set("CRUX_EXEC_PERSISTENT"); s.exec_no_kernarg = set("CRUX_EXEC_NO_KERNARG");
  s.force_generic = set("CRUX_FORCE_GENERIC"); s.quiet = set("CRUX_QUIET"); s.verbose = set("CRUX_VERBOSE"); s.sync_chains = set("CRUX_SYNC_CHAINS"); s.dqp_debug = set("CRUX_DQP_DEBUG");
  s.mfma_timing = set("CRUX_MFMA_TIMING"); s.small_solve_generic = set("CRUX_SMALL_SOLVE_GENERIC"); s.small_solve_timing = set("CRUX_SMALL_SOLVE_TIMING");
  s.sac_tile_ops = not0("CRUX_SAC_TILE_OPS"); s.per_fused_gather = not0("CRUX_PER_FUSED_GATHER"); s.pack_rows = not0("CRUX_PACK_ROWS"); s.spec_pair = not0("CRUX_SPEC_PAIR");
  s.dense_pair = not0("CRUX_DENSE_PAIR"); s.mfma_x2 = not0("CRUX_MFMA_X2"); s.push_fused = not0("CRUX_PUSH_FUSED"); s.host_zerocopy = not0("CRUX_HOST_ZEROCOPY");
  { const char* e = getenv("CRUX_DQN_PERSIST"); s.dqn_persist = e && e[0] == '1'; }
  { const char* e = getenv("CRUX_DENSE_FUSED"); s.dense_fused = !e ? 1 : e[0] == '0' ? 0 : e[0] == '2' ? 2 : 1; }
  s.fs = num("CRUX_FS", 1); s.exec_flags = num("CRUX_EXEC_FLAGS", 0);
  return s;
}
const CruxSwitches& crux_sw();      // context.hip: the snapshot (read on first use, refreshed by crux_reload_switches)
