// ctx.hip -- context lifecycle, error reporting, scratch memory, HIP-event kernel timing.
#include "common.h"
#include <cstdarg>
#include <mutex>

static std::mutex& crux_err_mu() { static std::mutex mu; return mu; }      // writers (crux_fail) and the reader (crux_last_error) of every context's message
int32_t crux_fail(crux_ctx* ctx, int32_t code, const char* fmt, ...) {
  char buf[1024];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  if (ctx) { std::lock_guard<std::mutex> lk(crux_err_mu()); ctx->err = buf; }      // (dense_pair drives two learner chains of one context from two host threads: ADVICE r3)
  return code;
}

int32_t crux_launch_check(crux_ctx* ctx, const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return crux_fail(ctx, CRUX_EHIP, "launch of %s failed: %s", what, hipGetErrorString(e));
  return CRUX_OK;
}

bool crux_exec_recording(const crux_ctx* c);          // exec.hip
void* crux_exec_scratch(crux_ctx* c, size_t bytes);
#undef hipFree
#include <atomic>
#include <mutex>
static std::atomic<int> g_same_device_members{0};      // contexts of this process currently attached to a replica group that shares a device
static std::mutex g_parked_mu; static std::vector<void*> g_parked;
hipError_t crux_hip_free(void* p) {
  if (!p) return hipSuccess;
  if (g_same_device_members.load(std::memory_order_acquire) > 0) { std::lock_guard<std::mutex> lk(g_parked_mu); g_parked.push_back(p); return hipSuccess; }
  return hipFree(p);
}
void crux_same_device_group_enter() { g_same_device_members.fetch_add(1, std::memory_order_acq_rel); }
void crux_same_device_group_leave() {
  if (g_same_device_members.fetch_sub(1, std::memory_order_acq_rel) != 1) return;
  std::vector<void*> v; { std::lock_guard<std::mutex> lk(g_parked_mu); v.swap(g_parked); }
  for (void* p : v) (void)hipFree(p);
}
#define hipFree(p) crux_hip_free((void*)(p))
void crux_sync_before_free(crux_ctx* ctx) { if (g_same_device_members.load(std::memory_order_acquire) == 0) (void)hipStreamSynchronize(ctx->stream); }
void crux_free_device(crux_ctx* ctx, void* p) {
  if (!p) return;
  crux_sync_before_free(ctx);
  (void)hipFree(p);
}
void* crux_scratch(crux_ctx* ctx, size_t bytes) {
  // while a fused sequence is being recorded (exec.hip) the pieces' scratch blocks must not alias: ops of different pieces may share a phase
  if (crux_exec_recording(ctx)) return crux_exec_scratch(ctx, bytes);
  if (bytes <= ctx->scratch_bytes) return ctx->scratch;
  if (ctx->scratch) { crux_free_device(ctx, ctx->scratch); ctx->scratch = nullptr; ctx->scratch_bytes = 0; }
  size_t want = bytes < (1u << 20) ? (1u << 20) : bytes + bytes / 4;
  if (hipMalloc(&ctx->scratch, want) != hipSuccess) { ctx->scratch = nullptr; return nullptr; }
  ctx->scratch_bytes = want;
  return ctx->scratch;
}

void* crux_pinned(crux_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->pinned_bytes) return ctx->pinned;
  if (ctx->pinned) { (void)hipStreamSynchronize(ctx->stream); (void)hipHostFree(ctx->pinned); ctx->pinned = nullptr; ctx->pinned_bytes = 0; }
  size_t want = bytes < (1u << 16) ? (1u << 16) : bytes + bytes / 4;
  if (hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault) != hipSuccess) { ctx->pinned = nullptr; return nullptr; }
  ctx->pinned_bytes = want;
  return ctx->pinned;
}

void* crux_pinned_mapped(crux_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->pinned_mapped_bytes) return ctx->pinned_mapped;
  if (ctx->pinned_mapped) { (void)hipStreamSynchronize(ctx->stream); (void)hipHostFree(ctx->pinned_mapped); ctx->pinned_mapped = nullptr; ctx->pinned_mapped_dev = nullptr; ctx->pinned_mapped_bytes = 0; }
  size_t want = bytes < (1u << 16) ? (1u << 16) : bytes + bytes / 4;
  if (hipHostMalloc(&ctx->pinned_mapped, want, hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); ctx->pinned_mapped = nullptr; return nullptr; }
  if (hipHostGetDevicePointer(&ctx->pinned_mapped_dev, ctx->pinned_mapped, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(ctx->pinned_mapped); ctx->pinned_mapped = nullptr; return nullptr; }
  ctx->pinned_mapped_bytes = want;
  return ctx->pinned_mapped;
}

static std::pair<hipEvent_t, hipEvent_t> take_events(crux_ctx* ctx) {
  if (!ctx->ev_pool.empty()) { auto p = ctx->ev_pool.back(); ctx->ev_pool.pop_back(); return p; }
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b); return {a, b};
}

void crux_prof_begin(crux_ctx* ctx, int slot) {
  if (!ctx->prof_on) return;
  auto ev = take_events(ctx);
  (void)hipEventRecord(ev.first, ctx->stream);
  ctx->pending.push_back({slot, ev});
}
void crux_prof_end(crux_ctx* ctx, int slot) {
  if (!ctx->prof_on) return;
  for (size_t i = ctx->pending.size(); i-- > 0;)
    if (ctx->pending[i].first == slot) { (void)hipEventRecord(ctx->pending[i].second.second, ctx->stream); break; }
}
static void prof_resolve(crux_ctx* ctx) {
  (void)hipStreamSynchronize(ctx->stream);
  for (auto& p : ctx->pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.second.first, p.second.second) == hipSuccess) { ctx->prof_ms[p.first] += ms; ctx->prof_n[p.first] += 1; }
    ctx->ev_pool.push_back(p.second);
  }
  ctx->pending.clear();
}

// the CRUX_* switches: one snapshot per process (switches.h)
// One process-wide snapshot, published through an atomic pointer: read from the environment on FIRST use and by crux_reload_switches() only -- creating another context
// does not change the switches under contexts that are already running. Superseded snapshots are kept (a launch path may still hold a reference; a few hundred bytes each).
static std::atomic<const CruxSwitches*> g_sw{nullptr};
const CruxSwitches& crux_sw() {
  const CruxSwitches* s = g_sw.load(std::memory_order_acquire);
  if (!s) { const CruxSwitches* fresh = new CruxSwitches(crux_switches_read()); const CruxSwitches* expect = nullptr;
    if (g_sw.compare_exchange_strong(expect, fresh, std::memory_order_acq_rel)) s = fresh; else { delete fresh; s = expect; } }
  return *s;
}

// every live context of the process (crux_abort_all: a watchdog that does not know which contexts a stuck call involves)
static std::mutex g_ctx_mu; static std::vector<crux_ctx*> g_ctxs;

extern "C" {

// raises the host abort word of EVERY live context (crux_peer_abort): replica-group launches of this process that are waiting for a peer return CRUX_EHIP within a
// slow-path poll (~100 us). No GPU work, no lock a training call holds: callable from a watchdog thread while the main thread sits in a training call.
int32_t crux_abort_all(void) {
  std::lock_guard<std::mutex> lk(g_ctx_mu); int n = 0;
  for (crux_ctx* c : g_ctxs) if (c->peer_host) { __atomic_store_n(&c->peer_host[0], 1u, __ATOMIC_RELEASE); ++n; }
  return n;
}

int32_t crux_reload_switches(void) { g_sw.store(new CruxSwitches(crux_switches_read()), std::memory_order_release); return CRUX_OK; }

const char* crux_version(void) { return "cruxhip 0.1 (gfx950)"; }

int32_t crux_ctx_create(int32_t device_id, void* stream, crux_ctx** out) {
  if (!out) return CRUX_EINVAL;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) return CRUX_EHIP;
  if (hipSetDevice(device_id) != hipSuccess) return CRUX_EHIP;
  (void)crux_sw();             // the environment switches are read on first use (normally here, by the process's first context), never on a launch path; crux_reload_switches() re-reads
  crux_ctx* c = new crux_ctx();
  c->device = device_id;
  if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
  else { if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return CRUX_EHIP; } c->own_stream = true; }
  { std::lock_guard<std::mutex> lk(g_ctx_mu); g_ctxs.push_back(c); }
  *out = c;
  return CRUX_OK;
}

int32_t crux_ctx_set_learner_cus(crux_ctx* c, int32_t cus) {
  if (!c || cus < 0 || cus > 2) return CRUX_EINVAL;
  c->learner_cus = cus; return CRUX_OK;
}
int32_t crux_ctx_destroy(crux_ctx* c) {
  if (!c) return CRUX_OK;
  { std::lock_guard<std::mutex> lk(g_ctx_mu); for (size_t i = 0; i < g_ctxs.size(); ++i) if (g_ctxs[i] == c) { g_ctxs.erase(g_ctxs.begin() + (long)i); break; } }
  (void)hipStreamSynchronize(c->stream);
  for (auto& p : c->pending) { (void)hipEventDestroy(p.second.first); (void)hipEventDestroy(p.second.second); }
  for (auto& p : c->ev_pool) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
  if (c->scratch) (void)hipFree(c->scratch);
  for (int k = 0; k < 2; ++k) { if (c->xbuf[k]) (void)hipFree(c->xbuf[k]); if (c->xmulti[k]) (void)hipFree(c->xmulti[k]); if (c->amulti[k]) (void)hipFree(c->amulti[k]); }
  if (c->pinned) (void)hipHostFree(c->pinned);
  if (c->pinned_mapped) (void)hipHostFree(c->pinned_mapped);
  for (int r = 0; r < 8; ++r) if (c->peer_ipc[r]) (void)hipIpcCloseMemHandle(c->peer_ptr[r]);
  if (c->peer_local) (void)hipFree(c->peer_local);
  if (c->peer_tab) (void)hipFree(c->peer_tab);
  if (c->lag_dev) (void)hipFree(c->lag_dev);
  if (c->peer_same_device) { c->peer_same_device = false; crux_same_device_group_leave(); }
  if (c->dense_tmp) (void)hipFree(c->dense_tmp);
  if (c->dense_tmp2) (void)hipFree(c->dense_tmp2);
  if (c->dense_pinned2) (void)hipHostFree(c->dense_pinned2);
  if (c->epoch_tmp) (void)hipFree(c->epoch_tmp);
  if (c->epoch_rows) (void)hipFree(c->epoch_rows);
  if (c->spec_abort) (void)hipHostFree(c->spec_abort);
  if (c->peer_host) (void)hipHostFree(c->peer_host);
  crux_exec_destroy(c);
  for (int k = 0; k < c->aux_n_rejected; ++k) (void)hipStreamDestroy(c->aux_rejected[k]);
  if (c->aux_stream) { (void)hipStreamSynchronize(c->aux_stream); (void)hipStreamDestroy(c->aux_stream); (void)hipEventDestroy(c->aux_ev0); (void)hipEventDestroy(c->aux_ev1); }
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return CRUX_OK;
}

const char* crux_last_error(crux_ctx* c) {      // a copy taken under the writers' lock, valid until this thread's next call
  if (!c) return "no context";
  static thread_local std::string copy; { std::lock_guard<std::mutex> lk(crux_err_mu()); copy = c->err; } return copy.c_str();
}

int32_t crux_sync(crux_ctx* c) { if (!c) return CRUX_EINVAL; HIPCHK(c, hipStreamSynchronize(c->stream)); return CRUX_OK; }

int32_t crux_device_alloc(crux_ctx* c, int64_t bytes, void** out) {
  if (!c || !out || bytes < 0) return CRUX_EINVAL;
  *out = nullptr; if (bytes == 0) return CRUX_OK;
  if (hipMalloc(out, (size_t)bytes) != hipSuccess) return crux_fail(c, CRUX_ENOMEM, "device_alloc(%lld) failed", (long long)bytes);
  return CRUX_OK;
}
int32_t crux_device_free(crux_ctx* c, void* p) { if (!c) return CRUX_EINVAL; if (p) { crux_sync_before_free(c); (void)hipFree(p); } return CRUX_OK; }
int32_t crux_memcpy_h2d(crux_ctx* c, void* d_dst, const void* src, int64_t bytes) {
  if (!c || bytes < 0) return CRUX_EINVAL; if (bytes == 0) return CRUX_OK;
  HIPCHK(c, hipMemcpyAsync(d_dst, src, (size_t)bytes, hipMemcpyHostToDevice, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream)); return CRUX_OK;
}
int32_t crux_memcpy_d2h(crux_ctx* c, void* dst, const void* d_src, int64_t bytes) {
  if (!c || bytes < 0) return CRUX_EINVAL; if (bytes == 0) return CRUX_OK;
  HIPCHK(c, hipMemcpyAsync(dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream)); return CRUX_OK;
}

int32_t crux_prof_enable(crux_ctx* c, int32_t on) { if (!c) return CRUX_EINVAL; if (!on) prof_resolve(c); c->prof_on = on != 0; return CRUX_OK; }
int32_t crux_prof_reset(crux_ctx* c) {
  if (!c) return CRUX_EINVAL; prof_resolve(c);
  for (int i = 0; i < CRUX_PROF_NSLOTS; ++i) { c->prof_ms[i] = 0; c->prof_n[i] = 0; }
  return CRUX_OK;
}
int32_t crux_prof_get(crux_ctx* c, int32_t slot, double* ms_total, int64_t* launches) {
  if (!c || slot < 0 || slot >= CRUX_PROF_NSLOTS) return CRUX_EINVAL;
  prof_resolve(c);
  if (ms_total) *ms_total = c->prof_ms[slot];
  if (launches) *launches = c->prof_n[slot];
  return CRUX_OK;
}

}  // extern "C"
