// comm.hip -- multi-GPU exchange for environment-shard replicas: RCCL (= NCCL API on ROCm) over xGMI, one process per GPU.
// replicas own their environments, buffers and RNG streams; the single exchange step of the path is the all-reduce of the
// replicated learner state. Here it is the periodic form north_star names: parameters and Adam moments are AVERAGED every `sync_every`
// epochs, enqueued on the library's stream right behind the learner kernels -- no host synchronisation between epochs.
// RCCL is loaded with dlopen on first use, so single-GPU users (and the CPU-side ABI tests) never need it.
#include "common.h"
#include <dlfcn.h>
#include <rccl/rccl.h>

struct RcclApi {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi* rccl(crux_ctx* c) {
  static RcclApi api; static bool tried = false;
  if (!tried) {
    tried = true;
    // an RCCL already living in the process (e.g. the one a host framework brought) is reused -- one collective runtime per process;
    // otherwise the ROCm installation's. CRUX_RCCL_LIB names a specific file.
    const char* envp = getenv("CRUX_RCCL_LIB");
    if (envp && *envp) api.h = dlopen(envp, RTLD_NOW | RTLD_LOCAL);
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) { if (api.h) break; api.h = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD); }
    for (const char* n : names) { if (api.h) break; api.h = dlopen(n, RTLD_NOW | RTLD_LOCAL); }
    if (api.h) {
      api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.h, "ncclGetUniqueId"); api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.h, "ncclCommInitRank");
      api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.h, "ncclCommDestroy"); api.AllReduce = (decltype(api.AllReduce))dlsym(api.h, "ncclAllReduce");
      api.GroupStart = (decltype(api.GroupStart))dlsym(api.h, "ncclGroupStart"); api.GroupEnd = (decltype(api.GroupEnd))dlsym(api.h, "ncclGroupEnd");
      api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.h, "ncclGetErrorString");
      if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce || !api.GroupStart || !api.GroupEnd) { dlclose(api.h); api.h = nullptr; }
    }
  }
  if (!api.h) { crux_fail(c, CRUX_ERCCL, "RCCL is not available (dlopen librccl.so.1 failed: %s)", dlerror()); return nullptr; }
  return &api;
}
#define RCCLCHK(c, api, expr) do { ncclResult_t r__ = (expr); if (r__ != ncclSuccess) return crux_fail((c), CRUX_ERCCL, "%s failed: %s", #expr, (api)->GetErrorString ? (api)->GetErrorString(r__) : "?"); } while (0)

__global__ void k_scale3(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, float s, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  p[i] *= s; m[i] *= s; v[i] *= s;
}

// stream-ordered: averaged parameters and Adam moments are in place when later work on ctx->stream runs
int32_t crux_comm_allreduce_mean_impl(crux_ctx* c, crux_mlp* const* nets, int n_nets) {
  if (!c->comm) return CRUX_OK;
  RcclApi* api = rccl(c); if (!api) return CRUX_ERCCL;
  RCCLCHK(c, api, api->GroupStart());
  for (int i = 0; i < n_nets; ++i) { const size_t cnt = (size_t)nets[i]->nd.n_params;
    RCCLCHK(c, api, api->AllReduce(nets[i]->p, nets[i]->p, cnt, ncclFloat, ncclSum, (ncclComm_t)c->comm, c->stream));
    RCCLCHK(c, api, api->AllReduce(nets[i]->m, nets[i]->m, cnt, ncclFloat, ncclSum, (ncclComm_t)c->comm, c->stream));
    RCCLCHK(c, api, api->AllReduce(nets[i]->v, nets[i]->v, cnt, ncclFloat, ncclSum, (ncclComm_t)c->comm, c->stream)); }
  RCCLCHK(c, api, api->GroupEnd());
  for (int i = 0; i < n_nets; ++i) { const int64_t cnt = nets[i]->nd.n_params;
    hipLaunchKernelGGL(k_scale3, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, c->stream, nets[i]->p, nets[i]->m, nets[i]->v, 1.0f / (float)c->comm_n, cnt); }
  return crux_launch_check(c, "k_scale3");
}

extern "C" {

int32_t crux_comm_unique_id(crux_ctx* c, uint8_t* id128) {
  if (!c || !id128) return CRUX_EINVAL;
  RcclApi* api = rccl(c); if (!api) return CRUX_ERCCL;
  ncclUniqueId id; RCCLCHK(c, api, api->GetUniqueId(&id));
  static_assert(sizeof(ncclUniqueId) == 128, "unique id size");
  memcpy(id128, &id, 128); return CRUX_OK;
}
int32_t crux_comm_init(crux_ctx* c, int32_t rank, int32_t nranks, const uint8_t* id128) {
  if (!c || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return CRUX_EINVAL;
  if (c->comm) return crux_fail(c, CRUX_EINVAL, "comm_init: this context already has a communicator");
  RcclApi* api = rccl(c); if (!api) return CRUX_ERCCL;
  HIPCHK(c, hipSetDevice(c->device));
  ncclUniqueId id; memcpy(&id, id128, 128); ncclComm_t comm = nullptr;
  RCCLCHK(c, api, api->CommInitRank(&comm, nranks, id, rank));
  c->comm = (void*)comm; c->comm_rank = rank; c->comm_n = nranks; return CRUX_OK;
}
int32_t crux_comm_destroy(crux_ctx* c) {
  if (!c) return CRUX_EINVAL;
  if (c->comm) { RcclApi* api = rccl(c); (void)hipStreamSynchronize(c->stream); if (api) (void)api->CommDestroy((ncclComm_t)c->comm); c->comm = nullptr; c->comm_n = 0; }
  return CRUX_OK;
}
int32_t crux_comm_size(const crux_ctx* c) { return c && c->comm ? c->comm_n : 1; }
// exact data-parallel step (k = 1): crux_loss_grad on the local minibatch -> crux_allreduce_grads (SUM over ranks, stream-ordered) ->
// crux_adam_apply(net, 1/nranks): every rank applies the same update, parameters and Adam state stay replicated
int32_t crux_allreduce_grads(crux_mlp* net) {
  if (!net) return CRUX_EINVAL;
  crux_ctx* c = net->ctx; if (!c->comm) return CRUX_OK;
  RcclApi* api = rccl(c); if (!api) return CRUX_ERCCL;
  RCCLCHK(c, api, api->AllReduce(net->g, net->g, (size_t)net->nd.n_params, ncclFloat, ncclSum, (ncclComm_t)c->comm, c->stream));
  return CRUX_OK;
}
// ---- replica group with direct peer slots ------------------------------------------------------------------------------------------
// The exchange step of the data-parallel path: a SUM all-reduce of the flattened minibatch gradient (C5 actor: 22.8 KB) before
// Adam, every minibatch. At that size a ring/tree collective is pure latency, and the learner is a persistent kernel that cannot return to
// the host between steps, so the all-reduce is done BY the kernel: every rank owns a fine-grained region; rank r writes its local gradient
// into slot [parity][r] of every peer's region over xGMI (one hop, fully connected), raises flag[r] there, waits for the N-1 flags in its own
// region and adds the N contributions in rank order -- every rank forms the same sum bit for bit, so parameters and Adam state stay
// replicated without ever being exchanged. These entries only set the regions up; the protocol itself is in train_mfma_x2.hip.
int32_t crux_make_streams_concurrent(crux_ctx* const* ctxs, int n);   // train.hip
int crux_x2_placement_ok(crux_ctx* c);                                  // train_mfma_x2.hip
extern "C" int32_t crux_peer_probe_launch(crux_ctx* c, int32_t rounds, int32_t first_bound_ms, int32_t round_bound_ms);
extern "C" int32_t crux_peer_probe_collect(crux_ctx* c, int32_t rounds, float* out_us4);
static int32_t peer_region(crux_ctx* c) {
  if (c->peer_local) return CRUX_OK;
  HIPCHK(c, hipSetDevice(c->device));
  void* p = nullptr;
  // The slots are written by peers over xGMI and read by a kernel that is already running: only a fine-grained region is coherent for that. A coarse-grained
  // hipMalloc block would look fine behind one L2 and serve stale slots across devices, so there is no fallback: no fine-grained memory, no replica group.
  const hipError_t ef = hipExtMallocWithFlags(&p, CRUX_PX_BYTES, hipDeviceMallocFinegrained);
  if (ef != hipSuccess) { (void)hipGetLastError(); c->peer_fine = false;
    return crux_fail(c, CRUX_EHIP, "peer region: fine-grained device memory is unavailable (hipExtMallocWithFlags: %s); replica groups need it for in-kernel coherence across devices", hipGetErrorString(ef)); }
  c->peer_fine = true;
  HIPCHK(c, hipMemset(p, 0, CRUX_PX_BYTES));
  HIPCHK(c, hipDeviceSynchronize());
  c->peer_local = p;
  if (!c->peer_host) {      // the host's side of the bounds of peer_wait.h: a pinned, device-mapped block (word 0: abort word; words 16..: probe results)
    if (hipHostMalloc((void**)&c->peer_host, 256, hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); c->peer_host = nullptr; return crux_fail(c, CRUX_ENOMEM, "peer region: pinned abort word"); }
    memset(c->peer_host, 0, 256);
    HIPCHK(c, hipHostGetDevicePointer(&c->peer_host_dev, c->peer_host, 0)); }
  return CRUX_OK;
}
// the words of the region that the kernels' slow path reads (peer_wait.h): where this context's host abort word lives and the per-launch wait budget
static int32_t peer_write_consts(crux_ctx* c) {
  HIPCHK(c, hipSetDevice(c->device));
  for (int w = 0; w < 2; ++w) { float* base = (float*)c->peer_local + (size_t)w * CRUX_PX_STREAM_FLOATS;
    HIPCHK(c, hipMemcpy(base + CRUX_PX_HABORT, &c->peer_host_dev, sizeof(void*), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(base + CRUX_PX_BUDGET, &c->peer_budget_ticks, sizeof(long long), hipMemcpyHostToDevice)); }
  return CRUX_OK;
}
// a new group starts from exchange 0 with clean flags (a previous group may have ended on a timeout); legal because no peer writes here before
// every rank has attached
static int32_t peer_reset(crux_ctx* c) {
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemset(c->peer_local, 0, CRUX_PX_BYTES)); HIPCHK(c, hipDeviceSynchronize());
  c->peer_host[0] = 0u; c->peer_probe_base = 0; c->peer_probe_bad = false;      // a new group: the host abort word and the rendezvous counters start over
  return peer_write_consts(c);
}
static int32_t peer_upload_table(crux_ctx* c) {
  if (!c->peer_tab) { if (hipMalloc(&c->peer_tab, sizeof(float*) * 2 * CRUX_PX_MAXR) != hipSuccess) return crux_fail(c, CRUX_ENOMEM, "peer table"); }
  float* h[2 * CRUX_PX_MAXR] = {};
  for (int w = 0; w < 2; ++w) for (int r = 0; r < c->peer_n; ++r) h[w * CRUX_PX_MAXR + r] = (float*)c->peer_ptr[r] + (size_t)w * CRUX_PX_STREAM_FLOATS;
  HIPCHK(c, hipMemcpy(c->peer_tab, h, sizeof h, hipMemcpyHostToDevice));
  return CRUX_OK;
}
int32_t crux_peer_export(crux_ctx* c, uint8_t* handle64) {
  if (!c || !handle64) return CRUX_EINVAL;
  if (crux_grouped(c)) return crux_fail(c, CRUX_EINVAL, "peer_export: this context is attached to a group (detach first)");
  int32_t rc = peer_region(c); if (rc) return rc;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
  hipIpcMemHandle_t h; HIPCHK(c, hipIpcGetMemHandle(&h, c->peer_local));
  memcpy(handle64, &h, 64); return CRUX_OK;
}
int32_t crux_peer_attach(crux_ctx* c, int32_t rank, int32_t nranks, const uint8_t* handles) {
  if (!c || !handles || nranks < 1 || nranks > CRUX_PX_MAXR || rank < 0 || rank >= nranks) return CRUX_EINVAL;
  if (crux_grouped(c)) return crux_fail(c, CRUX_EINVAL, "peer_attach: already attached");
  int32_t rc = peer_region(c); if (rc) return rc;
  rc = peer_reset(c); if (rc) return rc;
  int ndev = 0; (void)hipGetDeviceCount(&ndev);
  for (int d = 0; d < ndev; ++d) if (d != c->device) { (void)hipDeviceEnablePeerAccess(d, 0); (void)hipGetLastError(); }   // best effort: already enabled / not visible are both fine
  for (int r = 0; r < nranks; ++r) {
    if (r == rank) { c->peer_ptr[r] = c->peer_local; c->peer_ipc[r] = false; continue; }
    hipIpcMemHandle_t h; memcpy(&h, handles + 64 * (size_t)r, 64); void* p = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) { (void)hipGetLastError();      // (reported below; a sticky error would fail the next kernel launch check of this thread)
      for (int q = 0; q < r; ++q) if (c->peer_ipc[q]) { (void)hipIpcCloseMemHandle(c->peer_ptr[q]); c->peer_ipc[q] = false; }
      return crux_fail(c, CRUX_EHIP, "peer_attach: hipIpcOpenMemHandle for rank %d failed: %s", r, hipGetErrorString(e)); }
    c->peer_ptr[r] = p; c->peer_ipc[r] = true;
  }
  (void)crux_x2_placement_ok(c);
  c->peer_rank = rank; c->peer_n = nranks; c->peer_solo = nranks == 1;      // nranks == 1: a group of one (the replica-group kernels with no peer: the bit-exact reference of a group on identical shards)
  return peer_upload_table(c);
}
// the same wiring for contexts of ONE process (a multi-GPU single-process host, or several replicas on one device): ctxs[r] is rank r
int32_t crux_peer_attach_local(crux_ctx* const* ctxs, int32_t n) {
  if (!ctxs || n < 1 || n > CRUX_PX_MAXR) return CRUX_EINVAL;
  for (int r = 0; r < n; ++r) { if (!ctxs[r]) return CRUX_EINVAL; if (ctxs[r]->peer_n > 1) return crux_fail(ctxs[r], CRUX_EINVAL, "peer_attach_local: rank %d is already attached", r);
    int32_t rc = peer_region(ctxs[r]); if (rc) return rc; rc = peer_reset(ctxs[r]); if (rc) return rc; }
  for (int r = 0; r < n; ++r) { crux_ctx* c = ctxs[r];
    HIPCHK(c, hipSetDevice(c->device));
    for (int q = 0; q < n; ++q) { if (ctxs[q]->device != c->device) { (void)hipDeviceEnablePeerAccess(ctxs[q]->device, 0); (void)hipGetLastError(); }
      c->peer_ptr[q] = ctxs[q]->peer_local; c->peer_ipc[q] = false; }
    c->peer_rank = r; c->peer_n = n;
    { bool same = false; for (int q = 0; q < n; ++q) if (q != r && ctxs[q]->device == c->device) same = true;       // device-wide waits (hipFree) would deadlock against the peer's spinning kernel
      if (same && !c->peer_same_device) { c->peer_same_device = true; crux_same_device_group_enter(); } }
    const int32_t rc = peer_upload_table(c); if (rc) return rc; }
  for (int r = 0; r < n; ++r) { HIPCHK(ctxs[r], hipSetDevice(ctxs[r]->device)); (void)crux_x2_placement_ok(ctxs[r]); }
  auto undo = [&](int32_t rc) { for (int r = 0; r < n; ++r) { ctxs[r]->peer_n = 0; ctxs[r]->peer_rank = 0; if (ctxs[r]->peer_same_device) { ctxs[r]->peer_same_device = false; crux_same_device_group_leave(); } } return rc; };
  const int32_t rcs = crux_make_streams_concurrent(ctxs, n);        // replicas sharing a device: every learner stream on its own hardware queue (pairwise handshakes)
  if (rcs) return undo(rcs);
  if (n > 1) {      // and then all 2 n learner streams together, the way a training call uses them: 64 rendezvous rounds (crux_peer_probe above)
    const int rounds = 64; int32_t rc = CRUX_OK; int failed = -1; float worst = 0.f;
    for (int r = 0; r < n && !rc; ++r) rc = crux_peer_probe_launch(ctxs[r], rounds, 50, 20);
    if (rc) { for (int r = 0; r < n; ++r) { (void)hipStreamSynchronize(ctxs[r]->stream); if (ctxs[r]->aux_stream) (void)hipStreamSynchronize(ctxs[r]->aux_stream); } return undo(rc); }
    for (int r = 0; r < n; ++r) { float us[4]; const int32_t rcr = crux_peer_probe_collect(ctxs[r], rounds, us); if (rcr && failed < 0) { failed = r; rc = rcr; }
      for (int q = 1; q < 4; q += 2) if (us[q] > worst) worst = us[q]; }
    if (rc) { if (failed > 0) crux_fail(ctxs[0], rc, "peer_attach_local: replica %d: %s", failed, ctxs[failed]->err.c_str()); return undo(rc); }
    if (worst > 250.f) return undo(crux_fail(ctxs[0], CRUX_EHIP, "peer_attach_local: the replicas' learner kernels do not run side by side on this device: the slowest of %d rendezvous rounds took %.0f us (co-resident kernels: a few us). The device's hardware queues are time-sliced (other processes on the GPU, or more streams than GPU_MAX_HW_QUEUES) -- a group here would train at a scheduling quantum per exchange", rounds, (double)worst));
  }
  return CRUX_OK;
}
int32_t crux_peer_detach(crux_ctx* c) {
  if (!c) return CRUX_EINVAL;
  (void)hipStreamSynchronize(c->stream); if (c->aux_stream) (void)hipStreamSynchronize(c->aux_stream);
  for (int r = 0; r < CRUX_PX_MAXR; ++r) { if (c->peer_ipc[r]) (void)hipIpcCloseMemHandle(c->peer_ptr[r]); c->peer_ipc[r] = false; c->peer_ptr[r] = nullptr; }
  c->peer_n = 0; c->peer_rank = 0; c->peer_solo = false;
  if (c->peer_same_device) { c->peer_same_device = false; crux_same_device_group_leave(); }      // the last member to leave frees the parked blocks
  return CRUX_OK;
}
// flag-wait histogram of the in-kernel exchange (diagnostics of a multi-GPU run): while enabled, lane 0 of each learner workgroup bins the time it waited for the
// slowest peer's flag of every exchange, log2 of 10 ns ticks (bin b: [2^b, 2^(b+1)) x 10 ns). out: uint32 [2 learner streams][2 workgroups][32].
// The periodic form of the replica group (local SGD / "sync_every"): between exchanges every replica takes k - 1 LOCAL Adam steps; after every k-th step the group
// averages theta, m and v inside the persistent learner kernel (train_fs2_kernel.h). k = 1 (default) is the exact form: the gradient is exchanged every minibatch.
int32_t crux_peer_set_sync_every(crux_ctx* c, int32_t k) { if (!c) return CRUX_EINVAL; if (k < 1 || k > 65536) return crux_fail(c, CRUX_EINVAL, "peer_set_sync_every: k = %d", k); c->peer_every = k; return CRUX_OK; }
int32_t crux_peer_sync_every(const crux_ctx* c) { return c ? c->peer_every : 1; }
// how long a learner workgroup waits for a peer's flag of ONE exchange before it gives up: the launch then ends with CRUX_EHIP ("a replica of the group did not answer") and
// raises the abort word of every peer, instead of hanging the GPU behind a rank that died. Default 30 000 ms.
int32_t crux_peer_set_timeout_ms(crux_ctx* c, int32_t ms) { if (!c) return CRUX_EINVAL; if (ms < 1 || ms > 600000) return crux_fail(c, CRUX_EINVAL, "peer_set_timeout_ms: %d ms (1 .. 600 000)", ms);
  c->peer_timeout_ticks = (long long)ms * 100000ll; return CRUX_OK; }
int32_t crux_peer_hist_enable(crux_ctx* c, int32_t on) { if (!c) return CRUX_EINVAL; c->peer_hist = on != 0; return CRUX_OK; }
int32_t crux_peer_wait_hist(crux_ctx* c, uint32_t* out128, int32_t reset) {
  if (!c || !out128) return CRUX_EINVAL;
  if (!c->peer_local) { memset(out128, 0, 128 * sizeof(uint32_t)); return CRUX_OK; }
  (void)hipStreamSynchronize(c->stream); if (c->aux_stream) (void)hipStreamSynchronize(c->aux_stream);
  for (int w = 0; w < 2; ++w) { float* h = (float*)c->peer_local + (size_t)w * CRUX_PX_STREAM_FLOATS + CRUX_PX_HIST;
    HIPCHK(c, hipMemcpy(out128 + 64 * w, h, 64 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (reset) HIPCHK(c, hipMemset(h, 0, 64 * sizeof(uint32_t))); }
  return CRUX_OK;
}
// ---- the bounds of peer_wait.h, host side -----------------------------------------------------------------------------------------------
// what the flag waits of ONE learner launch may add up to before the launch gives up with CRUX_EHIP (bound 2): the per-exchange timeout only catches a peer that is
// absent; replicas whose hardware queues are time-sliced answer every exchange after a scheduling quantum and never trip it. 0 = no budget. Default 60 s.
int32_t crux_peer_set_budget_ms(crux_ctx* c, int32_t ms) { if (!c) return CRUX_EINVAL; if (ms < 0 || ms > 3600000) return crux_fail(c, CRUX_EINVAL, "peer_set_budget_ms: %d ms (0 .. 3 600 000)", ms);
  c->peer_budget_ticks = (long long)ms * 100000ll;
  if (c->peer_local) { (void)hipStreamSynchronize(c->stream); if (c->aux_stream) (void)hipStreamSynchronize(c->aux_stream); return peer_write_consts(c); }
  return CRUX_OK; }
// Calls off the replica-group launches of this context FROM THE HOST, without any GPU work: raises the context's pinned abort word, which the slow path of every flag
// wait polls (bound 4). Safe from any thread, a signal handler or a watchdog while another thread sits in a training call; the launch returns CRUX_EHIP and tells its
// peers. crux_peer_abort_clear (or the next attach) re-arms the context; the GROUP stays ended (the peers' abort words are terminal until they re-attach).
int32_t crux_peer_abort(crux_ctx* c) { if (!c) return CRUX_EINVAL; if (c->peer_host) __atomic_store_n(&c->peer_host[0], 1u, __ATOMIC_RELEASE); return CRUX_OK; }
int32_t crux_peer_abort_clear(crux_ctx* c) { if (!c) return CRUX_EINVAL; if (c->peer_host) __atomic_store_n(&c->peer_host[0], 0u, __ATOMIC_RELEASE); return CRUX_OK; }
const char* crux_peer_why_text(int why) {
  switch (why) { case 1: return "a replica did not answer one exchange within the timeout (crux_peer_set_timeout_ms)";
    case 2: return "the replicas answered, but the waits of this launch exceeded its budget (crux_peer_set_budget_ms): the group is running far below its speed, e.g. replicas sharing a device whose hardware queues are time-sliced";
    case 3: return "a peer raised this rank's abort word (it left the group; crux_peer_abort_reason says why)";
    case 4: return "the host called the launch off (crux_peer_abort)";
    case 5: return "a peer left the group on a NaN step (training.jl:20 on that replica)";
    default: return "a replica of the group did not answer or raised the abort word"; } }
// the abort words of this rank's region, one per learner stream (0 = none; 1 .. 5 as in crux_peer_why_text): why a peer -- or this rank -- ended the group
int32_t crux_peer_abort_reason(crux_ctx* c, int32_t* out2) {
  if (!c || !out2) return CRUX_EINVAL; out2[0] = out2[1] = 0;
  if (!c->peer_local) return CRUX_OK;
  HIPCHK(c, hipSetDevice(c->device));
  for (int w = 0; w < 2; ++w) { unsigned v = 0; HIPCHK(c, hipMemcpy(&v, (float*)c->peer_local + (size_t)w * CRUX_PX_STREAM_FLOATS + CRUX_PX_ABORT, 4, hipMemcpyDeviceToHost)); out2[w] = (int32_t)v; }
  return CRUX_OK;
}

// ---- rendezvous probe ---------------------------------------------------------------------------------------------------------------------
// Are the replicas of this group able to answer each other at the speed the in-kernel exchange assumes? One wave per (rank, learner stream) runs `rounds` rendezvous
// through the regions with the exchange's own primitives (system-scope flag stores into every peer's region, polls of the own region) and reports the wait of the first
// round (launch skew between the ranks' hosts) and the longest later one. Kernels that are co-resident answer in microseconds; streams that share a hardware queue never
// meet (the first kernel waits for one that cannot start); queues that the firmware time-slices (more user queues on the device than hardware slots: several
// processes on one GPU) meet once per scheduling quantum -- milliseconds. This replaces the "two 150 us spins took < 1.6 x one" guess of rounds 2-5.
__global__ void k_px_probe(float* const* __restrict__ tab, int rank, int n, unsigned long long base, int rounds, long long first_bound, long long bound, unsigned long long* __restrict__ out) {
  if (threadIdx.x != 0) return;
  float* const mine = tab[rank];
  const unsigned* const hab = *(const unsigned* const*)(mine + CRUX_PX_HABORT);
  unsigned long long firstw = 0ull, maxw = 0ull, done = 0ull; unsigned why = 0u;
  for (int k = 0; k < rounds && !why; ++k) {
    const unsigned long long want = base + (unsigned long long)k + 1ull;
    for (int r = 0; r < n; ++r) if (r != rank) __hip_atomic_store((unsigned long long*)(tab[r] + CRUX_PX_PROBE) + 8 * rank, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const long long t0 = wall_clock64(), lim = k == 0 ? first_bound : bound;
    for (int r = 0; r < n && !why; ++r) { if (r == rank) continue;
      const unsigned long long* fl = (const unsigned long long*)(mine + CRUX_PX_PROBE) + 8 * r; unsigned spins = 0;
      while (__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want) { __builtin_amdgcn_s_sleep(1);
        if ((++spins & 63u) == 0u) { if (wall_clock64() - t0 > lim) why = 1u; else if (hab && __hip_atomic_load(hab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) why = 4u; if (why) break; } } }
    if (why) break;
    const unsigned long long wt = (unsigned long long)(wall_clock64() - t0);
    if (k == 0) firstw = wt; else if (wt > maxw) maxw = wt;
    ++done;
    const long long p0 = wall_clock64(); while (wall_clock64() - p0 < 300) __builtin_amdgcn_s_sleep(4);      // ~3 us between rounds: the spacing of a learner's exchanges
  }
  __hip_atomic_store(out + 1, firstw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); __hip_atomic_store(out + 2, maxw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(out + 3, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); __hip_atomic_store(out + 0, (unsigned long long)why + 100ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
int32_t crux_ensure_aux_stream(crux_ctx* c);      // train.hip
// the two halves of crux_peer_probe (crux_peer_attach_local drives every rank of a same-process group from one host thread: all launches first, then all results)
int32_t crux_peer_probe_launch(crux_ctx* c, int32_t rounds, int32_t first_bound_ms, int32_t round_bound_ms) {
  if (c->peer_n < 2) return CRUX_OK;
  if (c->peer_probe_bad) return crux_fail(c, CRUX_EINVAL, "peer_probe: an earlier probe of this group failed (the ranks' round counters may differ): detach and attach again");
  HIPCHK(c, hipSetDevice(c->device));
  { const int32_t rca = crux_ensure_aux_stream(c); if (rca) return rca; }
  unsigned long long* out = (unsigned long long*)(c->peer_host + 16); unsigned long long* out_dev = (unsigned long long*)((unsigned*)c->peer_host_dev + 16);
  for (int w = 0; w < 2; ++w) { for (int q = 0; q < 4; ++q) out[4 * w + q] = 0ull;
    hipLaunchKernelGGL(k_px_probe, dim3(1), dim3(64), 0, w ? c->aux_stream : c->stream, (float* const*)(c->peer_tab + w * CRUX_PX_MAXR), c->peer_rank, c->peer_n, c->peer_probe_base, (int)rounds,
                       (long long)first_bound_ms * 100000ll, (long long)round_bound_ms * 100000ll, out_dev + 4 * w); }
  c->peer_probe_base += (unsigned long long)rounds;
  return crux_launch_check(c, "k_px_probe");
}
int32_t crux_peer_probe_collect(crux_ctx* c, int32_t rounds, float* out_us4) {
  if (out_us4) for (int q = 0; q < 4; ++q) out_us4[q] = 0.f;
  if (c->peer_n < 2) return CRUX_OK;
  HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipStreamSynchronize(c->aux_stream));
  const volatile unsigned long long* out = (const volatile unsigned long long*)(c->peer_host + 16);
  int bad = 0; unsigned why = 0; unsigned long long done = 0;
  for (int w = 0; w < 2; ++w) { if (out_us4) { out_us4[2 * w] = (float)out[4 * w + 1] * 0.01f; out_us4[2 * w + 1] = (float)out[4 * w + 2] * 0.01f; }
    if (out[4 * w] != 100ull) { bad |= 1 << w; why = (unsigned)(out[4 * w] >= 100ull ? out[4 * w] - 100ull : 0ull); done = out[4 * w + 3]; } }
  if (bad) { c->peer_probe_bad = true;
    return crux_fail(c, CRUX_EHIP, "peer_probe: the replicas did not meet on learner stream%s (round %llu of %d: %s): their kernels are not running at the same time -- streams sharing a hardware queue, or more user queues on the device than the firmware keeps resident",
                     bad == 3 ? "s 0 and 1" : bad == 1 ? " 0" : " 1", done, rounds, why == 4 ? "called off by the host" : "no answer within the bound"); }
  return CRUX_OK;
}
// COLLECTIVE: every rank of the group calls it at about the same time (after the attach; the first round absorbs up to first_bound_ms of skew between the ranks' hosts).
// out_us4 = {first-round wait, longest later wait} for learner stream 0, then 1, in microseconds. CRUX_EHIP when the replicas did not meet; the caller judges the waits
// (healthy: a few microseconds on one device, ~10 us over xGMI; time-sliced queues: milliseconds).
int32_t crux_peer_probe(crux_ctx* c, int32_t rounds, int32_t first_bound_ms, int32_t round_bound_ms, float* out_us4) {
  if (!c || rounds < 2 || rounds > 100000 || first_bound_ms < 1 || round_bound_ms < 1) return CRUX_EINVAL;
  if (c->peer_n < 2) { if (out_us4) for (int q = 0; q < 4; ++q) out_us4[q] = 0.f; return CRUX_OK; }
  const int32_t rc = crux_peer_probe_launch(c, rounds, first_bound_ms, round_bound_ms); if (rc) return rc;
  return crux_peer_probe_collect(c, rounds, out_us4);
}
int32_t crux_peer_size(const crux_ctx* c) { return c && c->peer_n > 1 ? c->peer_n : 1; }
int32_t crux_peer_rank(const crux_ctx* c) { return c && c->peer_n > 1 ? c->peer_rank : 0; }

int32_t crux_allreduce_mean(crux_mlp* net) {
  if (!net) return CRUX_EINVAL;
  crux_mlp* one[1] = {net}; return crux_comm_allreduce_mean_impl(net->ctx, one, 1);
}

}  // extern "C"
