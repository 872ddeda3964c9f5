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
  c->peer_local = p; return CRUX_OK;
}
// a new group starts from exchange 0 with clean flags (a previous group may have ended on a timeout); legal because no peer writes here before
// every rank has attached
static int32_t peer_reset(crux_ctx* c) {
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemset(c->peer_local, 0, CRUX_PX_BYTES)); HIPCHK(c, hipDeviceSynchronize()); return CRUX_OK;
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
  const int32_t rcs = crux_make_streams_concurrent(ctxs, n);        // replicas sharing a device: every learner stream on its own hardware queue
  if (rcs) { for (int r = 0; r < n; ++r) { ctxs[r]->peer_n = 0; ctxs[r]->peer_rank = 0; if (ctxs[r]->peer_same_device) { ctxs[r]->peer_same_device = false; crux_same_device_group_leave(); } } return rcs; }
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
// averages theta, m and v inside the persistent learner kernel (train_fs_kernel.h). k = 1 (default) is the exact form: the gradient is exchanged every minibatch.
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
int32_t crux_peer_size(const crux_ctx* c) { return c && c->peer_n > 1 ? c->peer_n : 1; }
int32_t crux_peer_rank(const crux_ctx* c) { return c && c->peer_n > 1 ? c->peer_rank : 0; }

int32_t crux_allreduce_mean(crux_mlp* net) {
  if (!net) return CRUX_EINVAL;
  crux_mlp* one[1] = {net}; return crux_comm_allreduce_mean_impl(net->ctx, one, 1);
}

}  // extern "C"
