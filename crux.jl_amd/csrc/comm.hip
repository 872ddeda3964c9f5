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
int32_t crux_allreduce_mean(crux_mlp* net) {
  if (!net) return CRUX_EINVAL;
  crux_mlp* one[1] = {net}; return crux_comm_allreduce_mean_impl(net->ctx, one, 1);
}

}  // extern "C"
