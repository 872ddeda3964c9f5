// per.hip -- prioritized / uniform replay sampling (src/experience_buffer.jl:303-349). Device scan + search land next;
// until then the entry points report CRUX_EUNSUP loudly rather than falling back to host arithmetic.
#include "common.h"

extern "C" {
int32_t crux_per_sample(crux_buffer* target, crux_buffer* source, int64_t B, const double* rands, float beta, uint64_t i) {
  (void)source; (void)B; (void)rands; (void)beta; (void)i;
  return crux_fail(target ? target->ctx : nullptr, CRUX_EUNSUP, "prioritized_sample!: device kernel not built yet");
}
int32_t crux_uniform_sample(crux_buffer* target, crux_buffer* source, int64_t B, const int64_t* ids, uint64_t i) {
  (void)source; (void)B; (void)ids; (void)i;
  return crux_fail(target ? target->ctx : nullptr, CRUX_EUNSUP, "uniform_sample!: device kernel not built yet");
}
int32_t crux_per_get(crux_buffer* b, float* priorities, float* max_priority, float* min_priority, float* cumsum) {
  if (!b) return CRUX_EINVAL;
  crux_ctx* c = b->ctx;
  if (!b->prioritized) return crux_fail(c, CRUX_EINVAL, "buffer is not prioritized");
  if (cumsum) return crux_fail(c, CRUX_EUNSUP, "cumsum: device scan not built yet");
  float mm[2];
  HIPCHK(c, hipMemcpyAsync(mm, b->pminmax, 8, hipMemcpyDeviceToHost, c->stream));
  if (priorities) HIPCHK(c, hipMemcpyAsync(priorities, b->priorities, 4 * (size_t)b->capacity, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (max_priority) *max_priority = mm[0]; if (min_priority) *min_priority = mm[1];
  return CRUX_OK;
}
}
