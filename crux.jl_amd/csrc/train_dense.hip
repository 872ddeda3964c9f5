// train_dense.hip -- batch_train! / train! (src/training.jl:13-55) for networks OUTSIDE the register-resident IN-64-64-OUT family, on the MFMA dense engine.
//
// The persistent learner kernels keep a whole 64-wide network in one or two compute units; everything else used to fall to the generic single-workgroup
// learner (scalar loops, ~0.5 ms per minibatch at 64 wide) or, beyond its LDS budget (hidden widths of 256), was refused. Here a minibatch step is the chain
//   gather the minibatch's observations through the composed shuffle order -> Chain forward (tile GEMMs, dense.hip) -> loss head (ppo_loss / a2c_loss /
//   critic mse: d(loss)/d(output), statistics, logSigma gradient) -> pullback (tile GEMMs) -> norm(grad), NaN check -> Flux.update! (gated Adam)
// of ~13 stream-ordered launches, driven from the host over epochs x minibatches like the reference's loop. The host synchronises once per epoch (the
// epoch's info row, training.jl:48), or once per minibatch when KL early stopping or max_batches need a decision (training.jl:45-46). NaN: the gated Adam
// leaves the parameters alone and every later step of the launch finds the status set and does the same (training.jl:20: error, no update).
// Compiled inside offpolicy_unit.hip (uses the heads' helpers of sac.hip).
#include "train_args.h"
#include "peer_wait.h"
#ifndef EPS32F
#define EPS32F 1.1920928955078125e-07f
#endif
#define CRUX_MAXEXTRA 64      // logSigma entries a Gaussian head may carry here (act_dim <= 64, as in check_sac)

__global__ void k_gather_obs(const float* __restrict__ S, int od, const int32_t* __restrict__ rows, int64_t nb, float* __restrict__ x) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i >= nb * od) return;
  const int64_t s = i / od; const int k = (int)(i - s * od);
  x[i] = S[(int64_t)rows[s] * od + k];
}

struct PgHeadArgs {
  const float* z; int nout; const int32_t* rows; int64_t nb;
  const void* A; int ad; const float* LP; const float* ADV; const float* RET;
  int loss, head; float lo, hi, lambda_p, lambda_e, squash;
  const float* ls;       // logSigma (p + xoff) for the Gaussian head
  float* dy;             // [nout x nb]
  float* gx;             // g + xoff: d(loss)/d(logSigma)
  double* stats;         // [8]: sum of the clipped surrogate terms, entropy, kl, advantage, return, clip count, squared error, lagrange cost terms
  const crux_lagrange* lag; const float* CADV;      // lagrange_ppo_loss (ppo.jl:70-131): penalty of THIS minibatch (written by k_lagrange_pid) and D[:cost_advantage]; NULL = plain ppo_loss
};
// the penalty update inside lagrange_ppo_loss (ppo.jl:80-116), once per evaluation of the loss: one block sums the minibatch's :cost and :episode_end and
// thread 0 advances the controller -- the arithmetic of the generic learner's in-kernel version (train_generic.h) and of the oracle
// Under a replica group (px_tab != NULL, N > 1) the minibatch is the GLOBAL one (ppo.jl:84-88 sums over the whole D): the two Float64 sums of every rank travel bit for bit
// (four 32-bit words) through one exchange of the peer slots -- the protocol of k_px_allreduce_flat below -- and every rank adds them in rank order, so all replicas advance
// identical controllers.
__global__ __launch_bounds__(256) void k_lagrange_pid(crux_lagrange* __restrict__ lgp, const float* __restrict__ COST, const uint8_t* __restrict__ EE, const int32_t* __restrict__ rows, int64_t nb,
                                                      float* const* __restrict__ px_tab, int rank, int N, int32_t* __restrict__ status, long long tmo) {
  __shared__ double red[4];
  double sc_ = 0.0, ne_ = 0.0;
  for (int64_t i = threadIdx.x; i < nb; i += 256) { const int64_t row = rows[i]; sc_ += (double)COST[row]; ne_ += EE[row] ? 1.0 : 0.0; }
  double t_sc = block_sum256(sc_, red), t_ne = block_sum256(ne_, red);
  if (threadIdx.x != 0) return;
  if (px_tab && N > 1) {
    float* const mine = px_tab[rank];
    const unsigned long long xg = *(const unsigned long long*)(mine + CRUX_PX_COUNT); const int par = (int)(xg & 1ull);
    const unsigned long long w0 = __builtin_bit_cast(unsigned long long, t_sc), w1 = __builtin_bit_cast(unsigned long long, t_ne);
    for (int r = 0; r < N; ++r) { if (r == rank) continue;
      unsigned long long* dst = (unsigned long long*)(px_tab[r] + (size_t)(par * CRUX_PX_MAXR + rank) * CRUX_PX_SLOT);
      __hip_atomic_store(dst, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); __hip_atomic_store(dst + 1, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    for (int r = 0; r < N; ++r) if (r != rank) __hip_atomic_store((unsigned long long*)(px_tab[r] + CRUX_PX_FLAGS) + 8 * rank, xg + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const long long t0 = wall_clock64();
    const unsigned gave_up = px_wait_peers(mine, N, rank, xg + 1ull, t0, tmo, 0, false);      // (one exchange per launch: the per-launch budget does not apply, the other bounds of peer_wait.h do)
    if (gave_up) { px_raise_abort(px_tab, N, gave_up); status[0] = CRUX_EHIP; status[1] = 16 + (int)gave_up; return; }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    double g_sc = 0.0, g_ne = 0.0;
    for (int r = 0; r < N; ++r) { double a_ = t_sc, b_ = t_ne;
      if (r != rank) { const unsigned long long* src = (const unsigned long long*)(mine + (size_t)(par * CRUX_PX_MAXR + r) * CRUX_PX_SLOT);
        a_ = __builtin_bit_cast(double, __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)); b_ = __builtin_bit_cast(double, __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)); }
      g_sc = r == 0 ? a_ : g_sc + a_; g_ne = r == 0 ? b_ : g_ne + b_; }
    t_sc = g_sc; t_ne = g_ne;
    *(unsigned long long*)(mine + CRUX_PX_COUNT) = xg + 1ull;
  }
  crux_lagrange lg = *lgp;
  const float Jc = (float)t_sc / (float)t_ne;
  const float dl = Jc - lg.target_cost;
  { const float x = lg.I + lg.Ki * dl; lg.I = x > lg.Ki_max ? lg.Ki_max : (x < 0.f ? 0.f : x); }
  lg.smooth_delta = (float)(lg.ema_alpha * (double)lg.smooth_delta + (1.0 - lg.ema_alpha) * (double)dl);
  lg.smooth_Jc = (float)(lg.ema_alpha * (double)lg.smooth_Jc + (1.0 - lg.ema_alpha) * (double)Jc);
  { const float x = lg.smooth_Jc - lg.Jc_prev; lg.deriv_term = (x != x) ? x : (x > 0.f ? x : 0.f); }
  lg.Jc_prev = lg.smooth_Jc;
  { const float x = (lg.Kp * lg.smooth_delta + lg.I) + lg.Kd * lg.deriv_term; lg.penalty = x > lg.penalty_max ? lg.penalty_max : (x < 0.f ? 0.f : x); }
  lg.cur_cost = Jc;
  *lgp = lg;
}
// ppo_loss (ppo.jl:4-21) / a2c_loss (a2c.jl:4-15) / Flux.mse(value, return) (ppo.jl:60): one thread per sample, the arithmetic of the generic learner's head
__global__ __launch_bounds__(256) void k_pg_head(PgHeadArgs q) {
  __shared__ double red[4];
  const int tid = threadIdx.x; const float invB = 1.0f / (float)q.nb; const bool a2c = q.loss == CRUX_LOSS_A2C;
  double s_lossp = 0, s_H = 0, s_kl = 0, s_adv = 0, s_ret = 0, s_clip = 0, s_sq = 0, s_cost = 0;
  const bool lagr = q.lag != nullptr; const float pen = lagr ? q.lag->penalty : 0.f;
  double exg[CRUX_MAXEXTRA];
  const bool gauss = CRUX_IS_PG(q.loss) && q.head == CRUX_HEAD_GAUSSIAN;
  if (gauss) for (int k = 0; k < q.ad; ++k) exg[k] = 0.0;
  for (int64_t s = tid; s < q.nb; s += 256) {
    const int64_t row = q.rows[s]; const float* z = q.z + s * q.nout; float* dy = q.dy + s * q.nout;
    if (q.loss == CRUX_LOSS_VALUE_MSE) { const float R = q.RET[row]; const float d = z[0] - R; s_sq += (double)(d * d); s_ret += (double)R; dy[0] = 2.f * d * invB; continue; }
    if (q.loss == CRUX_LOSS_MSE_ACTION) {                                            // Flux.mse(action(pi, s), a) (il/bc.jl:1): mean over act_dim x batch
      const float* av = (const float*)q.A + row * q.ad; const float inv = invB / (float)q.nout;
      for (int k = 0; k < q.nout; ++k) { const float d = z[k] - av[k]; s_sq += (double)(d * d) / (double)q.nout; dy[k] = 2.f * d * inv; }
      continue; }
    const float A = q.ADV[row], oldlp = q.LP[row]; float newlp = 0.f, H = 0.f, r, g;
    if (q.head == CRUX_HEAD_CATEGORICAL) {
      const uint8_t* av = (const uint8_t*)q.A + row * q.ad;
      float mx = z[0]; for (int k = 1; k < q.nout; ++k) mx = z[k] > mx ? z[k] : mx;
      float sum = 0.f; for (int k = 0; k < q.nout; ++k) sum += expf(z[k] - mx);
      float qq = 0.f, hp = 0.f;
      for (int k = 0; k < q.nout; ++k) { const float pk = expf(z[k] - mx) / sum; qq += pk * (av[k] ? 1.f : 0.f);
        const float lg = logf(pk + EPS32F); H -= pk * lg; hp += (-lg - pk / (pk + EPS32F)) * pk; }
      newlp = logf(qq);
      r = expf(newlp - oldlp); const float u = r * A, rc = fminf(fmaxf(r, q.lo), q.hi), cl = rc * A;
      g = (u <= cl) ? A : 0.f; s_lossp += (double)(a2c ? newlp * A : (u <= cl ? u : cl));
      if (a2c) { g = A; r = 1.f; }
      float gcr = 0.f;
      if (lagr) { const float Ac = q.CADV[row]; const float uc = r * Ac, clc = rc * Ac; s_cost += (double)(uc >= clc ? uc : clc); gcr = (uc >= clc ? Ac : 0.f) * r; }
      for (int k = 0; k < q.nout; ++k) { const float pk = expf(z[k] - mx) / sum; const float lg = logf(pk + EPS32F); const float hk = -lg - pk / (pk + EPS32F);
        const float dlogpi = pk * ((av[k] ? 1.f : 0.f) / qq) - pk;
        const float base = -q.lambda_p * g * r * dlogpi - q.lambda_e * (pk * (hk - hp));
        dy[k] = lagr ? invB * ((base + pen * gcr * dlogpi) / (1.f + pen)) : invB * base; }
    } else {                                                                        // GaussianPolicy / SquashedGaussianPolicy (policies.jl:333-348,374-396)
      const float* av = (const float*)q.A + row * q.ad; const float sq = q.squash;
      for (int k = 0; k < q.ad; ++k) { const float sg = expf(sq > 0.f ? sq_clampls(q.ls[k]) : q.ls[k]); const float uk = sq > 0.f ? sq_untanh(av[k], sq) : av[k]; const float d = uk - z[k];
        newlp += (-(d * d) / (2.f * sg * sg) - 0.9189385332046727f - q.ls[k]); if (sq > 0.f) newlp -= sq_corr(uk); }
      r = expf(newlp - oldlp); const float u = r * A, rc = fminf(fmaxf(r, q.lo), q.hi), cl = rc * A;
      g = (u <= cl) ? A : 0.f; s_lossp += (double)(a2c ? newlp * A : (u <= cl ? u : cl));
      if (a2c) { g = A; r = 1.f; }
      float cf = -q.lambda_p * g * r;
      if (lagr) { const float Ac = q.CADV[row]; const float uc = r * Ac, clc = rc * Ac; s_cost += (double)(uc >= clc ? uc : clc); cf = (cf + pen * ((uc >= clc ? Ac : 0.f) * r)) / (1.f + pen); }
      for (int k = 0; k < q.ad; ++k) { const float sg = expf(sq > 0.f ? sq_clampls(q.ls[k]) : q.ls[k]); const float s2 = sg * sg; const float uk = sq > 0.f ? sq_untanh(av[k], sq) : av[k]; const float d = uk - z[k];
        const float inr = (sq > 0.f && !(q.ls[k] >= -5.f && q.ls[k] <= 2.f)) ? 0.f : 1.f;
        dy[k] = invB * (cf * (d / s2));
        exg[k] += (double)(invB * (cf * (((d * d) / s2) * inr - 1.f))); }
    }
    s_H += (double)H; s_kl += (double)(oldlp - newlp); s_adv += (double)A; if (q.RET) s_ret += (double)q.RET[row];
    if (!a2c && (r > q.hi || r < q.lo)) s_clip += 1.0;
  }
  const double t0 = block_sum256(s_lossp, red), t1 = block_sum256(s_H, red), t2 = block_sum256(s_kl, red), t3 = block_sum256(s_adv, red);
  const double t4 = block_sum256(s_ret, red), t5 = block_sum256(s_clip, red), t6 = block_sum256(s_sq, red), t7 = block_sum256(s_cost, red);
  if (tid == 0) { q.stats[0] = t0; q.stats[1] = t1; q.stats[2] = t2; q.stats[3] = t3; q.stats[4] = t4; q.stats[5] = t5; q.stats[6] = t6; q.stats[7] = t7; }
  if (gauss) for (int k = 0; k < q.ad; ++k) { const double t = block_sum256(exg[k], red); if (tid == 0) q.gx[k] = (float)t + (lagr ? -q.lambda_e / (1.f + pen) : -q.lambda_e); }     // + d(-lambda_e * H)/dlogSigma, H = const + sum(logSigma)
}
__global__ void k_pg_info(const double* __restrict__ st, const double* __restrict__ ssq, int64_t nb, int loss, int head, float lambda_p, float lambda_e, const float* __restrict__ ls, int ad,
                          float* __restrict__ dinfo, const crux_lagrange* __restrict__ lag) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  ssq_finalize(ssq);
  for (int k = 0; k < CRUX_INFO_N; ++k) dinfo[k] = 0.f;
  if (CRUX_IS_PG(loss)) {
    const float p_loss = (float)(-(st[0] / (double)nb)); float entropy, e_loss;
    if (head == CRUX_HEAD_CATEGORICAL) { entropy = (float)(st[1] / (double)nb); e_loss = -entropy; }
    else { float Hs = 1.4189385332046727f; for (int k = 0; k < ad; ++k) Hs += ls[k]; entropy = Hs; e_loss = -Hs; }
    dinfo[CRUX_INFO_LOSS] = lambda_p * p_loss + lambda_e * e_loss; dinfo[CRUX_INFO_ENTROPY] = entropy; dinfo[CRUX_INFO_KL] = (float)(st[2] / (double)nb);
    dinfo[CRUX_INFO_CLIP_FRACTION] = (float)st[5] / (float)nb; dinfo[CRUX_INFO_AVG_ADVANTAGE] = (float)(st[3] / (double)nb); dinfo[CRUX_INFO_AVG_RETURN] = (float)(st[4] / (double)nb);
    if (lag) { const float pen = lag->penalty; const float cost_loss = pen * (float)(st[7] / (double)nb);                    // ppo.jl:119,131
      dinfo[CRUX_INFO_LOSS] = ((lambda_p * p_loss + lambda_e * e_loss) + cost_loss) / (1.f + pen);
      dinfo[CRUX_INFO_PENALTY] = pen; dinfo[CRUX_INFO_CUR_COST] = lag->cur_cost; dinfo[CRUX_INFO_COST_LOSS] = cost_loss; dinfo[CRUX_INFO_P_LOSS] = lambda_p * p_loss; }
  } else dinfo[CRUX_INFO_LOSS] = (float)(st[6] / (double)nb);
  dinfo[CRUX_INFO_GRAD_NORM] = (float)sqrt(ssq[0]);
}

// Replica group (comm.hip "peer") for the dense-engine learner: the SUM all-reduce of the flat minibatch gradient (and of the head's eight statistics sums) between the pullback
// (training.jl:18) and Flux.update! (:21) as ONE launch of one workgroup, in chunks of a peer slot: chunk k is exchange number x0 + k of this learner stream -- written into
// slot [parity][my rank] of every peer, system-scope release, flag, wait for the N - 1 flags in the own region, acquire, add the N contributions in rank order (the own one from
// the gradient buffer itself) and scale by 1/N: the protocol of the persistent kernels (train_mfma_kernel.h), driven from a stand-alone kernel. Every rank forms the same sums bit
// for bit, so parameters, Adam state and the host's early-stopping decisions stay replicated.
#define PXF_CHUNK (CRUX_PX_SLOT - 16)
__global__ __launch_bounds__(1024) void k_px_allreduce_flat(float* __restrict__ g, int64_t n, double* __restrict__ st, float* const* __restrict__ px_tab, int rank, int N, int32_t* __restrict__ status, long long tmo) {
  __shared__ int ok_s;
  const int tid = threadIdx.x;
  float* const mine = px_tab[rank];
  const unsigned long long x0 = *(const unsigned long long*)(mine + CRUX_PX_COUNT);
  const float inv = 1.0f / (float)N;
  const int nchunks = (int)((n + PXF_CHUNK - 1) / PXF_CHUNK);
  for (int k = 0; k < nchunks; ++k) {
    const unsigned long long xg = x0 + (unsigned long long)k; const int par = (int)(xg & 1ull);
    const int64_t off = (int64_t)k * PXF_CHUNK; const int len = (int)((n - off) < PXF_CHUNK ? (n - off) : PXF_CHUNK);
    for (int r = 0; r < N; ++r) { if (r == rank) continue;
      float* dst = px_tab[r] + (size_t)(par * CRUX_PX_MAXR + rank) * CRUX_PX_SLOT;
      for (int i = tid; i < len; i += 1024) dst[i] = g[off + i];
      if (k == 0 && tid < 8) dst[PXF_CHUNK + tid] = (float)st[tid]; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave drains its slot stores, then ONE lane issues the system-scope release and drains it before the flags
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      for (int r = 0; r < N; ++r) if (r != rank) __hip_atomic_store((unsigned long long*)(px_tab[r] + CRUX_PX_FLAGS) + 8 * rank, xg + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const long long t0 = wall_clock64();
      const unsigned gave_up = px_wait_peers(mine, N, rank, xg + 1ull, t0, tmo, 0, false); const bool ok = gave_up == 0u;
      if (!ok) { px_raise_abort(px_tab, N, gave_up); status[1] = 16 + (int)gave_up; }
      ok_s = ok ? 1 : 0;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");      // one lane: drops the L1 (the slot loads are system-scope atomic loads and pass it anyway)
    }
    __syncthreads();
    if (!ok_s) { if (tid == 0) status[0] = CRUX_EHIP; return; }
    for (int i = tid; i < len; i += 1024) { float acc = 0.f;
      for (int r = 0; r < N; ++r) { const float v = r == rank ? g[off + i] : __hip_atomic_load(mine + (size_t)(par * CRUX_PX_MAXR + r) * CRUX_PX_SLOT + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        acc = r == 0 ? v : acc + v; }
      g[off + i] = acc * inv; }
    if (k == 0 && tid < 8) { float acc = 0.f;
      for (int r = 0; r < N; ++r) { const float v = r == rank ? (float)st[tid] : __hip_atomic_load(mine + (size_t)(par * CRUX_PX_MAXR + r) * CRUX_PX_SLOT + PXF_CHUNK + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        acc = r == 0 ? v : acc + v; }
      st[tid] = (double)(acc * inv); }
    __syncthreads();
  }
  if (tid == 0) *(unsigned long long*)(mine + CRUX_PX_COUNT) = x0 + (unsigned long long)nchunks;
}

__global__ void k_nan_status(const double* __restrict__ ssq, int32_t* __restrict__ status) { if (isnan(ssq[0])) status[0] = CRUX_ENAN; }   // gradient-only calls: the NaN check of training.jl:20 without the update

// which learners take this path (called by launch_train after the MFMA family declined)
bool crux_train_dense_eligible(const TrainArgs& a, size_t generic_lds, bool force_generic) {
  if (!a.host_net || !(a.ids || a.ord_all)) return false;
  if (!(CRUX_IS_PG(a.loss) || a.loss == CRUX_LOSS_VALUE_MSE || (a.loss == CRUX_LOSS_MSE_ACTION && a.squash == 0.f))) return false;
  if (CRUX_IS_PG(a.loss) && a.head != CRUX_HEAD_CATEGORICAL && a.head != CRUX_HEAD_GAUSSIAN) return false;
  if (a.nd.L < 1 || a.nd.n_extra > CRUX_MAXEXTRA) return false;
  if (generic_lds > 160 * 1024 - 64) return true;                                   // the generic learner cannot hold it at all
  return !force_generic && a.nd.n_params >= 512 && a.bs >= 32 && !a.ids;            // single steps and tiny networks (the README's 2-8-4) stay on the one-launch generic learner
}

// `strm` / `which`: the stream the chain runs on and the resource set it uses (0: the context's main stream; 1: the second learner stream -- policy_gradient_training runs the
// critic's chain there, from a second host thread, beside the actor's: both chains are launch-bound, two queues interleave them on the device).
int32_t crux_train_dense_run(crux_ctx* c, TrainArgs& a, hipStream_t strm, int which) {
  if (!strm) strm = c->stream;
  void*& dtmp = which ? c->dense_tmp2 : c->dense_tmp; size_t& dtmp_bytes = which ? c->dense_tmp2_bytes : c->dense_tmp_bytes;
  crux_mlp* net = (crux_mlp*)a.host_net; const NetDesc& nd = net->nd; const int nout = nd.dims[nd.L], od = nd.dims[0];
  const int64_t total_rows = a.ids ? a.n_ids : a.len; const int n_epochs = a.ids ? 1 : a.epochs;
  const int64_t bmax = total_rows < a.bs ? total_rows : a.bs;
  // a block of the context that nothing else carves (the caller's status / info rows live in the scratch block)
  { const size_t need = 4 * (size_t)bmax * (size_t)(od + nout) + 256 * 8 + 4096;
    if (dtmp_bytes < need) { if (dtmp) { HIPCHK(c, hipStreamSynchronize(strm)); (void)hipFree(dtmp); dtmp = nullptr; dtmp_bytes = 0; }
      if (hipMalloc(&dtmp, 2 * need) != hipSuccess) return crux_fail(c, CRUX_ENOMEM, "batch_train! (dense): %zu bytes of staging", 2 * need);
      dtmp_bytes = 2 * need; } }
  Carve cv{(char*)dtmp, 0};
  float* x = cv.take<float>((size_t)bmax * od); float* dy = cv.take<float>((size_t)bmax * nout);
  float* dinfo = cv.take<float>(CRUX_INFO_N); double* st = cv.take<double>(8); double* ssq = cv.take<double>(2 + SUMSQ_BLOCKS); int32_t* status = cv.take<int32_t>(4);
  HIPCHK(c, hipMemsetAsync(status, 0, 16, strm));
  if (which && !c->dense_pinned2 && hipHostMalloc(&c->dense_pinned2, 256, hipHostMallocDefault) != hipSuccess) { c->dense_pinned2 = nullptr; return crux_fail(c, CRUX_ENOMEM, "batch_train! (dense): pinned staging"); }
  float* hinfo = which ? (float*)c->dense_pinned2 : (float*)crux_pinned(c, sizeof(float) * CRUX_INFO_N + 16); if (!hinfo) return crux_fail(c, CRUX_ENOMEM, "batch_train! (dense): pinned staging");
  const bool pg = CRUX_IS_PG(a.loss); const bool step_sync = (pg && a.target_kl >= 0.f);
  long long total_batches = 0; int epochs_run = 0, err = 0, why = 0; bool stop = false;      // why: 16 + the bound that ended a replica-group wait (peer_wait.h)
  std::vector<float> ei((size_t)CRUX_INFO_N * (size_t)n_epochs, 0.f);
  auto read_info = [&]() -> int32_t {
    HIPCHK(c, hipMemcpyAsync(hinfo, dinfo, sizeof(float) * CRUX_INFO_N, hipMemcpyDeviceToHost, strm));
    HIPCHK(c, hipMemcpyAsync(hinfo + CRUX_INFO_N, status, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, strm));
    HIPCHK(c, hipStreamSynchronize(strm));
    int32_t s[2]; memcpy(s, hinfo + CRUX_INFO_N, sizeof s); if (s[0] == CRUX_ENAN || s[0] == CRUX_EHIP) { err = s[0]; why = s[1]; }
    return CRUX_OK;
  };
  for (int ep = 0; ep < n_epochs && !stop && !err; ++ep) {
    const int32_t* order = a.ids ? a.ids : a.ord_all + (size_t)ep * (size_t)a.len;
    bool fresh = false;                      // hinfo holds the info row of the latest minibatch
    for (int64_t s0 = 0; s0 < total_rows; s0 += a.bs) {                                                    // partition(1:len, batch_size) (training.jl:40)
      const int64_t nb = (total_rows - s0) < a.bs ? (total_rows - s0) : a.bs;
      hipLaunchKernelGGL(k_gather_obs, dim3((unsigned)((nb * od + 255) / 256)), dim3(256), 0, strm, a.S, od, order + s0, nb, x);
      int32_t rc = crux_dense_forward(net, x, nb, strm); if (rc) return rc;
      if (a.lag) hipLaunchKernelGGL(k_lagrange_pid, dim3(1), dim3(256), 0, strm, a.lag, a.COST, a.EE, order + s0, nb, (float* const*)((a.need_px && c->peer_n > 1) ? c->peer_tab + which * CRUX_PX_MAXR : nullptr), c->peer_rank, c->peer_n, status, c->peer_timeout_ticks);
      PgHeadArgs q{}; q.lag = a.lag; q.CADV = a.CADV; q.z = crux_dense_act(net, nd.L); q.nout = nout; q.rows = order + s0; q.nb = nb; q.A = a.A; q.ad = a.ad; q.LP = a.LP; q.ADV = a.ADV; q.RET = a.RET;
      q.loss = a.loss; q.head = a.head; q.lo = 1.f - a.eps_clip; q.hi = 1.f + a.eps_clip; q.lambda_p = a.lambda_p; q.lambda_e = a.lambda_e; q.squash = a.squash;
      q.ls = net->p + nd.xoff; q.dy = dy; q.gx = net->g + nd.xoff; q.stats = st;
      hipLaunchKernelGGL(k_pg_head, dim3(1), dim3(256), 0, strm, q);
      // (round 4) the fused pullback of layers 1 / 0 where it applies (dense_fused.h: three launches instead of five); it leaves layer 0's gradient as quarter partials that
      // k_sumsq2 completes -- not in a replica group, whose flat all-reduce reads the whole gradient before the norm
      const bool group = a.need_px && c->peer_n > 1; Sumsq2Fix fx{};
      rc = crux_dense_backward(net, x, nb, dy, 1.0f, true, nullptr, strm, group ? nullptr : &fx, 0); if (rc) return rc;
      if (group)      // replica group: the gradient (and the statistics) of the GLOBAL minibatch, the same bits on every rank
        hipLaunchKernelGGL(k_px_allreduce_flat, dim3(1), dim3(1024), 0, strm, net->g, (int64_t)nd.n_params, st, (float* const*)(c->peer_tab + which * CRUX_PX_MAXR), c->peer_rank, c->peer_n, status, c->peer_timeout_ticks);
      hipLaunchKernelGGL(k_sumsq2, dim3(SUMSQ_BLOCKS), dim3(256), 0, strm, (float*)net->g, (int64_t)nd.n_params, (float*)nullptr, (int64_t)0, ssq, fx);
      hipLaunchKernelGGL(k_pg_info, dim3(1), dim3(1), 0, strm, (const double*)st, (const double*)ssq, nb, a.loss, a.head, a.lambda_p, a.lambda_e, (const float*)(net->p + nd.xoff), a.ad, dinfo, (const crux_lagrange*)a.lag);
      if (a.apply) { rc = adam_gated(net, ssq, status, true, strm); if (rc) return rc; }
      else hipLaunchKernelGGL(k_nan_status, dim3(1), dim3(1), 0, strm, (const double*)ssq, status);
      total_batches += 1; fresh = false;
      const bool last = s0 + a.bs >= total_rows, capped = a.max_batches > 0 && total_batches >= a.max_batches;
      if (step_sync || capped || last) { rc = read_info(); if (rc) return rc; fresh = true; }
      if (err) break;
      if (capped) break;                                                                                  // training.jl:45
      if (step_sync && hinfo[CRUX_INFO_KL] > a.target_kl) break;                                          // :46
    }
    if (err) break;
    if (!fresh) { const int32_t rc = read_info(); if (rc) return rc; if (err) break; }
    memcpy(&ei[(size_t)ep * CRUX_INFO_N], hinfo, sizeof(float) * CRUX_INFO_N);                            // aggregate_info(minibatch_infos) == the latest minibatch's (App. A-Q3)
    epochs_run += 1;
    if (pg && a.target_kl >= 0.f && hinfo[CRUX_INFO_KL] > a.target_kl) stop = true;                       // :49
    if (a.max_batches > 0 && total_batches >= a.max_batches) stop = true;                                 // :50
  }
  // status row and epoch infos where the persistent kernels leave them (run_batch / collect read them back)
  if (err && epochs_run == 0) { ei[CRUX_INFO_LOSS] = hinfo[CRUX_INFO_LOSS]; ei[CRUX_INFO_GRAD_NORM] = NAN; }
  int32_t hst[5] = {err, (int32_t)total_batches, epochs_run, 0, why ? why : 3};
  HIPCHK(c, hipMemcpyAsync(a.status, hst, sizeof hst, hipMemcpyHostToDevice, strm));
  if (a.epoch_infos) HIPCHK(c, hipMemcpyAsync(a.epoch_infos, ei.data(), sizeof(float) * ei.size(), hipMemcpyHostToDevice, strm));
  HIPCHK(c, hipStreamSynchronize(strm));            // hst / ei are stack / heap memory of this call
  return crux_launch_check(c, "batch_train! (dense)");
}
