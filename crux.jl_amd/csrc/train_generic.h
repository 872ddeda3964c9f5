// train_generic.h -- the shape-generic persistent learner body (train! / batch_train!, src/training.jl:13-55) shared by train.hip and env.hip.
#pragma once
#include "train_args.h"

#define TR_CH 32
#ifndef EPS32F
#define EPS32F 1.1920928955078125e-07f
#endif

__device__ __forceinline__ double block_sum_d(double v, double* red, int tid) {
  // 256 threads = 4 waves. Deterministic: wave butterfly then fixed-order sum of 4 partials.
  v = wave_sum_d(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  const double t = ((red[0] + red[1]) + red[2]) + red[3];
  return t;
}

// the whole learner as a device function: k_train_generic (train.hip) is this body launched as one workgroup; the small-network solve kernel (env.hip) calls it
// once per value_training epoch. 256 threads; sm = dynamic LDS of generic_lds_bytes(nd), red = 4 doubles of LDS.
__device__ __forceinline__ void train_generic_run(const TrainArgs& a, float* sm, double* red) {
#pragma clang fp contract(fast)      // train.hip's own mode: the same bits when this body is compiled inside env.hip (built with -ffp-contract=off for the dynamics)
  const NetDesc& nd = a.nd;
  const int tid = threadIdx.x;
  const int L = nd.L, nout = nd.dims[L], od = nd.dims[0];
  // LDS carve: acts[l] (l=0..L) [sample][feature], two delta buffers, per-sample extras scratch
  float* acts[CRUX_MAXL + 1]; int off = 0;
  for (int l = 0; l <= L; ++l) { acts[l] = sm + off; off += nd.dims[l] * TR_CH; }
  float* dA = sm + off; off += nd.maxdim * TR_CH;
  float* dB = sm + off; off += nd.maxdim * TR_CH;
  float* exs = sm + off;   // [TR_CH x n_extra] per-sample logSigma gradient contributions

  double bp1 = a.bp[0], bp2 = a.bp[1];   // every thread carries the beta powers in registers (identical values)
  const float lo = 1.f - a.eps_clip, hi = 1.f + a.eps_clip;
  const bool a2c = a.loss == CRUX_LOSS_A2C;
  int32_t* order_cur = a.order_a; int32_t* order_nxt = a.order_b;
  long long total_batches = 0; int epochs_run = 0; int err = 0; bool stop = false;
  float info[CRUX_INFO_N];
#pragma unroll
  for (int q = 0; q < CRUX_INFO_N; ++q) info[q] = 0.f;
  // lagrange_ppo_loss: every thread carries an identical copy of the PID state (ppo.jl:192-201)
  const bool lagr = a.lag != nullptr;
  crux_lagrange lg{}; if (lagr) lg = *a.lag;
  float pen = 0.f;

  const int n_epochs = a.ids ? 1 : a.epochs;
  if (!a.ids && !a.ord_all) { for (int64_t j = tid; j < a.len; j += 256) order_cur[j] = (int32_t)j; __syncthreads(); }
  if (!a.ids && !a.ord_all && a.pre_epochs > 0) {
    __syncthreads();
    for (int pe = 0; pe < a.pre_epochs; ++pe) {
      if (a.pre_perms) { for (int64_t j = tid; j < a.len; j += 256) order_nxt[j] = order_cur[a.pre_perms[(int64_t)pe * a.len + j]]; }
      else { const crux_perm pp = crux_perm_make(a.pre_seed, a.pre_counter + (uint64_t)pe, 0, (uint32_t)a.len);
        for (int64_t j = tid; j < a.len; j += 256) order_nxt[j] = order_cur[crux_perm_at(&pp, (uint32_t)j)]; }
      __syncthreads();
      int32_t* t = order_cur; order_cur = order_nxt; order_nxt = t;
    }
  }

  for (int ep = 0; ep < n_epochs && !stop && !err; ++ep) {
    if (!a.ids && a.ord_all) order_cur = const_cast<int32_t*>(a.ord_all) + (size_t)ep * (size_t)a.len;   // shuffle orders composed ahead of time by k_compose_order
    else if (!a.ids) {
      // shuffle!(D): new[:,j] = old[:,perm[j]]  (experience_buffer.jl:118-124) as an index composition
      if (a.perms) { for (int64_t j = tid; j < a.len; j += 256) order_nxt[j] = order_cur[a.perms[(int64_t)ep * a.len + j]]; }
      else { const crux_perm pp = crux_perm_make(a.shuffle_seed, a.shuffle_counter + (uint64_t)ep, 0, (uint32_t)a.len);
        for (int64_t j = tid; j < a.len; j += 256) order_nxt[j] = order_cur[crux_perm_at(&pp, (uint32_t)j)]; }
      __syncthreads();
      int32_t* t = order_cur; order_cur = order_nxt; order_nxt = t;
    }
    const int64_t total_rows = a.ids ? a.n_ids : a.len;
    for (int64_t st = 0; st < total_rows; st += a.bs) {                     // partition(1:len, batch_size) (training.jl:40)
      const int nb = (int)((total_rows - st) < a.bs ? (total_rows - st) : a.bs);
      const float invB = 1.0f / (float)nb;
      for (int i = tid; i < nd.n_params; i += 256) a.g[i] = 0.f;
      double s_lossp = 0, s_H = 0, s_kl = 0, s_adv = 0, s_ret = 0, s_clip = 0, s_sq = 0, s_q = 0, s_cost = 0;
      if (lagr) {   // the penalty update inside the loss (ppo.jl:80-116): it reads only the minibatch's :cost and :episode_end, so it runs ahead of the passes
        double sc_ = 0.0, ne_ = 0.0;
        for (int i = tid; i < nb; i += 256) { const int64_t row = a.ids ? (int64_t)a.ids[st + i] : (int64_t)order_cur[st + i]; sc_ += (double)a.COST[row]; ne_ += a.EE[row] ? 1.0 : 0.0; }
        const double t_sc = block_sum_d(sc_, red, tid), t_ne = block_sum_d(ne_, red, tid);
        const float Jc = (float)t_sc / (float)t_ne;                                   // sum(D[:cost]) / sum(D[:episode_end]) (:86): Float32 / Int
        const float dl = Jc - lg.target_cost;                                         // :91
        { const float x = lg.I + lg.Ki * dl; lg.I = x > lg.Ki_max ? lg.Ki_max : (x < 0.f ? 0.f : x); }             // :94 clamp(I + Ki*Delta, 0, Ki_max)
        lg.smooth_delta = (float)(lg.ema_alpha * (double)lg.smooth_delta + (1.0 - lg.ema_alpha) * (double)dl);      // :98 (Float64 arithmetic, Float32 store)
        lg.smooth_Jc = (float)(lg.ema_alpha * (double)lg.smooth_Jc + (1.0 - lg.ema_alpha) * (double)Jc);            // :99
        { const float x = lg.smooth_Jc - lg.Jc_prev; lg.deriv_term = (x != x) ? x : (x > 0.f ? x : 0.f); }           // :102 max(0, .) keeps NaN
        lg.Jc_prev = lg.smooth_Jc;                                                    // :105
        { const float x = (lg.Kp * lg.smooth_delta + lg.I) + lg.Kd * lg.deriv_term; pen = x > lg.penalty_max ? lg.penalty_max : (x < 0.f ? 0.f : x); }   // :108
        lg.penalty = pen; lg.cur_cost = Jc;
      }
      for (int c0 = 0; c0 < nb; c0 += TR_CH) {
        const int ns = (nb - c0) < TR_CH ? (nb - c0) : TR_CH;
        // ---- gather the chunk's observations (minibatch view, experience_buffer.jl:170)
        for (int idx = tid; idx < od * ns; idx += 256) { const int s = idx / od, k = idx - s * od;
          const int64_t row = a.ids ? (int64_t)a.ids[st + c0 + s] : (int64_t)order_cur[st + c0 + s];
          acts[0][s * od + k] = a.S[row * od + k]; }
        __syncthreads();
        // ---- forward
        for (int l = 0; l < L; ++l) {
          const int in = nd.dims[l], out = nd.dims[l + 1], act = nd.acts[l];
          const float* W = a.p + nd.woff[l]; const float* b = a.p + nd.boff[l];
          for (int idx = tid; idx < out * ns; idx += 256) { const int o = idx % out, s = idx / out;
            float acc = 0.f;
            for (int k = 0; k < in; ++k) acc = fmaf(W[o + out * k], acts[l][s * in + k], acc);
            acts[l + 1][s * out + o] = crux_act(act, acc + b[o]); }
          __syncthreads();
        }
        // ---- loss head: one thread per sample -> d(loss)/d(output) into dA[s*nout + k]
        if (tid < ns) {
          const int s = tid; const int64_t row = a.ids ? (int64_t)a.ids[st + c0 + s] : (int64_t)order_cur[st + c0 + s];
          const float* z = acts[L] + s * nout; float* dy = dA + s * nout;
          if (a.loss == CRUX_LOSS_MSE_ACTION) {                                  // Flux.mse(action(pi,s), a)  il/bc.jl:1: mean over act_dim x batch
            const float* av = (const float*)a.A + row * a.ad; const float inv = invB / (float)nout;
            for (int k = 0; k < nout; ++k) { const float d = z[k] - av[k]; s_sq += (double)(d * d) / (double)nout; dy[k] = 2.f * d * inv; }
          } else if (a.loss == CRUX_LOSS_VALUE_MSE) {                            // Flux.mse(value(pi,s), return)  ppo.jl:60
            const float d = z[0] - a.RET[row]; s_sq += (double)(d * d); dy[0] = 2.f * d * invB;
          } else if (a.loss == CRUX_LOSS_TD_INTERNAL) {                          // td_loss utils.jl:76-87
            const uint8_t* av = (const uint8_t*)a.A + row * a.ad; float Q = 0.f;
            for (int k = 0; k < nout; ++k) Q += z[k] * (av[k] ? 1.f : 0.f);
            const float d = Q - a.Y[row]; const float w = a.Wt ? a.Wt[row] : 1.f;
            s_sq += (double)(d * d * w); s_q += (double)Q;
            for (int k = 0; k < nout; ++k) dy[k] = av[k] ? 2.f * d * w * invB : 0.f;
          } else {                                                                // ppo_loss ppo.jl:4-21
            const float A = a.ADV[row], oldlp = a.LP[row]; float newlp = 0.f, H = 0.f, r, g;
            if (a.head == CRUX_HEAD_CATEGORICAL) {
              const uint8_t* av = (const uint8_t*)a.A + row * a.ad;
              float mx = z[0]; for (int k = 1; k < nout; ++k) mx = z[k] > mx ? z[k] : mx;
              float sum = 0.f; for (int k = 0; k < nout; ++k) sum += expf(z[k] - mx);
              float q = 0.f, hp = 0.f;
              for (int k = 0; k < nout; ++k) { const float pk = expf(z[k] - mx) / sum; q += pk * (av[k] ? 1.f : 0.f);
                const float lg = logf(pk + EPS32F); H -= pk * lg; hp += (-lg - pk / (pk + EPS32F)) * pk; }
              newlp = logf(q);
              r = expf(newlp - oldlp); const float u = r * A, rc = fminf(fmaxf(r, lo), hi), cl = rc * A;
              g = (u <= cl) ? A : 0.f; s_lossp += (double)(a2c ? newlp * A : (u <= cl ? u : cl));
              if (a2c) { g = A; r = 1.f; }                                        // a2c_loss (a2c.jl:4-15): d(-mean(logpdf .* A)); clip statistics off below
              float gcr = 0.f;                                                    // lagrange: d/dr of max(r Ac, clamp(r) Ac) times r (ppo.jl:119)
              if (lagr) { const float Ac = a.CADV[row]; const float uc = r * Ac, clc = rc * Ac; s_cost += (double)(uc >= clc ? uc : clc); gcr = (uc >= clc ? Ac : 0.f) * r; }
              for (int k = 0; k < nout; ++k) { const float pk = expf(z[k] - mx) / sum; const float lgk = logf(pk + EPS32F);
                const float hk = -lgk - pk / (pk + EPS32F);
                const float dlogpi = pk * ((av[k] ? 1.f : 0.f) / q) - pk;
                const float base = -a.lambda_p * g * r * dlogpi - a.lambda_e * (pk * (hk - hp));
                dy[k] = lagr ? invB * ((base + pen * gcr * dlogpi) / (1.f + pen)) : invB * base; }
            } else {                                                              // GaussianPolicy policies.jl:333-348
              const float* av = (const float*)a.A + row * a.ad; const float* ls = a.p + nd.xoff;
              const float sq = a.squash;                                         // > 0: SquashedGaussianPolicy (policies.jl:374-396)
              for (int k = 0; k < a.ad; ++k) { const float sg = expf(sq > 0.f ? sq_clampls(ls[k]) : ls[k]); const float uk = sq > 0.f ? sq_untanh(av[k], sq) : av[k]; const float d = uk - z[k];
                newlp += (-(d * d) / (2.f * sg * sg) - 0.9189385332046727f - ls[k]); if (sq > 0.f) newlp -= sq_corr(uk); }
              r = expf(newlp - oldlp); const float u = r * A, rc = fminf(fmaxf(r, lo), hi), cl = rc * A;
              g = (u <= cl) ? A : 0.f; s_lossp += (double)(a2c ? newlp * A : (u <= cl ? u : cl));
              if (a2c) { g = A; r = 1.f; }
              float cf = -a.lambda_p * g * r;                                     // coefficient of d logpdf in d loss
              if (lagr) { const float Ac = a.CADV[row]; const float uc = r * Ac, clc = rc * Ac; s_cost += (double)(uc >= clc ? uc : clc);
                cf = (cf + pen * ((uc >= clc ? Ac : 0.f) * r)) / (1.f + pen); }
              for (int k = 0; k < a.ad; ++k) { const float sg = expf(sq > 0.f ? sq_clampls(ls[k]) : ls[k]); const float s2 = sg * sg; const float uk = sq > 0.f ? sq_untanh(av[k], sq) : av[k]; const float d = uk - z[k];
                const float inr = (sq > 0.f && !(ls[k] >= -5.f && ls[k] <= 2.f)) ? 0.f : 1.f;    // d clamp/dx
                dy[k] = invB * (cf * (d / s2));
                exs[s * a.ad + k] = invB * (cf * (((d * d) / s2) * inr - 1.f)); }
            }
            s_H += (double)H; s_kl += (double)(oldlp - newlp); s_adv += (double)A; if (a.RET) s_ret += (double)a.RET[row];
            if (!a2c && (r > hi || r < lo)) s_clip += 1.0;
          }
        }
        __syncthreads();
        if (CRUX_IS_PG(a.loss) && a.head == CRUX_HEAD_GAUSSIAN && tid < a.ad) {   // deterministic per-dimension sum over the chunk
          float acc = 0.f; for (int s = 0; s < ns; ++s) acc += exs[s * a.ad + tid]; a.g[nd.xoff + tid] += acc; }
        // ---- backward through the layers
        float* dcur = dA; float* dnxt = dB;
        for (int l = L - 1; l >= 0; --l) {
          const int in = nd.dims[l], out = nd.dims[l + 1], act = nd.acts[l];
          for (int idx = tid; idx < out * ns; idx += 256) dcur[idx] = crux_act_grad(act, acts[l + 1][idx], dcur[idx]);
          __syncthreads();
          const float* W = a.p + nd.woff[l];
          for (int pidx = tid; pidx < out * in + out; pidx += 256) {
            float acc = 0.f;
            if (pidx < out * in) { const int o = pidx % out, k = pidx / out;
              for (int s = 0; s < ns; ++s) acc = fmaf(dcur[s * out + o], acts[l][s * in + k], acc);
              a.g[nd.woff[l] + pidx] += acc; }
            else { const int o = pidx - out * in; for (int s = 0; s < ns; ++s) acc += dcur[s * out + o]; a.g[nd.boff[l] + o] += acc; }
          }
          if (l > 0) {
            for (int idx = tid; idx < in * ns; idx += 256) { const int k = idx % in, s = idx / in; float acc = 0.f;
              for (int o = 0; o < out; ++o) acc = fmaf(W[o + out * k], dcur[s * out + o], acc);
              dnxt[s * in + k] = acc; }
          }
          __syncthreads();
          float* t = dcur; dcur = dnxt; dnxt = t;
        }
      }
      if (CRUX_IS_PG(a.loss) && a.head == CRUX_HEAD_GAUSSIAN && tid < a.ad) a.g[nd.xoff + tid] += lagr ? -a.lambda_e / (1.f + pen) : -a.lambda_e;   // d(-le*H)/dlogSigma, H scalar
      // ---- reductions: stats and grad norm (utils.jl:49-55)
      double ssq = 0.0; for (int i = tid; i < nd.n_params; i += 256) { const double gi = (double)a.g[i]; ssq += gi * gi; }
      const double t_ssq = block_sum_d(ssq, red, tid);
      const double t_lossp = block_sum_d(s_lossp, red, tid), t_H = block_sum_d(s_H, red, tid), t_kl = block_sum_d(s_kl, red, tid);
      const double t_adv = block_sum_d(s_adv, red, tid), t_ret = block_sum_d(s_ret, red, tid), t_clip = block_sum_d(s_clip, red, tid);
      const double t_sq = block_sum_d(s_sq, red, tid), t_q = block_sum_d(s_q, red, tid);
      const double t_cost = lagr ? block_sum_d(s_cost, red, tid) : 0.0;
      const float gnorm = (float)sqrt(t_ssq);
#pragma unroll
      for (int q = 0; q < CRUX_INFO_N; ++q) info[q] = 0.f;
      if (CRUX_IS_PG(a.loss)) {
        const float p_loss = (float)(-(t_lossp / (double)nb)); float entropy, e_loss;
        if (a.head == CRUX_HEAD_CATEGORICAL) { entropy = (float)(t_H / (double)nb); e_loss = -entropy; }
        else { float Hs = 1.4189385332046727f; for (int k = 0; k < a.ad; ++k) Hs += a.p[nd.xoff + k]; entropy = Hs; e_loss = -Hs; }
        info[CRUX_INFO_LOSS] = a.lambda_p * p_loss + a.lambda_e * e_loss; info[CRUX_INFO_ENTROPY] = entropy; info[CRUX_INFO_KL] = (float)(t_kl / (double)nb);
        info[CRUX_INFO_CLIP_FRACTION] = (float)t_clip / (float)nb; info[CRUX_INFO_AVG_ADVANTAGE] = (float)(t_adv / (double)nb); info[CRUX_INFO_AVG_RETURN] = (float)(t_ret / (double)nb);
        if (lagr) { const float cost_loss = pen * (float)(t_cost / (double)nb);                                    // ppo.jl:119
          info[CRUX_INFO_LOSS] = ((a.lambda_p * p_loss + a.lambda_e * e_loss) + cost_loss) / (1.f + pen);           // :131
          info[CRUX_INFO_PENALTY] = pen; info[CRUX_INFO_CUR_COST] = lg.cur_cost; info[CRUX_INFO_COST_LOSS] = cost_loss; info[CRUX_INFO_P_LOSS] = a.lambda_p * p_loss; }
      } else { info[CRUX_INFO_LOSS] = (float)(t_sq / (double)nb); if (a.loss == CRUX_LOSS_TD_INTERNAL) info[2] = (float)(t_q / (double)nb); }
      info[CRUX_INFO_GRAD_NORM] = gnorm;
      if (isnan(gnorm)) { err = CRUX_ENAN; break; }                            // training.jl:20 -- no update
      // ---- Flux.update!(Adam) (training.jl:21), Float64 per element like the reference
      if (a.apply) {
        for (int i = tid; i < nd.n_params; i += 256) {
          const double gd = (double)a.g[i];
          const float mi = (float)(a.b1 * (double)a.m[i] + (1.0 - a.b1) * gd);
          const float vi = (float)(a.b2 * (double)a.v[i] + ((1.0 - a.b2) * gd) * gd);
          const float d = (float)((double)mi / (1.0 - bp1) / (sqrt((double)vi / (1.0 - bp2)) + a.eps) * a.eta);
          a.m[i] = mi; a.v[i] = vi; a.p[i] = a.p[i] - d;
        }
        bp1 *= a.b1; bp2 *= a.b2;
      }
      __syncthreads();
      total_batches += 1;
      if (a.max_batches > 0 && total_batches >= a.max_batches) break;          // training.jl:45
      if (a.target_kl >= 0.f && CRUX_IS_PG(a.loss) && info[CRUX_INFO_KL] > a.target_kl) break;   // :46
    }
    if (err) break;
    if (tid < CRUX_INFO_N && a.epoch_infos) a.epoch_infos[(size_t)ep * CRUX_INFO_N + tid] = info[tid];   // aggregate == last minibatch (Q3)
    epochs_run += 1;
    if (a.target_kl >= 0.f && CRUX_IS_PG(a.loss) && info[CRUX_INFO_KL] > a.target_kl) stop = true;  // :49
    if (a.max_batches > 0 && total_batches >= a.max_batches) stop = true;                                // :50
  }
  if (tid == 0) {
    a.status[0] = err; a.status[1] = (int32_t)total_batches; a.status[2] = epochs_run; a.status[3] = (order_cur == a.order_a) ? 0 : 1;
    a.bp[0] = bp1; a.bp[1] = bp2;
    if (lagr) *a.lag = lg;
  }
}

