// env.hip -- vectorised Samplers: environment dynamics + policy forward + column writes in one kernel.
// Reference: Sampler / reset_sampler! / step! / steps! / terminate_episode! (src/sampler.jl:1-173),
// DiscreteNetwork exploration (src/policies.jl:137-142), GaussianPolicy exploration (:338-344),
// MixedPolicy (:474-494), GaussianNoiseExplorationPolicy (:510-514), LinearDecaySchedule (src/utils.jl:116-126),
// tovec whitening (src/spaces.jl:25). Environment dynamics restate gymnasium's classic_control definitions
// (the reference reaches them through POMDPGym/PyCall, src/sampler.jl:93).
#include "common.h"
#include "train_generic.h"
#include "mlp_forward.h"
#include "ops_small.h"
#include "per_tree.h"

int32_t crux_buffer_ring_indices(crux_buffer* b, int64_t N, std::vector<int64_t>& I);
void crux_buffer_ring_advance(crux_buffer* b, int64_t N);
int32_t crux_buffer_per_on_push(crux_buffer* b, const int64_t* d_I, int64_t N);
int32_t crux_buffer_ring_ids_device(crux_buffer* b, int64_t N, int64_t* d_out);
bool crux_per_push_fused(crux_buffer* b, int64_t n, int64_t* d_ids);
bool crux_per_push_plan(crux_buffer* b, int64_t n, int* touch);
void crux_per_push_done(crux_buffer* b, int touch);

#define ENV_MAXSD 32          // SYNTH keeps one Float64 per observation; CartPole 4, Pendulum / GridWorld 2
#define ENV_MAXOBS 32
#define PI_D 3.14159265358979323846

// ---- dynamics (float64, same operation order as the oracle) ------------------------------------------
__device__ __forceinline__ void cartpole_step(const double* s, int action, double* sn, float* r, uint8_t* done) {
  const double gravity = 9.8, masscart = 1.0, masspole = 0.1, total_mass = masspole + masscart, length = 0.5,
               polemass_length = masspole * length, force_mag = 10.0, tau = 0.02;
  double x = s[0], x_dot = s[1], theta = s[2], theta_dot = s[3];
  const double force = action == 1 ? force_mag : -force_mag;
  const double costheta = cos(theta), sintheta = sin(theta);
  const double temp = __ddiv_rn(__dadd_rn(force, __dmul_rn(__dmul_rn(polemass_length, __dmul_rn(theta_dot, theta_dot)), sintheta)), total_mass);
  const double den = __dmul_rn(length, __dsub_rn(4.0 / 3.0, __ddiv_rn(__dmul_rn(masspole, __dmul_rn(costheta, costheta)), total_mass)));
  const double thetaacc = __ddiv_rn(__dsub_rn(__dmul_rn(gravity, sintheta), __dmul_rn(costheta, temp)), den);
  const double xacc = __dsub_rn(temp, __ddiv_rn(__dmul_rn(__dmul_rn(polemass_length, thetaacc), costheta), total_mass));
  x = __dadd_rn(x, __dmul_rn(tau, x_dot)); x_dot = __dadd_rn(x_dot, __dmul_rn(tau, xacc));
  theta = __dadd_rn(theta, __dmul_rn(tau, theta_dot)); theta_dot = __dadd_rn(theta_dot, __dmul_rn(tau, thetaacc));
  sn[0] = x; sn[1] = x_dot; sn[2] = theta; sn[3] = theta_dot;
  const double xth = 2.4, thth = 12.0 * 2.0 * PI_D / 360.0;
  *done = (x < -xth || x > xth || theta < -thth || theta > thth) ? 1 : 0;
  *r = 1.0f;
}
__device__ __forceinline__ double angle_normalize(double x) { double t = fmod(x + PI_D, 2.0 * PI_D); if (t < 0) t += 2.0 * PI_D; return t - PI_D; }
__device__ __forceinline__ void pendulum_step(const double* s, float a, double* sn, float* r, uint8_t* done) {
  const double g = 10.0, m = 1.0, l = 1.0, dt = 0.05, max_speed = 8.0, max_torque = 2.0;
  const double th = s[0], thdot = s[1];
  double u = (double)a; if (u < -max_torque) u = -max_torque; if (u > max_torque) u = max_torque;
  const double an = angle_normalize(th);
  const double costs = __dadd_rn(__dadd_rn(__dmul_rn(an, an), __dmul_rn(0.1, __dmul_rn(thdot, thdot))), __dmul_rn(0.001, __dmul_rn(u, u)));
  double newthdot = __dadd_rn(thdot, __dmul_rn(__dadd_rn(__dmul_rn(3.0 * g / (2.0 * l), sin(th)), __dmul_rn(3.0 / (m * l * l), u)), dt));
  if (newthdot < -max_speed) newthdot = -max_speed; if (newthdot > max_speed) newthdot = max_speed;
  sn[0] = __dadd_rn(th, __dmul_rn(newthdot, dt)); sn[1] = newthdot; *r = (float)(-costs); *done = 0;
}
// POMDPModels.SimpleGridWorld restated (README example, configs[0]); see the oracle for the definition.
__device__ __forceinline__ float gridworld_reward(double x, double y) {
  if (x == 4 && y == 3) return -10.f; if (x == 4 && y == 6) return -5.f; if (x == 9 && y == 3) return 10.f; if (x == 8 && y == 8) return 3.f; return 0.f;
}
__device__ __forceinline__ void gridworld_step(const double* s, int a, double u, double* sn, float* r, uint8_t* done) {
  const double tprob = 0.7;
  const double x = s[0], y = s[1];
  const float rw = gridworld_reward(x, y);
  *r = rw;
  if (rw != 0.f) { sn[0] = -1; sn[1] = -1; *done = 1; return; }
  int dir = a;
  if (!(u < tprob)) { int k = (int)((u - tprob) / (1.0 - tprob) * 3.0); if (k > 2) k = 2; int cnt = 0;
    for (int d = 0; d < 4; ++d) { if (d == a) continue; if (cnt == k) { dir = d; break; } ++cnt; } }
  const double ddx = dir == 2 ? -1.0 : (dir == 3 ? 1.0 : 0.0), ddy = dir == 0 ? 1.0 : (dir == 1 ? -1.0 : 0.0);
  double nx = x + ddx, ny = y + ddy;
  if (nx < 1 || nx > 10 || ny < 1 || ny > 10) { nx = x; ny = y; }
  sn[0] = nx; sn[1] = ny; *done = 0;
}
// SYNTH: see include/cruxhip.h for the definition (the oracle's synth_step is the twin)
__device__ __forceinline__ bool env_is_synth(int kind) { return kind == CRUX_ENV_SYNTH || kind == CRUX_ENV_SYNTH_DISCRETE; }
__device__ __forceinline__ void env_writelane(uint32_t& dst, const uint32_t uniform_v, const int l) {      // dst[lane l] = the (wave-uniform) value
  const int sv = __builtin_amdgcn_readfirstlane((int)uniform_v);
  asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(dst) : "s"(sv), "n"(l));
}
__device__ __forceinline__ double env_readlane_f64(const double v, const int l) {
  const uint64_t b = __builtin_bit_cast(uint64_t, v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, l), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), l);
  return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
// par_lane >= 0: the whole wave executes this with identical arguments in every lane (the register-resident rollout kernels); lane i evaluates element i -- the Float64 sin
// is the cost of the step -- and the elements are collected with v_readlane (so, sa compile-time there). The same operations per element: the same bits.
__device__ __forceinline__ void synth_step(int so, int sa, bool discrete, const double* s, int ai, const float* a, double* sn, float* r, uint8_t* done, const int par_lane = -1) {
  double ss = 0.0;
  if (par_lane >= 0) {
    // lane i takes s[i] and s[i+1] with v_writelane (a select chain over the lane id is turned into an indexed load of the state array, which then lives in scratch)
    uint32_t il = 0, ih = 0, jl = 0, jh = 0; double u = 0.0;
#pragma unroll
    for (int i = 0; i < so; ++i) { const uint64_t bi = __builtin_bit_cast(uint64_t, s[i]), bj = __builtin_bit_cast(uint64_t, s[(i + 1) % so]);
      env_writelane(il, (uint32_t)bi, i); env_writelane(ih, (uint32_t)(bi >> 32), i); env_writelane(jl, (uint32_t)bj, i); env_writelane(jh, (uint32_t)(bj >> 32), i); }
    const double si = __builtin_bit_cast(double, ((uint64_t)ih << 32) | il), sj = __builtin_bit_cast(double, ((uint64_t)jh << 32) | jl);
    const int ii = par_lane < so ? par_lane : 0;
    if (discrete) u = ((ii + ai) % sa == 0) ? 1.0 : -0.25;
    else { float av = a[0];
#pragma unroll
      for (int q = 0; q < sa; ++q) if (ii % sa == q) av = a[q]; u = (double)av; if (u < -1.0) u = -1.0; if (u > 1.0) u = 1.0; }
    const double mine = __dadd_rn(__dmul_rn(0.9, si), __dmul_rn(0.1, sin(__dadd_rn(sj, u))));
#pragma unroll
    for (int i = 0; i < so; ++i) { sn[i] = env_readlane_f64(mine, i); ss = __dadd_rn(ss, __dmul_rn(sn[i], sn[i])); }
  } else
  for (int i = 0; i < so; ++i) {
    double u;
    if (discrete) u = ((i + ai) % sa == 0) ? 1.0 : -0.25;
    else { u = (double)a[i % sa]; if (u < -1.0) u = -1.0; if (u > 1.0) u = 1.0; }
    sn[i] = __dadd_rn(__dmul_rn(0.9, s[i]), __dmul_rn(0.1, sin(__dadd_rn(s[(i + 1) % so], u))));
    ss = __dadd_rn(ss, __dmul_rn(sn[i], sn[i]));
  }
  *r = (float)__dadd_rn(-(ss / (double)so), __dmul_rn(0.05, sn[0]));
  *done = sn[0] > 0.9 ? 1 : 0;
}
// the cost channel of the restated environments (include/cruxhip.h): info["cost"] of the step that reached s' (sampler.jl:65-66,114)
__device__ __forceinline__ float env_cost(int kind, int so, const double* sn) {
  if (env_is_synth(kind)) { const double x = sn[1 % so]; return (float)__dmul_rn(25.0, __dmul_rn(x, x)); }
  if (kind == CRUX_ENV_CARTPOLE) return fabs(sn[2]) > 0.05 ? 1.f : 0.f;
  if (kind == CRUX_ENV_PENDULUM) return fabs(sn[1]) > 4.0 ? 1.f : 0.f;
  return 0.f;
}
__device__ __forceinline__ void env_obs(int kind, const double* s, float* o, int od = 0) {
  if (kind == CRUX_ENV_CARTPOLE) { o[0] = (float)s[0]; o[1] = (float)s[1]; o[2] = (float)s[2]; o[3] = (float)s[3]; }
  else if (kind == CRUX_ENV_PENDULUM) { o[0] = (float)cos(s[0]); o[1] = (float)sin(s[0]); o[2] = (float)s[1]; }
  else if (env_is_synth(kind)) { for (int i = 0; i < od; ++i) o[i] = (float)s[i]; }
  else { o[0] = (float)s[0]; o[1] = (float)s[1]; }
}
__device__ __forceinline__ void env_draw_initial(int kind, uint64_t seed, uint64_t n_resets, uint32_t env, double* s, int sd = 0) {
  if (env_is_synth(kind)) {                                   // U(-0.05, 0.05)^so, two Float64 uniforms per Philox block
    for (int i = 0; i < sd; ++i) { const crux_u32x4 x = crux_philox(seed, 16 * n_resets + (uint64_t)(i >> 1), env, CRUX_RNG_RESET);
      const double ui = (i & 1) ? crux_u32x2_to_f64(x.v[2], x.v[3]) : crux_u32x2_to_f64(x.v[0], x.v[1]); s[i] = __dadd_rn(-0.05, __dmul_rn(0.1, ui)); }
    return;
  }
  const crux_u32x4 a = crux_philox(seed, 2 * n_resets, env, CRUX_RNG_RESET), b = crux_philox(seed, 2 * n_resets + 1, env, CRUX_RNG_RESET);
  const double u0 = crux_u32x2_to_f64(a.v[0], a.v[1]), u1 = crux_u32x2_to_f64(a.v[2], a.v[3]), u2 = crux_u32x2_to_f64(b.v[0], b.v[1]), u3 = crux_u32x2_to_f64(b.v[2], b.v[3]);
  if (kind == CRUX_ENV_CARTPOLE) { s[0] = __dadd_rn(-0.05, __dmul_rn(0.1, u0)); s[1] = __dadd_rn(-0.05, __dmul_rn(0.1, u1)); s[2] = __dadd_rn(-0.05, __dmul_rn(0.1, u2)); s[3] = __dadd_rn(-0.05, __dmul_rn(0.1, u3)); }
  else if (kind == CRUX_ENV_PENDULUM) { s[0] = __dadd_rn(-PI_D, __dmul_rn(2.0 * PI_D, u0)); s[1] = __dadd_rn(-1.0, __dmul_rn(2.0, u1)); }
  else { s[0] = 1.0 + floor(10.0 * u0); s[1] = 1.0 + floor(10.0 * u1); }
}
__device__ __forceinline__ float randn_f32(uint64_t seed, uint64_t ctr, uint32_t stream, int which) {
  const crux_u32x4 x = crux_philox(seed, ctr, stream, CRUX_RNG_NOISE);
  const double u1 = crux_u32x2_to_f64(x.v[0], x.v[1]), u2 = crux_u32x2_to_f64(x.v[2], x.v[3]);
  const double rr = sqrt(-2.0 * log(1.0 - u1)), th = 2.0 * PI_D * u2;
  return (float)(which ? rr * sin(th) : rr * cos(th));
}
__device__ __forceinline__ double linear_decay(double start, double stop, int64_t steps, int64_t i) {
  const double rate = (start - stop) / (double)steps; const double val = start - (double)i * rate; return val > stop ? val : stop;
}

struct RolloutArgs {
  NetDesc nd; const float* p;
  int32_t kind, E, max_steps, od, ad, sd, act_kind;
  uint64_t seed;
  const float* mu; const float* sigma;
  double* state; int64_t* ep_len; int64_t* n_resets; int64_t* steps_taken; float* svec; double* acc;
  float* S; void* A; float* SP; float* R; uint8_t* D; uint8_t* EE; float* LP; int64_t* TT; int64_t* II; float* W; float* RET; float* ADV;
  float* COST; float* CADV; float* CRET;      // :cost, :cost_advantage, :cost_return (cost-constrained solvers; NULL = absent)
  int64_t base, C, T;
  crux_rollout_cfg cfg;
  float squash;      // SquashedGaussianPolicy ascale, 0 = GaussianPolicy
  // push!'s priority bookkeeping finished by the rollout launch itself (k_rollout_res with one environment; per_tree.h: push_touch_block); pt_pr == NULL: not asked for
  int64_t* pt_ids; float* pt_pr; float* pt_pminmax; float* pt_run; float* pt_total; float pt_alpha; int32_t pt_nlev, pt_touch; int64_t pt_N;
};

// exploration(pi_explore, svec; pi_on, i) / action(pi, svec) (sampler.jl:73; policies.jl:124-144, 338-344, 372-394, 466-494, 499-514) given the network outputs z of ONE
// observation: the action (aout: one-hot floats for the discrete heads, the action vector otherwise; ai: the discrete index) and its log-probability. Every draw is
// crux_philox(seed, f(ctr), stream = e, purpose) with ctr = the number of steps sampler e has taken so far -- shared by the rollout kernels (device environments) and
// k_policy_explore (caller-stepped environments, crux_policy_explore), so the two produce the same actions from the same observations.
__device__ __forceinline__ void rollout_head(const RolloutArgs& a, const float* z, const int ad, const int nout, const int e, const uint64_t gi, const uint64_t ctr,
                                             float* aout, int& ai, float& logprob) {
      logprob = NAN; ai = 0;
      if (a.cfg.head == CRUX_HEAD_CATEGORICAL || a.cfg.head == CRUX_HEAD_GREEDY_Q) {
        int greedy = 0; for (int q = 1; q < nout; ++q) if (z[q] > z[greedy]) greedy = q;
        if (!a.cfg.explore) ai = greedy;
        else if (a.cfg.eps_steps > 0 || a.cfg.head == CRUX_HEAD_GREEDY_Q) {
          const double eps = a.cfg.eps_steps > 0 ? linear_decay(a.cfg.eps_start, a.cfg.eps_stop, a.cfg.eps_steps, (int64_t)gi) : 0.0;
          const crux_u32x4 x = crux_philox(a.seed, ctr, (uint32_t)e, CRUX_RNG_ACTION);
          const double u = crux_u32x2_to_f64(x.v[0], x.v[1]);
          if (u < eps) { const crux_u32x4 y = crux_philox(a.seed, ctr, (uint32_t)e, CRUX_RNG_RANDACT); ai = (int)(((uint64_t)y.v[0] * (uint64_t)nout) >> 32); }
          else ai = greedy;
          if (a.LP) logprob = (float)log(eps * (1.0 / (double)nout) + (1.0 - eps));      // (a Float64 log per step: only where a :logprob column takes it)
        } else {
          float pr[ENV_MAXOBS];
          const float ldiv = a.cfg.logit_div > 0.f ? a.cfg.logit_div : 1.f;                       // softmax(value ./ alpha) (softq.jl:53)
          float mx = __fdiv_rn(z[0], ldiv); for (int q = 1; q < nout; ++q) { const float zq = __fdiv_rn(z[q], ldiv); mx = zq > mx ? zq : mx; }
          float sum = 0.f; for (int q = 0; q < nout; ++q) { pr[q] = expf(__fdiv_rn(z[q], ldiv) - mx); sum = __fadd_rn(sum, pr[q]); }
          for (int q = 0; q < nout; ++q) pr[q] = __fdiv_rn(pr[q], sum);
          const crux_u32x4 x = crux_philox(a.seed, ctr, (uint32_t)e, CRUX_RNG_ACTION);
          const float draw = crux_u32_to_f32(x.v[0]);
          float cp = pr[0]; ai = 0; while (cp <= draw && ai < nout - 1) { ai += 1; cp = __fadd_rn(cp, pr[ai]); }
          logprob = logf(pr[ai]);
        }
        for (int q = 0; q < ad; ++q) aout[q] = (q == ai) ? 1.f : 0.f;
      } else if (a.cfg.head == CRUX_HEAD_GAUSSIAN) {
        const float* ls = a.p + a.nd.xoff; float lp = 0.f;
        for (int q = 0; q < ad; ++q) {
          const float mu = z[q];
          if (a.squash > 0.f) {                                                   // SquashedGaussianPolicy (policies.jl:372, 388-394)
            if (a.cfg.explore) { const float sg = expf(sq_clampls(ls[q]));
              const float epsn = randn_f32(a.seed, ctr * (uint64_t)((ad + 1) / 2) + (uint64_t)(q / 2), (uint32_t)e, q & 1);
              const float u = __fadd_rn(__fmul_rn(epsn, sg), mu); const float s2 = __fmul_rn(sg, sg); const float dd = __fsub_rn(u, mu);
              aout[q] = __fmul_rn(a.squash, tanhf(u));
              lp = __fadd_rn(lp, __fsub_rn(__fsub_rn(__fsub_rn(__fdiv_rn(-__fmul_rn(dd, dd), __fmul_rn(2.f, s2)), 0.9189385332046727f), ls[q]), sq_corr(u))); }
            else aout[q] = __fmul_rn(a.squash, tanhf(mu));
          } else
          if (a.cfg.explore) { const float sg = expf(ls[q]);
            const float epsn = randn_f32(a.seed, ctr * (uint64_t)((ad + 1) / 2) + (uint64_t)(q / 2), (uint32_t)e, q & 1);
            aout[q] = __fadd_rn(__fmul_rn(epsn, sg), mu); const float s2 = __fmul_rn(sg, sg); const float dd = __fsub_rn(aout[q], mu);
            lp = __fadd_rn(lp, __fsub_rn(__fsub_rn(__fdiv_rn(-__fmul_rn(dd, dd), __fmul_rn(2.f, s2)), 0.9189385332046727f), ls[q])); }
          else aout[q] = mu;
        }
        logprob = a.cfg.explore ? lp : NAN;
      } else {
        for (int q = 0; q < ad; ++q) { float av = z[q];
          if (a.cfg.explore && a.cfg.noise_sigma >= 0.f) {
            float n0 = __fmul_rn(randn_f32(a.seed, ctr * (uint64_t)((ad + 1) / 2) + (uint64_t)(q / 2), (uint32_t)e, q & 1), a.cfg.noise_sigma);
            n0 = n0 < a.cfg.noise_eps_min ? a.cfg.noise_eps_min : n0 > a.cfg.noise_eps_max ? a.cfg.noise_eps_max : n0; av = __fadd_rn(av, n0);
            av = av < a.cfg.a_min ? a.cfg.a_min : av > a.cfg.a_max ? a.cfg.a_max : av; }
          aout[q] = av; }
      }
      if (a.cfg.explore == 2) logprob = NAN;      // action(pi, s) of an always_stochastic policy: exploration(pi, s)[1] with logprob NaN (policies.jl:124, sampler.jl:73)
}
// Everything of step! after the policy forward (sampler.jl:73-136): action + logprob from the head, env transition, column writes,
// episode bookkeeping. `writer` lanes store to the buffer; all callers advance identical copies of the sampler state.
__device__ __forceinline__ void rollout_tail(const RolloutArgs& a, const float* z, const int od, const int ad, const int nout, const int kind, const int e,
                                             const int64_t t, const int64_t j, const bool writer, double* st, int64_t& ep_len, int64_t& n_resets,
                                             int64_t& steps_taken, double& sum_r, int64_t& nee, float* next_obs, const int par_lane = -1,
                                             const float* mu_cached = nullptr, const float* sigma_cached = nullptr) {
  const float* mu = mu_cached ? mu_cached : a.mu; const float* sg = sigma_cached ? sigma_cached : a.sigma;      // (a kernel may hold the observation normalisation in LDS / registers)
  const int sd = kind == CRUX_ENV_CARTPOLE ? 4 : (env_is_synth(kind) ? od : 2);      // (SYNTH keeps one Float64 per observation: a compile-time od makes the state array registers)
      const uint64_t gi = a.cfg.i0 + (uint64_t)t * (uint64_t)a.E + (uint64_t)e;    // i + (j-1), env-minor (sampler.jl:161-163)
      const uint64_t ctr = (uint64_t)steps_taken;
      float logprob; int ai; float aout[ENV_MAXOBS];
      rollout_head(a, z, ad, nout, e, gi, ctr, aout, ai, logprob);
      // ---- env transition (sampler.jl:93-97)
      double sn[ENV_MAXSD]; float r; uint8_t done; float o[ENV_MAXOBS], spv[ENV_MAXOBS];
      if (kind == CRUX_ENV_CARTPOLE) cartpole_step(st, ai, sn, &r, &done);
      else if (kind == CRUX_ENV_PENDULUM) pendulum_step(st, aout[0], sn, &r, &done);
      else if (env_is_synth(kind)) synth_step(od, ad, kind == CRUX_ENV_SYNTH_DISCRETE, st, ai, aout, sn, &r, &done, par_lane);
      else { const crux_u32x4 xd = crux_philox(a.seed, ctr, (uint32_t)e, CRUX_RNG_ENVDYN); gridworld_step(st, ai, crux_u32x2_to_f64(xd.v[0], xd.v[1]), sn, &r, &done); }
      env_obs(kind, sn, o, od);
      for (int q = 0; q < od; ++q) spv[q] = __fdiv_rn(__fsub_rn(o[q], mu[q]), sg[q]);
      // ---- column writes (sampler.jl:101-107)
      if (writer) {
        if (a.act_kind == CRUX_ACTION_DISCRETE) { uint8_t* A = (uint8_t*)a.A + (size_t)j * ad; for (int q = 0; q < ad; ++q) A[q] = aout[q] != 0.f; }
        else { float* A = (float*)a.A + (size_t)j * ad; for (int q = 0; q < ad; ++q) A[q] = aout[q]; }
        for (int q = 0; q < od; ++q) a.SP[(size_t)j * od + q] = spv[q];
        a.R[j] = r; a.D[j] = done;
        if (a.LP) a.LP[j] = logprob;
        if (a.TT) a.TT[j] = ep_len + 1;
        if (a.II) a.II[j] = (int64_t)gi + 1;
        if (a.W) a.W[j] = 1.0f;
        if (a.RET) a.RET[j] = 0.f;
        if (a.ADV) a.ADV[j] = 0.f;
        if (a.COST) a.COST[j] = env_cost(kind, od, sn);                              // data[:cost][1,j] = info["cost"] (sampler.jl:114)
        if (a.CADV) a.CADV[j] = 0.f;
        if (a.CRET) a.CRET[j] = 0.f;
      }
      sum_r += (double)r; steps_taken += 1;
      // ---- episode bookkeeping (sampler.jl:130-136; terminate_episode! :53-69)
      ep_len += 1;
      uint8_t ee = 0;
      if (done || ep_len >= a.max_steps) {
        ee = 1; ++nee;
        env_draw_initial(kind, a.seed, (uint64_t)n_resets, (uint32_t)e, st, sd); n_resets += 1; ep_len = 0;
        env_obs(kind, st, o, od);
        for (int q = 0; q < od; ++q) next_obs[q] = __fdiv_rn(__fsub_rn(o[q], mu[q]), sg[q]);
      } else {
        for (int i = 0; i < sd; ++i) st[i] = sn[i];
        for (int q = 0; q < od; ++q) next_obs[q] = spv[q];
      }
      if (a.cfg.reset_at_end && t == a.T - 1 && ep_len > 0) {                       // sampler.jl:148
        ee = 1; ++nee;
        env_draw_initial(kind, a.seed, (uint64_t)n_resets, (uint32_t)e, st, sd); n_resets += 1; ep_len = 0;
        env_obs(kind, st, o, od);
        for (int q = 0; q < od; ++q) next_obs[q] = __fdiv_rn(__fsub_rn(o[q], mu[q]), sg[q]);
      }
      if (writer) a.EE[j] = ee;
}


// ---- register-resident policy for the IN-64-64-OUT family ------------------------------------------------------------------------
// One wave per environment. Lane j keeps row j of W1/W2 and column j of W3 in VGPRs for the whole rollout; every lane carries an
// identical copy of the sampler state (Float64 dynamics, Philox draws and the head are evaluated redundantly), so the only
// cross-lane traffic per step is the broadcast of H1 (64 floats through LDS) and one wave reduction per output.
template <int CTRL> __device__ __forceinline__ float env_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float env_wave_sum_uniform(float v) {      // the same bits in every lane (readfirstlane of one reduction order)
  v += env_dpp<0x128>(v); v += env_dpp<0x124>(v); v += env_dpp<0x122>(v); v += env_dpp<0x121>(v);     // row_ror 8,4,2,1: row totals
  float a0 = v, b0 = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a0), "+v"(b0));
  a0 = a0 + b0; b0 = a0;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a0), "+v"(b0));
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a0 + b0)));
}
// the forward pass of the register-resident IN-64-64-OUT policy on ONE observation, by one wave: lane j holds row j of W1 / W2 and column j of W3. Layers 1 and 2 are fma chains over
// k ascending (+ bias, activation), layer 3 a product per lane summed over the wave in one fixed order. Shared by k_rollout_h64 and k_explore_h64 (same bits from the same observation).
template <int IN, int OUT, int ACT>
__device__ __forceinline__ void h64_forward(const float (&w1)[IN], const float (&w2)[64], const float (&w3)[OUT], const float (&b3)[OUT], const float b1, const float b2,
                                            const float (&x)[IN], float* sh, const int lane, float (&z)[OUT]) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < IN; ++k) acc = fmaf(w1[k], x[k], acc);
    sh[lane] = crux_act(ACT, acc + b1);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    acc = 0.f;
#pragma unroll
    for (int k4 = 0; k4 < 16; ++k4) { const float4 hv = *(const float4*)&sh[4 * k4];
      acc = fmaf(w2[4 * k4], hv.x, acc); acc = fmaf(w2[4 * k4 + 1], hv.y, acc); acc = fmaf(w2[4 * k4 + 2], hv.z, acc); acc = fmaf(w2[4 * k4 + 3], hv.w, acc); }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    const float h2 = crux_act(ACT, acc + b2);
#pragma unroll
    for (int o = 0; o < OUT; ++o) z[o] = env_wave_sum_uniform(w3[o] * h2) + b3[o];
}
template <int IN, int OUT, int ACT, int KIND>
__global__ __launch_bounds__(64) void k_rollout_h64(RolloutArgs a_single, const RolloutArgs* __restrict__ multi) {
  __shared__ __attribute__((aligned(16))) float sh[64];
  // multi != NULL: the rollouts of several independent samplers (own policy, envs, buffer) in one launch; problem r = blockIdx / E
  const RolloutArgs a = multi ? multi[blockIdx.x / a_single.E] : a_single;
  const int e = multi ? (int)(blockIdx.x % a_single.E) : (int)blockIdx.x, lane = threadIdx.x;
  const NetDesc& nd = a.nd;
  constexpr int SD = KIND == CRUX_ENV_CARTPOLE ? 4 : ((KIND == CRUX_ENV_SYNTH || KIND == CRUX_ENV_SYNTH_DISCRETE) ? IN : 2);
  float w1[IN], w2[64], w3[OUT], b3[OUT];
#pragma unroll
  for (int k = 0; k < IN; ++k) w1[k] = a.p[nd.woff[0] + lane + 64 * k];
#pragma unroll
  for (int k = 0; k < 64; ++k) w2[k] = a.p[nd.woff[1] + lane + 64 * k];
#pragma unroll
  for (int o = 0; o < OUT; ++o) { w3[o] = a.p[nd.woff[2] + o + OUT * lane]; b3[o] = a.p[nd.boff[2] + o]; }
  const float b1 = a.p[nd.boff[0] + lane], b2 = a.p[nd.boff[1] + lane];
  double st[ENV_MAXSD]; float x[IN], nx[ENV_MAXOBS];
#pragma unroll
  for (int i = 0; i < SD; ++i) st[i] = a.state[(size_t)e * SD + i];
  int64_t ep_len = a.ep_len[e], n_resets = a.n_resets[e], steps_taken = a.steps_taken[e]; double sum_r = 0.0; int64_t nee = 0;
#pragma unroll
  for (int k = 0; k < IN; ++k) x[k] = a.svec[(size_t)e * IN + k];
  for (int64_t t = 0; t < a.T; ++t) {
    const int64_t j = (a.base + (int64_t)e * a.T + t) % a.C;
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < IN; ++k) a.S[(size_t)j * IN + k] = x[k];
    }
    float z[OUT];
    h64_forward<IN, OUT, ACT>(w1, w2, w3, b3, b1, b2, x, sh, lane, z);
    rollout_tail(a, z, IN, OUT, OUT, KIND, e, t, j, lane == 0, st, ep_len, n_resets, steps_taken, sum_r, nee, nx, lane);
#pragma unroll
    for (int k = 0; k < IN; ++k) x[k] = nx[k];
  }
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < SD; ++i) a.state[(size_t)e * SD + i] = st[i];
    a.ep_len[e] = ep_len; a.n_resets[e] = n_resets; a.steps_taken[e] = steps_taken;
    a.acc[2 * e] = sum_r; a.acc[2 * e + 1] = (double)nee;
#pragma unroll
    for (int k = 0; k < IN; ++k) a.svec[(size_t)e * IN + k] = x[k];
  }
}

// one wave (64 lanes) per environment; lanes split the output units of each Dense layer, lane 0 runs the
// head, the dynamics and the bookkeeping. Environments never synchronise with each other.
// The body is a device function of ONE wave (its LDS traffic is ordered by a wave barrier, not a workgroup barrier), so that the small-network
// solve kernel below can run it on one of its waves.
#define RO_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
// NT = threads sharing the body: 64 (one wave, ordered by a wave barrier) or 256 (one workgroup per environment, ordered by __syncthreads: the wide-network rollout
// kernel below -- with 256-wide layers one wave needed 117 us per step). Thread 0 runs the tail either way; the layer arithmetic (fma over k ascending, + bias,
// activation) is the same, so the results are.
// Chain(Dense...) on ONE observation held in hbuf[0] (sampler.jl:73 -> policies.jl:94,120) by NT threads: thread o evaluates output unit o (fma over k ascending, + bias,
// activation); returns the index of the hbuf half that holds the outputs. Shared by the generic rollout kernels and k_explore_generic.
template <int NT>
__device__ __forceinline__ int rollout_generic_forward(const RolloutArgs& a, const int lane, float (*hbuf)[1024]) {
    const NetDesc& nd = a.nd;
    int cur = 0;
    for (int l = 0; l < nd.L; ++l) {
      const int in = nd.dims[l], out = nd.dims[l + 1], act = nd.acts[l];
      const float* Wl = a.p + nd.woff[l]; const float* bl = a.p + nd.boff[l];
      for (int o = lane; o < out; o += NT) {
        float accv = 0.f; int k = 0;
        if (NT > 64) {                         // wide layers (one workgroup per environment): 32 independent weight loads in flight -- a batch-1 layer is a chain of L2 round trips,
          for (; k + 32 <= in; k += 32) {      // and its length is in / (loads in flight); the fma chain keeps its order, so the result keeps its bits
            float wv[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) wv[u] = Wl[o + out * (k + u)];
#pragma unroll
            for (int u = 0; u < 32; ++u) accv = fmaf(wv[u], hbuf[cur][k + u], accv); } }
        for (; k + 8 <= in; k += 8) {          // eight independent weight loads in flight; the fma chain keeps its order
          float wv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) wv[u] = Wl[o + out * (k + u)];
#pragma unroll
          for (int u = 0; u < 8; ++u) accv = fmaf(wv[u], hbuf[cur][k + u], accv); }
        for (; k < in; ++k) accv = fmaf(Wl[o + out * k], hbuf[cur][k], accv);
        hbuf[cur ^ 1][o] = crux_act(act, accv + bl[o]);
      }
      if (NT == 64) { RO_WAVE_SYNC(); } else { __syncthreads(); }
      cur ^= 1;
    }
    return cur;
}
template <int NT = 64>
__device__ __forceinline__ void rollout_generic_wave(const RolloutArgs& a, const int e, const int lane, float (*hbuf)[1024], float* sh_misc) {
#define RO_SYNC() do { if (NT == 64) { RO_WAVE_SYNC(); } else { __syncthreads(); } } while (0)
  const NetDesc& nd = a.nd;
  const int od = a.od, ad = a.ad, nout = nd.dims[nd.L];
  double st[ENV_MAXSD]; int64_t ep_len = 0, n_resets = 0, steps_taken = 0; double sum_r = 0.0; int64_t nee = 0;
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < ENV_MAXSD; ++i) if (i < a.sd) st[i] = a.state[(size_t)e * a.sd + i];      // constant indices: the state array stays in registers
    ep_len = a.ep_len[e]; n_resets = a.n_resets[e]; steps_taken = a.steps_taken[e];
  }
  if (lane < od) hbuf[0][lane] = a.svec[(size_t)e * od + lane];
  RO_SYNC();
  for (int64_t t = 0; t < a.T; ++t) {
    const int64_t j = (a.base + (int64_t)e * a.T + t) % a.C;
    // current observation -> S column (sampler.jl:100)
    if (lane < od) a.S[(size_t)j * od + lane] = hbuf[0][lane];
    // ---- policy forward: Chain(Dense...) on a batch of one (sampler.jl:73 -> policies.jl:94,120)
    const int cur = rollout_generic_forward<NT>(a, lane, hbuf);
    // (the observation in hbuf[0] was already stored to the S column, so the ping-pong may overwrite it)
    // the tail with compile-time kind and dimensions for the restated environments (its per-step arrays are then registers; with run-time dimensions they are
    // private memory and the step costs ~3x more), the run-time form for everything else (SYNTH envs of any width)
    if (lane == 0) {
      if (a.kind == CRUX_ENV_CARTPOLE && od == 4 && ad == 2 && nout == 2) rollout_tail(a, hbuf[cur], 4, 2, 2, CRUX_ENV_CARTPOLE, e, t, j, true, st, ep_len, n_resets, steps_taken, sum_r, nee, sh_misc);
      else if (a.kind == CRUX_ENV_GRIDWORLD && od == 2 && ad == 4 && nout == 4) rollout_tail(a, hbuf[cur], 2, 4, 4, CRUX_ENV_GRIDWORLD, e, t, j, true, st, ep_len, n_resets, steps_taken, sum_r, nee, sh_misc);
      else if (a.kind == CRUX_ENV_PENDULUM && od == 3 && ad == 1 && nout == 1) rollout_tail(a, hbuf[cur], 3, 1, 1, CRUX_ENV_PENDULUM, e, t, j, true, st, ep_len, n_resets, steps_taken, sum_r, nee, sh_misc);
      else if (a.kind == CRUX_ENV_SYNTH_DISCRETE && od == 8 && ad == 4 && nout == 4) rollout_tail(a, hbuf[cur], 8, 4, 4, CRUX_ENV_SYNTH_DISCRETE, e, t, j, true, st, ep_len, n_resets, steps_taken, sum_r, nee, sh_misc);
      else rollout_tail(a, hbuf[cur], od, ad, nout, a.kind, e, t, j, true, st, ep_len, n_resets, steps_taken, sum_r, nee, sh_misc);
    }
    RO_SYNC();
    if (lane < od) hbuf[0][lane] = sh_misc[lane];
    RO_SYNC();
  }
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < ENV_MAXSD; ++i) if (i < a.sd) a.state[(size_t)e * a.sd + i] = st[i];
    a.ep_len[e] = ep_len; a.n_resets[e] = n_resets; a.steps_taken[e] = steps_taken;
    a.acc[2 * e] = sum_r; a.acc[2 * e + 1] = (double)nee;
  }
  if (lane < od) a.svec[(size_t)e * od + lane] = hbuf[0][lane];
}

__global__ __launch_bounds__(64) void k_rollout(RolloutArgs a) {
  __shared__ float hbuf[2][1024];
  __shared__ float sh_misc[ENV_MAXOBS + 8];
  rollout_generic_wave(a, blockIdx.x, threadIdx.x, hbuf, sh_misc);
}
// one WORKGROUP of 256 threads per environment: thread o evaluates output unit o of a layer (weights W[o + out k] coalesced over the threads, the previous layer
// broadcast from LDS), thread 0 runs the head, the dynamics and the bookkeeping. For policies with layers of 128 units or more.
__global__ __launch_bounds__(256) void k_rollout_wide(RolloutArgs a) {
  __shared__ float hbuf[2][1024];
  __shared__ float sh_misc[ENV_MAXOBS + 8];
  rollout_generic_wave<256>(a, blockIdx.x, threadIdx.x, hbuf, sh_misc);
}

// ---- register-resident WIDE policy: IN -> H -> H -> OUT with H = 256 or 128 (the off-policy configurations C3 / C4: 8->256->256->4, 3->256->256->1) -------------------------
// One workgroup of 256 threads (one wave per SIMD: up to 512 registers per thread) per environment. Thread o keeps row o of W1 and of W2 -- H weights of the second layer -- in
// registers for the whole launch and evaluates unit o of both hidden layers with the generic kernels' arithmetic (fma over k ascending, + bias, activation: the same bits); the
// output layer's weights sit in LDS. A batch-1 layer streamed from the L2 is a chain of H / (loads in flight) round trips per step (k_rollout_wide: 20 us per step at H = 256);
// from registers it is H fma. steps!(sampler, buffer; Nsteps = dN) of an off-policy solve calls this with T = dN = 4..50 steps per launch.
__device__ __forceinline__ void env_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }      // LDS traffic only: does not wait for the column stores
__device__ __forceinline__ float env_readlane(const float v, const int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
// acc = fma(w[i], h[L0 + i], acc) for i = 0..7, h[l] = lane l of hreg: v_readlane into a scalar register, v_fmac with it as the scalar operand (the product commutes: the bits
// of fmaf(w, h, acc)). Written out because the scheduler, left alone, hoists every v_readlane of a layer and then spills the scalar registers they fill.
template <int L0>
__device__ __forceinline__ void env_rl_fma8(float& acc, const float hreg, const float w0, const float w1, const float w2, const float w3, const float w4, const float w5,
                                            const float w6, const float w7) {
  int t0, t1;
  asm volatile("v_readlane_b32 %[t0], %[h], %[l0]\n\tv_readlane_b32 %[t1], %[h], %[l1]\n\t"
               "v_fmac_f32_e32 %[acc], %[t0], %[w0]\n\tv_readlane_b32 %[t0], %[h], %[l2]\n\t"
               "v_fmac_f32_e32 %[acc], %[t1], %[w1]\n\tv_readlane_b32 %[t1], %[h], %[l3]\n\t"
               "v_fmac_f32_e32 %[acc], %[t0], %[w2]\n\tv_readlane_b32 %[t0], %[h], %[l4]\n\t"
               "v_fmac_f32_e32 %[acc], %[t1], %[w3]\n\tv_readlane_b32 %[t1], %[h], %[l5]\n\t"
               "v_fmac_f32_e32 %[acc], %[t0], %[w4]\n\tv_readlane_b32 %[t0], %[h], %[l6]\n\t"
               "v_fmac_f32_e32 %[acc], %[t1], %[w5]\n\tv_readlane_b32 %[t1], %[h], %[l7]\n\t"
               "v_fmac_f32_e32 %[acc], %[t0], %[w6]\n\tv_fmac_f32_e32 %[acc], %[t1], %[w7]"
               : [acc] "+v"(acc), [t0] "=&s"(t0), [t1] "=&s"(t1)
               : [h] "v"(hreg), [w0] "v"(w0), [w1] "v"(w1), [w2] "v"(w2), [w3] "v"(w3), [w4] "v"(w4), [w5] "v"(w5), [w6] "v"(w6), [w7] "v"(w7),
                 [l0] "n"(L0), [l1] "n"(L0 + 1), [l2] "n"(L0 + 2), [l3] "n"(L0 + 3), [l4] "n"(L0 + 4), [l5] "n"(L0 + 5), [l6] "n"(L0 + 6), [l7] "n"(L0 + 7));
}
template <int H, int K0 = 0>
__device__ __forceinline__ void env_rl_layer(float& acc, const float (&hr)[H / 64], const float (&w)[H]) {
  if constexpr (K0 < H) {
    env_rl_fma8<(K0 & 63)>(acc, hr[K0 >> 6], w[K0], w[K0 + 1], w[K0 + 2], w[K0 + 3], w[K0 + 4], w[K0 + 5], w[K0 + 6], w[K0 + 7]);
    env_rl_layer<H, K0 + 8>(acc, hr, w);
  }
}
template <int H, int K0 = 0>
__device__ __forceinline__ void env_rl_layer_lds(float& acc, const float (&hr)[H / 64], const float* wrow) {      // the weights of the row from LDS, 32 at a time
  if constexpr (K0 < H) {
    float4 wv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) wv[q] = *(const float4*)&wrow[K0 + 4 * q];
    env_rl_fma8<(K0 & 63)>(acc, hr[K0 >> 6], wv[0].x, wv[0].y, wv[0].z, wv[0].w, wv[1].x, wv[1].y, wv[1].z, wv[1].w);
    env_rl_fma8<((K0 + 8) & 63)>(acc, hr[K0 >> 6], wv[2].x, wv[2].y, wv[2].z, wv[2].w, wv[3].x, wv[3].y, wv[3].z, wv[3].w);
    env_rl_fma8<((K0 + 16) & 63)>(acc, hr[K0 >> 6], wv[4].x, wv[4].y, wv[4].z, wv[4].w, wv[5].x, wv[5].y, wv[5].z, wv[5].w);
    env_rl_fma8<((K0 + 24) & 63)>(acc, hr[K0 >> 6], wv[6].x, wv[6].y, wv[6].z, wv[6].w, wv[7].x, wv[7].y, wv[7].z, wv[7].w);
    env_rl_layer_lds<H, K0 + 32>(acc, hr, wrow);
  }
}
template <int H>
__global__ __launch_bounds__(256) void k_rollout_res(RolloutArgs a) {
  __shared__ __attribute__((aligned(16))) float hb[2][H];      // the hidden activations of the two layers
  __shared__ __attribute__((aligned(16))) float w3s[4 * (H + 4)];      // row o of W3 contiguous (+4: the nout <= 4 rows start in different banks)
  __shared__ float xin[ENV_MAXOBS], smu[8], ssg[8];
  const int e = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
#ifdef CRUX_RES_TIMING
  const long long t_entry = wall_clock64();
#endif
  const NetDesc& nd = a.nd; const int od = a.od, nout = nd.dims[3];
  const bool on = tid < H;
  float w1[8], w2[H];      // (obs_dim <= 8 for the shapes dispatched here)
#pragma unroll
  for (int k = 0; k < 8; ++k) w1[k] = (on && k < od) ? a.p[nd.woff[0] + tid + H * k] : 0.f;
  // (element by element: 256-byte loads per wave, ~64 KB in flight per compute unit -- 10 us for the 256 KB of H = 256. Wider pieces dealt to their owners through LDS need the
  //  registers twice or > 64 KB of staging to keep enough bytes in flight; measured slower within the static 64 KB)
#pragma unroll
  for (int k = 0; k < H; ++k) w2[k] = on ? a.p[nd.woff[1] + tid + H * k] : 0.f;
  const float b1 = on ? a.p[nd.boff[0] + tid] : 0.f, b2 = on ? a.p[nd.boff[1] + tid] : 0.f;
  for (int q = tid; q < nout * H; q += 256) w3s[(q % nout) * (H + 4) + q / nout] = a.p[nd.woff[2] + q];      // W3[o + nout k] in the flat vector -> row o
  const int row3 = lane < nout ? lane : 0;
  const float b3 = a.p[nd.boff[2] + row3];
  const int act1 = nd.acts[0], act2 = nd.acts[1], act3 = nd.acts[2];
  // wave 0 carries the sampler state, identically in every lane (the head, the dynamics and the bookkeeping are evaluated redundantly; lane 0 writes)
  double st[ENV_MAXSD]; int64_t ep_len = 0, n_resets = 0, steps_taken = 0; double sum_r = 0.0; int64_t nee = 0;
  if (tid < 64) {
#pragma unroll
    for (int i = 0; i < ENV_MAXSD; ++i) if (i < a.sd) st[i] = a.state[(size_t)e * a.sd + i];
    ep_len = a.ep_len[e]; n_resets = a.n_resets[e]; steps_taken = a.steps_taken[e]; }
  if (tid < od) { xin[tid] = a.svec[(size_t)e * od + tid]; smu[tid] = a.mu[tid]; ssg[tid] = a.sigma[tid]; }
  __syncthreads();
#ifdef CRUX_RES_TIMING
  long long tk[6] = {0, 0, 0, 0, 0, 0}, t0 = wall_clock64(), t1;
#define RT(i) { t1 = wall_clock64(); tk[i] += t1 - t0; t0 = t1; }
#else
#define RT(i)
#endif
  int64_t j = (a.base + (int64_t)e * a.T) % a.C;      // the ring row of step t: (base + e T + t) % C, advanced instead of divided
  for (int64_t t = 0; t < a.T; ++t, j = j + 1 == a.C ? 0 : j + 1) {
    if (tid < od) a.S[(size_t)j * od + tid] = xin[tid];      // current observation -> S column (sampler.jl:100)
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) if (k < od) acc = fmaf(w1[k], xin[k], acc);
    if (on) hb[0][tid] = crux_act(act1, acc + b1);
    env_lds_barrier();
    RT(0)
    // layer 2: the H activations sit H / 64 per lane; v_readlane broadcasts h[k] as the scalar operand of the fma (an LDS broadcast read per k is a round trip per k)
    float hr[H / 64];
#pragma unroll
    for (int c = 0; c < H / 64; ++c) hr[c] = hb[0][lane + 64 * c];
    acc = 0.f;
    env_rl_layer<H>(acc, hr, w2);
    if (on) hb[1][tid] = crux_act(act2, acc + b2);
    env_lds_barrier();
    RT(1)
    if (tid < 64) {      // wave 0: the output layer (lane o = unit o; the other lanes repeat unit 0), then everything of step! after the forward pass
#pragma unroll
      for (int c = 0; c < H / 64; ++c) hr[c] = hb[1][lane + 64 * c];
      float z = 0.f;
      env_rl_layer_lds<H>(z, hr, &w3s[row3 * (H + 4)]);
      z = crux_act(act3, z + b3);
      RT(2)
      float zz[4], nx[ENV_MAXOBS];
      // (the host dispatches here only for these two environment shapes)
      if (a.kind == CRUX_ENV_PENDULUM) { zz[0] = env_readlane(z, 0);
        rollout_tail(a, zz, 3, 1, 1, CRUX_ENV_PENDULUM, e, t, j, tid == 0, st, ep_len, n_resets, steps_taken, sum_r, nee, nx, lane, smu, ssg);
        if (tid == 0) { xin[0] = nx[0]; xin[1] = nx[1]; xin[2] = nx[2]; } }
      else { zz[0] = env_readlane(z, 0); zz[1] = env_readlane(z, 1); zz[2] = env_readlane(z, 2); zz[3] = env_readlane(z, 3);
        rollout_tail(a, zz, 8, 4, 4, CRUX_ENV_SYNTH_DISCRETE, e, t, j, tid == 0, st, ep_len, n_resets, steps_taken, sum_r, nee, nx, lane, smu, ssg);
        if (tid == 0) {
#pragma unroll
          for (int q = 0; q < 8; ++q) xin[q] = nx[q]; } }
      RT(3)
    }
    env_lds_barrier();
    RT(4)
  }
#ifdef CRUX_RES_TIMING
  if (tid == 0 && e == 0) printf("[res-timing] prologue %lld\n", t0 - t_entry - tk[0] - tk[1] - tk[2] - tk[3] - tk[4]);
  if (tid == 0 && e == 0) printf("[res-timing] T=%lld ticks: l1 %lld l2 %lld l3 %lld tail %lld bar %lld\n", (long long)a.T, tk[0], tk[1], tk[2], tk[3], tk[4]);
#endif
#undef RT
  if (tid == 0) {
#pragma unroll
    for (int i = 0; i < ENV_MAXSD; ++i) if (i < a.sd) a.state[(size_t)e * a.sd + i] = st[i];
    a.ep_len[e] = ep_len; a.n_resets[e] = n_resets; a.steps_taken[e] = steps_taken;
    a.acc[2 * e] = sum_r; a.acc[2 * e + 1] = (double)nee; }
  if (tid < od) a.svec[(size_t)e * od + tid] = xin[tid];
  // the T rows just written get the maximal priority (experience_buffer.jl:254), in this launch (one environment: the rows are this workgroup's)
#ifdef CRUX_RES_TIMING
  const long long tp0 = wall_clock64();
#endif
  if (a.pt_pr) { if (a.pt_touch && a.T <= PUSH_SMALL_MAX && a.T <= a.C) push_touch_small(a.pt_ids, (int)a.T, a.base, a.C, a.pt_pr, a.pt_pminmax, a.pt_alpha, a.pt_N, a.pt_nlev, a.pt_run, a.pt_total);
    else push_touch_block(a.pt_ids, a.T, a.base, a.C, a.pt_pr, a.pt_pminmax, a.pt_alpha, a.pt_N, a.pt_nlev, a.pt_run, a.pt_total, a.pt_touch); }
#ifdef CRUX_RES_TIMING
  if (tid == 0 && e == 0) printf("[res-timing] push %lld ticks (touch %d)\n", wall_clock64() - tp0, a.pt_touch);
#endif
}

// ---- caller-stepped environments: step! with an arbitrary mdp on the host (sampler.jl:71-137) ----------------------------------------------------------------
// The reference's step! calls @gen(:sp,:r)(mdp, s, a) on whatever mdp the user handed to solve (sampler.jl:89-97). For environments that only exist on the host the
// device part of a step is the policy: exploration(pi_explore, svec; pi_on, i) / action(pi, svec) (sampler.jl:73) for the E current observations at once --
// a.svec = observations [od x E], a.steps_taken = draws each sampler has made so far, a.cfg.i0 = the interaction counter of sampler 0; outputs a.A (one-hot bytes or
// Float32 [ad x E]) and a.LP [E]. The forward pass and the head are the rollout kernels' own (h64_forward / rollout_generic_forward, rollout_head): a caller that steps the
// restated CartPole on the host gets the transitions crux_rollout produces, bit for bit (tests/test_gpu_host_env.py).
__device__ __forceinline__ void explore_store(const RolloutArgs& a, const int e, const int ad, const float* aout, const float logprob) {
  if (a.act_kind == CRUX_ACTION_DISCRETE) { uint8_t* A = (uint8_t*)a.A + (size_t)e * ad; for (int q = 0; q < ad; ++q) A[q] = aout[q] != 0.f; }
  else { float* A = (float*)a.A + (size_t)e * ad; for (int q = 0; q < ad; ++q) A[q] = aout[q]; }
  a.LP[e] = logprob;
}
template <int IN, int OUT, int ACT>
__global__ __launch_bounds__(64) void k_explore_h64(RolloutArgs a) {
  __shared__ __attribute__((aligned(16))) float sh[64];
  const int e = blockIdx.x, lane = threadIdx.x;
  const NetDesc& nd = a.nd;
  float w1[IN], w2[64], w3[OUT], b3[OUT], x[IN], z[OUT];
#pragma unroll
  for (int k = 0; k < IN; ++k) w1[k] = a.p[nd.woff[0] + lane + 64 * k];
#pragma unroll
  for (int k = 0; k < 64; ++k) w2[k] = a.p[nd.woff[1] + lane + 64 * k];
#pragma unroll
  for (int o = 0; o < OUT; ++o) { w3[o] = a.p[nd.woff[2] + o + OUT * lane]; b3[o] = a.p[nd.boff[2] + o]; }
  const float b1 = a.p[nd.boff[0] + lane], b2 = a.p[nd.boff[1] + lane];
#pragma unroll
  for (int k = 0; k < IN; ++k) x[k] = a.svec[(size_t)e * IN + k];
  h64_forward<IN, OUT, ACT>(w1, w2, w3, b3, b1, b2, x, sh, lane, z);
  float aout[ENV_MAXOBS]; int ai; float logprob;
  rollout_head(a, z, OUT, OUT, e, a.cfg.i0 + (uint64_t)e, (uint64_t)a.steps_taken[e], aout, ai, logprob);
  if (lane == 0) explore_store(a, e, OUT, aout, logprob);
}
template <int NT>
__global__ __launch_bounds__(NT) void k_explore_generic(RolloutArgs a) {
  __shared__ float hbuf[2][1024];
  const int e = blockIdx.x, lane = threadIdx.x, od = a.od, nout = a.nd.dims[a.nd.L];
  if (lane < od) hbuf[0][lane] = a.svec[(size_t)e * od + lane];
  if (NT == 64) { RO_WAVE_SYNC(); } else { __syncthreads(); }
  const int cur = rollout_generic_forward<NT>(a, lane, hbuf);
  if (lane == 0) { float aout[ENV_MAXOBS]; int ai; float logprob;
    rollout_head(a, hbuf[cur], a.ad, nout, e, a.cfg.i0 + (uint64_t)e, (uint64_t)a.steps_taken[e], aout, ai, logprob);
    explore_store(a, e, a.ad, aout, logprob); }
}

// ---- the whole off-policy solve loop of a small network in ONE launch ------------------------------------------------------------------------
// solve(::OffPolicySolver) (src/model_free/off_policy.jl:133-147) for the DQN family on networks that fit one workgroup (the README example: DQN on
// SimpleGridWorld, 2-8-4, dN = 4, B = 128): per iteration steps!(dN) -> dN epochs of {rand! (uniform) -> dqn_target -> train!(td_loss)} ->
// polyak_average!. Launched piece by piece this is ~10 kernels and several host round trips per gradient step for a few hundred flops of work. Here
// one workgroup runs `iters` iterations back to back: wave e steps environment e (the generic rollout body above), then all four waves run the epochs
// (the uniform draw and gather, the target network through mlp_forward_run, the step through train_generic_run -- the bodies the separate calls
// use, so the results are the same bits), then the target update. Per-epoch info rows are left in device memory for the host to average.
struct SmallSolveArgs {
  RolloutArgs ro;                 // steps! of dN transitions per iteration into the replay ring (base / cfg.i0 advance in the kernel)
  TrainArgs tr;                   // train!(pi, td_loss) on the B rows of the staging buffer (explicit rows 0..B-1)
  NetDesc ndt; float* pt;         // pi_minus
  // staging buffer <- replay buffer gather table (rand!: push!(target, source, ids))
  void* gdst[CRUX_NCOLS]; const void* gsrc[CRUX_NCOLS]; int32_t gre[CRUX_NCOLS]; int32_t gesz[CRUX_NCOLS]; int32_t gn;
  const float* bSP; const float* bR; const uint8_t* bD;     // staging columns the target reads
  int64_t* bidx;                  // staging buffer's indices (device)
  float* qtmp; float* y;          // [nout x B], [B]
  int64_t elements, next, C;      // replay ring state at the first iteration
  int32_t B, epochs, iters, dN, E;
  float gamma, tau;
  uint64_t sample_seed; uint32_t sample_stream; uint64_t i0;
  float* infos;                   // [iters x epochs x 4]: loss, grad norm, Qavg, status
  int32_t* status;
  int32_t lds_train, lds_fwd;     // floats
};
__global__ __launch_bounds__(256) void k_dqn_small_solve(SmallSolveArgs q) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __shared__ double red[4];
  __shared__ int32_t st_s[16];
  float* sm_train = sm; float* sm_fwd = sm + q.lds_train; float* sm_ro = sm_fwd + q.lds_fwd;      // rollout: per wave hbuf[2][1024] + misc
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nout = q.ndt.dims[q.ndt.L], npar = q.tr.nd.n_params, B = q.B;
  float* P = q.tr.p; float* G = q.tr.g; float* M = q.tr.m; float* V = q.tr.v; float* PT = q.pt; double* BP = q.tr.bp;
  const float* bS = q.tr.S; const void* bA = q.tr.A; const float* bSP = q.bSP; const float* bR = q.bR; const uint8_t* bD = q.bD; float* Y = q.y; float* QT = q.qtmp; const int32_t* IDS = q.tr.ids;
  if (tid < 16) st_s[tid] = 0;
  __syncthreads();
  int64_t elements = q.elements, next = q.next; int err = 0;
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl = __builtin_amdgcn_s_memtime();
#define SS_T(k_) do { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); tacc[k_] += tn_ - tl; tl = tn_; } while (0)
  for (int it = 0; it < q.iters && !err; ++it) {
    const uint64_t si = q.i0 + (uint64_t)it * (uint64_t)q.dN;             // S.i of this iteration (off_policy.jl:134)
    // ---- steps!(sampler, buffer, Nsteps = dN, explore = true, i = S.i) (:138)
    if (wv < q.E) { RolloutArgs ro = q.ro; ro.base = next; ro.cfg.i0 = si; ro.p = P;
      rollout_generic_wave(ro, wv, lane, (float (*)[1024])(sm_ro + (size_t)wv * (2 * 1024 + ENV_MAXOBS + 8)), sm_ro + (size_t)wv * (2 * 1024 + ENV_MAXOBS + 8) + 2 * 1024); }
    next = (next + q.dN) % q.C; elements = elements + q.dN < q.C ? elements + q.dN : q.C;
    __threadfence_block(); __syncthreads(); SS_T(0);
    // ---- value_training (:66-111)
    for (int ep = 0; ep < q.epochs; ++ep) {
      const uint64_t ictr = si * (uint64_t)q.epochs + (uint64_t)ep;
      // rand!(D, buffer): uniform_sample! (experience_buffer.jl:317-321) with the library's Philox draw (per.hip k_uniform_ids), then the row gather
      for (int j = tid; j < B; j += 256) { const crux_u32x4 x = crux_philox(q.sample_seed, ictr * (uint64_t)B + (uint64_t)j, q.sample_stream, CRUX_RNG_SAMPLE);
        q.bidx[j] = (int64_t)(((uint64_t)x.v[0] * (uint64_t)elements) >> 32); }
      __threadfence_block(); __syncthreads(); SS_T(1);
      for (int k = 0; k < q.gn; ++k) { const int re = q.gre[k];
        for (int t = tid; t < B * re; t += 256) { const int j = t / re, e2 = t - j * re; const int64_t sidx = q.bidx[j] * re + e2;
          if (q.gesz[k] == 4) ((uint32_t*)q.gdst[k])[t] = ((const uint32_t*)q.gsrc[k])[sidx];
          else ((uint8_t*)q.gdst[k])[t] = ((const uint8_t*)q.gsrc[k])[sidx]; } }
      __threadfence_block(); __syncthreads(); SS_T(2);
      // y = dqn_target(pi_minus, D) (dqn.jl:4-6)
      mlp_forward_run(q.ndt, PT, bSP, B, QT, sm_fwd, 0u, 1u);
      __threadfence_block(); __syncthreads();
      for (unsigned b = 0; b * 256u < (unsigned)B; ++b) DqnTargetOp::run(b, 1u, QT, nout, bR, bD, q.gamma, (int64_t)B, Y);
      __threadfence_block(); __syncthreads(); SS_T(3);
      // train!(pi, td_loss) (training.jl:13-25)
      { TrainArgs tr = q.tr; tr.p = P; tr.g = G; tr.m = M; tr.v = V; tr.bp = BP; tr.S = bS; tr.A = bA; tr.Y = Y; tr.ids = IDS; tr.status = st_s;
        tr.epoch_infos = q.infos + ((size_t)it * q.epochs + ep) * CRUX_INFO_N; train_generic_run(tr, sm_train, red); }
      __threadfence_block(); __syncthreads(); SS_T(4);
      if (st_s[0] != 0) { err = st_s[0]; break; }
    }
    if (err) break;
    // target_update: polyak_average!(pi_minus, pi, tau) (:108, policies.jl:48-59)
    { const float omt = __fsub_rn(1.0f, q.tau);
      for (int i = tid; i < npar; i += 256) PT[i] = __fadd_rn(__fmul_rn(q.tau, P[i]), __fmul_rn(omt, PT[i])); }
    __threadfence_block(); __syncthreads();
  }
  if (tid == 0) { q.status[0] = st_s[0]; q.status[8] = err; for (int z = 0; z < 8; ++z) ((unsigned long long*)(q.status + 16))[z] = tacc[z]; }
}

typedef float f32x4_env __attribute__((ext_vector_type(4)));
// ---- the same solve loop for a TINY network, wave-resident ----------------------------------------------------------------------------------
// The README example's DQN (SimpleGridWorld, 2 -> 8 -> 4, 60 parameters, B = 128) is a few hundred flops per gradient step: in k_dqn_small_solve the step
// still cost 68 us, almost all of it workgroup barriers and Float64 block reductions of the shape-generic learner body -- one host core does the step in
// 21 us. Here ONE wave owns the whole problem: parameter i, its Adam moments and the target network's copy live in lane i's registers and are broadcast with
// v_readlane; the replay ring is mirrored in LDS (the rollout body still writes the global columns, the new rows are copied over after each steps!);
// lane j evaluates samples j and j + 64 of the minibatch (target network, Q network, td_loss and the per-sample parameter gradients in registers), the
// 60 x 64 partial gradients are transposed through LDS so that lane i adds parameter i's contributions in lane order, and applies Flux's Adam (Float64 per
// element, like the generic body). No workgroup barrier exists in the loop. Same mathematics as the call-by-call loop with a different (fixed) summation
// order of the minibatch gradient: parity against the oracle is to float tolerance (tests/test_gpu_components.py: the solve loop keeps the oracle's
// trajectories and replay contents exactly, the networks to 1e-5), not bit-identity with k_dqn_small_solve, which stays as the shape-generic form.
template <int IN, int H, int OUT>
__global__ __launch_bounds__(64) void k_dqn_tiny_solve(SmallSolveArgs q) {
  constexpr int NP = H * IN + H + OUT * H + OUT, oW1 = 0, oB1 = H * IN, oW2 = oB1 + H, oB2 = oW2 + OUT * H;
  static_assert(NP <= 64 && OUT == 4, "one parameter per lane; the one-hot action row is read as one 32-bit word");
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x; const int B = q.B; const int64_t C = q.C;
  float* sm_ro = sm;                                                   // rollout body: hbuf[2][1024] + misc
  float* mS = sm_ro + (2 * 1024 + ENV_MAXOBS + 8);                     // ring mirror: S [C x IN], SP [C x IN], R [C], A (action index) [C], D [C]
  float* mSP = mS + C * IN; float* mR = mSP + C * IN; int32_t* mA = (int32_t*)(mR + C); int32_t* mD = mA + C;
  float* red = (float*)(mD + C);                                       // [NP][64] transpose buffer
  const float* gS = q.ro.S; const float* gSP = q.ro.SP; const float* gR = q.ro.R; const uint8_t* gA = (const uint8_t*)q.ro.A; const uint8_t* gD = q.ro.D;
  auto mirror_row = [&](int64_t row) {
    for (int k = 0; k < IN; ++k) { mS[row * IN + k] = gS[row * IN + k]; mSP[row * IN + k] = gSP[row * IN + k]; }
    mR[row] = gR[row]; int ai = 0; for (int k = 0; k < OUT; ++k) ai = gA[row * OUT + k] ? k : ai; mA[row] = ai; mD[row] = gD[row] ? 1 : 0;
  };
  for (int64_t row = lane; row < q.elements; row += 64) mirror_row(row);
  __shared__ float th_s[64]; __shared__ double st_s[ENV_MAXSD]; __shared__ int64_t cnt_s[3]; __shared__ float sv_s[ENV_MAXOBS]; __shared__ double acc_s[2];
  if (lane == 0) { for (int i = 0; i < q.ro.sd; ++i) st_s[i] = q.ro.state[i];
    cnt_s[0] = q.ro.ep_len[0]; cnt_s[1] = q.ro.n_resets[0]; cnt_s[2] = q.ro.steps_taken[0]; acc_s[0] = q.ro.acc[0]; acc_s[1] = q.ro.acc[1]; }
  if (lane < q.ro.od) sv_s[lane] = q.ro.svec[lane];
  float p = lane < NP ? q.tr.p[lane] : 0.f, pt = lane < NP ? q.pt[lane] : 0.f, am = lane < NP ? q.tr.m[lane] : 0.f, av = lane < NP ? q.tr.v[lane] : 0.f;
  double bp1 = q.tr.bp[0], bp2 = q.tr.bp[1];
  int64_t elements = q.elements, next = q.next; int err = 0;
  const float invB = 1.0f / (float)B;
  auto W = [&](float reg, int i) -> float { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, reg), i)); };   // parameter i, wave-uniform
  auto forward = [&](float reg, const float (&x)[IN], float (&h)[H], float (&o)[OUT]) {
#pragma unroll
    for (int j = 0; j < H; ++j) { float acc = 0.f;
#pragma unroll
      for (int k = 0; k < IN; ++k) acc = fmaf(W(reg, oW1 + j + H * k), x[k], acc);
      const float z = acc + W(reg, oB1 + j); h[j] = z > 0.f ? z : 0.f; }
#pragma unroll
    for (int j = 0; j < OUT; ++j) { float acc = 0.f;
#pragma unroll
      for (int k = 0; k < H; ++k) acc = fmaf(W(reg, oW2 + j + OUT * k), h[k], acc);
      o[j] = acc + W(reg, oB2 + j); }
  };
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
  const bool grid = q.ro.kind == CRUX_ENV_GRIDWORLD && q.ro.od == IN && q.ro.ad == OUT;
  double gst[ENV_MAXSD]; float gx[IN]; int64_t g_ep_len = cnt_s[0], g_n_resets = cnt_s[1], g_steps_taken = cnt_s[2], g_nee = 0; double g_sum_r = 0.0;
#pragma unroll
  for (int i = 0; i < 2; ++i) gst[i] = st_s[i];
#pragma unroll
  for (int k = 0; k < IN; ++k) gx[k] = sv_s[k];
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl = __builtin_amdgcn_s_memtime();      // phase shares (CRUX_SMALL_SOLVE_TIMING): rollout | sample + forward / backward | transpose-reduce | statistics + Adam
#define TS_T(k_) do { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); tacc[k_] += tn_ - tl; tl = tn_; } while (0)
  for (int it = 0; it < q.iters && !err; ++it) {
    const uint64_t si = q.i0 + (uint64_t)it * (uint64_t)q.dN;
    // ---- steps!(sampler, buffer, Nsteps = dN, explore = true, i = S.i): the generic rollout body, with theta and the sampler's state (env state, episode
    // counters, current observation) pointed at LDS copies -- read from global memory they cost a dependent L2 round trip per layer and per call (the rollout was
    // 2/3 of the solve); the buffer columns it writes stay in global memory, where the call-by-call loop leaves them
    if (grid) {
      // SimpleGridWorld (the README example): the whole sampler lives in registers, every lane carrying an identical copy like the 64-wide rollout kernel does. The
      // policy forward reads theta from the lane registers (the arithmetic order of the generic body: fma over k ascending, + bias, relu), and the tail runs with
      // compile-time environment kind and dimensions, so that its per-step arrays are registers -- called with run-time dimensions they are private memory and
      // a step cost 7.4 us (2/3 of the whole solve).
      RolloutArgs ro = q.ro; ro.base = next; ro.cfg.i0 = si; g_sum_r = 0.0; g_nee = 0;
      for (int64_t t = 0; t < ro.T; ++t) {
        const int64_t j = (ro.base + t) % ro.C;
        if (lane == 0) {
#pragma unroll
          for (int k = 0; k < IN; ++k) ro.S[(size_t)j * IN + k] = gx[k]; }
        float h[H], z[OUT]; forward(p, gx, h, z);
        float nx[ENV_MAXOBS];
        rollout_tail(ro, z, IN, OUT, OUT, CRUX_ENV_GRIDWORLD, 0, t, j, lane == 0, gst, g_ep_len, g_n_resets, g_steps_taken, g_sum_r, g_nee, nx);
#pragma unroll
        for (int k = 0; k < IN; ++k) gx[k] = nx[k];
      }
    } else {
    if (lane < NP) th_s[lane] = p;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
    { RolloutArgs ro = q.ro; ro.base = next; ro.cfg.i0 = si;
      ro.p = th_s; ro.state = st_s; ro.ep_len = &cnt_s[0]; ro.n_resets = &cnt_s[1]; ro.steps_taken = &cnt_s[2]; ro.svec = sv_s; ro.acc = acc_s;
      rollout_generic_wave(ro, 0, lane, (float (*)[1024])sm_ro, sm_ro + 2 * 1024); }
    }
    TS_T(5);
    __threadfence();
    if (lane < q.dN) mirror_row((next + lane) % C);
    next = (next + q.dN) % C; elements = elements + q.dN < C ? elements + q.dN : C;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
    TS_T(0);
    for (int ep = 0; ep < q.epochs; ++ep) {
      const uint64_t ictr = si * (uint64_t)q.epochs + (uint64_t)ep;
      float g[NP];
#pragma unroll
      for (int k = 0; k < NP; ++k) g[k] = 0.f;
      double s_sq = 0.0, s_q = 0.0;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int j = lane + 64 * half;
        if (j < B) {
          // rand!(D, buffer): uniform_sample! with the library's Philox draw (per.hip k_uniform_ids)
          const crux_u32x4 xr = crux_philox(q.sample_seed, ictr * (uint64_t)B + (uint64_t)j, q.sample_stream, CRUX_RNG_SAMPLE);
          const int64_t row = (int64_t)(((uint64_t)xr.v[0] * (uint64_t)elements) >> 32);
          q.bidx[j] = row;
          float x[IN], xp[IN];
#pragma unroll
          for (int k = 0; k < IN; ++k) { x[k] = mS[row * IN + k]; xp[k] = mSP[row * IN + k]; }
          const float r = mR[row]; const int ai = mA[row]; const float nd = 1.f - (mD[row] ? 1.f : 0.f);
          // y = dqn_target(pi_minus, D) (dqn.jl:4-6): r + gamma (1 - done) max_a Q-(sp, a), un-fused
          float h[H], o[OUT]; forward(pt, xp, h, o);
          float mx = o[0];
#pragma unroll
          for (int k = 1; k < OUT; ++k) mx = o[k] > mx ? o[k] : mx;
          const float gn = q.gamma * nd; const float tt = gn * mx; const float y = r + tt;
          // td_loss (utils.jl:76-87): mean((Q(s, a) - y)^2) and its pullback
          forward(p, x, h, o);
          float Q = 0.f;
#pragma unroll
          for (int k = 0; k < OUT; ++k) Q += o[k] * (k == ai ? 1.f : 0.f);
          const float d = Q - y; s_sq += (double)(d * d); s_q += (double)Q;
          float dq[OUT], dh[H];
#pragma unroll
          for (int k = 0; k < OUT; ++k) dq[k] = (k == ai) ? 2.f * d * 1.f * invB : 0.f;
#pragma unroll
          for (int k = 0; k < H; ++k) { float acc = 0.f;
#pragma unroll
            for (int oo = 0; oo < OUT; ++oo) { g[oW2 + oo + OUT * k] = fmaf(dq[oo], h[k], g[oW2 + oo + OUT * k]); acc = fmaf(W(p, oW2 + oo + OUT * k), dq[oo], acc); }
            dh[k] = h[k] > 0.f ? acc : 0.f; }                                                                   // relu'(0) = 0
#pragma unroll
          for (int oo = 0; oo < OUT; ++oo) g[oB2 + oo] += dq[oo];
#pragma unroll
          for (int k = 0; k < IN; ++k)
#pragma unroll
            for (int oo = 0; oo < H; ++oo) g[oW1 + oo + H * k] = fmaf(dh[oo], x[k], g[oW1 + oo + H * k]);
#pragma unroll
          for (int oo = 0; oo < H; ++oo) g[oB1 + oo] += dh[oo];
        }
      }
      TS_T(1);
      // transpose: lane i adds parameter i's 64 partials in lane order
#pragma unroll
      for (int k = 0; k < NP; ++k) red[k * 64 + lane] = g[k];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
      float gi = 0.f;
      if (lane < NP) {
#pragma unroll
        for (int l4 = 0; l4 < 16; ++l4) { const f32x4_env v4 = *(const f32x4_env*)&red[lane * 64 + 4 * l4]; gi = ((((gi + v4[0]) + v4[1]) + v4[2]) + v4[3]); } }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
      TS_T(2);
      const double t_ssq = wave_sum_d(lane < NP ? (double)gi * (double)gi : 0.0), t_sq = wave_sum_d(s_sq), t_q = wave_sum_d(s_q);
      const float gnorm = (float)sqrt(t_ssq);
      if (lane == 0) { float* e = q.infos + ((size_t)it * q.epochs + ep) * CRUX_INFO_N;
        e[CRUX_INFO_LOSS] = (float)(t_sq / (double)B); e[CRUX_INFO_GRAD_NORM] = gnorm; e[2] = (float)(t_q / (double)B); }
      if (isnan(gnorm)) { err = CRUX_ENAN; break; }                                                               // training.jl:20 -- no update
      if (lane < NP) {   // Flux.update!(Adam) (training.jl:21), Float64 per element like the reference
        const double gd = (double)gi; const double b1 = q.tr.b1, b2 = q.tr.b2;
        const float mi = (float)(b1 * (double)am + (1.0 - b1) * gd);
        const float vi = (float)(b2 * (double)av + ((1.0 - b2) * gd) * gd);
        const float dd = (float)((double)mi / (1.0 - bp1) / (sqrt((double)vi / (1.0 - bp2)) + q.tr.eps) * q.tr.eta);
        am = mi; av = vi; p = p - dd; }
      bp1 *= q.tr.b1; bp2 *= q.tr.b2;
      TS_T(4);
    }
    if (err) break;
    { const float omt = __fsub_rn(1.0f, q.tau); pt = __fadd_rn(__fmul_rn(q.tau, p), __fmul_rn(omt, pt)); }          // polyak_average!(pi_minus, pi, tau) (:108)
  }
  // ---- leave everything where the call-by-call loop leaves it: networks, Adam state, and the staging batch = the last minibatch drawn
  if (lane < NP) { q.tr.p[lane] = p; q.pt[lane] = pt; q.tr.m[lane] = am; q.tr.v[lane] = av; q.tr.g[lane] = 0.f; }
  if (grid && lane == 0) { st_s[0] = gst[0]; st_s[1] = gst[1]; cnt_s[0] = g_ep_len; cnt_s[1] = g_n_resets; cnt_s[2] = g_steps_taken; acc_s[0] = g_sum_r; acc_s[1] = (double)g_nee;
#pragma unroll
    for (int k = 0; k < IN; ++k) sv_s[k] = gx[k]; }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
  if (lane == 0) { for (int i = 0; i < q.ro.sd; ++i) q.ro.state[i] = st_s[i];           // the sampler's state back where the call-by-call loop keeps it
    q.ro.ep_len[0] = cnt_s[0]; q.ro.n_resets[0] = cnt_s[1]; q.ro.steps_taken[0] = cnt_s[2]; q.ro.acc[0] = acc_s[0]; q.ro.acc[1] = acc_s[1]; }
  if (lane < q.ro.od) q.ro.svec[lane] = sv_s[lane];
  __threadfence();
  for (int k = 0; k < q.gn; ++k) { const int re = q.gre[k];
    for (int t = lane; t < B * re; t += 64) { const int j = t / re, e2 = t - j * re; const int64_t sidx = q.bidx[j] * re + e2;
      if (q.gesz[k] == 4) ((uint32_t*)q.gdst[k])[t] = ((const uint32_t*)q.gsrc[k])[sidx];
      else ((uint8_t*)q.gdst[k])[t] = ((const uint8_t*)q.gsrc[k])[sidx]; } }
  if (lane == 0) { q.tr.bp[0] = bp1; q.tr.bp[1] = bp2; q.status[0] = err; q.status[8] = err; for (int z = 0; z < 8; ++z) ((unsigned long long*)(q.status + 16))[z] = tacc[z]; }
}

__global__ void k_env_init(int kind, int E, int od, int sd, uint64_t seed, const float* mu, const float* sigma, double* state, int64_t* ep_len,
                           int64_t* n_resets, float* svec, int fresh) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int64_t nr = fresh ? 0 : n_resets[e];
  if (env_is_synth(kind)) {                                  // straight to memory: no per-thread arrays (observation i == state i)
    for (int i = 0; i < sd; ++i) { const crux_u32x4 x = crux_philox(seed, 16 * (uint64_t)nr + (uint64_t)(i >> 1), (uint32_t)e, CRUX_RNG_RESET);
      const double ui = (i & 1) ? crux_u32x2_to_f64(x.v[2], x.v[3]) : crux_u32x2_to_f64(x.v[0], x.v[1]); const double xi = __dadd_rn(-0.05, __dmul_rn(0.1, ui));
      state[(size_t)e * sd + i] = xi; svec[(size_t)e * od + i] = __fdiv_rn(__fsub_rn((float)xi, mu[i]), sigma[i]); }
    n_resets[e] = nr + 1; ep_len[e] = 0; return;
  }
  double st[4]; float o[4];
  st[0] = st[1] = st[2] = st[3] = 0.0; o[0] = o[1] = o[2] = o[3] = 0.f;
  env_draw_initial(kind, seed, (uint64_t)nr, (uint32_t)e, st);
#pragma unroll
  for (int i = 0; i < 4; ++i) if (i < sd) state[(size_t)e * sd + i] = st[i];      // static indices: keep the arrays in registers
  n_resets[e] = nr + 1; ep_len[e] = 0;
  env_obs(kind, st, o);
#pragma unroll
  for (int q = 0; q < 4; ++q) if (q < od) svec[(size_t)e * od + q] = __fdiv_rn(__fsub_rn(o[q], mu[q]), sigma[q]);
}

__global__ void k_env_step(int kind, int64_t n, const double* state, const void* action, const double* uniforms, double* next_state, float* obs, float* r, uint8_t* done) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  if (kind == CRUX_ENV_GRIDWORLD) {
    const uint8_t* a = (const uint8_t*)action + 4 * j; int act = 0; for (int q = 0; q < 4; ++q) if (a[q]) act = q;
    double s[2] = {state[2 * j], state[2 * j + 1]}, sn[2]; float rr; uint8_t dd;
    gridworld_step(s, act, uniforms ? uniforms[j] : 0.0, sn, &rr, &dd);
    next_state[2 * j] = sn[0]; next_state[2 * j + 1] = sn[1]; obs[2 * j] = (float)sn[0]; obs[2 * j + 1] = (float)sn[1]; r[j] = rr; done[j] = dd; return;
  }
  if (kind == CRUX_ENV_CARTPOLE) {
    const uint8_t* a = (const uint8_t*)action + 2 * j; double s[4], sn[4]; for (int i = 0; i < 4; ++i) s[i] = state[4 * j + i];
    float rr; uint8_t dd; cartpole_step(s, a[1] ? 1 : 0, sn, &rr, &dd);
    for (int i = 0; i < 4; ++i) { next_state[4 * j + i] = sn[i]; obs[4 * j + i] = (float)sn[i]; } r[j] = rr; done[j] = dd;
  } else {
    double s[2] = {state[2 * j], state[2 * j + 1]}, sn[2]; float rr; uint8_t dd; float o[3];
    pendulum_step(s, ((const float*)action)[j], sn, &rr, &dd); env_obs(kind, sn, o);
    next_state[2 * j] = sn[0]; next_state[2 * j + 1] = sn[1]; obs[3 * j] = o[0]; obs[3 * j + 1] = o[1]; obs[3 * j + 2] = o[2]; r[j] = rr; done[j] = dd;
  }
}

static void env_dims(int kind, int so, int sa, int* obs, int* act, int* sd) {
  switch (kind) {
    case CRUX_ENV_CARTPOLE: *obs = 4; *act = 2; *sd = 4; break;
    case CRUX_ENV_PENDULUM: *obs = 3; *act = 1; *sd = 2; break;
    case CRUX_ENV_GRIDWORLD: *obs = 2; *act = 4; *sd = 2; break;
    case CRUX_ENV_SYNTH: case CRUX_ENV_SYNTH_DISCRETE: *obs = so; *act = sa; *sd = so; break;
    default: *obs = so; *act = sa; *sd = 1; break;
  }
}

static void fill_rollout_args(RolloutArgs& a, crux_env* e, crux_mlp* policy, const crux_rollout_cfg* cfg, crux_buffer* buf, int64_t T) {
  a = RolloutArgs{};
  a.nd = policy->nd; a.p = policy->p; a.kind = e->kind; a.E = e->n_envs; a.max_steps = e->max_steps; a.od = e->obs_dim; a.ad = e->act_dim; a.sd = e->state_dim;
  a.act_kind = buf->act_kind; a.seed = e->seed; a.mu = e->mu; a.sigma = e->sigma; a.state = e->state; a.ep_len = e->ep_len; a.n_resets = e->n_resets;
  a.steps_taken = e->steps_taken; a.svec = e->svec; a.acc = e->acc;
  a.S = (float*)buf->col[CRUX_COL_S]; a.A = buf->col[CRUX_COL_A]; a.SP = (float*)buf->col[CRUX_COL_SP]; a.R = (float*)buf->col[CRUX_COL_R];
  a.D = (uint8_t*)buf->col[CRUX_COL_DONE]; a.EE = (uint8_t*)buf->col[CRUX_COL_EPISODE_END];
  a.LP = has_col(buf, CRUX_COL_LOGPROB) ? (float*)buf->col[CRUX_COL_LOGPROB] : nullptr; a.TT = has_col(buf, CRUX_COL_T) ? (int64_t*)buf->col[CRUX_COL_T] : nullptr;
  a.II = has_col(buf, CRUX_COL_I) ? (int64_t*)buf->col[CRUX_COL_I] : nullptr; a.W = has_col(buf, CRUX_COL_WEIGHT) ? (float*)buf->col[CRUX_COL_WEIGHT] : nullptr;
  a.RET = has_col(buf, CRUX_COL_RETURN) ? (float*)buf->col[CRUX_COL_RETURN] : nullptr; a.ADV = has_col(buf, CRUX_COL_ADVANTAGE) ? (float*)buf->col[CRUX_COL_ADVANTAGE] : nullptr;
  a.COST = has_col(buf, CRUX_COL_COST) ? (float*)buf->col[CRUX_COL_COST] : nullptr; a.CADV = has_col(buf, CRUX_COL_COST_ADVANTAGE) ? (float*)buf->col[CRUX_COL_COST_ADVANTAGE] : nullptr;
  a.CRET = has_col(buf, CRUX_COL_COST_RETURN) ? (float*)buf->col[CRUX_COL_COST_RETURN] : nullptr;
  a.base = buf->next_ind; a.C = buf->capacity; a.T = T; a.cfg = *cfg; a.squash = policy->squash;
}

extern "C" {

int32_t crux_env_create(crux_ctx* ctx, int32_t kind, int32_t n_envs, int32_t max_steps, float gamma, const float* obs_mu, const float* obs_sigma,
                        uint64_t seed, int32_t synth_obs_dim, int32_t synth_act_dim, crux_env** out) {
  if (!ctx || !out) return CRUX_EINVAL;
  if (kind < CRUX_ENV_CARTPOLE || kind > CRUX_ENV_SYNTH_DISCRETE) return crux_fail(ctx, CRUX_EUNSUP, "env kind %d has no device dynamics", kind);
  if ((kind == CRUX_ENV_SYNTH || kind == CRUX_ENV_SYNTH_DISCRETE) && (synth_obs_dim < 1 || synth_obs_dim > ENV_MAXOBS || synth_act_dim < 1 || synth_act_dim > ENV_MAXOBS))
    return crux_fail(ctx, CRUX_EINVAL, "synthetic env: obs/act dims (%d, %d) must be in 1..%d", synth_obs_dim, synth_act_dim, ENV_MAXOBS);
  if (n_envs < 1 || max_steps < 1) return crux_fail(ctx, CRUX_EINVAL, "env_create: n_envs=%d max_steps=%d", n_envs, max_steps);
  crux_env* e = new crux_env(); e->ctx = ctx; e->kind = kind; e->n_envs = n_envs; e->max_steps = max_steps; e->gamma = gamma; e->seed = seed;
  env_dims(kind, synth_obs_dim, synth_act_dim, &e->obs_dim, &e->act_dim, &e->state_dim);
  const int od = e->obs_dim;
  if (hipMalloc(&e->mu, 4 * od) != hipSuccess || hipMalloc(&e->sigma, 4 * od) != hipSuccess || hipMalloc(&e->state, 8 * (size_t)n_envs * e->state_dim) != hipSuccess ||
      hipMalloc(&e->ep_len, 8 * (size_t)n_envs) != hipSuccess || hipMalloc(&e->n_resets, 8 * (size_t)n_envs) != hipSuccess ||
      hipMalloc(&e->steps_taken, 8 * (size_t)n_envs) != hipSuccess || hipMalloc(&e->svec, 4 * (size_t)n_envs * od) != hipSuccess ||
      hipMalloc(&e->acc, 16 * (size_t)n_envs) != hipSuccess) { crux_env_destroy(e); return crux_fail(ctx, CRUX_ENOMEM, "env_create: hipMalloc"); }
  std::vector<float> mu(od, 0.f), sg(od, 1.f);
  if (obs_mu) for (int i = 0; i < od; ++i) mu[i] = obs_mu[i];
  if (obs_sigma) for (int i = 0; i < od; ++i) sg[i] = obs_sigma[i];
  HIPCHK(ctx, hipMemcpyAsync(e->mu, mu.data(), 4 * od, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(e->sigma, sg.data(), 4 * od, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(e->steps_taken, 0, 8 * (size_t)n_envs, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(e->acc, 0, 16 * (size_t)n_envs, ctx->stream));
  hipLaunchKernelGGL(k_env_init, dim3((n_envs + 63) / 64), dim3(64), 0, ctx->stream, kind, n_envs, od, e->state_dim, seed, e->mu, e->sigma, e->state, e->ep_len, e->n_resets, e->svec, 1);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  *out = e; return crux_launch_check(ctx, "k_env_init");
}

int32_t crux_env_destroy(crux_env* e) {
  if (!e) return CRUX_OK;
  crux_sync_before_free(e->ctx);
  (void)hipFree(e->mu); (void)hipFree(e->sigma); (void)hipFree(e->state); (void)hipFree(e->ep_len); (void)hipFree(e->n_resets);
  (void)hipFree(e->steps_taken); (void)hipFree(e->svec); (void)hipFree(e->acc);
  delete e; return CRUX_OK;
}
int32_t crux_env_obs_dim(const crux_env* e) { return e ? e->obs_dim : -1; }
int32_t crux_env_act_dim(const crux_env* e) { return e ? e->act_dim : -1; }
int32_t crux_env_state_dim(const crux_env* e) { return e ? e->state_dim : -1; }

int32_t crux_env_reset(crux_env* e) {
  if (!e) return CRUX_EINVAL;
  hipLaunchKernelGGL(k_env_init, dim3((e->n_envs + 63) / 64), dim3(64), 0, e->ctx->stream, e->kind, e->n_envs, e->obs_dim, e->state_dim, e->seed, e->mu, e->sigma,
                     e->state, e->ep_len, e->n_resets, e->svec, 0);
  return crux_launch_check(e->ctx, "k_env_init");
}

int32_t crux_env_get_state(crux_env* e, double* state, int64_t* ep_len, int64_t* n_resets) {
  if (!e) return CRUX_EINVAL;
  crux_ctx* c = e->ctx;
  if (state) HIPCHK(c, hipMemcpyAsync(state, e->state, 8 * (size_t)e->n_envs * e->state_dim, hipMemcpyDeviceToHost, c->stream));
  if (ep_len) HIPCHK(c, hipMemcpyAsync(ep_len, e->ep_len, 8 * (size_t)e->n_envs, hipMemcpyDeviceToHost, c->stream));
  if (n_resets) HIPCHK(c, hipMemcpyAsync(n_resets, e->n_resets, 8 * (size_t)e->n_envs, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return CRUX_OK;
}

int32_t crux_rollout(crux_env* e, crux_mlp* policy, const crux_rollout_cfg* cfg, crux_buffer* buf, int64_t T, double* sum_r, int64_t* n_episode_end) {
  if (!e || !policy || !cfg || !buf || T < 1) return CRUX_EINVAL;
  crux_ctx* c = e->ctx;
  const int64_t N = (int64_t)e->n_envs * T;
  if (buf->obs_dim != e->obs_dim || buf->act_dim != e->act_dim) return crux_fail(c, CRUX_EINVAL, "steps!: buffer columns (%d,%d) do not match the env (%d,%d)", buf->obs_dim, buf->act_dim, e->obs_dim, e->act_dim);
  if (N > buf->capacity) return crux_fail(c, CRUX_EINVAL, "steps!: %lld transitions exceed buffer capacity %lld", (long long)N, (long long)buf->capacity);
  if (policy->nd.dims[0] != e->obs_dim) return crux_fail(c, CRUX_EINVAL, "steps!: policy input %d != obs dim %d", policy->nd.dims[0], e->obs_dim);
  if (policy->nd.maxdim > 1024) return crux_fail(c, CRUX_EUNSUP, "steps!: layer width %d > 1024", policy->nd.maxdim);
  const int nout = policy->nd.dims[policy->nd.L];
  if ((cfg->head == CRUX_HEAD_CATEGORICAL || cfg->head == CRUX_HEAD_GREEDY_Q) && (nout != e->act_dim || buf->act_kind != CRUX_ACTION_DISCRETE))
    return crux_fail(c, CRUX_EINVAL, "steps!: discrete head needs %d logits and a one-hot action column", e->act_dim);
  if ((cfg->head == CRUX_HEAD_GAUSSIAN || cfg->head == CRUX_HEAD_DETERMINISTIC) && (nout != e->act_dim || buf->act_kind != CRUX_ACTION_CONTINUOUS))
    return crux_fail(c, CRUX_EINVAL, "steps!: continuous head needs %d outputs and a Float32 action column", e->act_dim);
  if (cfg->head == CRUX_HEAD_GAUSSIAN && policy->nd.n_extra != e->act_dim) return crux_fail(c, CRUX_EINVAL, "steps!: GaussianPolicy needs %d logSigma extras", e->act_dim);
  RolloutArgs a; fill_rollout_args(a, e, policy, cfg, buf, T);
  crux_prof_begin(c, CRUX_PROF_ROLLOUT);
  const NetDesc& pn = policy->nd;
  const bool h64 = pn.L == 3 && pn.dims[1] == 64 && pn.dims[2] == 64 && pn.acts[0] == pn.acts[1] && pn.acts[2] == CRUX_ACT_IDENTITY && !crux_sw().force_generic;
  bool push_done = false; int pt_touch = 0;
#define RO_CASE(I, O, A_, K) if (h64 && pn.dims[0] == I && nout == O && pn.acts[0] == A_ && e->kind == K) hipLaunchKernelGGL((k_rollout_h64<I, O, A_, K>), dim3(e->n_envs), dim3(64), 0, c->stream, a, (const RolloutArgs*)nullptr); else
  RO_CASE(4, 2, CRUX_ACT_RELU, CRUX_ENV_CARTPOLE)
  RO_CASE(4, 2, CRUX_ACT_TANH, CRUX_ENV_CARTPOLE)
  RO_CASE(3, 1, CRUX_ACT_RELU, CRUX_ENV_PENDULUM)
  RO_CASE(3, 1, CRUX_ACT_TANH, CRUX_ENV_PENDULUM)
  RO_CASE(17, 6, CRUX_ACT_TANH, CRUX_ENV_SYNTH)       // C5-shaped: 17 obs / 6 continuous actions
  RO_CASE(17, 6, CRUX_ACT_RELU, CRUX_ENV_SYNTH)
#undef RO_CASE
  // wide two-hidden-layer policies (C3 / C4): the register-resident kernel when the launch is long enough to pay for loading W2 into registers (a step costs 8 round trips
  // streamed; the load costs 8 once) -- always, from two steps on
  if (pn.L == 3 && pn.dims[1] == pn.dims[2] && (pn.dims[1] == 256 || pn.dims[1] == 128) && T >= 2 && !crux_sw().force_generic &&
      ((e->kind == CRUX_ENV_PENDULUM && e->obs_dim == 3 && e->act_dim == 1 && nout == 1) || (e->kind == CRUX_ENV_SYNTH_DISCRETE && e->obs_dim == 8 && e->act_dim == 4 && nout == 4))) {
    if (e->n_envs == 1 && crux_per_push_plan(buf, N, &pt_touch)) { push_done = true;
      a.pt_ids = buf->d_indices; a.pt_pr = buf->priorities; a.pt_pminmax = buf->pminmax; a.pt_run = buf->cumsum; a.pt_total = buf->topo_total; a.pt_alpha = buf->alpha;
      a.pt_nlev = (int32_t)buf->topo_levels; a.pt_touch = pt_touch; a.pt_N = (int64_t)buf->topo_n; }
    if (pn.dims[1] == 256) hipLaunchKernelGGL(k_rollout_res<256>, dim3(e->n_envs), dim3(256), 0, c->stream, a);
    else hipLaunchKernelGGL(k_rollout_res<128>, dim3(e->n_envs), dim3(256), 0, c->stream, a); }
  else if (policy->nd.maxdim >= 128) hipLaunchKernelGGL(k_rollout_wide, dim3(e->n_envs), dim3(256), 0, c->stream, a);
  else hipLaunchKernelGGL(k_rollout, dim3(e->n_envs), dim3(64), 0, c->stream, a);
  crux_prof_end(c, CRUX_PROF_ROLLOUT);
  int32_t rc = crux_launch_check(c, "k_rollout"); if (rc) return rc;
  if (push_done) crux_per_push_done(buf, pt_touch);
  else if (buf->prioritized) {      // push!: the new rows get max_priority (experience_buffer.jl:254); their ring rows are formed on the device, nothing to wait for
    if (crux_per_push_fused(buf, N, buf->d_indices)) { rc = crux_launch_check(c, "k_push_touch"); if (rc) return rc; }      // a few rows: the whole bookkeeping as one launch
    else {
      rc = crux_buffer_ring_ids_device(buf, N, buf->d_indices); if (rc) return rc;
      rc = crux_buffer_per_on_push(buf, buf->d_indices, N); if (rc) return rc; }
  }
  crux_buffer_ring_advance(buf, N);
  if (sum_r || n_episode_end) {
    std::vector<double> acc(2 * (size_t)e->n_envs);
    HIPCHK(c, hipMemcpyAsync(acc.data(), e->acc, 16 * (size_t)e->n_envs, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    double sr = 0; int64_t ne = 0; for (int k = 0; k < e->n_envs; ++k) { sr += acc[2 * k]; ne += (int64_t)acc[2 * k + 1]; }
    if (sum_r) *sum_r = sr; if (n_episode_end) *n_episode_end = ne;
  }
  return CRUX_OK;
}

// steps! for n independent samplers of equal shape in ONE launch (multi-seed runs): problem r rolls out its own policy on its own environments into its own buffer
// exploration(pi_explore, svec; pi_on, i) / action(pi, svec) of step! (sampler.jl:73) for n_envs caller-stepped samplers at once: see k_explore_* above.
int32_t crux_policy_explore(crux_mlp* policy, const crux_rollout_cfg* cfg, int32_t n_envs, const float* obs, uint64_t seed, const int64_t* steps_taken,
                            void* actions_out, float* logprob_out) {
  if (!policy || !cfg || !obs || !actions_out || n_envs < 1) return CRUX_EINVAL;
  crux_ctx* c = policy->ctx; const NetDesc& pn = policy->nd;
  if (pn.L < 1) return crux_fail(c, CRUX_EINVAL, "policy_explore: the policy has no layers");
  const int od = pn.dims[0], nout = pn.dims[pn.L];
  if (od > ENV_MAXOBS || nout > ENV_MAXOBS) return crux_fail(c, CRUX_EUNSUP, "policy_explore: observation / action widths (%d, %d) above %d", od, nout, ENV_MAXOBS);
  if (pn.maxdim > 1024) return crux_fail(c, CRUX_EUNSUP, "policy_explore: layer width %d > 1024", pn.maxdim);
  if (cfg->head < CRUX_HEAD_CATEGORICAL || cfg->head > CRUX_HEAD_DETERMINISTIC) return crux_fail(c, CRUX_EINVAL, "policy_explore: head %d", cfg->head);
  if (cfg->head == CRUX_HEAD_GAUSSIAN && pn.n_extra != nout) return crux_fail(c, CRUX_EINVAL, "policy_explore: GaussianPolicy needs %d logSigma extras", nout);
  const bool disc = cfg->head == CRUX_HEAD_CATEGORICAL || cfg->head == CRUX_HEAD_GREEDY_Q;
  const size_t E = (size_t)n_envs, b_obs = (4 * E * od + 255) / 256 * 256, b_st = (8 * E + 255) / 256 * 256, b_act = (E * nout * (disc ? 1 : 4) + 255) / 256 * 256, b_lp = (4 * E + 255) / 256 * 256;
  // ONE launch per call: observations, step counters, actions and log-probabilities live in a pinned, device-mapped block of the context -- the kernel
  // reads the E observations straight from host memory and writes the E actions back into it (a few hundred bytes to a few KB over the link, inside the launch) instead of
  // two staged uploads before and two read-backs after the kernel: five dependent stream operations (~60-80 us per call from pageable memory) become one.
  const bool zc = crux_sw().host_zerocopy;      // "0": the round-5 form (device scratch, two staged uploads, two read-backs) -- kept for the A/B of bench_hostenv.py and a bit-identity test
  char* hs = zc ? (char*)crux_pinned_mapped(c, b_obs + b_st + b_act + b_lp) : nullptr; if (zc && !hs) return crux_fail(c, CRUX_ENOMEM, "policy_explore: pinned staging");
  char* sc = zc ? (char*)c->pinned_mapped_dev : (char*)crux_scratch(c, b_obs + b_st + b_act + b_lp); if (!sc) return crux_fail(c, CRUX_ENOMEM, "policy_explore: scratch");
  RolloutArgs a = RolloutArgs{};
  a.nd = pn; a.p = policy->p; a.E = n_envs; a.od = od; a.ad = nout; a.act_kind = disc ? CRUX_ACTION_DISCRETE : CRUX_ACTION_CONTINUOUS; a.seed = seed;
  a.svec = (float*)sc; a.steps_taken = (int64_t*)(sc + b_obs); a.A = sc + b_obs + b_st; a.LP = (float*)(sc + b_obs + b_st + b_act); a.cfg = *cfg; a.squash = policy->squash;
  if (zc) { memcpy(hs, obs, 4 * E * od);
    if (steps_taken) memcpy(hs + b_obs, steps_taken, 8 * E); else memset(hs + b_obs, 0, 8 * E); }
  else { HIPCHK(c, hipMemcpyAsync(a.svec, obs, 4 * E * od, hipMemcpyHostToDevice, c->stream));
    if (steps_taken) HIPCHK(c, hipMemcpyAsync(a.steps_taken, steps_taken, 8 * E, hipMemcpyHostToDevice, c->stream));
    else HIPCHK(c, hipMemsetAsync(a.steps_taken, 0, 8 * E, c->stream)); }
  // the arithmetic of the rollout kernel that serves this policy shape (crux_rollout: the register-resident form for the instantiated IN-64-64-OUT shapes, else the generic one)
  const bool h64 = pn.L == 3 && pn.dims[1] == 64 && pn.dims[2] == 64 && pn.acts[0] == pn.acts[1] && pn.acts[2] == CRUX_ACT_IDENTITY && !crux_sw().force_generic;
#define EX_CASE(I, O, A_) if (h64 && od == I && nout == O && pn.acts[0] == A_) hipLaunchKernelGGL((k_explore_h64<I, O, A_>), dim3(n_envs), dim3(64), 0, c->stream, a); else
  EX_CASE(4, 2, CRUX_ACT_RELU) EX_CASE(4, 2, CRUX_ACT_TANH) EX_CASE(3, 1, CRUX_ACT_RELU) EX_CASE(3, 1, CRUX_ACT_TANH) EX_CASE(17, 6, CRUX_ACT_TANH) EX_CASE(17, 6, CRUX_ACT_RELU)
#undef EX_CASE
  if (pn.maxdim >= 128) hipLaunchKernelGGL(k_explore_generic<256>, dim3(n_envs), dim3(256), 0, c->stream, a);
  else hipLaunchKernelGGL(k_explore_generic<64>, dim3(n_envs), dim3(64), 0, c->stream, a);
  int32_t rc = crux_launch_check(c, "k_explore"); if (rc) return rc;
  if (!zc) { HIPCHK(c, hipMemcpyAsync(actions_out, a.A, E * nout * (disc ? 1 : 4), hipMemcpyDeviceToHost, c->stream));
    if (logprob_out) HIPCHK(c, hipMemcpyAsync(logprob_out, a.LP, 4 * E, hipMemcpyDeviceToHost, c->stream)); }
  HIPCHK(c, hipStreamSynchronize(c->stream));      // the one synchronisation of the call: the actions are what the caller steps its environment with
  if (zc) { memcpy(actions_out, hs + b_obs + b_st, E * nout * (disc ? 1 : 4));
    if (logprob_out) memcpy(logprob_out, hs + b_obs + b_st + b_act, 4 * E); }
  return CRUX_OK;
}

// the tail of steps! for a block the CALLER stepped (sampler.jl:139-155): push!(buffer, data) of the n transitions (ring write, experience_buffer.jl:232-259) and, on the ring rows
// just written, what terminate_episode! ran on the reference's `data` before the push (sampler.jl:53-66): fill_gae! / fill_returns!, the importance-weight products, the cost
// advantage / return -- for whichever of those columns the buffer has. cols[CRUX_COL_EPISODE_END] carries the caller's episode cuts (done, max_steps, the reset at the end).
int32_t crux_steps_push(crux_buffer* buf, int64_t n, const void* const* cols, int64_t rows_per_env, int32_t close_last, crux_mlp* critic, float lambda, float gamma,
                        crux_mlp* cost_critic, crux_mlp* nominal, int32_t nominal_head, int64_t* first_row_out) {
  if (!buf || !cols || n < 0) return CRUX_EINVAL;
  crux_ctx* c = buf->ctx;
  if (n > buf->capacity && (has_col(buf, CRUX_COL_ADVANTAGE) || has_col(buf, CRUX_COL_RETURN) || has_col(buf, CRUX_COL_COST_ADVANTAGE) || has_col(buf, CRUX_COL_COST_RETURN) ||
                            has_col(buf, CRUX_COL_FWD_IMPORTANCE_WEIGHT) || has_col(buf, CRUX_COL_CUM_IMPORTANCE_WEIGHT) || has_col(buf, CRUX_COL_REV_IMPORTANCE_WEIGHT) || (nominal && has_col(buf, CRUX_COL_IMPORTANCE_WEIGHT))))
    return crux_fail(c, CRUX_EINVAL, "steps!: a block of %lld transitions does not fit the buffer (capacity %lld) whose advantage / return / importance-weight columns it must fill", (long long)n, (long long)buf->capacity);
  if (!cols[CRUX_COL_EPISODE_END]) return crux_fail(c, CRUX_EINVAL, "steps_push: the block needs its :episode_end column (the caller's Sampler cuts the episodes: sampler.jl:130-136,148)");
  // every argument is checked BEFORE the ring moves: a failed call must not leave pushed rows whose advantage / return columns hold stale data
  if (n > 0 && n <= buf->capacity) {
    if (has_col(buf, CRUX_COL_ADVANTAGE) && !critic) return crux_fail(c, CRUX_EINVAL, "steps_push: the buffer has an :advantage column but no critic was given (fill_gae!, sampler.jl:56)");
    if (has_col(buf, CRUX_COL_COST_ADVANTAGE) && !cost_critic) return crux_fail(c, CRUX_EINVAL, "steps_push: the buffer has a :cost_advantage column but no cost critic (Sampler.Vc, sampler.jl:65)"); }
  const int64_t first = buf->next_ind;
  if (first_row_out) *first_row_out = first;
  int32_t rc = crux_buffer_push_host(buf, n, cols, nullptr); if (rc) return rc;
  if (n == 0 || n > buf->capacity) return CRUX_OK;
  if (has_col(buf, CRUX_COL_ADVANTAGE)) {
    rc = crux_fill_gae_rows(buf, critic, lambda, gamma, first, n, rows_per_env, close_last); if (rc) return rc; }
  if (has_col(buf, CRUX_COL_RETURN)) { rc = crux_fill_returns_rows(buf, gamma, first, n, rows_per_env, close_last); if (rc) return rc; }
  if (nominal && has_col(buf, CRUX_COL_IMPORTANCE_WEIGHT)) {      // step! wrote exp(logpdf(pa, s, a) - logprob) per row (sampler.jl:108-111): the same on the pushed rows (a wrapped block takes two ranges)
    const int64_t n1 = n < buf->capacity - first ? n : buf->capacity - first;
    rc = crux_importance_weight_rows(buf, nominal, nominal_head, first, n1); if (rc) return rc;
    if (n1 < n) { rc = crux_importance_weight_rows(buf, nominal, nominal_head, 0, n - n1); if (rc) return rc; } }
  rc = crux_fill_importance_weights_rows(buf, first, n, rows_per_env, close_last); if (rc) return rc;
  if (has_col(buf, CRUX_COL_COST_ADVANTAGE)) {
    rc = crux_fill_gae_rows_keys(buf, cost_critic, lambda, gamma, first, n, rows_per_env, close_last, CRUX_COL_COST, CRUX_COL_COST_ADVANTAGE); if (rc) return rc; }
  if (has_col(buf, CRUX_COL_COST_RETURN)) { rc = crux_fill_returns_rows_keys(buf, gamma, first, n, rows_per_env, close_last, CRUX_COL_COST, CRUX_COL_COST_RETURN); if (rc) return rc; }
  return CRUX_OK;
}

int32_t crux_rollout_multi(int32_t n, crux_env* const* envs, crux_mlp* const* policies, const crux_rollout_cfg* cfg, crux_buffer* const* bufs, int64_t T, double* sum_r, int64_t* n_episode_end) {
  if (n < 1 || !envs || !policies || !cfg || !bufs || T < 1) return CRUX_EINVAL;
  crux_ctx* c = envs[0]->ctx; crux_env* e0 = envs[0]; const NetDesc& pn = policies[0]->nd; const int nout = pn.dims[pn.L];
  const bool h64 = pn.L == 3 && pn.dims[1] == 64 && pn.dims[2] == 64 && pn.acts[0] == pn.acts[1] && pn.acts[2] == CRUX_ACT_IDENTITY;
  if (!h64) return crux_fail(c, CRUX_EUNSUP, "steps! (multi): only the 64-wide register-resident rollout kernel is batched");
  std::vector<RolloutArgs> as((size_t)n);
  for (int i = 0; i < n; ++i) {
    crux_env* e = envs[i]; crux_buffer* b = bufs[i]; crux_mlp* p = policies[i];
    if (!e || !b || !p || e->kind != e0->kind || e->n_envs != e0->n_envs || p->nd.n_params != pn.n_params || b->obs_dim != e->obs_dim || b->act_dim != e->act_dim || (int64_t)e->n_envs * T > b->capacity)
      return crux_fail(c, CRUX_EINVAL, "steps! (multi): problem %d does not match problem 0 or its buffer is too small", i);
    fill_rollout_args(as[(size_t)i], e, p, cfg, b, T);
  }
  const size_t bytes = sizeof(RolloutArgs) * (size_t)n;
  RolloutArgs* d_args = (RolloutArgs*)crux_scratch(c, bytes + 256); if (!d_args) return crux_fail(c, CRUX_ENOMEM, "steps! (multi): scratch");
  HIPCHK(c, hipMemcpyAsync(d_args, as.data(), bytes, hipMemcpyHostToDevice, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream));
  const dim3 grid((unsigned)((size_t)n * (size_t)e0->n_envs));
  crux_prof_begin(c, CRUX_PROF_ROLLOUT);
  bool done = false;
#define ROM_CASE(I, O, A_, K) if (!done && pn.dims[0] == I && nout == O && pn.acts[0] == A_ && e0->kind == K) { hipLaunchKernelGGL((k_rollout_h64<I, O, A_, K>), grid, dim3(64), 0, c->stream, as[0], (const RolloutArgs*)d_args); done = true; }
  ROM_CASE(4, 2, CRUX_ACT_RELU, CRUX_ENV_CARTPOLE)
  ROM_CASE(4, 2, CRUX_ACT_TANH, CRUX_ENV_CARTPOLE)
  ROM_CASE(3, 1, CRUX_ACT_RELU, CRUX_ENV_PENDULUM)
  ROM_CASE(3, 1, CRUX_ACT_TANH, CRUX_ENV_PENDULUM)
  ROM_CASE(17, 6, CRUX_ACT_TANH, CRUX_ENV_SYNTH)
  ROM_CASE(17, 6, CRUX_ACT_RELU, CRUX_ENV_SYNTH)
#undef ROM_CASE
  crux_prof_end(c, CRUX_PROF_ROLLOUT);
  if (!done) return crux_fail(c, CRUX_EUNSUP, "steps! (multi): no batched rollout kernel for this policy / environment");
  int32_t rc = crux_launch_check(c, "k_rollout_h64 (multi)"); if (rc) return rc;
  for (int i = 0; i < n; ++i) { if (bufs[i]->prioritized) return crux_fail(c, CRUX_EUNSUP, "steps! (multi): prioritized buffers are not batched"); crux_buffer_ring_advance(bufs[i], (int64_t)envs[i]->n_envs * T); }
  if (sum_r || n_episode_end) {     // n small copies enqueued on the library's stream (never the null stream), one synchronisation
    const size_t per = 2 * (size_t)e0->n_envs; std::vector<double> acc(per * (size_t)n);
    for (int i = 0; i < n; ++i) HIPCHK(c, hipMemcpyAsync(acc.data() + per * (size_t)i, envs[i]->acc, 8 * per, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < n; ++i) { double sr = 0; int64_t ne = 0; for (int k = 0; k < e0->n_envs; ++k) { sr += acc[per * (size_t)i + 2 * k]; ne += (int64_t)acc[per * (size_t)i + 2 * k + 1]; }
      if (sum_r) sum_r[i] = sr; if (n_episode_end) n_episode_end[i] = ne; }
  }
  return CRUX_OK;
}

int32_t crux_env_step_host(crux_ctx* c, int32_t kind, int64_t n, const double* state, const void* action, const double* uniforms, double* next_state,
                           float* obs, float* r, uint8_t* done) {
  if (!c || n < 0 || !state || !action) return CRUX_EINVAL;
  if (kind != CRUX_ENV_CARTPOLE && kind != CRUX_ENV_PENDULUM && kind != CRUX_ENV_GRIDWORLD) return crux_fail(c, CRUX_EUNSUP, "env kind %d has no device dynamics yet", kind);
  if (n == 0) return CRUX_OK;
  const int sd = kind == CRUX_ENV_CARTPOLE ? 4 : 2, od = kind == CRUX_ENV_CARTPOLE ? 4 : (kind == CRUX_ENV_PENDULUM ? 3 : 2);
  const size_t ab = kind == CRUX_ENV_CARTPOLE ? 2 * (size_t)n : 4 * (size_t)n;     // 2 / 4 one-hot bytes, or one Float32
  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  const size_t o_s = 0, o_a = o_s + al(8 * (size_t)n * sd), o_ns = o_a + al(ab), o_o = o_ns + al(8 * (size_t)n * sd), o_r = o_o + al(4 * (size_t)n * od), o_d = o_r + al(4 * (size_t)n),
               o_u = o_d + al((size_t)n);
  char* sc = (char*)crux_scratch(c, o_u + al(8 * (size_t)n));
  if (!sc) return crux_fail(c, CRUX_ENOMEM, "env_step: scratch");
  HIPCHK(c, hipMemcpyAsync(sc + o_s, state, 8 * (size_t)n * sd, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(sc + o_a, action, ab, hipMemcpyHostToDevice, c->stream));
  if (uniforms) HIPCHK(c, hipMemcpyAsync(sc + o_u, uniforms, 8 * (size_t)n, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_env_step, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, kind, n, (const double*)(sc + o_s), (const void*)(sc + o_a),
                     uniforms ? (const double*)(sc + o_u) : (const double*)nullptr, (double*)(sc + o_ns), (float*)(sc + o_o), (float*)(sc + o_r), (uint8_t*)(sc + o_d));
  int32_t rc = crux_launch_check(c, "k_env_step"); if (rc) return rc;
  if (next_state) HIPCHK(c, hipMemcpyAsync(next_state, sc + o_ns, 8 * (size_t)n * sd, hipMemcpyDeviceToHost, c->stream));
  if (obs) HIPCHK(c, hipMemcpyAsync(obs, sc + o_o, 4 * (size_t)n * od, hipMemcpyDeviceToHost, c->stream));
  if (r) HIPCHK(c, hipMemcpyAsync(r, sc + o_r, 4 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  if (done) HIPCHK(c, hipMemcpyAsync(done, sc + o_d, (size_t)n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return CRUX_OK;
}

}  // extern "C"

// solve(::OffPolicySolver) for a small DQN (see k_dqn_small_solve). Preconditions (CRUX_EUNSUP otherwise, the caller then loops piece by piece): uniform replay,
// batch capacity == B <= 256, at most 4 environments, dN a multiple of them, both networks narrower than the dense engine's threshold, a rollout that the
// generic kernel would run. infos: host [iters x epochs x CRUX_INFO_N] (loss, grad norm, [2] = Qavg per epoch).
extern "C" int32_t crux_dqn_small_solve(crux_mlp* net, crux_mlp* target_net, crux_env* e, const crux_rollout_cfg* cfg, crux_buffer* source, crux_buffer* batch,
                                        int32_t iters, int32_t dN, int32_t epochs, float gamma, float tau, int32_t use_weight, uint64_t i0, float* infos, double* sum_r, int64_t* n_episode_end) {
  if (!net || !target_net || !e || !cfg || !source || !batch || iters < 1 || dN < 1 || epochs < 1) return CRUX_EINVAL;
  crux_ctx* c = net->ctx; const int64_t B = batch->capacity; const int E = e->n_envs;
  const NetDesc& pn = net->nd;
  const bool h64 = pn.L == 3 && pn.dims[1] == 64 && pn.dims[2] == 64 && pn.acts[0] == pn.acts[1] && pn.acts[2] == CRUX_ACT_IDENTITY;
  if (source->prioritized || batch->prioritized || B > 256 || B < 1 || E > 4 || dN % E || pn.maxdim >= CRUX_DENSE_MIN_WIDTH || target_net->nd.maxdim >= CRUX_DENSE_MIN_WIDTH || h64 ||
      target_net->nd.n_params != pn.n_params || !net->has_adam || cfg->head != CRUX_HEAD_GREEDY_Q || source->act_kind != CRUX_ACTION_DISCRETE || pn.dims[pn.L] != source->act_dim || dN > source->capacity || use_weight)
    return crux_fail(c, CRUX_EUNSUP, "dqn_small_solve: configuration outside the one-launch solve kernel");
  if (source->obs_dim != e->obs_dim || source->act_dim != e->act_dim || batch->obs_dim != source->obs_dim || batch->act_dim != source->act_dim || pn.dims[0] != e->obs_dim) return crux_fail(c, CRUX_EINVAL, "dqn_small_solve: shapes of env / buffers / network differ");
  SmallSolveArgs q{};
  fill_rollout_args(q.ro, e, net, cfg, source, dN / E);
  // train!(pi, td_loss) on rows 0..B-1 of the staging buffer (the ids live in the staging buffer's order scratch)
  TrainArgs& a = q.tr; memset(&a, 0, sizeof a);
  a.nd = net->nd; a.p = net->p; a.g = net->g; a.m = net->m; a.v = net->v; a.bp = net->bp; a.eta = net->eta; a.b1 = net->b1; a.b2 = net->b2; a.eps = net->eps;
  a.S = (const float*)batch->col[CRUX_COL_S]; a.A = batch->col[CRUX_COL_A]; a.od = batch->obs_dim; a.ad = batch->act_dim; a.act_kind = batch->act_kind;
  a.loss = CRUX_LOSS_TD_INTERNAL; a.head = CRUX_HEAD_GREEDY_Q; a.bs = (int32_t)B; a.epochs = 1; a.max_batches = 0; a.target_kl = -1.f; a.len = B; a.apply = 1;
  a.order_a = batch->order_a; a.order_b = batch->order_b;
  const size_t nout = (size_t)pn.dims[pn.L];
  const size_t small = 4 * ((size_t)B * nout + (size_t)B) + 4 * (size_t)B + 2048;
  char* sc = (char*)crux_scratch(c, small + sizeof(float) * CRUX_INFO_N * (size_t)iters * (size_t)epochs + 4096); if (!sc) return crux_fail(c, CRUX_ENOMEM, "dqn_small_solve: scratch");
  int32_t* d_ids = (int32_t*)sc; q.qtmp = (float*)(sc + ((4 * (size_t)B + 255) / 256) * 256); q.y = q.qtmp + B * nout; q.status = (int32_t*)(q.y + B); q.infos = (float*)(sc + ((small + 255) / 256) * 256);
  { std::vector<int32_t> h((size_t)B); for (int64_t j = 0; j < B; ++j) h[(size_t)j] = (int32_t)j;
    HIPCHK(c, hipMemcpyAsync(d_ids, h.data(), 4 * (size_t)B, hipMemcpyHostToDevice, c->stream)); HIPCHK(c, hipMemsetAsync(q.status, 0, 256, c->stream));
    HIPCHK(c, hipMemsetAsync(q.infos, 0, sizeof(float) * CRUX_INFO_N * (size_t)iters * (size_t)epochs, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream)); }
  a.ids = d_ids; a.n_ids = B; a.Y = q.y; a.Wt = nullptr; a.status = q.status;
  q.ndt = target_net->nd; q.pt = target_net->p;
  q.gn = 0; for (int k = 0; k < CRUX_NCOLS; ++k) { if (!has_col(batch, k) || !has_col(source, k)) continue; const size_t st = col_stride(batch, k); const int z = q.gn++;
    q.gdst[z] = batch->col[k]; q.gsrc[z] = source->col[k]; q.gesz[z] = st % 4 == 0 ? 4 : 1; q.gre[z] = (int32_t)(st % 4 == 0 ? st / 4 : st);
  }
  q.bSP = (const float*)batch->col[CRUX_COL_SP]; q.bR = (const float*)batch->col[CRUX_COL_R]; q.bD = (const uint8_t*)batch->col[CRUX_COL_DONE]; q.bidx = batch->d_indices;
  q.elements = source->elements; q.next = source->next_ind; q.C = source->capacity; q.B = (int32_t)B; q.epochs = epochs; q.iters = iters; q.dN = dN; q.E = E;
  q.gamma = gamma; q.tau = tau; q.sample_seed = source->sample_seed; q.sample_stream = source->sample_stream; q.i0 = i0;
  size_t fl = 0; for (int l = 0; l <= pn.L; ++l) fl += (size_t)pn.dims[l] * 32; fl += 2 * (size_t)pn.maxdim * 32 + 32 * (size_t)(pn.n_extra > 0 ? pn.n_extra : 1);     // = generic_lds_bytes (train.hip)
  q.lds_train = (int32_t)((fl + 63) / 64 * 64); q.lds_fwd = (int32_t)(2 * (size_t)target_net->nd.maxdim * FWD_TS);
  size_t lds = sizeof(float) * ((size_t)q.lds_train + (size_t)q.lds_fwd + (size_t)E * (2 * 1024 + ENV_MAXOBS + 8));
  if (lds > 150 * 1024) return crux_fail(c, CRUX_EUNSUP, "dqn_small_solve: %zu bytes of LDS", lds);
  // the README shape (2 -> 8 -> 4, relu, one environment, B <= 128) with a ring that fits LDS: the wave-resident kernel
  const size_t tiny_lds = sizeof(float) * ((size_t)(2 * 1024 + ENV_MAXOBS + 8) + (size_t)source->capacity * (2 * 2 + 3) + 60 * 64) + 64;
  const bool tiny = pn.L == 2 && pn.dims[0] == 2 && pn.dims[1] == 8 && pn.dims[2] == 4 && pn.acts[0] == CRUX_ACT_RELU && pn.acts[1] == CRUX_ACT_IDENTITY && pn.n_extra == 0 &&
                    target_net->nd.L == 2 && target_net->nd.dims[1] == 8 && target_net->nd.acts[0] == CRUX_ACT_RELU && E == 1 && B <= 128 && dN <= 64 && tiny_lds <= 60 * 1024 &&
                    !crux_sw().small_solve_generic;
  crux_prof_begin(c, tiny ? CRUX_PROF_TINY_SOLVE : CRUX_PROF_TD_STEP);
  if (tiny) hipLaunchKernelGGL((k_dqn_tiny_solve<2, 8, 4>), dim3(1), dim3(64), tiny_lds, c->stream, q);
  else {
    static size_t attr_set = 0;
    if (lds > attr_set) { HIPCHK(c, hipFuncSetAttribute((const void*)k_dqn_small_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr_set = lds; }
    hipLaunchKernelGGL(k_dqn_small_solve, dim3(1), dim3(256), lds, c->stream, q);
  }
  crux_prof_end(c, tiny ? CRUX_PROF_TINY_SOLVE : CRUX_PROF_TD_STEP);
  int32_t rc = crux_launch_check(c, "k_dqn_small_solve"); if (rc) return rc;
  int32_t hst9[9] = {0}; int32_t hst[2];
  HIPCHK(c, hipMemcpyAsync(hst9, q.status, sizeof hst9, hipMemcpyDeviceToHost, c->stream));
  if (infos) HIPCHK(c, hipMemcpyAsync(infos, q.infos, sizeof(float) * CRUX_INFO_N * (size_t)iters * (size_t)epochs, hipMemcpyDeviceToHost, c->stream));
  std::vector<double> acc(2 * (size_t)E);
  HIPCHK(c, hipMemcpyAsync(acc.data(), e->acc, 16 * (size_t)E, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  // host-side bookkeeping of the rings (push!, experience_buffer.jl:256-257) and of the staging buffer's sample state
  for (int it = 0; it < iters; ++it) crux_buffer_ring_advance(source, dN);
  batch->elements = B; batch->next_ind = 0; batch->total_count += (int64_t)B * epochs * iters; batch->indices_n = B; batch->indices_stale = true;
  if (sum_r || n_episode_end) { double sr = 0; int64_t ne = 0; for (int k = 0; k < E; ++k) { sr += acc[2 * k]; ne += (int64_t)acc[2 * k + 1]; } if (sum_r) *sum_r = sr; if (n_episode_end) *n_episode_end = ne; }
  hst[0] = hst9[0]; hst[1] = hst9[8];
  if (crux_sw().small_solve_timing) { unsigned long long tt[8]; (void)hipMemcpy(tt, q.status + 16, sizeof tt, hipMemcpyDeviceToHost); unsigned long long tot = 0; for (auto v : tt) tot += v;
    fprintf(stderr, "[small-solve] rollout %.1f%% ids (tiny kernel: sample + forward / backward) %.1f%% gather (tiny: transpose-reduce) %.1f%% target %.1f%% train (tiny: statistics + Adam) %.1f%%\n", 100.0 * tt[0] / tot, 100.0 * tt[1] / tot, 100.0 * tt[2] / tot, 100.0 * tt[3] / tot, 100.0 * tt[4] / tot); fprintf(stderr, "[small-solve] tiny: rollout body %.1f%%, fence + mirror %.1f%%\n", 100.0 * tt[5] / tot, 100.0 * tt[0] / tot); }
  if (hst[1] == CRUX_ENAN || hst[0] == CRUX_ENAN) return crux_fail(c, CRUX_ENAN, "NaN detected! (grad norm is NaN, src/training.jl:20)");
  if (hst[1]) return crux_fail(c, hst[1], "small solve kernel reported status %d", hst[1]);
  return CRUX_OK;
}
