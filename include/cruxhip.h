/* cruxhip.h -- C ABI of libcruxhip.so: the MI355X (gfx950) actor-learner hot path behind Crux.jl's
 * Sampler / ExperienceBuffer / batch_train! seams.
 *
 * Every entry point names the reference interface it replaces (file:line under sisl/Crux.jl v0.1.4).
 * The reference has no FFI; its seams are Julia multiple dispatch on the container type
 * (src/devices.jl:1-21, src/experience_buffer.jl:53,87-95) and function-valued solver fields
 * (src/training.jl:1-11, src/model_free/on_policy.jl:44-45). A Julia shim binds these symbols with
 * `ccall`; the Python mirror in crux.jl_amd/ binds them with ctypes.
 *
 * Conventions
 *  - every function returns int32 status: 0 ok, <0 error (message via crux_last_error).
 *  - arrays are laid out exactly like the Julia arrays: column-major (features..., batch), batch last;
 *    Dense weight is out x in column-major; Bool columns are 1 byte.
 *  - indices cross the ABI 0-based (Julia callers subtract 1).
 *  - host pointers are borrowed for the duration of the call; `d_` prefixed pointers are device memory.
 *  - one context per host thread; calls on a context are stream-ordered and asynchronous unless they
 *    return host data.
 */
#ifndef CRUXHIP_H
#define CRUXHIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes ------------------------------------------------------------------------------ */
#define CRUX_OK       0
#define CRUX_EINVAL  -1  /* shape/dtype mismatch  == @assert at src/experience_buffer.jl:251        */
#define CRUX_ENAN    -2  /* NaN grad norm / advantage == src/training.jl:20, src/sampler.jl:270     */
#define CRUX_EHIP    -3  /* HIP runtime failure                                                    */
#define CRUX_ERCCL   -4  /* collective failure                                                     */
#define CRUX_ENOMEM  -5
#define CRUX_EUNSUP  -6  /* configuration not supported by any kernel                              */

typedef struct crux_ctx crux_ctx;
typedef struct crux_mlp crux_mlp;
typedef struct crux_buffer crux_buffer;
typedef struct crux_env crux_env;

/* lifecycle --------------------------------------------------------------------------------- */
/* stream: a hipStream_t to run on (e.g. torch's current stream) or NULL to create a private one. */
int32_t crux_ctx_create(int32_t device_id, void* stream, crux_ctx** out);
/* The CRUX_* environment switches (development / test knobs) are read ONCE per process, when its first context is created, into one snapshot (creating further
 * contexts does not re-read them under the ones already running); a process that changes one afterwards calls this to have it take effect. */
int32_t crux_reload_switches(void);
int32_t crux_ctx_destroy(crux_ctx* ctx);
/* How many CUs one small-MLP learner (batch_train!, src/training.jl:28-55) occupies: 0 = automatic (two CUs of one XCD per learner; the batched
 * multi-learner call switches to one CU per learner above 64 learners), 1 = always one CU (k_train_mfma8), 2 = two CUs where the shape allows
 * (populations above 64 always take the one-CU form: two launches x 2 n workgroups would not be co-resident).
 * The two kernels sum the minibatch gradient in different orders: results agree to fp32 tolerance, bitwise only for equal settings.              */
int32_t crux_ctx_set_learner_cus(crux_ctx* ctx, int32_t cus);   /* 0 (default): automatic = the feature-split kernel on four CUs for full batch_train! loops, else the two- / one-CU kernels */
const char* crux_last_error(crux_ctx* ctx);
int32_t crux_sync(crux_ctx* ctx);
const char* crux_version(void);

/* raw device memory for callers without their own allocator (targets, td errors, ...). */
int32_t crux_device_alloc(crux_ctx* ctx, int64_t bytes, void** d_out);
int32_t crux_device_free(crux_ctx* ctx, void* d_ptr);
int32_t crux_memcpy_h2d(crux_ctx* ctx, void* d_dst, const void* src, int64_t bytes);
int32_t crux_memcpy_d2h(crux_ctx* ctx, void* dst, const void* d_src, int64_t bytes);

/* kernel timing on the context's stream (HIP events). slot = one of CRUX_PROF_*.                */
enum { CRUX_PROF_ROLLOUT = 0, CRUX_PROF_VALUES = 1, CRUX_PROF_GAE = 2, CRUX_PROF_WHITEN = 3,
       CRUX_PROF_TRAIN_ACTOR = 4, CRUX_PROF_TRAIN_CRITIC = 5, CRUX_PROF_PER_SCAN = 6,
       CRUX_PROF_PER_SEARCH = 7, CRUX_PROF_GATHER = 8, CRUX_PROF_TD_STEP = 9,
       CRUX_PROF_TINY_SOLVE = 10 /* launches of the wave-resident k_dqn_tiny_solve (the README shape) */, CRUX_PROF_NSLOTS = 16 };
int32_t crux_prof_enable(crux_ctx* ctx, int32_t on);
int32_t crux_prof_reset(crux_ctx* ctx);
/* returns accumulated milliseconds and launch count for a slot (synchronises the stream). */
int32_t crux_prof_get(crux_ctx* ctx, int32_t slot, double* ms_total, int64_t* launches);

/* networks: Chain(Dense...) --------------------------------------------------------------------
 * replaces Flux Chain(Dense(in,out,act)...) wrapped by ContinuousNetwork / DiscreteNetwork
 * (src/policies.jl:68-98,104-157). Flat parameter vector = Flux.params order: W1,b1,W2,b2,... then
 * `n_extra` trailing trainables (GaussianPolicy's ConstantLayer logSigma, src/policies.jl:315-320,
 * src/utils.jl:31-36). n_layers = 0 with n_extra >= 1 is a bare trainable vector (a ConstantLayer on
 * its own; SAC's log_alpha, src/model_free/rl/sac.jl:96) that owns parameters and Adam state only. */
enum { CRUX_ACT_IDENTITY = 0, CRUX_ACT_RELU = 1, CRUX_ACT_TANH = 2 };
int32_t crux_mlp_create(crux_ctx* ctx, int32_t n_layers, const int32_t* dims /*n_layers+1*/,
                        const int32_t* acts /*n_layers*/, int32_t n_extra, crux_mlp** out);
int32_t crux_mlp_destroy(crux_mlp* net);
int64_t crux_mlp_n_params(const crux_mlp* net);
int32_t crux_mlp_set_params(crux_mlp* net, const float* host_flat, int64_t n);
int32_t crux_mlp_get_params(crux_mlp* net, float* host_flat, int64_t n);
float*  crux_mlp_params_ptr(crux_mlp* net);        /* device pointer to the flat parameters      */
float*  crux_mlp_grads_ptr(crux_mlp* net);         /* device pointer to the flat gradient scratch */
/* Flux glorot_uniform: U(+-sqrt(6/(in+out))) weights, zero bias; extras set to `extra_init`.    */
int32_t crux_mlp_init_glorot(crux_mlp* net, uint64_t seed, uint32_t stream, float extra_init);
/* y[out x B] = net(x[in x B]); value(pi, s) (src/policies.jl:94,120). Device pointers.          */
int32_t crux_mlp_forward(crux_mlp* net, const float* d_x, int64_t B, float* d_y);
/* same with host pointers (copies in/out, synchronous).                                          */
int32_t crux_mlp_forward_host(crux_mlp* net, const float* x, int64_t B, float* y);
/* The pullback Zygote builds for value(pi, x) (src/training.jl:16-18) as an explicit call, on the MFMA dense engine:
 * forward with cached activations, then for d(loss)/d(y) = d_dy [out x B]: parameter gradients (scaled by grad_scale,
 * written to crux_mlp_grads_ptr; skipped when want_param_grads = 0) and d(loss)/d(x) into d_dx [in x B] (NULL = skip).   */
int32_t crux_mlp_forward_cached(crux_mlp* net, const float* d_x, int64_t B, float* d_y /* NULL = keep internal only */);
int32_t crux_mlp_backward(crux_mlp* net, const float* d_x, int64_t B, const float* d_dy, float grad_scale,
                          int32_t want_param_grads, float* d_dx);
/* SquashedGaussianPolicy(mu, logSigma::Array, ascale) (src/policies.jl:353-400): ascale > 0 switches the Gaussian head of this handle (rollout and
 * the policy-gradient / BC learners) to a = ascale*tanh(mu + sigma*eps), sigma = exp(clamp(logSigma, -5, 2)), logpdf with the tanh correction and
 * atanh(clamp(a/ascale, -1+1f-5, 1-1f-5)) for stored actions; entropy stays 1.4189385 + sum(logSigma) (:398). ascale = 0 restores GaussianPolicy. */
int32_t crux_mlp_set_squash(crux_mlp* net, float ascale);
float crux_mlp_get_squash(const crux_mlp* net);
/* copyto!(to, from) (src/policies.jl:61-65) and polyak_average!(to, from, tau) (:48-59).         */
int32_t crux_mlp_copy(crux_mlp* to, const crux_mlp* from);
int32_t crux_polyak(crux_mlp* to, const crux_mlp* from, float tau);

/* optimiser: Flux.Optimise.Adam(eta, (b1,b2), eps) attached to one network
 * (TrainingParams.optimizer, src/training.jl:3; Flux.update! at :21). State = (m, v, beta powers). */
int32_t crux_adam_init(crux_mlp* net, double eta, double beta1, double beta2, double eps);
int32_t crux_adam_get_state(crux_mlp* net, float* m_host, float* v_host, double* beta_pow /*2*/);
int32_t crux_adam_set_state(crux_mlp* net, const float* m_host, const float* v_host, const double* beta_pow);
/* device pointers to the moment vectors (replica averaging in the multi-GPU path). */
int32_t crux_adam_state_ptrs(crux_mlp* net, float** d_m, float** d_v);

/* experience buffer ------------------------------------------------------------------------------
 * replaces mdp_data / ExperienceBuffer (src/experience_buffer.jl:4-35,53-80). Columns are separate
 * device arrays (SoA across keys; one transition's features contiguous within a key).           */
enum { CRUX_COL_S = 0, CRUX_COL_A = 1, CRUX_COL_SP = 2, CRUX_COL_R = 3, CRUX_COL_DONE = 4,
       CRUX_COL_EPISODE_END = 5, CRUX_COL_RETURN = 6, CRUX_COL_LOGPROB = 7, CRUX_COL_ADVANTAGE = 8,
       CRUX_COL_WEIGHT = 9, CRUX_COL_T = 10, CRUX_COL_I = 11, CRUX_COL_VALUE = 12,
       /* cost-constrained solvers (LagrangePPO, rl/ppo.jl:211): info["cost"] of the step (sampler.jl:114), its GAE under the cost critic Vc and its return (:65-66) */
       CRUX_COL_COST = 13, CRUX_COL_COST_ADVANTAGE = 14, CRUX_COL_COST_RETURN = 15,
       /* importance-weight columns (experience_buffer.jl:17-19: Float32, initialised to ONE like :weight): the per-step ratio exp(logpdf(pa, s, a) - logprob) of the nominal
        * action policy `pa` (sampler.jl:108-111) and its running products over the episode, filled by terminate_episode! (sampler.jl:58-62,283-308) */
       CRUX_COL_IMPORTANCE_WEIGHT = 16, CRUX_COL_FWD_IMPORTANCE_WEIGHT = 17, CRUX_COL_REV_IMPORTANCE_WEIGHT = 18, CRUX_COL_CUM_IMPORTANCE_WEIGHT = 19, CRUX_COL_TRAJ_IMPORTANCE_WEIGHT = 20,
       CRUX_NCOLS = 21 };
#define CRUX_COL_INIT_ONE(k) ((k) == CRUX_COL_WEIGHT || ((k) >= CRUX_COL_IMPORTANCE_WEIGHT && (k) <= CRUX_COL_TRAJ_IMPORTANCE_WEIGHT))
enum { CRUX_ACTION_DISCRETE = 0 /* Bool one-hot, 1 byte each (src/spaces.jl:18,24) */,
       CRUX_ACTION_CONTINUOUS = 1 /* Float32 */ };
/* column_mask: bit k set => optional column k present (S,A,SP,R,DONE,EPISODE_END always are).   */
int32_t crux_buffer_create(crux_ctx* ctx, int32_t obs_dim, int32_t act_dim, int32_t act_kind,
                           int64_t capacity, uint32_t column_mask, int32_t prioritized, float alpha,
                           crux_buffer** out);
int32_t crux_buffer_destroy(crux_buffer* b);
int64_t crux_buffer_len(const crux_buffer* b);         /* Base.length   (:182)                    */
int64_t crux_buffer_capacity(const crux_buffer* b);    /* capacity      (:184)                    */
int64_t crux_buffer_next_ind(const crux_buffer* b);    /* 0-based next_ind                        */
int64_t crux_buffer_total_count(const crux_buffer* b);
int32_t crux_buffer_has_column(const crux_buffer* b, int32_t key);
int32_t crux_buffer_clear(crux_buffer* b);             /* clear!        (:97-104)                 */
/* element size in bytes and rows-per-transition of a column (e.g. S: 4, obs_dim).               */
int32_t crux_buffer_column_info(const crux_buffer* b, int32_t key, int32_t* elem_bytes, int32_t* rows);
int32_t crux_buffer_column_ptr(crux_buffer* b, int32_t key, void** d_ptr);
/* push!(b, data) (:232-259): ring write of n transitions from host columns. cols[k]==NULL means
 * "data has no key k" (skipped, :238-241). Writes the destination indices I (0-based) if I_out.   */
int32_t crux_buffer_push_host(crux_buffer* b, int64_t n, const void* const* cols /*CRUX_NCOLS*/,
                              int64_t* I_out);
/* push_reservoir!(buffer, data; weighted) (src/experience_buffer.jl:262-288): reservoir sampling into a full buffer. Row i of this call draws
 * x = Philox(seed, counter + i, 0, CRUX_RNG_RESERVOIR) (crux_rng.h): rand() of the weight test from x[0..1], rand(1:total_count) from x[2..3].
 * Kept quirks: total_count grows by 2 per element while the buffer fills (:272 + push! :235); replaced slots keep their priorities.            */
int32_t crux_buffer_push_reservoir(crux_buffer* b, int64_t N, const void* const* cols /* [CRUX_NCOLS] host, NULL = absent */, int32_t weighted, uint64_t seed, uint64_t counter);
/* push!(target, source, ids=ids) (:232-259): device gather of rows `ids` (host array, 0-based; NULL
 * => 0..n-1) of `src` into the ring of `dst`.                                                     */
int32_t crux_buffer_push_buffer(crux_buffer* dst, const crux_buffer* src, const int64_t* ids, int64_t n,
                                int64_t* I_out);
/* copy the first `n` (<= len) transitions of a column to the host: b[key] (:176).                */
int32_t crux_buffer_read_column(crux_buffer* b, int32_t key, void* host_out, int64_t n);
/* overwrite the first n transitions of a column from the host (b[key] .= x).                      */
int32_t crux_buffer_write_column(crux_buffer* b, int32_t key, const void* host_in, int64_t n);
/* shuffle!(b) with an explicit permutation (:118-124): new[:,j] = old[:,perm[j]], every column.    */
int32_t crux_buffer_permute(crux_buffer* b, const int64_t* perm /*len*/);
/* shuffle!(b) with the library's permutation stream (crux_rng.h: crux_perm_make(seed, counter, 0, length(b))), composed and applied on the device. */
int32_t crux_buffer_shuffle(crux_buffer* b, uint64_t seed, uint64_t counter);
/* get_last_N_indices (:223-229), 0-based; returns count written.                                  */
int64_t crux_buffer_last_n_indices(const crux_buffer* b, int64_t N, int64_t* out);
/* minibatch_copy(b, indices) (:171): gather rows to host; outs[k]==NULL skips a column.            */
int32_t crux_buffer_gather_host(crux_buffer* b, const int64_t* ids, int64_t n, void* const* outs);
/* device copy of the same indices (int64, 0-based), e.g. to feed crux_per_update_device without a host round trip. */
int64_t* crux_buffer_indices_ptr(crux_buffer* b);
/* indices of the last sample!() into this (staging) buffer: target.indices (:319,338).            */
int32_t crux_buffer_indices(const crux_buffer* b, int64_t* out, int64_t n);

/* prioritized replay (src/experience_buffer.jl:38-50,290-301,324-349) ---------------------------- */
/* update_priorities!(b, I, v): v Float64 (v_is_f64=1) or Float32.                                 */
int32_t crux_per_update(crux_buffer* b, const int64_t* I, const void* v, int32_t v_is_f64, int64_t n);
/* same with device-resident ids (int64) and Float32 values (the td-error path, off_policy.jl:83). Duplicate ids: the reference's scalar loop lets the LAST
 * occurrence win (:296-298); that order is reproduced exactly for n <= 512. For 512 < n the winner among duplicates is unspecified -- harmless when duplicated
 * ids carry equal values, which holds for td_error of the same row (the only caller inside the library); pass unique ids or equal values otherwise. */
int32_t crux_per_update_device(crux_buffer* b, const int64_t* d_ids, const float* d_v, int64_t n);
/* prioritized_sample!(target, source; i, B): `rands` = B Float64 uniforms (NULL => Philox draw with
 * counter `i`). Writes ids into target.indices, IS weights into source[:weight], gathers rows.     */
int32_t crux_per_sample(crux_buffer* target, crux_buffer* source, int64_t B, const double* rands,
                        float beta, uint64_t i);
/* uniform_sample!(target, source; B) (:317-321): `ids` host array or NULL => Philox draw.          */
int32_t crux_uniform_sample(crux_buffer* target, crux_buffer* source, int64_t B, const int64_t* ids,
                            uint64_t i);
/* The Philox draws of both samplers are crux_philox(seed, i*B + j, stream, CRUX_RNG_SAMPLE) (crux_rng.h) with the SOURCE buffer's key and stream
 * (defaults 0x5EED5A3F, 0): rand!(target, sources...) (:303-315) samples every source with an independent draw, so sources get distinct streams. */
int32_t crux_buffer_set_sample_stream(crux_buffer* source, uint64_t seed, uint32_t stream);
int32_t crux_per_get(crux_buffer* b, float* priorities /*capacity or NULL*/, float* max_priority,
                     float* min_priority, float* cumsum /*len or NULL*/);

/* environments + rollout (src/sampler.jl:1-173) --------------------------------------------------- */
enum { CRUX_ENV_CARTPOLE = 0, CRUX_ENV_PENDULUM = 1, CRUX_ENV_GRIDWORLD = 2,
       CRUX_ENV_SYNTH = 3 /* synthetic dynamics, synth_obs_dim observations, synth_act_dim continuous actions (C5-shaped: 17 / 6) */,
       CRUX_ENV_SYNTH_DISCRETE = 4 /* the same with synth_act_dim discrete actions (C3-shaped: 8 / 4) */ };
/* SYNTH (for the configurations whose simulators -- LunarLander, HalfCheetah -- cannot be restated): state x in R^so (Float64),
 * observation Float32(x); u_i = clamp(a[i mod sa], -1, 1) or, for discrete action k, ((i + k) mod sa == 0 ? 1 : -0.25);
 * x'_i = 0.9 x_i + 0.1 sin(x_{(i+1) mod so} + u_i); r = -mean(x'^2) + 0.05 x'_0; done = x'_0 > 0.9; reset x_i ~ U(-0.05, 0.05).
 * Cost channel (the info["cost"] a safety-gym style mdp returns from @gen, sampler.jl:65-66,114), written to the :cost column when the buffer has one:
 * SYNTH 25 x'_{1 mod so}^2 (Float32 of the Float64 product); CartPole 1 when the pole angle |theta'| > 0.05 rad else 0; Pendulum 1 when |thetadot'| > 4 else 0; GridWorld 0.  */
enum { CRUX_HEAD_CATEGORICAL = 0 /* DiscreteNetwork softmax head (policies.jl:104-157)            */,
       CRUX_HEAD_GAUSSIAN = 1    /* GaussianPolicy, const logSigma extras (policies.jl:315-350)    */,
       CRUX_HEAD_GREEDY_Q = 2    /* action(::DiscreteNetwork) argmax (policies.jl:124)             */,
       CRUX_HEAD_DETERMINISTIC = 3 /* ContinuousNetwork action (policies.jl:92)                    */ };
/* n_envs Samplers of one mdp kind (Sampler struct, src/sampler.jl:1-22). obs_mu/obs_sigma: the
 * ContinuousSpace whitening of tovec (src/spaces.jl:25); NULL => 0 / 1. synth_* used by SYNTH.    */
int32_t crux_env_create(crux_ctx* ctx, int32_t kind, int32_t n_envs, int32_t max_steps, float gamma,
                        const float* obs_mu, const float* obs_sigma, uint64_t seed,
                        int32_t synth_obs_dim, int32_t synth_act_dim, crux_env** out);
int32_t crux_env_destroy(crux_env* env);
int32_t crux_env_obs_dim(const crux_env* env);
int32_t crux_env_act_dim(const crux_env* env);
int32_t crux_env_reset(crux_env* env);                /* reset_sampler! for every env (:31-43)   */
/* read back the per-env sampler state: f64 state [state_dim x n_envs], episode_length, resets.   */
int32_t crux_env_get_state(crux_env* env, double* state, int64_t* episode_length, int64_t* n_resets);
int32_t crux_env_state_dim(const crux_env* env);

typedef struct {
  int32_t explore;        /* steps!(...; explore=) (:139); 2 = explore=false with an always_stochastic policy:
                             action(pi, s) = exploration(pi, s)[1] (policies.jl:124), logprob stored as NaN       */
  int32_t reset_at_end;   /* steps!(...; reset=)   (:148)                                          */
  int32_t head;           /* CRUX_HEAD_*                                                           */
  /* MixedPolicy / eps-greedy (policies.jl:466-494): eps(i)=max(stop, start - i*(start-stop)/steps)
     (utils.jl:116-126); eps_steps==0 disables. */
  double eps_start, eps_stop; int64_t eps_steps;
  /* GaussianNoiseExplorationPolicy (policies.jl:499-514); noise_sigma<0 disables. */
  float noise_sigma, noise_eps_min, noise_eps_max, a_min, a_max;
  float logit_div;        /* categorical head: probabilities = softmax(value ./ logit_div) (SoftQ's logit_conversion,
                             rl/softq.jl:53); 0 = plain softmax. Occupies the former padding: the struct stays 72 bytes. */
  uint64_t i0;            /* steps!(...; i=) global interaction counter at the first step          */
} crux_rollout_cfg;

/* steps!(samplers, buffer; Nsteps=T, explore, reset, i) (:139-173) for all envs at once, env-major:
 * rows [e*T, (e+1)*T) of the pushed block belong to env e (== hcat of E single-Sampler rollouts).
 * The T*n_envs transitions are ring-pushed into `buf` (push!, experience_buffer.jl:232-259). The
 * policy forward (exploration / action, sampler.jl:73), the env transition (:89-97), the column
 * writes (:100-107) and episode bookkeeping (:130-136, terminate_episode! :53-69 minus the GAE
 * fill) run in one kernel. sum_r/n_episode_end feed record_avgr (ppo.jl:52-54).                 */
int32_t crux_rollout(crux_env* env, crux_mlp* policy, const crux_rollout_cfg* cfg, crux_buffer* buf,
                     int64_t T, double* sum_r, int64_t* n_episode_end);
/* steps! for n independent samplers of equal shape (own policy, environments, buffer, seed) in one launch -- the rollout half of a
 * multi-seed run (see crux_policy_gradient_training_multi). sum_r / n_episode_end: host [n] or NULL.                               */
int32_t crux_rollout_multi(int32_t n, crux_env* const* envs, crux_mlp* const* policies, const crux_rollout_cfg* cfg,
                           crux_buffer* const* bufs, int64_t T, double* sum_r, int64_t* n_episode_end);

/* Caller-stepped environments: step! with an ARBITRARY mdp (src/sampler.jl:71-137). The reference's step! calls @gen(:sp,:r)(mdp, s, a) on whatever POMDPs.jl mdp the user
 * handed to solve (:89-97); an mdp that only exists on the host keeps its Sampler (state s, svec, episode_length, was_reset -- sampler.jl:1-22) in the host language and
 * uses these two entry points for the device part of steps!:
 *
 * crux_policy_explore = the first line of step! (:73) for n_envs samplers at once:
 *     a, logprob = explore ? exploration(agent.pi_explore, svec; pi_on = agent.pi, i = i) : (action(agent.pi, svec), NaN)
 *   policy / cfg: as crux_rollout (cfg->head, explore, eps_*, noise_*, logit_div; cfg->i0 = the interaction counter `i` of sampler 0 in this step, sampler e uses i0 + e
 *   -- steps!(::Vector{Sampler}) numbers them env-minor, sampler.jl:161-163; reset_at_end is ignored). obs: host [obs_dim x n_envs], the samplers' svec (already tovec'd,
 *   src/spaces.jl:25). Draws: crux_philox(seed, f(steps_taken[e]), stream = e, purpose) exactly as the rollout kernel draws them for its sampler e after steps_taken[e] steps
 *   (crux_rng.h; steps_taken: host [n_envs], NULL = zeros), so a caller that steps one of the restated environments itself reproduces crux_rollout's buffer bit for bit.
 *   actions_out: host [act_dim x n_envs], Bool one-hot bytes for the discrete heads (CATEGORICAL / GREEDY_Q), Float32 otherwise; logprob_out: host [n_envs] or NULL
 *   (NaN where the reference stores NaN). Synchronous (it returns host data).
 *
 * crux_steps_push = the tail of steps! (:148-155) for the block the caller stepped: push!(buffer, data) (experience_buffer.jl:232-259) of n = n_envs x T transitions,
 *   env-major (environment e in rows [e*rows_per_env, (e+1)*rows_per_env) of the block), then -- on the ring rows just written -- what terminate_episode! did to `data`
 *   (:53-66): fill_gae!(critic, lambda, gamma) / fill_returns!(gamma), :importance_weight = exp(logpdf(nominal, s, a) - logprob) (:108-111; nominal NULL = the caller
 *   supplied the column) and its fwd / cum / rev products, fill_gae! / fill_returns! on :cost with cost_critic -- each only if the buffer has the column.
 *   cols: host columns as crux_buffer_push_host, :episode_end included (the caller's Sampler cuts episodes: done, max_steps, and the reset at the end of the block);
 *   close_last = steps!(...; reset=true). first_row_out (optional): 0-based ring row of the block's first transition.                                                */
int32_t crux_policy_explore(crux_mlp* policy, const crux_rollout_cfg* cfg, int32_t n_envs, const float* obs, uint64_t seed, const int64_t* steps_taken,
                            void* actions_out, float* logprob_out);
int32_t crux_steps_push(crux_buffer* buf, int64_t n, const void* const* cols /*CRUX_NCOLS*/, int64_t rows_per_env, int32_t close_last, crux_mlp* critic, float lambda, float gamma,
                        crux_mlp* cost_critic, crux_mlp* nominal, int32_t nominal_head, int64_t* first_row_out);

/* test hook: apply the env dynamics once to explicit states/actions (host arrays):
 * state [state_dim x n] f64, action [act_dim x n] (Bool one-hot bytes or f32), uniforms [n] f64 for
 * stochastic envs or NULL; out: next state, obs(sp) [obs_dim x n] f32, r f32, done u8.            */
int32_t crux_env_step_host(crux_ctx* ctx, int32_t kind, int64_t n, const double* state, const void* action,
                           const double* uniforms, double* next_state, float* obs, float* r,
                           uint8_t* done);

/* advantage pipeline (src/sampler.jl:255-281, src/utils.jl:41-42) --------------------------------- */
/* fill_gae!(d::ExperienceBuffer, V, lambda, gamma) over episodes(d) (sampler.jl:255-273).          */
int32_t crux_fill_gae(crux_buffer* b, crux_mlp* critic, float lambda, float gamma);
/* fill_returns! for every episode of the buffer (sampler.jl:275-281).                             */
int32_t crux_fill_returns(crux_buffer* b, float gamma);
/* fill_gae!(data, ep, ...) / fill_returns!(data, ep, gamma) as terminate_episode! applies them to the block a steps! call produced (sampler.jl:53-57):
 * the same scans restricted to the ring rows [first_row, first_row + n_rows) mod capacity the block was pushed to, whatever else the buffer holds.
 * close_last = steps!(...; reset=true) (:148): the block's last row closes an episode; otherwise the rows of an episode still open at the end of
 * the block get 0, the value mdp_data gave them in the reference's fresh `data` (experience_buffer.jl:14-16).
 * rows_per_env: env-major blocks hold environment e in rows [e*rows_per_env, (e+1)*rows_per_env) of the block; episodes never cross that boundary
 * (0 = the block is one environment's).                                                                                                          */
int32_t crux_fill_gae_rows(crux_buffer* b, crux_mlp* critic, float lambda, float gamma, int64_t first_row, int64_t n_rows, int64_t rows_per_env, int32_t close_last);
int32_t crux_fill_returns_rows(crux_buffer* b, float gamma, int64_t first_row, int64_t n_rows, int64_t rows_per_env, int32_t close_last);
/* the `source=` / `target=` keywords of fill_gae! / fill_returns! (sampler.jl:255,275), as terminate_episode! uses them for cost constraints (:65-66):
 * fill_gae!(data, ep, Vc, lambda, gamma, source=:cost, target=:cost_advantage), fill_returns!(data, ep, gamma, source=:cost, target=:cost_return).
 * source / target: CRUX_COL_* of Float32 one-row columns.                                                                                         */
int32_t crux_fill_gae_keys(crux_buffer* b, crux_mlp* critic, float lambda, float gamma, int32_t source, int32_t target);
int32_t crux_fill_returns_keys(crux_buffer* b, float gamma, int32_t source, int32_t target);
/* :importance_weight of buffer rows [first_row, first_row + n_rows) (no ring wrap): exp.(logpdf(pa, s, a) .- logprob) with the nominal action policy `pa`
 * (step!, src/sampler.jl:108-111; head = CRUX_HEAD_CATEGORICAL | CRUX_HEAD_GAUSSIAN as in crux_train_cfg; needs :logprob and :importance_weight). */
int32_t crux_importance_weight_rows(crux_buffer* b, crux_mlp* nominal, int32_t head, int64_t first_row, int64_t n_rows);
/* fill_fwd_importance_weight! / fill_cum_importance_weight! / fill_rev_importance_weight! (src/sampler.jl:283-308) over the episodes of a block of rows, for
 * whichever of the three columns the buffer has (terminate_episode!, :58-60); block geometry as crux_fill_gae_rows. Rows of an episode left open at the end of a
 * segment keep the value mdp_data gave them (1). */
int32_t crux_fill_importance_weights_rows(crux_buffer* b, int64_t first_row, int64_t n_rows, int64_t rows_per_env, int32_t close_last);
int32_t crux_fill_gae_rows_keys(crux_buffer* b, crux_mlp* critic, float lambda, float gamma, int64_t first_row, int64_t n_rows, int64_t rows_per_env, int32_t close_last, int32_t source, int32_t target);
int32_t crux_fill_returns_rows_keys(crux_buffer* b, float gamma, int64_t first_row, int64_t n_rows, int64_t rows_per_env, int32_t close_last, int32_t source, int32_t target);
/* b[key] .= whiten(b[key]) (utils.jl:41-42; PPO post_batch_callback ppo.jl:61). Bessel-corrected. */
int32_t crux_whiten(crux_buffer* b, int32_t key);

/* evaluation (episodes! / undiscounted_return / discounted_return / failure, src/sampler.jl:175-251): after a rollout of
 * n_envs freshly reset environments for T = max_steps steps into an otherwise empty buffer, the FIRST episode of every
 * environment: sum of rewards, discounted return (reverse Float32 recursion, :223-229), length, and whether it ended within T.
 * Host output arrays [n_envs] (NULL = skip).                                                                          */
int32_t crux_first_episode_metrics(crux_buffer* buf, int32_t n_envs, int64_t T, float gamma, float* undiscounted,
                                   float* discounted, int64_t* length, uint8_t* complete);

/* learner (src/training.jl:1-55, src/model_free/rl/ppo.jl:4-21,59-60) ------------------------------ */
enum { CRUX_LOSS_PPO = 0      /* ppo_loss with the head's logpdf/entropy (ppo.jl:4-21)            */,
       CRUX_LOSS_VALUE_MSE = 1 /* Flux.mse(value(pi, s), return) (ppo.jl:60)                      */,
       CRUX_LOSS_A2C = 3       /* a2c_loss (a2c.jl:4-15): -lambda_p mean(logpdf .* advantage) - lambda_e mean(entropy) */,
       CRUX_LOSS_REINFORCE = 4 /* reinforce_loss (reinforce.jl:4-13): -mean(logpdf .* return); entropy and kl are reported only */,
       CRUX_LOSS_LOGPDF_BC = 5 /* logpdf_bc_loss (il/bc.jl:10-18): -mean(logpdf(pi, s, a)) - lambda_e mean(entropy); needs only :s, :a.
                                  info: LOSS, GRAD_NORM, ENTROPY, KL = the -mean(logpdf) term (info[:logpdf])                    */,
       CRUX_LOSS_MSE_ACTION = 6 /* mse_action_loss (il/bc.jl:1): Flux.mse(action(pi, s), a) for a ContinuousNetwork, mean over act_dim x batch */,
       CRUX_LOSS_LAGRANGE_PPO = 7 /* lagrange_ppo_loss (rl/ppo.jl:70-131): only through crux_batch_train_lagrange (it carries the PID state) */ };

typedef struct {
  int32_t loss;           /* CRUX_LOSS_*                                                          */
  int32_t head;           /* CRUX_HEAD_* for PPO                                                  */
  int32_t batch_size;     /* TrainingParams.batch_size (training.jl:5)                            */
  int32_t epochs;         /* TrainingParams.epochs     (training.jl:6)                            */
  int64_t max_batches;    /* TrainingParams.max_batches, <=0 => Inf (training.jl:10)               */
  float eps_clip;         /* P[:eps]  (ppo.jl:9)                                                  */
  float lambda_p;         /* P[:lp]   (ppo.jl:20)                                                 */
  float lambda_e;         /* P[:le]   (ppo.jl:20)                                                 */
  float target_kl;        /* early_stopping = infos[end][:kl] > target_kl (ppo.jl:59); <0 => off   */
  uint64_t shuffle_seed;  /* Philox key for epoch permutations when perms==NULL                    */
  uint64_t shuffle_counter; /* first epoch's permutation counter (advanced by the caller)          */
  int32_t reserved0;      /* must be 0. (Was `sync_every`, never read: replica groups exchange the gradient EVERY minibatch inside the learner kernel, crux_peer_*; the
                             periodic alternative -- parameters + Adam moments averaged every k epochs -- takes its period as an argument of crux_policy_gradient_training_synced.) */
  int32_t target_col;     /* CRUX_LOSS_VALUE_MSE: the Float32 column the value is regressed on; 0 = :return (ppo.jl:60), CRUX_COL_COST_RETURN for LagrangePPO's cost critic (ppo.jl:210) */
} crux_train_cfg;

/* info keys written by train!/batch_train! (training.jl:22-23,53; ppo.jl:13-19).                   */
enum { CRUX_INFO_LOSS = 0, CRUX_INFO_GRAD_NORM = 1, CRUX_INFO_ENTROPY = 2, CRUX_INFO_KL = 3,
       CRUX_INFO_CLIP_FRACTION = 4, CRUX_INFO_AVG_ADVANTAGE = 5, CRUX_INFO_AVG_RETURN = 6,
       CRUX_INFO_BATCHES_TRAINED = 7, CRUX_INFO_EPOCHS_RUN = 8, CRUX_INFO_Q1AVG = 9, CRUX_INFO_Q2AVG = 10,
       CRUX_INFO_ALPHA = 11 /* "SAC alpha" (sac.jl:47) */,
       /* lagrange_ppo_loss (ppo.jl:111-127): info["penalty"], info["cur_cost"], info["cost_loss"], info["p_loss"] (= lambda_p * p_loss) */
       CRUX_INFO_PENALTY = 12, CRUX_INFO_CUR_COST = 13, CRUX_INFO_COST_LOSS = 14, CRUX_INFO_P_LOSS = 15, CRUX_INFO_N = 16 };

/* batch_train!(pi, p, P, D) (training.jl:28-55): epochs x (shuffle!, partition, train!) with
 * max_batches and early stopping (incl. the aliased-info semantics), executed by
 * one persistent kernel. perms: NULL (Philox permutations) or host int64 [epochs x len] 0-based
 * permutations, one per epoch, applied exactly like shuffle! (new[:,j] = old[:,perm[j]]).
 * The buffer's row ORDER after the call equals the reference's (all epoch shuffles applied).
 * info_out[CRUX_INFO_N]: aggregate over epochs (mean of each epoch's last-minibatch info) +
 * batches_trained. epoch_infos (optional, host [epochs x CRUX_INFO_N]) gets the per-epoch rows.   */
int32_t crux_batch_train(crux_mlp* net, crux_buffer* buf, const crux_train_cfg* cfg, const int64_t* perms,
                         float* info_out, float* epoch_infos);

/* policy_gradient_training(S, D) (src/model_free/on_policy.jl:56-78): batch_train!(actor) then batch_train!(critic) on one
 * buffer. Results are those of the sequential reference order; when the actor's epoch count is fixed in advance (target_kl < 0,
 * no max_batches) the two persistent learner kernels run concurrently on two CUs (the critic composes the actor's shuffles
 * into its starting order), otherwise they run back to back. Learners of the dense engine (shapes outside the register-resident family) run their
 * two chains of launches side by side under the same condition: the critic's on the second learner stream, driven by a second host thread. */
int32_t crux_policy_gradient_training(crux_mlp* actor, crux_mlp* critic, crux_buffer* buf, const crux_train_cfg* cfg_actor,
                                      const crux_train_cfg* cfg_critic, const int64_t* perms_actor, const int64_t* perms_critic,
                                      float* info_actor, float* info_critic, float* epoch_infos_actor, float* epoch_infos_critic);
/* The same for n independent (actor, critic, buffer) triples of equal shape -- multi-seed / population training -- as two batched launches
 * (all critics || all actors; each learner on two CUs, consecutive learners on consecutive XCDs). Replica i shuffles with seed + i.
 * Requires target_kl < 0 and max_batches = 0 (the exact-overlap condition). info_a / info_c: host [n x CRUX_INFO_N] or NULL.            */
/* fill_gae! (+ fill_returns! when with_returns) and whiten for n buffers of a multi-seed run: launches for all buffers are enqueued back to back
 * and the NaN assertion of src/sampler.jl:270 is checked with one host synchronisation for the whole set.                                    */
int32_t crux_fill_gae_multi(int32_t n, crux_buffer* const* bufs, crux_mlp* const* critics, float lambda, float gamma, int32_t with_returns);
int32_t crux_whiten_multi(int32_t n, crux_buffer* const* bufs, int32_t key);
int32_t crux_policy_gradient_training_multi(int32_t n, crux_mlp* const* actors, crux_mlp* const* critics, crux_buffer* const* bufs,
                                            const crux_train_cfg* cfg_actor, const crux_train_cfg* cfg_critic, float* info_a, float* info_c);

/* ---- multi-GPU: environment-shard replicas, one process per GPU, RCCL over xGMI -------------------------------------
 * The reference has no distributed path (Sampler.step!/batch_train! run in one Julia process, src/sampler.jl:137-160,
 * src/training.jl:31-56); these entries are what a Distributed.jl / MPI.jl launcher around it would bind. Rank 0 calls
 * crux_comm_unique_id and ships the 128 bytes to the other ranks by any means; every rank then calls crux_comm_init.
 * crux_allreduce_mean averages a network's parameters AND Adam moments over the group, stream-ordered on the context's stream
 * (no host synchronisation). RCCL is dlopen'ed on first use; without it these return CRUX_ERCCL and nothing else is affected.       */
int32_t crux_comm_unique_id(crux_ctx* ctx, uint8_t* id128 /* [128] out */);
int32_t crux_comm_init(crux_ctx* ctx, int32_t rank, int32_t nranks, const uint8_t* id128);
int32_t crux_comm_destroy(crux_ctx* ctx);
int32_t crux_comm_size(const crux_ctx* ctx);               /* 1 when no communicator is attached */
int32_t crux_allreduce_mean(crux_mlp* net);
/* SUM all-reduce of the flat gradient left by crux_loss_grad (crux_mlp_grads_ptr), stream-ordered; follow with crux_adam_apply(net, 1/nranks):
 * the exact data-parallel minibatch step (global batch = nranks x local batch), every rank applying the identical update.                */
int32_t crux_allreduce_grads(crux_mlp* net);
/* Replica group with direct peer slots -- the exact data-parallel step of SURVEY 8(e): with a group attached, every minibatch step of
 * crux_batch_train / crux_policy_gradient_training SUM-all-reduces the flattened local gradient (and the loss / KL statistics) over the group
 * INSIDE the persistent learner kernel, between the pullback (src/training.jl:18) and Flux.update! (:21), and applies Adam to sum / nranks:
 * nranks replicas with local minibatches of B reproduce one learner with minibatches of nranks x B (up to the summation order), and all replicas
 * hold bit-identical parameters and Adam state at every step. Transport: every rank's gradient is written straight into a slot of each peer's
 * fine-grained region over xGMI (hipIpc-mapped; one hop on the fully connected node) and summed locally in rank order. KL early stopping and
 * max_batches work (every rank sees the same global statistics). Learners with the exchange: the register-resident kernels (IN->64->{64,32}->OUT, batch
 * 65..128: inside the persistent launch) and the dense-engine learner of every other Chain(Dense...) shape (one exchange launch per minibatch between the pullback
 * and the gated Adam, the flat gradient in slot-sized chunks; lagrange_ppo_loss always takes this learner under a group: the controller step first exchanges the
 * minibatch's cost sums, so every replica advances the same controller on the global minibatch); the generic one-workgroup learner returns CRUX_EUNSUP while a
 * group is attached rather than training un-synchronised. Needs equal buffer lengths on all ranks and every rank making the same sequence of training calls.
 *   crux_peer_export  allocates this context's region and returns its 64-byte IPC handle; ship the handles of all ranks to all ranks;
 *   crux_peer_attach  maps the peers' regions (handles: [nranks][64], own entry ignored). All ranks must have attached before any trains.
 *   crux_peer_attach_local wires contexts of one process directly (ctxs[r] = rank r), e.g. two replicas on one device.
 *   nranks = 1 is a group of ONE: the learner runs the replica-group instantiation of its kernel with no peer (sum of one contribution x 1/1). It is the reference
 *   a group on identical shards is compared with bit for bit (N = 2: g + g and the halving are exact), which the un-grouped kernel -- another compilation -- is not.      */
int32_t crux_peer_export(crux_ctx* ctx, uint8_t* handle64);
int32_t crux_peer_attach(crux_ctx* ctx, int32_t rank, int32_t nranks, const uint8_t* handles);
int32_t crux_peer_attach_local(crux_ctx* const* ctxs, int32_t n);
int32_t crux_peer_detach(crux_ctx* ctx);
/* diagnostics of the in-kernel exchange: while enabled every learner workgroup bins how long it waited for the slowest peer's flag at each exchange
 * (log2 bins of 10 ns ticks). out: uint32 [2 learner streams][2 workgroups][32]; reset != 0 clears the bins after reading.                       */
/* periodic form (Crux.jl itself is single-process: this extends SURVEY 8(e)'s k > 1 row into the kernel): k = 1 (default) exchanges the minibatch gradient every step and equals
 * the single learner on the concatenated batch; k > 1 lets every replica take local Adam steps on its shard and averages theta, m and v (sum in rank order x 1/N) after every
 * k-th step inside the persistent kernel -- 3 exchanges per k steps instead of k. Needs the register-resident learner family, no KL early stopping / max_batches, and k must
 * divide the minibatches per epoch so that every call returns with identical replicas; other calls fail with CRUX_EUNSUP / CRUX_EINVAL.                                        */
int32_t crux_peer_set_sync_every(crux_ctx* ctx, int32_t k);
int32_t crux_peer_sync_every(const crux_ctx* ctx);
/* how long a learner workgroup waits for a peer's flag of ONE exchange before it gives up (default 30 000 ms; 1 .. 600 000): the training call then returns CRUX_EHIP and
 * the abort word of every peer is raised, so the surviving ranks of a group whose member died leave their kernels too instead of hanging the GPU.                      */
int32_t crux_peer_set_timeout_ms(crux_ctx* ctx, int32_t ms);
/* The waits of the in-kernel exchange are bounded four ways (csrc/peer_wait.h): the per-exchange timeout above (a peer that is ABSENT); a per-launch budget -- what the flag
 * waits of ONE learner launch may add up to (default 60 000 ms; 0 = none): a peer that is SLOW, e.g. replicas sharing a device whose hardware queues the firmware time-slices
 * answer every exchange after a scheduling quantum and never trip the timeout; a peer's abort word (it LEFT: failure, NaN step); and the host's abort word: crux_peer_abort
 * raises a pinned word that every flag wait of this context polls -- no GPU work, callable from any thread or a signal handler while another thread sits in a training call,
 * which then returns CRUX_EHIP within ~100 us (crux_abort_all: the same for every live context of the process -- a test watchdog, a launcher tearing a job down).
 * An abort is TERMINAL for the group (a rank that left holds other parameters than its peers): detach and attach again. crux_peer_abort_reason reads this rank's abort words
 * (one per learner stream): 0 none, 1 timeout, 2 budget, 3 passed on, 4 host, 5 a peer left on a NaN step -- crux_last_error of the failed call already names it.          */
int32_t crux_peer_set_budget_ms(crux_ctx* ctx, int32_t ms);
int32_t crux_peer_abort(crux_ctx* ctx);
int32_t crux_peer_abort_clear(crux_ctx* ctx);
int32_t crux_abort_all(void);                               /* returns the number of contexts told */
int32_t crux_peer_abort_reason(crux_ctx* ctx, int32_t* out2);
/* COLLECTIVE rendezvous probe: do the replicas answer each other at the speed the in-kernel exchange assumes? Every rank calls it at about the same time after the attach
 * (the first round absorbs up to first_bound_ms of skew between the hosts); one wave per learner stream runs `rounds` (2 .. 100 000) rendezvous through the regions with the
 * exchange's own primitives. out_us4 = {first-round wait, longest later wait} of learner stream 0, then 1, in microseconds: a few us on one device, ~10 us over xGMI;
 * milliseconds mean time-sliced hardware queues (several processes on one GPU) -- a group there runs 100x below its speed (profiles/r06_same_device_oversubscription.txt).
 * CRUX_EHIP when the replicas did not meet within the bounds. crux_peer_attach_local runs it itself and refuses such a group.                                             */
int32_t crux_peer_probe(crux_ctx* ctx, int32_t rounds, int32_t first_bound_ms, int32_t round_bound_ms, float* out_us4);
int32_t crux_peer_hist_enable(crux_ctx* ctx, int32_t on);
int32_t crux_peer_wait_hist(crux_ctx* ctx, uint32_t* out128, int32_t reset);
int32_t crux_peer_size(const crux_ctx* ctx);               /* 1 when no group is attached */
int32_t crux_peer_rank(const crux_ctx* ctx);
/* policy_gradient_training (src/model_free/on_policy.jl:56-78) for replicas: the epochs run in chunks of sync_every, each chunk followed
 * by crux_allreduce_mean(actor), (critic) on the stream. With no communicator it is bit-identical to crux_policy_gradient_training.
 * Requires no KL early stopping / max_batches (replicas must run the same number of epochs) and equal actor/critic epoch counts.     */
int32_t crux_policy_gradient_training_synced(crux_mlp* actor, crux_mlp* critic, crux_buffer* buf, const crux_train_cfg* cfg_a, const crux_train_cfg* cfg_c,
                                             int32_t sync_every, float* info_a /* [CRUX_INFO_N] */, float* info_c);

/* train!(pi, loss, p) (training.jl:13-25): one gradient step on explicit rows `ids` (host, 0-based)
 * of the buffer. Returns CRUX_ENAN (without updating) when the grad norm is NaN (:20).            */
/* LagrangePPO (rl/ppo.jl:70-215): batch_train!(actor, a_opt, P, D) with loss = lagrange_ppo_loss. The loss runs a PID controller on the minibatch's
 * average episode cost EVERY time it is evaluated (ppo.jl:80-116, inside ignore_derivatives): Jc = sum(D[:cost]) / sum(D[:episode_end]) over the
 * minibatch, Delta = Jc - target_cost, I = clamp(I + Ki Delta, 0, Ki_max), exponential smoothing of Delta and Jc with ema_alpha, derivative term
 * max(0, smooth_Jc - Jc_prev), penalty = clamp(Kp smooth_Delta + I + Kd d, 0, penalty_max); then
 *   loss = (lambda_p p_loss + lambda_e e_loss + penalty mean(max(r Ac, clamp(r, 1-eps, 1+eps) Ac))) / (1 + penalty),  Ac = D[:cost_advantage].
 * The one-element arrays the reference keeps in P (I, Jc_prev, smooth_Delta, smooth_Jc, ppo.jl:192-201) are the state fields below: read at the
 * start of the call, advanced on the device once per executed minibatch (a one-block kernel ahead of the loss head on the dense-engine learner; inside
 * the generic learner kernel for tiny networks, inside the register-resident kernels for the 64-wide actors), written back at the end. The buffer needs
 * :logprob, :advantage, :cost, :cost_advantage. info adds CRUX_INFO_PENALTY / CUR_COST / COST_LOSS / P_LOSS of the last minibatch of each epoch.
 * A minibatch without an episode end makes Jc Inf or NaN exactly as in the reference (a NaN loss is CRUX_ENAN, training.jl:20).                  */
typedef struct {
  float target_cost, penalty_max, Ki_max, Ki, Kp, Kd;     /* LagrangePPO keywords (ppo.jl:167-176); penalty_max may be INFINITY                  */
  double ema_alpha;                                         /* Float64 in the reference (0.95): the smoothing is evaluated in Float64, stored Float32 */
  float I, Jc_prev, smooth_delta, smooth_Jc;                /* state                                                                                */
  float penalty, cur_cost, deriv_term, reserved;            /* out: values of the last executed minibatch (info["penalty"], ["cur_cost"], ["deriv_term"]) */
} crux_lagrange;
int32_t crux_batch_train_lagrange(crux_mlp* net, crux_buffer* buf, const crux_train_cfg* cfg, crux_lagrange* lag, const int64_t* perms,
                                  float* info_out, float* epoch_infos);

int32_t crux_train_step(crux_mlp* net, crux_buffer* buf, const crux_train_cfg* cfg, const int64_t* ids,
                        int64_t n, float* info_out);
/* gradient only (no optimiser step): writes the flat gradient to crux_mlp_grads_ptr(net); used by
 * the multi-GPU path (all-reduce between crux_loss_grad and crux_adam_apply) and by parity tests. */
int32_t crux_loss_grad(crux_mlp* net, crux_buffer* buf, const crux_train_cfg* cfg, const int64_t* ids,
                       int64_t n, float* info_out);
int32_t crux_loss_grad_device_ids(crux_mlp* net, crux_buffer* buf, const crux_train_cfg* cfg,
                                  const int32_t* d_ids, int64_t n, float* d_info);
/* Flux.update!(opt, params, grads) from crux_mlp_grads_ptr(net) scaled by `grad_scale`.            */
int32_t crux_adam_apply(crux_mlp* net, float grad_scale);

/* off-policy pieces (src/model_free/rl/dqn.jl:4-6, src/utils.jl:76-87,112) --------------------------- */
/* y = r + gamma*(1-done)*max_a Q_target(sp)   (dqn_target) over the staging buffer -> d_y [len].  */
int32_t crux_dqn_target(crux_mlp* target_net, crux_buffer* batch, float gamma, float* d_y);
/* softq_target(alpha) (rl/softq.jl:4-13): y = r + gamma*(1-done)*alpha*logsumexp(Q_target(sp) ./ alpha) -> d_y [len].        */
int32_t crux_softq_target(crux_mlp* target_net, crux_buffer* batch, float gamma, float alpha, float* d_y);
/* td_error = |Q(s,a) - y| (utils.jl:112) -> d_err [len].                                          */
int32_t crux_td_error(crux_mlp* net, crux_buffer* batch, const float* d_y, float* d_err);
/* train!(critic, td_loss) (utils.jl:76-87): one Adam step on mean((Q(s,a)-y)^2 [.* weight]).       */
int32_t crux_td_step(crux_mlp* net, crux_buffer* batch, const float* d_y, int32_t use_weight,
                     float* info_out /* LOSS, GRAD_NORM, [2]=Qavg */);
/* td_error(pi, D, y) (src/utils.jl:112) and train!(pi, td_loss) (:76-87) of one value_training epoch (off_policy.jl:83-93) in one call: both evaluate
 * Q(s, a) with the same parameters, so the forward pass is shared. d_err (device, [B]) receives |Q - y| for update_priorities!; the step is
 * identical to crux_td_step. The priorities do not enter the loss (the :weight column was written by prioritized_sample!), so updating them after
 * the step instead of before gives the reference's result.                                                                                       */
int32_t crux_td_step_with_error(crux_mlp* net, crux_buffer* batch, const float* d_y, int32_t use_weight, float* d_err, float* info_out);


/* One epoch of value_training (src/model_free/off_policy.jl:69-93) for the DQN family: rand!(batch, source; i) (:71, uniform or prioritized with
 * exponent beta) -> y = dqn_target(target_net, batch) (:80) -> when source is prioritized td_error (:83, shares the forward pass of the step) and
 * update_priorities!(source, batch.indices, td_error) -> train!(net, td_loss[, weight]) (:91-93). Results equal the separate calls
 * (crux_per_sample / crux_uniform_sample, crux_dqn_target, crux_td_step[_with_error], crux_per_update_device) in that order, bit for bit. For networks
 * at least 128 wide the ~25 kernel bodies of the epoch are recorded as ops and run by the executor (csrc/exec.hip): ops that do not depend on each
 * other share a launch (13 phase launches over the whole chip instead of ~25, info rows read back once), or -- CRUX_EXEC_PERSISTENT=1 -- by ONE
 * persistent launch on one XCD with L2 counter barriers between dependent ops. info_out: LOSS, GRAD_NORM, [2] = Qavg.                             */
int32_t crux_dqn_epoch(crux_mlp* net, crux_mlp* target_net, crux_buffer* source, crux_buffer* batch, float gamma, int32_t use_weight, float beta,
                       uint64_t sample_counter, float* info_out);
/* The epoch loop of value_training (off_policy.jl:69: `for epoch in 1:c_opt.epochs`; DQN's c_opt.epochs = dN, rl/dqn.jl) as ONE recorded list: n_epochs epochs back to
 * back, one upload, 10 phase launches per chained epoch (13 phases; the next epoch's sampling and first layer share launches with the optimizer tail), one read-back -- no host round trip between the epochs of an iteration (~60 us each). Epoch e draws with sample counter
 * sample_counter0 + e; infos: host [n_epochs x CRUX_INFO_N]. Same results as n_epochs calls of crux_dqn_epoch.
 * Round 4: networks of the C3 family (L = 3, hidden 128 / 192 / 256, out <= 4) take the tile plan -- 4 launches per chained epoch (exec.hip dqn_epoch_tiles).
 * NaN (training.jl:20 "NaN detected!"): the step that sees it and every later step of the SAME network in the chain leave parameters and Adam state untouched (the status word
 * gates them on the device), but the chain itself runs to its end -- sampling, target-network updates and steps of OTHER networks continue -- and CRUX_ENAN is reported when the
 * chain's rows are read back (at the latest at the end of the call; the *_async forms report it when the caller resolves the info rows). The reference stops at the first NaN
 * step; a caller that needs that state should use the per-epoch entry points.                                                                                       */
int32_t crux_dqn_epochs(crux_mlp* net, crux_mlp* target_net, crux_buffer* source, crux_buffer* batch, float gamma, int32_t use_weight, float beta,
                        uint64_t sample_counter0, int32_t n_epochs, float* infos);
/* The same epoch loop with softq_target(alpha) (rl/softq.jl:4-13) in place of dqn_target: value_training of the SoftQ solver (rl/softq.jl:31-58).             */
/* crux_dqn_epochs without the host in the loop: nothing is read back, nothing is waited for. The info row of epoch e (LOSS, GRAD_NORM, [2] = Qavg) is copied
 * to d_infos[e * CRUX_INFO_N] (DEVICE memory, n_epochs rows) by the recorded list itself; the caller fetches the rows when it wants them (crux_ctx_sync +
 * crux_memcpy_d2h). Meant for the iteration loop of solve(::OffPolicySolver) (off_policy.jl:133-147): the host records and enqueues iteration k + 1 (steps!,
 * value_training, target update) while the device runs iteration k -- up to three chains ahead. A NaN gradient norm skips the update on the device as always
 * (training.jl:20) and shows as NaN in the row; the "NaN detected!" error is the caller's to raise when it reads the rows. CRUX_EUNSUP: the networks do not take
 * the recorded form (narrower than the dense engine's minimum width) -- use crux_dqn_epochs.                                                               */
int32_t crux_dqn_epochs_async(crux_mlp* net, crux_mlp* target_net, crux_buffer* source, crux_buffer* batch, float gamma, int32_t use_weight, float beta,
                              uint64_t sample_counter0, int32_t n_epochs, float* d_infos);
/* value_training of the DQN family INCLUDING its target update (off_policy.jl:66-111: the epoch loop, then `target_update(pi_minus, pi)` once, :108), without the host:
 * the chain of crux_dqn_epochs_async (softq_alpha > 0: of crux_softq_epochs_async) with polyak_average!(target_net, net, tau) (src/utils.jl polyak_average!) as an op of the
 * chain's last phase instead of a launch of its own. tau < 0: no target update. Same results, bit for bit, as the async entry followed by crux_polyak.                   */
int32_t crux_dqn_value_training_async(crux_mlp* net, crux_mlp* target_net, crux_buffer* source, crux_buffer* batch, float gamma, float softq_alpha, int32_t use_weight, float beta,
                                      uint64_t sample_counter0, int32_t n_epochs, float tau, float* d_infos);
int32_t crux_softq_epochs(crux_mlp* net, crux_mlp* target_net, crux_buffer* source, crux_buffer* batch, float gamma, float alpha, int32_t use_weight, float beta,
                          uint64_t sample_counter0, int32_t n_epochs, float* infos);
int32_t crux_softq_epochs_async(crux_mlp* net, crux_mlp* target_net, crux_buffer* source, crux_buffer* batch, float gamma, float alpha, int32_t use_weight, float beta,
                          uint64_t sample_counter0, int32_t n_epochs, float* d_infos);      /* the same without the host in the loop (see crux_dqn_epochs_async) */
/* One epoch of value_training with SAC's pieces (off_policy.jl:69-104, rl/sac.jl:4-52,94-104) as one recorded op list (30 phases, see crux_dqn_epoch): rand! -> sac_target ->
 * train!(log_alpha, sac_temp_loss) -> [update_critic: train!(critic, double_Q_loss)] -> [update_actor: train!(actor, sac_actor_loss), then
 * polyak_average!(target, online, tau) for the actor (when actor_targ != NULL) and both critics (:100)]. The three exploration draws use noise counters
 * noise_counter0, +1, +2 like the separate calls. info_*: host [CRUX_INFO_N] each (NULL = not wanted).                                            */
int32_t crux_sac_epoch(crux_mlp* actor, crux_mlp* q1, crux_mlp* q2, crux_mlp* actor_targ, crux_mlp* q1_targ, crux_mlp* q2_targ, crux_mlp* log_alpha,
                       crux_buffer* source, crux_buffer* batch, float gamma, float H_target, float tau, int32_t use_weight, int32_t update_critic, int32_t update_actor,
                       uint64_t sample_counter, uint64_t noise_seed, uint64_t noise_counter0, float* info_temp, float* info_critic, float* info_actor);

/* The epoch loop of value_training with SAC's pieces in chains of up to 8 epochs per recorded list (SAC's c_opt.epochs = dN = 50, rl/sac.jl): epoch e has global index
 * epoch0 + e within the iteration, trains the critic when that index % critic_every == 0 and the actor (+ target update) when % actor_every == 0 (off_policy.jl:91,96),
 * draws with sample counter sample_counter0 + e and noise counters noise_counter0 + 3 e (+1, +2). infos_*: host [n_epochs x CRUX_INFO_N] (NULL = not wanted). Same
 * results as n_epochs calls of crux_sac_epoch.                                                                                                          */
int32_t crux_sac_epochs(crux_mlp* actor, crux_mlp* q1, crux_mlp* q2, crux_mlp* actor_targ, crux_mlp* q1_targ, crux_mlp* q2_targ, crux_mlp* log_alpha,
                        crux_buffer* source, crux_buffer* batch, float gamma, float H_target, float tau, int32_t use_weight, int32_t epoch0, int32_t n_epochs,
                        int32_t critic_every, int32_t actor_every, uint64_t sample_counter0, uint64_t noise_seed, uint64_t noise_counter0,
                        float* infos_temp, float* infos_critic, float* infos_actor);
/* The same chains without the host in the loop (see crux_dqn_epochs_async): nothing is read back or waited for; d_infos is DEVICE memory, [n_epochs][3][CRUX_INFO_N] =
 * the temperature | critic | actor info rows of every epoch, copied there by the recorded lists themselves (rows of steps an epoch skips stay untouched).          */
int32_t crux_sac_epochs_async(crux_mlp* actor, crux_mlp* q1, crux_mlp* q2, crux_mlp* actor_targ, crux_mlp* q1_targ, crux_mlp* q2_targ, crux_mlp* log_alpha,
                              crux_buffer* source, crux_buffer* batch, float gamma, float H_target, float tau, int32_t use_weight, int32_t epoch0, int32_t n_epochs,
                              int32_t critic_every, int32_t actor_every, uint64_t sample_counter0, uint64_t noise_seed, uint64_t noise_counter0, float* d_infos);

/* The epoch loop of value_training with DDPG's / TD3's pieces (off_policy.jl:69-104 with rl/ddpg.jl:4-11, rl/td3.jl:4-12) in chains of up to 8 epochs per recorded list:
 * rand! -> ddpg_target | td3_target -> [train!(critic, td_loss | double_Q_loss)] -> [train!(actor, -mean(Q(s, mu(s)))) -> polyak_average!(pi_minus, pi, tau)].
 * q2 = q2_targ = NULL: one critic (DDPG). sigma < 0: no target-policy smoothing; otherwise TD3's clamp(a' + clamp(sigma randn, eps_min, eps_max), a_min, a_max) drawn with
 * noise counter noise_counter0 + e. Epoch e has global index epoch0 + e, trains the critic when that index % critic_every == 0 and the actor (+ target update) when
 * % actor_every == 0 (TD3's delayed policy update). Same results as the call-by-call sequence crux_uniform_sample, crux_dpg_target, crux_q_step | crux_double_q_step,
 * crux_dpg_actor_step, crux_polyak.                                                                                                                       */
int32_t crux_dpg_epochs(crux_mlp* actor, crux_mlp* q1, crux_mlp* q2, crux_mlp* actor_targ, crux_mlp* q1_targ, crux_mlp* q2_targ, crux_buffer* source, crux_buffer* batch,
                        float gamma, float tau, float sigma, float eps_min, float eps_max, float a_min, float a_max, int32_t use_weight, int32_t epoch0, int32_t n_epochs,
                        int32_t critic_every, int32_t actor_every, uint64_t sample_counter0, uint64_t noise_seed, uint64_t noise_counter0, float* infos_critic, float* infos_actor);
/* The same chains without the host in the loop (see crux_dqn_epochs_async): d_infos is DEVICE memory, [n_epochs][2][CRUX_INFO_N] = critic | actor rows.      */
int32_t crux_dpg_epochs_async(crux_mlp* actor, crux_mlp* q1, crux_mlp* q2, crux_mlp* actor_targ, crux_mlp* q1_targ, crux_mlp* q2_targ, crux_buffer* source, crux_buffer* batch,
                        float gamma, float tau, float sigma, float eps_min, float eps_max, float a_min, float a_max, int32_t use_weight, int32_t epoch0, int32_t n_epochs,
                        int32_t critic_every, int32_t actor_every, uint64_t sample_counter0, uint64_t noise_seed, uint64_t noise_counter0, float* d_infos);

/* solve(::OffPolicySolver) (src/model_free/off_policy.jl:133-147) for a DQN on a SMALL network (the README example: SimpleGridWorld, 2-8-4), `iters` iterations
 * in ONE launch: per iteration steps!(sampler, buffer, Nsteps = dN, explore = true, i = S.i) (:138), then value_training (:66-111): dN.. `epochs` epochs of
 * rand! (uniform) -> dqn_target -> train!(td_loss), then polyak_average!(target_net, net, tau) (:108). One workgroup runs the loop; the bodies are the ones
 * the separate calls use (crux_rollout's generic kernel, crux_uniform_sample, crux_dqn_target, crux_td_step, crux_polyak), so the results are the same bits.
 * The README shape itself (2-8-4, one environment, B <= 128, a ring that fits LDS) runs wave-resident (parameters in lane registers, replay ring mirrored in LDS; the
 * minibatch gradient is summed in a fixed but different order, so that form agrees with the separate calls to float tolerance instead of bit for bit).
 * CRUX_EUNSUP when the configuration needs the call-by-call loop (prioritized replay, weighted loss, batch > 256 rows, > 4 environments, wide or 64-64 networks).
 * i0 = S.i of the first iteration; infos: host [iters x epochs x CRUX_INFO_N] (LOSS, GRAD_NORM, [2] = Qavg of every epoch).                       */
int32_t crux_dqn_small_solve(crux_mlp* net, crux_mlp* target_net, crux_env* env, const crux_rollout_cfg* cfg, crux_buffer* source, crux_buffer* batch,
                             int32_t iters, int32_t dN, int32_t epochs, float gamma, float tau, int32_t use_weight, uint64_t i0, float* infos,
                             double* sum_r, int64_t* n_episode_end);

/* SAC (src/model_free/rl/sac.jl) -----------------------------------------------------------------------
 * actor: GaussianPolicy handle (mean network + n_extra = act_dim trainable logSigma, policies.jl:315-348);
 * critic: DoubleNetwork = two ContinuousNetwork handles over vcat(s, a) (policies.jl:96,162-187); log_alpha:
 * a bare-vector handle (n_layers 0, n_extra 1; sac.jl:96). `batch` is the staging buffer rand! filled
 * (continuous actions). exploration's randn(Float32, size(mu)) (policies.jl:341) for batch column j, action
 * dim d is randn(Philox(seed, counter, stream = j*act_dim + d, NOISE)); the three draws of one epoch (target,
 * temperature, actor) take three different counters. Every *_step is train! (training.jl:13-25): loss ->
 * gradient -> norm (NaN => CRUX_ENAN, no update) -> Adam. info_out is host [CRUX_INFO_N].                 */
/* sac_target (sac.jl:4-9): y = r + gamma (1-done) (min(Q1-,Q2-)(sp,a') - exp(log_alpha) logprob(a')), a' ~ actor(sp). */
int32_t crux_sac_target(crux_mlp* actor, crux_mlp* q1_targ, crux_mlp* q2_targ, crux_mlp* log_alpha, crux_buffer* batch,
                        float gamma, uint64_t seed, uint64_t counter, float* d_y);
/* train!(params(SAC_log_alpha), sac_temp_loss) (sac.jl:45-52): LOSS, GRAD_NORM, ALPHA (pre-update).        */
int32_t crux_sac_temp_step(crux_mlp* actor, crux_mlp* log_alpha, crux_buffer* batch, float H_target,
                           uint64_t seed, uint64_t counter, float* info_out);
/* train!(critic(pi), double_Q_loss) (utils.jl:89-96): 0.5 (mse(Q1(s,a),y) + mse(Q2(s,a),y)), optional
 * weighted_mean(:weight); one gradient norm over both networks. LOSS, GRAD_NORM, Q1AVG, Q2AVG.             */
int32_t crux_double_q_step(crux_mlp* q1, crux_mlp* q2, crux_buffer* batch, const float* d_y, int32_t use_weight,
                           float* info_out);
/* train!(actor(pi), sac_actor_loss) (sac.jl:34-40): mean(exp(log_alpha) logprob(a) - min(Q1,Q2)(s,a)),
 * a ~ actor(s) reparameterised; LOSS, GRAD_NORM, ENTROPY (= -mean(logprob)).                               */
int32_t crux_sac_actor_step(crux_mlp* actor, crux_mlp* q1, crux_mlp* q2, crux_mlp* log_alpha, crux_buffer* batch,
                            uint64_t seed, uint64_t counter, float* info_out);

/* DDPG / TD3 (src/model_free/rl/ddpg.jl, td3.jl) -----------------------------------------------------------
 * actor: deterministic ContinuousNetwork s -> a; critics: ContinuousNetwork over vcat(s, a).              */
/* OnPolicyGAIL (src/model_free/il/on_policy_gail.jl): train!(D, gail_d_loss(GAN_BCELoss())) on rows [off_ex, off_ex+n_ex) of the expert buffer
 * and [off_pi, off_pi+n_pi) of the policy buffer: L = LBCE(D(vcat(a_ex, s_ex)), 1) + LBCE(D(vcat(a_pi, s_pi)), 0) (:1-5, src/extras/gans.jl:7-9,
 * LBCE = Flux.Losses.logitbinarycrossentropy) -> gradient -> norm (NaN => CRUX_ENAN, no update) -> Adam. The row ranges are the zipped minibatch
 * partitions of batch_train! over two shuffled buffers (src/training.jl:28-44). D maps act_dim + obs_dim -> 1; one-hot actions enter as 0/1.       */
int32_t crux_gail_d_step(crux_mlp* D, crux_buffer* expert, int64_t off_ex, int64_t n_ex, crux_buffer* policy, int64_t off_pi, int64_t n_pi, float* info_out);
/* GAIL_callback reward (:49-55): D_out = value(D, a, s); r = ar*logsigmoid(D_out) - (1-ar)*logcompsigmoid(D_out) (src/utils.jl:140-143);
 * buffer[:r] .= r .* Rscale; *mean_r = mean(r) (info["disc_reward"]). fill_gae!/fill_returns!/whiten follow as separate calls.                      */
int32_t crux_gail_reward(crux_mlp* D, crux_buffer* buf, float alpha_r, float rscale, float* mean_r);

/* ddpg_target (ddpg.jl:6-8) when q2_targ = NULL and smooth_sigma < 0; td3_target (td3.jl:4-7) with both targets and the
 * smoothing policy GaussianNoiseExplorationPolicy(sigma; eps_min, eps_max, a_min, a_max) (policies.jl:510-514):
 * y = r + gamma (1-done) min_k Q_k^-(sp, clamp(mu^-(sp) + clamp(sigma randn, eps_min, eps_max), a_min, a_max)).          */
int32_t crux_dpg_target(crux_mlp* actor_targ, crux_mlp* q1_targ, crux_mlp* q2_targ, crux_buffer* batch, float gamma,
                        float smooth_sigma, float eps_min, float eps_max, float a_min, float a_max,
                        uint64_t seed, uint64_t counter, float* d_y);
/* train!(critic, td_loss) for a critic over vcat(s, a) (utils.jl:76-87): LOSS, GRAD_NORM, Q1AVG.                           */
int32_t crux_q_step(crux_mlp* q, crux_buffer* batch, const float* d_y, int32_t use_weight, float* info_out);
/* train!(actor, ddpg_actor_loss | td3_actor_loss) (ddpg.jl:26, td3.jl:12): -mean(Q(s, mu(s))) with Q = critic (DDPG) or
 * critic.N1 (TD3); only the actor is updated. LOSS, GRAD_NORM.                                                              */
int32_t crux_dpg_actor_step(crux_mlp* actor, crux_mlp* q, crux_buffer* batch, float* info_out);

#ifdef __cplusplus
}
#endif
#endif /* CRUXHIP_H */
