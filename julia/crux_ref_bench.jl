# crux_ref_bench.jl -- times the REFERENCE (sisl/Crux.jl, CPU path) on the host cores of the box bench.py runs on: SURVEY 8(d)(1), BASELINE.md 3.
# bench.py's julia_reference_probe() starts it as `julia --project=julia julia/crux_ref_bench.jl` when a `julia` binary exists and reads the last JSON
# line it prints. Nothing here is needed by the library: without Julia + Crux.jl + POMDPGym (there is no network on the build or GPU boxes, so they cannot
# be installed there) the probe reports "unavailable" and the C restatement under oracle/ stays the CPU baseline.
#
# Workload = BASELINE.json configs[1], the configuration the headline metric is quoted on: PPO on CartPole-v1, DiscreteNetwork 4-64-64-2 + critic
# 4-64-64-1, 32 environments x 2048 steps per iteration, 80 + 80 epochs at batch 128 (reference constructor: src/model_free/rl/ppo.jl:40-66; sampler
# src/sampler.jl:154-170; batch_train! src/training.jl:28-55). One warm-up iteration (compilation), then ITER timed iterations.
using Printf
const ITER = parse(Int, get(ENV, "CRUX_REF_ITERS", "1"))
try
    @eval using POMDPs, Crux, Flux, POMDPGym
catch e
    println("{\"available\": false, \"why\": \"", replace(sprint(showerror, e), '"' => '\''), "\"}")
    exit(0)
end

n_envs, T, B, EP = 32, 2048, 128, 80
mdp = GymPOMDP(:CartPole, version = :v1)
as = actions(mdp)
S = state_space(mdp)
A() = DiscreteNetwork(Chain(Dense(Crux.dim(S)..., 64, relu), Dense(64, 64, relu), Dense(64, length(as))), as)
V() = ContinuousNetwork(Chain(Dense(Crux.dim(S)..., 64, relu), Dense(64, 64, relu), Dense(64, 1)))

function run(iters)
    # ΔN transitions per iteration come from n_envs samplers stepped round-robin (Sampler(mdps::Vector), src/sampler.jl:16-29)
    solver = PPO(π = ActorCritic(A(), V()), S = S, N = iters * n_envs * T, ΔN = n_envs * T, λ_gae = 0.95f0,
                 a_opt = (epochs = EP, batch_size = B), c_opt = (epochs = EP, batch_size = B), target_kl = Inf32,
                 log = (period = typemax(Int),))
    t = @elapsed solve(solver, [GymPOMDP(:CartPole, version = :v1) for _ in 1:n_envs])
    return t
end

run(1)
t = run(ITER)
steps = ITER * n_envs * T
grad = ITER * 2 * EP * div(n_envs * T, B)
@printf("{\"value\": %.3f, \"unit\": \"env-steps/s\", \"kind\": \"reference\", \"cores\": %d, \"grad_steps_per_s\": %.3f, \"seconds\": %.3f, \"sample\": \"sisl/Crux.jl PPO CartPole-v1, %d envs x %d steps, %d + %d epochs at batch %d, %d iteration(s) after one warm-up\"}\n",
        steps / t, Threads.nthreads(), grad / t, t, n_envs, T, EP, EP, B, ITER)
