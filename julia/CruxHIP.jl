# CruxHIP.jl -- the reference-side binding a Crux.jl maintainer would add to put libcruxhip.so behind Crux's own API.
#
# NOT executed in this repository's build environment (no Julia toolchain there); the tested twin of every call below is the
# ctypes binding crux.jl_amd/_lib.py + crux.jl_amd/api.py, which uses the same symbols, argument order and struct layouts.
# Every method cites the Crux.jl definition it overloads (paths relative to the Crux.jl repository root).
# `julia julia/check_syntax.jl` parses this file (Meta.parseall) without loading Crux -- the CI-style check for boxes that have a julia binary.
module CruxHIP

using Crux, Flux, POMDPs, Statistics

const LIB = get(ENV, "CRUXHIP_LIB", joinpath(@__DIR__, "..", "crux.jl_amd", "libcruxhip.so"))

# ---------------------------------------------------------------------------------------------------- context, errors
mutable struct Ctx
    h::Ptr{Cvoid}
    function Ctx(device::Integer=0)
        r = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:crux_ctx_create, LIB), Int32, (Int32, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), device, C_NULL, r)
        rc == 0 || error("crux_ctx_create failed ($rc)")
        c = new(r[]); finalizer(x -> ccall((:crux_ctx_destroy, LIB), Int32, (Ptr{Cvoid},), x.h), c); c
    end
end
struct CruxHIPError <: Exception; code::Int32; msg::String; end
function check(c::Ctx, rc::Int32)                                   # CRUX_ENAN == error("NaN detected!") (src/training.jl:20), CRUX_EINVAL == the @asserts
    rc == 0 && return rc
    throw(CruxHIPError(rc, unsafe_string(ccall((:crux_last_error, LIB), Cstring, (Ptr{Cvoid},), c.h))))
end

# isbits mirrors of crux_rollout_cfg (72 bytes) and crux_train_cfg (64 bytes), passed by Ref
struct RolloutCfg
    explore::Int32; reset_at_end::Int32; head::Int32
    eps_start::Float64; eps_stop::Float64; eps_steps::Int64
    noise_sigma::Float32; noise_eps_min::Float32; noise_eps_max::Float32; a_min::Float32; a_max::Float32; logit_div::Float32
    i0::UInt64
end
struct TrainCfg
    loss::Int32; head::Int32; batch_size::Int32; epochs::Int32; max_batches::Int64
    eps_clip::Float32; lambda_p::Float32; lambda_e::Float32; target_kl::Float32
    shuffle_seed::UInt64; shuffle_counter::UInt64; reserved0::Int32; target_col::Int32
end
@assert sizeof(RolloutCfg) == 72 && sizeof(TrainCfg) == 64

const NCOLS = 21   # CRUX_NCOLS (cruxhip.h)
const COL = Dict(:s => 0, :a => 1, :sp => 2, :r => 3, :done => 4, :episode_end => 5, :return => 6, :logprob => 7, :advantage => 8, :weight => 9, :t => 10, :i => 11, :value => 12, :cost => 13, :cost_advantage => 14, :cost_return => 15,
                 :importance_weight => 16, :fwd_importance_weight => 17, :rev_importance_weight => 18, :cum_importance_weight => 19, :traj_importance_weight => 20)
const HEAD_CATEGORICAL, HEAD_GAUSSIAN, HEAD_GREEDY_Q, HEAD_DETERMINISTIC = Int32(0), Int32(1), Int32(2), Int32(3)
const LOSS_PPO, LOSS_VALUE_MSE, LOSS_A2C, LOSS_REINFORCE, LOSS_LOGPDF_BC, LOSS_MSE_ACTION = Int32(0), Int32(1), Int32(3), Int32(4), Int32(5), Int32(6)
const INFO_N = 16

# ---------------------------------------------------------------------------------------------------- networks
# A HipNetwork owns a crux_mlp handle mirroring a Flux Chain(Dense...) (src/policies.jl:68-157); n_extra trailing trainables = GaussianPolicy's logΣ (:315-320)
mutable struct HipNetwork <: Crux.NetworkPolicy
    ctx::Ctx; h::Ptr{Cvoid}; dims::Vector{Int32}; head::Int32; outputs
end
act_id(σ) = σ === relu ? Int32(1) : σ === tanh ? Int32(2) : σ === identity ? Int32(0) : error("activation $σ has no device kernel")
function HipNetwork(ctx::Ctx, chain::Chain; head=HEAD_DETERMINISTIC, logΣ=Float32[], outputs=nothing)
    dims = Int32[size(chain[1].weight, 2); [size(l.weight, 1) for l in chain]...]
    acts = Int32[act_id(l.σ) for l in chain]
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(ctx, ccall((:crux_mlp_create, LIB), Int32, (Ptr{Cvoid}, Int32, Ptr{Int32}, Ptr{Int32}, Int32, Ref{Ptr{Cvoid}}), ctx.h, length(chain), dims, acts, length(logΣ), r))
    flat = Float32[vcat([vec(p) for p in Flux.params(chain)]...); logΣ]        # Flux.params order == the library's flat layout (W out×in column-major, b, ...)
    check(ctx, ccall((:crux_mlp_set_params, LIB), Int32, (Ptr{Cvoid}, Ptr{Float32}, Int64), r[], flat, length(flat)))
    n = HipNetwork(ctx, r[], dims, head, outputs); finalizer(x -> ccall((:crux_mlp_destroy, LIB), Int32, (Ptr{Cvoid},), x.h), n); n
end
HipNetwork(ctx::Ctx, π::Crux.DiscreteNetwork) = HipNetwork(ctx, π.network; head=HEAD_CATEGORICAL, outputs=π.outputs)
HipNetwork(ctx::Ctx, π::Crux.ContinuousNetwork) = HipNetwork(ctx, π.network)
HipNetwork(ctx::Ctx, π::Crux.GaussianPolicy) = HipNetwork(ctx, π.μ.network; head=HEAD_GAUSSIAN, logΣ=vec(Float32.(first(Flux.params(π.logΣ)))))
n_params(π::HipNetwork) = ccall((:crux_mlp_n_params, LIB), Int64, (Ptr{Cvoid},), π.h)
function Flux.params(π::HipNetwork)                                            # flat copy; write back with set_params!
    v = Vector{Float32}(undef, n_params(π)); check(π.ctx, ccall((:crux_mlp_get_params, LIB), Int32, (Ptr{Cvoid}, Ptr{Float32}, Int64), π.h, v, length(v))); v
end
function POMDPs.value(π::HipNetwork, s::AbstractMatrix{Float32})               # src/policies.jl:94,120
    y = Matrix{Float32}(undef, π.dims[end], size(s, 2))
    check(π.ctx, ccall((:crux_mlp_forward_host, LIB), Int32, (Ptr{Cvoid}, Ptr{Float32}, Int64, Ptr{Float32}), π.h, s, size(s, 2), y)); y
end
Crux.polyak_average!(to::HipNetwork, from::HipNetwork, τ=1f0) = check(to.ctx, ccall((:crux_polyak, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Float32), to.h, from.h, τ))   # src/policies.jl:48-59
Base.copyto!(to::HipNetwork, from::HipNetwork) = check(to.ctx, ccall((:crux_mlp_copy, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), to.h, from.h))                          # :61-65
attach!(π::HipNetwork, o::Flux.Optimise.Adam) = check(π.ctx, ccall((:crux_adam_init, LIB), Int32, (Ptr{Cvoid}, Float64, Float64, Float64, Float64), π.h, o.eta, o.beta[1], o.beta[2], o.epsilon))

# ---------------------------------------------------------------------------------------------------- buffer
mutable struct HipBuffer                                                       # stands in for ExperienceBuffer{CuArray} (src/experience_buffer.jl:53-80)
    ctx::Ctx; h::Ptr{Cvoid}; obs_dim::Int; act_dim::Int; discrete::Bool; keys::Vector{Symbol}; prioritized::Bool; β
end
function HipBuffer(ctx::Ctx, S, A, capacity::Integer, extras=Symbol[]; prioritized=false, α=0.6f0, β=Crux.LinearDecaySchedule(0.5f0, 1f0, 10_000))   # PriorityParams (:38-50)
    mask = UInt32(0); for k in extras; mask |= UInt32(1) << COL[k]; end
    r = Ref{Ptr{Cvoid}}(C_NULL); disc = A isa Crux.DiscreteSpace
    check(ctx, ccall((:crux_buffer_create, LIB), Int32, (Ptr{Cvoid}, Int32, Int32, Int32, Int64, UInt32, Int32, Float32, Ref{Ptr{Cvoid}}),
                     ctx.h, prod(Crux.dim(S)), prod(Crux.dim(A)), disc ? 0 : 1, capacity, mask, prioritized, α, r))
    b = HipBuffer(ctx, r[], prod(Crux.dim(S)), prod(Crux.dim(A)), disc, [:s, :a, :sp, :r, :done, :episode_end, extras...], prioritized, β)
    finalizer(x -> ccall((:crux_buffer_destroy, LIB), Int32, (Ptr{Cvoid},), x.h), b); b
end
Base.length(b::HipBuffer) = Int(ccall((:crux_buffer_len, LIB), Int64, (Ptr{Cvoid},), b.h))
Base.haskey(b::HipBuffer, k::Symbol) = k in b.keys
function Base.push!(b::HipBuffer, data::Dict{Symbol,<:AbstractArray})          # src/experience_buffer.jl:232-259; returns the 1-based ring indices
    cols = fill(C_NULL, NCOLS); keep = Any[]
    for (k, v) in data; haskey(COL, k) || continue; a = collect(v); push!(keep, a); cols[COL[k] + 1] = pointer(a); end
    N = size(first(values(data)), 2); I = Vector{Int64}(undef, N)
    GC.@preserve keep check(b.ctx, ccall((:crux_buffer_push_host, LIB), Int32, (Ptr{Cvoid}, Int64, Ptr{Ptr{Cvoid}}, Ptr{Int64}), b.h, N, cols, I))
    I .+ 1                                                                     # indices cross the ABI 0-based
end
Crux.shuffle!(b::HipBuffer; seed=0, counter=0) = check(b.ctx, ccall((:crux_buffer_shuffle, LIB), Int32, (Ptr{Cvoid}, UInt64, UInt64), b.h, seed, counter))   # :118-124 with the library's permutation stream
function priorities(b::HipBuffer)                                              # b.priorities, b.max_priority, b.min_priority (:38-50) as host values
    pr = Vector{Float32}(undef, Crux.capacity(b)); mx = Ref{Float32}(0); mn = Ref{Float32}(0)
    check(b.ctx, ccall((:crux_per_get, LIB), Int32, (Ptr{Cvoid}, Ptr{Float32}, Ref{Float32}, Ref{Float32}, Ptr{Float32}), b.h, pr, mx, mn, C_NULL)); (pr, mx[], mn[])
end
function Crux.update_priorities!(b::HipBuffer, I::AbstractVector{<:Integer}, v::AbstractVector)   # :290-301
    check(b.ctx, ccall((:crux_per_update, LIB), Int32, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Cvoid}, Int32, Int64), b.h, Int64.(I) .- 1, Float64.(v), 1, length(I)))
end

# ---------------------------------------------------------------------------------------------------- sampler / advantage pipeline
mutable struct HipSampler; ctx::Ctx; h::Ptr{Cvoid}; agent; n_envs::Int; max_steps::Int; γ::Float32; λ::Float32; Vc; end      # Vc: the cost value network (Sampler.Vc, sampler.jl:17)
function HipSampler(ctx::Ctx, kind::Integer, agent; n_envs=1, max_steps=100, γ=0.99f0, λ=NaN32, S=nothing, seed=0, Vc=nothing)   # kind: 0 CartPole-v1, 1 Pendulum-v1, 2 SimpleGridWorld
    μ = isnothing(S) ? C_NULL : Float32.(S.μ); σ = isnothing(S) ? C_NULL : Float32.(S.σ); r = Ref{Ptr{Cvoid}}(C_NULL)
    check(ctx, ccall((:crux_env_create, LIB), Int32, (Ptr{Cvoid}, Int32, Int32, Int32, Float32, Ptr{Float32}, Ptr{Float32}, UInt64, Int32, Int32, Ref{Ptr{Cvoid}}),
                     ctx.h, kind, n_envs, max_steps, γ, μ, σ, seed, 0, 0, r))
    HipSampler(ctx, r[], agent, n_envs, max_steps, γ, λ, Vc)
end
# exploration configuration of the agent -> the fields of crux_rollout_cfg (cruxhip.h): the device rollout evaluates them per step, like
# exploration(π_explore, s; π_on, i) does (src/sampler.jl:73, src/policies.jl:474-514)
function explore_fields(π_explore)
    eps_start, eps_stop, eps_steps = 0.0, 0.0, Int64(0)
    σ, ϵmin, ϵmax, amin, amax = -1f0, -Inf32, Inf32, -Inf32, Inf32
    if π_explore isa Crux.MixedPolicy                                   # ϵGreedyPolicy(ϵ, actions) = MixedPolicy(ϵ, ObjectCategorical) (policies.jl:471-472)
        sched = π_explore.ϵ
        if sched isa Crux.LinearDecaySchedule                           # src/utils.jl:115-126
            eps_start, eps_stop, eps_steps = Float64(sched.start), Float64(sched.stop), Int64(sched.steps)
        else                                                            # a constant ϵ: MixedPolicy(ϵ::Real, policy) wraps it as (i) -> ϵ
            e = Float64(sched(0)); e == Float64(sched(10^9)) || error("ϵ schedule $(typeof(sched)) has no device form (LinearDecaySchedule or a constant)")
            eps_start, eps_stop, eps_steps = e, e, Int64(1)
        end
    elseif π_explore isa Crux.GaussianNoiseExplorationPolicy            # policies.jl:498-514
        σ = Float32(π_explore.σ(0)); σ == Float32(π_explore.σ(10^9)) || error("a σ schedule needs one crux_rollout call per value of σ(i)")
        ϵmin, ϵmax, amin, amax = π_explore.ϵ_min, π_explore.ϵ_max, π_explore.a_min[1], π_explore.a_max[1]
    elseif !isnothing(π_explore)
        error("exploration policy $(typeof(π_explore)) has no device form")
    end
    (eps_start, eps_stop, eps_steps, σ, ϵmin, ϵmax, amin, amax)
end
function Crux.steps!(s::HipSampler, b::HipBuffer; Nsteps=1, explore=false, i=0, reset=false, cb=(D) -> nothing, kw...)   # src/sampler.jl:139-173
    π = Crux.actor(s.agent.π)
    e0, e1, en, σ, ϵmin, ϵmax, amin, amax = explore ? explore_fields(s.agent.π_explore) : explore_fields(nothing)
    # explore = 2: action(π, s) of an always_stochastic DiscreteNetwork samples but records logprob NaN (policies.jl:124, sampler.jl:73)
    mode = explore ? Int32(1) : (hasproperty(π, :always_stochastic) && π.always_stochastic ? Int32(2) : Int32(0))
    cfg = RolloutCfg(mode, reset, π.head, e0, e1, en, σ, ϵmin, ϵmax, amin, amax, 0f0, i)
    sr = Ref{Float64}(0); ne = Ref{Int64}(0)
    first = ccall((:crux_buffer_next_ind, LIB), Int64, (Ptr{Cvoid},), b.h)
    check(s.ctx, ccall((:crux_rollout, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{RolloutCfg}, Ptr{Cvoid}, Int64, Ref{Float64}, Ref{Int64}),
                       s.h, π.h, cfg, b.h, Nsteps ÷ s.n_envs, sr, ne))
    # terminate_episode! (:53-69) fills advantage / return per finished episode of the rows just written: the block entry points take the ring
    # position of the block, so a buffer larger than one batch is handled like the reference handles it (rows outside the block are not touched)
    T = Nsteps ÷ s.n_envs
    haskey(b, :advantage) && check(b.ctx, ccall((:crux_fill_gae_rows, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Float32, Float32, Int64, Int64, Int64, Int32),
                                                b.h, Crux.critic(s.agent.π).h, s.λ, s.γ, first, Nsteps, T, reset))
    haskey(b, :return) && check(b.ctx, ccall((:crux_fill_returns_rows, LIB), Int32, (Ptr{Cvoid}, Float32, Int64, Int64, Int64, Int32), b.h, s.γ, first, Nsteps, T, reset))
    # importance weights (:58-62, 108-111): the per-step ratio against the nominal action policy agent.pa, then its running products per episode
    if haskey(b, :importance_weight)
        pa = s.agent.pa; head = pa isa DiscreteNetwork ? Int32(0) : Int32(1); C = capacity(b); n1 = min(Nsteps, C - first)
        check(b.ctx, ccall((:crux_importance_weight_rows, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int64, Int64), b.h, pa.h, head, first, n1))
        n1 < Nsteps && check(b.ctx, ccall((:crux_importance_weight_rows, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int64, Int64), b.h, pa.h, head, 0, Nsteps - n1))
    end
    (haskey(b, :fwd_importance_weight) || haskey(b, :cum_importance_weight) || haskey(b, :rev_importance_weight)) &&
        check(b.ctx, ccall((:crux_fill_importance_weights_rows, LIB), Int32, (Ptr{Cvoid}, Int64, Int64, Int64, Int32), b.h, first, Nsteps, T, reset))
    # cost constraints (:65-66): fill_gae!(data, ep, Vc, λ, γ, source=:cost, target=:cost_advantage), fill_returns!(data, ep, γ, source=:cost, target=:cost_return)
    haskey(b, :cost_advantage) && check(b.ctx, ccall((:crux_fill_gae_rows_keys, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Float32, Float32, Int64, Int64, Int64, Int32, Int32, Int32),
                                                     b.h, s.Vc.h, s.λ, s.γ, first, Nsteps, T, reset, COL[:cost], COL[:cost_advantage]))
    haskey(b, :cost_return) && check(b.ctx, ccall((:crux_fill_returns_rows_keys, LIB), Int32, (Ptr{Cvoid}, Float32, Int64, Int64, Int64, Int32, Int32, Int32),
                                                  b.h, s.γ, first, Nsteps, T, reset, COL[:cost], COL[:cost_return]))
    cb(b); Dict("avg_r" => sr[] / ne[])
end

# ---------------------------------------------------------------------------------------------------- steps! with ANY POMDPs.jl mdp (caller-stepped environments)
# A plain Crux.Sampler -- whatever `mdp` the user handed to solve: LunarLander, MuJoCo, a custom @gen -- filling a HipBuffer with HipNetwork policies. The Sampler struct, its
# state (s, svec, episode_length, was_reset) and the @gen call stay exactly the reference's (sampler.jl:1-43, 89-97); the device does the first line of step! (:73) for the
# sampler's observation -- crux_policy_explore: policy forward, exploration draw, logprob -- and the tail of steps! (:148-155) -- crux_steps_push: push! + terminate_episode!'s
# fills on the rows just written. `samplers` may be one Sampler or the Vector{Sampler} the reference builds for a vector of mdps (sampler.jl:27-29); the block is env-major
# (sampler e owns rows (e-1)*T+1 : e*T, SURVEY §8a R7), the interaction counter env-minor (:161-163). Draw streams: sampler e = Philox stream e-1 of `seed`, counter = the
# steps that sampler has taken (kept here in STEPS_TAKEN).
const STEPS_TAKEN = IdDict{Any,Int64}()
function policy_explore(π::HipNetwork, cfg::RolloutCfg, svec::Matrix{Float32}; seed=0, steps=zeros(Int64, size(svec, 2)))
    E = size(svec, 2); nout = Int(π.dims[end]); disc = π.head == HEAD_CATEGORICAL || cfg.head == HEAD_GREEDY_Q
    a = disc ? zeros(Bool, nout, E) : zeros(Float32, nout, E); lp = Vector{Float32}(undef, E)
    check(π.ctx, ccall((:crux_policy_explore, LIB), Int32, (Ptr{Cvoid}, Ref{RolloutCfg}, Int32, Ptr{Float32}, UInt64, Ptr{Int64}, Ptr{Cvoid}, Ptr{Float32}),
                       π.h, cfg, E, svec, seed, steps, a, lp))
    a, lp
end
function Crux.steps!(samplers::Union{Crux.Sampler,Vector{<:Crux.Sampler}}, b::HipBuffer; Nsteps=1, explore=false, i=0, reset=false, cb=(D) -> nothing, seed=0, kw...)
    ss = samplers isa Crux.Sampler ? [samplers] : samplers; E = length(ss); T = Nsteps ÷ E; s1 = ss[1]
    π = Crux.actor(s1.agent.π)
    e0, e1, en, σ, ϵmin, ϵmax, amin, amax = explore ? explore_fields(s1.agent.π_explore) : explore_fields(nothing)
    mode = explore ? Int32(1) : (hasproperty(π, :always_stochastic) && π.always_stochastic ? Int32(2) : Int32(0))
    head = explore && s1.agent.π_explore isa Crux.MixedPolicy ? HEAD_GREEDY_Q : explore && s1.agent.π_explore isa Crux.GaussianNoiseExplorationPolicy ? HEAD_DETERMINISTIC : π.head
    data = Crux.mdp_data(s1.S, s1.agent.space, Nsteps, Crux.extra_columns(b))                                       # :140
    for t in 1:T
        cfg = RolloutCfg(mode, 0, head, e0, e1, en, σ, ϵmin, ϵmax, amin, amax, 0f0, i + (t - 1) * E)
        svec = reduce(hcat, [Float32.(vec(s.svec)) for s in ss]); steps = Int64[get(STEPS_TAKEN, s, 0) for s in ss]
        A, LP = policy_explore(π, cfg, svec; seed, steps)                                                           # :73 for all samplers
        for (e, s) in enumerate(ss)
            j = (e - 1) * T + t
            s.was_reset = false                                                                                     # :72
            a = π.head == HEAD_CATEGORICAL || head == HEAD_GREEDY_Q ? π.outputs[argmax(A[:, e])] : (size(A, 1) == 1 ? A[1, e] : A[:, e])
            info = Dict(); kwargs = haskey(data, :cost) ? (info=info,) : ()
            sp, r = POMDPs.@gen(:sp, :r)(s.mdp, s.s, a; kwargs...)                                                   # :93  (POMDPs with observations: :90-91, the same two lines)
            spvec = Crux.tovec(POMDPs.convert_s(AbstractArray, sp, s.mdp), s.S)                                     # :95-96
            done = POMDPs.isterminal(s.mdp, sp)                                                                     # :97
            Crux.bslice(data[:s], j:j) .= s.svec; data[:a][:, j:j] .= Crux.tovec(a, s.agent.space); Crux.bslice(data[:sp], j:j) .= spvec   # :100-102
            data[:r][1, j] = r; data[:done][1, j] = done                                                            # :103-104
            haskey(data, :logprob) && (data[:logprob][:, j] .= LP[e])                                               # :107
            haskey(data, :t) && (data[:t][1, j] = s.episode_length + 1)                                             # :112
            haskey(data, :i) && (data[:i][1, j] = i + (t - 1) * E + e)                                              # :113
            haskey(data, :cost) && (data[:cost][1, j] = info["cost"])                                               # :114
            STEPS_TAKEN[s] = get(STEPS_TAKEN, s, 0) + 1
            s.episode_length += 1                                                                                   # :130
            if done || s.episode_length >= s.max_steps                                                              # :131-132: the cut here, terminate_episode!'s fills after the push
                data[:episode_end][1, j] = true; Crux.reset_sampler!(s)
            else
                s.s = sp; s.svec = spvec                                                                            # :134-135
            end
        end
    end
    if reset                                                                                                        # :148
        for (e, s) in enumerate(ss); data[:episode_end][1, e * T] = true; Crux.reset_sampler!(s); end
    end
    cols = fill(C_NULL, NCOLS); keep = Any[]
    for (k, v) in data; haskey(COL, k) || continue; a = collect(v); push!(keep, a); cols[COL[k] + 1] = pointer(a); end
    V = haskey(b, :advantage) ? Crux.critic(s1.agent.π).h : C_NULL; Vc = haskey(b, :cost_advantage) ? s1.Vc.h : C_NULL
    pa = haskey(b, :importance_weight) ? s1.agent.pa : nothing; first = Ref{Int64}(0)
    GC.@preserve keep check(b.ctx, ccall((:crux_steps_push, LIB), Int32,
                                         (Ptr{Cvoid}, Int64, Ptr{Ptr{Cvoid}}, Int64, Int32, Ptr{Cvoid}, Float32, Float32, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ref{Int64}),
                                         b.h, Nsteps, cols, T, reset, V, s1.λ, s1.γ, Vc, isnothing(pa) ? C_NULL : pa.h, isnothing(pa) ? Int32(0) : pa.head, first))
    cb(b); data
end
Crux.fill_gae!(b::HipBuffer, V::HipNetwork, λ::Float32, γ::Float32) = check(b.ctx, ccall((:crux_fill_gae, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Float32, Float32), b.h, V.h, λ, γ))   # :255-273
Crux.fill_returns!(b::HipBuffer, γ::Float32) = check(b.ctx, ccall((:crux_fill_returns, LIB), Int32, (Ptr{Cvoid}, Float32), b.h, γ))                                                  # :275-281
whiten!(b::HipBuffer, k::Symbol=:advantage) = check(b.ctx, ccall((:crux_whiten, LIB), Int32, (Ptr{Cvoid}, Int32), b.h, COL[k]))                                                      # src/utils.jl:41-42 via ppo.jl:61

# evaluation: undiscounted_return / discounted_return / failure (sampler.jl:202-251) over Neps episodes = a rollout of Neps freshly reset device environments for max_steps
# steps into an empty buffer, then the first episode of every environment
function episode_metrics(s::HipSampler, b::HipBuffer; explore=false)
    E = s.n_envs; und, dis = Vector{Float32}(undef, E), Vector{Float32}(undef, E); len = Vector{Int64}(undef, E); complete = Vector{UInt8}(undef, E)
    ccall((:crux_buffer_clear, LIB), Int32, (Ptr{Cvoid},), b.h); ccall((:crux_env_reset, LIB), Int32, (Ptr{Cvoid},), s.h)
    Crux.steps!(s, b; Nsteps=E * s.max_steps, explore=explore)
    check(b.ctx, ccall((:crux_first_episode_metrics, LIB), Int32, (Ptr{Cvoid}, Int32, Int64, Float32, Ptr{Float32}, Ptr{Float32}, Ptr{Int64}, Ptr{UInt8}), b.h, E, s.max_steps, s.γ, und, dis, len, complete))
    (undiscounted=und, discounted=dis, length=len, complete=complete .!= 0)
end
Crux.undiscounted_return(s::HipSampler, b::HipBuffer; kw...) = Statistics.mean(episode_metrics(s, b; kw...).undiscounted)      # :202-212
Crux.discounted_return(s::HipSampler, b::HipBuffer; kw...) = Statistics.mean(episode_metrics(s, b; kw...).discounted)          # :214-232

# ---------------------------------------------------------------------------------------------------- learner
loss_id(f) = f === Crux.ppo_loss ? LOSS_PPO : f === Crux.a2c_loss ? LOSS_A2C : f === Crux.reinforce_loss ? LOSS_REINFORCE : f === Crux.logpdf_bc_loss ? LOSS_LOGPDF_BC :
             f === Crux.mse_action_loss ? LOSS_MSE_ACTION : LOSS_VALUE_MSE      # the critic loss of PPO/A2C is an anonymous mse closure (ppo.jl:60)
# PPO's KL early stop lives in a closure, `early_stopping = (infos) -> (infos[end][:kl] > target_kl)` (ppo.jl:59): the number cannot be read back from
# the TrainingParams, so the solver constructor below records it here and batch_train! / policy_gradient_training look it up (keyword overrides).
const TARGET_KL = IdDict{Any,Float32}()
target_kl_of(p) = get(TARGET_KL, p, -1f0)
"""PPO(...) with the learners on the device: Crux.PPO's own constructor (ppo.jl:40-66) + the target_kl it captured, remembered for the kernel."""
function HipPPO(; target_kl=0.012f0, kwargs...)
    # post_batch_callback: PPO's own `𝒟[:advantage] .= whiten(𝒟[:advantage])` (ppo.jl:61) works on a HipBuffer through ColumnRef (a host round trip of the column);
    # this replaces it by the device kernel of the same arithmetic. A splatted keyword given twice takes the later value, so a caller's own callback still wins.
    𝒮 = Crux.PPO(; target_kl=target_kl, post_batch_callback=(𝒟; kw...) -> (𝒟 isa HipBuffer ? whiten!(𝒟, :advantage) : (𝒟[:advantage] .= Crux.whiten(𝒟[:advantage]))), kwargs...)
    TARGET_KL[𝒮.a_opt] = Float32(target_kl)
    𝒮
end
function train_cfg(π::HipNetwork, p::Crux.TrainingParams, 𝒫; target_kl=target_kl_of(p), seed=0, counter=0)
    TrainCfg(loss_id(p.loss), π.head, p.batch_size, p.epochs, isinf(p.max_batches) ? 0 : Int64(p.max_batches),
             get(𝒫, :ϵ, 0.2f0), get(𝒫, :λp, 1f0), get(𝒫, :λe, 0.1f0), target_kl, seed, counter, 0, 0)
end
function Crux.batch_train!(π::HipNetwork, p::Crux.TrainingParams, 𝒫, 𝒟::HipBuffer; info=Dict(), target_kl=target_kl_of(p), seed=0, counter=0)   # src/training.jl:28-55
    out = zeros(Float32, INFO_N)
    check(π.ctx, ccall((:crux_batch_train, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{TrainCfg}, Ptr{Int64}, Ptr{Float32}, Ptr{Float32}),
                       π.h, 𝒟.h, train_cfg(π, p, 𝒫; target_kl, seed, counter), C_NULL, out, C_NULL))
    info[string(p.name, "loss")] = out[1]; info[string(p.name, "grad_norm")] = out[2]; info[:entropy] = out[3]; info[:kl] = out[4]
    info[string(p.name, "batches_trained")] = Int(out[8]); info
end
function Crux.policy_gradient_training(𝒮::Crux.OnPolicySolver, 𝒟::HipBuffer)                                                          # src/model_free/on_policy.jl:56-78
    A, C_ = Crux.actor(𝒮.agent.π), Crux.critic(𝒮.agent.π); ia, ic = zeros(Float32, INFO_N), zeros(Float32, INFO_N)
    check(A.ctx, ccall((:crux_policy_gradient_training, LIB), Int32,
                       (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{TrainCfg}, Ref{TrainCfg}, Ptr{Int64}, Ptr{Int64}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
                       A.h, C_.h, 𝒟.h, train_cfg(A, 𝒮.a_opt, 𝒮.𝒫), train_cfg(C_, 𝒮.c_opt, 𝒮.𝒫), C_NULL, C_NULL, ia, ic, C_NULL, C_NULL))
    Dict("actor_loss" => ia[1], "actor_grad_norm" => ia[2], :kl => ia[4], :entropy => ia[3], "critic_loss" => ic[1], "critic_grad_norm" => ic[2])
end

# --- replica groups (one Julia process per GPU, e.g. under MPI.jl or Distributed.jl): RCCL communicator owned by the library
comm_unique_id(c::Ctx) = (id = zeros(UInt8, 128); check(c, ccall((:crux_comm_unique_id, LIB), Int32, (Ptr{Cvoid}, Ptr{UInt8}), c.h, id)); id)   # rank 0, then MPI.Bcast!(id, 0, comm)
comm_init!(c::Ctx, rank::Integer, nranks::Integer, id::Vector{UInt8}) = check(c, ccall((:crux_comm_init, LIB), Int32, (Ptr{Cvoid}, Int32, Int32, Ptr{UInt8}), c.h, rank, nranks, id))
allreduce_mean!(π::HipNetwork) = check(π.ctx, ccall((:crux_allreduce_mean, LIB), Int32, (Ptr{Cvoid},), π.h))   # parameters + Adam moments, stream-ordered
function policy_gradient_training_synced(𝒮::Crux.OnPolicySolver, 𝒟::HipBuffer; sync_every=1)                    # on_policy.jl:56-78 for env-shard replicas
    A, C_ = Crux.actor(𝒮.agent.π), Crux.critic(𝒮.agent.π); ia, ic = zeros(Float32, INFO_N), zeros(Float32, INFO_N)
    check(A.ctx, ccall((:crux_policy_gradient_training_synced, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{TrainCfg}, Ref{TrainCfg}, Int32, Ptr{Float32}, Ptr{Float32}),
                       A.h, C_.h, 𝒟.h, train_cfg(A, 𝒮.a_opt, 𝒮.𝒫), train_cfg(C_, 𝒮.c_opt, 𝒮.𝒫), sync_every, ia, ic))
    Dict("actor_loss" => ia[1], "actor_grad_norm" => ia[2], :kl => ia[4], :entropy => ia[3], "critic_loss" => ic[1], "critic_grad_norm" => ic[2])
end

# --- SquashedGaussianPolicy, reservoir pushes, GAIL pieces: thin wrappers over the same handles
set_squash!(π::HipNetwork, ascale::Real) = check(π.ctx, ccall((:crux_mlp_set_squash, LIB), Int32, (Ptr{Cvoid}, Float32), π.h, Float32(ascale)))   # SquashedGaussianPolicy(μ, logΣ, ascale) (policies.jl:353-400)
function Crux.push_reservoir!(b::HipBuffer, data::Dict{Symbol,<:AbstractArray}; weighted=false, seed=0, counter=0)                                # experience_buffer.jl:262-288
    cols = fill(C_NULL, NCOLS); keep = Any[]
    for (k, v) in data; haskey(COL, k) || continue; a = Array(v); push!(keep, a); cols[COL[k] + 1] = pointer(a); end
    GC.@preserve keep check(b.ctx, ccall((:crux_buffer_push_reservoir, LIB), Int32, (Ptr{Cvoid}, Int64, Ptr{Ptr{Cvoid}}, Int32, UInt64, UInt64),
                                         b.h, size(first(values(data)), 2), cols, weighted, seed, counter))
end
gail_d_step!(D::HipNetwork, ex::HipBuffer, r_ex::UnitRange, pol::HipBuffer, r_pol::UnitRange, info=zeros(Float32, INFO_N)) =                     # il/on_policy_gail.jl:1-5 under training.jl:40-44
    (check(D.ctx, ccall((:crux_gail_d_step, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Int64, Int64, Ptr{Float32}),
                        D.h, ex.h, first(r_ex) - 1, length(r_ex), pol.h, first(r_pol) - 1, length(r_pol), info)); info)
gail_reward!(D::HipNetwork, 𝒟::HipBuffer; αr=0.5f0, Rscale=1f0) = (m = Ref{Float32}(0); check(D.ctx, ccall((:crux_gail_reward, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Float32, Float32, Ref{Float32}), D.h, 𝒟.h, αr, Rscale, m)); m[])   # :50-55

# ---------------------------------------------------------------------------------------------------- LagrangePPO (rl/ppo.jl:70-215)
# isbits mirror of crux_lagrange (64 bytes): hyper-parameters, the PID state the reference keeps in one-element arrays of 𝒫 (:192-201), last values
mutable struct Lagrange
    target_cost::Float32; penalty_max::Float32; Ki_max::Float32; Ki::Float32; Kp::Float32; Kd::Float32; ema_α::Float64
    I::Float32; Jc_prev::Float32; smooth_Δ::Float32; smooth_Jc::Float32; penalty::Float32; cur_cost::Float32; deriv_term::Float32; reserved::Float32
end
Lagrange(𝒫) = Lagrange(𝒫[:target_cost], 𝒫[:penalty_max], 𝒫[:Ki_max], 𝒫[:Ki], 𝒫[:Kp], 𝒫[:Kd], 𝒫[:ema_α], 𝒫[:I][1], 𝒫[:Jc_prev][1], 𝒫[:smooth_Δ][1], 𝒫[:smooth_Jc][1], 0, 0, 0, 0)
const LOSS_LAGRANGE_PPO = Int32(7)
"""batch_train!(actor, a_opt, 𝒫, 𝒟) with lagrange_ppo_loss: the penalty controller (ppo.jl:80-116) runs once per minibatch inside the learner kernel;
its state is copied back into 𝒫's arrays afterwards. The sampler side: HipBuffer with :cost, :cost_advantage, :cost_return and steps! filling them
through crux_fill_gae_rows_keys / crux_fill_returns_rows_keys with Vc (sampler.jl:65-66); the cost critic: TrainCfg.target_col = COL[:cost_return]."""
function batch_train_lagrange!(π::HipNetwork, p::Crux.TrainingParams, 𝒫, 𝒟::HipBuffer; info=Dict(), target_kl=target_kl_of(p), seed=0, counter=0)
    lag = Lagrange(𝒫); out = zeros(Float32, INFO_N); c0 = train_cfg(π, p, 𝒫; target_kl, seed, counter)
    cfg = TrainCfg(LOSS_LAGRANGE_PPO, c0.head, c0.batch_size, c0.epochs, c0.max_batches, c0.eps_clip, c0.lambda_p, c0.lambda_e, c0.target_kl, c0.shuffle_seed, c0.shuffle_counter, 0, 0)
    check(π.ctx, ccall((:crux_batch_train_lagrange, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{TrainCfg}, Ref{Lagrange}, Ptr{Int64}, Ptr{Float32}, Ptr{Float32}), π.h, 𝒟.h, cfg, lag, C_NULL, out, C_NULL))
    𝒫[:I][1], 𝒫[:Jc_prev][1], 𝒫[:smooth_Δ][1], 𝒫[:smooth_Jc][1] = lag.I, lag.Jc_prev, lag.smooth_Δ, lag.smooth_Jc
    info["penalty"], info["cur_cost"], info["cost_loss"], info["p_loss"] = out[13], out[14], out[15], out[16]
    info[string(p.name, "loss")] = out[1]; info[:kl] = out[4]; info[:entropy] = out[3]; info
end

# ---------------------------------------------------------------------------------------------------- replica groups over xGMI peer slots
# The exact multi-GPU algorithm (SURVEY 8e): every minibatch step of the persistent learner SUM-all-reduces the gradient between back(1f0) and
# Flux.update! (training.jl:18,21) through peer-mapped slots -- no host call per step. One process per GPU: export, exchange the 64-byte handles
# (MPI.Allgather or any side channel), attach; crux_batch_train / crux_policy_gradient_training then synchronise by themselves.
peer_export(c::Ctx) = (h = zeros(UInt8, 64); check(c, ccall((:crux_peer_export, LIB), Int32, (Ptr{Cvoid}, Ptr{UInt8}), c.h, h)); h)
peer_attach!(c::Ctx, rank::Integer, nranks::Integer, handles::Vector{UInt8}) = check(c, ccall((:crux_peer_attach, LIB), Int32, (Ptr{Cvoid}, Int32, Int32, Ptr{UInt8}), c.h, rank, nranks, handles))
peer_detach!(c::Ctx) = check(c, ccall((:crux_peer_detach, LIB), Int32, (Ptr{Cvoid},), c.h))
# periodic form (round 4): k = 1 exchanges the gradient every minibatch (exact); k > 1: local Adam steps, theta / m / v averaged in the kernel after every k-th
peer_set_sync_every!(c::Ctx, k::Integer) = check(c, ccall((:crux_peer_set_sync_every, LIB), Int32, (Ptr{Cvoid}, Int32), c.h, k))
peer_set_timeout_ms!(c::Ctx, ms::Integer) = check(c, ccall((:crux_peer_set_timeout_ms, LIB), Int32, (Ptr{Cvoid}, Int32), c.h, ms))   # a rank whose peer died gets CRUX_EHIP after this long instead of a hung GPU
# the other bounds of the in-kernel waits (csrc/peer_wait.h): a per-launch wait budget (slow peers), the host's abort word (no GPU work: callable from another task while a
# training call is in flight), the abort reason of a failed call, and the collective rendezvous probe every rank runs after the attach
peer_set_budget_ms!(c::Ctx, ms::Integer) = check(c, ccall((:crux_peer_set_budget_ms, LIB), Int32, (Ptr{Cvoid}, Int32), c.h, ms))
peer_abort!(c::Ctx) = ccall((:crux_peer_abort, LIB), Int32, (Ptr{Cvoid},), c.h)
peer_abort_clear!(c::Ctx) = ccall((:crux_peer_abort_clear, LIB), Int32, (Ptr{Cvoid},), c.h)
abort_all!() = ccall((:crux_abort_all, LIB), Int32, ())
peer_abort_reason(c::Ctx) = (o = zeros(Int32, 2); check(c, ccall((:crux_peer_abort_reason, LIB), Int32, (Ptr{Cvoid}, Ptr{Int32}), c.h, o)); o)
peer_probe(c::Ctx; rounds::Integer=64, first_bound_ms::Integer=2000, round_bound_ms::Integer=20) =
    (o = zeros(Float32, 4); check(c, ccall((:crux_peer_probe, LIB), Int32, (Ptr{Cvoid}, Int32, Int32, Int32, Ptr{Float32}), c.h, rounds, first_bound_ms, round_bound_ms, o)); o)
# multi-seed / population training: n independent (actor, critic, buffer) triples in two batched launches (on_policy.jl:56-78 per triple)
function policy_gradient_training_multi(𝒮::Crux.OnPolicySolver, actors::Vector{HipNetwork}, critics::Vector{HipNetwork}, bufs::Vector{HipBuffer})
    n = length(actors); ia, ic = zeros(Float32, INFO_N, n), zeros(Float32, INFO_N, n)
    check(actors[1].ctx, ccall((:crux_policy_gradient_training_multi, LIB), Int32, (Int32, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Ref{TrainCfg}, Ref{TrainCfg}, Ptr{Float32}, Ptr{Float32}),
                               n, [a.h for a in actors], [c.h for c in critics], [b.h for b in bufs], train_cfg(actors[1], 𝒮.a_opt, 𝒮.𝒫), train_cfg(critics[1], 𝒮.c_opt, 𝒮.𝒫), ia, ic))
    [Dict("actor_loss" => ia[1, k], :kl => ia[4, k], "critic_loss" => ic[1, k]) for k in 1:n]
end
reload_switches!() = ccall((:crux_reload_switches, LIB), Int32, ())      # after changing a CRUX_* switch inside a running process

# ---------------------------------------------------------------------------------------------------- off-policy: value_training and its pieces
# src/model_free/off_policy.jl:66-111. 𝒟 is the staging HipBuffer of batch_size rows, 𝒮.buffer the replay HipBuffer.
device_vec(c::Ctx, n) = (r = Ref{Ptr{Cvoid}}(C_NULL); check(c, ccall((:crux_device_alloc, LIB), Int32, (Ptr{Cvoid}, Int64, Ref{Ptr{Cvoid}}), c.h, 4n, r)); r[])
device_free(c::Ctx, p) = ccall((:crux_device_free, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), c.h, p)

Crux.isprioritized(b::HipBuffer) = b.prioritized                                                                                    # experience_buffer.jl:84
function Crux.uniform_sample!(t::HipBuffer, s::HipBuffer; B=Int(ccall((:crux_buffer_capacity, LIB), Int64, (Ptr{Cvoid},), t.h)), i=1)   # :317-321
    check(t.ctx, ccall((:crux_uniform_sample, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Int64}, UInt64), t.h, s.h, B, C_NULL, i))
end
function Crux.prioritized_sample!(t::HipBuffer, s::HipBuffer; B=Int(ccall((:crux_buffer_capacity, LIB), Int64, (Ptr{Cvoid},), t.h)), i=1)   # :324-349
    β = Float32(s.β(i))                                                                                                             # priority_params.β(i) (:331)
    check(t.ctx, ccall((:crux_per_sample, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Float64}, Float32, UInt64), t.h, s.h, B, C_NULL, β, i))
end
function Base.rand!(t::HipBuffer, sources::HipBuffer...; i=1, fracs=ones(Float32, length(sources)) ./ length(sources))              # :303-315
    ccall((:crux_buffer_clear, LIB), Int32, (Ptr{Cvoid},), t.h)
    lens = Crux.split_batches(Int(ccall((:crux_buffer_capacity, LIB), Int64, (Ptr{Cvoid},), t.h)), fracs)
    for (k, (b, B)) in enumerate(zip(sources, lens))
        check(t.ctx, ccall((:crux_buffer_set_sample_stream, LIB), Int32, (Ptr{Cvoid}, UInt64, UInt32), b.h, 0x5EED5A3F, k - 1))     # independent draws per source
        Crux.isprioritized(b) ? Crux.prioritized_sample!(t, b; B, i) : Crux.uniform_sample!(t, b; B, i)
    end
end

# targets and losses: every device call is the reference's closure of the same name
struct DeviceTarget; p::Ptr{Cvoid}; n::Int; end                                     # y stays on the device between target_fn and the losses
dqn_target(π⁻::HipNetwork, 𝒫, 𝒟::HipBuffer, γ; y=device_vec(π⁻.ctx, length(𝒟)), kw...) =                                           # rl/dqn.jl:4-6
    (check(π⁻.ctx, ccall((:crux_dqn_target, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Float32, Ptr{Cvoid}), π⁻.h, 𝒟.h, γ, y)); DeviceTarget(y, length(𝒟)))
softq_target(α) = (π⁻::HipNetwork, 𝒫, 𝒟::HipBuffer, γ; y=device_vec(π⁻.ctx, length(𝒟)), kw...) ->                                   # rl/softq.jl:4-13
    (check(π⁻.ctx, ccall((:crux_softq_target, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Float32, Float32, Ptr{Cvoid}), π⁻.h, 𝒟.h, γ, α, y)); DeviceTarget(y, length(𝒟)))
function td_step!(π::HipNetwork, 𝒟::HipBuffer, y::DeviceTarget; weighted=false, err=C_NULL, info=zeros(Float32, INFO_N))            # train!(critic, td_loss) utils.jl:76-87 (+ td_error :112)
    check(π.ctx, ccall((:crux_td_step_with_error, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ptr{Cvoid}, Ptr{Float32}), π.h, 𝒟.h, y.p, weighted, err, info)); info
end

"""One epoch of value_training for the DQN family (off_policy.jl:69-93) as one fused call: rand! -> dqn_target -> td_error / update_priorities! ->
train!(critic, td_loss). `i` is 𝒮.i; the sample counter makes every epoch of every iteration draw fresh rows."""
function dqn_epoch!(𝒮, 𝒟::HipBuffer, γ, epoch)
    π, π⁻ = 𝒮.agent.π, 𝒮.agent.π⁻; info = zeros(Float32, INFO_N)
    β = Crux.isprioritized(𝒮.buffer) ? Float32(𝒮.buffer.β(𝒮.i)) : 0f0
    check(π.ctx, ccall((:crux_dqn_epoch, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float32, Int32, Float32, UInt64, Ptr{Float32}),
                       π.h, π⁻.h, 𝒮.buffer.h, 𝒟.h, γ, Crux.isprioritized(𝒮.buffer), β, 𝒮.i * 𝒮.c_opt.epochs + epoch - 1, info))
    Dict("critic_loss" => info[1], "critic_grad_norm" => info[2], "Qavg" => info[3])
end
"""One epoch of value_training with SAC's pieces (off_policy.jl:69-104, rl/sac.jl:4-52,94-104) as one fused call."""
function sac_epoch!(𝒮, 𝒟::HipBuffer, γ, epoch; noise_seed=0)
    A, Q = Crux.actor(𝒮.agent.π), Crux.critic(𝒮.agent.π); A⁻, Q⁻ = Crux.actor(𝒮.agent.π⁻), Crux.critic(𝒮.agent.π⁻)
    it, ic, ia = zeros(Float32, INFO_N), zeros(Float32, INFO_N), zeros(Float32, INFO_N); ctr = 𝒮.i * 𝒮.c_opt.epochs + epoch - 1
    check(A.ctx, ccall((:crux_sac_epoch, LIB), Int32,
                       (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float32, Float32, Float32, Int32, Int32, Int32,
                        UInt64, UInt64, UInt64, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
                       A.h, Q.N1.h, Q.N2.h, C_NULL, Q⁻.N1.h, Q⁻.N2.h, 𝒮.𝒫[:SAC_log_α].h, 𝒮.buffer.h, 𝒟.h, γ, 𝒮.𝒫[:SAC_H_target], 0.005f0,
                       Crux.isprioritized(𝒮.buffer), (epoch - 1) % 𝒮.c_opt.update_every == 0, (epoch - 1) % 𝒮.a_opt.update_every == 0,
                       ctr, noise_seed, 3ctr, it, ic, ia))
    Dict("SAC alpha" => it[3], "critic_loss" => ic[1], "actor_loss" => ia[1], "entropy" => ia[3])
end
"""The epoch loop of value_training with DDPG's / TD3's pieces (off_policy.jl:69-104, rl/ddpg.jl:4-26, rl/td3.jl:4-12) as one chained call: `smooth` is TD3's
target-policy smoothing (σ, ϵmin, ϵmax, amin, amax) or `nothing` (DDPG); a DoubleNetwork critic selects the twin form."""
function dpg_epochs!(𝒮, 𝒟::HipBuffer, γ; smooth=nothing, noise_seed=0)
    A, Q = Crux.actor(𝒮.agent.π), Crux.critic(𝒮.agent.π); A⁻, Q⁻ = Crux.actor(𝒮.agent.π⁻), Crux.critic(𝒮.agent.π⁻)
    twin = hasproperty(Q, :N1); n = 𝒮.c_opt.epochs; ic, ia = zeros(Float32, INFO_N, n), zeros(Float32, INFO_N, n); ctr0 = 𝒮.i * n
    σ, ϵmin, ϵmax, amin, amax = isnothing(smooth) ? (-1f0, 0f0, 0f0, 0f0, 0f0) : Float32.(smooth)
    check(A.ctx, ccall((:crux_dpg_epochs, LIB), Int32,
                       (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float32, Float32, Float32, Float32, Float32, Float32, Float32,
                        Int32, Int32, Int32, Int32, Int32, UInt64, UInt64, UInt64, Ptr{Float32}, Ptr{Float32}),
                       A.h, twin ? Q.N1.h : Q.h, twin ? Q.N2.h : C_NULL, A⁻.h, twin ? Q⁻.N1.h : Q⁻.h, twin ? Q⁻.N2.h : C_NULL, 𝒮.buffer.h, 𝒟.h, γ, 0.005f0, σ, ϵmin, ϵmax, amin, amax,
                       false, 0, n, 𝒮.c_opt.update_every, 𝒮.a_opt.update_every, ctr0, noise_seed, ctr0, ic, ia))
    [Dict("critic_loss" => ic[1, e], "critic_grad_norm" => ic[2, e], "actor_loss" => ia[1, e], "actor_grad_norm" => ia[2, e]) for e in 1:n]
end
# value_training (off_policy.jl:66-111) through the CHAINED epoch calls -- the paths bench.py times. crux_*_epochs_async record the whole `for epoch in 1:c_opt.epochs` loop
# into lists of up to 8 epochs and enqueue them without read-back or synchronisation: the info rows land in device memory, and the iteration loop of `solve`
# (off_policy.jl:133-147) never waits for the device. The rows are fetched (one crux_sync + one copy) only when somebody looks: `resolve` below, called by `solve` when a
# logger is attached and at the end. crux_*_epochs are the same chains with the read-back inside the call (one synchronisation per call). The per-epoch
# functions above are the readable form of one epoch. Tested twin: crux.jl_amd/off_policy.py `value_training` / `_solve_off_policy` (tests/test_gpu_round3.py: async == sync, bit for bit).
mutable struct PendingInfo                       # value_training's info of an iteration whose chain is still on its way
    ctx::Ctx; d_rows::Ptr{Cvoid}; n_epochs::Int; rows_per_epoch::Int; decode::Function; value
end
function resolve(p::PendingInfo)                 # aggregate_info of the epoch rows (off_policy.jl:110), "NaN detected!" (training.jl:20) raised here for the asynchronous chains
    isnothing(p.value) || return p.value
    rows = Array{Float32}(undef, INFO_N, p.rows_per_epoch, p.n_epochs)
    check(p.ctx, ccall((:crux_sync, LIB), Int32, (Ptr{Cvoid},), p.ctx.h))
    check(p.ctx, ccall((:crux_memcpy_d2h, LIB), Int32, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Cvoid}, Int64), p.ctx.h, rows, p.d_rows, sizeof(rows)))
    device_free(p.ctx, p.d_rows)
    any(isnan, rows[2, :, :]) && error("NaN detected! (grad norm is NaN, src/training.jl:20)")
    p.value = Crux.aggregate_info([p.decode(rows[:, :, e]) for e in 1:p.n_epochs])
end
resolve(d::AbstractDict) = d
const EUNSUP = Int32(-6)
function Crux.value_training(𝒮::Crux.OffPolicySolver, 𝒟::HipBuffer, γ; async=isnothing(𝒮.log), noise_seed=0)                       # off_policy.jl:66-111
    n = 𝒮.c_opt.epochs; ctx = 𝒟.ctx; per = Int32(Crux.isprioritized(𝒮.buffer)); β = per == 1 ? Float32(𝒮.buffer.β(𝒮.i)) : 0f0; ctr0 = UInt64(𝒮.i * n)
    sac = haskey(𝒮.𝒫, :SAC_log_α); dpg = !isnothing(𝒮.a_opt) && !sac; softq = haskey(𝒮.𝒫, :alpha) && !sac && !dpg
    rows_per_epoch = sac ? 3 : dpg ? 2 : 1
    d_rows = async ? device_vec(ctx, INFO_N * rows_per_epoch * n) : C_NULL
    if sac
        A, Q = Crux.actor(𝒮.agent.π), Crux.critic(𝒮.agent.π); Q⁻ = Crux.critic(𝒮.agent.π⁻)
        decode = r -> Dict("SAC alpha" => r[3, 1], "critic_loss" => r[1, 2], "actor_loss" => r[1, 3], "entropy" => r[3, 3])
        lα, H, ce, ae = 𝒮.𝒫[:SAC_log_α].h, Float32(𝒮.𝒫[:SAC_H_target]), Int32(𝒮.c_opt.update_every), Int32(𝒮.a_opt.update_every)
        if async
            rc = ccall((:crux_sac_epochs_async, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float32, Float32, Float32,
                                                             Int32, Int32, Int32, Int32, Int32, UInt64, UInt64, UInt64, Ptr{Cvoid}),
                       A.h, Q.N1.h, Q.N2.h, C_NULL, Q⁻.N1.h, Q⁻.N2.h, lα, 𝒮.buffer.h, 𝒟.h, γ, H, 0.005f0, per, 0, n, ce, ae, ctr0, noise_seed, 3ctr0, d_rows)
            rc == EUNSUP || (check(ctx, rc); return PendingInfo(ctx, d_rows, n, 3, decode, nothing))
            device_free(ctx, d_rows)
        end
        it, ic, ia = zeros(Float32, INFO_N, n), zeros(Float32, INFO_N, n), zeros(Float32, INFO_N, n)
        check(ctx, ccall((:crux_sac_epochs, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float32, Float32, Float32,
                                                           Int32, Int32, Int32, Int32, Int32, UInt64, UInt64, UInt64, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
                         A.h, Q.N1.h, Q.N2.h, C_NULL, Q⁻.N1.h, Q⁻.N2.h, lα, 𝒮.buffer.h, 𝒟.h, γ, H, 0.005f0, per, 0, n, ce, ae, ctr0, noise_seed, 3ctr0, it, ic, ia))
        return Crux.aggregate_info([decode(hcat(it[:, e], ic[:, e], ia[:, e])) for e in 1:n])
    elseif dpg
        A, Q = Crux.actor(𝒮.agent.π), Crux.critic(𝒮.agent.π); A⁻, Q⁻ = Crux.actor(𝒮.agent.π⁻), Crux.critic(𝒮.agent.π⁻); twin = hasproperty(Q, :N1)
        smooth = get(𝒮.𝒫, :π_smooth_params, nothing); σ, ϵmin, ϵmax, amin, amax = isnothing(smooth) ? (-1f0, 0f0, 0f0, 0f0, 0f0) : Float32.(smooth)
        decode = r -> Dict("critic_loss" => r[1, 1], "critic_grad_norm" => r[2, 1], "actor_loss" => r[1, 2], "actor_grad_norm" => r[2, 2])
        q1, q2, q1⁻, q2⁻ = twin ? Q.N1.h : Q.h, twin ? Q.N2.h : C_NULL, twin ? Q⁻.N1.h : Q⁻.h, twin ? Q⁻.N2.h : C_NULL
        if async
            rc = ccall((:crux_dpg_epochs_async, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float32, Float32, Float32, Float32, Float32, Float32, Float32,
                                                             Int32, Int32, Int32, Int32, Int32, UInt64, UInt64, UInt64, Ptr{Cvoid}),
                       A.h, q1, q2, A⁻.h, q1⁻, q2⁻, 𝒮.buffer.h, 𝒟.h, γ, 0.005f0, σ, ϵmin, ϵmax, amin, amax, 0, 0, n, 𝒮.c_opt.update_every, 𝒮.a_opt.update_every, ctr0, noise_seed, ctr0, d_rows)
            rc == EUNSUP || (check(ctx, rc); return PendingInfo(ctx, d_rows, n, 2, decode, nothing))
            device_free(ctx, d_rows)
        end
        return Crux.aggregate_info(dpg_epochs!(𝒮, 𝒟, γ; smooth, noise_seed))
    end
    π, π⁻ = 𝒮.agent.π, 𝒮.agent.π⁻                                                                                                   # DQN / SoftQ: one critic, polyak inside the chain
    decode = r -> Dict("critic_loss" => r[1, 1], "critic_grad_norm" => r[2, 1], "Qavg" => r[3, 1])
    infos = zeros(Float32, INFO_N, n)
    if softq
        α = Float32(𝒮.𝒫[:alpha])
        if async
            # the epoch loop and :108's polyak_average!(π⁻, π, 0.005f0) in ONE chain (round 6: the target update rides in the last epoch's final phase)
            rc = ccall((:crux_dqn_value_training_async, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float32, Float32, Int32, Float32, UInt64, Int32, Float32, Ptr{Cvoid}),
                       π.h, π⁻.h, 𝒮.buffer.h, 𝒟.h, γ, α, per, β, ctr0, n, 0.005f0, d_rows)
            rc == EUNSUP || (check(ctx, rc); return PendingInfo(ctx, d_rows, n, 1, decode, nothing))
            device_free(ctx, d_rows)
        end
        check(ctx, ccall((:crux_softq_epochs, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float32, Float32, Int32, Float32, UInt64, Int32, Ptr{Float32}),
                         π.h, π⁻.h, 𝒮.buffer.h, 𝒟.h, γ, α, per, β, ctr0, n, infos))
    else
        if async
            rc = ccall((:crux_dqn_value_training_async, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float32, Float32, Int32, Float32, UInt64, Int32, Float32, Ptr{Cvoid}),
                       π.h, π⁻.h, 𝒮.buffer.h, 𝒟.h, γ, 0f0, per, β, ctr0, n, 0.005f0, d_rows)                              # softq_alpha = 0: dqn_target; tau = 0.005: :108 inside the chain
            rc == EUNSUP || (check(ctx, rc); return PendingInfo(ctx, d_rows, n, 1, decode, nothing))
            device_free(ctx, d_rows)
        end
        check(ctx, ccall((:crux_dqn_epochs, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float32, Int32, Float32, UInt64, Int32, Ptr{Float32}),
                         π.h, π⁻.h, 𝒮.buffer.h, 𝒟.h, γ, per, β, ctr0, n, infos))
    end
    Crux.polyak_average!(π⁻, π, 0.005f0)                                                                                           # :108 with the default target_update (stream-ordered behind the chain)
    Crux.aggregate_info([decode(infos[:, e:e]) for e in 1:n])
end
# The unfused pieces for solvers that compose their own epoch (DDPG / TD3 / custom param_optimizers) follow the same pattern:
#   sac_target -> :crux_sac_target   sac_temp_loss -> :crux_sac_temp_step   double_Q_loss -> :crux_double_q_step   sac_actor_loss -> :crux_sac_actor_step
#   ddpg/td3   -> :crux_dpg_target, :crux_q_step, :crux_dpg_actor_step (fused: dpg_epochs! above)      episodes! -> :crux_rollout over Neps fresh envs + :crux_first_episode_metrics
# solve(::OffPolicySolver) for a small DQN (the README example) in one launch: :crux_dqn_small_solve (off_policy.jl:133-147).

# ---------------------------------------------------------------------------------------------------- user-written losses and the regularizer
# The reference differentiates ANY loss(π, 𝒫, 𝒟) with Zygote (training.jl:16-18). The library's fast paths cover a closed list (loss_id above);
# everything else composes the explicit pullback: forward with cached activations -> the user's d(loss)/d(output) -> parameter gradients ->
# + the regularizer's gradient (training.jl:13) -> Flux.update!. Tested twin: crux.jl_amd/api.py `CustomLoss`, tests/test_gpu_seams.py.
"""train!(π, loss, p) for a loss given as `f(y, mb) -> (l, dl_dy)`, y = value(π, mb[:s]) (training.jl:13-25). `mb` is a Dict of host arrays."""
function train_custom!(π::HipNetwork, f, p::Crux.TrainingParams, mb::Dict; regularizer_grad=nothing, info=Dict())
    c = π.ctx; x = Float32.(mb[:s]); B = size(x, 2); out = Int(π.dims[end])
    d_x, d_y = device_vec(c, length(x)), device_vec(c, out * B)
    try
        check(c, ccall((:crux_memcpy_h2d, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float32}, Int64), c.h, d_x, x, sizeof(x)))
        check(c, ccall((:crux_mlp_forward_cached, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}), π.h, d_x, B, d_y))
        y = Matrix{Float32}(undef, out, B)
        check(c, ccall((:crux_memcpy_d2h, LIB), Int32, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Cvoid}, Int64), c.h, y, d_y, sizeof(y)))
        l, dy = f(y, mb); dy = Float32.(dy)                                                                                        # the piece Zygote would derive
        check(c, ccall((:crux_memcpy_h2d, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float32}, Int64), c.h, d_y, dy, sizeof(dy)))
        check(c, ccall((:crux_mlp_backward, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Float32, Int32, Ptr{Cvoid}), π.h, d_x, B, d_y, 1f0, 1, C_NULL))
        g = Vector{Float32}(undef, n_params(π)); gp = ccall((:crux_mlp_grads_ptr, LIB), Ptr{Cvoid}, (Ptr{Cvoid},), π.h)
        check(c, ccall((:crux_memcpy_d2h, LIB), Int32, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Cvoid}, Int64), c.h, g, gp, sizeof(g)))
        if !isnothing(regularizer_grad)                                                                                            # p.regularizer(π) and its gradient w.r.t. θ
            rv, rg = regularizer_grad(Flux.params(π)); l += rv; g .+= rg
            check(c, ccall((:crux_memcpy_h2d, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float32}, Int64), c.h, gp, g, sizeof(g)))
        end
        gnorm = sqrt(sum(abs2, Float64.(g))); isnan(gnorm) && error("NaN detected! Loss: $l")                                      # training.jl:19-20
        check(c, ccall((:crux_adam_apply, LIB), Int32, (Ptr{Cvoid}, Float32), π.h, 1f0))                                           # Flux.update! (:21)
        info[string(p.name, "loss")] = l; info[string(p.name, "grad_norm")] = Float32(gnorm); info
    finally
        device_free(c, d_x); device_free(c, d_y)
    end
end
# example: a Huber critic loss with an L2 penalty
#   huber(y, mb) = (d = y .- mb[:return]; a = abs.(d); (mean(ifelse.(a .< 1, 0.5f0 .* d .^ 2, a .- 0.5f0)), clamp.(d, -1, 1) ./ length(d)))
#   l2(θ; λ=1f-4) = (λ * sum(abs2, θ), 2λ .* θ)
#   train_custom!(V, huber, 𝒮.c_opt, minibatch(𝒟, 1:128); regularizer_grad=l2)

# ---------------------------------------------------------------------------------------------------- the solve entry seam
# The reference's `solve` builds its own buffer and sampler from the policy's device (src/model_free/on_policy.jl:80-85, off_policy.jl:113-117, src/devices.jl:1-8):
#     𝒟 = ExperienceBuffer(𝒮.S, 𝒮.agent.space, 𝒮.ΔN, 𝒮.required_columns, device=device(𝒮.agent.π));   s = Sampler(mdp, 𝒮.agent, ...)
# Neither constructor can be re-pointed by dispatch (keyword arguments do not dispatch, and re-defining the methods would replace Crux's own), so the hook is `solve` itself,
# on the ENVIRONMENT argument: an `HipMDP` names one of the library's device environments, and `solve(𝒮, ::HipMDP)` below is the reference's loop with the device buffer and
# sampler in the two places where the reference constructs the host ones. Everything inside the loop dispatches on HipSampler / HipBuffer / HipNetwork:
#     steps!(::HipSampler, ::HipBuffer)                     (this file)  <- sampler.jl:139-173
#     𝒮.post_batch_callback(𝒟::HipBuffer)                    getindex / setindex! / `.=` on a HipBuffer column (this file; PPO's whiten callback, ppo.jl:61, runs unchanged)
#     policy_gradient_training(::OnPolicySolver, ::HipBuffer) (this file) <- on_policy.jl:56-78
#     value_training(::OffPolicySolver, ::HipBuffer, γ)       (this file) <- off_policy.jl:66-111
# `solve(𝒮, mdp)` with a plain POMDPs environment is untouched and stays on the CPU path.
"""An environment that lives in the rollout kernel (cruxhip.h CRUX_ENV_*): `kind` 0 CartPole-v1, 1 Pendulum-v1, 2 SimpleGridWorld, 3/4 the synthetic dynamics;
`n_envs` independent-seed copies are stepped together (env-major, SURVEY §8a R7). `host` optionally keeps the POMDPs model for evaluation loggers."""
struct HipMDP{M} <: POMDPs.MDP{Vector{Float32},Any}
    ctx::Ctx; kind::Int32; n_envs::Int; seed::UInt64; γ::Float32; host::M
end
HipMDP(ctx::Ctx, kind::Integer; n_envs=1, seed=0, γ=0.99f0, host=nothing) = HipMDP(ctx, Int32(kind), Int(n_envs), UInt64(seed), Float32(γ), host)
POMDPs.discount(m::HipMDP) = m.γ
HipSampler(m::HipMDP, agent; S=nothing, max_steps=100, λ=NaN32, kw...) = HipSampler(m.ctx, m.kind, agent; n_envs=m.n_envs, max_steps=max_steps, γ=m.γ, λ=λ, S=S, seed=m.seed)

# device(π) (src/devices.jl:1-8, policies.jl:28): a marker the reference's `device(𝒮.agent.π)` call sites can receive
hip(x) = x
Crux.device(π::HipNetwork) = hip
Crux.device(π::Crux.ActorCritic{<:HipNetwork}) = hip
ctx_of(π::HipNetwork) = π.ctx
ctx_of(π) = ctx_of(Crux.actor(π))
# the greedy action for evaluation samplers (policies.jl:124,96): argmax over the outputs of a categorical / Q head, the network output otherwise
function POMDPs.action(π::HipNetwork, s::AbstractVector)
    y = vec(POMDPs.value(π, reshape(Float32.(s), :, 1)))
    (π.head == HEAD_CATEGORICAL || π.head == HEAD_GREEDY_Q) && !isnothing(π.outputs) ? π.outputs[argmax(y)] : y
end
Crux.action_space(π::HipNetwork) = isnothing(π.outputs) ? Crux.ContinuousSpace(Int(π.dims[end])) : Crux.DiscreteSpace(length(π.outputs), π.outputs)

# ---- columns of a HipBuffer as the callbacks of the reference use them: 𝒟[:r], 𝒟[:advantage] .= whiten(𝒟[:advantage]) (ppo.jl:44,61)
col_eltype(b::HipBuffer, k::Symbol) = k in (:done, :episode_end) ? Bool : k === :a && b.discrete ? Bool : k in (:t, :i) ? Int64 : Float32
col_rows(b::HipBuffer, k::Symbol) = k in (:s, :sp) ? b.obs_dim : k === :a ? b.act_dim : 1
function Base.getindex(b::HipBuffer, k::Symbol)                                  # b[key] = view(b.data[key], :, 1:length(b)) (experience_buffer.jl:176) -- a host copy here
    n = length(b); out = Matrix{col_eltype(b, k)}(undef, col_rows(b, k), n)
    check(b.ctx, ccall((:crux_buffer_read_column, LIB), Int32, (Ptr{Cvoid}, Int32, Ptr{Cvoid}, Int64), b.h, COL[k], out, n)); out
end
function Base.setindex!(b::HipBuffer, v, k::Symbol)
    n = length(b); a = Matrix{col_eltype(b, k)}(undef, col_rows(b, k), n); a .= v
    check(b.ctx, ccall((:crux_buffer_write_column, LIB), Int32, (Ptr{Cvoid}, Int32, Ptr{Cvoid}, Int64), b.h, COL[k], a, n)); v
end
struct ColumnRef; b::HipBuffer; k::Symbol; end                                  # the left-hand side of `𝒟[k] .= rhs`
Base.dotview(b::HipBuffer, k::Symbol) = ColumnRef(b, k)
Base.Broadcast.materialize!(dest::ColumnRef, bc::Base.Broadcast.Broadcasted) = (dest.b[dest.k] = Base.Broadcast.materialize(bc); dest)
Base.Broadcast.materialize!(dest::ColumnRef, x::AbstractArray) = (dest.b[dest.k] = x; dest)
Crux.capacity(b::HipBuffer) = Int(ccall((:crux_buffer_capacity, LIB), Int64, (Ptr{Cvoid},), b.h))
Crux.extra_columns(b::HipBuffer) = [k for k in b.keys if !(k in (:s, :a, :sp, :r, :done, :episode_end))]   # experience_buffer.jl:178
"""buffer_like(b; capacity) (experience_buffer.jl:82-85) for the staging buffer of an off-policy solver"""
hip_buffer_like(b::HipBuffer, S, A; capacity) = HipBuffer(b.ctx, S, A, capacity, Crux.extra_columns(b); prioritized=b.prioritized, β=b.β)
"""The replay buffer of an OffPolicySolver on the device: the solver's constructor made a host ExperienceBuffer (off_policy.jl:58); its rows (if any) move over once."""
function to_hip(ctx::Ctx, b::Crux.ExperienceBuffer, S, A)
    extras = [k for k in keys(b.data) if !(k in (:s, :a, :sp, :r, :done, :episode_end))]
    h = HipBuffer(ctx, S, A, Crux.capacity(b), extras; prioritized=Crux.isprioritized(b))
    length(b) > 0 && push!(h, Dict{Symbol,AbstractArray}(k => collect(b[k]) for k in keys(b.data)))
    h
end
to_hip(ctx::Ctx, b::HipBuffer, S, A) = b

"""solve(𝒮::OnPolicySolver, mdp) (src/model_free/on_policy.jl:80-109) for a device environment. Requires 𝒮.agent.π = ActorCritic(HipNetwork, HipNetwork)."""
function POMDPs.solve(𝒮::Crux.OnPolicySolver, mdp::HipMDP)
    ctx = ctx_of(𝒮.agent.π)
    𝒟 = HipBuffer(ctx, 𝒮.S, 𝒮.agent.space, 𝒮.ΔN, Symbol[𝒮.required_columns...])                                # :82  ExperienceBuffer(...; device=device(π))
    γ, λ = Float32(POMDPs.discount(mdp)), 𝒮.λ_gae                                                               # :83
    s = HipSampler(mdp, 𝒮.agent; S=𝒮.S, λ=λ, max_steps=𝒮.max_steps)                                            # :84  Sampler(mdp, 𝒮.agent, ...)
    evaluating = !isnothing(𝒮.log) && !isnothing(mdp.host)                                                      # the reference's loggers roll out on a host Sampler (logging.jl:73)
    evaluating && isnothing(𝒮.log.sampler) && (𝒮.log.sampler = Crux.Sampler(mdp.host, 𝒮.agent, S=𝒮.S, max_steps=𝒮.max_steps))   # :85
    evaluating && Crux.log(𝒮.log, 𝒮.i, 𝒮=𝒮)                                                                    # :88
    for 𝒮.i = range(𝒮.i, stop=𝒮.i + 𝒮.N - 𝒮.ΔN, step=𝒮.ΔN)                                                     # :91
        info = Dict()
        Crux.steps!(s, 𝒟, Nsteps=𝒮.ΔN, explore=true, i=𝒮.i, reset=true, cb=(D) -> 𝒮.post_sample_callback(D, info=info, 𝒮=𝒮))       # :96  -> steps!(::HipSampler, ::HipBuffer)
        𝒮.post_batch_callback(𝒟, info=info, 𝒮=𝒮)                                                                # :99  PPO: 𝒟[:advantage] .= whiten(𝒟[:advantage]) through ColumnRef
        training_info = Crux.policy_gradient_training(𝒮, 𝒟)                                                    # :102 -> policy_gradient_training(::OnPolicySolver, ::HipBuffer)
        evaluating && Crux.log(𝒮.log, 𝒮.i + 1:𝒮.i + 𝒮.ΔN, training_info, info, 𝒮=𝒮)                             # :105
    end
    𝒮.i += 𝒮.ΔN                                                                                                 # :107
    𝒮.agent.π
end

"""solve(𝒮::OffPolicySolver, mdp) (src/model_free/off_policy.jl:113-150) for a device environment. Requires HipNetwork policies in 𝒮.agent (π, π⁻)."""
function POMDPs.solve(𝒮::Crux.OffPolicySolver, mdp::HipMDP)
    ctx = ctx_of(𝒮.agent.π)
    𝒮.buffer = to_hip(ctx, 𝒮.buffer, 𝒮.S, 𝒮.agent.space)                                                        # the replay ring the constructor made (:58), on the device
    𝒟 = hip_buffer_like(𝒮.buffer, 𝒮.S, 𝒮.agent.space; capacity=𝒮.c_opt.batch_size)                              # :115 buffer_like(𝒮.buffer, capacity=batch_size, device=device(π))
    γ = Float32(POMDPs.discount(mdp))                                                                           # :116
    s = HipSampler(mdp, 𝒮.agent; S=𝒮.S, max_steps=𝒮.max_steps)                                                 # :117 Sampler(mdp, 𝒮.agent, ...)
    evaluating = !isnothing(𝒮.log) && !isnothing(mdp.host)
    evaluating && isnothing(𝒮.log.sampler) && (𝒮.log.sampler = Crux.Sampler(mdp.host, 𝒮.agent, S=𝒮.S, max_steps=𝒮.max_steps))   # :118
    info = Dict()
    Nfill = max(0, 𝒮.buffer_init - length(𝒮.buffer)); istart = 𝒮.i                                              # :122-123
    if Nfill > 0
        𝒮.i += Nfill                                                                                            # :125
        Crux.steps!(s, 𝒮.buffer, Nsteps=Nfill, explore=true, i=𝒮.i, cb=(D) -> 𝒮.post_sample_callback(D, 𝒮=𝒮, info=info))             # :126
    end
    evaluating && Crux.log(𝒮.log, 𝒮.i, info, 𝒮=𝒮)                                                              # :130
    pending = PendingInfo[]          # asynchronous chains whose info rows nobody has fetched yet. Declared OUTSIDE the loop (a name first assigned in a `for` body is local to it: ADVICE r5) and bounded
    for 𝒮.i in range(𝒮.i, stop=istart + 𝒮.N - 𝒮.ΔN, step=𝒮.ΔN)                                                 # :133
        info = Dict()
        Crux.steps!(s, 𝒮.buffer, Nsteps=𝒮.ΔN, explore=true, i=𝒮.i, cb=(D) -> 𝒮.post_sample_callback(D, 𝒮=𝒮, info=info))            # :138
        𝒮.pre_train_callback(𝒮, info=info)                                                                      # :140
        training_info = Crux.value_training(𝒮, 𝒟, γ)                                                            # :143 -> value_training(::OffPolicySolver, ::HipBuffer, γ): the asynchronous chains when nobody logs
        evaluating && Crux.log(𝒮.log, 𝒮.i + 1:𝒮.i + 𝒮.ΔN, resolve(training_info), info, 𝒮=𝒮)                    # :146
        training_info isa PendingInfo && isnothing(training_info.value) && push!(pending, training_info)
        length(pending) >= 64 && (foreach(resolve, pending); empty!(pending))                                   # one synchronisation per 64 iterations: the device rows are freed and a NaN of an earlier iteration surfaces (training.jl:20)
    end
    foreach(resolve, pending)                                                                                   # every chain's rows fetched and freed, "NaN detected!" raised for any of them, before solve returns
    𝒮.i += 𝒮.ΔN                                                                                                 # :148
    𝒮.agent.π
end
# What a user writes (README / examples/rl/cartpole.jl, three changed lines -- the environment and the two networks):
#     ctx = CruxHIP.Ctx(0); mdp = CruxHIP.HipMDP(ctx, 0; n_envs=32, host=GymPOMDP(:CartPole))
#     A() = CruxHIP.HipNetwork(ctx, DiscreteNetwork(Chain(Dense(4, 64, relu), Dense(64, 64, relu), Dense(64, 2)), [1, 2]))
#     V() = CruxHIP.HipNetwork(ctx, ContinuousNetwork(Chain(Dense(4, 64, relu), Dense(64, 64, relu), Dense(64, 1))))
#     solve(CruxHIP.HipPPO(π=ActorCritic(A(), V()), S=state_space(mdp.host), N=65536 * 20, ΔN=65536), mdp)      # dispatches to the method above

end # module
