# CruxHIP.jl -- the reference-side binding a Crux.jl maintainer would add to put libcruxhip.so behind Crux's own API.
#
# NOT executed in this repository's build environment (no Julia toolchain there); the tested twin of every call below is the
# ctypes binding crux.jl_amd/_lib.py + crux.jl_amd/api.py, which uses the same symbols, argument order and struct layouts.
# Every method cites the Crux.jl definition it overloads (paths relative to the Crux.jl repository root).
module CruxHIP

using Crux, Flux, POMDPs

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
    shuffle_seed::UInt64; shuffle_counter::UInt64; sync_every::Int32; reserved::Int32
end
@assert sizeof(RolloutCfg) == 72 && sizeof(TrainCfg) == 64

const NCOLS = 13   # CRUX_NCOLS (cruxhip.h)
const COL = Dict(:s => 0, :a => 1, :sp => 2, :r => 3, :done => 4, :episode_end => 5, :return => 6, :logprob => 7, :advantage => 8, :weight => 9, :t => 10, :i => 11, :value => 12)
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
    ctx::Ctx; h::Ptr{Cvoid}; obs_dim::Int; act_dim::Int; discrete::Bool; keys::Vector{Symbol}
end
function HipBuffer(ctx::Ctx, S, A, capacity::Integer, extras=Symbol[]; prioritized=false, α=0.6f0)
    mask = UInt32(0); for k in extras; mask |= UInt32(1) << COL[k]; end
    r = Ref{Ptr{Cvoid}}(C_NULL); disc = A isa Crux.DiscreteSpace
    check(ctx, ccall((:crux_buffer_create, LIB), Int32, (Ptr{Cvoid}, Int32, Int32, Int32, Int64, UInt32, Int32, Float32, Ref{Ptr{Cvoid}}),
                     ctx.h, prod(Crux.dim(S)), prod(Crux.dim(A)), disc ? 0 : 1, capacity, mask, prioritized, α, r))
    b = HipBuffer(ctx, r[], prod(Crux.dim(S)), prod(Crux.dim(A)), disc, [:s, :a, :sp, :r, :done, :episode_end, extras...])
    finalizer(x -> ccall((:crux_buffer_destroy, LIB), Int32, (Ptr{Cvoid},), x.h), b); b
end
Base.length(b::HipBuffer) = Int(ccall((:crux_buffer_len, LIB), Int64, (Ptr{Cvoid},), b.h))
Base.haskey(b::HipBuffer, k::Symbol) = k in b.keys
function Base.push!(b::HipBuffer, data::Dict{Symbol,<:AbstractArray})          # src/experience_buffer.jl:232-259; returns the 1-based ring indices
    cols = fill(C_NULL, 13); keep = Any[]
    for (k, v) in data; haskey(COL, k) || continue; a = collect(v); push!(keep, a); cols[COL[k] + 1] = pointer(a); end
    N = size(first(values(data)), 2); I = Vector{Int64}(undef, N)
    GC.@preserve keep check(b.ctx, ccall((:crux_buffer_push_host, LIB), Int32, (Ptr{Cvoid}, Int64, Ptr{Ptr{Cvoid}}, Ptr{Int64}), b.h, N, cols, I))
    I .+ 1                                                                     # indices cross the ABI 0-based
end
function Crux.update_priorities!(b::HipBuffer, I::AbstractVector{<:Integer}, v::AbstractVector)   # :290-301
    check(b.ctx, ccall((:crux_per_update, LIB), Int32, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Cvoid}, Int32, Int64), b.h, Int64.(I) .- 1, Float64.(v), 1, length(I)))
end

# ---------------------------------------------------------------------------------------------------- sampler / advantage pipeline
mutable struct HipSampler; ctx::Ctx; h::Ptr{Cvoid}; agent; n_envs::Int; max_steps::Int; γ::Float32; λ::Float32; end
function HipSampler(ctx::Ctx, kind::Integer, agent; n_envs=1, max_steps=100, γ=0.99f0, λ=NaN32, S=nothing, seed=0)   # kind: 0 CartPole-v1, 1 Pendulum-v1, 2 SimpleGridWorld
    μ = isnothing(S) ? C_NULL : Float32.(S.μ); σ = isnothing(S) ? C_NULL : Float32.(S.σ); r = Ref{Ptr{Cvoid}}(C_NULL)
    check(ctx, ccall((:crux_env_create, LIB), Int32, (Ptr{Cvoid}, Int32, Int32, Int32, Float32, Ptr{Float32}, Ptr{Float32}, UInt64, Int32, Int32, Ref{Ptr{Cvoid}}),
                     ctx.h, kind, n_envs, max_steps, γ, μ, σ, seed, 0, 0, r))
    HipSampler(ctx, r[], agent, n_envs, max_steps, γ, λ)
end
function Crux.steps!(s::HipSampler, b::HipBuffer; Nsteps=1, explore=false, i=0, reset=false, cb=(D) -> nothing, kw...)   # src/sampler.jl:139-173
    π = Crux.actor(s.agent.π)
    cfg = RolloutCfg(explore, reset, π.head, 0.0, 0.0, 0, -1f0, -Inf32, Inf32, -Inf32, Inf32, 0f0, i)
    sr = Ref{Float64}(0); ne = Ref{Int64}(0)
    check(s.ctx, ccall((:crux_rollout, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{RolloutCfg}, Ptr{Cvoid}, Int64, Ref{Float64}, Ref{Int64}),
                       s.h, π.h, cfg, b.h, Nsteps ÷ s.n_envs, sr, ne))
    haskey(b, :advantage) && Crux.fill_gae!(b, Crux.critic(s.agent.π), s.λ, s.γ)          # terminate_episode! (:56-57)
    haskey(b, :return) && Crux.fill_returns!(b, s.γ)
    cb(b); Dict("avg_r" => sr[] / ne[])
end
Crux.fill_gae!(b::HipBuffer, V::HipNetwork, λ::Float32, γ::Float32) = check(b.ctx, ccall((:crux_fill_gae, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Float32, Float32), b.h, V.h, λ, γ))   # :255-273
Crux.fill_returns!(b::HipBuffer, γ::Float32) = check(b.ctx, ccall((:crux_fill_returns, LIB), Int32, (Ptr{Cvoid}, Float32), b.h, γ))                                                  # :275-281
whiten!(b::HipBuffer, k::Symbol=:advantage) = check(b.ctx, ccall((:crux_whiten, LIB), Int32, (Ptr{Cvoid}, Int32), b.h, COL[k]))                                                      # src/utils.jl:41-42 via ppo.jl:61

# ---------------------------------------------------------------------------------------------------- learner
loss_id(f) = f === Crux.ppo_loss ? LOSS_PPO : f === Crux.a2c_loss ? LOSS_A2C : f === Crux.reinforce_loss ? LOSS_REINFORCE : f === Crux.logpdf_bc_loss ? LOSS_LOGPDF_BC :
             f === Crux.mse_action_loss ? LOSS_MSE_ACTION : LOSS_VALUE_MSE      # the critic loss of PPO/A2C is an anonymous mse closure (ppo.jl:60)
function train_cfg(π::HipNetwork, p::Crux.TrainingParams, 𝒫; target_kl=-1f0, seed=0, counter=0)
    TrainCfg(loss_id(p.loss), π.head, p.batch_size, p.epochs, isinf(p.max_batches) ? 0 : Int64(p.max_batches),
             get(𝒫, :ϵ, 0.2f0), get(𝒫, :λp, 1f0), get(𝒫, :λe, 0.1f0), target_kl, seed, counter, 0, 0)
end
function Crux.batch_train!(π::HipNetwork, p::Crux.TrainingParams, 𝒫, 𝒟::HipBuffer; info=Dict(), target_kl=-1f0, seed=0, counter=0)   # src/training.jl:28-55
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
                       A.h, C_.h, 𝒟.h, train_cfg(A, 𝒮.a_opt, 𝒮.𝒫; target_kl=get(𝒮.𝒫, :target_kl, -1f0)), train_cfg(C_, 𝒮.c_opt, 𝒮.𝒫), C_NULL, C_NULL, ia, ic, C_NULL, C_NULL))
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

# off-policy seams (value_training, src/model_free/off_policy.jl:66-111) follow the same pattern:
#   dqn_target  -> :crux_dqn_target      td_error -> :crux_td_error        train!(critic, td_loss)        -> :crux_td_step / :crux_q_step
#   sac_target  -> :crux_sac_target      sac_temp_loss -> :crux_sac_temp_step    double_Q_loss -> :crux_double_q_step    sac_actor_loss -> :crux_sac_actor_step
#   ddpg/td3    -> :crux_dpg_target, :crux_dpg_actor_step                  softq_target  -> :crux_softq_target
#   rand!       -> :crux_uniform_sample / :crux_per_sample                 episodes!     -> :crux_rollout over Neps fresh envs + :crux_first_episode_metrics

end # module
