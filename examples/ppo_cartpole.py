#!/usr/bin/env python
"""PPO on CartPole-v1 -- the counterpart of the reference's examples/rl/cartpole.jl (PPO part) on cruxhip.

    reference (Julia)                                         this library (Python host mirror over the C ABI)
    A() = DiscreteNetwork(Chain(Dense(4,64,relu), ...), as)   crux.DiscreteNetwork(crux.Chain(crux.Dense(4, 64, "relu"), ...), [1, 2])
    V() = ContinuousNetwork(Chain(Dense(4,64,relu), ..., 1))  crux.ContinuousNetwork(...)
    S = PPO(pi=ActorCritic(A(), V()), S=S, N=N, dN=dN)        crux.PPO(crux.ActorCritic(A(), V()), S, N=N, dN=dN)
    solve(S, mdp)                                             crux.solve(solver, mdp)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crux_jl_amd as crux


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=10)
    ap.add_argument("--envs", type=int, default=32)
    ap.add_argument("--T", type=int, default=2048)
    a = ap.parse_args()
    mdp = crux.CartPoleMDP(n_envs=a.envs, seed=0, discount=0.99)
    S = mdp.state_space()
    A = lambda: crux.DiscreteNetwork(crux.Chain(crux.Dense(4, 64, "relu"), crux.Dense(64, 64, "relu"), crux.Dense(64, 2)), [1, 2], seed=1, stream=0)
    V = lambda: crux.ContinuousNetwork(crux.Chain(crux.Dense(4, 64, "relu"), crux.Dense(64, 64, "relu"), crux.Dense(64, 1)), seed=1, stream=1)
    dN = a.envs * a.T
    solver = crux.PPO(crux.ActorCritic(A(), V()), S, N=a.iterations * dN, dN=dN, max_steps=500, lambda_e=0.1,
                      a_opt={"batch_size": 128, "epochs": 80}, c_opt={"batch_size": 128, "epochs": 80})
    pi = crux.solve(solver, mdp)
    for k, h in enumerate(solver.history):
        print("iter %2d  avg_r %7.2f  actor_loss %+.4f  kl %+.5f  entropy %.4f  critic_loss %9.3f  batches %d" % (
            k, h.get("avg_r", float("nan")), h["actor_loss"], h["kl"], h["entropy"], h["critic_loss"], h["actor_batches_trained"] + h["critic_batches_trained"]))
    smp = crux.Sampler(mdp, pi, max_steps=500)
    print("greedy evaluation over 100 episodes: undiscounted return %.1f" % crux.undiscounted_return(smp, Neps=100))


if __name__ == "__main__":
    main()
