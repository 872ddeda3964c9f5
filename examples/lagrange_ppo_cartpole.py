#!/usr/bin/env python
"""LagrangePPO on CartPole-v1 with a cost channel -- the constrained counterpart of examples/ppo_cartpole.py (reference: src/model_free/rl/ppo.jl:70-215).

The restated CartPole emits cost 1 on every step whose pole angle exceeds 0.05 rad (include/cruxhip.h), so an episode's cost counts the steps spent away
from upright. `lagrange_ppo_loss` estimates the average episode cost from EVERY minibatch (sum(cost) / sum(episode_end), ppo.jl:86), which needs episode
ends in every minibatch: short episodes (max_steps 50) and minibatches of 1 024 keep that estimate finite, and penalty_max bounds the controller.
Plain PPO on the same environment is run next to it: it maximises the return and ignores the cost.

    reference (Julia)                                                        this library
    S = LagrangePPO(pi=ActorCritic(A(), V()), Vc=V(), S=S, target_cost=..)   crux.LagrangePPO(crux.ActorCritic(A(), V()), V(seed), S, target_cost=..)
    solve(S, mdp)                                                            crux.solve(solver, mdp)
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crux_jl_amd as crux


def nets():
    A = crux.DiscreteNetwork(crux.Chain(crux.Dense(4, 64, "relu"), crux.Dense(64, 64, "relu"), crux.Dense(64, 2)), [1, 2], seed=1, stream=0)
    V = lambda st: crux.ContinuousNetwork(crux.Chain(crux.Dense(4, 64, "relu"), crux.Dense(64, 64, "relu"), crux.Dense(64, 1)), seed=1, stream=st)
    return A, V(1), V(2)


def episode_cost(buf):
    c, ee = buf["cost"][0], buf["episode_end"][0]
    return float(c.sum() / max(1, ee.sum()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=15)
    ap.add_argument("--target-cost", type=float, default=8.0)
    a = ap.parse_args()
    E, T, ms = 32, 512, 50
    opt = {"batch_size": 1024, "epochs": 20}
    for name in ("LagrangePPO", "PPO"):
        mdp = crux.CartPoleMDP(n_envs=E, seed=0, discount=0.99)
        S = mdp.state_space(); A, V, Vc = nets(); costs = []
        if name == "LagrangePPO":
            solver = crux.LagrangePPO(crux.ActorCritic(A, V), Vc, S, N=a.iterations * E * T, dN=E * T, max_steps=ms, target_cost=a.target_cost, penalty_max=20.0, Ki=5e-3,
                                      a_opt=dict(opt), c_opt=dict(opt), cost_opt=dict(opt), target_kl=0.02)
        else:
            solver = crux.PPO(crux.ActorCritic(A, V), S, N=a.iterations * E * T, dN=E * T, max_steps=ms, a_opt=dict(opt), c_opt=dict(opt), target_kl=0.02,
                              required_columns=["cost"])
        base_cb = solver.post_batch_callback
        def cb(D, info, base_cb=base_cb, costs=costs):
            costs.append(episode_cost(D)); base_cb(D, info)
        solver.post_batch_callback = cb
        crux.solve(solver, mdp)
        print("== %s (target cost per episode %.1f)" % (name, a.target_cost))
        for k, h in enumerate(solver.history):
            extra = "  penalty %6.3f  cur_cost(minibatch) %6.2f" % (h["penalty"], h["cur_cost"]) if "penalty" in h else ""
            print("iter %2d  avg_r %6.2f  episode cost %6.2f  kl %+.4f%s" % (k, h.get("avg_r", float("nan")), costs[k], h["kl"], extra))
        print("mean episode cost over the last 5 iterations: %.2f; mean return: %.2f" % (np.mean(costs[-5:]), np.mean([h["avg_r"] for h in solver.history[-5:]])))


if __name__ == "__main__":
    main()
