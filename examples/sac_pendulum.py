#!/usr/bin/env python
"""SAC on Pendulum-v1 -- the SAC part of the reference's examples/rl/pendulum.jl (GaussianPolicy actor, DoubleNetwork critic), BASELINE configs C4 at 64-wide nets."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crux_jl_amd as crux


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--N", type=int, default=4000); ap.add_argument("--width", type=int, default=64); a = ap.parse_args()
    mdp = crux.PendulumMDP(n_envs=1, seed=0)
    S = mdp.state_space()
    w = a.width
    G = crux.GaussianPolicy(crux.Chain(crux.Dense(3, w, "relu"), crux.Dense(w, w, "relu"), crux.Dense(w, 1)), np.zeros(1, np.float32), seed=1)
    QSA = lambda s: crux.ContinuousNetwork(crux.Chain(crux.Dense(4, w, "relu"), crux.Dense(w, w, "relu"), crux.Dense(w, 1)), seed=s)
    opt = {"batch_size": 100, "optimizer": crux.Adam(np.float32(1e-3))}
    solver = crux.SAC(crux.ActorCritic(G, crux.DoubleNetwork(QSA(2), QSA(3))), S, N=a.N, dN=50, c_opt=dict(opt), a_opt=dict(opt), SAC_alpha_opt=dict(opt),
                      buffer_size=100000, buffer_init=1000, max_steps=200, pi_explore=crux.GaussianNoiseExplorationPolicy(0.5, a_min=-2.0, a_max=2.0))
    crux.solve(solver, mdp)
    h = solver.history[-1]
    print("iterations %d  critic_loss %.4f  actor_loss %.4f  alpha %.4f  entropy %.3f" % (len(solver.history), h["critic_loss"], h["actor_loss"], h["SAC alpha"], h["entropy"]))
    print("evaluation: undiscounted return %.1f" % crux.undiscounted_return(crux.Sampler(mdp, crux.PolicyParams(solver.agent.pi), max_steps=200), Neps=20))


if __name__ == "__main__":
    main()
