#!/usr/bin/env python
"""DQN on SimpleGridWorld -- the reference's README example (BASELINE configs[0]): Q network 2 -> 8 -> 4, N = 100000 interactions."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crux_jl_amd as crux


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--N", type=int, default=20000); a = ap.parse_args()
    mdp = crux.SimpleGridWorld(n_envs=1, seed=0)
    S = mdp.state_space()
    Q = crux.DiscreteNetwork(crux.Chain(crux.Dense(2, 8, "relu"), crux.Dense(8, 4)), ["up", "down", "left", "right"], seed=0)
    solver = crux.DQN(Q, S, N=a.N, dN=4, c_opt={"batch_size": 128, "optimizer": crux.Adam(np.float32(1e-3))}, buffer_size=1000, max_steps=100)
    crux.solve(solver, mdp)
    h = solver.history
    print("iterations %d  last critic_loss %.5f  Qavg %.4f" % (len(h), h[-1]["critic_loss"], h[-1]["Qavg"]))
    print("greedy evaluation: discounted return %.3f" % crux.discounted_return(crux.Sampler(mdp, Q, max_steps=100), Neps=100))


if __name__ == "__main__":
    main()
