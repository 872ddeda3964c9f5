import os, sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import crux_jl_amd as crux, parity
S = crux.ContinuousSpace(3); acts = ["relu", "relu", "identity"]; B = 256
pi = crux.ActorCritic(crux.GaussianPolicy(parity.chain([3, 256, 256, 1], acts), np.zeros(1, np.float32), seed=2),
                      crux.DoubleNetwork(crux.ContinuousNetwork(parity.chain([4, 256, 256, 1], acts), seed=3), crux.ContinuousNetwork(parity.chain([4, 256, 256, 1], acts), seed=4)))
sv = crux.SAC(pi, S, N=420, dN=6, buffer_size=1000, buffer_init=300, max_steps=50, c_opt={"batch_size": B}, a_opt={"batch_size": B}, SAC_alpha_opt={"batch_size": B})
crux.solve(sv, crux.PendulumMDP(n_envs=1, seed=8))
for h in sv.history[-3:]:
    print({k: (float(v) if np.isscalar(v) else v) for k, v in h.items() if "actor" in k or "entropy" in k or "critic" in k or "temp" in k})
