import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CRUX_MFMA_TIMING"] = "1"; os.environ["CRUX_FS"] = "0"
import numpy as np, bench
import crux_jl_amd as crux
ctx = crux.default_context(); ctx.set_learner_cus(1)
pi, buf, sampler = bench.build_problem(crux, 1000)
P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": 0.1}
a_opt = crux.TrainingParams(loss=crux.ppo_loss, batch_size=128, epochs=4, target_kl=None, name="actor_", shuffle_seed=300)
c_opt = crux.TrainingParams(loss=crux.value_mse_loss, batch_size=128, epochs=4, name="critic_", shuffle_seed=400)
import time
nb, info = bench.ppo_iteration(crux, pi, buf, sampler, a_opt, c_opt, P, 0, None)
ctx.sync(); t0 = time.perf_counter()
nb, info = bench.ppo_iteration(crux, pi, buf, sampler, a_opt, c_opt, P, 1, None)
ctx.sync(); print("iteration", time.perf_counter() - t0, nb)
