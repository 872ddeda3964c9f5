"""step time of the reference's HalfCheetah PPO networks (17 -tanh-> 64 -tanh-> 32 -> 6 Gaussian actor, examples/rl/half_cheetah_mujoco.jl:33-38) on the feature-split learner
(k_train_fs2 with a 32-wide second layer) against the dense-engine learner they took before round 3 (CRUX_FS=0)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import dense_learner_bench as d

if __name__ == "__main__":
    for fs in ("1", "0"):
        os.environ["CRUX_FS"] = fs; d.crux.reload_switches()
        r = d.run([17, 64, 32, 6], ["tanh", "tanh", "identity"], False, 17, 6, E=32, T=512, epochs=4)
        print("17-64-32-6 tanh Gaussian actor, CRUX_FS=%s (%s): actor %.2f us/step, critic (17-64-32-1 tanh tanh) %.2f us/step" % (fs, "feature-split kernel" if fs == "1" else "dense engine", r["actor"], r["critic"]))
