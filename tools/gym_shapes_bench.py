"""step time of the standard Gym shapes added to the feature-split learner's dispatch in round 3 (64-64 hidden), against the dense-engine learner they took before (CRUX_FS=0)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import dense_learner_bench as d

if __name__ == "__main__":
    for label, od, ad, disc, act in (("Acrobot 6 / 3", 6, 3, True, "relu"), ("MountainCar 2 / 3", 2, 3, True, "relu"), ("LunarLanderContinuous 8 / 2", 8, 2, False, "tanh"),
                                     ("Hopper 11 / 3", 11, 3, False, "tanh"), ("BipedalWalker 24 / 4", 24, 4, False, "tanh"), ("Ant 27 / 8", 27, 8, False, "tanh")):
        r = {}
        for fs in ("1", "0"):
            os.environ["CRUX_FS"] = fs; d.crux.reload_switches()
            r[fs] = d.run([od, 64, 64, ad], [act, act, "identity"], disc, od, ad, E=32, T=512, epochs=4)
        print("%-28s %s: k_train_fs2 actor %.2f / critic %.2f us per step; dense engine %.1f / %.1f" % (label, act, r["1"]["actor"], r["1"]["critic"], r["0"]["actor"], r["0"]["critic"]), flush=True)
