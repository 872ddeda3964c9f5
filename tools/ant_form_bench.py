"""Ant-shaped PPO learner (27-64-64-8 tanh Gaussian actor): the helper-wave form (8 waves per workgroup, 256 registers per wave: 52 spilled) against four waves per workgroup (no spills)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import dense_learner_bench as d
for wg in ("8", "4", "8", "4"):
    os.environ["CRUX_FS_WG"] = wg; d.crux.reload_switches()
    r = d.run([27, 64, 64, 8], ["tanh", "tanh", "identity"], False, 27, 8, E=32, T=512, epochs=4)
    print("27-64-64-8 tanh Gaussian actor, CRUX_FS_WG=%s: actor %.2f us/step, critic %.2f us/step" % (wg, r["actor"], r["critic"]))
