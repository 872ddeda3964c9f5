"""step time of the dense-engine on-policy learner (train_dense.hip) against the generic learner, for shapes outside the IN-64-64-OUT family"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import crux_jl_amd as crux


def run(dims_a, acts, disc, od, ad, E=32, T=512, bs=128, epochs=2, force_generic=False):
    if force_generic:
        os.environ["CRUX_FORCE_GENERIC"] = "1"; crux.reload_switches()
    else:
        os.environ.pop("CRUX_FORCE_GENERIC", None); crux.reload_switches()
    ch = crux.Chain(*[crux.Dense(dims_a[i], dims_a[i + 1], acts[i]) for i in range(len(acts))])
    A = crux.DiscreteNetwork(ch, list(range(1, ad + 1)), seed=1) if disc else crux.GaussianPolicy(ch, np.full(ad, -0.5, np.float32), seed=1)
    cd = dims_a[:-1] + [1]
    Cn = crux.ContinuousNetwork(crux.Chain(*[crux.Dense(cd[i], cd[i + 1], acts[i]) for i in range(len(acts))]), seed=2)
    mdp = crux.SynthMDP(od, ad, discrete=disc, n_envs=E, seed=3)
    extras = ["return", "logprob", "advantage"]
    buf = crux.ExperienceBuffer(crux.ContinuousSpace(od), crux.DiscreteSpace(ad) if disc else crux.ContinuousSpace(ad), E * T, extras)
    s = crux.Sampler(mdp, crux.ActorCritic(A, Cn), max_steps=200, required_columns=extras, lam=0.95)
    crux.steps_(s, buf, Nsteps=E * T, explore=True, i=0, reset=True); crux.whiten_(buf, "advantage")
    P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": 0.0 if not disc else 0.1}
    out = {}
    for name, net, loss in (("actor", A, crux.ppo_loss), ("critic", Cn, crux.value_mse_loss)):
        p = crux.TrainingParams(loss=loss, batch_size=bs, epochs=epochs, target_kl=None, name="x_")
        crux.batch_train_(net, crux.TrainingParams(loss=loss, batch_size=bs, epochs=1, target_kl=None, name="x_", max_batches=4), P, buf)
        buf.ctx.sync(); t0 = time.perf_counter(); info = crux.batch_train_(net, p, P, buf); buf.ctx.sync(); dt = time.perf_counter() - t0
        out[name] = 1e6 * dt / info["x_batches_trained"]
    return out


if __name__ == "__main__":
    tanh3, relu3 = ["tanh", "tanh", "identity"], ["relu", "relu", "identity"]
    for label, dims, acts, disc, od, ad in (("17-256-256-6 gaussian tanh", [17, 256, 256, 6], tanh3, False, 17, 6), ("8-128-128-4 categorical relu", [8, 128, 128, 4], relu3, True, 8, 4),
                                            ("8-64-32-4 categorical relu", [8, 64, 32, 4], relu3, True, 8, 4)):
        d = run(dims, acts, disc, od, ad)
        line = "%-32s dense engine: actor %.1f us/step, critic %.1f us/step" % (label, d["actor"], d["critic"])
        try:
            g = run(dims, acts, disc, od, ad, epochs=1, T=64, force_generic=True)
            line += " | generic learner: actor %.1f, critic %.1f" % (g["actor"], g["critic"])
        except crux.CruxError as e:
            line += " | generic learner: refused (%s)" % str(e)[:60]
        print(line)
