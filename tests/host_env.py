"""The restated CartPole as a HOST environment (test helper, not a test module): the four functions a crux.HostMDP needs, stepping with a supplied dynamics function --
the oracle's (orc_env_step_host, CPU) or the library's test hook (crux_env_step_host, the rollout kernel's own device functions) -- and drawing initial states exactly as
the device samplers do (csrc/env.hip: env_draw_initial = -0.05 + 0.1 u, u = the four Float64 uniforms of crux_philox(seed, 2 n_resets (+1), env, CRUX_RNG_RESET))."""
import ctypes as C

import numpy as np

import oracle as O
from crux_jl_amd import _lib as L

RNG_RESET = 3


def _f64(hi, lo):
    return float(((int(hi) << 32 | int(lo)) >> 11) * 1.1102230246251565e-16)


def cartpole_initialstate(seed):
    def f(e, n_resets):
        a, b = np.zeros(4, np.uint32), np.zeros(4, np.uint32)
        O.lib().orc_philox(seed, 2 * n_resets, e, RNG_RESET, O.vpz(a)); O.lib().orc_philox(seed, 2 * n_resets + 1, e, RNG_RESET, O.vpz(b))
        u = np.array([_f64(a[0], a[1]), _f64(a[2], a[3]), _f64(b[0], b[1]), _f64(b[2], b[3])], np.float64)
        return np.float64(-0.05) + np.float64(0.1) * u
    return f


def cartpole_gen(step_fn):
    """step_fn(state[4 x 1] f64, action one-hot bytes [2 x 1]) -> (next_state [4], r)"""
    def gen(e, s, a, n_steps):
        onehot = np.zeros((2, 1), np.uint8, order="F"); onehot[int(a), 0] = 1
        sn, r = step_fn(np.asfortranarray(np.asarray(s, np.float64).reshape(4, 1)), onehot)
        return sn, r, {"cost": 1.0 if abs(sn[2]) > 0.05 else 0.0}
    return gen


def oracle_step(state, onehot):
    sn = np.empty((4, 1), np.float64, order="F"); obs = np.empty((4, 1), np.float32, order="F"); r = np.empty(1, np.float32); d = np.empty(1, np.uint8)
    O.chk(O.lib().orc_env_step_host(L.ENV["cartpole"], 1, O.vpz(state), O.vpz(onehot), None, O.vpz(sn), O.vpz(obs), O.vpz(r), O.vpz(d)))
    return sn[:, 0].copy(), float(r[0])


def device_step(ctx):
    def f(state, onehot):
        sn = np.empty((4, 1), np.float64, order="F"); obs = np.empty((4, 1), np.float32, order="F"); r = np.empty(1, np.float32); d = np.empty(1, np.uint8)
        ctx.check(ctx.lib.crux_env_step_host(ctx.h, L.ENV["cartpole"], 1, O.vpz(state), O.vpz(onehot), None, O.vpz(sn), O.vpz(obs), O.vpz(r), O.vpz(d)))
        return sn[:, 0].copy(), float(r[0])
    return f


def cartpole_isterminal(s):
    # CartPole-v1 termination (the restated dynamics: |x| > 2.4 or |theta| > 12 degrees)
    return bool(abs(s[0]) > 2.4 or abs(s[2]) > 12 * 2 * np.pi / 360)


def cartpole_observation(s):
    return np.asarray(s, np.float64).astype(np.float32)
