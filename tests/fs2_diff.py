"""development aid (test infrastructure: it builds its shards with the oracle, like the tests): k_train_fs2 against k_train_fs (CRUX_FS2=0) after a few minibatch steps, difference per parameter block (W1 b1 W2 b2 W3 b3 extra)"""
import os, sys, subprocess, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import test_gpu_fs2 as T
    from parity import crux
    import parity
    family, nb = sys.argv[2], int(sys.argv[3])
    data = T._shard(family, 900, 8, 128)
    le = 0.1 if parity.FAMILIES[family][2] else 0.0
    P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": le}
    a, c = T._nets(family); out = {}
    for name, net, loss in (("actor", a, crux.ppo_loss), ("critic", c, crux.value_mse_loss)):
        b = T._buffer(family, data)
        crux.batch_train_(net, crux.TrainingParams(loss=loss, batch_size=128, epochs=3, name="n_", shuffle_seed=17, max_batches=nb), P, b)
        out[name] = [x.tolist() for x in T._state(net)[:3]]
    print("RESULT" + json.dumps(out))
else:
    family = sys.argv[1] if len(sys.argv) > 1 else "cartpole"; nb = sys.argv[2] if len(sys.argv) > 2 else "1"
    res = {}
    for form in ("1", "0"):
        env = dict(os.environ, CRUX_FS2=form)
        o = subprocess.run([sys.executable, __file__, "child", family, nb], env=env, capture_output=True, text=True)
        line = [l for l in o.stdout.splitlines() if l.startswith("RESULT")]
        if not line: print(o.stdout[-2000:], o.stderr[-3000:]); sys.exit(1)
        res[form] = json.loads(line[0][6:])
    import parity
    od, ad, disc, adims, cdims = parity.FAMILIES[family][:5]
    for name, dims in (("actor", adims), ("critic", cdims)):
        th2, m2, v2 = [np.array(x, np.float32) for x in res["1"][name]]; th1, m1, v1 = [np.array(x, np.float32) for x in res["0"][name]]
        o = 0
        for l in range(len(dims) - 1):
            for nm, n in (("W%d" % (l + 1), dims[l] * dims[l + 1]), ("b%d" % (l + 1), dims[l + 1])):
                d = np.abs(th2[o:o + n] - th1[o:o + n]); dm = np.abs(m2[o:o + n] - m1[o:o + n])
                print(name, nm, "max|dtheta| %.3g (%d of %d differ)  max|dm| %.3g" % (d.max(), int((d > 0).sum()), n, dm.max()))
                if nm == "W2" and d.max() > 0:
                    bad = np.nonzero(d.reshape(dims[l], dims[l + 1]).T > 0)   # [o][i]
                    M2 = m2[o:o + n].reshape(dims[l], dims[l + 1]).T; M1 = m1[o:o + n].reshape(dims[l], dims[l + 1]).T   # [o][i]
                    for oo in (0, 5, 17, 33, 50):
                        print("   m[o=%d][i=0,1,16,17,40]: fs2" % oo, M2[oo, [0, 1, 16, 17, 40]], " fs", M1[oo, [0, 1, 16, 17, 40]])
                    print("   W2 rows(o) differing:", sorted(set(bad[0].tolist()))[:70], " cols(i):", sorted(set(bad[1].tolist()))[:70])
                o += n
        if o < th2.size: print(name, "extra", np.abs(th2[o:] - th1[o:]).max())
