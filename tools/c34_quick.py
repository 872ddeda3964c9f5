import sys, json
sys.path.insert(0, "/root/repo")
import crux_jl_amd as crux, bench_offpolicy as b
ctx = crux.default_context()
for name, f in (("c3", b.c3), ("c4", b.c4)):
    try:
        d = f(crux, ctx, cpu=False)
        print(name, json.dumps({k: v for k, v in d.items() if "us_per_epoch" in k or k in ("grad_steps_per_s",)}))
    except Exception as e:
        print(name, "ERR", repr(e)[:300])
