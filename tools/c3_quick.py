import sys, json
sys.path.insert(0, "/root/repo")
import crux_jl_amd as crux, bench_offpolicy as b
ctx = crux.default_context()
print(json.dumps({k: v for k, v in b.c3(crux, ctx, cpu=False).items() if k in ("us_per_epoch", "grad_steps_per_s")}))
