#!/bin/bash
# same-box A/B of two builds of the library on the replica-group path: two PROCESSES on one GPU (hipIpc peer slots), per-step exchange (k = 1) and the periodic form (k = 8),
# C2 and the C5 shard; crux.jl_amd/libcruxhip_base.so (a saved copy of an earlier build) against the current one; prints env-steps/s and us per actor step
for v in base new; do
  if [ $v = base ]; then export CRUXHIP_LIB=$PWD/crux.jl_amd/libcruxhip_base.so; else unset CRUXHIP_LIB; fi
  for wl in c2 c5; do for k in 1 8; do
    extra=""; [ $wl = c5 ] && extra="--workload c5"; [ $k = 8 ] && extra="$extra --sync-every 8"
    python bench.py --gpus 2 --same-device $extra --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=$v $wl k=$k', round(d['value']), round(d['roofline']['us_per_grad_step'],3), d.get('exchange_kind'))"
  done; done
done
