#!/bin/bash
# A round's evidence in one GPU call: gpurun -- 'ROUND=r05 bash tools/artifacts.sh [parts]'  (parts: tests bench rocprof phase solve traffic sq px; default all). Output: gpurun_out/$ROUND/
# (copy what is to be judged into profiles/ with the round's prefix)
ROUND=${ROUND:-r05}; R=$PWD; OUT=$R/gpurun_out/$ROUND; mkdir -p $OUT; export TMPDIR=/tmp
PARTS=${1:-"tests bench rocprof phase solve traffic sq px"}
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has tests; then timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|error" > $OUT/pytest_gpu.txt; cat $OUT/pytest_gpu.txt; fi
if has bench; then timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 600 $OUT/bench.json; echo; fi
if has rocprof; then
  (cd /tmp && rm -rf /tmp/kst && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o k -- python $R/bench.py --no-measure-traffic --no-early-stop --replicas 0 --replicas-wide 0 --no-extra --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
   cp /tmp/kst/k_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null || find /tmp/kst -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;)
  head -4 $OUT/kernel_stats.csv | cut -c1-200
fi
if has phase; then C3_N=10 C4_N=28 bash tools/phase_trace.sh > /dev/null 2>&1; grep -E "sum of durations|us per epoch|epoch" $OUT/offpolicy_phase_trace.txt | head; fi
if has solve; then bash tools/c3_solve_trace.sh > /dev/null 2>&1; tail -4 $OUT/c3_solve_trace.txt; fi
if has traffic; then
  bash tools/pmc_traffic.sh > $OUT/pmc_traffic.txt 2>&1; BENCH_ARGS="--workload c5" bash tools/pmc_traffic.sh > $OUT/pmc_traffic_c5.txt 2>&1; tail -n 4 $OUT/pmc_traffic.txt | cut -c1-200; tail -n 4 $OUT/pmc_traffic_c5.txt | cut -c1-200
fi
if has sq; then bash tools/pmc_learner.sh > $OUT/pmc_learner_sq.txt 2>&1; tail -30 $OUT/pmc_learner_sq.txt | cut -c1-120; fi
if has px; then
  for k in 1 8; do timeout 600 python bench.py --gpus 2 --same-device --steps 3 --warmup 1 --no-cpu-baseline --replicas 0 --no-extra --sync-every $k $( [ $k = 1 ] && echo --selftest ) > $OUT/bench_2ranks_same_device_k$k.json 2> $OUT/bench_2ranks_same_device_k$k.err; done
  python - <<'PY'
import json, os
for k in (1, 8):
    try:
        d = json.loads(open('gpurun_out/%s/bench_2ranks_same_device_k%d.json' % (os.environ.get('ROUND', 'r05'), k)).read().strip().splitlines()[-1])
        print(k, d['value'], d['roofline']['us_per_grad_step'], d.get('replicas_bit_identical_after_run'), d['exchange']['flag_wait_per_rank'], (d.get('selftest') or {}).get('passed'))
    except Exception as e: print(k, 'ERR', e)
PY
fi
