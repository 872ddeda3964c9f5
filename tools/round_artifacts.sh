#!/bin/bash
# The round's judged artifacts in one GPU call: the default bench line, rocprofv3 --kernel-trace --stats of the same command, the PMC passes (matrix-pipe / issue counters and
# HBM traffic, separate --pmc runs as the guide prescribes), the population counters, the smoke run. Output under gpurun_out/$ROUND; copy what is judged into profiles/.
ROUND=${ROUND:-r06}; R=$PWD; OUT=$R/gpurun_out/$ROUND; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?" >> $OUT/smoke.txt
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o k -- python $R/bench.py --no-measure-traffic --no-early-stop --replicas 0 --replicas-wide 0 --no-extra --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
cd $R
[ -x tools/pmc_learner.sh ] && ROUND=$ROUND bash tools/pmc_learner.sh > $OUT/pmc_learner_sq.txt 2>&1
[ -x tools/pmc_traffic.sh ] && ROUND=$ROUND bash tools/pmc_traffic.sh > $OUT/pmc_traffic.txt 2>&1
bash tools/pmc_population.sh 128 > $OUT/pmc_population.txt 2>&1
cd $R
{ echo "== python examples/ppo_cartpole.py --iterations 12"; timeout 300 python examples/ppo_cartpole.py --iterations 12 2>&1 | tail -14
  echo "== python examples/dqn_gridworld.py"; timeout 300 python examples/dqn_gridworld.py 2>&1 | tail -3
  echo "== python examples/lagrange_ppo_cartpole.py"; timeout 300 python examples/lagrange_ppo_cartpole.py 2>&1 | tail -7
  echo "== python examples/sac_pendulum.py"; timeout 300 python examples/sac_pendulum.py 2>&1 | tail -3; } > $OUT/learning.txt 2>&1
timeout 1500 python -m pytest tests -q -m gpu -x -rs 2>&1 | tail -15 > $OUT/pytest_gpu.txt
ROUND=$ROUND bash tools/c3_solve_trace.sh > /dev/null 2>&1
for k in 1 8; do timeout 600 python bench.py --gpus 2 --same-device --steps 3 --warmup 1 --no-cpu-baseline --replicas 0 --no-extra --sync-every $k $( [ $k = 1 ] && echo --selftest ) > $OUT/bench_2ranks_same_device_k$k.json 2> $OUT/bench_2ranks_same_device_k$k.err; done
