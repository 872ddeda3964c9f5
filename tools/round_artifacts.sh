#!/bin/bash
# The round's judged artifacts in one GPU call: the default bench line, rocprofv3 --kernel-trace --stats of the same command, the PMC passes (matrix-pipe / issue counters and
# HBM traffic, separate --pmc runs as the guide prescribes), the population counters, the smoke run. Output under gpurun_out/$ROUND; copy what is judged into profiles/.
ROUND=${ROUND:-r06}; R=$PWD; OUT=$R/gpurun_out/$ROUND; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?" >> $OUT/smoke.txt
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o k -- python $R/bench.py --no-cpu-baseline --no-extra > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
cd $R
[ -x tools/pmc_learner.sh ] && ROUND=$ROUND bash tools/pmc_learner.sh > $OUT/pmc_learner_sq.txt 2>&1
[ -x tools/pmc_traffic.sh ] && ROUND=$ROUND bash tools/pmc_traffic.sh > $OUT/pmc_traffic.txt 2>&1
bash tools/pmc_population.sh 128 > $OUT/pmc_population.txt 2>&1
