#!/bin/bash
# stand-alone durations of the replay kernels (tools/per_bench.py: N = 1 M, B = 128) from rocprofv3 --kernel-trace --stats -> gpurun_out/r04/per_micro.txt
R=$PWD; OUT=$R/gpurun_out/r04; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
: > $OUT/per_micro.txt
for f in 1 0; do
  rm -rf /tmp/pm; CRUX_PER_FUSED_GATHER=$f timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm -o t -- python $R/tools/per_bench.py > /tmp/pm.log 2>&1
  echo "== CRUX_PER_FUSED_GATHER=$f : $(grep 'per step' /tmp/pm.log)" >> $OUT/per_micro.txt
  python - >> $OUT/per_micro.txt <<'PY'
import csv
for r in list(csv.DictReader(open("/tmp/pm/t_kernel_stats.csv")))[:9]:
    print("  %-60s calls %6s avg %9.1f ns min %8s max %8s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"]))
PY
done
cat $OUT/per_micro.txt
