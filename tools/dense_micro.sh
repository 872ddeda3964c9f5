#!/bin/bash
# per-kernel durations of the stand-alone dense-engine launches (tools/dense_micro.py) from rocprofv3 --kernel-trace --stats -> gpurun_out/r04/dense_micro.txt
R=$PWD; OUT=$R/gpurun_out/r04; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
: > $OUT/dense_micro.txt
for cfg in "128 8" "256 4"; do
  for f in 1 0; do
    rm -rf /tmp/dm; CRUX_DENSE_FUSED=$f timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dm -o t -- python $R/tools/dense_micro.py $cfg > /tmp/dm.log 2>&1
    echo "== B, in = $cfg  CRUX_DENSE_FUSED=$f : $(grep 'us per forward' /tmp/dm.log)" >> $OUT/dense_micro.txt
    python - >> $OUT/dense_micro.txt <<'PY'
import csv
for r in list(csv.DictReader(open("/tmp/dm/t_kernel_stats.csv")))[:9]:
    print("  %-70s calls %6s avg %9.1f ns min %8s max %8s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"]))
PY
  done
done
cat $OUT/dense_micro.txt
