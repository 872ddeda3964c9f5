#!/bin/bash
# HBM-side traffic of the learner kernels: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (MI355X_MICROARCH.md, HBM section).
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/pmc_traffic; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --replicas 0 --replicas-wide 0 --no-extra --no-measure-traffic --no-early-stop $BENCH_ARGS"   # BENCH_ARGS="--workload c5": the C5 shard
for c in FETCH_SIZE WRITE_SIZE; do timeout 400 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -o p -- $CMD > $OUT/$c.log 2>&1; done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_train" in k or "k_rollout" in k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for n, v in sorted(d.items()): print("   %-12s per launch: %s  (rocprofv3 units as reported; FETCH_SIZE is KB-granular, x2 correction for wide reads per the guide)" % (n, ["%.1f" % x for x in v]))
PY
