#!/bin/bash
# Collect SQ counters for the learner kernels (run on the GPU box: gpurun -- 'bash tools/pmc_learner.sh').
# Separate --pmc passes, no trace domains (see MI355X_MICROARCH.md, rocprofv3 section). Output: gpurun_out/pmc/<pass>/...
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --replicas 0 --replicas-wide 0 --no-extra --no-measure-traffic --no-early-stop $BENCH_ARGS"   # BENCH_ARGS="--workload c5": the C5 shard
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM" ; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o p -- $CMD > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_train" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in agg.items():
    print(k)
    for n, v in sorted(d.items()): print(f"   {n:32s} {v:16.0f}")
PY
