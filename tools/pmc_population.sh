#!/bin/bash
# Matrix-pipe occupancy of the batched (population) learner launches from hardware counters: rocprofv3 --pmc on tools/multi_breakdown.py S.
# Counter collection serialises dispatches, so each launch (S actors, then S critics) is measured alone. Output: one table on stdout.
S=${1:-128}
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/pmc_pop; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAVES" "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o p -- python $R/tools/multi_breakdown.py $S > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_train" not in k: continue
        per[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for (k, d), c in per.items():
        for n, v in c.items(): agg[k][n].append(v)
print("S = $S learners per launch; per-launch counter totals (mean over the launches of the run)")
for k, d in agg.items():
    m = {n: sum(v) / len(v) for n, v in d.items()}
    print(k)
    for n, v in sorted(m.items()): print("   %-28s %18.0f" % (n, v))
    if "SQ_WAVE_CYCLES" in m and "SQ_WAVES" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m and m["SQ_WAVES"] > 0:
        T = 4.0 * m["SQ_WAVE_CYCLES"] / m["SQ_WAVES"]          # cycles one wave (= the launch) lasted; SQ_WAVE_CYCLES counts quad-cycles
        print("   launch duration %.0f cycles; matrix pipe busy %.1f %% of (1024 SIMDs x duration), %.1f %% of the SIMDs that hosted a wave" % (
            T, 100.0 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * T), 100.0 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["SQ_WAVES"] / (2 if ", 8, 1," in k else 1) * T)))
PY
rm -rf $OUT
