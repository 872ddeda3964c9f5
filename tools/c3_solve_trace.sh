#!/bin/bash
# Timeline of whole solve(DQN + PER) iterations at the C3 shapes (1 M-row ring): rocprofv3 kernel trace of tools/c3_solve_bench.py; prints, for two iterations from the middle of the
# run, every launch with its duration and the idle gap before it. Output: gpurun_out/r05/c3_solve_trace.txt
R=$PWD; OUT=$R/gpurun_out/${ROUND:-r05}; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/c3s
C3S_BUF=1000000 C3S_ITERS=300 C3S_PREFILL=1 timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/c3s -o t -- python $R/tools/c3_solve_bench.py > /tmp/c3s.log 2>&1
python - > $OUT/c3_solve_trace.txt <<'PY'
import csv
rows = list(csv.DictReader(open("/tmp/c3s/t_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_rollout" in r["Kernel_Name"]]
a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 2]
prev = None; print("two solve iterations from the middle of the run: kernel, duration us, gap before us")
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-44s %8.2f %7.2f" % (r["Kernel_Name"].split("(")[0][:44], (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0)); prev = e
print("span of the two iterations: %.1f us" % ((int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3))
print(open("/tmp/c3s.log").read().strip().splitlines()[-1])
PY
cat $OUT/c3_solve_trace.txt
