#!/bin/bash
# Phase durations of the chained off-policy epochs (C3: tools/c3_trace.py, C4: tools/c4_trace.py) from a rocprofv3 kernel trace:
# per launch its grid, duration and the idle gap before it, for one epoch in the middle of the run, plus the kernel-stats summary.
# Output: gpurun_out/$ROUND/offpolicy_phase_trace.txt (copied to profiles/<round>_offpolicy_phase_trace.txt)
R=$PWD; OUT=$R/gpurun_out/${ROUND:-r05}; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
: > $OUT/offpolicy_phase_trace.txt
for w in c3 c4; do
  rm -rf /tmp/pt_$w
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt_$w -o t -- python $R/tools/${w}_trace.py > /tmp/pt_$w.log 2>&1
  python - $w >> $OUT/offpolicy_phase_trace.txt <<'PY'
import csv, sys
w = sys.argv[1]
rows = [r for r in csv.DictReader(open("/tmp/pt_%s/t_kernel_trace.csv" % w)) if "k_phase" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = {"c3": int(__import__("os").environ.get("C3_N", 7)), "c4": int(__import__("os").environ.get("C4_N", 26))}[w]
mid = rows[len(rows) // 2: len(rows) // 2 + 2 * n]
print("== %s: %d consecutive phase launches from the middle of the run (kernel, grid threads, duration us, gap before us)" % (w, len(mid)))
prev = None; tot = 0.0
for r in mid:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-22s %8s %7.2f %6.2f" % (r["Kernel_Name"][:22], r.get("Grid_Size_X", r.get("Grid_Size")), (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0)); tot += (e - s) / 1e3
    prev = e
print("sum of durations: %.1f us over %d launches = %.2f us per launch" % (tot, len(mid), tot / len(mid)))
print("-- kernel stats (rocprofv3 --stats), top rows")
for r in list(csv.DictReader(open("/tmp/pt_%s/t_kernel_stats.csv" % w)))[:8]:
    print("%-60s calls %6s avg %9.1f ns  %5s %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
print(open("/tmp/pt_%s.log" % w).read().strip().splitlines()[0] if True else "")
PY
done
cat $OUT/offpolicy_phase_trace.txt | tail -5
