#!/bin/bash
# N consecutive runs of the whole GPU suite exactly as the driver runs it (`python -m pytest tests/ -x -q -m gpu`): one line per run with pytest's summary and the wall time
mkdir -p gpurun_out/r06; out=gpurun_out/r06/full_suite_soak.txt; : > $out
for i in $(seq 1 ${1:-5}); do s=$(date +%s); r=$(timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -1); echo "run $i: $r [rc ${PIPESTATUS[0]}; $(( $(date +%s) - s )) s wall]" >> $out; done
cat $out
