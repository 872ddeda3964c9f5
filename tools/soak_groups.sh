#!/bin/bash
# soak of the same-device replica-group tests: N runs of tests/test_gpu_peer.py (in-process groups of 2, 3, 4) and M of tests/test_gpu_peer_xproc.py (two processes), one line per run
mkdir -p gpurun_out/r06; out=gpurun_out/r06/soak_groups.txt; : > $out
for i in $(seq 1 ${1:-20}); do s=$(date +%s%N); r=$(timeout 300 python -m pytest tests/test_gpu_peer.py -x -q 2>&1 | tail -1); echo "peer run $i: $r [$(( ($(date +%s%N) - s) / 1000000 )) ms wall]" >> $out; done
for i in $(seq 1 ${2:-5}); do s=$(date +%s%N); r=$(timeout 600 python -m pytest tests/test_gpu_peer_xproc.py -x -q 2>&1 | tail -1); echo "xproc run $i: $r [$(( ($(date +%s%N) - s) / 1000000 )) ms wall]" >> $out; done
cat $out
