#!/bin/bash
# same-box A/B of two builds of the library: crux.jl_amd/libcruxhip_base.so (a saved copy of an earlier build) against the current one; prints env-steps/s and us per actor step
for v in base new base new; do
  if [ $v = base ]; then export CRUXHIP_LIB=$PWD/crux.jl_amd/libcruxhip_base.so; else unset CRUXHIP_LIB; fi
  echo "lib=$v"; bash tools/headline_quick.sh
done
