#!/bin/bash
# Upper bound of what a four-waves-per-tile decomposition of k_train_fs2 could gain, measured before building it: same-box A/B of the shipped library
# against two TIMING-ONLY builds of train_fs2.o (-DCRUX_FS2_EXP=1|2; wrong results): 1 = every compute wave runs the shortened MFMA chain of a quad form (half the L2, dH1, dW1
# MFMAs), nothing else changes; 2 = additionally the helper wave on the same SIMD issues the MFMAs / VALU of the second pair a real quad form would put there.
# build: see the commands in profiles/r06_fs4_bound.txt
for v in base exp1 exp2 exp3 base exp1 exp2 exp3; do
  if [ $v = base ]; then unset CRUXHIP_LIB; else export CRUXHIP_LIB=$PWD/crux.jl_amd/libcruxhip_$v.so; fi
  echo "lib=$v"; bash tools/headline_quick.sh
done
