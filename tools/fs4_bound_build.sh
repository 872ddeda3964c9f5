#!/bin/bash
# builds crux.jl_amd/libcruxhip_exp{1,2,3}.so for tools/fs4_bound.sh: the shipped objects with train_fs2.o replaced by a -DCRUX_FS2_EXP=n timing build (WRONG results by design)
set -e
cd "$(dirname "$0")/../crux.jl_amd/csrc"
make -s -j8
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -mllvm -amdgpu-sched-strategy=iterative-ilp"
OBJS=$(ls *.o | grep -v "^train_fs2.o$" | tr '\n' ' ')
for e in 1 2 3; do
  /opt/rocm/bin/hipcc $F -DCRUX_FS2_EXP=$e -c train_fs2.hip -o /tmp/train_fs2_exp$e.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-z,defs -o ../libcruxhip_exp$e.so $OBJS /tmp/train_fs2_exp$e.o -ldl
done
