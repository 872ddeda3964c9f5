#!/bin/bash
# Prints VGPR / AGPR / SGPR / scratch (private segment) / LDS of the kernels in crux.jl_amd/csrc/*.o (reads the code-object notes; no GPU needed).
# usage: tools/kernel_resources.sh [name-regex] [object ...]
BIN=/opt/rocm/lib/llvm/bin
DIR=$(dirname "$0")/../crux.jl_amd/csrc
F="${1:-.}"; shift
OBJS=("$@"); [ ${#OBJS[@]} -eq 0 ] && OBJS=($DIR/*.o)
TMP=$(mktemp -d)
for o in "${OBJS[@]}"; do
  $BIN/llvm-objcopy --dump-section .hip_fatbin=$TMP/fat.bin "$o" 2>/dev/null || continue
  $BIN/clang-offload-bundler --unbundle --type=o --input=$TMP/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$TMP/a.co 2>/dev/null || continue
  $BIN/llvm-readelf --notes $TMP/a.co | awk -v f="$F" -v o="$(basename $o)" '
    /\.agpr_count:/ {a=$2} /\.group_segment_fixed_size:/ {l=$2} /\.name:/ {n=$2} /\.private_segment_fixed_size:/ {p=$2} /\.sgpr_count:/ {s=$2}
    /\.vgpr_count:/ {v=$2} /\.vgpr_spill_count:/ {sp=$2; if (n ~ f) printf "%s vgpr %3d agpr %3d sgpr %3d scratch %5d spill %3d lds %6d  [%s]\n", n, v, a, s, p, sp, l, o}'
done | c++filt
rm -rf $TMP
