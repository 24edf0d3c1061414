#!/bin/bash
# Register / spill counts of the kernels of one csrc file (hipcc -save-temps in a scratch directory):  tools/kernel_resources.sh rnn_fused2 [name filter]
set -u
src=${1:?csrc file stem}; filt=${2:-.}
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/kres.XXXXXX)
( cd "$tmp" && hipcc --offload-arch=gfx950 -O3 -std=c++17 -I"$root/include" -I"$root/icassp2022-depression_amd/csrc" \
    -c "$root/icassp2022-depression_amd/csrc/$src.hip" -save-temps -o k.o 2>&1 | grep -v warning | grep -i "error" )
grep -E "^\s+\.(vgpr_count|vgpr_spill_count|sgpr_spill_count)|\.name:" "$tmp"/*gfx950.s | awk '{print $2, $3}' | paste - - - - | grep -E "$filt"
echo "ISA: $tmp/$src-hip-amdgcn-amd-amdhsa-gfx950.s"
