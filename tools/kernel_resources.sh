#!/bin/bash
# Register / scratch / spill figures of every kernel in an object file built by the Makefile (default: swmi_kernels.o).
# usage: tools/kernel_resources.sh [sparksmithwaterman_amd/lib/obj/swmi_tfused.o]
set -e
OBJ=${1:-$(dirname "$0")/../sparksmithwaterman_amd/lib/obj/swmi_kernels.o}
T=$(mktemp -d)
LLVM=/opt/rocm/lib/llvm/bin
$LLVM/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin "$OBJ"
$LLVM/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/k.co --unbundle
echo "kernel scratch_bytes sgprs sgpr_spills vgprs vgpr_spills"
$LLVM/llvm-readelf --notes $T/k.co | grep -E "\.name:|\.vgpr_count|\.sgpr_count|private_segment_fixed_size|spill_count" | paste - - - - - - | awk '{print $2, $4, $6, $8, $10, $12}'
rm -rf $T
