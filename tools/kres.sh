#!/bin/bash
# usage: tools/kres.sh <lib.so> [pattern]: VGPRs, spills, LDS and scratch of the kernels in a built library
set -e
T=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin "$1" $T/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/dev.co | grep -E "^\s+\.name:|\.vgpr_count|\.vgpr_spill_count|\.group_segment_fixed_size|\.private_segment_fixed_size" \
  | paste - - - - - | sed 's/ \+/ /g; s/_ZN5bpgpu//' | grep -E "${2:-.}"
rm -rf $T
