#!/bin/bash
# Register / LDS / spill figures of the kernels of one translation unit (cross-compiled, no GPU needed):
#   tools/kres.sh kvhip.hip 'k_ltile|k_part2'
set -e
cd "$(dirname "$0")/../tfplus_amd/csrc"
out=/tmp/kres_$$; mkdir -p $out
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics --offload-device-only -c -o $out/u.co "$1"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --input=$out/u.co --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$out/u.elf --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $out/u.elf | grep -E "^\s+\.name:|\.vgpr_count|\.sgpr_count|vgpr_spill_count|group_segment_fixed|private_segment_fixed" | paste - - - - - - \
  | sed 's/ \+/ /g; s/\.group_segment_fixed_size/lds/; s/\.private_segment_fixed_size/scratch/; s/\.sgpr_count/sgpr/; s/\.vgpr_count/vgpr/; s/\.vgpr_spill_count/spill/' \
  | grep -E "${2:-.}" | while read -r l; do n=$(echo "$l" | sed 's/.*\.name: \([^ \t]*\).*/\1/'); echo "$(echo "$n" | /opt/rocm/lib/llvm/bin/llvm-cxxfilt | cut -c1-110)  $(echo "$l" | sed 's/\.name: [^ \t]*//')"; done
rm -rf $out
