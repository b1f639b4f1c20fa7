#!/bin/bash
# Register / LDS / spill figures of the kernels of one translation unit (cross-compiled, no GPU needed):
#   tools/kres.sh kvhip.hip 'k_ltile|k_part2'
set -e
cd "$(dirname "$0")/../tfplus_amd/csrc"
out=/tmp/kres_$$; mkdir -p $out
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -w $KRES_FLAGS --offload-device-only -c -o $out/u.co "$1"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --input=$out/u.co --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$out/u.elf --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $out/u.elf | grep -E "^\s+\.name:|\.vgpr_count|\.sgpr_count|vgpr_spill_count|group_segment_fixed|private_segment_fixed" | paste - - - - - - \
  | python3 -c '
import re, subprocess, sys
pat = sys.argv[1] if len(sys.argv) > 1 else "."
for l in sys.stdin:
  f = dict(re.findall(r"\.(\w+):\s+(\S+)", l))
  name = subprocess.run(["c++filt", f["name"]], capture_output=True, text=True).stdout.strip()
  name = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
  if re.search(pat, name):
    print("%-64s vgpr %3s sgpr %3s lds %6s scratch %s spill %s" % (name[:64], f["vgpr_count"], f["sgpr_count"], f["group_segment_fixed_size"], f["private_segment_fixed_size"], f["vgpr_spill_count"]))
' "${2:-.}"
rm -rf $out
