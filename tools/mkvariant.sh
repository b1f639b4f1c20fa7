#!/bin/bash
# A variant build of libkvhip.so for an A/B on the box: tools/mkvariant.sh <name> [-DKNOB=value ...]
# -> build/ab/<name>.so (run it with `python tools/ab_bench.py build/ab/<name>.so [bench args]`).
# Every translation unit is compiled with the extra flags (objects under build/ab/obj_<name>/, not shipped).
set -e
name=$1; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
src=$root/tfplus_amd/csrc
obj=$root/build/ab/obj_$name
mkdir -p $obj
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -w $*"
pids=()
for u in ${UNITS:-kvhip kv_apply_a kv_apply_b kv_papply_a kv_papply_b kv_papply_c}; do
  /opt/rocm/bin/hipcc $FLAGS -c -o $obj/$u.o $src/$u.hip &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
# units not rebuilt come from the product build
objs=""
for u in kvhip kv_apply_a kv_apply_b kv_papply_a kv_papply_b kv_papply_c; do
  if [ -f $obj/$u.o ]; then objs="$objs $obj/$u.o"; else objs="$objs $src/_obj/$u.o"; fi
done
/opt/rocm/bin/hipcc $FLAGS -shared -o $root/build/ab/$name.so $objs
echo build/ab/$name.so
