#!/bin/bash
# the numbers DESIGN.md quotes from the diagnostic scripts, collected in one file (-> profiles/r03_tools_output.txt)
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/tools_r03.txt
: > $out
run() { echo "### $*" >> $out; timeout -k 5 300 "$@" >> $out 2>&1; echo "(rc=$?)" >> $out; echo >> $out; }
run python tools/opt_bench.py
run python tools/goz_dims.py
run python tools/inference_gather.py
run python tools/small_batch.py 2048 1
run python tools/small_batch.py 2048 1 multi
run python tools/config5_bench.py
run python tools/config5_bench.py --per-table
run python tools/sparse_lookup.py
run python bench.py --keys 125000000 --dim 64 --steps 20 --warmup 3 --no-cpu-baseline --no-extras
run python bench.py --deterministic --steps 20 --warmup 3 --no-cpu-baseline --no-extras
run python bench.py --force-sharded --steps 20 --warmup 3 --no-cpu-baseline --no-extras
run env KV_NO_FUSED=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras
run env KV_NO_DEFER_PART=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras
run python bench.py --overlap --steps 20 --warmup 3 --no-cpu-baseline --no-extras
run python bench.py --graph --steps 20 --warmup 3 --no-cpu-baseline --no-extras
grep -v "amdgpu.ids" $out > $out.tmp && mv $out.tmp $out
tail -3 $out
