#!/bin/bash
# the numbers DESIGN.md quotes from the diagnostic scripts and A/B switches, collected in one file (-> profiles/r04_tools_output.txt)
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/tools_r04.txt
: > $out
B="--steps 20 --warmup 3 --no-cpu-baseline --no-extras"
short() {   # the bench line cut down to what the A/B is about
python3 -c '
import json, sys
for l in sys.stdin:
  if l.startswith("{"):
    d = json.loads(l)
    k = {a: round(v * 1e3, 1) for a, v in d.get("kernels_ms", {}).items() if v}
    print("ms_per_step %.4f  kernels_us %s" % (d["ms_per_step"], k))
  elif "amdgpu.ids" not in l: sys.stdout.write(l)'
}
run() { echo "### $*" >> $out; timeout -k 5 400 "$@" >> $out 2>&1; echo "(rc=$?)" >> $out; echo >> $out; }
runb() { echo "### $*" >> $out; timeout -k 5 400 "$@" 2>&1 | short >> $out; echo >> $out; }
runb python bench.py $B
runb env KV_NO_PAPPLY=1 python bench.py $B
runb env KV_BUCKET=1 python bench.py $B
runb env KV_DEFER_TILE=1 python bench.py $B
runb env KV_NO_FUSED=1 python bench.py $B
runb env KV_FORCE_P=512 python bench.py $B
runb env KV_FORCE_P=2048 python bench.py $B
runb python bench.py --no-token $B
runb python bench.py --deterministic $B
runb python bench.py --force-sharded $B
runb python bench.py --force-sharded --lossless $B
runb python bench.py --keys 125000000 --dim 64 $B
run python tools/opt_bench.py
run python tools/inference_gather.py
run python tools/small_batch.py 2048 1
run python tools/small_batch.py 2048 1 multi
run python tools/small_batch.py 2048 1 multi notoken
run python tools/config5_bench.py
run python tools/config5_bench.py --per-table
run python tools/config5_bench.py --sharded
run env KV_MULTI_SHARD_PER_TABLE=1 python tools/config5_bench.py --sharded
runb env KV_SHARD_SELF_COPY=1 python bench.py --force-sharded $B
run python tools/sparse_lookup.py
grep -v "amdgpu.ids" $out > $out.tmp && mv $out.tmp $out
tail -3 $out
