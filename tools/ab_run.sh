# usage: bash tools/ab_run.sh "<env>" variant ... : bench.py with the product build, then with each build/ab/<variant>.so, under <env>
B="--steps 20 --warmup 3 --no-cpu-baseline --no-extras"
E="$1"; shift
run() { tag=$1; shift; env $E "$@" > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err; python3 - <<PY
import json
try:
  d=json.loads(open('gpurun_out/ab_$tag.json').read().strip().splitlines()[-1]); k=d['kernels_ms']
  print('%-14s step %.4f ' % ('$tag', d['ms_per_step']), {a:round(v*1e3,1) for a,v in k.items() if v})
except Exception as e: print('$tag', 'failed', e)
PY
}
run base python bench.py $B
for v in "$@"; do run $v python tools/ab_bench.py build/ab/$v.so $B; done
