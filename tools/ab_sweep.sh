# usage: bash tools/ab_sweep.sh variant ... : bench.py's skew sweep / op-boundary records with the product build, then with each build/ab/<variant>.so
B="--steps 20 --warmup 3 --no-cpu-baseline"
run() { tag=$1; shift; "$@" > gpurun_out/abs_$tag.json 2> gpurun_out/abs_$tag.err; python3 - <<PY
import json
try:
  d=json.loads(open('gpurun_out/abs_$tag.json').read().strip().splitlines()[-1])
  print('%-10s step %.4f' % ('$tag', d['ms_per_step']), 'no_token %.4f' % d['no_token']['ms_per_step'], ' sweep:', [(e['zipf'], round(e['lookup_ms'],4), round(e['lookup_rows_ready_ms'],4), round(e['ms_per_step'],4)) for e in d['skew_sweep']])
except Exception as e: print('$tag', 'failed', e)
PY
}
run base python bench.py $B
for v in "$@"; do run $v python tools/ab_bench.py build/ab/$v.so $B; done
