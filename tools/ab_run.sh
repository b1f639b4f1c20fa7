B="--steps 20 --warmup 3 --no-cpu-baseline --no-extras"
run() { tag=$1; shift; "$@" > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err; python3 - <<PY
import json
try:
  d=json.loads(open('gpurun_out/ab_$tag.json').read().strip().splitlines()[-1]); k=d['kernels_ms']
  print('%-14s step %.4f  ltile %.1f part %.1f tsum %.1f apply %.1f' % ('$tag', d['ms_per_step'], k['lookup_tile']*1e3, k['lookup_part']*1e3, k['apply_tsum']*1e3, k['apply_sorted']*1e3))
except Exception as e: print('$tag', 'failed', e)
PY
}
run base python bench.py $B
for v in "$@"; do
  run $v python tools/ab_bench.py build/ab/$v.so $B
  KV_FORCE_P=1024 run ${v}_p1024 python tools/ab_bench.py build/ab/$v.so $B
done
