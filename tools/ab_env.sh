# usage: bash tools/ab_env.sh "ENV=1 ENV2=2" "..." : bench.py under each environment setting
B="--steps 20 --warmup 3 --no-cpu-baseline --no-extras"
i=0
for e in "$@"; do
  i=$((i+1))
  env $e python bench.py $B > gpurun_out/abe_$i.json 2> gpurun_out/abe_$i.err
  python3 - <<PY
import json
try:
  d=json.loads(open('gpurun_out/abe_$i.json').read().strip().splitlines()[-1]); k=d['kernels_ms']
  print('%-28s step %.4f ' % ('$e', d['ms_per_step']), {a:round(v*1e3,1) for a,v in k.items() if v})
except Exception as ex: print('$e', 'failed', ex)
PY
done
