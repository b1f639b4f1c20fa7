# usage: bash tools/ab_small.sh variant : sharded world-1 phases and small / mid batches at low skew, product build against build/ab/<variant>.so
for so in "" build/ab/$1.so; do
  echo "== ${so:-product}"
  run() { if [ -n "$so" ]; then python tools/ab_bench.py $so "$@"; else python bench.py "$@"; fi; }
  run --force-sharded --no-cpu-baseline --no-extras --steps 30 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('sharded', round(d['ms_per_step'],4), {k:(round(v*1e3,1) if v is not None else None) for k,v in d['phases_ms'].items()})"
  for b in 100000 200000; do
    run --batch $b --zipf 0.3 --keys 20000000 --no-cpu-baseline --no-extras --steps 40 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $b zipf 0.3', round(d['ms_per_step'],4), {k:round(v*1e3,1) for k,v in d['kernels_ms'].items() if v})"
  done
done
