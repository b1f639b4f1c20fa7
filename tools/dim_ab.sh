#!/bin/bash
# entry-list pipeline against the sorted-position one (KV_NO_FUSED=1) per embedding dim, configs[1] batch shape
cd ${GRAFT_REPO_ROOT:-/root/repo}
for d in ${@:-8 16 32 64 128}; do
  for nf in 0 1; do
    echo -n "dim $d KV_NO_FUSED=$nf  "
    KV_NO_FUSED=$nf timeout -k 10 200 python bench.py --dim $d --keys 30000000 --no-extras --cpu-steps 0 --steps 30 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), {k:round(v*1e3,1) for k,v in d['kernels_ms'].items() if v>0})"
  done
done
