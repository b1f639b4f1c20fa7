#!/bin/bash
# k_tsum / k_papply with fewer CUs (ROC_GLOBAL_CU_MASK): what a CU-masked side stream would leave the bandwidth kernels -> profiles/r06_cu_mask.txt
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
F=$(python3 -c "print('0x' + 'f'*64)")
E=$(python3 -c "print('0x' + 'e'*64)")      # 3 of every 4 CUs: 192
A=$(python3 -c "print('0x' + 'a'*64)")      # every other CU: 128
L192=$(python3 -c "print('0x' + 'f'*48)")   # the first 192
L128=$(python3 -c "print('0x' + 'f'*32)")   # the first 128
for name in all E A L192 L128; do
  case $name in all) unset ROC_GLOBAL_CU_MASK;; E) export ROC_GLOBAL_CU_MASK=$E;; A) export ROC_GLOBAL_CU_MASK=$A;; L192) export ROC_GLOBAL_CU_MASK=$L192;; L128) export ROC_GLOBAL_CU_MASK=$L128;; esac
  rm -rf gpurun_out/r6w_$name
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6w_$name -o p -- python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r6w_$name.json 2> gpurun_out/r6w_$name.err
  echo "== $name rc=$? mask=${ROC_GLOBAL_CU_MASK:-none}"
  python3 - $name <<'PY'
import csv, glob, json, sys
name = sys.argv[1]
try:
  d = json.loads([l for l in open("gpurun_out/r6w_%s.json" % name) if l.startswith("{")][-1])
  print("  ms_per_step %.4f" % d["ms_per_step"])
except Exception as e:
  print("  no bench line:", e)
f = glob.glob("gpurun_out/r6w_%s/**/*kernel_trace.csv" % name, recursive=True)
if f:
  rows = list(csv.DictReader(open(f[0])))
  by = {}
  for r in rows:
    by.setdefault(r["Kernel_Name"], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
  for k, v in by.items():
    if any(t in k for t in ("k_tsum", "k_papply<0, 4, 8, 1, 512>", "k_ltile<long long, 8, true, false, false>")) and len(v) >= 20:
      last = v[-20:]
      print("  %-60s %6.1f us (last 20 of %d)" % (k[k.find("k_"):][:60], sum(last) / len(last) / 1e3, len(v)))
PY
done
