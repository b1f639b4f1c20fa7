cd /tmp && export TMPDIR=/tmp
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/prof_r04_sharded
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $repo/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --force-sharded > $out/bench.json 2> $out/trace.err
python3 - <<PY
import csv,glob,collections
f=glob.glob("$out/trace/*/*_kernel_trace.csv")[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"]].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
rows=[]
for k,v in d.items():
    v=v[-20:]
    rows.append((sum(v)/len(v)/1e3,len(d[k]),k[:110]))
for a,c,k in sorted(rows,reverse=True)[:24]: print("%8.1f us  x%-5d %s"%(a,c,k))
PY
