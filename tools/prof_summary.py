"""Condenses a scripts_prof.sh output directory into the per-kernel table kept under profiles/."""
import collections, csv, glob, json, os, sys
out = sys.argv[1]
KEYS = {"k_ltile<": "lookup_tile  k_ltile", "k_part2": "lookup_part  k_part2 (a lookup's bookkeeping when no apply takes it over)",
        "k_tsum<": "apply_tsum   k_tsum", "k_papply<0": "apply_sorted k_papply<ADAM_V4>", "k_ltsum<": "apply_tile   k_ltsum",
        "k_uapply<0": "apply_unique k_uapply<ADAM_V4> (unique ids + pre-summed rows: the reference's op boundary)",
        "k_tile<": "old_tile     k_tile (sorted-position pipeline: dims outside the entry-list pipeline)", "k_part_keys<0>": "old_part     k_part_keys<LOOKUP>",
        "k_gather<8, true>": "lookup_order k_gather<8,ORDER>", "k_part_keys<6>": "old_index    k_part_keys<APPLYIDX>",
        "k_order": "apply_index  k_order",
        "k_apply<1, 0": "apply_sorted k_apply<APPLY,ADAM_V4>", "k_apply_fin<1, 0": "apply_span   k_apply_fin<APPLY,ADAM_V4>",
        "k_gather_or_zeros": "gather_or_zeros"}
def name_of(k):
  for a, b in KEYS.items():
    if a in k: return b
  return None
tr = glob.glob(os.path.join(out, "trace", "*", "*_kernel_trace.csv"))
dur = collections.defaultdict(list)
if tr:
  for r in csv.DictReader(open(tr[0])):
    n = name_of(r["Kernel_Name"])
    if n: dur[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("rocprofv3 --kernel-trace --stats  (bench.py --steps 20 --warmup 3; last 20 launches of each kernel)")
print("%-44s %8s %10s %10s %10s" % ("kernel", "calls", "avg_us", "min_us", "max_us"))
for n, v in dur.items():
  v = v[-20:]
  print("%-44s %8d %10.1f %10.1f %10.1f" % (n, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3))
pm = collections.defaultdict(dict)
for f in sorted(glob.glob(os.path.join(out, "pmc_*", "*", "*_counter_collection.csv"))):
  agg = collections.defaultdict(lambda: collections.defaultdict(list))
  for r in csv.DictReader(open(f)):
    n = name_of(r["Kernel_Name"])
    if n: agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
  for n, v in agg.items():
    for c, vals in v.items():
      pm[n][c] = sum(vals[-3:]) / len(vals[-3:])
print()
print("rocprofv3 --pmc (separate passes; mean of the last 3 launches).  FETCH_SIZE / WRITE_SIZE are KiB;")
print("hbm_read_MB doubles FETCH_SIZE (gfx950 counts 128-B requests as 64 B, MI355X_MICROARCH.md §HBM).")
for n, v in pm.items():
  fs, ws = v.get("FETCH_SIZE"), v.get("WRITE_SIZE")
  extra = ""
  if fs is not None and ws is not None:
    extra = "  hbm_read_MB %.1f  hbm_write_MB %.1f" % (2 * fs * 1024 / 1e6, ws * 1024 / 1e6)
  print("%-44s%s" % (n, extra))
  for c in sorted(v): print("      %-24s %16.1f" % (c, v[c]))
# machine-readable HBM traffic per launch for bench.py's roofline.traffic (bytes; reads = 2 x FETCH_SIZE)
traffic = {}
for n, v in pm.items():
  fs, ws = v.get("FETCH_SIZE"), v.get("WRITE_SIZE")
  if fs is not None and ws is not None:
    traffic[n.split()[0]] = {"hbm_bytes": 2 * fs * 1024 + ws * 1024, "read_bytes": 2 * fs * 1024, "write_bytes": ws * 1024,
                             "fetch_size_kib_raw": fs, "write_size_kib_raw": ws}
# the workload the counters were taken on (bench.py only reports traffic for the same one)
workload = None
try:
  for line in open(os.path.join(out, "bench_under_trace.json")):
    if line.startswith("{"):
      c = json.loads(line)["config"]
      workload = [c["keys"], c["batch"], c["dim"], c["zipf"]]
except (OSError, ValueError, KeyError):
  pass
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), mean of the last 3 launches; "
                     "reads = 2 x FETCH_SIZE (gfx950 tallies 128-B requests as 64 B)", "kernels": traffic,
           "workload": workload},
          open(os.path.join(out, "traffic.json"), "w"), indent=1)
try:
  b = json.loads([l for l in open(os.path.join(out, "bench_under_trace.json")) if l.startswith("{")][-1])   # (RCCL prints a banner to stdout)
  print(); print("bench.py line of the traced run:"); print(json.dumps({k: b[k] for k in ("ms_per_step", "kernels_ms", "roofline")}))
except Exception as e:
  print("no bench json:", e)
