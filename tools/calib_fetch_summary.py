"""Divides the FETCH_SIZE / WRITE_SIZE counters of tools/calib_fetch.hip's launches by their known request counts.
python tools/calib_fetch_summary.py <dir with pmc_FETCH_SIZE/ and pmc_WRITE_SIZE/>  -> profiles/r06_fetch_calibration.txt"""
import collections, csv, glob, os, sys
out = sys.argv[1]
NREQ, N16 = 1 << 20, (256 << 20) // 16
SH = {"k_rd16_scatter": (NREQ, 16, NREQ), "k_rd16x2_scatter": (NREQ, 32, NREQ), "k_rd64_scatter": (NREQ, 64, NREQ),
      "k_rd128_rows": (NREQ, 128, NREQ), "k_rd_stream": (N16 // 8, 128, N16 // 8),
      "k_wr4_scatter": (NREQ, 4, NREQ), "k_wr16_scatter": (NREQ, 16, NREQ), "k_wr128_rows": (NREQ, 128, NREQ),
      "k_wr_stream": (N16 // 8, 128, N16 // 8)}
val = collections.defaultdict(dict)
for f in sorted(glob.glob(os.path.join(out, "pmc_*", "*", "*_counter_collection.csv"))):
  agg = collections.defaultdict(lambda: collections.defaultdict(list))
  for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    if k in SH: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
  for k, v in agg.items():
    for c, xs in v.items():
      val[k][c] = xs[-1]          # the last of the three launches (another salt each: no line is touched twice in a launch)
print("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/calib_fetch.hip; counters are KiB")
print("%-18s %10s %7s %14s %14s %12s %12s" % ("kernel", "lines", "B/line", "FETCH_SIZE KiB", "WRITE_SIZE KiB", "fetch B/line", "write B/line"))
for k, (lines, per, _) in SH.items():
  fs, ws = val[k].get("FETCH_SIZE"), val[k].get("WRITE_SIZE")
  print("%-18s %10d %7d %14s %14s %12s %12s" % (k, lines, per, "%.0f" % fs if fs is not None else "-", "%.0f" % ws if ws is not None else "-",
                                               "%.1f" % (fs * 1024 / lines) if fs is not None else "-",
                                               "%.1f" % (ws * 1024 / lines) if ws is not None else "-"))
