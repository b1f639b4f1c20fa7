"""Print the kernel timeline of the last full step from a rocprofv3 results .db (rocpd): start, duration, name.
usage: python tools/timeline.py gpurun_out/prof_x/x_results.db [anchor-kernel-substring] [anchor launches per step] [step counted from the end]"""
import sqlite3
import sys

db = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_tile"
per_step = int(sys.argv[3]) if len(sys.argv) > 3 else 2   # anchor launches per step
back = int(sys.argv[4]) if len(sys.argv) > 4 else 1        # which step, counted from the end
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))
idx = [i for i, r in enumerate(rows) if anchor in r[0]]
a, b = idx[-(back + 1) * per_step], idx[-back * per_step]
t0 = rows[a][1]
busy = 0.0
for r in rows[a:b]:
  busy += (r[2] - r[1]) / 1e3
  print("%8.1f %7.1f  %6d  %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3] // max(r[4], 1), r[0].replace("(anonymous namespace)::", "")[:100]))
print("step span %.1f us, kernels busy %.1f us" % ((rows[b][1] - t0) / 1e3, busy))
