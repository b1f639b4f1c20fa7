"""Diagnostic: phase times inside the entry-list pipeline's kernels (-DKV_STAMPS build:
make -C tfplus_amd/csrc libkvhip_stamps.so).  One lookup + GroupAdam apply (batch token) of 1 M Zipf ids."""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tfplus_amd import _lib
_lib.SO_PATH = os.path.join(_lib.CSRC, "libkvhip_stamps.so")
from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as ops
L = _lib.lib()
dev = torch.device("cuda", 0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
S = float(sys.argv[2]) if len(sys.argv) > 2 else 1.2
N, D = 1_000_000, 32
gen = torch.Generator(device=dev).manual_seed(1)
var = ops.kv_variable([D], capacity_hint=K + 4 * N)
slot = ops.kv_variable([3 * D], capacity_hint=K + 4 * N)
ops.init_kv_variable_v2(var, torch.randn(1000, D, device=dev))
ops.init_kv_variable_v2(slot, torch.zeros(4, 3 * D, device=dev))
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
buf = torch.empty((1 << 21, 3 * D), device=dev)
for i in range(0, K, 1 << 21):
  keys = bench.splitmix64(torch.arange(i + 1, min(i + (1 << 21), K) + 1, device=dev))
  _lib.check(L.kv_gather_or_insert(var.ptr, keys.data_ptr(), None, keys.numel(), buf.data_ptr(), st))
  _lib.check(L.kv_gather_or_insert(slot.ptr, keys.data_ptr(), None, keys.numel(), buf.data_ptr(), st))
ops.kv_attach_slot(var, slot)
z = bench.Zipf(K, S, dev)
out = torch.empty((N, D), device=dev)
for rep in range(3):
  ids = bench.splitmix64(z.sample(N, gen))
  grad = torch.randn(N, D, device=dev, generator=gen) * 1e-2
  tok = ctypes.c_uint64(0)
  _lib.check(L.kv_gather_or_insert_tok(var.ptr, ids.data_ptr(), None, N, out.data_ptr(), ctypes.byref(tok), st))
  _lib.check(L.kv_apply_group_adam_tok(var.ptr, slot.ptr, grad.data_ptr(), ids.data_ptr(), N, 1e-3, 0.9, 0.999, 0.9, 0.999, 1e-8, 0., 0., 0., 4, tok.value, st))
torch.cuda.synchronize()
a = np.zeros((16384, 16), np.uint64)
L.kv_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
L.kv_debug_read_stamps(var.ptr, a.ctypes.data, 16384)
def med(x): return "median %6.0f p90 %6.0f max %6.0f" % (np.median(x), np.percentile(x, 90), x.max()) if len(x) else "-"
print("ticks of 10 ns; K = %d, Zipf %.1f" % (K, S))
tt = a[0:(N + 2047) // 2048].astype(np.int64); tt = tt[tt[:, 4] > 0]
if len(tt):
  b0 = tt[:, 0].min()
  print("k_ltile: %d blocks" % len(tt))
  for nm, i, j in (("ids + LDS hash insert", 0, 1), ("compact + probes out + counting sort + entries", 1, 2), ("probes back (+ inserts)", 2, 3),
                   ("entry scan, torder, mlist", 3, 4), ("output rows", 4, 5)):
    print("   %-48s %s" % (nm, med(tt[:, j] - tt[:, i])))
  print("   block total %s ; start %s ; last end %d" % (med(tt[:, 5] - tt[:, 0]), med(tt[:, 0] - b0), (tt[:, 5] - b0).max()))
  if tt[:, 9].max() > 0:
    for nm, i, j in (("  ids requested + LDS cleared + barrier", 0, 7), ("  hash insert", 7, 8), ("  probes requested", 8, 9), ("  barrier", 9, 1)):
      print("   %-48s %s" % (nm, med(tt[:, j] - tt[:, i])))
  if tt[:, 10].max() > 0:   # where the blocks ran
    hw = tt[:, 10]
    xcc = (hw >> 32) & 0xF; cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    where = xcc * 1000 + se * 100 + sh * 16 + cu
    ids_, inv, cnt = np.unique(where, return_inverse=True, return_counts=True)
    tot = tt[:, 5] - tt[:, 0]
    print("   distinct CUs used %d; blocks per CU: %s" % (len(ids_), dict(zip(*np.unique(cnt, return_counts=True)))))
    for c in sorted(set(cnt)):
      sel = cnt[inv] == c
      print("   blocks on a CU holding %d: total %s ; rows phase %s" % (c, med(tot[sel]), med((tt[:, 5] - tt[:, 4])[sel])))
    for x in range(8):
      sel = xcc == x
      if sel.any():
        print("   XCC %d: %3d blocks, total %s ; end %s" % (x, sel.sum(), med(tot[sel]), med((tt[:, 5] - b0)[sel])))
    slow = np.argsort(-(tt[:, 5] - b0))[:12]
    print("   last blocks to end (tile, xcc, se, cu, start, phases..., end):")
    for i in slow:
      print("     ", i, int(xcc[i]), int(se[i]), int(cu[i]), int(tt[i, 0] - b0), [int(tt[i, j + 1] - tt[i, j]) for j in range(5)], int(tt[i, 5] - b0))
pt = a[4096:4096 + 1024].astype(np.int64); pt = pt[pt[:, 4] > 0]
if len(pt):
  b0 = pt[:, 0].min()
  print("k_part2: %d blocks" % len(pt))
  for nm, i, j in (("directory + pass 1 (entries -> LDS hash)", 0, 1), ("scan + owner (record hop, frequency, key records)", 1, 2),
                   ("lane work (new rows, flags)", 2, 3), ("pass 2 (entry list)", 3, 4)):
    print("   %-48s %s" % (nm, med(pt[:, j] - pt[:, i])))
  print("   block total %s ; start %s ; last end %d" % (med(pt[:, 4] - pt[:, 0]), med(pt[:, 0] - b0), (pt[:, 4] - b0).max()))
  if pt[:, 6].max() > 0:   # entries, distinct keys and place of every partition block
    tot = pt[:, 4] - pt[:, 0]
    E_, nu_ = pt[:, 6], pt[:, 7]
    print("   entries per partition %s ; distinct keys %s" % (med(E_), med(nu_)))
    print("   corr(total, entries) %.2f  corr(total, keys) %.2f" % (np.corrcoef(tot, E_)[0, 1], np.corrcoef(tot, nu_)[0, 1]))
    for lo, hi in ((0, 400), (400, 700), (700, 1200), (1200, 1 << 30)):
      sel = (E_ >= lo) & (E_ < hi)
      if sel.any():
        print("   partitions with %4d <= entries < %-6d: %4d blocks, total %s" % (lo, min(hi, 99999), sel.sum(), med(tot[sel])))
    xcc = (pt[:, 8] >> 32) & 0xF
    for x in range(8):
      sel = xcc == x
      if sel.any():
        print("   XCC %d: %4d blocks, total %s" % (x, sel.sum(), med(tot[sel])))
    slow = np.argsort(-(pt[:, 4] - b0))[:10]
    print("   last blocks to end (partition block, xcc, entries, keys, phases..., end):")
    for i in slow:
      print("     ", i, int(xcc[i]), int(E_[i]), int(nu_[i]), [int(pt[i, j + 1] - pt[i, j]) for j in range(4)], int(pt[i, 4] - b0))
ts = a[2048:2048 + 16 + 4 * ((N + 2047) // 2048)].astype(np.int64); ts = ts[ts[:, 1] > 0]
if len(ts):
  b0 = ts[:, 0].min()
  print("k_tsum: %d blocks; block time %s ; start %s ; end %s" % (len(ts), med(ts[:, 1] - ts[:, 0]), med(ts[:, 0] - b0), med(ts[:, 1] - b0)))
t = a[8192:16384].astype(np.int64); t = t[t[:, 1] > 0]
if len(t):
  print("k_apply: %d waves ran; items %d, hot chunks %d" % (len(t), t[0, 6], t[0, 7]))
  t0 = t[:, 0].min()
  print("  wave start  ", med(t[:, 0] - t0))
  print("  wave end    ", med(t[:, 1] - t0), " kernel span %d" % (t[:, 1].max() - t0))
  h = t[:, 4] > 0; c = t[:, 5] > 0
  print("  hot items per wave ", med(t[h, 4]), " time per hot item ", med(t[h, 2] / t[h, 4]))
  print("  cold items per wave", med(t[c, 5]), " time per cold item", med(t[c, 3] / t[c, 5]))
  if t[:, 8].max() > 0:
    print("  cold item: start -> rows arrived", med(t[c, 8] / t[c, 5]), " (rest = the update)")
