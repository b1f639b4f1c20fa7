"""Diagnostic: phase times inside k_papply (kv_papply.h) from the -DKV_STAMPS build
(make -C tfplus_amd/csrc libkvhip_stamps.so).  One lookup + GroupAdam apply (batch token) of 1 M Zipf ids.
usage: python tools/stamps_r4.py [keys] [zipf s]"""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tfplus_amd import _lib
_lib.SO_PATH = os.environ.get("KV_STAMPS_SO", os.path.join(_lib.CSRC, "libkvhip_stamps.so"))
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
for rep in range(4):
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
pt = a[4096:4096 + 2048].astype(np.int64); pt = pt[pt[:, 3] > 0]
pt = pt[pt[:, 0] > pt[:, 0].max() - 30000]   # the last launch only (earlier batches ran other partition counts)
if len(pt):
  b0 = pt[:, 0].min()
  print("k_papply: %d blocks" % len(pt))
  for nm, i, j in (("directory", 0, 5), ("pass 1 (entries -> LDS hash)", 5, 1), ("scans + source list", 1, 2), ("apply phase", 2, 3)):
    print("   %-48s %s" % (nm, med(pt[:, j] - pt[:, i])))
  print("   block total %s ; start %s ; last end %d" % (med(pt[:, 3] - pt[:, 0]), med(pt[:, 0] - b0), (pt[:, 3] - b0).max()))
  E_, nu_, nh_ = pt[:, 6], pt[:, 7], pt[:, 8]
  print("   entries %s" % med(E_)); print("   keys    %s" % med(nu_)); print("   hot     %s" % med(nh_))
  tot = pt[:, 3] - pt[:, 0]
  print("   corr(total, entries) %.2f  corr(total, keys) %.2f corr(apply, hot) %.2f" % (np.corrcoef(tot, E_)[0, 1], np.corrcoef(tot, nu_)[0, 1], np.corrcoef(pt[:, 3] - pt[:, 2], nh_)[0, 1]))
  slow = np.argsort(-(pt[:, 3] - b0))[:8]
  for i in slow: print("     slow block: E %d keys %d hot %d phases %s end %d" % (E_[i], nu_[i], nh_[i], [int(pt[i, j + 1] - pt[i, j]) for j in range(3)], pt[i, 3] - b0))
wv = a[8192:8192 + 8192].astype(np.int64); wv = wv[wv[:, 1] > 0]
wv = wv[wv[:, 0] > wv[:, 0].max() - 30000]
if len(wv):
  print("   per wave (%d): apply span %s" % (len(wv), med(wv[:, 1] - wv[:, 0])))
  print("   hot items per wave %s ; time per hot item %s" % (med(wv[:, 4]), med(wv[:, 2][wv[:, 4] > 0] / wv[:, 4][wv[:, 4] > 0])))
  print("   cold batches per wave %s ; time per batch %s" % (med(wv[:, 5]), med(wv[:, 3][wv[:, 5] > 0] / wv[:, 5][wv[:, 5] > 0])))

tt = a[2048:2048 + 1024].astype(np.int64); tt = tt[tt[:, 1] > 0]
if len(tt):
  tt = tt[tt[:, 0] > tt[:, 0].max() - 30000]
  print("k_tsum: %d blocks, block time %s, last end %d" % (len(tt), med(tt[:, 1] - tt[:, 0]), (tt[:, 1] - tt[:, 0].min()).max()))
