"""Diagnostic: per-phase cycle shares of k_dedup_find from the -DKV_STAMPS build."""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tfplus_amd import _lib
_lib.SO_PATH = os.path.join(_lib.CSRC, "libkvhip_stamps.so")
from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as ops
L = _lib.lib()
dev = torch.device("cuda", 0)
K, N, D = 5_000_000, 1_000_000, 32
gen = torch.Generator(device=dev).manual_seed(1)
var = ops.kv_variable([D], capacity_hint=K + 4 * N)
ops.init_kv_variable_v2(var, torch.randn(1000, D, device=dev))
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
buf = torch.empty((1 << 22, D), device=dev)
for i in range(0, K, 1 << 22):
  keys = bench.splitmix64(torch.arange(i + 1, min(i + (1 << 22), K) + 1, device=dev))
  _lib.check(L.kv_gather_or_insert(var.ptr, keys.data_ptr(), None, keys.numel(), buf.data_ptr(), st))
z = bench.Zipf(K, float(sys.argv[1]) if len(sys.argv) > 1 else 1.2, dev)
out = torch.empty((N, D), device=dev)
for rep in range(3):
  ids = bench.splitmix64(z.sample(N, gen))
  _lib.check(L.kv_gather_or_insert(var.ptr, ids.data_ptr(), None, N, out.data_ptr(), st))
nb = (N + 2047) // 2048
a = np.zeros((nb, 16), np.uint64)
L.kv_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
L.kv_debug_read_stamps(var.ptr, a.ctypes.data, nb)
t = a[:, :8].astype(np.int64)
d = np.diff(t, axis=1)
names = ["init+p1 LDS insert", "p2a compact", "p2b scratch", "ctr atomic", "p2c owners", "p3 init+p4 write", "-"]
print("blocks", nb, "nwork mean", a[:, 8].mean(), "nown mean", a[:, 9].mean())
for k in range(6):
  print("%-22s median %8.0f  p90 %8.0f  max %8.0f cycles" % (names[k], np.median(d[:, k]), np.percentile(d[:, k], 90), d[:, k].max()))
tot = t[:, 6] - t[:, 0]
print("block total: median %.0f p90 %.0f max %.0f cycles; kernel span %.0f cycles" % (np.median(tot), np.percentile(tot, 90), tot.max(), t[:, 6].max() - t[:, 0].min()))
print("start spread: %.0f cycles" % (t[:, 0].max() - t[:, 0].min()))
