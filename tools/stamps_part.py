"""Diagnostic: per-phase cycle shares of k_part<LOOKUP> from the -DKV_STAMPS build."""
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
buf = torch.empty((1 << 21, D), device=dev)
for i in range(0, K, 1 << 21):
  keys = bench.splitmix64(torch.arange(i + 1, min(i + (1 << 21), K) + 1, device=dev))
  _lib.check(L.kv_gather_or_insert(var.ptr, keys.data_ptr(), None, keys.numel(), buf.data_ptr(), st))
z = bench.Zipf(K, float(sys.argv[1]) if len(sys.argv) > 1 else 1.2, dev)
out = torch.empty((N, D), device=dev)
for rep in range(3):
  ids = bench.splitmix64(z.sample(N, gen))
  _lib.check(L.kv_gather_or_insert(var.ptr, ids.data_ptr(), None, N, out.data_ptr(), st))
nb = 1024
a = np.zeros((nb, 16), np.uint64)
L.kv_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
L.kv_debug_read_stamps(var.ptr, a.ctypes.data, nb)
t = a[:, :8].astype(np.int64)
E = a[:, 8].astype(np.int64); R = a[:, 9].astype(np.int64); NU = a[:, 10].astype(np.int64)
names = ["pass1 stream+hash", "owner (1 thread/key)", "lane work (new/dirty)", "pass2 writeback"]
for k in range(4):
  d = t[:, k + 1] - t[:, k]
  print("%-18s median %8.0f  p90 %8.0f  max %8.0f cycles" % (names[k], np.median(d), np.percentile(d, 90), d.max()))
tot = t[:, 4] - t[:, 0]
print("block total: median %.0f p90 %.0f max %.0f; kernel span %.0f" % (np.median(tot), np.percentile(tot, 90), tot.max(), t[:, 4].max() - t[:, 0].min()))
print("uniques: median %d max %d ; rounds max %d ; lane-work rows median %d max %d" % (np.median(E), E.max(), R.max(), np.median(NU), NU.max()))
i = int(np.argmax(tot)); print("slowest block", i, "uniq", E[i], "R", R[i], "lanework", NU[i], (t[i, 1:5] - t[i, 0:4]))
