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
a = np.zeros((8192, 16), np.uint64)
L.kv_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
L.kv_debug_read_stamps(var.ptr, a.ctypes.data, 8192)
tt = a[:(N + 2047) // 2048, :4].astype(np.int64)
print("k_tile<LOOKUP> phases (median ticks(10ns)):", [int(np.median(tt[:, k + 1] - tt[:, k])) for k in range(3)])
a = a[4096:4096 + nb]
bid = np.nonzero(a[:, 4] > 0)[0]
a = a[a[:, 4] > 0]
t0 = a[:, 0].astype(np.int64); t4 = a[:, 4].astype(np.int64)
o = np.arange(len(t0)); cl = bid % 8     # workgroups are dealt round-robin to the 8 XCDs, one clock each
for c in range(cl.max() + 1):
  m = o[cl == c]; b = t0[m].min()
  print("  clock domain %d: %4d blocks, start skew median %6d p90 %6d max %6d ; last end %6d" % (c, len(m), np.median(t0[m] - b), np.percentile(t0[m] - b, 90), (t0[m] - b).max(), (t4[m] - b).max()))
t = a[:, :8].astype(np.int64)
E = a[:, 8].astype(np.int64); R = a[:, 9].astype(np.int64); NU = a[:, 10].astype(np.int64)
names = ["pass1 stream+hash", "owner (1 thread/key)", "lane work (new/dirty)", "pass2 writeback"]
for k in range(4):
  d = t[:, k + 1] - t[:, k]
  print("%-18s median %8.0f  p90 %8.0f  max %8.0f ticks(10ns)" % (names[k], np.median(d), np.percentile(d, 90), d.max()))
tot = t[:, 4] - t[:, 0]
print("block total: median %.0f p90 %.0f max %.0f; kernel span %.0f" % (np.median(tot), np.percentile(tot, 90), tot.max(), t[:, 4].max() - t[:, 0].min()))
print("uniques: median %d max %d ; rounds max %d ; lane-work rows median %d max %d" % (np.median(E), E.max(), R.max(), np.median(NU), NU.max()))
i = int(np.argmax(tot)); print("slowest block", i, "uniq", E[i], "R", R[i], "lanework", NU[i], (t[i, 1:5] - t[i, 0:4]))
