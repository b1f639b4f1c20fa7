"""Diagnostic: phase times inside k_apply_sorted (-DKV_STAMPS build: make -C tfplus_amd/csrc libkvhip_stamps.so).
One lookup + GroupAdam apply (batch token) of 1 M Zipf(1.2) ids on a K-key table."""
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
z = bench.Zipf(K, 1.2, dev)
out = torch.empty((N, D), device=dev)
for rep in range(3):
  ids = bench.splitmix64(z.sample(N, gen))
  grad = torch.randn(N, D, device=dev, generator=gen) * 1e-2
  tok = ctypes.c_uint64(0)
  _lib.check(L.kv_gather_or_insert_tok(var.ptr, ids.data_ptr(), None, N, out.data_ptr(), ctypes.byref(tok), st))
  _lib.check(L.kv_apply_group_adam_tok(var.ptr, slot.ptr, grad.data_ptr(), ids.data_ptr(), N, 1e-3, 0.9, 0.999, 0.9, 0.999, 1e-8, 0., 0., 0., 4, tok.value, st))
a = np.zeros((16384, 16), np.uint64)
L.kv_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
L.kv_debug_read_stamps(var.ptr, a.ctypes.data, 16384)
nch = (N + 255) // 256
t = a[8192:8192 + nch].astype(np.int64)
print("k_apply_sorted: %d blocks; ticks of 10 ns" % nch)
def med(x): return "median %6.0f p90 %6.0f max %6.0f" % (np.median(x), np.percentile(x, 90), x.max())
print("  preamble (order, heads, scan)   ", med(t[:, 1] - t[:, 0]))
print("  long segments                   ", med(t[:, 2] - t[:, 1]))
print("  group loop                      ", med(t[:, 11] - t[:, 2]))
print("  block total                     ", med(t[:, 11] - t[:, 0]), " kernel span %d" % (t[:, 11].max() - t[:, 0].min()))
nseg, nl = t[:, 12], t[:, 13]
print("  segments per chunk median %d p10 %d p90 %d ; long segments mean %.2f" % (np.median(nseg), np.percentile(nseg, 10), np.percentile(nseg, 90), nl.mean()))
for r in range(4):
  ok = t[:, 4 + 2 * r] > 0
  if ok.sum() == 0: continue
  prev = t[ok, 2] if r == 0 else t[ok, 2 + 2 * r]
  print("  group 0 round %d (%4d blocks): fold+touch %s | resolve+update %s" % (r, ok.sum(), med(t[ok, 3 + 2 * r] - prev), med(t[ok, 4 + 2 * r] - t[ok, 3 + 2 * r])))
for lo, hi in ((0, 1), (1, 8), (8, 64), (64, 160), (160, 257)):
  m = (nseg >= lo) & (nseg < hi)
  if m.any():
    print("  chunks with %3d..%3d keys: %4d blocks, total %s" % (lo, hi - 1, m.sum(), med((t[:, 11] - t[:, 0])[m])))
st0 = t[:, 0] - t[:, 0].min()
print("  block start times: median %d p90 %d max %d" % (np.median(st0), np.percentile(st0, 90), st0.max()))
