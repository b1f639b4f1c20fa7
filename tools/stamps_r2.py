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
t = a[8192:16384].astype(np.int64)
t = t[t[:, 1] > 0]
print("k_apply: %d waves ran; items %d, hot chunks %d; ticks of 10 ns" % (len(t), t[0, 6], t[0, 7]))
def med(x): return "median %6.0f p90 %6.0f max %6.0f" % (np.median(x), np.percentile(x, 90), x.max()) if len(x) else "-"
t0 = t[:, 0].min()
print("  wave start  ", med(t[:, 0] - t0))
print("  wave end    ", med(t[:, 1] - t0), " kernel span %d" % (t[:, 1].max() - t0))
h = t[:, 4] > 0; c = t[:, 5] > 0
print("  hot items per wave ", med(t[h, 4]), " time per hot item ", med(t[h, 2] / t[h, 4]))
print("  cold items per wave", med(t[c, 5]), " time per cold item", med(t[c, 3] / t[c, 5]))
print("  sum of hot time %.0f us*waves, cold %.0f us*waves" % (t[:, 2].sum() / 100.0, t[:, 3].sum() / 100.0))
# ---- k_part_keys_gather: the partition blocks (stamps 0..4 at rows 4096 + block)
pt = a[4096:4096 + 1024].astype(np.int64)
pt = pt[pt[:, 4] > 0]
if len(pt):
  b0 = pt[:, 0].min()
  print("k_part_keys (partition role of the lookup): %d blocks" % len(pt))
  for nm, i, j in (("directory + pass 1 (entries -> LDS hash)", 0, 1), ("scan + owner (probe, frequency, records)", 1, 2),
                   ("lane work (new rows, flags)", 2, 3), ("pass 2 (entries learn row / base)", 3, 4)):
    print("   %-44s %s" % (nm, med(pt[:, j] - pt[:, i])))
  print("   block total %s ; start %s ; last end %d" % (med(pt[:, 4] - pt[:, 0]), med(pt[:, 0] - b0), (pt[:, 4] - b0).max()))
  print("   uniques per block median %d max %d" % (np.median(pt[:, 8]), pt[:, 8].max()))
