"""Diagnostic: phase shares of k_tile<APPLY> and k_part_sum<APPLY> (-DKV_STAMPS build)."""
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
slot = ops.kv_variable([3 * D], capacity_hint=K + 4 * N)
ops.init_kv_variable_v2(var, torch.randn(1000, D, device=dev))
ops.init_kv_variable_v2(slot, torch.zeros(4, 3 * D, device=dev))
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
buf = torch.empty((1 << 21, 3 * D), device=dev)
for i in range(0, K, 1 << 21):
  keys = bench.splitmix64(torch.arange(i + 1, min(i + (1 << 21), K) + 1, device=dev))
  _lib.check(L.kv_gather_or_insert(var.ptr, keys.data_ptr(), None, keys.numel(), buf.data_ptr(), st))
  _lib.check(L.kv_gather_or_insert(slot.ptr, keys.data_ptr(), None, keys.numel(), buf.data_ptr(), st))
z = bench.Zipf(K, 1.2, dev)
for rep in range(3):
  ids = bench.splitmix64(z.sample(N, gen))
  grad = torch.randn(N, D, device=dev, generator=gen) * 1e-2
  _lib.check(L.kv_apply_group_adam(var.ptr, slot.ptr, grad.data_ptr(), ids.data_ptr(), N, 1e-3, 0.9, 0.999, 0.9, 0.999, 1e-8, 0., 0., 0., 4, st))
a = np.zeros((8192, 16), np.uint64)
L.kv_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
L.kv_debug_read_stamps(var.ptr, a.ctypes.data, 8192)
def rep(t, names, label):
  t = t.astype(np.int64)
  print(label, "blocks", len(t))
  for k, nm in enumerate(names):
    d = t[:, k + 1] - t[:, k]
    print("   %-34s median %8.0f  p90 %8.0f  max %8.0f" % (nm, np.median(d), np.percentile(d, 90), d.max()))
  tot = t[:, len(names)] - t[:, 0]
  print("   block total median %.0f p90 %.0f max %.0f ; kernel span %.0f ticks(10ns)" % (np.median(tot), np.percentile(tot, 90), tot.max(), t[:, len(names)].max() - t[:, 0].min()))
nt = (N + 2047) // 2048
rep(a[:nt, :7], ["init + LDS hash insert", "compact + partition sort + entries", "slot_of_id writes", "multi-key offsets + perm", "chunk fold (all multi rows)", "spanning keys store"], "k_tile<APPLY>")
pt = a[4096:4096 + 1024]
bid = np.nonzero(pt[:, 4] > 0)[0]
pt = pt[pt[:, 4] > 0]
t0 = pt[:, 0].astype(np.int64); t4 = pt[:, 4].astype(np.int64)
o = np.arange(len(t0)); cl = bid % 8     # workgroups are dealt round-robin to the 8 XCDs, one clock each
for c in range(cl.max() + 1):
  m = o[cl == c]; b = t0[m].min()
  print("  clock domain %d: %4d blocks, start skew median %6d p90 %6d max %6d ; last end %6d" % (c, len(m), np.median(t0[m] - b), np.percentile(t0[m] - b, 90), (t0[m] - b).max(), (t4[m] - b).max()))
rep(pt[:, :5], ["count + copy entries + hash", "group + probes (1 thread/key)", "heavy keys (flattened fold)", "per-key rows + update"], "k_part_sum<APPLY>")
print("   entries/round median %d max %d ; rounds max %d ; uniques median %d max %d" % (np.median(pt[:, 8]), pt[:, 8].max(), pt[:, 9].max(), np.median(pt[:, 10]), pt[:, 10].max()))

r = pt[:, 11:15].astype(np.int64); t3 = pt[:, 3].astype(np.int64); t4 = pt[:, 4].astype(np.int64)
ok = (r[:, 0] > 0) & (r[:, 2] > 0)
print("   group 0 of each block, rows phase: start -> grads of round 0 summed %d ; round 0 -> 1 %d ; round 1 -> 2 %d ; last sum -> phase end %d (median ticks)" % (
    np.median(r[ok, 0] - t3[ok]), np.median(r[ok, 1] - r[ok, 0]), np.median(r[ok, 2] - r[ok, 1]), np.median(t4[ok] - r[ok, 2])))
# blocks by entry count: where the tail comes from
E = pt[:, 8].astype(np.int64); ph = np.diff(pt[:, :5].astype(np.int64), axis=1); tot = ph.sum(1)
print("   by entries in the partition (the hottest keys add one entry per tile):")
for lo, hi in ((0, 400), (400, 700), (700, 1000), (1000, 1300), (1300, 10**6)):
  m = (E >= lo) & (E < hi)
  if m.any():
    print("     E in [%4d,%5s): %4d blocks  phases (median) %s  total median %d max %d" % (
        lo, hi if hi < 10**6 else "inf", m.sum(), np.median(ph[m], axis=0).astype(int).tolist(), np.median(tot[m]), tot[m].max()))
