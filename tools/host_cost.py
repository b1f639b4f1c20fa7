"""Host time of the C-ABI calls themselves (raw ctypes, arguments prepared once) for a small batch: what a graph of many
small per-table ops pays per op.  python tools/host_cost.py [n]"""
import ctypes, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfplus_amd import _lib
from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda", 0)
D = 16
L = _lib.lib()
var = ops.kv_variable([D], capacity_hint=1 << 20)
slot = ops.kv_variable([3 * D], capacity_hint=1 << 20)
ops.init_kv_variable_v2(var, torch.randn(64, D, device=dev))
ops.init_kv_variable_v2(slot, torch.zeros(4, 3 * D, device=dev))
ids = torch.randint(0, 100000, (n,), device=dev)
uids = torch.unique(ids)
grad = torch.randn(n, D, device=dev) * 1e-2
out = torch.empty(n, D, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
tok = ctypes.c_uint64(0)
P = lambda t: ctypes.c_void_p(t.data_ptr())
hp = [ctypes.c_float(x) for x in (1e-3, 0.9, 0.999, 0.9, 0.999, 1e-8, 0.0, 0.0, 0.0)]
def lookup_tok(): _lib.check(L.kv_gather_or_insert_tok(var.ptr, P(ids), None, n, P(out), ctypes.byref(tok), st))
def lookup(): _lib.check(L.kv_gather_or_insert(var.ptr, P(ids), None, n, P(out), st))
def apply_tok(): _lib.check(L.kv_apply_group_adam_tok(var.ptr, slot.ptr, P(grad), P(ids), n, *hp, 4, tok, st))
def apply_plain(): _lib.check(L.kv_apply_group_adam(var.ptr, slot.ptr, P(grad), P(ids), n, *hp, 4, st))
def apply_unique(): _lib.check(L.kv_apply_group_adam_unique(var.ptr, slot.ptr, P(grad), P(uids), uids.numel(), *hp, 4, st))
def goz(): _lib.check(L.kv_gather_or_zeros(var.ptr, P(ids), n, P(out), st))
def bench(name, fns, reps=300):
  for _ in range(50):
    for f in fns: f()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(reps):
    for f in fns: f()
  t1 = time.perf_counter()
  torch.cuda.synchronize()
  t2 = time.perf_counter()
  print("%-34s host issue %6.1f us per iteration   (drain after the loop %.1f ms)" % (name, (t1 - t0) / reps * 1e6, (t2 - t1) * 1e3))
bench("gather_or_zeros", [goz])
bench("gather_or_insert", [lookup])
bench("lookup_tok + apply_tok", [lookup_tok, apply_tok])
bench("lookup + apply (no token)", [lookup, apply_plain])
bench("apply_unique", [apply_unique])
def empty(): pass
bench("(python loop + empty call)", [empty])
