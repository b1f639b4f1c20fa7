"""Per-op cost at dcn-like shapes: 26 tables, 2048 ids each (config 3)."""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfplus_amd import _lib
from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as ops
L = _lib.lib(); dev = torch.device("cuda", 0)
T_, N, D = 26, int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 64
g = torch.Generator(device=dev).manual_seed(0)
tabs = []
for t in range(T_):
  v = ops.kv_variable([D], capacity_hint=200_000); ops.init_kv_variable_v2(v, torch.randn(10000, D, device=dev))
  s = ops.kv_variable([3 * D], capacity_hint=200_000); ops.init_kv_variable_v2(s, torch.zeros(16, 3 * D, device=dev))
  tabs.append((v, s))
ids = [torch.randint(0, 10000, (N,), device=dev, generator=g) for _ in range(T_)]
grads = [torch.randn(N, D, device=dev, generator=g) * 1e-2 for _ in range(T_)]
outs = [torch.empty(N, D, device=dev) for _ in range(T_)]
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 1     # tables are independent: spread them over NS streams
streams = [torch.cuda.Stream(dev) for _ in range(NS)] if NS > 1 else [torch.cuda.current_stream(dev)]
sts = [ctypes.c_void_p(s.cuda_stream) for s in streams]
def step():
  for t in range(T_):
    v, s = tabs[t]
    _lib.check(L.kv_gather_or_insert(v.ptr, ids[t].data_ptr(), None, N, outs[t].data_ptr(), sts[t % NS]))
  for t in range(T_):
    v, s = tabs[t]
    _lib.check(L.kv_apply_group_adam(v.ptr, s.ptr, grads[t].data_ptr(), ids[t].data_ptr(), N, 1e-3, 0.9, 0.999, 0.9, 0.999, 1e-8, 0., 0., 0., 4, sts[t % NS]))
if len(sys.argv) > 3 and sys.argv[3] == "multi":
  # one batched call per op kind: 3 + 2 launches for all tables
  vp = (ctypes.c_void_p * T_)(*[t[0].ptr for t in tabs]); sp = (ctypes.c_void_p * T_)(*[t[1].ptr for t in tabs])
  ip = (ctypes.c_void_p * T_)(*[i.data_ptr() for i in ids]); gp = (ctypes.c_void_p * T_)(*[g_.data_ptr() for g_ in grads])
  op = (ctypes.c_void_p * T_)(*[o.data_ptr() for o in outs]); nsz = (ctypes.c_int64 * T_)(*([N] * T_))
  f = ctypes.c_float
  toks = (ctypes.c_uint64 * T_)()
  use_tok = not (len(sys.argv) > 4 and sys.argv[4] == "notoken")   # batch tokens: the apply takes over the lookup's index
  def step():
    _lib.check(L.kv_multi_gather_or_insert_tok(T_, vp, ip, None, nsz, op, toks if use_tok else None, sts[0]))
    _lib.check(L.kv_multi_apply_group_adam_tok(T_, vp, sp, gp, ip, nsz, f(1e-3), f(0.9), f(0.999), f(0.9), f(0.999), f(1e-8),
                                               f(0), f(0), f(0), 4, toks if use_tok else None, sts[0]))
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 50
for _ in range(K): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("streams=%d " % NS, end=""); print("N=%d: %d tables lookup+apply: host issue %.3f ms/step, total %.3f ms/step -> %.1f us per op pair, %.2f M ids/s" % (
    N, T_, (t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3, (t2 - t0) / K / T_ * 1e6, T_ * N / ((t2 - t0) / K) / 1e6))
