"""Why the same gather costs 29 us alone and 50 us inside the training loop: the gather (kv_gather_or_zeros, 1 M Zipf(1.2)
ids, dim 32) timed by events behind different predecessors.  python tools/inloop_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfplus_amd import _lib
if len(sys.argv) > 1:
  _lib.SO_PATH = os.path.abspath(sys.argv[1])   # (another build of the library)
import bench
from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as ops
dev = torch.device("cuda", 0)
K, N, D = 20_000_000, 1_000_000, 32
gen = torch.Generator(device=dev).manual_seed(1)
var = ops.kv_variable([D], capacity_hint=K + 8 * N)
slot = ops.kv_variable([3 * D], capacity_hint=K + 8 * N)
ops.init_kv_variable_v2(var, torch.randn(1000, D, device=dev) * 0.05)
ops.init_kv_variable_v2(slot, torch.zeros(16, 3 * D, device=dev))
for i in range(0, K, 1 << 22):
  keys = bench.splitmix64(torch.arange(i + 1, min(i + (1 << 22), K) + 1, device=dev))
  ops.kv_variable_gather_or_insert_v2(var, keys); ops.kv_variable_gather_or_insert_v2(slot, keys)
ops.kv_attach_slot(var, slot)
z = bench.Zipf(K, 1.2, dev)
pool = [(bench.splitmix64(z.sample(N, gen)), torch.randn(N, D, device=dev, generator=gen) * 1e-2) for _ in range(4)]
hp = (1e-3, 0.9, 0.999, 0.9, 0.999, 1e-8, 0.0, 0.0, 0.0)
big_r = torch.empty(64 << 20, dtype=torch.float32, device=dev).normal_()      # 256 MB
big_w = torch.empty(32 << 20, dtype=torch.float32, device=dev)                # 128 MB
def ev(): return torch.cuda.Event(enable_timing=True)
def apply(s): ops.kv_variable_group_sparse_apply_adam_v4(var, slot, pool[s % 4][1], pool[s % 4][0], *hp)
def rd(s): big_r.sum()
def wr(s): big_w.fill_(1.0)
def rdwr(s): big_r.sum(); big_w.fill_(1.0)
def none(s): pass
def run(name, pred, reps=30, twice=False, gap=False):
  t1 = t2 = 0.0
  for s in range(reps + 3):
    ids = pool[s % 4][0]
    pred(s)
    if gap: torch.cuda.synchronize()
    a, b, c = ev(), ev(), ev()
    a.record(); ops.kv_variable_gather_or_zeros_v2(var, ids); b.record()
    if twice: ops.kv_variable_gather_or_zeros_v2(var, ids)
    c.record(); torch.cuda.synchronize()
    if s >= 3: t1 += a.elapsed_time(b); t2 += b.elapsed_time(c)
  print("%-58s gather %.1f us%s" % (name, t1 / reps * 1e3, ("   the same gather again %.1f us" % (t2 / reps * 1e3)) if twice else ""))
run("alone (the previous gather in front)", none)
run("behind the optimizer apply (no token)", apply, twice=True)
run("behind the apply + a stream synchronisation", apply, gap=True)
run("behind a 256 MB read (torch sum)", rd)
run("behind a 128 MB write (torch fill)", wr)
run("behind a 256 MB read + a 128 MB write", rdwr, twice=True)
