"""Occurrence-order mode (kv_set_deterministic(h, 2)) at the headline's batch shape: 1 M ids, Zipf 1.2, dim 32, GroupAdam —
time per step next to the default mode on the same table shape.  python tools/occ_step.py [keys [zipf]]  """
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as ops

dev = torch.device("cuda", 0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
SKEW = float(sys.argv[2]) if len(sys.argv) > 2 else 1.2
N, D = 1_000_000, 32
gen = torch.Generator(device=dev).manual_seed(bench.SEED)
z = bench.Zipf(K, SKEW, dev)
print("Zipf %.1f over %d keys" % (SKEW, K))
pool = [(bench.splitmix64(z.sample(N, gen)), torch.randn(N, D, device=dev, generator=gen) * 1e-2) for _ in range(4)]
for mode in (0, 1, 2):
  var = ops.kv_variable([D], capacity_hint=K + 4 * N)
  slot = ops.kv_variable([3 * D], capacity_hint=K + 4 * N)
  ops.init_kv_variable_v2(var, torch.randn(10000, D, device=dev) * 0.05)
  ops.init_kv_variable_v2(slot, torch.zeros(16, 3 * D, device=dev))
  for i in range(0, K, 1 << 21):
    keys = bench.splitmix64(torch.arange(i + 1, min(i + (1 << 21), K) + 1, device=dev))
    ops.kv_variable_gather_or_insert_v2(var, keys); ops.kv_variable_gather_or_insert_v2(slot, keys)
  ops.kv_attach_slot(var, slot)
  ops.kv_set_deterministic(var, mode)
  tl = ta = 0.0
  steps, warm = 12, 3
  for s in range(steps + warm):
    ids, grad = pool[s % len(pool)]
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    ops.kv_variable_gather_or_insert_v2(var, ids)
    e[1].record()
    ops.kv_variable_group_sparse_apply_adam_v4(var, slot, grad, ids, 1e-3, 0.9, 0.999, 0.9, 0.999, 1e-8, 0.0, 0.0, 0.0)
    e[2].record()
    torch.cuda.synchronize()
    if s >= warm:
      tl += e[0].elapsed_time(e[1]); ta += e[1].elapsed_time(e[2])
  print("mode %d (%s): lookup %.3f ms, apply %.3f ms, step %.3f ms" % (
      mode, ("arrival order", "fixed order", "occurrence order")[mode], tl / steps, ta / steps, (tl + ta) / steps))
  del var, slot
