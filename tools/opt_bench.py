"""Apply-kernel times of the other optimizers at the configs[1] batch shape (1 M Zipf(1.2) ids, dim 32):
Adagrad (1 slot of dim D) and SparseGroupFtrl (2 slots of dim D), next to GroupAdam (1 slot of 3 D)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as ops
dev = torch.device("cuda", 0)
K, N, D = int(os.environ.get("KEYS", 20_000_000)), 1_000_000, int(os.environ.get("DIM", 32))
gen = torch.Generator(device=dev).manual_seed(1)
keys = [bench.splitmix64(torch.arange(i + 1, min(i + (1 << 21), K) + 1, device=dev)) for i in range(0, K, 1 << 21)]
z = bench.Zipf(K, 1.2, dev)
batches = [(bench.splitmix64(z.sample(N, gen)), torch.randn(N, D, device=dev, generator=gen) * 1e-2) for _ in range(4)]
U = np.mean([int(torch.unique(b[0]).numel()) for b in batches])
def table(dim, val):
  h = ops.kv_variable([dim], capacity_hint=K + 4 * N)
  ops.init_kv_variable_v2(h, torch.full((16, dim), float(val), device=dev) if val is not None else torch.randn(1000, dim, device=dev) * 0.05)
  for k in keys: ops.kv_variable_gather_or_insert_v2(h, k)
  return h
for name in ("group_adam", "adagrad", "ftrl"):
  var = table(D, None)
  if name == "group_adam":
    slots = [table(3 * D, 0.0)]
    step = lambda g, i: ops.kv_variable_group_sparse_apply_adam_v4(var, slots[0], g, i, 1e-3, 0.9, 0.999, 0.9, 0.999, 1e-8, 0, 0, 0)
    state = 4
  elif name == "adagrad":
    slots = [table(D, 0.1)]
    step = lambda g, i: ops.kv_variable_sparse_apply_adagrad(var, slots[0], 0.01, g, i, use_locking=True)
    state = 2
  else:
    slots = [table(D, 0.1), table(D, 0.0)]
    step = lambda g, i: ops.kv_variable_sparse_group_sparse_apply_ftrl_v2(var, slots[0], slots[1], g, i, 0.1, 0.0, 0.0, 0.0, 0.0, -0.5)
    state = 3
  for k in range(4): step(batches[k][1], batches[k][0])
  ops.kv_profile_enable(var, 64)
  for k in range(20): step(batches[k % 4][1], batches[k % 4][0])
  torch.cuda.synchronize()
  p = ops.kv_profile_read(var); ops.kv_profile_enable(var, 0)
  per = {k: p[k][0] / max(p[k][1], 1) for k in p}
  idx, srt, fin = per["apply_index"] * 3, per["apply_sorted"], per["apply_span"]   # index pass = 3 launches (no token here)
  bytes_ = N * (8 + 4 * D) + U * (16 + state * 4 * D) + U * state * 4 * D
  print("%-10s index pass %6.1f us  k_apply %6.1f us  k_apply_fin %5.1f us  -> %5.2f TB/s algorithmic (%d B of state per unique key)" % (
      name, idx * 1e3, srt * 1e3, fin * 1e3, bytes_ / ((idx + srt + fin) * 1e-3) / 1e12, 2 * state * 4 * D))
  del var, slots
