"""What a streaming per-position gather costs INSIDE the training loop (behind the previous step's apply) — the question
behind round 4's k_lrows (49 us in the loop against 29 us for the inference gather alone).  Steps of
{kv_gather_or_zeros; kv_apply_group_adam without a token} against {token lookup; token apply}, configs[1] shape.
python tools/goz_loop.py [keys]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as ops
dev = torch.device("cuda", 0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
N, D = 1_000_000, 32
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
def ev(): return torch.cuda.Event(enable_timing=True)
def run(kind, steps=40):
  t_l = t_a = 0.0
  for s in range(steps + 5):
    ids, g = pool[s % 4]
    a, b, c = ev(), ev(), ev()
    a.record()
    if kind == "goz":
      ops.kv_variable_gather_or_zeros_v2(var, ids)
      b.record()
      ops.kv_variable_group_sparse_apply_adam_v4(var, slot, g, ids, *hp)
    else:
      ops.kv_variable_gather_or_insert_v2(var, ids)   # (hands its batch token to the apply below: same ids tensor)
      b.record()
      ops.kv_variable_group_sparse_apply_adam_v4(var, slot, g, ids, *hp)
    c.record()
    torch.cuda.synchronize()
    if s >= 5:
      t_l += a.elapsed_time(b); t_a += b.elapsed_time(c)
  print("%-6s lookup %.1f us  apply %.1f us  step %.1f us" % (kind, t_l / steps * 1e3, t_a / steps * 1e3, (t_l + t_a) / steps * 1e3))
run("goz"); run("train"); run("goz")
