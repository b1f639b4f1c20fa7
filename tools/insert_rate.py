"""All-new-key lookups (cold start): 1 M distinct unseen keys per call into a pre-sized table."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as ops
dev = torch.device("cuda", 0)
h = ops.kv_variable([32], capacity_hint=40_000_000)
ops.init_kv_variable_v2(h, torch.randn(10000, 32, device=dev))
batches = [bench.splitmix64(torch.arange(i + 1, i + (1 << 20) + 1, device=dev)) for i in range(0, 1 << 25, 1 << 20)]
ops.kv_variable_gather_or_insert_v2(h, batches[0]); torch.cuda.synchronize()
ops.kv_profile_enable(h, 400)
t0 = time.perf_counter()
for b in batches[1:]:
  ops.kv_variable_gather_or_insert_v2(h, b)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
p = ops.kv_profile_read(h)
print("all-new 1M-key lookups: %.3f ms each; kernels (us):" % (dt / 31 * 1e3), {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in p.items() if v[1]})
