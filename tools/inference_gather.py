"""GatherOrZeros (inference, one probe per id, no dedup) next to GatherOrInsert at 1 M ids: Zipf(1.2) 30 vs 56 us per call, Zipf(0.3) 83 vs 221 us."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as ops
dev = torch.device("cuda", 0)
K, N, D = 20_000_000, 1_000_000, 32
h = ops.kv_variable([D], capacity_hint=K + N)
ops.init_kv_variable_v2(h, torch.randn(1000, D, device=dev))
for i in range(0, K, 1 << 21):
  ops.kv_variable_gather_or_insert_v2(h, bench.splitmix64(torch.arange(i + 1, min(i + (1 << 21), K) + 1, device=dev)))
gen = torch.Generator(device=dev).manual_seed(1)
for zipf in (1.2, 0.3):
  z = bench.Zipf(K, zipf, dev)
  batches = [bench.splitmix64(z.sample(N, gen)) for _ in range(4)]
  for name, fn in (("gather_or_zeros", ops.kv_variable_gather_or_zeros_v2), ("gather_or_insert", ops.kv_variable_gather_or_insert_v2)):
    for b in batches: fn(h, b)
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for k in range(20): fn(h, batches[k % 4])
    e.record(); torch.cuda.synchronize()
    print("zipf %.1f %-17s %.1f us per 1M-id call" % (zipf, name, s.elapsed_time(e) / 20 * 1e3))
