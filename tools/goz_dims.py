"""kv_gather_or_zeros per 1 M-id call at several dims (Zipf 1.2 and uniform ids).  python tools/goz_dims.py [other.so]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfplus_amd import _lib
if len(sys.argv) > 1:
  _lib.SO_PATH = os.path.abspath(sys.argv[1])
import bench
from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as ops
dev = torch.device("cuda", 0)
K, N = 4_000_000, 1_000_000
gen = torch.Generator(device=dev).manual_seed(1)
zb = [bench.splitmix64(bench.Zipf(K, 1.2, dev).sample(N, gen)) for _ in range(4)]
ub = [bench.splitmix64(torch.randint(1, K + 1, (N,), device=dev, generator=gen)) for _ in range(4)]
for D in (8, 16, 32, 64, 128, 256):
  h = ops.kv_variable([D], capacity_hint=K + N)
  ops.init_kv_variable_v2(h, torch.randn(1000, D, device=dev))
  for i in range(0, K, 1 << 21):
    ops.kv_variable_gather_or_insert_v2(h, bench.splitmix64(torch.arange(i + 1, min(i + (1 << 21), K) + 1, device=dev)))
  res = []
  for batches in (zb, ub):
    for b in batches: ops.kv_variable_gather_or_zeros_v2(h, b)
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for k in range(20): ops.kv_variable_gather_or_zeros_v2(h, batches[k % 4])
    e.record(); torch.cuda.synchronize()
    res.append(s.elapsed_time(e) / 20 * 1e3)
  print("dim %3d: zipf1.2 %.1f us  uniform %.1f us  (write floor at 5.5 TB/s: %.1f us)" % (D, res[0], res[1], N * D * 4 / 5.5e6))
  del h
