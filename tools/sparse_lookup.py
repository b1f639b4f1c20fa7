"""kv_lookup_sparse (embedding_lookup_sparse in one call) per 1 M ids at several segment lengths, dim 32, Zipf(1.2)
ids, mean combiner.  python tools/sparse_lookup.py [other.so]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfplus_amd import _lib
if len(sys.argv) > 1:
  _lib.SO_PATH = os.path.abspath(sys.argv[1])
import bench
from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as ops
dev = torch.device("cuda", 0)
K, N, D = 4_000_000, 1_000_000, 32
gen = torch.Generator(device=dev).manual_seed(1)
h = ops.kv_variable([D], capacity_hint=K + N)
ops.init_kv_variable_v2(h, torch.randn(1000, D, device=dev))
for i in range(0, K, 1 << 21):
  ops.kv_variable_gather_or_insert_v2(h, bench.splitmix64(torch.arange(i + 1, min(i + (1 << 21), K) + 1, device=dev)))
ids = [bench.splitmix64(bench.Zipf(K, 1.2, dev).sample(N, gen)) for _ in range(4)]
for L in (1, 4, 16, 64, 512):
  seg = (torch.arange(N, device=dev) // L).to(torch.int64)
  nseg = (N + L - 1) // L
  for b in ids: ops.kv_variable_lookup_sparse(h, b, seg, None, nseg, "mean")
  torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s.record()
  for k in range(20): ops.kv_variable_lookup_sparse(h, ids[k % 4], seg, None, nseg, "mean")
  e.record(); torch.cuda.synchronize()
  print("segment length %3d: %.1f us per 1M-id call" % (L, s.elapsed_time(e) / 20 * 1e3))
