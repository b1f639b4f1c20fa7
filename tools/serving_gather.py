"""BatchKvVariableGatherOrZerosV2 (one launch for all tables) at two serving shapes: 26 tables x 2048 ids
(configs[2]'s feature count) and 8 tables x 200 k ids; dims 64 / 128.  python tools/serving_gather.py [other.so]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfplus_amd import _lib
if len(sys.argv) > 1:
  _lib.SO_PATH = os.path.abspath(sys.argv[1])
import bench
from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as ops
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
for ntab, n, keys in ((26, 2048, 200_000), (8, 200_000, 2_000_000)):
  hs, idl = [], []
  for i in range(ntab):
    D = 64 if i % 2 == 0 else 128
    h = ops.kv_variable([D], capacity_hint=keys + 1024)
    ops.init_kv_variable_v2(h, torch.randn(256, D, device=dev))
    for j in range(0, keys, 1 << 20):
      ops.kv_variable_gather_or_insert_v2(h, torch.arange(j, min(j + (1 << 20), keys), device=dev))
    hs.append(h)
    idl.append((torch.rand(n, device=dev, generator=g) ** 3 * keys).long())     # mildly skewed towards small keys
  z = bench.Zipf(keys, 1.2, dev)
  zl = [(z.sample(n, g) - 1) for _ in range(ntab)]                              # Zipf(1.2) over the same keys
  for name, lists in (("cubic skew", idl), ("Zipf(1.2)", zl)):
    for _ in range(3): ops.batch_kv_variable_gather_or_zeros_v2(hs, lists)
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): ops.batch_kv_variable_gather_or_zeros_v2(hs, lists)
    e.record(); torch.cuda.synchronize()
    print("%2d tables x %6d ids, %-10s: %.1f us per batched call" % (ntab, n, name, s.elapsed_time(e) / 20 * 1e3))
