"""Capacity check: K keys x dim D in one table (default 1e9 x 8), grown batch by batch without a
capacity hint (index rebuilds and slab chunks on the way), then verified by sampling:
  python tools/big_table.py [K] [D] [capacity_hint]
Rows carry a value derived from the key, so any lost / duplicated / misplaced row shows up."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as ops

K = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 8
CH = 1 << 23
dev = torch.device("cuda", 0)
HINT = int(float(sys.argv[3])) if len(sys.argv) > 3 else 0
h = ops.kv_variable([D], capacity_hint=HINT) if HINT else ops.kv_variable([D])
ops.init_kv_variable_v2(h, torch.zeros(16, D, device=dev))


def rows_of(keys):
  # a float pattern of the key: low 20 bits and the next 20 bits, exactly representable
  lo = (keys & 0xFFFFF).to(torch.float32)
  hi = ((keys >> 20) & 0xFFFFF).to(torch.float32)
  cols = torch.arange(D, device=dev, dtype=torch.float32)
  return lo[:, None] + hi[:, None] * 0.5 + cols[None, :] * 1e-3 * 0 + (cols[None, :] % 2) * hi[:, None]


t0 = time.perf_counter()
for i in range(0, K, CH):
  keys = bench.splitmix64(torch.arange(i + 1, min(i + CH, K) + 1, dtype=torch.int64, device=dev))
  ops.kv_variable_insert_v2(h, keys, rows_of(keys))
  if (i // CH) % 16 == 0:
    torch.cuda.synchronize()
    print("inserted %11d keys  %.1f s  %.1f GB allocated by torch, %.1f GB free on device" %
          (min(i + CH, K), time.perf_counter() - t0, torch.cuda.memory_allocated() / 1e9,
           torch.cuda.mem_get_info()[0] / 1e9), flush=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
size = int(ops.kv_variable_size_v2(h))
print("size", size, "expected", K, "insert rate %.2f G keys/s" % (K / dt / 1e9))
assert size == K

gen = torch.Generator(device=dev); gen.manual_seed(1)
bad = 0
for rep in range(8):
  idx = torch.randint(1, K + 1, (1 << 20,), device=dev, generator=gen, dtype=torch.int64)
  keys = bench.splitmix64(idx)
  got = ops.kv_variable_gather_or_zeros_v2(h, keys)
  bad += int((got != rows_of(keys)).any(1).sum())
  absent = bench.splitmix64(idx + K)          # never inserted
  z = ops.kv_variable_gather_or_zeros_v2(h, absent)
  bad += int((z != 0).any(1).sum())
print("sampled 8 M present + 8 M absent keys, mismatching rows:", bad)
assert bad == 0
t0 = time.perf_counter()
for rep in range(20):
  got = ops.kv_variable_gather_or_zeros_v2(h, keys)
torch.cuda.synchronize()
print("gather_or_zeros of 1 M uniform ids from the full table: %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
print("free device memory at the end: %.1f GB" % (torch.cuda.mem_get_info()[0] / 1e9))
