"""Stage timing of the sharded lookup / apply at world = 1 (nccl self-exchange)."""
import ctypes, os, sys, time
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tfplus_amd import _lib
from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as ops, sharded
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
K, N, D = 5_000_000, 1_000_000, 32
gen = torch.Generator(device=dev).manual_seed(1)
var = ops.kv_variable([D], capacity_hint=K + 4 * N); ops.init_kv_variable_v2(var, torch.randn(1000, D, device=dev))
for i in range(0, K, 1 << 21):
  ops.kv_variable_gather_or_insert_v2(var, bench.splitmix64(torch.arange(i + 1, min(i + (1 << 21), K) + 1, device=dev)))
ids = bench.splitmix64(bench.Zipf(K, 1.2, dev).sample(N, gen)); grad = torch.randn(N, D, device=dev) * 1e-2
def T(fn, n=10):
  fn(); torch.cuda.synchronize(); t = time.perf_counter()
  for _ in range(n): r = fn()
  torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3, r
t, (uniq, ucnt, inv) = T(lambda: ops.kv_unique(var, ids, None)); print("kv_unique              %.3f ms  U=%d" % (t, uniq.numel()))
t, (b, perm, cnts) = T(lambda: ops.kv_bucket_by_owner(var, uniq, 1)); print("bucket_by_owner        %.3f ms" % t)
t, rt = T(lambda: sharded.route(uniq, None, lambda i, w: ops.kv_bucket_by_owner(var, i, w))); print("route (bucket+a2a cnt+sync) %.3f ms" % t)
t, served = T(lambda: sharded.exchange(rt, uniq, presorted=rt.bucketed_ids)); print("exchange ids           %.3f ms" % t)
t, sc = T(lambda: sharded.exchange(rt, ucnt)); print("exchange counts        %.3f ms" % t)
t, rows = T(lambda: ops.kv_variable_gather_or_insert_with_counts(var, served, sc)); print("owner lookup           %.3f ms" % t)
t, urows = T(lambda: sharded.exchange(rt, rows, reverse=True)); print("exchange rows back     %.3f ms" % t)
t, out = T(lambda: ops.kv_take_rows(urows, inv)); print("expand (kv_take_rows)   %.3f ms" % t)
t, (u2, summed, _) = T(lambda: ops.kv_dedup_segment_sum(var, ids, grad)); print("kv_dedup_segment_sum   %.3f ms" % t)
t, g = T(lambda: sharded.exchange(rt, summed[:uniq.numel()])); print("exchange grads         %.3f ms" % t)
slot = ops.kv_variable([3 * D], capacity_hint=K + 4 * N); ops.init_kv_variable_v2(slot, torch.zeros(4, 3 * D, device=dev))
t, _ = T(lambda: ops.kv_variable_group_sparse_apply_adam_v4(var, slot, g, served, 1e-3, 0.9, 0.999, 0.9, 0.999, 1e-8, 0., 0., 0.)); print("owner apply            %.3f ms" % t)
skv = sharded.ShardedKvVariable(type("S", (), {"sparse_read_with_counts": lambda self, i, c: ops.kv_variable_gather_or_insert_with_counts(var, i, c)})(),
                                bucket_fn=lambda i, w, nd=None, c=None: ops.kv_bucket_by_owner(var, i, w, nd, c, with_payload=nd is not None), unique_fn=lambda i, c: ops.kv_unique(var, i, c),
                                segsum_fn=lambda i, gg: ops.kv_dedup_segment_sum(var, i, gg), take_fn=ops.kv_take_rows,
                                    index_sum_fn=lambda g, i, n: ops.kv_unsorted_segment_sum(var, g, i, n),
                                    unique_async_fn=lambda i, c: ops.kv_unique(var, i, c, sync=False))
t, _ = T(lambda: skv.lookup(ids)); print("sharded lookup total   %.3f ms" % t)
t, _ = T(lambda: skv.apply_gradients(lambda sh, gg, sv: ops.kv_variable_group_sparse_apply_adam_v4(var, slot, gg, sv, 1e-3, 0.9, 0.999, 0.9, 0.999, 1e-8, 0., 0., 0.), grad, ids)); print("sharded apply total    %.3f ms" % t)
dist.destroy_process_group()
