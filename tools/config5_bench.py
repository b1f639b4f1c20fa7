"""BASELINE.json configs[4] shape on ONE GPU's share: 256 tables over 8 GPUs = 32 tables per GPU
(table-wise placement, no collective), dims cycled over {8, 16, 32, 64, 128}, cardinalities
log-uniform in [1e2, 4e7] (capped so 32 tables fit), half GroupAdam, half SparseGroupFtrl(lr .1,
accum .1) — SURVEY.md §8(d).  One step = lookup + apply on every table with --batch ids each,
through the batched ops grouped by (optimizer, dim): 10 groups x 5 launches."""
import argparse, ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as ops

ap = argparse.ArgumentParser()
ap.add_argument("--tables", type=int, default=32)
ap.add_argument("--batch", type=int, default=32768)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--max-keys", type=float, default=4e6, help="cap of the log-uniform cardinalities (4e7 in the config)")
ap.add_argument("--per-table", action="store_true", help="baseline: one op per table instead of the batched ops")
ap.add_argument("--sharded", action="store_true",
                help="the sharded ops at world 1 (kv_multi_shard_lookup / _apply over a one-rank communicator: every phase but the wire), "
                     "one call per optimizer over all its tables")
ap.add_argument("--streams", type=int, default=1, help="the groups of same-shaped tables are independent: spread them over this many HIP streams")
args = ap.parse_args()
dev = torch.device("cuda", 0)
rng = np.random.default_rng(20250215)
gen = torch.Generator(device=dev).manual_seed(5)
DIMS = [8, 16, 32, 64, 128]
groups = {}
total_keys = 0
for t in range(args.tables):
  D = DIMS[t % 5]
  opt = "adam" if (t // 5) % 2 == 0 else "ftrl"
  K = int(np.exp(rng.uniform(np.log(1e2), np.log(args.max_keys))))
  total_keys += K
  var = ops.kv_variable([D], capacity_hint=K + args.batch)
  ops.init_kv_variable_v2(var, torch.randn(1000, D, device=dev, generator=gen) * 0.05)
  if opt == "adam":
    slots = [ops.kv_variable([3 * D], capacity_hint=K + args.batch)]
    ops.init_kv_variable_v2(slots[0], torch.zeros(4, 3 * D, device=dev))
  else:
    slots = [ops.kv_variable([D], capacity_hint=K + args.batch), ops.kv_variable([D], capacity_hint=K + args.batch)]
    ops.init_kv_variable_v2(slots[0], torch.full((4, D), 0.1, device=dev))
    ops.init_kv_variable_v2(slots[1], torch.zeros(4, D, device=dev))
  z = bench.Zipf(K, 1.1, dev)
  ids = [bench.splitmix64(z.sample(args.batch, gen)) for _ in range(4)]
  grads = [torch.randn(args.batch, D, device=dev, generator=gen) * 1e-2 for _ in range(2)]
  groups.setdefault((opt, D), []).append((var, slots, ids, grads))

comm = None
if args.sharded:
  comm = ops.KvComm(1, 0, None, 0)
  for key, ms in groups.items():
    groups[key] = [m + (ops.KvShard(m[0], 1, 0, ops.KV_OWNER_HASH, max_ids=args.batch, peer_capacity=args.batch),) for m in ms]
    for m in groups[key]:
      m[4].set_lossless(False)   # (capacity = the batch: cannot overflow)
  by_opt = {"adam": [m for (o, _), ms in groups.items() if o == "adam" for m in ms],
            "ftrl": [m for (o, _), ms in groups.items() if o == "ftrl" for m in ms]}

def sharded_step(k):
  for opt, ms in by_opt.items():
    if not ms:
      continue
    sh = [m[4] for m in ms]
    ops.kv_multi_shard_lookup(sh, comm, [m[2][k % 4] for m in ms])
    gr = [m[3][k % 2] for m in ms]
    if opt == "adam":
      ops.kv_multi_shard_apply(sh, comm, ops.OPT_GROUP_ADAM_V4, [m[1] for m in ms], gr, (1e-3, 0.9, 0.999, 0.9, 0.999, 1e-8, 0, 0, 0))
    else:
      ops.kv_multi_shard_apply(sh, comm, ops.OPT_SPARSE_GROUP_FTRL, [m[1] for m in ms], gr, (0.1, 0, 0, 0, 0, -0.5))

streams = [torch.cuda.Stream(dev) for _ in range(args.streams)] if args.streams > 1 else None

def step(k):
  if args.sharded:
    return sharded_step(k)
  for gi, ((opt, D), ms) in enumerate(groups.items()):
    if streams:
      with torch.cuda.stream(streams[gi % len(streams)]):
        group_step(k, opt, ms)
    else:
      group_step(k, opt, ms)

def group_step(k, opt, ms):
  if True:
    vs = [m[0] for m in ms]; ids = [m[2][k % 4] for m in ms]; gr = [m[3][k % 2] for m in ms]
    if args.per_table:
      for m, i, g in zip(ms, ids, gr):
        ops.kv_variable_gather_or_insert_v2(m[0], i)
        if opt == "adam":
          ops.kv_variable_group_sparse_apply_adam_v4(m[0], m[1][0], g, i, 1e-3, 0.9, 0.999, 0.9, 0.999, 1e-8, 0, 0, 0)
        else:
          ops.kv_variable_sparse_group_sparse_apply_ftrl_v2(m[0], m[1][0], m[1][1], g, i, 0.1, 0, 0, 0, 0, -0.5)
      return
    ops.kv_multi_gather_or_insert(vs, ids)
    if opt == "adam":
      ops.kv_multi_group_sparse_apply_adam(vs, [m[1][0] for m in ms], gr, ids, 1e-3, 0.9, 0.999, 0.9, 0.999, 1e-8, 0, 0, 0)
    else:
      ops.kv_multi_sparse_group_sparse_apply_ftrl(vs, [m[1][0] for m in ms], [m[1][1] for m in ms], gr, ids, 0.1, 0, 0, 0, 0, -0.5)

for k in range(6): step(k)
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(args.steps): step(k)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / args.steps
print("%s%s: %d tables (%d groups), %d ids each, key space %.1f M: %.3f ms/step, %.1f M ids/s (lookup + apply)" % (
    "sharded ops, world 1" if args.sharded else "per-table ops" if args.per_table else "batched ops", ", %d streams" % args.streams if streams else "", args.tables, len(groups), args.batch, total_keys / 1e6, dt * 1e3,
    args.tables * args.batch / dt / 1e6))
