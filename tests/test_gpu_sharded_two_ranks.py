"""The N > 1 path with the REAL GPU pieces (kv_unique, kv_bucket_by_owner, kv_dedup_segment_sum,
kv_take_rows, HBM table shards) at world_size 2 on one GPU: two processes share cuda:0, the
all_to_all is staged through gloo on the host (RCCL refuses two ranks on one device), everything
else is the product path.  Checked against ONE unsharded oracle table fed every rank's ids."""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DAY = 20000


def _a2a_via_host(out, inp, output_split_sizes=None, input_split_sizes=None, group=None):
  o = torch.empty(out.shape, dtype=out.dtype)
  dist.all_to_all_single(o, inp.cpu(), output_split_sizes=output_split_sizes, input_split_sizes=input_split_sizes,
                         group=group)
  out.copy_(o)


def _worker(rank, world, port, q, D):
  sys.path.insert(0, ROOT)
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  try:
    from oracle import kv_oracle as ko
    from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as ops, sharded

    class _Dist(object):  # sharded.py's `dist`, with the exchange staged through the host
      get_world_size = staticmethod(dist.get_world_size)
      get_rank = staticmethod(dist.get_rank)
      all_to_all_single = staticmethod(_a2a_via_host)
    sharded.dist = _Dist

    rng = np.random.default_rng(5)
    table = rng.standard_normal((64, D)).astype(np.float32)
    var = ops.kv_variable([D]); slot = ops.kv_variable([3 * D])
    for h, t in ((var, table), (slot, np.zeros((4, 3 * D), np.float32))):
      ops.kv_set_clock_days(h, DAY); ops.kv_set_seed(h, 3); ops.init_kv_variable_v2(h, t)

    class Shard(object):
      def sparse_read_with_counts(self, ids, counts=None):
        return ops.kv_variable_gather_or_insert_with_counts(var, ids, counts) if counts is not None \
            else ops.kv_variable_gather_or_insert_v2(var, ids)

      def sparse_read_pairs(self, pairs):
        return ops.kv_variable_gather_or_insert_pairs(var, pairs)

      def apply(self, g, ids):
        ops.kv_variable_group_sparse_apply_adam_v4(var, slot, g, ids, 0.1, 0.9, 0.999, 0.9, 0.999, 1e-8, 0, 0, 0)

    sh = sharded.ShardedKvVariable(Shard(), bucket_fn=lambda i, w, nd=None, c=None: ops.kv_bucket_by_owner(var, i, w, nd, c, with_payload=nd is not None),
                                   unique_fn=lambda i, c: ops.kv_unique(var, i, c),
                                   segsum_fn=lambda i, g: ops.kv_dedup_segment_sum(var, i, g),
                                   take_fn=ops.kv_take_rows,
                                    index_sum_fn=lambda g, i, n: ops.kv_unsorted_segment_sum(var, g, i, n),
                                    unique_async_fn=lambda i, c: ops.kv_unique(var, i, c, sync=False))
    ref = ko.OracleKv(D, 0, table, day=DAY, picker=1, seed=3)          # the unsharded truth, same on every rank
    rslot = ko.OracleKv(3 * D, 0, np.zeros((4, 3 * D), np.float32), day=DAY)
    for step in range(4):
      batches = [rng.integers(-400, 400, 3000 + 517 * r) for r in range(world)]   # heavy repeats, negative ids
      # one-signed gradients: the summed gradient of a repeated id never cancels to ~epsilon, where
      # Adam's quotient would amplify the order-dependent fp32 rounding of the per-rank partial sums
      sign = rng.choice([-1.0, 1.0], (1, D))
      grads = [(rng.uniform(0.5, 1.5, (b.size, D)) * 1e-2 * sign).astype(np.float32) for b in batches]
      mine = torch.from_numpy(batches[rank]).cuda()
      out = sh.lookup(mine).cpu().numpy()
      want_all = ref.gather_or_insert(np.concatenate(batches))
      off = sum(b.size for b in batches[:rank])
      want = want_all[off:off + mine.numel()]
      if step == 0:
        np.testing.assert_array_equal(out, want)                        # rows are copies
      else:
        np.testing.assert_allclose(out, want, rtol=2e-5, atol=2e-6)     # per-rank partial sums: fp32 order
      sh.apply_gradients(lambda shard, g, i: shard.apply(g, i), torch.from_numpy(grads[rank]).cuda(), mine)
      u, s, _ = ko.dedup_segment_sum(np.concatenate(batches), np.concatenate(grads))
      ko.apply_group_adam(ref, rslot, s, u, 0.1, 0.9, 0.999, 0.9, 0.999, 1e-8)
      keys, vals = ops.read_kv_variable_op_v2(var)
      keys = keys.cpu().numpy()
      assert keys.size and np.all(np.mod(keys, world) == rank)          # ownership: floor-mod, negatives included
      mine_ref = {k: v for k, v in ref.as_dict().items() if k % world == rank}
      assert set(keys.tolist()) == set(mine_ref)
      got = dict(zip(keys.tolist(), vals.cpu().numpy()))
      for k in mine_ref:
        np.testing.assert_allclose(got[k], mine_ref[k], rtol=2e-5, atol=2e-6)
      tot = torch.tensor([ops.kv_variable_frequency(var)])
      dist.all_reduce(tot)
      assert int(tot) == ref.sum_freq()                                 # every occurrence counted exactly once
    q.put((rank, "ok"))
  except Exception:  # pragma: no cover
    import traceback
    q.put((rank, traceback.format_exc()))
  finally:
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,D", [(2, 16), (4, 64)])      # (4, 64): configs[3]'s shape — dim 64, more than two owners
def test_sharded_world2_on_one_gpu(world, D):
  if not torch.cuda.is_available():
    pytest.skip("needs a GPU")
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  port = s.getsockname()[1]
  s.close()
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  procs = [ctx.Process(target=_worker, args=(r, world, port, q, D)) for r in range(world)]
  for p in procs:
    p.start()
  res = [q.get(timeout=300) for _ in procs]
  for p in procs:
    p.join(timeout=60)
  assert all(r[1] == "ok" for r in res), res
