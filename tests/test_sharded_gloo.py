"""The N > 1 path on CPU: world_size 2, gloo.  The exchange logic (ownership, bucketing,
all_to_all, un-permute, routing reuse) is the product code; the rank-local shard is the CPU oracle
standing in for the HBM table (test infrastructure), and the check is against ONE unsharded oracle
table fed the same ids."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DAY, D = 20000, 8


class OracleShard(object):
  """KvVariable look-alike over the oracle (sparse_read_with_counts + a GroupAdam apply)."""

  def __init__(self, table):
    from oracle import kv_oracle as ko
    self.ko = ko
    self.var = ko.OracleKv(D, 0, table, day=DAY, picker=1, seed=3)
    self.slot = ko.OracleKv(3 * D, 0, np.zeros((4, 3 * D), np.float32), day=DAY)

  def sparse_read_with_counts(self, ids, counts=None):
    c = None if counts is None else counts.numpy().astype(np.int32)
    return torch.from_numpy(self.var.gather_or_insert(ids.numpy(), c))

  def apply(self, grad, ids):
    u, s, _ = self.ko.dedup_segment_sum(ids.numpy(), grad.numpy())
    self.ko.apply_group_adam(self.var, self.slot, s, u, 0.1, 0.9, 0.999, 0.9, 0.999, 1e-8)


def _worker(rank, world, port, q, rule):
  sys.path.insert(0, ROOT)
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  try:
    from tfplus_amd.kv_variable.python.ops import sharded
    rng = np.random.default_rng(11)
    table = rng.standard_normal((32, D)).astype(np.float32)
    sh = sharded.ShardedKvVariable(OracleShard(table), owner_rule=rule)
    own_of = lambda k: int(sharded.owner_of(torch.tensor([int(k)]), world, rule))
    ref = OracleShard(table)                                   # the unsharded truth, same on every rank
    for step in range(3):
      batches = [torch.from_numpy(rng.integers(-50, 200, 64 + 13 * r)) for r in range(world)]
      grads = [torch.from_numpy(rng.standard_normal((b.numel(), D)).astype(np.float32)) for b in batches]
      mine = batches[rank]
      mine2d = mine.reshape(-1, 1)
      out = sh.lookup(mine2d).reshape(-1, D)                   # 2-D ids keep their shape
      want_all = ref.sparse_read_with_counts(torch.cat(batches))
      off = sum(b.numel() for b in batches[:rank])
      # bit-equal on the first step (rows are copies); later steps carry the optimizer state, whose
      # gradient sums were formed per rank first (dedup before the exchange) -> fp32 order differs
      want = want_all[off:off + mine.numel()]
      if step == 0:
        assert torch.equal(out, want), "lookup rows differ from the unsharded table"
      else:
        torch.testing.assert_close(out, want, rtol=1e-5, atol=1e-6)
      # even steps hand the apply the very tensor the lookup saw (bucket sizes reused, no size
      # exchange); odd steps a fresh view (sizes exchanged again)
      sh.apply_gradients(lambda shard, g, i: shard.apply(g, i), grads[rank],
                         mine2d if step % 2 == 0 else mine.reshape(-1, 1))
      ref.apply(torch.cat(grads), torch.cat(batches))
      # ownership: this rank's shard holds exactly the keys with floor_mod(key, world) == rank
      keys, vals, *_ = sh.shard.var.export(2)
      assert all(own_of(k) == rank for k in keys)
      rk, rv, *_ = ref.var.export(2)
      own = {int(k): v for k, v in zip(rk, rv) if own_of(k) == rank}
      got = {int(k): v for k, v in zip(keys, vals)}
      assert set(own) == set(got)
      for k in own:
        np.testing.assert_allclose(got[k], own[k], rtol=1e-5, atol=1e-6)
      cnt = torch.tensor([sh.shard.var.sum_freq()])
      dist.all_reduce(cnt)
      assert int(cnt) == ref.var.sum_freq()                    # frequency words add up across shards
    # routing primitives
    ids = torch.tensor([-3, 4, 7, -8, 5])
    assert sharded.owner_of(ids, 2, "mod").tolist() == [1, 0, 1, 0, 1]
    # the hashed rule is the high half of the library's mix64(id) (kv_device.h) modulo the world, spelled out in Python
    def mix(x):
      x &= (1 << 64) - 1
      x ^= x >> 33; x = x * 0xff51afd7ed558ccd & ((1 << 64) - 1)
      x ^= x >> 33; x = x * 0xc4ceb9fe1a85ec53 & ((1 << 64) - 1)
      return x ^ (x >> 33)
    probe = torch.tensor([-3, 4, 7, -8, 5, 0, -1, 2**62, -2**63, 2**63 - 1])
    for w in (2, 3, 7, 8):
      assert sharded.owner_of(probe, w).tolist() == [(mix(int(v)) >> 32) % w for v in probe.tolist()]
    rt = sharded.route(ids, rule=rule)
    assert sorted(rt.perm.tolist()) == list(range(5)) and sum(rt.send_counts) == 5
    # occurrence counts survive the dedup-before-exchange: lookup with per-id counts
    extra = sh.lookup(torch.tensor([1000 + rank, 1000 + rank, 1001]), counts=torch.tensor([2, 3, 4], dtype=torch.int32))
    assert extra.shape == (3, D) and torch.equal(extra[0], extra[1])
    back = sharded.exchange(rt, sharded.exchange(rt, ids.reshape(-1, 1)), reverse=True)
    assert torch.equal(back.reshape(-1), ids)                  # exchange then reverse is the identity
    q.put((rank, "ok"))
  except Exception as e:  # pragma: no cover
    import traceback
    q.put((rank, traceback.format_exc()))
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize("rule", ["hash", "mod"])
def test_sharded_lookup_and_apply_world2(rule):
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  port = s.getsockname()[1]
  s.close()
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  procs = [ctx.Process(target=_worker, args=(r, 2, port, q, rule)) for r in range(2)]
  for p in procs:
    p.start()
  res = [q.get(timeout=240) for _ in procs]
  for p in procs:
    p.join(timeout=60)
  assert all(r[1] == "ok" for r in res), res
