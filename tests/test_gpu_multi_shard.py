"""Sharded lookup + apply of SEVERAL tables (kvhip.h kv_multi_shard_lookup / kv_multi_shard_apply): the tables of one model
share two grouped exchanges per lookup and one per apply.  (a) the whole ops on a world of one through RCCL's grouped
send / recv, bit-identical to the per-table sharded ops; (b) world 8 x 40 tables in ONE process on one device, phase by
phase with the in-process exchange, against one unsharded oracle table per embedding table."""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DAY = 19000
HP = (0.1, 0.9, 0.999, 0.9, 0.999, 1e-8, 0, 0, 0)


def _table(ops, D, init, det):
  h = ops.kv_variable([D])
  ops.kv_set_clock_days(h, DAY); ops.kv_set_seed(h, 3); ops.init_kv_variable_v2(h, init)
  if det:
    ops.kv_set_deterministic(h, True)
  return h


@pytest.mark.gpu
def test_multi_shard_ops_equal_the_per_table_sharded_ops():
  if not torch.cuda.is_available():
    pytest.skip("needs a GPU")
  sys.path.insert(0, ROOT)
  from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as ops
  os.environ["KV_COMM_SELF_VIA_RCCL"] = "1"
  rng = np.random.default_rng(8)
  dims = [16, 32, 8, 64, 32]
  T = len(dims)
  sets = []
  for which in range(2):   # 0: driven by the multi ops, 1: by the per-table ops
    vs, ss, shs = [], [], []
    for k, D in enumerate(dims):
      init = np.random.default_rng(100 + k).standard_normal((32, D)).astype(np.float32)
      v = _table(ops, D, init, True); s = _table(ops, 3 * D, np.zeros((4, 3 * D), np.float32), True)
      vs.append(v); ss.append(s); shs.append(ops.KvShard(v, 1, 0, ops.KV_OWNER_HASH, max_ids=1 << 14))
    sets.append((vs, ss, shs))
  comm = ops.KvComm(1, 0, ops.kv_comm_unique_id())
  for step in range(3):
    ids = [torch.from_numpy(rng.integers(0, 3000, 5000 + 777 * k)).cuda() for k in range(T)]
    if step == 2:
      ids[1] = ids[1][:0]                                                  # a table nobody looks up this step
    grads = [torch.from_numpy((rng.standard_normal((i.numel(), D)) * 1e-2).astype(np.float32)).cuda() for i, D in zip(ids, dims)]
    outs = ops.kv_multi_shard_lookup(sets[0][2], comm, ids, join=(step != 1))
    if step == 1:
      sets[0][2][0].join()
    want = [sets[1][2][k].lookup(comm, ids[k]) for k in range(T)]
    for k in range(T):
      assert torch.equal(outs[k], want[k]), (step, k)
    ops.kv_multi_shard_apply(sets[0][2], comm, ops.OPT_GROUP_ADAM_V4, [[s] for s in sets[0][1]], grads, HP)
    for k in range(T):
      sets[1][2][k].apply(comm, ops.OPT_GROUP_ADAM_V4, [sets[1][1][k]], grads[k], HP)
  torch.cuda.synchronize()
  for k in range(T):
    for a, b in ((sets[0][0][k], sets[1][0][k]), (sets[0][1][k], sets[1][1][k])):
      ka, va = ops.read_kv_variable_op_v2(a); kb, vb = ops.read_kv_variable_op_v2(b)
      oa, ob = torch.argsort(ka), torch.argsort(kb)
      assert torch.equal(ka[oa], kb[ob]) and torch.equal(va[oa], vb[ob]), k   # deterministic mode: bit for bit
  # a shard listed twice, a communicator of another world: refused before anything is queued
  with pytest.raises(Exception, match="listed twice"):
    ops.kv_multi_shard_lookup([sets[0][2][0], sets[0][2][0]], comm, [ids[0], ids[0]])
  del comm


@pytest.mark.gpu
@pytest.mark.parametrize("opt", ["adam", "ftrl"])
def test_multi_shard_batched_phases_equal_the_per_table_ops(opt):
  """The default (not deterministic) mode: kv_multi_shard_lookup / _apply run every phase of same-shaped tables in one
  launch (route, owner lookup, finish, gradient pre-sum, owner apply: kvhip.hip multi_*_impl).  Against the per-table
  sharded ops on twin tables: the rows a lookup returns are bit-equal, the tables hold the same keys and — the fp32
  sums of repeated ids being taken in another order — the same values within the summation tolerance.  One table is
  read between its lookup and its apply (its pending pass is settled: it leaves the batched apply for the per-table
  one), one sits out a step, dims repeat and do not."""
  if not torch.cuda.is_available():
    pytest.skip("needs a GPU")
  sys.path.insert(0, ROOT)
  from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as ops
  rng = np.random.default_rng(18)
  dims = [16, 32, 32, 8, 32, 16]
  T = len(dims)
  sets = []
  for which in range(2):   # 0: driven by the multi ops, 1: by the per-table ops
    vs, ss, shs = [], [], []
    for k, D in enumerate(dims):
      init = np.random.default_rng(200 + k).standard_normal((32, D)).astype(np.float32)
      v = _table(ops, D, init, False)
      if opt == "adam":
        sl = [_table(ops, 3 * D, np.zeros((4, 3 * D), np.float32), False)]
      else:
        sl = [_table(ops, D, np.full((4, D), 0.1, np.float32), False), _table(ops, D, np.zeros((4, D), np.float32), False)]
      vs.append(v); ss.append(sl); shs.append(ops.KvShard(v, 1, 0, ops.KV_OWNER_HASH, max_ids=1 << 15))
    sets.append((vs, ss, shs))
  comm = ops.KvComm(1, 0, None)
  code = ops.OPT_GROUP_ADAM_V4 if opt == "adam" else ops.OPT_SPARSE_GROUP_FTRL
  hp = HP if opt == "adam" else (0.1, 0, 0, 0, 0, -0.5)
  for step in range(4):
    ids = [torch.from_numpy(rng.integers(0, 6000, 9000 + 1111 * k)).cuda() for k in range(T)]
    if step == 2:
      ids[2] = ids[2][:0]
    sign = [rng.choice([-1.0, 1.0], (1, D)) for D in dims]   # one sign per element: sums of repeated ids do not cancel
    grads = [torch.from_numpy((rng.uniform(0.5, 1.5, (i.numel(), D)) * 1e-2 * sg).astype(np.float32)).cuda()
             for i, D, sg in zip(ids, dims, sign)]
    outs = ops.kv_multi_shard_lookup(sets[0][2], comm, ids)
    want = [sets[1][2][k].lookup(comm, ids[k]) for k in range(T)]
    for k in range(T):
      if step == 0:
        assert torch.equal(outs[k], want[k]), (step, k)
      else:
        torch.testing.assert_close(outs[k], want[k], rtol=2e-5, atol=1e-6)
    if step == 1:   # another op on a table between its lookup and its apply
      ops.kv_variable_gather_or_zeros_v2(sets[0][0][4], ids[4][:100])
    ops.kv_multi_shard_apply(sets[0][2], comm, code, sets[0][1], grads, hp)
    for k in range(T):
      sets[1][2][k].apply(comm, code, sets[1][1][k], grads[k], hp)
  torch.cuda.synchronize()
  for k in range(T):
    pairs = [(sets[0][0][k], sets[1][0][k])] + list(zip(sets[0][1][k], sets[1][1][k]))
    for a, b in pairs:
      ka, va = ops.read_kv_variable_op_v2(a); kb, vb = ops.read_kv_variable_op_v2(b)
      oa, ob = torch.argsort(ka), torch.argsort(kb)
      assert torch.equal(ka[oa], kb[ob]), k
      torch.testing.assert_close(va[oa], vb[ob], rtol=2e-5, atol=1e-6)
  del comm


@pytest.mark.gpu
def test_multi_shard_lossless_tables_grow_together():
  """Lossless tables in a multi-table step: ONE agreement (an all-reduce per table in one group, one synchronisation)
  raises the capacity of exactly the tables that need it; rows and state equal the unsharded ops."""
  if not torch.cuda.is_available():
    pytest.skip("needs a GPU")
  sys.path.insert(0, ROOT)
  from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as ops
  os.environ["KV_COMM_SELF_VIA_RCCL"] = "1"
  rng = np.random.default_rng(12)
  D, T = 16, 3
  vs, ss, shs, v2, s2 = [], [], [], [], []
  for k in range(T):
    init = np.random.default_rng(300 + k).standard_normal((32, D)).astype(np.float32)
    v = _table(ops, D, init, False); s = _table(ops, 3 * D, np.zeros((4, 3 * D), np.float32), False)
    vs.append(v); ss.append(s)
    sh = ops.KvShard(v, 1, 0, ops.KV_OWNER_HASH, max_ids=1 << 14, peer_capacity=4096 if k == 1 else 16)
    sh.set_lossless(k != 1)                                                # table 1: default mode, roomy capacity
    shs.append(sh)
    v2.append(_table(ops, D, init, False)); s2.append(_table(ops, 3 * D, np.zeros((4, 3 * D), np.float32), False))
  comm = ops.KvComm(1, 0, ops.kv_comm_unique_id())
  for step in range(2):
    ids = [torch.from_numpy(rng.choice(50000, 2500 + 500 * k, replace=False)).cuda() for k in range(T)]
    grads = [torch.from_numpy((rng.uniform(0.5, 1.5, (i.numel(), D)) * 1e-2).astype(np.float32)).cuda() for i in ids]
    outs = ops.kv_multi_shard_lookup(shs, comm, ids)
    for k in range(T):
      assert torch.equal(outs[k], ops.kv_variable_gather_or_insert_v2(v2[k], ids[k])), (step, k)
    assert shs[0].peer_capacity >= 2500 and shs[2].peer_capacity >= 3500 and shs[1].peer_capacity == 4096
    ops.kv_multi_shard_apply(shs, comm, ops.OPT_GROUP_ADAM_V4, [[s] for s in ss], grads, HP)
    for k in range(T):
      ops.kv_variable_group_sparse_apply_adam_v4(v2[k], s2[k], grads[k], ids[k], *HP)
  torch.cuda.synchronize()
  for k in range(T):
    ka, va = ops.read_kv_variable_op_v2(vs[k]); kb, vb = ops.read_kv_variable_op_v2(v2[k])
    oa, ob = torch.argsort(ka), torch.argsort(kb)
    assert torch.equal(ka[oa], kb[ob])
    torch.testing.assert_close(va[oa], vb[ob], rtol=1e-6, atol=1e-7)      # distinct ids: nothing to reorder
  del comm


@pytest.mark.gpu
def test_world_of_eight_forty_tables_in_one_process_match_unsharded_oracles():
  if not torch.cuda.is_available():
    pytest.skip("needs a GPU")
  sys.path.insert(0, ROOT)
  from oracle import kv_oracle as ko
  from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as ops, sharded
  from _reorder import adam_hp, adam_reorder_check
  world, T, D = 8, 40, 8
  rng = np.random.default_rng(21)
  inits = [np.random.default_rng(500 + k).standard_normal((16, D)).astype(np.float32) for k in range(T)]
  vars_ = [[_table(ops, D, inits[k], False) for r in range(world)] for k in range(T)]
  slots = [[_table(ops, 3 * D, np.zeros((4, 3 * D), np.float32), False) for r in range(world)] for k in range(T)]
  shards = [[ops.KvShard(vars_[k][r], world, r, ops.KV_OWNER_HASH, max_ids=2048) for r in range(world)] for k in range(T)]
  refs = [ko.OracleKv(D, 0, inits[k], day=DAY, picker=1, seed=3) for k in range(T)]
  rslots = [ko.OracleKv(3 * D, 0, np.zeros((4, 3 * D), np.float32), day=DAY) for k in range(T)]
  b1p, b2p = np.float32(0.9), np.float32(0.999)
  hp = adam_hp(0.1, float(b1p), float(b2p))
  for step in range(2):
    batches = [[rng.integers(0, 300 + 40 * k, 200 + 13 * r + k) for r in range(world)] for k in range(T)]
    grads = [[(rng.standard_normal((b.size, D)) * 1e-2).astype(np.float32) for b in batches[k]] for k in range(T)]
    # every table's route phase, then every table's exchange (a multi-table step queues them in this order), ...
    for k in range(T):
      for r in range(world):
        shards[k][r].lookup_route(torch.from_numpy(batches[k][r]).cuda())
    for k in range(T):
      ops.kv_shard_exchange_local(shards[k], 0)
    for k in range(T):
      for r in range(world):
        shards[k][r].lookup_serve()
    for k in range(T):
      ops.kv_shard_exchange_local(shards[k], 1)
    for k in range(T):
      allb = np.concatenate(batches[k])
      want_all = refs[k].gather_or_insert(allb)
      # the rows as the owners hold them now (the state itself is held to the oracle by the reorder bound below)
      u = np.unique(allb)
      own = sharded.owner_of(torch.from_numpy(u), world, "hash").numpy()
      held = np.zeros((u.size, D), np.float32)
      for r in range(world):
        held[own == r] = ops.kv_variable_gather_or_zeros_v2(vars_[k][r], torch.from_numpy(u[own == r]).cuda()).cpu().numpy()
      off = 0
      for r in range(world):
        got = shards[k][r].lookup_finish().cpu().numpy()
        want = want_all[off:off + batches[k][r].size]
        off += batches[k][r].size
        if step == 0:
          np.testing.assert_array_equal(got, want)                          # rows are copies of the init rule's rows
        np.testing.assert_array_equal(got, held[np.searchsorted(u, batches[k][r])])   # ... and of the owner's rows
    for k in range(T):
      for r in range(world):
        shards[k][r].apply_route(torch.from_numpy(grads[k][r]).cuda())
    for k in range(T):
      ops.kv_shard_exchange_local(shards[k], 1)
    for k in range(T):
      allb, allg = np.concatenate(batches[k]), np.concatenate(grads[k])
      u = np.unique(allb)
      own = sharded.owner_of(torch.from_numpy(u), world, "hash").numpy()
      # each owner's state before the step (its own fp32 state), for the per-element reorder bound
      pre = {}
      for r in range(world):
        ur = torch.from_numpy(u[own == r]).cuda()
        st = ops.kv_variable_gather_or_zeros_v2(slots[k][r], ur).cpu().numpy()
        pre[r] = (ops.kv_variable_gather_or_zeros_v2(vars_[k][r], ur).cpu().numpy(), st[:, :D], st[:, D:2 * D], st[:, 2 * D:])
      for r in range(world):
        shards[k][r].apply_serve(ops.OPT_GROUP_ADAM_V4, [slots[k][r]], (0.1, b1p, b2p, 0.9, 0.999, 1e-8, 0, 0, 0))
      us, s, _ = ko.dedup_segment_sum(allb, allg)
      ko.apply_group_adam(refs[k], rslots[k], s, us, 0.1, float(b1p), float(b2p), 0.9, 0.999, 1e-8)
      total = 0
      for r in range(world):
        keys, _ = ops.read_kv_variable_op_v2(vars_[k][r])
        keys = keys.cpu().numpy()
        mine = np.array(sorted(refs[k].as_dict()), np.int64)
        mine = mine[sharded.owner_of(torch.from_numpy(mine), world, "hash").numpy() == r]
        assert set(keys.tolist()) == set(mine.tolist()), (k, r)             # every key on its owner, only there
        total += keys.size
        ur = u[own == r]
        sel = np.isin(allb, ur)
        x1 = ops.kv_variable_gather_or_zeros_v2(vars_[k][r], torch.from_numpy(ur).cuda()).cpu().numpy()
        x0, m0, v0, z0 = pre[r]
        adam_reorder_check(x0, m0, v0, z0, allb[sel], allg[sel], x1, hp, what="table %d rank %d step %d" % (k, r, step))
      assert total == len(refs[k].as_dict())


@pytest.mark.gpu
def test_multi_shard_ops_with_ranks_as_threads_match_unsharded_oracles():
  """kv_multi_shard_lookup / kv_multi_shard_apply — batched phases — at world 3 with the ranks as threads of this
  process over staged communicators (the rank's own segments are NOT delivered by the transport: they are read in
  place): five tables (dims repeat, so the per-dim groups hold several tables), every rank's ids against one unsharded
  oracle table per table."""
  if not torch.cuda.is_available():
    pytest.skip("needs a GPU")
  import threading
  sys.path.insert(0, ROOT)
  from oracle import kv_oracle as ko
  from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as ops, sharded
  world, dims = 3, [16, 32, 16, 8, 32]
  T = len(dims)
  rng = np.random.default_rng(91)
  inits = [np.random.default_rng(300 + k).standard_normal((32, D)).astype(np.float32) for k, D in enumerate(dims)]
  vs = [[_table(ops, D, inits[k], False) for k, D in enumerate(dims)] for _ in range(world)]
  ss = [[_table(ops, 3 * D, np.zeros((4, 3 * D), np.float32), False) for D in dims] for _ in range(world)]
  shs = [[ops.KvShard(vs[r][k], world, r, ops.KV_OWNER_HASH, max_ids=1 << 14) for k in range(T)] for r in range(world)]
  refs = [ko.OracleKv(D, 0, inits[k], day=DAY, picker=1, seed=3) for k, D in enumerate(dims)]
  rsl = [ko.OracleKv(3 * D, 0, np.zeros((4, 3 * D), np.float32), day=DAY) for D in dims]
  dev = torch.device("cuda", 0)
  bar = threading.Barrier(world, timeout=120)
  sent, vals = [None] * world, [0] * world

  def make_comm(r):
    def exchange(send, recv, per_peer):
      n = per_peer * world
      sent[r] = ops.KvCommStaged.raw(send, n, dev)
      torch.cuda.synchronize()
      bar.wait()
      dst = ops.KvCommStaged.raw(recv, n, dev)
      for p in range(world):
        if p == r and per_peer != 32:
          dst[p * per_peer:(p + 1) * per_peer].fill_(0x5A)
        else:
          dst[p * per_peer:(p + 1) * per_peer].copy_(sent[p][r * per_peer:(r + 1) * per_peer])
      torch.cuda.synchronize()
      bar.wait()

    def max_u32(v):
      vals[r] = v
      bar.wait()
      m = max(vals)
      bar.wait()
      return m
    return ops.KvCommStaged(0, world=world, rank=r, exchange=exchange, max_u32=max_u32)

  comms = [make_comm(r) for r in range(world)]
  for step in range(3):
    ids = [[rng.integers(-200, 2000, 1500 + 300 * k + 77 * r) for k in range(T)] for r in range(world)]
    if step == 1:
      ids[1][2] = ids[1][2][:0]
    sign = [rng.choice([-1.0, 1.0], (1, D)) for D in dims]
    grads = [[(rng.uniform(0.5, 1.5, (ids[r][k].size, D)) * 1e-2 * sign[k]).astype(np.float32) for k, D in enumerate(dims)]
             for r in range(world)]
    outs, errs = [None] * world, []

    def rank_step(r):
      try:
        torch.cuda.set_device(0)
        o = ops.kv_multi_shard_lookup(shs[r], comms[r], [torch.from_numpy(x).cuda() for x in ids[r]])
        torch.cuda.synchronize()
        outs[r] = [x.cpu().numpy() for x in o]
        ops.kv_multi_shard_apply(shs[r], comms[r], ops.OPT_GROUP_ADAM_V4, [[s] for s in ss[r]],
                                 [torch.from_numpy(g).cuda() for g in grads[r]], HP)
        torch.cuda.synchronize()
      except Exception as e:
        errs.append((r, repr(e)))
        bar.abort()
    ts = [threading.Thread(target=rank_step, args=(r,)) for r in range(world)]
    for t in ts:
      t.start()
    for t in ts:
      t.join()
    assert not errs, errs
    bar.reset()
    for k, D in enumerate(dims):
      allids = np.concatenate([ids[r][k] for r in range(world)])
      want_all = refs[k].gather_or_insert(allids)
      off = 0
      for r in range(world):
        want = want_all[off:off + ids[r][k].size]
        off += ids[r][k].size
        if step == 0:
          np.testing.assert_array_equal(outs[r][k], want)
        else:
          np.testing.assert_allclose(outs[r][k], want, rtol=2e-5, atol=2e-6)
      u, s, _ = ko.dedup_segment_sum(allids, np.concatenate([grads[r][k] for r in range(world)]))
      ko.apply_group_adam(refs[k], rsl[k], s, u, *HP)
  for k in range(T):
    allk = np.array(sorted(refs[k].as_dict()), np.int64)
    own = sharded.owner_of(torch.from_numpy(allk), world, "hash").numpy()
    total = 0
    for r in range(world):
      keys, vals_r = ops.read_kv_variable_op_v2(vs[r][k])
      keys = keys.cpu().numpy()
      assert set(keys.tolist()) == set(allk[own == r].tolist()), (k, r)
      got = dict(zip(keys.tolist(), vals_r.cpu().numpy()))
      for key in keys.tolist():
        np.testing.assert_allclose(got[key], refs[k].as_dict()[key], rtol=2e-5, atol=2e-6)
      total += ops.kv_variable_frequency(vs[r][k])
    assert total == refs[k].sum_freq(), k
  del comms
