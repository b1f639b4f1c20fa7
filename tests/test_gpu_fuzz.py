"""Randomised differential test: a random program of KvVariable ops runs on the GPU table and on the
oracle side by side; after every op the observable state must agree: key set, frequency words, sizes
exactly; rows and optimizer state to north_star's 1e-6 relative.  The optimizer ops get what the reference's
ops get — unique ids and TF-core's occurrence-ordered segment sums (variable_scope.py:1096-1106 runs
tf.unique + unsorted_segment_sum in front of them) — so no summation-order allowance is needed here; the
fused segment reduction of repeated ids is bounded per element in test_gpu_config_sizes.py,
test_gpu_parity.py::test_full_size_batch_properties and test_large_batches_match_oracle below."""
RTOL, ATOL = 1e-6, 1e-7      # ATOL: FTRL rebuilds the row from a linear slot that is a difference of O(1) terms
import numpy as np
import pytest

from oracle import kv_oracle as ko

torch = pytest.importorskip("torch")
DAY0 = 20000


@pytest.fixture(scope="module")
def ops():
  if not torch.cuda.is_available():
    pytest.skip("needs a GPU")
  from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as g
  return g


class Pair(object):
  def __init__(self, ops, D, thr, table, seed, key_dtype=None):
    self.ops, self.D = ops, D
    self.h = ops.kv_variable([D], enter_threshold=thr, capacity_hint=64,     # tiny hint: growth + rebuilds happen
                             key_dtype=key_dtype or torch.int64)
    ops.kv_set_seed(self.h, seed); ops.init_kv_variable_v2(self.h, table)
    self.o = ko.OracleKv(D, thr, table, day=DAY0, picker=1, seed=seed)
    self.set_day(DAY0)

  def set_day(self, day):
    self.ops.kv_set_clock_days(self.h, day); self.o.set_day(day)

  def check(self, rng, keyspace, tag):
    ops, h, o = self.ops, self.h, self.o
    assert ops.kv_variable_shape_v2(h)[0] == o.map_size(), tag
    assert ops.kv_variable_size_v2(h) == o.size() and ops.kv_variable_frequency(h) == o.sum_freq(), tag
    q = rng.integers(-keyspace - 5, keyspace + 5, 200)
    np.testing.assert_allclose(ops.kv_variable_gather_or_zeros_v2(h, q).cpu().numpy(), o.gather_or_zeros(q),
                               rtol=RTOL, atol=ATOL, err_msg=tag)
    np.testing.assert_array_equal(ops.kv_variable_get_count_v2(h, q).cpu().numpy(), o.get_count(q), err_msg=tag)
    np.testing.assert_array_equal(ops.kv_variable_get_time_stamp(h, q).cpu().numpy(), o.get_timestamp(q), err_msg=tag)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8))
def test_random_program_matches_oracle(ops, seed):
  _random_program(ops, seed, False)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(100, 108))
def test_random_program_occurrence_order(ops, seed):
  """the same programs on a var in occurrence-order mode (kv_set_deterministic(h, 2)), and the optimizer ops get the RAW ids
  and gradient rows — repeats and all — while the oracle gets TF-core's unique + segment sum of them: same 1e-6, and
  kv_dedup_segment_sum's sums are the oracle's bit for bit"""
  _random_program(ops, seed, True)


def _random_program(ops, seed, occ):
  rng = np.random.default_rng(1000 + seed)
  kd = torch.int32 if seed % 4 == 3 else torch.int64    # int32 keys: same values, the narrow id path of every kernel
  D = int(rng.choice([4, 8, 20, 32, 64]))
  thr = int(rng.choice([0, 0, 2]))
  opt = ["adam", "adagrad", "ftrl"][seed % 3]
  keyspace = int(rng.choice([50, 400, 5000]))
  table = rng.standard_normal((32, D)).astype(np.float32)
  var = Pair(ops, D, thr, table, seed, kd)
  slots = {"adam": [(3 * D, 0.0)], "adagrad": [(D, 0.1)], "ftrl": [(D, 0.1), (D, 0.0)]}[opt]
  sl = [Pair(ops, d, 0, np.full((4, d), v, np.float32), seed, kd) for d, v in slots]
  if occ:
    ops.kv_set_deterministic(var.h, ops.KV_ORDER_OCCURRENCE)
  day = DAY0
  b1p, b2p = np.float32(0.9), np.float32(0.999)
  for step in range(40):
    op = rng.choice(["lookup", "lookup_counts", "apply", "apply", "scatter", "insert", "delete", "expire", "roundtrip", "sparse"],
                    p=[.15, .1, .2, .15, .1, .05, .1, .05, .05, .05])
    n = int(rng.choice([1, 7, 300, 3000]))
    ids = rng.integers(-keyspace, keyspace, n)
    if op == "roundtrip" and kd == torch.int32:
      op = "lookup"                                       # kv_import is int64-only
    tag = "seed %d step %d %s n=%d D=%d" % (seed, step, op, n, D)
    if op == "lookup":
      got = ops.kv_variable_gather_or_insert_v2(var.h, ids).cpu().numpy()
      np.testing.assert_allclose(got, var.o.gather_or_insert(ids), rtol=RTOL, atol=ATOL, err_msg=tag)
    elif op == "lookup_counts":
      c = rng.integers(1, 40000, n).astype(np.int32)
      got = ops.kv_variable_gather_or_insert_with_counts(var.h, ids, c).cpu().numpy()
      np.testing.assert_allclose(got, var.o.gather_or_insert(ids, c), rtol=RTOL, atol=ATOL, err_msg=tag)
    elif op == "apply":
      g = (rng.uniform(0.5, 1.5, (n, D)) * 1e-2 * rng.choice([-1.0, 1.0], (1, D))).astype(np.float32)
      u, s, _ = ko.dedup_segment_sum(ids, g)
      # TF-core's dedup on the GPU (kv_dedup_segment_sum / kv_unique): the same keys, sums up to the order of the additions
      gu, gs, ginv = ops.kv_dedup_segment_sum(var.h, ids, g)
      gu, gs, ginv = gu.cpu().numpy(), gs.cpu().numpy(), ginv.cpu().numpy()
      assert np.array_equal(gu[ginv], ids) and np.array_equal(np.sort(gu), np.sort(u)), tag
      if occ:
        assert np.array_equal(gs[np.argsort(gu)].view(np.uint32), s[np.argsort(u)].view(np.uint32)), tag   # TF-core's chain
      else:
        np.testing.assert_allclose(gs[np.argsort(gu)], s[np.argsort(u)], rtol=1e-5, atol=1e-8, err_msg=tag)
      qu, qc, qinv = ops.kv_unique(var.h, ids)
      assert np.array_equal(qu.cpu().numpy()[qinv.cpu().numpy()], ids) and int(qc.sum()) == n, tag
      uq = dict(unique_indices=bool(rng.integers(0, 2)))   # the op as an unchanged graph calls it: ids promised unique (one launch)
      gi, gg = (ids, g) if occ else (u, s)                 # occurrence order: the op sums the repeats itself, in TF-core's order
      if occ:
        uq = {}
      if opt == "adam":
        ops.kv_variable_group_sparse_apply_adam_v4(var.h, sl[0].h, gg, gi, 1e-2, b1p, b2p, 0.9, 0.999, 1e-8, 0, 0, 0, **uq)
        ko.apply_group_adam(var.o, sl[0].o, s, u, 1e-2, b1p, b2p, 0.9, 0.999, 1e-8)
        b1p, b2p = np.float32(b1p * np.float32(0.9)), np.float32(b2p * np.float32(0.999))
      elif opt == "adagrad":
        ops.kv_variable_sparse_apply_adagrad(var.h, sl[0].h, 0.05, gg, gi, use_locking=True, **uq)
        ko.apply_adagrad(var.o, sl[0].o, 0.05, s, u)
      else:
        ops.kv_variable_sparse_group_sparse_apply_ftrl_v2(var.h, sl[0].h, sl[1].h, gg, gi, 0.05, 0.0, 1e-3, 0.0, 0.0, -0.5, **uq)
        ko.apply_sparse_group_ftrl(var.o, sl[0].o, sl[1].o, s, u, 0.05, 0.0, 1e-3, 0.0, 0.0, -0.5)
    elif op == "sparse":   # embedding_lookup_sparse in one call: unique -> GatherOrInsert -> segment sum in position order
      nseg = max(1, n // 3)
      seg = np.sort(rng.integers(0, nseg, n))
      got = ops.kv_variable_lookup_sparse(var.h, ids, seg, None, nseg, "sum", count_occurrences=False).cpu().numpy()
      uniq, idx = np.unique(ids, return_inverse=True)
      emb = var.o.gather_or_insert(uniq)[idx]
      want = np.zeros((nseg, D), np.float32)
      for j in range(n):
        want[seg[j]] = want[seg[j]] + emb[j]
      np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-7, err_msg=tag)
    elif op == "scatter":
      uids = np.unique(ids)
      upd = rng.uniform(0.5, 2.0, (uids.size, D)).astype(np.float32)
      which = int(rng.integers(0, 7))
      fn = [ops.kv_variable_scatter_update_v2, ops.kv_variable_scatter_add_v2, ops.kv_variable_scatter_sub_v2,
            ops.kv_variable_scatter_mul_v2, ops.kv_variable_scatter_div_v2, ops.kv_variable_scatter_min_v2,
            ops.kv_variable_scatter_max_v2][which]
      fn(var.h, uids, upd); var.o.scatter_update(uids, upd, which)
    elif op == "insert":
      uids = np.unique(ids)
      vals = rng.standard_normal((uids.size, D)).astype(np.float32)
      ops.kv_variable_insert_v2(var.h, uids, vals); var.o.insert(uids, vals)
    elif op == "delete":
      for p in [var] + sl:
        assert ops.kv_variable_delete(p.h, ids) == p.o.delete(ids), tag
    elif op == "expire":
      day += int(rng.integers(1, 5))
      for p in [var] + sl:
        p.set_day(day)
      thr_days = int(rng.integers(2, 8))
      assert sorted(ops.kv_variable_delete_with_timestamp(var.h, thr_days).cpu().numpy().tolist()) == \
          sorted(var.o.delete_with_timestamp(thr_days).tolist()), tag
    else:   # export -> import into the same table (clear + reload)
      k, v, bl, fk, fv = ops.kv_variable_export(var.h, first_n=6)
      ok, ov, obl, ofk, ofv = var.o.export(6)
      assert sorted(k.cpu().numpy().tolist()) == sorted(ok.tolist()) and sorted(bl.cpu().numpy().tolist()) == sorted(obl.tolist()), tag
      ops.kv_variable_import(var.h, k, v, bl, fk, fv)
      var.o.import_(ok, ov, obl, ofk, ofv)
    var.check(rng, keyspace, tag)
    for i, p in enumerate(sl):
      p.check(rng, keyspace, tag + " slot%d" % i)


@pytest.mark.gpu
@pytest.mark.parametrize("n,keyspace,D", [(150_000, 2_000, 8), (150_000, 10_000_000, 64), (700_000, 600_000, 4)])
def test_large_batches_match_oracle(ops, n, keyspace, D):
  """Multi-tile batches: keys present in every tile (block-wide folds), and ~700 distinct keys per
  partition (LDS key tables overflow and the partition splits into sub-hash classes)."""
  rng = np.random.default_rng(n + D)
  table = rng.standard_normal((32, D)).astype(np.float32)
  var = Pair(ops, D, 0, table, 5)
  slot = Pair(ops, 3 * D, 0, np.zeros((4, 3 * D), np.float32), 5)
  b1p, b2p = np.float32(0.9), np.float32(0.999)
  for step in range(3):
    ids = rng.integers(-keyspace, keyspace, n)
    got = ops.kv_variable_gather_or_insert_v2(var.h, ids).cpu().numpy()
    # ids repeat up to ~n / keyspace x 2048 times and the fused reduce sums them in another order than TF-core:
    # one-signed gradients keep |sum| ~ sum|g|, where (count - 1) 2^-24 relative on the sum moves an Adam step by
    # far less than 1e-6 of the row; the allowance below is for the sum itself re-entering m and v
    np.testing.assert_allclose(got, var.o.gather_or_insert(ids), rtol=1e-5, atol=1e-6)
    g = (rng.uniform(0.5, 1.5, (n, D)) * 1e-2 * rng.choice([-1.0, 1.0], (1, D))).astype(np.float32)
    ops.kv_variable_group_sparse_apply_adam_v4(var.h, slot.h, g, ids, 1e-2, b1p, b2p, 0.9, 0.999, 1e-8, 0, 0, 0)
    u, s, _ = ko.dedup_segment_sum(ids, g)
    ko.apply_group_adam(var.o, slot.o, s, u, 1e-2, b1p, b2p, 0.9, 0.999, 1e-8)
    b1p, b2p = np.float32(b1p * np.float32(0.9)), np.float32(b2p * np.float32(0.999))
    kill = rng.integers(-keyspace, keyspace, n // 10)
    assert ops.kv_variable_delete(var.h, kill) == var.o.delete(kill)
    var.check(rng, keyspace, "large step %d" % step)
    slot.check(rng, keyspace, "large slot step %d" % step)
