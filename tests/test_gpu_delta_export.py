"""DeltaExport (KvVariableFullOrDeltaExport with need_full_export = false, dynamic_save.hpp:198-451): the
train / prediction delta lists kept by the GPU table against the oracle's restatement of the reference
sets, over random programs of lookups, applies (filtered keys, blacklisting), scatters, inserts,
deletes, expiry, full exports and imports.  Key lists: exact (as sets); rows: as in test_gpu_fuzz — 1e-6 relative,
times the amplification of the group-lasso step for a row that one has just written: x = u * (1 - t) / y with t =
l21_norm / ||u|| (training_ops.cc:7166-7195, :713-751), so the rounding of the norm (a sum of squares taken in another order
than the checker's: SURVEY 8c "parity unpinned") reaches the row multiplied by t / (1 - t), without bound as the row nears
the threshold under which it is blacklisted.  The test derives that factor per key from the oracle's own slot rows; a fixed
1e-6 met such a row twice in 50 000 soak programs (tools/README.md: seeds 40039, 165681)."""
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
  def __init__(self, ops, D, thr, table, seed, track=(True, False)):
    self.ops, self.D = ops, D
    self.h = ops.kv_variable([D], enter_threshold=thr, capacity_hint=64)
    ops.kv_set_seed(self.h, seed); ops.init_kv_variable_v2(self.h, table)
    self.o = ko.OracleKv(D, thr, table, day=DAY0, picker=1, seed=seed)
    self.set_day(DAY0)
    if track is not None:
      ops.kv_set_delta_tracking(self.h, *track); self.o.set_delta_tracking(*track)
    self.amp = {}        # key -> amplification of the last optimizer step that wrote its row (1: none)

  def set_day(self, day):
    self.ops.kv_set_clock_days(self.h, day); self.o.set_day(day)

  def compare_delta(self, first_n, tag):
    k, v, bl, fk, fv, need_full, dk = self.ops.kv_variable_full_or_delta_export(self.h, do_full_export=False, first_n=first_n)
    ok, ov, obl, ofk, ofv, odk = self.o.export_delta(first_n)
    assert need_full is False
    k, bl, fk, dk = (x.cpu().numpy() for x in (k, bl, fk, dk))
    fv = fv.cpu().numpy().view(np.uint32)
    assert sorted(k.tolist()) == sorted(ok.tolist()), tag
    assert sorted(bl.tolist()) == sorted(obl.tolist()), tag
    assert sorted(dk.tolist()) == sorted(odk.tolist()), tag
    assert dict(zip(fk.tolist(), fv.tolist())) == dict(zip(ofk.tolist(), ofv.tolist())) and fk.size == ofk.size, tag
    got = dict(zip(k.tolist(), v.cpu().numpy()))
    for key, row in zip(ok.tolist(), ov):
      np.testing.assert_allclose(got[key], row, rtol=1e-6 * self.amp.get(key, 1.0), atol=1e-7, err_msg=tag)
    return k.size + bl.size + dk.size


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
def test_delta_lists_match_oracle(ops, seed):
  rng = np.random.default_rng(7000 + seed)
  D = int(rng.choice([4, 8, 32]))
  thr = int(rng.choice([0, 2]))
  opt = ["adam", "ftrl"][seed % 2]
  pred = bool(seed % 3 == 0)                      # SUPPORT_PREDICTION_DELTA_EXPORT
  keyspace = int(rng.choice([60, 600]))
  table = rng.standard_normal((32, D)).astype(np.float32)
  var = Pair(ops, D, thr, table, seed, (True, pred))
  slots = {"adam": [(3 * D, 0.0)], "ftrl": [(D, 0.1), (D, 0.0)]}[opt]
  sl = [Pair(ops, d, 0, np.full((4, d), v, np.float32), seed, (True, pred)) for d, v in slots]
  day = DAY0
  b1p, b2p = np.float32(0.9), np.float32(0.999)
  exported = 0
  for step in range(60):
    op = rng.choice(["lookup", "apply", "apply_lasso", "scatter", "insert", "delete", "expire", "full", "delta", "import"],
                    p=[.2, .15, .1, .08, .05, .1, .04, .04, .2, .04])
    n = int(rng.choice([1, 9, 200, 2500]))
    ids = rng.integers(-keyspace, keyspace, n)
    tag = "seed %d step %d %s n=%d D=%d thr=%d" % (seed, step, op, n, D, thr)
    if op == "lookup":
      ops.kv_variable_gather_or_insert_v2(var.h, ids); var.o.gather_or_insert(ids)
    elif op in ("apply", "apply_lasso"):
      g = (rng.uniform(0.5, 1.5, (n, D)) * 1e-2 * rng.choice([-1.0, 1.0], (1, D))).astype(np.float32)
      u, s, _ = ko.dedup_segment_sum(ids, g)
      l21 = 10.0 if op == "apply_lasso" else 0.0     # group lasso large enough to blacklist what it touches
      if opt == "adam":
        ops.kv_variable_group_sparse_apply_adam_v4(var.h, sl[0].h, s, u, 1e-2, b1p, b2p, 0.9, 0.999, 1e-8, 0, 0, l21)   # what the reference op gets: unique ids, TF-core's sums
        ko.apply_group_adam(var.o, sl[0].o, s, u, 1e-2, b1p, b2p, 0.9, 0.999, 1e-8, 0.0, 0.0, l21)
        b1p, b2p = np.float32(b1p * np.float32(0.9)), np.float32(b2p * np.float32(0.999))
      else:
        ops.kv_variable_sparse_group_sparse_apply_ftrl_v2(var.h, sl[0].h, sl[1].h, s, u, 0.05, 0.0, 1e-3, l21, 0.0, -0.5)
        ko.apply_sparse_group_ftrl(var.o, sl[0].o, sl[1].o, s, u, 0.05, 0.0, 1e-3, l21, 0.0, -0.5)
      # what the step may have amplified: with l1 = 0 the lasso's u is minus the linear slot's row (Adam: the third block of
      # m | v | z), its norm the one the op compared with l21 * sqrt(D); rows it blacklisted are zeros on both sides
      for key in u.tolist():
        var.amp[key] = 1.0
      if l21 > 0:
        lin = (sl[0].o.gather_or_zeros(u)[:, 2 * D:] if opt == "adam" else sl[1].o.gather_or_zeros(u)).astype(np.float64)
        t = l21 * np.sqrt(float(D)) / np.maximum(np.sqrt((lin * lin).sum(1)), 1e-30)
        for key, tk in zip(u.tolist(), t.tolist()):
          if tk < 1.0:
            var.amp[key] = 1.0 + 2.0 * tk / (1.0 - tk)
    elif op == "scatter":
      uids = np.unique(ids)
      upd = rng.uniform(0.5, 2.0, (uids.size, D)).astype(np.float32)
      ops.kv_variable_scatter_add_v2(var.h, uids, upd); var.o.scatter_update(uids, upd, 1)
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
      thr_days = int(rng.integers(2, 6))
      assert sorted(ops.kv_variable_delete_with_timestamp(var.h, thr_days).cpu().numpy().tolist()) == \
          sorted(var.o.delete_with_timestamp(thr_days).tolist()), tag
    elif op == "full":      # a full export with first_n > 2 ends the current delta period too
      fn = int(rng.choice([2, 3, 6]))
      for p in [var] + sl:
        k = p.ops.kv_variable_full_or_delta_export(p.h, do_full_export=True, first_n=fn)[0]
        assert sorted(k.cpu().numpy().tolist()) == sorted(p.o.export(fn)[0].tolist()), tag
    elif op == "import":    # ImportValues empties both lists
      k, v, bl, fk, fv = ops.kv_variable_export(var.h, first_n=6)
      ok, ov, obl, ofk, ofv = var.o.export(6)
      ops.kv_variable_import(var.h, k, v, bl, fk, fv)
      var.o.import_(ok, ov, obl, ofk, ofv)
    else:
      fn = int(rng.choice([3, 4, 6]))
      for i, p in enumerate([var] + sl):
        exported += p.compare_delta(fn, tag + " table%d first_n=%d" % (i, fn))
  for i, p in enumerate([var] + sl):
    for fn in (6, 3):
      exported += p.compare_delta(fn, "seed %d final table%d first_n=%d" % (seed, i, fn))
  assert exported > 0


@pytest.mark.gpu
def test_delta_export_known_answers(ops):
  """Hand-checked: touched keys only; a deleted key is a delete key with frequency 0; a second export is empty;
  tracking off records nothing; a delete followed by a new lookup is an update, not a delete."""
  D = 4
  table = np.ones((8, D), np.float32)
  p = Pair(ops, D, 0, table, 1, track=None)
  ops.kv_variable_gather_or_insert_v2(p.h, np.arange(10))           # not tracked yet
  ops.kv_set_delta_tracking(p.h, True, False)
  ops.kv_variable_gather_or_insert_v2(p.h, np.array([1, 2, 3, 2]))
  assert ops.kv_variable_delete(p.h, np.array([3, 99])) == 1
  k, v, bl, fk, fv, full, dk = ops.kv_variable_full_or_delta_export(p.h, do_full_export=False, first_n=6)
  assert sorted(k.tolist()) == [1, 2] and bl.numel() == 0 and sorted(dk.tolist()) == [3, 99] and full is False
  np.testing.assert_array_equal(v.cpu().numpy(), np.ones((2, D), np.float32))
  f = dict(zip(fk.tolist(), fv.cpu().numpy().view(np.uint32).tolist()))
  assert f == {1: (DAY0 << 16) | 2, 2: (DAY0 << 16) | 3, 3: 0, 99: 0}
  out = ops.kv_variable_full_or_delta_export(p.h, do_full_export=False, first_n=6)
  assert all(x.numel() == 0 for x in (out[0], out[2], out[3], out[6]))
  assert ops.kv_variable_delete(p.h, np.array([5])) == 1
  ops.kv_variable_gather_or_insert_v2(p.h, np.array([5]))
  k, _, _, _, _, _, dk = ops.kv_variable_full_or_delta_export(p.h, do_full_export=False, first_n=4)
  assert k.tolist() == [5] and dk.numel() == 0
  ops.kv_set_delta_tracking(p.h, False, False)
  ops.kv_variable_gather_or_insert_v2(p.h, np.array([7, 8]))
  out = ops.kv_variable_full_or_delta_export(p.h, do_full_export=False, first_n=6)
  assert out[0].numel() == 0 and out[6].numel() == 0
