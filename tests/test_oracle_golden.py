"""Pins the CPU oracle (oracle/kv_oracle.cc) to the known-answer content of the
reference's own tests (restated as fixtures by tests/golden/make_golden.py), plus
closed-form cases derived from the reference kernels' row math.

CPU only; no GPU, no product code.
"""
import os

import numpy as np
import pytest

from oracle import kv_oracle as ko

DAY = 20000


def _load(golden_dir, name):
  return np.load(os.path.join(golden_dir, name))


def _mk(dim, table_val=1.0, rows=1024, thr=0):
  return ko.OracleKv(dim, thr, np.full((rows, dim), table_val, np.float32), day=DAY)


# --- G1: py_ut/tests/test_kv_variable_ops.py:234-268 -----------------------------------
def test_G1_gather_zeros_then_ones(golden_dir):
  g = _load(golden_dir, "G1G2_gather_frequency.npz")
  kv = _mk(8)
  assert kv.map_size() == 0 and kv.size() == 0           # test_kv_variable_shape/size :79-148
  np.testing.assert_array_equal(kv.gather_or_zeros(g["ids0"]), g["expect_zeros"])
  assert kv.map_size() == 0                               # GatherOrZeros never inserts
  np.testing.assert_array_equal(kv.gather_or_insert(g["ids0"]), g["expect_ones"])
  np.testing.assert_array_equal(kv.gather_or_zeros(g["ids0"]), g["expect_ones"])  # rows persist
  assert kv.map_size() == 5


# --- G2: test_kv_variable_ops.py:150-189 --------------------------------------------------
def test_G2_frequency_enter_threshold(golden_dir):
  g = _load(golden_dir, "G1G2_gather_frequency.npz")
  rng = np.random.default_rng(1)
  kv = ko.OracleKv(8, 2, rng.standard_normal((1024, 8)).astype(np.float32), day=DAY)
  kv.gather_or_insert(g["ids0"])
  assert kv.sum_freq() == g["expect_sum_freq"][0]
  kv.gather_or_insert(g["ids1"])
  assert kv.sum_freq() == g["expect_sum_freq"][1]
  kv.gather_or_zeros(g["ids0"])
  assert kv.sum_freq() == g["expect_sum_freq"][2]
  assert kv.size() == 3 and kv.map_size() == 7


# --- F1: kernels/kv_variable_test.cc:359-382 ---------------------------------------------
def test_F1_freq_word_packing(golden_dir):
  g = _load(golden_dir, "F1_freq_word.npz")
  for hi, lo, word in zip(g["hi"], g["lo"], g["word"]):
    kv = ko.OracleKv(4, 0, np.ones((4, 4), np.float32), day=int(hi))
    kv.gather_or_insert(np.array([7]), counts=np.array([int(lo)], np.int32))
    m = kv.meta(7)
    assert (m["day"] << 16 | m["freq"]) == int(word)
  # saturating add (utility.h:63-69): 65535 stays 65535
  kv = ko.OracleKv(4, 0, np.ones((4, 4), np.float32), day=1)
  kv.gather_or_insert(np.array([1]), counts=np.array([65534], np.int32))
  kv.gather_or_insert(np.array([1, 1, 1]))
  assert kv.meta(1)["freq"] == 65535
  # counts above uint16 clamp (utility.h:57-59)
  kv.gather_or_insert(np.array([2]), counts=np.array([1 << 20], np.int32))
  assert kv.meta(2)["freq"] == 65535


# --- A1: py_ut/tests/test_training_ops.py:437-473 ------------------------------------------
@pytest.mark.parametrize("D", [64, 1])
def test_A1_group_adam_v4_equals_tf_adam(golden_dir, D):
  g = _load(golden_dir, "A1_group_adam_v4_D%d.npz" % D)
  var = _mk(D, 1.0, rows=10000)                       # get_kv_variable: [10000, D] ones
  slot = _mk(3 * D, 0.0, rows=10000)                  # m_v_linear zeros slot (group_adam.py:143-152)
  b1, b2 = 0.9, 0.999
  ko.apply_group_adam(var, slot, g["grad"], g["ids"], 0.5, b1, b2, b1, b2, 1e-8)
  got = var.as_dict()
  res = np.stack([got[int(k)] for k in g["ids"]])
  # the reference asserts np.allclose(..., atol=1e-8) with the default rtol 1e-5
  np.testing.assert_allclose(res, g["expect_var"], rtol=1e-5, atol=1e-8)
  sl = slot.gather_or_zeros(g["ids"])
  np.testing.assert_allclose(sl[:, :D], g["expect_m"], rtol=1e-5, atol=1e-8)
  np.testing.assert_allclose(sl[:, D:2 * D], g["expect_v"], rtol=1e-5, atol=1e-8)
  # second step: beta powers advanced by TF-core _finish (beta1 > beta1_power branch)
  b1p, b2p = (float(x) for x in g["beta_powers"][1])
  ko.apply_group_adam(var, slot, g["grad2"], g["ids"], 0.5, b1p, b2p, b1, b2, 1e-8)
  got = var.as_dict()
  res = np.stack([got[int(k)] for k in g["ids"]])
  # step 2 is not asserted by the reference; x2 = x1 - O(.5) cancels to ~1e-2, so the fp32
  # rounding of the O(.5) terms (6e-8 relative) shows up as ~2e-7 absolute
  np.testing.assert_allclose(res, g["expect_var2"], rtol=1e-5, atol=1e-6)
  sl = slot.gather_or_zeros(g["ids"])
  np.testing.assert_allclose(sl[:, :D], g["expect_m2"], rtol=1e-5, atol=1e-8)
  np.testing.assert_allclose(sl[:, D:2 * D], g["expect_v2"], rtol=1e-5, atol=1e-8)


def test_A1_group_adam_v3_equals_tf_adam_step1(golden_dir):
  g = _load(golden_dir, "A1_group_adam_v4_D64.npz")
  var = _mk(64, 1.0)
  slot = _mk(192, 0.0)
  ko.apply_group_adam(var, slot, g["grad"], g["ids"], 0.5, 0.9, 0.999, 0.9, 0.999, 1e-8, version=3)
  got = var.as_dict()
  res = np.stack([got[int(k)] for k in g["ids"]])
  np.testing.assert_allclose(res, g["expect_var"], rtol=1e-5, atol=1e-8)


# --- A2: test_training_ops.py:418-435 -------------------------------------------------------
def test_A2_adagrad_equals_tf_adagrad(golden_dir):
  g = _load(golden_dir, "A2_adagrad.npz")
  var = _mk(64, 1.0)
  acc = _mk(64, 0.1)                                   # initial_accumulator_value = 0.1
  ko.apply_adagrad(var, acc, 0.5, g["grad"], g["ids"])
  got = var.as_dict()
  res = np.stack([got[int(k)] for k in g["ids"]])
  np.testing.assert_allclose(res, g["expect_var"], rtol=1e-5, atol=1e-8)
  np.testing.assert_allclose(acc.gather_or_zeros(g["ids"]), g["expect_acc"], rtol=1e-6)


# --- A3: test_training_ops.py:475-543: group-lasso FTRL must differ from plain FTRL -------
@pytest.mark.parametrize("D", [64, 1])
def test_A3_sparse_group_ftrl_runs_and_differs(D):
  rng = np.random.default_rng(3)
  grad = rng.random((10, D)).astype(np.float32)
  ids = np.arange(10)
  var, acc, lin = _mk(D, 1.0), _mk(D, 0.1), _mk(D, 0.0)
  ko.apply_sparse_group_ftrl(var, acc, lin, grad, ids, 0.5, 0.01, 0.05, 0.05, 0.0, -0.5)
  # plain TF FTRL (l21 = 0) in float64
  g = grad.astype(np.float64)
  a0, x0 = 0.1, 1.0
  na = a0 + g * g
  z = g - (np.sqrt(na) - np.sqrt(a0)) / 0.5 * x0
  plain = np.where(np.abs(z) > 0.01, (np.sign(z) * 0.01 - z) / (np.sqrt(na) / 0.5 + 2 * 0.05), 0.0)
  got = var.gather_or_zeros(ids)
  assert not np.allclose(got, plain, atol=1e-8)
  # and with l21 = 0 it IS plain FTRL
  var, acc, lin = _mk(D, 1.0), _mk(D, 0.1), _mk(D, 0.0)
  ko.apply_sparse_group_ftrl(var, acc, lin, grad, ids, 0.5, 0.01, 0.05, 0.0, 0.0, -0.5)
  if D > 1:
    np.testing.assert_allclose(var.gather_or_zeros(ids), plain, rtol=1e-5, atol=1e-7)
  np.testing.assert_allclose(acc.gather_or_zeros(ids), na, rtol=1e-6)


# --- A4: test_training_ops.py:68-205: one FTRL-V2 step, 300 ids x 64, equals TF's FTRL -----
def test_A4_ftrl_v2_equals_tf_ftrl(golden_dir):
  g = _load(golden_dir, "A4_ftrl_v2.npz")
  var, acc, lin = _mk(64, 0.03), _mk(64, 0.1), _mk(64, 0.0)
  ko.apply_sparse_group_ftrl(var, acc, lin, g["grad"], g["ids"], 0.01, 0.0, 0.0, 0.0, 0.0, -0.5)
  np.testing.assert_allclose(var.gather_or_zeros(g["ids"]), g["expect_var"], rtol=1e-5, atol=1e-8)
  np.testing.assert_allclose(acc.gather_or_zeros(g["ids"]), g["expect_accum"], rtol=1e-6)
  np.testing.assert_allclose(lin.gather_or_zeros(g["ids"]), g["expect_linear"], rtol=1e-5, atol=1e-6)


# --- closed-form cases authored from the kernels ("additional fixtures") ---
def test_group_lasso_blacklist_cycle():
  """training_ops.cc:7182-7192 + kv_variable.h:404-408 + table_manager.h:335-372"""
  D = 8
  var, slot = _mk(D, 1.0), _mk(3 * D, 0.0)
  ids = np.array([5])
  g = np.full((1, D), 1e-3, np.float32)
  var.gather_or_insert(ids)
  # huge l21 -> ||l1_linear|| <= l21_norm -> blacklist
  ko.apply_group_adam(var, slot, g, ids, 0.5, 0.9, 0.999, 0.9, 0.999, 1e-8, l21=1e6)
  m = var.meta(5)
  assert m["blacklist"] and m["under_threshold"]
  np.testing.assert_array_equal(var.gather_or_insert(ids), np.zeros((1, D), np.float32))
  np.testing.assert_array_equal(var.gather_or_zeros(ids), np.zeros((1, D), np.float32))
  assert 5 not in var.as_dict() and var.size() == 0
  keys, vals, bl, fk, fv = var.export(first_n=6)
  assert list(bl) == [5] and len(keys) == 0 and list(fk) == [5]
  # next apply un-blacklists to a fresh ZERO row, then updates it
  ko.apply_group_adam(var, slot, g, ids, 0.5, 0.81, 0.998001, 0.9, 0.999, 1e-8)
  m = var.meta(5)
  assert not m["blacklist"]
  assert np.all(np.abs(var.gather_or_zeros(ids)) > 0)


def test_enter_threshold_skips_apply():
  """training_ops.cc:7150-7152, kv_variable.h:402-405"""
  D = 4
  var = ko.OracleKv(D, 3, np.ones((16, D), np.float32), day=DAY)
  slot = _mk(3 * D, 0.0)
  ids = np.array([1, 2])
  var.gather_or_insert(np.array([1, 1, 1, 2]))          # freq(1)=3, freq(2)=1
  g = np.ones((2, D), np.float32)
  ko.apply_group_adam(var, slot, g, ids, 0.5, 0.9, 0.999, 0.9, 0.999, 1e-8)
  out = var.gather_or_zeros(ids)
  assert np.all(out[0] != 1.0) and np.all(out[1] == 1.0)
  assert slot.map_size() == 1                             # filtered key never reaches the slot table
  # a key first seen by the optimizer is inserted with freq word 1 and is NOT filtered
  ko.apply_group_adam(var, slot, g[:1], np.array([9]), 0.5, 0.9, 0.999, 0.9, 0.999, 1e-8)
  assert var.meta(9) == {"freq": 1, "day": 0, "blacklist": False, "under_threshold": False}
  assert np.all(var.gather_or_zeros(np.array([9])) != 1.0)


def test_under_threshold_rows_vanish_from_export():
  """kv_variable.h:837-861, dynamic_save.hpp:77-84"""
  var = ko.OracleKv(4, 0, np.zeros((4, 4), np.float32), day=DAY)
  var.gather_or_insert(np.array([1, 2]))
  assert var.map_size() == 2 and var.size() == 2
  assert var.as_dict() == {}


def test_slot_frequency_bumped_by_apply():
  """kv_variable.h:409-414: slot tables count one hit per apply after the inserting one"""
  D = 4
  var, slot = _mk(D, 1.0), _mk(3 * D, 0.0)
  ids = np.array([3])
  g = np.ones((1, D), np.float32)
  for t in range(3):
    ko.apply_group_adam(var, slot, g, ids, 0.5, 0.9, 0.999, 0.9, 0.999, 1e-8)
  assert slot.meta(3)["freq"] == 3 and slot.meta(3)["day"] == DAY
  assert var.meta(3)["freq"] == 1 and var.meta(3)["day"] == 0


def test_dedup_segment_sum_first_occurrence_order():
  ids = np.array([7, 3, 7, 7, 3, 9], np.int64)
  g = np.arange(12, dtype=np.float32).reshape(6, 2)
  u, s, pos = ko.dedup_segment_sum(ids, g)
  assert list(u) == [7, 3, 9] and list(pos) == [0, 1, 0, 0, 1, 2]
  np.testing.assert_array_equal(s, np.array([[0 + 4 + 6, 1 + 5 + 7], [2 + 8, 3 + 9], [10, 11]],
                                            np.float32))


def test_multithreaded_matches_single_thread():
  rng = np.random.default_rng(5)
  D = 16
  table = rng.standard_normal((64, D)).astype(np.float32)
  ids = rng.integers(0, 5000, 20000)
  a = ko.OracleKv(D, 0, table, day=DAY, picker=1, seed=7, threads=1)
  b = ko.OracleKv(D, 0, table, day=DAY, picker=1, seed=7, threads=4)
  np.testing.assert_array_equal(a.gather_or_insert(ids), b.gather_or_insert(ids))
  u, s, _ = ko.dedup_segment_sum(ids, rng.standard_normal((ids.size, D)).astype(np.float32))
  sa, sb = _mk(3 * D, 0.0), _mk(3 * D, 0.0)
  sb.threads = 4
  ko.apply_group_adam(a, sa, s, u, 1e-3, 0.9, 0.999, 0.9, 0.999, 1e-8)
  ko.apply_group_adam(b, sb, s, u, 1e-3, 0.9, 0.999, 0.9, 0.999, 1e-8)
  np.testing.assert_array_equal(a.gather_or_zeros(u), b.gather_or_zeros(u))
  # the reference bumps the frequency under a SHARED segment lock (kv_variable.h:320-332 inside
  # table_manager.h:167-190), so concurrent hits on one key can lose counts: only the
  # single-threaded sum is exact
  assert a.sum_freq() == ids.size and b.sum_freq() <= ids.size


def test_delta_export_known_answers():
  """DeltaExport (dynamic_save.hpp:198-451) of the oracle on a hand-checked sequence: touched keys only, a
  deleted key is a delete key with frequency 0, exports end the delta period, prediction exports (first_n <= 3)
  read train + prediction lists and move blacklisted keys to the delete list."""
  D = 4
  t = ko.OracleKv(D, 2, np.ones((8, D), np.float32), day=DAY, picker=1, seed=1)
  t.gather_or_insert(np.arange(10))                      # before tracking: in no list
  t.set_delta_tracking(True, True)
  t.gather_or_insert(np.array([1, 2, 3, 2, 11]))         # 11 is new: frequency 1 < enter_threshold 2
  t.delete(np.array([3, 99]))
  k, v, bl, fk, fv, dk = t.export_delta(6)
  assert sorted(k.tolist()) == [1, 2] and bl.size == 0 and sorted(dk.tolist()) == [3, 99]
  np.testing.assert_array_equal(v, np.ones((2, D), np.float32))
  assert dict(zip(fk.tolist(), fv.tolist())) == {1: (DAY << 16) | 2, 2: (DAY << 16) | 3, 3: 0, 99: 0, 11: (DAY << 16) | 1}
  assert all(x.size == 0 for x in t.export_delta(6))     # the training export emptied the train list ...
  k, v, bl, fk, fv, dk = t.export_delta(3)               # ... into the prediction list (SUPPORT_PREDICTION_DELTA_EXPORT)
  assert sorted(k.tolist()) == [1, 2] and sorted(dk.tolist()) == [3, 99] and fk.size == 0
  assert all(x.size == 0 for x in t.export_delta(3))
  # a key touched by scatter and then blacklisted by a group-lasso apply (key 2: frequency 3 >= enter_threshold)
  slot = ko.OracleKv(3 * D, 0, np.zeros((4, 3 * D), np.float32), day=DAY, picker=1, seed=1)
  slot.set_delta_tracking(True, False)
  t.scatter_update(np.array([2, 5]), np.ones((2, D), np.float32), 1)
  ko.apply_group_adam(t, slot, np.full((1, D), 0.01, np.float32), np.array([2]), 1e-2, 0.9, 0.999, 0.9, 0.999, 1e-8,
                      0.0, 0.0, 10.0)
  ko.apply_group_adam(t, slot, np.full((1, D), 0.01, np.float32), np.array([11]), 1e-2, 0.9, 0.999, 0.9, 0.999, 1e-8)
  k, v, bl, fk, fv, dk = t.export_delta(4)
  # 5 (frequency 1) is listed by the scatter but exports nothing; 11 was filtered by the apply: never listed
  assert k.size == 0 and bl.tolist() == [2] and dk.size == 0 and fk.size == 0
  assert sorted(slot.export_delta(4)[0].tolist()) == [2]                           # the slot table keeps its own list
  assert t.export_delta(3)[5].tolist() == [2]                                      # prediction export: blacklisted -> delete


def test_I1_import_v2_known_answer(golden_dir):
  """py_ut/tests/test_kv_variable_ops.py:345-435 (the reference's own KAT for import / export)."""
  z = np.load(os.path.join(golden_dir, "I1_import_v2.npz"))
  D = z["values"].shape[1]
  o = ko.OracleKv(D, int(z["enter_threshold"][0]), np.ones((1024, D), np.float32), day=20000)
  for i, first_n in enumerate(z["first_n"]):
    o.import_(z["keys"], z["values"], z["blacklist"], z["freq_keys"], z["freq_values"], first_n=int(first_n))
    k, v, b, fk, fv = o.export(first_n=6)
    assert k.shape == (z["expect_rows"][i],) and v.shape == (z["expect_rows"][i], D)
    assert b.shape == (z["expect_blacklist"][i],) and fk.shape == fv.shape == (z["expect_freq"][i],)
    assert sorted(k) == list(z["keys"]) and dict(zip(k, map(bytes, v))) == dict(zip(z["keys"], map(bytes, z["values"])))
    assert list(b) == ([] if first_n <= 3 else list(z["blacklist"]))
