"""GPU parity: the HIP path (through the C ABI, via the op-level host functions) against the
CPU oracle on the same seeded inputs, plus the reference tests' known answers re-run on the GPU.

Bars (BASELINE.json north_star): ids / indices / frequency words / flags bit-exact; fp32 rows and
optimizer state within 1e-6 relative.
"""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu

from oracle import kv_oracle as ko  # noqa: E402  (checker only)

DAY = 20000
RTOL = 1e-6


@pytest.fixture(scope="module")
def ops():
  if not torch.cuda.is_available():
    pytest.skip("needs a GPU")
  from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as g
  return g


def _pair(ops, D, thr=0, table=None, seed=0, rows=64, rng=None, cap=0):
  """A GPU table and its oracle twin with the same init table, seed and day."""
  if table is None:
    table = (rng or np.random.default_rng(0)).standard_normal((rows, D)).astype(np.float32)
  h = ops.kv_variable([D], enter_threshold=thr, capacity_hint=cap)
  ops.kv_set_clock_days(h, DAY)
  ops.kv_set_seed(h, seed)
  ops.init_kv_variable_v2(h, table)
  o = ko.OracleKv(D, thr, table, day=DAY, picker=1, seed=seed)
  return h, o


def _const(ops, D, val, thr=0, rows=16):
  return _pair(ops, D, thr, np.full((rows, D), val, np.float32))


def _np(t):
  return t.detach().cpu().numpy()


def _zipf_ids(rng, n, universe, s=1.2):
  ranks = np.arange(1, universe + 1, dtype=np.float64)
  p = ranks**-s
  p /= p.sum()
  return rng.choice(universe, size=n, p=p).astype(np.int64)


def _assert_same_table(ops, h, o, keys, rtol=0.0, atol=0.0):
  """rows, frequency words and flags of `keys` agree; table-wide counters agree."""
  keys = np.unique(np.asarray(keys, np.int64))
  got = _np(ops.kv_variable_gather_or_zeros_v2(h, keys))
  exp = o.gather_or_zeros(keys)
  if rtol == 0.0 and atol == 0.0:
    np.testing.assert_array_equal(got, exp)
  else:
    np.testing.assert_allclose(got, exp, rtol=rtol, atol=atol)
  metas = ops.kv_get_meta(h, keys)
  for k, m in zip(keys, metas):
    assert m == o.meta(int(k)), (int(k), m, o.meta(int(k)))
  assert ops.kv_variable_size_v2(h) == o.size()
  assert ops.kv_variable_frequency(h) == o.sum_freq()
  assert ops.kv_variable_shape_v2(h) == [o.map_size(), o.dim]


# ---------------------------------------------------------------------------------------------
# the reference tests' known answers, on the GPU
# ---------------------------------------------------------------------------------------------
def test_G1_gather_zeros_then_ones(ops, golden_dir):
  g = np.load(os.path.join(golden_dir, "G1G2_gather_frequency.npz"))
  h = ops.kv_variable([8])
  assert not ops.kv_variable_is_initialized_v2(h)
  ops.init_kv_variable_v2(h, np.ones((1024, 8), np.float32))
  assert ops.kv_variable_is_initialized_v2(h)
  assert ops.kv_variable_shape_v2(h) == [0, 8] and ops.kv_variable_size_v2(h) == 0
  np.testing.assert_array_equal(_np(ops.kv_variable_gather_or_zeros_v2(h, g["ids0"])), g["expect_zeros"])
  assert ops.kv_variable_shape_v2(h) == [0, 8]
  np.testing.assert_array_equal(_np(ops.kv_variable_gather_or_insert_v2(h, g["ids0"])), g["expect_ones"])
  np.testing.assert_array_equal(_np(ops.kv_variable_gather_or_zeros_v2(h, g["ids0"])), g["expect_ones"])
  assert ops.kv_variable_shape_v2(h) == [5, 8]


def test_G2_frequency_enter_threshold(ops, golden_dir):
  g = np.load(os.path.join(golden_dir, "G1G2_gather_frequency.npz"))
  h, _ = _pair(ops, 8, thr=2, rows=1024)
  ops.kv_variable_gather_or_insert_v2(h, g["ids0"])
  assert ops.kv_variable_frequency(h) == g["expect_sum_freq"][0]
  ops.kv_variable_gather_or_insert_v2(h, g["ids1"])
  assert ops.kv_variable_frequency(h) == g["expect_sum_freq"][1]
  ops.kv_variable_gather_or_zeros_v2(h, g["ids0"])
  assert ops.kv_variable_frequency(h) == g["expect_sum_freq"][2]
  assert ops.kv_variable_size_v2(h) == 3


def test_F1_freq_word(ops, golden_dir):
  g = np.load(os.path.join(golden_dir, "F1_freq_word.npz"))
  for hi, lo, word in zip(g["hi"], g["lo"], g["word"]):
    h = ops.kv_variable([4])
    ops.init_kv_variable_v2(h, np.ones((4, 4), np.float32))
    ops.kv_set_clock_days(h, int(hi))
    ops.kv_variable_gather_or_insert_with_counts(h, np.array([7]), np.array([int(lo)], np.int32))
    m = ops.kv_get_meta(h, [7])[0]
    assert (m["day"] << 16 | m["freq"]) == int(word)


@pytest.mark.parametrize("D", [64, 1])
def test_A1_group_adam_v4_equals_tf_adam(ops, golden_dir, D):
  g = np.load(os.path.join(golden_dir, "A1_group_adam_v4_D%d.npz" % D))
  var, _ = _const(ops, D, 1.0)
  slot, _ = _const(ops, 3 * D, 0.0)
  b1, b2 = 0.9, 0.999
  ops.kv_variable_group_sparse_apply_adam_v4(var, slot, g["grad"], g["ids"], 0.5, b1, b2, b1, b2,
                                             1e-8, 0.0, 0.0, 0.0)
  k, v = ops.read_kv_variable_op_v2(var)
  d = {int(a): b for a, b in zip(_np(k), _np(v))}
  res = np.stack([d[int(i)] for i in g["ids"]])
  np.testing.assert_allclose(res, g["expect_var"], rtol=1e-5, atol=1e-8)  # the reference's tolerance
  sl = _np(ops.kv_variable_gather_or_zeros_v2(slot, g["ids"]))
  np.testing.assert_allclose(sl[:, :D], g["expect_m"], rtol=1e-5, atol=1e-8)
  np.testing.assert_allclose(sl[:, D:2 * D], g["expect_v"], rtol=1e-5, atol=1e-8)
  b1p, b2p = (float(x) for x in g["beta_powers"][1])
  ops.kv_variable_group_sparse_apply_adam_v4(var, slot, g["grad2"], g["ids"], 0.5, b1p, b2p, b1, b2,
                                             1e-8, 0.0, 0.0, 0.0)
  res = _np(ops.kv_variable_gather_or_zeros_v2(var, g["ids"]))
  np.testing.assert_allclose(res, g["expect_var2"], rtol=1e-5, atol=1e-6)


def test_A2_adagrad_equals_tf_adagrad(ops, golden_dir):
  g = np.load(os.path.join(golden_dir, "A2_adagrad.npz"))
  var, _ = _const(ops, 64, 1.0)
  acc, _ = _const(ops, 64, 0.1)
  ops.kv_variable_sparse_apply_adagrad(var, acc, 0.5, g["grad"], g["ids"], use_locking=True)
  np.testing.assert_allclose(_np(ops.kv_variable_gather_or_zeros_v2(var, g["ids"])), g["expect_var"],
                             rtol=1e-5, atol=1e-8)
  np.testing.assert_allclose(_np(ops.kv_variable_gather_or_zeros_v2(acc, g["ids"])), g["expect_acc"],
                             rtol=1e-6)


def test_A4_ftrl_v2_equals_tf_ftrl(ops, golden_dir):
  g = np.load(os.path.join(golden_dir, "A4_ftrl_v2.npz"))
  var, _ = _const(ops, 64, 0.03)
  acc, _ = _const(ops, 64, 0.1)
  lin, _ = _const(ops, 64, 0.0)
  ops.kv_variable_sparse_group_sparse_apply_ftrl_v2(var, acc, lin, g["grad"], g["ids"], 0.01, 0.0, 0.0, 0.0, 0.0, -0.5)
  np.testing.assert_allclose(_np(ops.kv_variable_gather_or_zeros_v2(var, g["ids"])), g["expect_var"],
                             rtol=1e-5, atol=1e-8)
  np.testing.assert_allclose(_np(ops.kv_variable_gather_or_zeros_v2(acc, g["ids"])), g["expect_accum"], rtol=1e-6)
  np.testing.assert_allclose(_np(ops.kv_variable_gather_or_zeros_v2(lin, g["ids"])), g["expect_linear"],
                             rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------------------------------------
# lookup parity vs the oracle
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D", [1, 5, 8, 12, 32, 64, 128, 260])
def test_lookup_parity_random_init(ops, D):
  rng = np.random.default_rng(100 + D)
  h, o = _pair(ops, D, thr=2, seed=11, rng=rng)
  for step in range(3):
    ids = _zipf_ids(rng, 5000, 3000)
    ids[::97] = -ids[::97] - 1                       # negative keys are ordinary keys
    got = _np(ops.kv_variable_gather_or_insert_v2(h, ids.reshape(50, 100)))
    exp = o.gather_or_insert(ids.reshape(50, 100))
    assert got.shape == (50, 100, D)
    np.testing.assert_array_equal(got, exp)          # rows are copies: bit-exact
    _assert_same_table(ops, h, o, ids)
  # inference lookup: hits, misses, no side effects
  probe = np.concatenate([ids[:100], np.arange(10**9, 10**9 + 50)])
  np.testing.assert_array_equal(_np(ops.kv_variable_gather_or_zeros_v2(h, probe)), o.gather_or_zeros(probe))
  _assert_same_table(ops, h, o, ids)


def test_partition_hint_survives_changing_batch_sizes(ops):
  """The entry-list pipeline sizes its partition pass by the distinct keys of the table's PREVIOUS batch of the same
  length (a pinned word, no synchronisation).  The hint may be stale or come from a batch of another length: mostly
  distinct ids, then tiny batches, then heavy repeats, then distinct ids again — rows and table state equal the oracle
  every time (a wrong partition count must cost time, never results or memory)."""
  D = 32
  rng = np.random.default_rng(55)
  h, o = _pair(ops, D, seed=5, rng=rng, cap=1 << 20)
  hs, os_ = _const(ops, 3 * D, 0.0)
  plans = [("distinct", 300000), ("small", 2048), ("small", 2048), ("small", 2048), ("repeats", 300000),
           ("repeats", 300000), ("distinct", 300000), ("distinct", 300000), ("small", 5000), ("repeats", 300000)]
  for step, (kind, n) in enumerate(plans):
    if kind == "distinct":
      ids = rng.choice(2_000_000, n, replace=False).astype(np.int64)
    elif kind == "repeats":
      ids = rng.integers(0, 300, n).astype(np.int64)
    else:
      ids = rng.integers(0, 100000, n).astype(np.int64)
    got = _np(ops.kv_variable_gather_or_insert_v2(h, ids))
    if step <= 2:     # no optimizer step yet: rows are copies of the init rule's rows
      np.testing.assert_array_equal(got, o.gather_or_insert(ids), err_msg="step %d %s" % (step, kind))
    else:             # the applies below sum repeated ids in another fp32 order than the oracle's pre-summed input
      np.testing.assert_allclose(got, o.gather_or_insert(ids), rtol=2e-5, atol=1e-6, err_msg="step %d %s" % (step, kind))
    if step % 3 == 2:
      grad = rng.uniform(0.5, 1.5, (n, D)).astype(np.float32) * 1e-3
      b1p, b2p = _beta_pows(step // 3)
      _apply_both(ops, "adam4", (h, hs), (o, os_), grad, ids, lr=1e-2, b1p=b1p, b2p=b2p, l1=0.0, l2=0.0, l21=0.0)
    sample = ids[:: max(1, n // 2000)]
    got = _np(ops.kv_variable_gather_or_zeros_v2(h, sample))
    np.testing.assert_allclose(got, o.gather_or_zeros(sample), rtol=2e-5, atol=1e-6, err_msg="state, step %d" % step)
    assert ops.kv_variable_size_v2(h) == o.size() and ops.kv_variable_frequency(h) == o.sum_freq()


def test_lookup_with_counts_and_saturation(ops):
  rng = np.random.default_rng(7)
  h, o = _pair(ops, 16, thr=3, seed=5, rng=rng)
  ids = rng.integers(0, 200, 4000)
  counts = rng.integers(0, 40000, 4000).astype(np.int32)
  counts[::13] = 70000                               # clamps to 65535 (utility.h:57-59)
  got = _np(ops.kv_variable_gather_or_insert_with_counts(h, ids, counts))
  np.testing.assert_array_equal(got, o.gather_or_insert(ids, counts))
  _assert_same_table(ops, h, o, ids)
  with pytest.raises(ValueError):
    ops.kv_variable_gather_or_insert_with_counts(h, ids, counts[:-1])
  with pytest.raises(ValueError):
    ops.kv_variable_gather_or_insert_with_counts(h, ids, counts.astype(np.int64))


def test_lookup_edge_cases(ops):
  h, o = _pair(ops, 8, seed=3)
  # empty input (kv_variable_ops.cc:530-532)
  out = ops.kv_variable_gather_or_insert_v2(h, np.zeros((0,), np.int64))
  assert tuple(out.shape) == (0, 8)
  # the key that equals the table's EMPTY sentinel, INT64 extremes, all-duplicates, ragged tail
  ids = np.array([np.iinfo(np.int64).min, np.iinfo(np.int64).max, 0, -1,
                  np.iinfo(np.int64).min] + [42] * 1500, np.int64)
  np.testing.assert_array_equal(_np(ops.kv_variable_gather_or_insert_v2(h, ids)), o.gather_or_insert(ids))
  _assert_same_table(ops, h, o, ids)
  np.testing.assert_array_equal(_np(ops.kv_variable_gather_or_zeros_v2(h, ids[:5])), o.gather_or_zeros(ids[:5]))
  # int32 keys
  h32 = ops.kv_variable([8], key_dtype=torch.int32)
  ops.init_kv_variable_v2(h32, np.ones((4, 8), np.float32))
  out = ops.kv_variable_gather_or_insert_v2(h32, np.array([-5, 7, 7], np.int32))
  np.testing.assert_array_equal(_np(out), np.ones((3, 8), np.float32))
  assert ops.kv_variable_shape_v2(h32) == [2, 8]


def test_uninitialized_is_failed_precondition(ops):
  from tfplus_amd import _lib
  h = ops.kv_variable([8])
  with pytest.raises(_lib.FailedPreconditionError):
    ops.kv_variable_gather_or_insert_v2(h, np.array([1]))
  with pytest.raises(_lib.FailedPreconditionError):   # FindOrZeros checks too (kv_variable.h:242)
    ops.kv_variable_gather_or_zeros_v2(h, np.array([1]))
  s = ops.kv_variable([24])
  with pytest.raises(_lib.FailedPreconditionError):   # training_ops.cc:7001-7008
    ops.kv_variable_group_sparse_apply_adam_v4(h, s, np.zeros((1, 8), np.float32), np.array([1]), 0.1,
                                               0.9, 0.999, 0.9, 0.999, 1e-8, 0, 0, 0)


def test_growth_rehash_keeps_every_key(ops):
  rng = np.random.default_rng(9)
  h, o = _pair(ops, 4, seed=2, rng=rng)               # default capacity: 65536 rows
  all_ids = []
  for step in range(4):
    ids = rng.integers(-2**62, 2**62, 60000)
    all_ids.append(ids)
    np.testing.assert_array_equal(_np(ops.kv_variable_gather_or_insert_v2(h, ids)), o.gather_or_insert(ids))
  ids = np.concatenate(all_ids)
  assert ops.kv_variable_shape_v2(h)[0] == o.map_size() > 200000
  np.testing.assert_array_equal(_np(ops.kv_variable_gather_or_zeros_v2(h, ids)), o.gather_or_zeros(ids))


# ---------------------------------------------------------------------------------------------
# optimizer parity vs the oracle
# ---------------------------------------------------------------------------------------------
def _apply_both(ops, name, hs, os_, grad, ids, **kw):
  """Runs one optimizer op on the GPU tables `hs` and on the oracle tables `os_`.  The oracle gets
  the TF-core de-duplicated (unique ids, summed grads), i.e. exactly what the reference op sees."""
  u, s, _ = ko.dedup_segment_sum(ids, grad)
  if name == "adam4" or name == "adam3":
    v = 4 if name == "adam4" else 3
    fn = ops.kv_variable_group_sparse_apply_adam_v4 if v == 4 else ops.kv_variable_group_sparse_apply_adam_v3
    fn(hs[0], hs[1], grad, ids, kw["lr"], kw["b1p"], kw["b2p"], 0.9, 0.999, 1e-8, kw["l1"], kw["l2"], kw["l21"])
    ko.apply_group_adam(os_[0], os_[1], s, u, kw["lr"], kw["b1p"], kw["b2p"], 0.9, 0.999, 1e-8,
                        kw["l1"], kw["l2"], kw["l21"], version=v)
  elif name == "adagrad":
    ops.kv_variable_sparse_apply_adagrad(hs[0], hs[1], kw["lr"], grad, ids, update_slots=kw.get("us", True))
    ko.apply_adagrad(os_[0], os_[1], kw["lr"], s, u, kw.get("us", True))
  else:
    ops.kv_variable_sparse_group_sparse_apply_ftrl_v2(hs[0], hs[1], hs[2], grad, ids, kw["lr"], kw["l1"],
                                                      kw["l2"], kw["l21"], kw["l2s"], kw["lrp"])
    ko.apply_sparse_group_ftrl(os_[0], os_[1], os_[2], s, u, kw["lr"], kw["l1"], kw["l2"], kw["l21"],
                               kw["l2s"], kw["lrp"])


def _beta_pows(t):
  p1, p2 = np.float32(0.9), np.float32(0.999)
  for _ in range(t):
    p1, p2 = np.float32(p1 * np.float32(0.9)), np.float32(p2 * np.float32(0.999))
  return float(p1), float(p2)


@pytest.mark.parametrize("D", [1, 6, 8, 32, 64, 128])
@pytest.mark.parametrize("ver", ["adam4", "adam3"])
def test_group_adam_parity_unique_ids(ops, D, ver):
  rng = np.random.default_rng(200 + D)
  hv, ov = _pair(ops, D, seed=1, rng=rng)
  hs, os_ = _const(ops, 3 * D, 0.0)
  seen = []
  for t in range(3):
    ids = rng.choice(5000, 1500, replace=False).astype(np.int64)
    seen.append(ids)
    if t == 1:                                        # forward lookups interleave like a training step
      np.testing.assert_array_equal(_np(ops.kv_variable_gather_or_insert_v2(hv, ids)), ov.gather_or_insert(ids))
    grad = rng.normal(0, 1e-2, (ids.size, D)).astype(np.float32)
    b1p, b2p = _beta_pows(t)
    _apply_both(ops, ver, (hv, hs), (ov, os_), grad, ids, lr=1e-2, b1p=b1p, b2p=b2p, l1=0.0, l2=0.0, l21=0.0)
    keys = np.concatenate(seen)
    _assert_same_table(ops, hv, ov, keys, rtol=RTOL, atol=1e-9)
    _assert_same_table(ops, hs, os_, keys, rtol=RTOL, atol=1e-12)


def test_delete_between_lookup_and_apply_makes_the_token_stale(ops):
  """lookup(ids) -> delete(some of them) -> apply(the SAME ids tensor): the Python layer hands the optimizer op the
  lookup's batch token (same tensor object); the delete must have invalidated it, so the apply re-creates the deleted
  keys like the reference's FindOrInsertUnsafe would (and does not write into freed rows)."""
  D = 32
  rng = np.random.default_rng(77)
  hv, ov = _pair(ops, D, seed=1, rng=rng)
  hs, os_ = _const(ops, 3 * D, 0.0)
  for t in range(3):
    ids_np = rng.choice(4000, 1200, replace=False).astype(np.int64)
    ids = torch.from_numpy(ids_np).cuda()                                  # ONE tensor object for lookup and apply
    np.testing.assert_array_equal(_np(ops.kv_variable_gather_or_insert_v2(hv, ids)), ov.gather_or_insert(ids_np))
    gone = ids_np[rng.choice(ids_np.size, 300, replace=False)]
    if t == 2:
      assert ops.kv_variable_delete_with_timestamp(hv, threshold=0).numel() == len(ov.delete_with_timestamp(0))
    else:
      assert ops.kv_variable_delete(hv, gone) == 300
      ov.delete(gone)
    grad = torch.from_numpy(rng.normal(0, 1e-2, (ids_np.size, D)).astype(np.float32)).cuda()
    b1p, b2p = _beta_pows(t)
    ops.kv_variable_group_sparse_apply_adam_v4(hv, hs, grad, ids, 1e-2, b1p, b2p, 0.9, 0.999, 1e-8, 0.0, 0.0, 0.0)
    ko.apply_group_adam(ov, os_, _np(grad), ids_np, 1e-2, b1p, b2p, 0.9, 0.999, 1e-8, 0.0, 0.0, 0.0, version=4)
    _assert_same_table(ops, hv, ov, ids_np, rtol=RTOL, atol=1e-9)
    _assert_same_table(ops, hs, os_, ids_np, rtol=RTOL, atol=1e-12)


def test_reserve_between_token_lookup_and_apply(ops):
  """a token lookup returns with its partition pass pending (new keys sit in the index with no row contents
  yet).  kv_reserve in between grows the slab and rebuilds the index: the pending pass must have run first.  Then the
  apply with the same tensor object (the token is stale after the growth: the general path) against the oracle."""
  D = 32
  rng = np.random.default_rng(404)
  hv, ov = _pair(ops, D, seed=2, rng=rng)
  hs, os_ = _const(ops, 3 * D, 0.0)
  for t in range(3):
    ids_np = (rng.choice(50000, 6000, replace=True) + 100000 * t).astype(np.int64)      # new keys every step, with repeats
    ids = torch.from_numpy(ids_np).cuda()
    np.testing.assert_array_equal(_np(ops.kv_variable_gather_or_insert_v2(hv, ids)), ov.gather_or_insert(ids_np))
    ops.kv_reserve(hv, 200000 * (t + 1))                # larger than the table: slab chunks + index rebuild
    u, gs, _ = ko.dedup_segment_sum(ids_np, rng.normal(0, 1e-2, (ids_np.size, D)).astype(np.float32))
    b1p, b2p = _beta_pows(t)
    ops.kv_variable_group_sparse_apply_adam_v4(hv, hs, gs, u, 1e-2, b1p, b2p, 0.9, 0.999, 1e-8, 0.0, 0.0, 0.0)
    ko.apply_group_adam(ov, os_, gs, u, 1e-2, b1p, b2p, 0.9, 0.999, 1e-8, 0.0, 0.0, 0.0, version=4)
    _assert_same_table(ops, hv, ov, ids_np, rtol=RTOL, atol=1e-9)
    _assert_same_table(ops, hs, os_, ids_np, rtol=RTOL, atol=1e-12)


def test_seed_change_after_a_token_lookup_keeps_the_rows_it_returned(ops):
  """The rows a lookup returned for new keys are the rows the table holds afterwards, whatever happens to the seed
  before the deferred pass writes them (kv_set_seed settles pending work first)."""
  D = 16
  rng = np.random.default_rng(9)
  hv, ov = _pair(ops, D, seed=3, rng=rng)
  ids_np = rng.choice(10 ** 6, 3000, replace=True).astype(np.int64)
  ids = torch.from_numpy(ids_np).cuda()
  got = _np(ops.kv_variable_gather_or_insert_v2(hv, ids))
  np.testing.assert_array_equal(got, ov.gather_or_insert(ids_np))
  ops.kv_set_seed(hv, 12345)
  np.testing.assert_array_equal(_np(ops.kv_variable_gather_or_zeros_v2(hv, ids)), got)
  _assert_same_table(ops, hv, ov, ids_np)


def test_two_optimizer_steps_on_one_lookup_token(ops):
  """lookup(ids) -> apply(ids, token) -> apply(ids, token) again (a second optimizer step on the same batch index): the
  first apply consumed the lookup's pending bookkeeping (k_papply, PA_LOOKUP), the second must find the batch indexed
  and bookkept (PA_NONE) — frequencies count the lookup once."""
  D = 32
  rng = np.random.default_rng(1234)
  hv, ov = _pair(ops, D, seed=5, rng=rng)
  hs, os_ = _const(ops, 3 * D, 0.0)
  from tfplus_amd import _lib
  import ctypes
  L = _lib.lib()
  st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
  for t in range(2):
    ids_np = rng.integers(0, 8000, 20000).astype(np.int64) + 8000 * t      # a few repeats per id (the summation order hardly matters), new keys in step 1
    ids = torch.from_numpy(ids_np).cuda()
    out = torch.empty((ids_np.size, D), device="cuda")
    tok = ctypes.c_uint64(0)
    _lib.check(L.kv_gather_or_insert_tok(hv.ptr, ids.data_ptr(), None, ids_np.size, out.data_ptr(), ctypes.byref(tok), st))
    np.testing.assert_array_equal(_np(out), ov.gather_or_insert(ids_np))
    u = np.unique(ids_np)
    for rep in range(2):
      # one gradient row per id, repeated for every occurrence (the sum then hardly depends on its order)
      g_np = rng.normal(0, 1e-2, (u.size, D)).astype(np.float32)[np.searchsorted(u, ids_np)]
      grad = torch.from_numpy(g_np).cuda()
      b1p, b2p = _beta_pows(2 * t + rep)
      _lib.check(L.kv_apply_group_adam_tok(hv.ptr, hs.ptr, grad.data_ptr(), ids.data_ptr(), ids_np.size, 1e-2, float(b1p), float(b2p),
                                           0.9, 0.999, 1e-8, 0.0, 0.0, 0.0, 4, tok.value, st))
      uu, gs, _ = ko.dedup_segment_sum(ids_np, g_np)
      ko.apply_group_adam(ov, os_, gs, uu, 1e-2, b1p, b2p, 0.9, 0.999, 1e-8, 0.0, 0.0, 0.0, version=4)
    _assert_same_table(ops, hv, ov, ids_np, rtol=2e-5, atol=1e-7)      # repeated ids: summation order (tests/_reorder.py has the bound)
    _assert_same_table(ops, hs, os_, ids_np, rtol=2e-5, atol=1e-7)


def test_group_adam_parity_with_regularizers_and_blacklist(ops):
  rng = np.random.default_rng(31)
  D = 32
  hv, ov = _pair(ops, D, seed=4, rng=rng)
  hs, os_ = _const(ops, 3 * D, 0.0)
  ids = np.arange(3000, dtype=np.int64)
  ops.kv_variable_gather_or_insert_v2(hv, ids)
  ov.gather_or_insert(ids)
  for t in range(3):
    # per-row gradient scale spreads ||l1_linear|| around the l21 threshold: some rows blacklist
    grad = (rng.normal(0, 1, (ids.size, D)) * rng.uniform(1e-4, 3e-2, (ids.size, 1))).astype(np.float32)
    b1p, b2p = _beta_pows(t)
    _apply_both(ops, "adam4", (hv, hs), (ov, os_), grad, ids, lr=0.05, b1p=b1p, b2p=b2p, l1=1e-3, l2=1e-2,
                l21=2e-2)
    nb = sum(1 for k in ids if ov.meta(int(k))["blacklist"])
    assert 0 < nb < ids.size, nb                      # the case really exercises both branches
    # x = u * (1 - l21n/||u||) / y: the norm's fp32 reduction order (Eigen packets in the
    # reference, sequential in the oracle, an 8-lane shuffle tree on the GPU) moves ||u|| by an
    # ulp, which the subtraction amplifies by 1/scale.  Tolerance per row = 1e-6 / scale;
    # rows closer than 1e-4 relative to the threshold may legitimately flip and are skipped.
    lr, l1s, l21n = np.float32(0.05), np.float32(1e-3 * 0.05), np.float32(2e-2 * 0.05) * np.sqrt(np.float32(D))
    z = os_.gather_or_zeros(ids)[:, 2 * D:]
    nrm = np.sqrt(((np.clip(z, -l1s, l1s) - z).astype(np.float64)**2).sum(1))
    scale = 1.0 - l21n / np.maximum(nrm, 1e-30)
    clear = np.abs(scale) > 1e-4
    assert clear.sum() > ids.size * 0.99
    got, exp = _np(ops.kv_variable_gather_or_zeros_v2(hv, ids)), ov.gather_or_zeros(ids)
    tol = RTOL / np.maximum(scale, 1e-4)[:, None] * np.abs(exp) + 1e-9
    assert np.all((np.abs(got - exp) <= tol)[clear])
    gm, om = ops.kv_get_meta(hv, ids), [ov.meta(int(k)) for k in ids]
    assert all(a == b for a, b, c in zip(gm, om, clear) if c)
    _assert_same_table(ops, hs, os_, ids, rtol=RTOL, atol=1e-10)   # z cancels towards 0
    # blacklisted keys read as zeros through both lookups, and still count frequency
    gi, oi = _np(ops.kv_variable_gather_or_insert_v2(hv, ids)), ov.gather_or_insert(ids)
    np.testing.assert_array_equal(gi, got)             # training lookup == inference lookup
    black = np.array([m["blacklist"] for m in om])
    assert np.all(gi[black & clear] == 0) and np.all(oi[black] == 0)
    assert [m["freq"] for m in ops.kv_get_meta(hv, ids)] == [ov.meta(int(k))["freq"] for k in ids]
  ek, ev, ebl, efk, efv = ops.kv_variable_export(hv, first_n=6)
  ok, ovv, obl, ofk, ofv = ov.export(first_n=6)
  assert sorted(_np(ebl)) == sorted(obl) and sorted(_np(ek)) == sorted(ok)
  assert dict(zip(_np(efk), _np(efv).view(np.uint32))) == dict(zip(ofk, ofv))


def test_enter_threshold_skips_rows(ops):
  D = 8
  hv, ov = _pair(ops, D, thr=3, seed=6)
  hs, os_ = _const(ops, 3 * D, 0.0)
  warm = np.array([1, 1, 1, 2, 3, 3, 3, 3, 4], np.int64)
  ops.kv_variable_gather_or_insert_v2(hv, warm)
  ov.gather_or_insert(warm)
  ids = np.array([1, 2, 3, 4, 9], np.int64)          # 2 and 4 are below threshold; 9 is new
  grad = np.ones((5, D), np.float32)
  _apply_both(ops, "adam4", (hv, hs), (ov, os_), grad, ids, lr=0.1, b1p=0.9, b2p=0.999, l1=0, l2=0, l21=0)
  _assert_same_table(ops, hv, ov, ids, rtol=RTOL)
  _assert_same_table(ops, hs, os_, ids, rtol=RTOL)
  assert ops.kv_variable_shape_v2(hs)[0] == 3


@pytest.mark.parametrize("D", [1, 8, 32, 100])
def test_adagrad_parity(ops, D):
  rng = np.random.default_rng(300 + D)
  hv, ov = _pair(ops, D, seed=8, rng=rng)
  ha, oa = _const(ops, D, 0.1)
  for t in range(3):
    ids = rng.choice(4000, 1200, replace=False).astype(np.int64)
    grad = rng.normal(0, 1e-1, (ids.size, D)).astype(np.float32)
    _apply_both(ops, "adagrad", (hv, ha), (ov, oa), grad, ids, lr=0.05, us=(t != 1))
    _assert_same_table(ops, hv, ov, ids, rtol=RTOL, atol=1e-9)
    _assert_same_table(ops, ha, oa, ids, rtol=RTOL)


@pytest.mark.parametrize("lrp,l2s", [(-0.5, 0.0), (-0.5, 0.01), (-0.7, 0.0)])
@pytest.mark.parametrize("D", [1, 16, 32])
def test_sparse_group_ftrl_parity(ops, D, lrp, l2s):
  rng = np.random.default_rng(400 + D)
  hv, ov = _pair(ops, D, seed=9, rng=rng)
  ha, oa = _const(ops, D, 0.1)
  hl, ol = _const(ops, D, 0.0)
  ids = np.arange(2000, dtype=np.int64)
  for t in range(3):
    grad = (rng.normal(0, 1, (ids.size, D)) * rng.uniform(1e-3, 1e-1, (ids.size, 1))).astype(np.float32)
    _apply_both(ops, "ftrl", (hv, ha, hl), (ov, oa, ol), grad, ids, lr=0.1, l1=1e-3, l2=1e-2, l21=1e-2,
                l2s=l2s, lrp=lrp)
    if lrp == -0.5:
      tol, atol = RTOL, 1e-8                          # rows sit around 1e-2: 1e-8 abs = 1e-6 of scale
    else:
      # lr_power != -0.5 goes through powf (Eigen -> libm in the reference, ocml on the GPU): the
      # two differ by an ulp and (new_accum^p - accum^p) cancels ~200x, so this third-party-math
      # case is held to 1e-4 relative instead of 1e-6
      tol, atol = 1e-4, 1e-6
    _assert_same_table(ops, hv, ov, ids, rtol=tol, atol=atol)
    _assert_same_table(ops, ha, oa, ids, rtol=tol)
    _assert_same_table(ops, hl, ol, ids, rtol=tol, atol=atol)


def test_apply_with_repeated_ids_is_segment_sum_then_apply(ops):
  """The fused path: duplicates are combined like TF-core's tf.unique + unsorted_segment_sum.
  Few addends per key -> fp32 sums agree with the occurrence-ordered sum to an ulp."""
  rng = np.random.default_rng(55)
  D = 32
  hv, ov = _pair(ops, D, seed=10, rng=rng)
  hs, os_ = _const(ops, 3 * D, 0.0)
  for t in range(3):
    ids = _zipf_ids(rng, 6000, 4000, s=0.7)
    # one-signed gradients: Adam's step m/sqrt(v) is discontinuous at g = 0, so a sum that cancels
    # to ~0 would turn an ulp of summation-order difference into a visible difference in x
    grad = np.abs(rng.normal(0, 1e-2, (ids.size, D))).astype(np.float32)
    b1p, b2p = _beta_pows(t)
    _apply_both(ops, "adam4", (hv, hs), (ov, os_), grad, ids, lr=1e-2, b1p=b1p, b2p=b2p, l1=0, l2=0, l21=0)
    _assert_same_table(ops, hv, ov, ids, rtol=5e-6, atol=1e-7)
    _assert_same_table(ops, hs, os_, ids, rtol=5e-6, atol=1e-9)


def test_dedup_segment_sum_against_fp64(ops):
  """Heavy-hitter sums: the occurrence-ordered fp32 sum of the reference is itself ~1e-5 off for
  thousands of addends, so the fused reduce is bounded against the exact (fp64) sum instead."""
  rng = np.random.default_rng(66)
  D = 32
  h, _ = _const(ops, D, 0.0)
  ids = _zipf_ids(rng, 200000, 50000)
  grad = rng.normal(0, 1e-2, (ids.size, D)).astype(np.float32)
  u, s, inv = ops.kv_dedup_segment_sum(h, ids, grad)
  u, s, inv = _np(u), _np(s), _np(inv)
  assert np.array_equal(np.sort(u), np.unique(ids))
  assert np.array_equal(u[inv], ids)                  # inverse maps every position to its key: bit-exact
  exact = np.zeros((u.size, D), np.float64)
  np.add.at(exact, inv, grad.astype(np.float64))
  absg = np.zeros((u.size, D), np.float64)
  np.add.at(absg, inv, np.abs(grad.astype(np.float64)))
  cnt = np.bincount(inv, minlength=u.size)[:, None]
  # |fl(sum) - sum| <= gamma_n * sum|g| for ANY summation order
  assert np.all(np.abs(s - exact) <= (cnt * 6e-8 * 1.01) * absg + 1e-30)
  single = cnt[:, 0] == 1
  np.testing.assert_array_equal(s[single], exact[single].astype(np.float32))  # singletons are copies


# ---------------------------------------------------------------------------------------------
# full-size properties (BASELINE config 2 shape, smaller key space so it fits a test run)
# ---------------------------------------------------------------------------------------------
def test_full_size_batch_properties(ops):
  rng = np.random.default_rng(20250213)
  D, N, KEYS = 32, 1 << 20, 2_000_000
  table = rng.standard_normal((10000, D)).astype(np.float32)
  h = ops.kv_variable([D], capacity_hint=KEYS)
  ops.kv_set_clock_days(h, DAY)
  ops.init_kv_variable_v2(h, table)
  s = ops.kv_variable([3 * D], capacity_hint=KEYS)
  ops.init_kv_variable_v2(s, np.zeros((16, 3 * D), np.float32))
  ids = torch.from_numpy(_zipf_ids(rng, N, KEYS)).cuda()
  out1 = ops.kv_variable_gather_or_insert_v2(h, ids)
  out2 = ops.kv_variable_gather_or_insert_v2(h, ids)
  assert torch.equal(out1, out2)                       # idempotent
  uniq, inv = torch.unique(ids, return_inverse=True)
  first = torch.full((uniq.numel(),), N, device="cuda", dtype=torch.int64).scatter_reduce(
      0, inv, torch.arange(N, device="cuda"), "amin")
  assert torch.equal(out1, out1[first][inv])           # every occurrence of a key reads the same row
  assert ops.kv_variable_shape_v2(h)[0] == uniq.numel()
  assert ops.kv_variable_frequency(h) == int(torch.clamp(torch.bincount(inv) * 2, max=65535).sum())
  assert torch.equal(ops.kv_variable_gather_or_zeros_v2(h, ids), out1)
  # one fused GroupAdam step == the same step on (unique ids, fp64-exact summed grads) within 1e-5
  grad = torch.from_numpy(rng.normal(0, 1e-2, (N, D)).astype(np.float32)).cuda()
  ops.kv_variable_group_sparse_apply_adam_v4(h, s, grad, ids, 1e-3, 0.9, 0.999, 0.9, 0.999, 1e-8, 0, 0, 0)
  gsum = torch.zeros((uniq.numel(), D), dtype=torch.float64, device="cuda").index_add_(0, inv, grad.double())
  x0 = out1[first].double()
  omb1 = float(np.float32(1) - np.float32(0.9))      # hyper-parameters are fp32 at the op boundary
  omb2 = float(np.float32(1) - np.float32(0.999))
  alpha = float(np.float32(1e-3)) * np.sqrt(omb2) / omb1
  expect = x0 - alpha * (omb1 * gsum) / ((omb2 * gsum * gsum).sqrt() + float(np.float32(1e-8)))
  got = ops.kv_variable_gather_or_zeros_v2(h, uniq).double()
  # fp32 summation in any order is off the exact sum by at most (count-1) * 2^-24 * sum|g_i|; where
  # the summed gradient cancels to ~epsilon the Adam quotient amplifies that by
  # d(update)/dg = alpha*omb1*eps / (sqrt(omb2)|g| + eps)^2, so the bound is per element
  cnt = torch.bincount(inv).double().unsqueeze(1)
  gabs = torch.zeros_like(gsum).index_add_(0, inv, grad.double().abs())
  dg = (cnt - 1).clamp(min=0) * 2.0 ** -24 * gabs
  eps32 = float(np.float32(1e-8))
  gmin = (gsum.abs() - dg).clamp(min=0)
  slope = alpha * omb1 * eps32 / (np.sqrt(omb2) * gmin + eps32) ** 2
  bound = 1e-6 + slope * dg
  excess = ((got - expect).abs() - bound).max().item()
  assert excess < 0, excess
  assert (bound > 2e-6).double().mean().item() < 1e-4     # the amplified cases are rare
  # export round trip: every key once, rows equal to a lookup
  k, vals = ops.read_kv_variable_op_v2(h)
  assert k.numel() == uniq.numel() and torch.equal(torch.sort(k).values, uniq)
  assert torch.equal(ops.kv_variable_gather_or_zeros_v2(h, k), vals)


# ---------------------------------------------------------------------------------------------
# deterministic reduction mode (kv_set_deterministic): the same batch gives bit-identical state
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D", [8, 32, 64])
def test_deterministic_mode_is_bit_reproducible(ops, D):
  rng = np.random.default_rng(77 + D)
  table = rng.standard_normal((64, D)).astype(np.float32)
  N = 300_000
  ids = torch.from_numpy(_zipf_ids(rng, N, 40_000)).cuda()
  grads = [torch.from_numpy(rng.normal(0, 1e-2, (N, D)).astype(np.float32)).cuda() for _ in range(2)]   # two-signed
  states = []
  for rep in range(3):
    hv = ops.kv_variable([D]); hs = ops.kv_variable([3 * D])
    for h, t in ((hv, table), (hs, np.zeros((4, 3 * D), np.float32))):
      ops.kv_set_clock_days(h, DAY); ops.kv_set_seed(h, 5); ops.init_kv_variable_v2(h, t)
      ops.kv_set_deterministic(h, True)
    b1p, b2p = np.float32(0.9), np.float32(0.999)
    for g in grads:
      if rep < 2:
        out = ops.kv_variable_gather_or_insert_v2(hv, ids)          # batch token path
        ops.kv_variable_group_sparse_apply_adam_v4(hv, hs, g, ids, 1e-2, b1p, b2p, 0.9, 0.999, 1e-8, 0, 0, 0)
      else:                                                          # the apply builds the index itself
        ops.kv_variable_gather_or_insert_v2(hv, ids.clone())
        ops.kv_variable_group_sparse_apply_adam_v4(hv, hs, g, ids.clone(), 1e-2, b1p, b2p, 0.9, 0.999, 1e-8, 0, 0, 0)
      b1p, b2p = np.float32(b1p * np.float32(0.9)), np.float32(b2p * np.float32(0.999))
    u = torch.unique(ids)
    states.append((ops.kv_variable_gather_or_zeros_v2(hv, u), ops.kv_variable_gather_or_zeros_v2(hs, u)))
  for x, y in zip(states[0], states[1]):
    assert torch.equal(x, y)                                         # run to run
  for x, y in zip(states[0], states[2]):
    assert torch.equal(x, y)                                         # with and without the lookup's index
