"""Slot mirrors (csrc/kv_device.h SlotMirror; kvhip.hip mirror_*): the lean GroupAdam / Adagrad update keeps a write-back copy
of the slot row's frequency word and flags in the VAR row's own 32-byte record unit — one 128-byte line less to read and one
32-byte request less to write per key (profiles/r06_fetch_calibration.txt) — and every other op that enters either table
first flushes the dirty copies back and invalidates them all (one epoch number).  The reference has no such thing
(kernels/training_ops.cc:7142-7197 reads and writes both tables' records for every id: kv_variable.h:382-416); what must hold
is that NO op can tell: every test below interleaves mirror applies with ops that read or write the slot table's own
records, or free / reuse the var's rows, and compares both tables — rows, frequency words, flags, sizes — with the CPU
oracle, which knows nothing of mirrors.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu

from oracle import kv_oracle as ko  # noqa: E402  (checker only)
from test_gpu_parity import _pair, _const, _np, _beta_pows, _assert_same_table, RTOL, DAY  # noqa: E402
from test_gpu_unique_apply import _run, _oracle  # noqa: E402


@pytest.fixture(scope="module")
def ops():
  if not torch.cuda.is_available():
    pytest.skip("needs a GPU")
  from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as g
  return g


def _mk(ops, name, D, seed, cap=200_000):
  rng = np.random.default_rng(seed)
  hv, ov = _pair(ops, D, seed=seed, rng=rng, cap=cap)
  val = 0.0 if name.startswith("adam") else 0.1
  mult = 3 if name.startswith("adam") else 1
  table = np.full((16, mult * D), val, np.float32)
  hs = ops.kv_variable([mult * D], capacity_hint=cap)
  ops.kv_set_clock_days(hs, DAY); ops.kv_set_seed(hs, 0); ops.init_kv_variable_v2(hs, table)
  os_ = ko.OracleKv(mult * D, 0, table, day=DAY, picker=1, seed=0)
  return (hv, hs), (ov, os_)


def _step(ops, name, hs, os_, rng, t, ids, unique=False):
  D = hs[0].dim
  grad = rng.normal(0, 1e-2, (ids.size, D)).astype(np.float32)
  b1p, b2p = _beta_pows(t)
  kw = dict(lr=0.05, b1p=b1p, b2p=b2p)
  _run(ops, name, hs, grad, ids, unique, **kw)
  if unique:
    _oracle(name, os_, grad, ids, **kw)
  else:
    u, s, _ = ko.dedup_segment_sum(ids, grad)
    _oracle(name, os_, s, u, **kw)


def _check(ops, hs, os_, keys):
  for h, o in zip(hs, os_):
    _assert_same_table(ops, h, o, keys, rtol=RTOL, atol=1e-7)


@pytest.mark.parametrize("name,D", [("adam4", 32), ("adam3", 8), ("adagrad", 64), ("adam4", 100)])
def test_lean_applies_use_mirrors_and_nobody_can_tell(ops, name, D):
  """steps of unique ids (exact sums): the first apply of a key takes the general path and leaves a mirror, the later ones
  work on it; the slot table is only LOOKED AT (records and rows against the oracle: an op on it, i.e. an epoch end) every
  third step, so dirty mirrors live across several applies and are flushed when somebody asks"""
  rng = np.random.default_rng(11 + D)
  hs, os_ = _mk(ops, name, D, seed=11 + D)
  seen = []
  for t in range(9):
    ids = rng.choice(3000, 1200, replace=False).astype(np.int64) - 100
    seen.append(ids)
    if t % 2 == 0:
      want = os_[0].gather_or_insert(ids)                     # a training lookup in between (keeps the epoch)
      np.testing.assert_array_equal(_np(ops.kv_variable_gather_or_insert_v2(hs[0], ids)), want)
    _step(ops, name, hs, os_, rng, t, ids, unique=(t % 3 == 1))
    if t == 1:   # two plain steps (lookup, apply, apply): the epoch that began when the pair formed is still running
      assert ops.kv_get_stat(hs[0], ops.KV_STAT_MIRROR_EPOCHS) == 1
    if t % 3 == 2:
      e0 = ops.kv_get_stat(hs[0], ops.KV_STAT_MIRROR_EPOCHS)
      _check(ops, hs, os_, np.concatenate(seen))
      assert ops.kv_get_stat(hs[0], ops.KV_STAT_MIRROR_EPOCHS) > e0      # somebody looked: the copies went back
  e1 = ops.kv_get_stat(hs[0], ops.KV_STAT_MIRROR_EPOCHS)
  assert ops.kv_get_stat(hs[0], ops.KV_STAT_MIRROR_APPLIES) == 9
  _check(ops, hs, os_, np.concatenate(seen))
  assert ops.kv_get_stat(hs[0], ops.KV_STAT_MIRROR_EPOCHS) > e1


def test_ops_on_the_slot_table_between_applies(ops):
  """the slot table's own records change between two mirror applies — a training lookup on the SLOT table (its frequency
  words move), a scatter into it, a delete of some of its keys: the next apply must start from those records, not from its
  stale copies"""
  name, D = "adam4", 16
  rng = np.random.default_rng(5)
  hs, os_ = _mk(ops, name, D, seed=5)
  ids = np.arange(2000, dtype=np.int64) * 3 - 700
  for t in range(3):
    _step(ops, name, hs, os_, rng, t, ids, unique=True)
  # (a) a lookup on the slot table itself: every occurrence counts in ITS frequency words
  some = ids[::7]
  np.testing.assert_array_equal(_np(ops.kv_variable_gather_or_insert_v2(hs[1], some)), os_[1].gather_or_insert(some))
  _step(ops, name, hs, os_, rng, 3, ids, unique=True)
  _check(ops, hs, os_, ids)
  # (b) slot rows overwritten from outside
  upd = rng.normal(0, 1e-3, (some.size, 3 * D)).astype(np.float32)
  ops.kv_variable_scatter_update_v2(hs[1], some, upd)
  os_[1].scatter_update(some, upd, 0)
  _step(ops, name, hs, os_, rng, 4, ids, unique=False)
  _check(ops, hs, os_, ids)
  # (c) keys deleted from the slot table only: their slot rows start over at the next apply, the freed rows are reused
  gone = ids[5::11]
  assert ops.kv_variable_delete(hs[1], gone) == os_[1].delete(gone)
  fresh = np.arange(90_000, 90_400, dtype=np.int64)
  _step(ops, name, hs, os_, rng, 5, np.concatenate([ids, fresh]), unique=True)
  _check(ops, hs, os_, np.concatenate([ids, fresh]))


def test_var_rows_released_and_reused(ops):
  """keys deleted from the VAR (their rows go to the free list with whatever their mirror units hold), other keys inserted
  into those rows, then applies: a reused row must not inherit the old key's mirror"""
  name, D = "adagrad", 32
  rng = np.random.default_rng(6)
  hs, os_ = _mk(ops, name, D, seed=6)
  ids = np.arange(3000, dtype=np.int64)
  for t in range(2):
    _step(ops, name, hs, os_, rng, t, ids, unique=True)
  gone = ids[::3]
  assert ops.kv_variable_delete(hs[0], gone) == os_[0].delete(gone)
  new = np.arange(50_000, 51_000, dtype=np.int64)
  np.testing.assert_array_equal(_np(ops.kv_variable_gather_or_insert_v2(hs[0], new)), os_[0].gather_or_insert(new))
  allk = np.concatenate([ids, new])
  for t in range(2, 5):
    batch = np.concatenate([ids[1::3], ids[2::3], new, gone[:200]])        # (some deleted keys come back, too)
    _step(ops, name, hs, os_, rng, t, batch, unique=(t == 3))
    _check(ops, hs, os_, allk)


def test_two_vars_on_one_slot_table_and_a_change_of_slot_table(ops):
  """a slot table attached by a second var serves both without mirrors from then on; a var that moves to another slot table
  hands its dirty copies back to the first"""
  name, D = "adam4", 16
  rng = np.random.default_rng(8)
  (hv1, hs1), (ov1, os1) = _mk(ops, name, D, seed=8)
  (hv2, hs2), (ov2, os2) = _mk(ops, name, D, seed=9)
  ids = np.arange(1500, dtype=np.int64) - 200
  for t in range(2):
    _step(ops, name, (hv1, hs1), (ov1, os1), rng, t, ids, unique=True)
  # var 1 moves on to slot table 2 (its mirrors for slot table 1 are flushed), var 2 joins slot table 2 as well
  for t in range(2, 4):
    _step(ops, name, (hv1, hs2), (ov1, os2), rng, t, ids, unique=True)
  ids2 = ids + 700                                                           # (overlapping keys: both vars hold slot rows of them)
  for t in range(4, 6):
    _step(ops, name, (hv2, hs2), (ov2, os2), rng, t, ids2, unique=True)
    _step(ops, name, (hv1, hs2), (ov1, os2), rng, t, ids, unique=False)
  allk = np.concatenate([ids, ids2])
  for h, o in ((hv1, ov1), (hs1, os1), (hv2, ov2), (hs2, os2)):
    _assert_same_table(ops, h, o, allk, rtol=RTOL, atol=1e-7)


def test_batched_applies_share_the_mirrors_with_the_single_table_ops(ops):
  """kv_multi_apply_* over the same pairs as the single-table applies: both are lean, both work on the mirrors"""
  D, T = 16, 4
  rng = np.random.default_rng(21)
  pairs = [_mk(ops, "adam4", D, seed=30 + j) for j in range(T)]
  hv = [p[0][0] for p in pairs]; hsl = [p[0][1] for p in pairs]
  ov = [p[1][0] for p in pairs]; osl = [p[1][1] for p in pairs]
  keys = [np.arange(800 + 50 * j, dtype=np.int64) * 2 + j for j in range(T)]
  for t in range(4):
    b1p, b2p = _beta_pows(t)
    grads = [rng.normal(0, 1e-2, (k.size, D)).astype(np.float32) for k in keys]
    if t % 2 == 0:
      ops.kv_multi_group_sparse_apply_adam(hv, hsl, grads, keys, 0.05, b1p, b2p, 0.9, 0.999, 1e-8, 0.0, 0.0, 0.0, version=4)
    else:
      for j in range(T):
        ops.kv_variable_group_sparse_apply_adam_v4(hv[j], hsl[j], grads[j], keys[j], 0.05, b1p, b2p, 0.9, 0.999, 1e-8, 0.0, 0.0, 0.0,
                                                   unique_indices=True)
    for j in range(T):
      ko.apply_group_adam(ov[j], osl[j], grads[j], keys[j], 0.05, b1p, b2p, 0.9, 0.999, 1e-8)
  for j in range(T):
    _assert_same_table(ops, hv[j], ov[j], keys[j], rtol=RTOL, atol=1e-7)
    _assert_same_table(ops, hsl[j], osl[j], keys[j], rtol=RTOL, atol=1e-7)
    assert ops.kv_get_stat(hv[j], ops.KV_STAT_MIRROR_APPLIES) == 4


def test_the_epoch_number_wraps(ops):
  """65 536 epoch ends (a point query on the slot table each): the 16-bit epoch on the device wraps, every mirror is cleared,
  and the applies on both sides of the wrap agree with the oracle"""
  name, D = "adam4", 8
  rng = np.random.default_rng(13)
  hs, os_ = _mk(ops, name, D, seed=13, cap=20_000)
  ids = np.arange(500, dtype=np.int64)
  q = torch.tensor([1, 2, 3], dtype=torch.int64, device="cuda")
  for t in range(2):
    _step(ops, name, hs, os_, rng, t, ids, unique=True)
  e0 = ops.kv_get_stat(hs[0], ops.KV_STAT_MIRROR_EPOCHS)
  for k in range(66_000):
    ops.kv_get_meta(hs[1], q)
    if k in (20_000, 65_520, 65_540):
      _step(ops, name, hs, os_, rng, 2 + k % 7, ids, unique=True)            # an apply now and then: mirrors of several epochs
  assert ops.kv_get_stat(hs[0], ops.KV_STAT_MIRROR_EPOCHS) - e0 >= 66_000
  for t in range(3):
    _step(ops, name, hs, os_, rng, t, ids, unique=(t != 1))
  _check(ops, hs, os_, ids)


def test_stream_capture_gives_the_mirrors_up(ops):
  """No host code runs when a hipGraph replays, so nothing could flush or re-validate a mirror there.  kv_prepare_capture (the
  documented precondition of capturing training ops) dissolves the pair — the dirty copies go back, the table never pairs
  again — and an op that would have to end an epoch INSIDE a capture without it is refused, not silently stale."""
  from tfplus_amd import _lib
  name, D = "adam4", 16
  rng = np.random.default_rng(17)
  hs, os_ = _mk(ops, name, D, seed=17, cap=50_000)
  ids = np.arange(4000, dtype=np.int64) - 50
  for t in range(2):
    _step(ops, name, hs, os_, rng, t, ids, unique=True)
  assert ops.kv_get_stat(hs[0], ops.KV_STAT_MIRROR_APPLIES) == 2
  # (a) capturing a read of the SLOT table while its records' newest words live in the var's mirrors: refused
  q = torch.from_numpy(ids[:256]).cuda()
  side = torch.cuda.Stream()
  side.wait_stream(torch.cuda.current_stream())
  g = torch.cuda.CUDAGraph()
  with pytest.raises(_lib.KvError):
    with torch.cuda.graph(g, stream=side):
      ops.kv_variable_gather_or_zeros_v2(hs[1], q)
  torch.cuda.synchronize()
  # ... and so is an optimizer apply of the pair (its end-of-epoch flush would be recorded instead of run)
  dids = torch.from_numpy(ids).cuda()
  dgrad = torch.zeros((ids.size, D), device="cuda")
  g2 = torch.cuda.CUDAGraph()
  side.wait_stream(torch.cuda.current_stream())
  with pytest.raises(_lib.KvError):
    with torch.cuda.graph(g2, stream=side):
      ops.kv_variable_group_sparse_apply_adam_v4(hs[0], hs[1], dgrad, dids, 0.05, 0.5, 0.9, 0.9, 0.999, 1e-8, 0.0, 0.0, 0.0)
  torch.cuda.synchronize()
  # (b) behind kv_prepare_capture the pair is gone for good: the applies go on (general path), everything still agrees
  for h in hs:
    ops.kv_prepare_capture(h, 10_000)
  for t in range(2, 5):
    _step(ops, name, hs, os_, rng, t, ids, unique=(t != 3))
  _check(ops, hs, os_, ids)
  assert ops.kv_get_stat(hs[0], ops.KV_STAT_MIRROR_APPLIES) == 2


def test_a_query_on_another_stream_and_another_thread(ops):
  """the slot table is queried from a second host thread on a second stream while the first thread steps (lookup + apply) on its
  own stream, no synchronisation between them: the query ends the epoch behind the apply it follows (a flush that overtook
  the apply would orphan the copies that apply wrote), the steps go on; at the end both tables are the oracle's"""
  import threading
  name, D = "adam4", 16
  rng = np.random.default_rng(23)
  hs, os_ = _mk(ops, name, D, seed=23, cap=60_000)
  ids = np.arange(20_000, dtype=np.int64) - 1000
  STEPS = 25
  grads = [rng.normal(0, 1e-2, (ids.size, D)).astype(np.float32) for _ in range(STEPS)]
  errs, stop = [], threading.Event()

  def stepper():
    try:
      with torch.cuda.stream(torch.cuda.Stream()):
        dids = torch.from_numpy(ids).cuda()
        for t in range(STEPS):
          b1p, b2p = _beta_pows(t)
          ops.kv_variable_gather_or_insert_v2(hs[0], dids)
          ops.kv_variable_group_sparse_apply_adam_v4(hs[0], hs[1], torch.from_numpy(grads[t]).cuda(), dids, 0.05, b1p, b2p, 0.9, 0.999,
                                                     1e-8, 0.0, 0.0, 0.0, unique_indices=(t % 2 == 0))
        torch.cuda.current_stream().synchronize()
    except Exception as e:  # pragma: no cover
      errs.append(repr(e))
    finally:
      stop.set()

  def asker():
    try:
      with torch.cuda.stream(torch.cuda.Stream()):
        q = torch.from_numpy(ids[::97].copy()).cuda()
        while not stop.is_set():
          ops.kv_get_meta(hs[1], q)                 # a point query on the slot table: ends the running epoch
          ops.kv_variable_size_v2(hs[0])            # ... and one on the var
    except Exception as e:  # pragma: no cover
      errs.append(repr(e))

  th = [threading.Thread(target=stepper), threading.Thread(target=asker)]
  for t in th:
    t.start()
  for t in th:
    t.join()
  torch.cuda.synchronize()
  assert not errs, errs
  for t in range(STEPS):
    b1p, b2p = _beta_pows(t)
    os_[0].gather_or_insert(ids)
    ko.apply_group_adam(os_[0], os_[1], grads[t], ids, 0.05, b1p, b2p, 0.9, 0.999, 1e-8)
  _check(ops, hs, os_, ids)
  assert ops.kv_get_stat(hs[0], ops.KV_STAT_MIRROR_EPOCHS) > 2


def test_slot_queries_from_another_thread_while_the_var_grows(ops):
  """the stepper's batches move on (new keys every step: the var's index is rebuilt and both tables leave their first slab, so
  lean and general applies alternate) while a second thread keeps ending epochs from the SLOT side, under the slot table's
  lock alone: the flush goes through the pair's views kept in the slot table, never through the var's host state"""
  import threading
  name, D = "adam4", 16
  rng = np.random.default_rng(29)
  hs, os_ = _mk(ops, name, D, seed=29, cap=9_000)          # small slabs: growth on the way
  STEPS, N, SHIFT = 30, 6_000, 700
  batches = [np.arange(N, dtype=np.int64) + t * SHIFT - 500 for t in range(STEPS)]
  grads = [rng.normal(0, 1e-2, (N, D)).astype(np.float32) for _ in range(STEPS)]
  errs, stop = [], threading.Event()

  def stepper():
    try:
      with torch.cuda.stream(torch.cuda.Stream()):
        for t in range(STEPS):
          b1p, b2p = _beta_pows(t)
          dids = torch.from_numpy(batches[t]).cuda()
          ops.kv_variable_gather_or_insert_v2(hs[0], dids)
          ops.kv_variable_group_sparse_apply_adam_v4(hs[0], hs[1], torch.from_numpy(grads[t]).cuda(), dids, 0.05, b1p, b2p, 0.9, 0.999,
                                                     1e-8, 0.0, 0.0, 0.0)
        torch.cuda.current_stream().synchronize()
    except Exception as e:  # pragma: no cover
      errs.append(repr(e))
    finally:
      stop.set()

  def asker():
    try:
      with torch.cuda.stream(torch.cuda.Stream()):
        q = torch.arange(0, 20_000, 53, device="cuda")
        while not stop.is_set():
          ops.kv_get_meta(hs[1], q)
          ops.kv_variable_get_count_v2(hs[1], q)
    except Exception as e:  # pragma: no cover
      errs.append(repr(e))

  th = [threading.Thread(target=stepper), threading.Thread(target=asker)]
  for t in th:
    t.start()
  for t in th:
    t.join()
  torch.cuda.synchronize()
  assert not errs, errs
  for t in range(STEPS):
    b1p, b2p = _beta_pows(t)
    os_[0].gather_or_insert(batches[t])
    ko.apply_group_adam(os_[0], os_[1], grads[t], batches[t], 0.05, b1p, b2p, 0.9, 0.999, 1e-8)
  _check(ops, hs, os_, np.unique(np.concatenate(batches)))
