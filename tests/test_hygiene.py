"""GetCount / GetTimeStamp / Delete / DeleteWithTimestamp
(kv_variable.h:503-561,737-789) — oracle known answers on the CPU, parity on the GPU."""
import numpy as np
import pytest

from oracle import kv_oracle as ko

DAY = 20000


def _oracle(D=8, thr=0, seed=3, rows=64, day=DAY):
  table = np.random.default_rng(seed).standard_normal((rows, D)).astype(np.float32)
  return table, ko.OracleKv(D, thr, table, day=day, picker=1, seed=seed)


def test_oracle_count_timestamp_delete():
  _, o = _oracle()
  o.gather_or_insert([1, 2, 2, 3, 3, 3])
  np.testing.assert_array_equal(o.get_count([1, 2, 3, 4]), [1, 2, 3, 0])          # absent -> 0
  np.testing.assert_array_equal(o.get_timestamp([1, 4]), [DAY, DAY])               # absent -> today
  assert o.delete([2, 2, 99]) == 1 and o.map_size() == 2
  np.testing.assert_array_equal(o.get_count([2]), [0])
  assert float(np.abs(o.gather_or_zeros([2])).sum()) == 0.0
  o.gather_or_insert([2])                                                           # comes back as a new key
  np.testing.assert_array_equal(o.get_count([2]), [1])


def test_oracle_delete_with_timestamp():
  _, o = _oracle()
  o.gather_or_insert([1, 2])
  o.set_day(DAY + 3)
  o.gather_or_insert([3])
  o.insert([4], np.ones((1, 8), np.float32))               # InsertOrUpdate: day stamp 0 -> never expires
  o.set_day(DAY + 7)
  assert sorted(o.delete_with_timestamp(7).tolist()) == [1, 2]     # 7 days old: >= threshold
  assert sorted(o.as_dict()) == [3, 4]
  assert o.delete_with_timestamp(7).size == 0


# ----------------------------------------------------------------------------------------------------
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ops():
  if not torch.cuda.is_available():
    pytest.skip("needs a GPU")
  from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as g
  return g


def _pair(ops, D=8, thr=0, seed=3, rows=64, cap=0):
  table, o = _oracle(D, thr, seed, rows)
  h = ops.kv_variable([D], enter_threshold=thr, capacity_hint=cap)
  ops.kv_set_clock_days(h, DAY)
  ops.kv_set_seed(h, seed)
  ops.init_kv_variable_v2(h, table)
  return h, o


def _same_table(ops, h, o):
  k, v = ops.read_kv_variable_op_v2(h)
  got = dict(zip(k.cpu().numpy().tolist(), map(bytes, v.cpu().numpy())))
  assert got == {kk: bytes(vv) for kk, vv in o.as_dict().items()}
  assert ops.kv_variable_size_v2(h) == o.size() and ops.kv_variable_frequency(h) == o.sum_freq()
  assert ops.kv_variable_shape_v2(h)[0] == o.map_size()


@pytest.mark.gpu
def test_count_and_timestamp_parity(ops):
  h, o = _pair(ops)
  rng = np.random.default_rng(0)
  ids = rng.integers(-30, 30, 500)
  ops.kv_variable_gather_or_insert_v2(h, ids); o.gather_or_insert(ids)
  q = np.arange(-40, 40).reshape(8, 10)                     # output keeps the shape of indices
  c = ops.kv_variable_get_count_v2(h, q)
  assert c.dtype == torch.int32 and tuple(c.shape) == (8, 10)
  np.testing.assert_array_equal(c.cpu().numpy(), o.get_count(q))
  np.testing.assert_array_equal(ops.kv_variable_get_time_stamp(h, q).cpu().numpy(), o.get_timestamp(q))
  assert ops.kv_variable_get_count_v2(h, np.zeros((0,), np.int64)).numel() == 0


@pytest.mark.gpu
def test_delete_parity_and_row_recycling(ops):
  h, o = _pair(ops, D=16, cap=4096)
  s_h, s_o = _pair(ops, D=48, seed=4, cap=4096)
  rng = np.random.default_rng(1)
  for rnd in range(6):
    ids = rng.integers(-300, 300, 700)
    np.testing.assert_array_equal(ops.kv_variable_gather_or_insert_v2(h, ids).cpu().numpy(), o.gather_or_insert(ids))
    g = rng.normal(0, 1e-2, (ids.size, 16)).astype(np.float32)
    ops.kv_variable_group_sparse_apply_adam_v4(h, s_h, g, ids, 1e-3, 0.9, 0.999, 0.9, 0.999, 1e-8, 0, 0, 0)
    u, inv = np.unique(ids, return_inverse=True)
    uu, summed, _ = ko.dedup_segment_sum(ids, g)
    ko.apply_group_adam(o, s_o, summed, uu, 1e-3, 0.9, 0.999, 0.9, 0.999, 1e-8)
    dele = rng.integers(-300, 300, 150)                     # with repeats and absent keys
    n_gpu = ops.kv_variable_delete(h, dele)
    assert n_gpu == o.delete(dele)
    assert ops.kv_variable_delete(s_h, dele[:50]) == s_o.delete(dele[:50])
    assert float(ops.kv_variable_gather_or_zeros_v2(h, dele).abs().sum()) == 0.0
    np.testing.assert_array_equal(ops.kv_variable_get_count_v2(h, dele).cpu().numpy(), 0)
  # rows were recycled: the slab never grew past the largest live set (+ one batch)
  k, v = ops.read_kv_variable_op_v2(h)
  got = {int(a): b for a, b in zip(k.cpu().numpy(), v.cpu().numpy())}
  want = o.as_dict()
  assert set(got) == set(want)
  for kk in want:
    np.testing.assert_allclose(got[kk], want[kk], rtol=2e-6, atol=1e-7)
  assert ops.kv_variable_shape_v2(h)[0] == o.map_size() and ops.kv_variable_frequency(h) == o.sum_freq()
  assert ops.kv_variable_shape_v2(s_h)[0] == s_o.map_size()


@pytest.mark.gpu
def test_delete_then_index_rebuild_and_growth(ops):
  """Tombstones must not survive a rebuild, freed rows must not be re-indexed, and a table that
  keeps deleting and inserting fresh keys keeps working past its initial capacity."""
  h, o = _pair(ops, D=4, cap=256)
  nxt = 0
  for rnd in range(40):
    ids = np.arange(nxt, nxt + 200); nxt += 200
    ops.kv_variable_gather_or_insert_v2(h, ids); o.gather_or_insert(ids)
    kill = ids[::2] if rnd % 3 else ids                    # some rounds delete everything they added
    assert ops.kv_variable_delete(h, kill) == o.delete(kill)
  _same_table(ops, h, o)
  probe = np.arange(0, nxt, 7)
  np.testing.assert_array_equal(ops.kv_variable_gather_or_zeros_v2(h, probe).cpu().numpy(), o.gather_or_zeros(probe))
  # the EMPTY-sentinel key and a revived key
  big = np.array([np.iinfo(np.int64).min, 5, 5], np.int64)
  ops.kv_variable_gather_or_insert_v2(h, big); o.gather_or_insert(big)
  assert ops.kv_variable_delete(h, big) == o.delete(big)
  ops.kv_variable_gather_or_insert_v2(h, big); o.gather_or_insert(big)
  _same_table(ops, h, o)


@pytest.mark.gpu
def test_delete_with_timestamp_parity(ops):
  h, o = _pair(ops, D=8)
  ops.kv_variable_gather_or_insert_v2(h, np.arange(0, 100)); o.gather_or_insert(np.arange(0, 100))
  ops.kv_set_clock_days(h, DAY + 4); o.set_day(DAY + 4)
  ops.kv_variable_gather_or_insert_v2(h, np.arange(50, 150)); o.gather_or_insert(np.arange(50, 150))
  vals = np.ones((10, 8), np.float32)
  ops.kv_variable_insert_v2(h, np.arange(200, 210), vals); o.insert(np.arange(200, 210), vals)   # day 0: kept
  ops.kv_set_clock_days(h, DAY + 9); o.set_day(DAY + 9)
  gone = ops.kv_variable_delete_with_timestamp(h, 7)
  assert sorted(gone.cpu().numpy().tolist()) == sorted(o.delete_with_timestamp(7).tolist()) == list(range(0, 50))
  _same_table(ops, h, o)
  assert ops.kv_variable_delete_with_timestamp(h, 7).numel() == 0
  gone = ops.kv_variable_delete_with_timestamp(h, 5)
  assert sorted(gone.cpu().numpy().tolist()) == sorted(o.delete_with_timestamp(5).tolist()) == list(range(50, 150))
  _same_table(ops, h, o)


@pytest.mark.gpu
def test_batch_gather_or_zeros(ops):
  """BatchKvVariableGatherOrZerosV2 == per-table GatherOrZeros, mixed dims / key types / shapes."""
  rng = np.random.default_rng(8)
  hs, idl, want = [], [], []
  for j, (D, kd, n) in enumerate([(8, torch.int64, 300), (64, torch.int64, 2048), (5, torch.int32, 77), (128, torch.int64, 0),
                                  (32, torch.int64, 1)]):
    h = ops.kv_variable([D], key_dtype=kd)
    ops.init_kv_variable_v2(h, rng.standard_normal((32, D)).astype(np.float32))
    ops.kv_variable_gather_or_insert_v2(h, np.arange(-20, 60))
    ids = rng.integers(-40, 90, (n,)) if n != 300 else rng.integers(-40, 90, (10, 30))
    hs.append(h); idl.append(ids); want.append(ops.kv_variable_gather_or_zeros_v2(h, ids))
  for rep in range(3):                                    # the descriptor staging buffer is reused
    got = ops.batch_kv_variable_gather_or_zeros_v2(hs, idl)
    assert [tuple(g.shape) for g in got] == [tuple(w.shape) for w in want]
    assert all(torch.equal(g, w) for g, w in zip(got, want))
  assert ops.kv_variable_frequency(hs[0]) == 80           # inference lookups count nothing


@pytest.mark.gpu
def test_create_use_destroy_releases_device_memory(ops):
  """Every buffer a table owns (index, slab chunks, workspace, free list, scratch) goes away with it."""
  import gc
  rng = np.random.default_rng(0)

  def cycle():
    h = ops.kv_variable([32], capacity_hint=200_000)
    s = ops.kv_variable([96], capacity_hint=200_000)
    ops.init_kv_variable_v2(h, rng.standard_normal((16, 32)).astype(np.float32))
    ops.init_kv_variable_v2(s, np.zeros((4, 96), np.float32))
    ids = rng.integers(0, 300_000, 100_000)
    ops.kv_variable_gather_or_insert_v2(h, ids)
    ops.kv_variable_group_sparse_apply_adam_v4(h, s, rng.standard_normal((ids.size, 32)).astype(np.float32), ids, 1e-3, 0.9,
                                               0.999, 0.9, 0.999, 1e-8, 0, 0, 0)
    ops.kv_variable_scatter_add_v2(h, ids[:1000], np.ones((1000, 32), np.float32))
    ops.kv_variable_delete(h, ids[:5000])
    ops.kv_variable_lookup_sparse(h, ids[:4000], np.repeat(np.arange(1000), 4), None, 1000, "mean")
    del h, s
  cycle()
  gc.collect(); torch.cuda.synchronize()
  free0 = torch.cuda.mem_get_info()[0]
  for _ in range(20):
    cycle()
  gc.collect(); torch.cuda.synchronize()
  free1 = torch.cuda.mem_get_info()[0]
  assert free0 - free1 < 64 << 20, (free0, free1)          # one cycle holds ~250 MB


@pytest.mark.gpu
def test_concurrent_host_threads_on_separate_tables_and_streams(ops):
  """OpKernel::Compute may run on several inter-op threads at once: tables are
  independent, the library takes a per-table lock, errors are thread-local."""
  import threading
  rng = np.random.default_rng(3)
  T, D, STEPS = 4, 16, 25
  data = [[(rng.integers(-200, 200, 3000), (rng.uniform(0.5, 1.5, (3000, D)) * 1e-2).astype(np.float32))
           for _ in range(STEPS)] for _ in range(T)]

  def make(j):
    v, s = ops.kv_variable([D]), ops.kv_variable([3 * D])
    ops.kv_set_seed(v, j); ops.kv_set_clock_days(v, DAY); ops.kv_set_clock_days(s, DAY)
    ops.init_kv_variable_v2(v, np.random.default_rng(j).standard_normal((32, D)).astype(np.float32))
    ops.init_kv_variable_v2(s, np.zeros((4, 3 * D), np.float32))
    ops.kv_set_deterministic(v, True); ops.kv_set_deterministic(s, True)
    return v, s

  def run(j, tabs, stream, errs):
    try:
      with torch.cuda.stream(stream):
        v, s = tabs
        for ids, g in data[j]:
          ops.kv_variable_gather_or_insert_v2(v, ids)
          ops.kv_variable_group_sparse_apply_adam_v4(v, s, g, ids, 1e-2, 0.9, 0.999, 0.9, 0.999, 1e-8, 0, 0, 0)
        stream.synchronize()
    except Exception as e:  # pragma: no cover
      errs.append(repr(e))
  par = [make(j) for j in range(T)]
  seq = [make(j) for j in range(T)]
  errs = []
  th = [threading.Thread(target=run, args=(j, par[j], torch.cuda.Stream(), errs)) for j in range(T)]
  for t in th:
    t.start()
  for t in th:
    t.join()
  assert not errs, errs
  for j in range(T):
    run(j, seq[j], torch.cuda.current_stream(), errs)
  assert not errs, errs
  for (pv, _), (sv, _) in zip(par, seq):
    kp, vp = ops.read_kv_variable_op_v2(pv); ks, vs = ops.read_kv_variable_op_v2(sv)
    op_, os_ = torch.argsort(kp), torch.argsort(ks)
    assert torch.equal(kp[op_], ks[os_]) and ops.kv_variable_frequency(pv) == ops.kv_variable_frequency(sv)
    assert torch.equal(vp[op_], vs[os_])                                 # deterministic mode: bit-identical whatever the thread / stream


@pytest.mark.gpu
def test_two_streams_on_one_table_are_serialised_by_the_library(ops):
  """The reference lets Compute run concurrently on one table under shared locks (training_ops.cc:96-184);
  here ops of one table run in issue order whatever their streams (the library makes a new stream wait for
  the table's previous one), so two host threads driving ONE table through two streams end in a state some
  serial order of their ops produces — compared with the oracle run in the order the ops were issued."""
  import threading
  rng = np.random.default_rng(9)
  D, STEPS = 16, 30
  v, s = ops.kv_variable([D]), ops.kv_variable([3 * D])
  table = rng.standard_normal((32, D)).astype(np.float32)
  for h, t in ((v, table), (s, np.zeros((4, 3 * D), np.float32))):
    ops.kv_set_seed(h, 2); ops.kv_set_clock_days(h, DAY); ops.init_kv_variable_v2(h, t)
  ov = ko.OracleKv(D, 0, table, day=DAY, picker=1, seed=2)
  os_ = ko.OracleKv(3 * D, 0, np.zeros((4, 3 * D), np.float32), day=DAY, picker=1, seed=2)
  issue = threading.Lock()      # the oracle sees the ops in the order they were issued to the library
  errs = []

  def lookups(stream):
    try:
      with torch.cuda.stream(stream):
        for k in range(STEPS):
          ids = np.random.default_rng(100 + k).integers(-300, 300, 2000)
          with issue:
            got = ops.kv_variable_gather_or_insert_v2(v, ids)
            want = ov.gather_or_insert(ids)
          np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-6, atol=1e-9)
    except Exception as e:  # pragma: no cover
      errs.append(repr(e))

  def applies(stream):
    try:
      with torch.cuda.stream(stream):
        for k in range(STEPS):
          r = np.random.default_rng(200 + k)
          u = np.unique(r.integers(-300, 300, 1500))
          g = r.normal(0, 1e-2, (u.size, D)).astype(np.float32)
          with issue:
            ops.kv_variable_group_sparse_apply_adam_v4(v, s, g, u, 1e-2, 0.9, 0.999, 0.9, 0.999, 1e-8, 0, 0, 0)
            ko.apply_group_adam(ov, os_, g, u, 1e-2, 0.9, 0.999, 0.9, 0.999, 1e-8)
    except Exception as e:  # pragma: no cover
      errs.append(repr(e))

  th = [threading.Thread(target=lookups, args=(torch.cuda.Stream(),)), threading.Thread(target=applies, args=(torch.cuda.Stream(),))]
  for t in th:
    t.start()
  for t in th:
    t.join()
  torch.cuda.synchronize()
  assert not errs, errs
  q = np.arange(-300, 300)
  np.testing.assert_allclose(ops.kv_variable_gather_or_zeros_v2(v, q).cpu().numpy(), ov.gather_or_zeros(q), rtol=1e-6, atol=1e-9)
  np.testing.assert_allclose(ops.kv_variable_gather_or_zeros_v2(s, q).cpu().numpy(), os_.gather_or_zeros(q), rtol=1e-6, atol=1e-9)
  assert ops.kv_variable_frequency(v) == ov.sum_freq() and ops.kv_variable_frequency(s) == os_.sum_freq()


@pytest.mark.gpu
def test_a_table_survives_the_stream_of_its_last_op(ops):
  """Ops of one table are ordered across streams by an event recorded on the PREVIOUS op's stream at the next op.  A stream
  the library owns (a communicator's) may be destroyed in between: kv_comm_destroy retires it from every table first.
  (Round 6: bench.py's sharded_world1 sub-record destroyed its communicator, the next lookup on the default stream crashed
  in hipEventRecord on the dead stream — under rocprofv3, silently wrong without it.)  kv_forget_stream is the same service
  for a stream the caller owns."""
  rng = np.random.default_rng(3)
  D = 16
  table = rng.standard_normal((32, D)).astype(np.float32)
  h = ops.kv_variable([D])
  ops.kv_set_seed(h, 4); ops.kv_set_clock_days(h, DAY); ops.init_kv_variable_v2(h, table)
  o = ko.OracleKv(D, 0, table, day=DAY, picker=1, seed=4)
  comm = ops.KvComm(1, 0, ops.kv_comm_unique_id(), 0)
  cs = comm.stream()
  ids = rng.integers(-500, 500, 3000)
  with torch.cuda.stream(cs):                      # the table's last op runs on the communicator's stream ...
    got = ops.kv_variable_gather_or_insert_v2(h, ids).cpu().numpy()      # (read back on that stream too)
  np.testing.assert_array_equal(got, o.gather_or_insert(ids))
  from tfplus_amd import _lib
  _lib.check(_lib.lib().kv_comm_destroy(comm.ptr)); comm.ptr = None     # ... which goes away
  ids2 = rng.integers(-500, 500, 3000)
  got = ops.kv_variable_gather_or_insert_v2(h, ids2)                    # the next op, on the default stream
  np.testing.assert_array_equal(got.cpu().numpy(), o.gather_or_insert(ids2))
  side = torch.cuda.Stream()
  with torch.cuda.stream(side):
    got = ops.kv_variable_gather_or_insert_v2(h, ids).cpu().numpy()
  np.testing.assert_array_equal(got, o.gather_or_insert(ids))
  ops.kv_forget_stream(side)                                            # the caller's own stream, about to be dropped
  del side
  got = ops.kv_variable_gather_or_insert_v2(h, ids2)
  np.testing.assert_array_equal(got.cpu().numpy(), o.gather_or_insert(ids2))
  assert ops.kv_variable_frequency(h) == o.sum_freq()
