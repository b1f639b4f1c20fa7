"""embedding_lookup_sparse fused behind the lookup (kv_lookup_sparse) against
the oracle lookup + a NumPy restatement of TF's segment ops (embedding_ops.py:359-441)."""
import numpy as np
import pytest

from oracle import kv_oracle as ko

torch = pytest.importorskip("torch")
DAY = 20000


@pytest.fixture(scope="module")
def ops():
  if not torch.cuda.is_available():
    pytest.skip("needs a GPU")
  from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as g
  return g


def _pair(ops, D, thr=0, seed=3, rows=64):
  table = np.random.default_rng(seed).standard_normal((rows, D)).astype(np.float32)
  h = ops.kv_variable([D], enter_threshold=thr)
  ops.kv_set_clock_days(h, DAY)
  ops.kv_set_seed(h, seed)
  ops.init_kv_variable_v2(h, table)
  return h, ko.OracleKv(D, thr, table, day=DAY, picker=1, seed=seed)


def _tf_combine(emb, seg, w, nseg, combiner):
  """tf.segment_sum(emb * w) [/ segment_sum(w) | / sqrt(segment_sum(w^2))], fp32, position order;
  without weights tf.sparse_segment_{sum,mean,sqrt_n} (empty segment -> zeros)."""
  D = emb.shape[1]
  out = np.zeros((nseg, D), np.float32)
  ws = np.zeros(nseg, np.float32)
  w2 = np.zeros(nseg, np.float32)
  ww = np.ones(len(seg), np.float32) if w is None else w.astype(np.float32)
  for j, s in enumerate(seg):
    out[s] = out[s] + emb[j] * ww[j]
    ws[s] = np.float32(ws[s] + ww[j])
    w2[s] = np.float32(w2[s] + np.float32(ww[j] * ww[j]))
  if combiner == "sum":
    return out
  den = ws if combiner == "mean" else np.sqrt(w2)
  with np.errstate(invalid="ignore", divide="ignore"):
    res = out / den[:, None]
  if w is None:
    res[den == 0] = 0.0
  return res.astype(np.float32)


def _ragged(rng, nseg, maxlen, keyspace):
  lens = rng.integers(0, maxlen + 1, nseg)
  lens[rng.integers(0, nseg)] = 0                      # at least one empty segment in the middle
  lens[-1] = max(lens[-1], 1)                          # dense_shape[0] = last segment + 1
  seg = np.repeat(np.arange(nseg), lens)
  ids = rng.integers(-keyspace, keyspace, seg.size)
  return ids, seg


@pytest.mark.gpu
@pytest.mark.parametrize("D", [32, 8, 5, 256, 12, 100])
@pytest.mark.parametrize("combiner", ["sum", "mean", "sqrtn"])
@pytest.mark.parametrize("weighted", [False, True])
def test_lookup_sparse_matches_oracle(ops, D, combiner, weighted):
  rng = np.random.default_rng(D * 7 + len(combiner) + weighted)
  h, o = _pair(ops, D, seed=5)
  for step in range(2):                                # step 0 inserts, step 1 finds
    ids, seg = _ragged(rng, 200, 9, 150)
    nseg = 200
    w = rng.uniform(0.1, 2.0, ids.size).astype(np.float32) if weighted else None
    got = ops.kv_variable_lookup_sparse(h, ids, seg, w, nseg, combiner, count_occurrences=False).cpu().numpy()
    uniq, idx = np.unique(ids, return_inverse=True)
    emb = o.gather_or_insert(uniq)[idx]
    want = _tf_combine(emb, seg, w, nseg, combiner)
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-7, equal_nan=True)
    assert ops.kv_variable_size_v2(h) == o.size() and ops.kv_variable_frequency(h) == o.sum_freq()
  k, v = ops.read_kv_variable_op_v2(h)
  assert dict(zip(k.cpu().numpy().tolist(), map(bytes, v.cpu().numpy()))) == \
      {kk: bytes(vv) for kk, vv in o.as_dict().items()}


@pytest.mark.gpu
@pytest.mark.parametrize("D,n", [(32, 3_000_000), (20, 700_001)])
def test_lookup_sparse_large_batch_cold_then_warm(ops, D, n):
  """One call over more ids than the sorted-position kernels take (2^21), every key new in the first call — a key
  repeated in several tiles is inserted by one of them and the others find its row through the index — then the same
  batch again, found.  Against the rows read back + a float64 segment sum; frequency = one per distinct key and call."""
  dev = torch.device("cuda", 0)
  gen = torch.Generator(device=dev).manual_seed(D)
  h = ops.kv_variable([D], capacity_hint=2 * n)
  ops.kv_set_seed(h, 11)
  ops.init_kv_variable_v2(h, torch.randn(257, D, device=dev, generator=gen))
  ids = torch.randint(-(1 << 40), 1 << 40, (n,), device=dev, generator=gen)
  ids[1::3] = ids[0::3][: ids[1::3].numel()].flip(0)          # a third of the positions repeat a key from far away
  seg = (torch.arange(n, device=dev) // 3).to(torch.int64)
  nseg = (n + 2) // 3
  w = torch.rand(n, device=dev, generator=gen) + 0.5
  distinct = int(torch.unique(ids).numel())
  for call in range(2):
    got = ops.kv_variable_lookup_sparse(h, ids, seg, w, nseg, "sum", count_occurrences=False)
    assert ops.kv_variable_size_v2(h) == distinct and ops.kv_variable_frequency(h) == distinct * (call + 1)
    rows = torch.cat([ops.kv_variable_gather_or_zeros_v2(h, ids[i:i + (1 << 20)]) for i in range(0, n, 1 << 20)])
    want = torch.zeros((nseg, D), device=dev, dtype=torch.float64).index_add_(0, seg, rows.double() * w.double()[:, None])
    torch.testing.assert_close(got.double(), want, rtol=2e-6, atol=1e-6)
  from tfplus_amd import _lib
  with pytest.raises(_lib.InvalidArgumentError):             # the limit of one call is stated, not silently chunked
    big = torch.zeros((1 << 23) + 1, dtype=torch.int64, device=dev)
    ops.kv_variable_lookup_sparse(h, big, big, None, 1, "sum")


@pytest.mark.gpu
def test_lookup_sparse_counts_every_occurrence_when_asked(ops):
  """enter_threshold > 0 -> unique_with_counts + GatherOrInsertWithCounts (embedding_ops.py:362-382)."""
  h, o = _pair(ops, 16, thr=3, seed=9)
  rng = np.random.default_rng(4)
  ids, seg = _ragged(rng, 64, 12, 20)
  ops.kv_variable_lookup_sparse(h, ids, seg, None, 64, "mean", count_occurrences=True)
  uniq, cnt = np.unique(ids, return_counts=True)
  o.gather_or_insert(uniq, cnt)
  assert ops.kv_variable_frequency(h) == o.sum_freq() and ops.kv_variable_size_v2(h) == o.size()
  for key in uniq[:10]:
    assert ops.kv_get_meta(h, [int(key)])[0]["freq"] == o.meta(int(key))["freq"]


@pytest.mark.gpu
def test_lookup_sparse_edges(ops):
  h, _ = _pair(ops, 8)
  out = ops.kv_variable_lookup_sparse(h, np.zeros(0, np.int64), np.zeros(0, np.int64), None, 5, "mean")
  assert out.shape == (5, 8) and float(out.abs().sum()) == 0.0
  one = ops.kv_variable_lookup_sparse(h, [7, 7, 7], np.array([2, 2, 2], np.int32), None, 3, "sqrtn")
  row = ops.kv_variable_gather_or_zeros_v2(h, [7])[0]
  torch.testing.assert_close(one[2], row * 3 / np.sqrt(np.float32(3)), rtol=1e-6, atol=0)
  assert float(one[:2].abs().sum()) == 0.0
  from tfplus_amd import _lib
  with pytest.raises(ValueError):
    ops.kv_variable_lookup_sparse(h, [1], [0], None, 1, "max")
  with pytest.raises(_lib.InvalidArgumentError):
    ops.kv_variable_lookup_sparse(h, [1, 2], [0], None, 1, "sum")


@pytest.mark.gpu
def test_embedding_lookup_sparse_gradient_matches_unfused():
  """The fused op's IndexedSlices gradient == the torch-op chain's, and one GroupAdam step agrees."""
  from tfplus_amd.kv_variable.python.ops import embedding_ops, kv_variable_ops, variable_scope
  from tfplus_amd.kv_variable.python import training
  kv_variable_ops.set_training(True)
  rng = np.random.default_rng(12)
  ids, seg = _ragged(rng, 50, 6, 40)
  w = rng.uniform(0.5, 1.5, ids.size).astype(np.float32)
  ind = np.stack([seg, np.zeros_like(seg)], 1)
  results = []
  for fused in (True, False):
    variable_scope.reset_default_store()
    var = variable_scope.get_kv_variable("t_%d" % fused, embedding_dim=16, initializer=variable_scope.ones_initializer())
    sp = embedding_ops.SparseTensor(ind, ids, [50, 6])
    spw = embedding_ops.SparseTensor(ind, w, [50, 6])
    emb = embedding_ops.embedding_lookup_sparse(var, sp, spw, combiner="mean") if fused else \
        _unfused(embedding_ops, var, sp, spw)
    loss = (emb * torch.arange(16, device=emb.device)).sum()
    loss.backward()
    g = var.pop_gradients()
    # the fused op reports one slice per occurrence, the op chain one per distinct id (gather's
    # gradient already summed them): compare after the optimizer's own dedup (unique + segment sum)
    u, inv = torch.unique(g.indices, return_inverse=True)
    summed = torch.zeros((u.numel(), 16), device=g.values.device).index_add_(0, inv, g.values)
    results.append((emb.detach().cpu(), u.cpu(), summed.cpu()))
  torch.testing.assert_close(results[0][0], results[1][0], rtol=1e-6, atol=1e-7, equal_nan=True)  # empty rows: 0/0
  assert torch.equal(results[0][1], results[1][1])
  torch.testing.assert_close(results[0][2], results[1][2], rtol=1e-5, atol=1e-6)


def _unfused(embedding_ops, var, sp, spw):
  """The reference's op chain spelled out with torch ops (what embedding_lookup_sparse falls back to)."""
  dev = var.device
  seg = torch.as_tensor(sp.indices).to(dev)[:, 0]
  ids = torch.as_tensor(sp.values).to(dev)
  uniq, idx = torch.unique(ids, return_inverse=True)
  emb = var.sparse_read(uniq).index_select(0, idx)
  wts = torch.as_tensor(spw.values, dtype=emb.dtype).to(dev).reshape(-1, 1)
  nseg = int(seg.max()) + 1
  summed = torch.zeros((nseg, emb.shape[1]), device=dev).index_add(0, seg, emb * wts)
  return summed / torch.zeros((nseg, 1), device=dev).index_add(0, seg, wts)
