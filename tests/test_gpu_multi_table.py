"""Batched launches over many tables (kv_multi_gather_or_insert / kv_multi_apply_group_adam): the
same kernels with grid.y = table, so every table must end bit-identical to one driven by the
single-table ops, and the first table is additionally checked against the oracle."""
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


def _tables(ops, n, D, seed):
  rng = np.random.default_rng(seed)
  out = []
  for j in range(n):
    table = rng.standard_normal((64, D)).astype(np.float32)
    pair = []
    for dim, tab in ((D, table), (3 * D, np.zeros((4, 3 * D), np.float32))):
      h = ops.kv_variable([dim])
      ops.kv_set_clock_days(h, DAY); ops.kv_set_seed(h, 3 + j); ops.init_kv_variable_v2(h, tab)
      ops.kv_set_deterministic(h, True)     # repeated ids are summed in an order fixed by the batch alone
      pair.append(h)
    out.append((pair[0], pair[1], table))
  return out


def _dump(ops, h):
  k, v = ops.read_kv_variable_op_v2(h)
  o = torch.argsort(k)
  return k[o].cpu(), v[o].cpu(), ops.kv_variable_frequency(h), ops.kv_variable_size_v2(h)


@pytest.mark.gpu
@pytest.mark.parametrize("D,sizes", [(32, [2048] * 6), (64, [2048, 1, 0, 5000, 300]), (8, [70000, 100]),
                                     (8, [2_500_000, 3000])])   # (more than 2^21 ids in one table of a batched call)
def test_multi_ops_equal_single_ops(ops, D, sizes):
  T = len(sizes)
  A = _tables(ops, T, D, seed=1)        # driven by the batched ops
  B = _tables(ops, T, D, seed=1)        # driven table by table
  ref = ko.OracleKv(D, 0, A[0][2], day=DAY, picker=1, seed=3)
  rslot = ko.OracleKv(3 * D, 0, np.zeros((4, 3 * D), np.float32), day=DAY)
  rng = np.random.default_rng(2)
  b1p, b2p = np.float32(0.9), np.float32(0.999)
  for step in range(3):
    ids = [rng.integers(-500, 500, n) for n in sizes]
    ids[0] = ids[0].reshape(-1, 1)                                   # output keeps the shape of indices
    # one-signed gradients: sums of repeated ids never cancel to ~epsilon, where Adam's quotient would
    # amplify the (order-dependent) fp32 rounding of the sum
    grads = [(rng.uniform(0.5, 1.5, (i.size, D)) * 1e-2 * rng.choice([-1, 1], (1, D))).astype(np.float32) for i in ids]
    outs = ops.kv_multi_gather_or_insert([a[0] for a in A], ids)
    for j in range(T):
      want = ops.kv_variable_gather_or_insert_v2(B[j][0], ids[j])
      assert tuple(outs[j].shape) == tuple(np.shape(ids[j])) + (D,)
      assert torch.equal(outs[j], want), (step, j)                 # same kernels, deterministic mode: bit-identical
    # against the oracle: the fused reduce sums repeated ids (one-signed here) in another order than TF-core
    np.testing.assert_allclose(outs[0].cpu().numpy(), ref.gather_or_insert(ids[0]), rtol=1e-5, atol=1e-6)
    ops.kv_multi_group_sparse_apply_adam([a[0] for a in A], [a[1] for a in A], grads, ids, 1e-2, b1p, b2p, 0.9, 0.999,
                                         1e-8, 0, 0, 0)
    for j in range(T):
      ops.kv_variable_group_sparse_apply_adam_v4(B[j][0], B[j][1], grads[j], ids[j].reshape(-1), 1e-2, b1p, b2p, 0.9,
                                                 0.999, 1e-8, 0, 0, 0)
    u, sm, _ = ko.dedup_segment_sum(ids[0].reshape(-1), grads[0])
    ko.apply_group_adam(ref, rslot, sm, u, 1e-2, b1p, b2p, 0.9, 0.999, 1e-8, 0, 0, 0)
    b1p, b2p = np.float32(b1p * np.float32(0.9)), np.float32(b2p * np.float32(0.999))
  for j in range(T):
    for which in (0, 1):
      ka, va, fa, sa = _dump(ops, A[j][which])
      kb, vb, fb, sb = _dump(ops, B[j][which])
      assert torch.equal(ka, kb) and fa == fb and sa == sb
      assert torch.equal(va, vb)                                     # same kernels, deterministic mode
  k0, v0, _, _ = _dump(ops, A[0][0])
  want = ref.as_dict()
  assert sorted(want) == k0.tolist()
  np.testing.assert_allclose(v0.numpy(), np.stack([want[k] for k in k0.tolist()]), rtol=1e-5, atol=1e-6)   # summation order (see above)


@pytest.mark.gpu
@pytest.mark.parametrize("opt", ["adam", "adagrad", "ftrl"])
def test_multi_ops_take_over_the_batched_lookups_index(ops, opt):
  """The batched lookup hands every table a batch token; the batched optimizer op given the very same id tensors skips
  its index pass (kv_multi_*_tok).  Same results as the single-table ops (which take the single lookup's token), bit for
  bit in deterministic mode; a stale token (an op in between, another tensor) falls back to indexing again."""
  D, sizes = 32, [3000, 2048, 1, 0, 9000, 40]
  T = len(sizes)
  rng = np.random.default_rng(5)
  nslot = {"adam": 1, "adagrad": 1, "ftrl": 2}[opt]

  def make():
    out = []
    for j in range(T):
      hs = []
      for si in range(1 + nslot):
        dim = 3 * D if (opt == "adam" and si == 1) else D
        h = ops.kv_variable([dim])
        ops.kv_set_clock_days(h, DAY); ops.kv_set_seed(h, 21 + j)
        tab = np.random.default_rng(90 + j).standard_normal((32, D)).astype(np.float32) if si == 0 \
            else np.full((4, dim), 0.1 if (opt != "adam" and si == 1) else 0.0, np.float32)
        ops.init_kv_variable_v2(h, tab)
        ops.kv_set_deterministic(h, True)
        hs.append(h)
      out.append(hs)
    return out

  def apply(multi, tabs, grads, ids):
    if opt == "adam":
      if multi:
        ops.kv_multi_group_sparse_apply_adam([a[0] for a in tabs], [a[1] for a in tabs], grads, ids, 1e-2, 0.9, 0.999, 0.9, 0.999, 1e-8, 0, 0, 0)
      else:
        for b, g, i in zip(tabs, grads, ids):
          ops.kv_variable_group_sparse_apply_adam_v4(b[0], b[1], g, i, 1e-2, 0.9, 0.999, 0.9, 0.999, 1e-8, 0, 0, 0)
    elif opt == "adagrad":
      if multi:
        ops.kv_multi_sparse_apply_adagrad([a[0] for a in tabs], [a[1] for a in tabs], 0.05, grads, ids)
      else:
        for b, g, i in zip(tabs, grads, ids):
          ops.kv_variable_sparse_apply_adagrad(b[0], b[1], 0.05, g, i, use_locking=True)
    else:
      if multi:
        ops.kv_multi_sparse_group_sparse_apply_ftrl([a[0] for a in tabs], [a[1] for a in tabs], [a[2] for a in tabs], grads, ids,
                                                    0.05, 1e-3, 1e-3, 1e-4, 0.0, -0.5)
      else:
        for b, g, i in zip(tabs, grads, ids):
          ops.kv_variable_sparse_group_sparse_apply_ftrl_v2(b[0], b[1], b[2], g, i, 0.05, 1e-3, 1e-3, 1e-4, 0.0, -0.5)

  A, B = make(), make()
  for step in range(4):
    ids = [torch.from_numpy(rng.integers(-400, 400, n)).cuda() for n in sizes]          # the SAME tensor objects go to both ops
    grads = [torch.from_numpy((rng.standard_normal((n, D)) * 1e-2).astype(np.float32)).cuda() for n in sizes]
    outs = ops.kv_multi_gather_or_insert([a[0] for a in A], ids)
    assert all(a[0].batch is not None for a, n in zip(A, sizes) if n > 0)               # tokens were handed out
    for j in range(T):
      assert torch.equal(outs[j], ops.kv_variable_gather_or_insert_v2(B[j][0], ids[j])), (step, j)
    if step == 2:     # an op in between makes one token stale: every table is indexed again, same result
      assert ops.kv_variable_size_v2(A[0][0]) == ops.kv_variable_size_v2(B[0][0])
      ops.kv_variable_delete(A[4][0], ids[4][:5]); ops.kv_variable_delete(B[4][0], ids[4][:5])
    if step == 3:     # other tensor objects with the same contents: no token is passed at all
      apply(True, A, grads, [i.clone() for i in ids])
    else:
      apply(True, A, grads, ids)
    apply(False, B, grads, ids)
  for a, b in zip(A, B):
    for ha, hb in zip(a, b):
      ka, va, fa, sa = _dump(ops, ha)
      kb, vb, fb, sb = _dump(ops, hb)
      assert torch.equal(ka, kb) and fa == fb and sa == sb
      assert torch.equal(va, vb)


@pytest.mark.gpu
def test_multi_ops_argument_checks(ops):
  from tfplus_amd import _lib
  (v8, s8, _), = _tables(ops, 1, 8, seed=5)
  (v16, s16, _), = _tables(ops, 1, 16, seed=5)
  with pytest.raises(_lib.InvalidArgumentError):
    ops.kv_multi_gather_or_insert([v8, v16], [[1], [2]])             # dims differ
  with pytest.raises(_lib.InvalidArgumentError):
    ops.kv_multi_gather_or_insert([v8, v8], [[1], [2]])              # listed twice
  with pytest.raises(_lib.InvalidArgumentError):
    ops.kv_multi_group_sparse_apply_adam([v8], [s16], [np.zeros((1, 8), np.float32)], [[1]], 0.1, 0.9, 0.999, 0.9, 0.999,
                                         1e-8, 0, 0, 0)
  raw = ops.kv_variable([8])
  with pytest.raises(_lib.FailedPreconditionError):
    ops.kv_multi_gather_or_insert([raw], [[1]])
  assert ops.kv_multi_gather_or_insert([v8], [np.zeros((0,), np.int64)])[0].shape == (0, 8)


@pytest.mark.gpu
def test_dcn_example_trains(ops):
  """configs[2] shape end to end: 26 KvVariables behind the batched ops + a torch dense tower."""
  import importlib.util
  import os
  spec = importlib.util.spec_from_file_location(
      "dcn_train", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "dcn_train.py"))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  losses = mod.main(["--steps", "120", "--batch_size", "1024"])
  assert np.isfinite(losses).all() and losses[-1] < 0.95 * losses[0], losses


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["adagrad", "ftrl"])
def test_multi_adagrad_and_ftrl_equal_single_ops(ops, which):
  D, sizes = 16, [700, 2048, 3, 0, 5000]
  rng = np.random.default_rng(7)

  def make():
    out = []
    for j in range(len(sizes)):
      hs = []
      for val in ((None, 0.1, 0.0) if which == "ftrl" else (None, 0.1)):
        h = ops.kv_variable([D])
        ops.kv_set_clock_days(h, DAY); ops.kv_set_seed(h, 11 + j)
        tab = np.random.default_rng(50 + j).standard_normal((32, D)).astype(np.float32) if val is None \
            else np.full((4, D), val, np.float32)
        ops.init_kv_variable_v2(h, tab)
        ops.kv_set_deterministic(h, True)
        hs.append(h)
      out.append(hs)
    return out
  A, B = make(), make()
  for step in range(3):
    ids = [rng.integers(-300, 300, n) for n in sizes]
    grads = [(rng.uniform(0.5, 1.5, (i.size, D)) * 1e-1).astype(np.float32) for i in ids]
    if which == "adagrad":
      ops.kv_multi_sparse_apply_adagrad([a[0] for a in A], [a[1] for a in A], 0.05, grads, ids)
      for b, g, i in zip(B, grads, ids):
        ops.kv_variable_sparse_apply_adagrad(b[0], b[1], 0.05, g, i, use_locking=True)
    else:
      ops.kv_multi_sparse_group_sparse_apply_ftrl([a[0] for a in A], [a[1] for a in A], [a[2] for a in A], grads, ids,
                                                  0.05, 1e-3, 1e-3, 1e-4, 0.0, -0.5)
      for b, g, i in zip(B, grads, ids):
        ops.kv_variable_sparse_group_sparse_apply_ftrl_v2(b[0], b[1], b[2], g, i, 0.05, 1e-3, 1e-3, 1e-4, 0.0, -0.5)
  for a, b in zip(A, B):
    for ha, hb in zip(a, b):
      ka, va, fa, sa = _dump(ops, ha)
      kb, vb, fb, sb = _dump(ops, hb)
      assert torch.equal(ka, kb) and fa == fb and sa == sb
      assert torch.equal(va, vb)                                     # same kernels, deterministic mode: bit-identical
