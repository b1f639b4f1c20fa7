"""tf.unique_with_counts / tf.unique + unsorted_segment_sum on the entry-list kernels (kv_unique, kv_dedup_segment_sum:
TF-core's _deduplicate_indexed_slices, variable_scope.py:1096-1106, and embedding_ops.py:362-372): dense numbers, the
inverse, saturating counts, every key type and dim, batches longer than the sorted-position kernels' 2^21."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
  if not torch.cuda.is_available():
    pytest.skip("needs a GPU")
  from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as g
  return g


def _table(ops, D, key_dtype=torch.int64):
  h = ops.kv_variable([D], key_dtype=key_dtype)
  ops.init_kv_variable_v2(h, np.zeros((4, D), np.float32))
  return h


def _check_unique(ids, counts, u, c, inv):
  ids_np = ids.cpu().numpy().astype(np.int64)
  u, c, inv = u.cpu().numpy(), c.cpu().numpy(), inv.cpu().numpy()
  ref_u, ref_inv = np.unique(ids_np, return_inverse=True)
  assert np.array_equal(np.sort(u), ref_u)                   # every distinct id exactly once
  assert inv.min() >= 0 and inv.max() < u.size
  assert np.array_equal(u[inv], ids_np)                      # the inverse names every position's id
  w = np.ones(ids_np.size, np.int64) if counts is None else counts.cpu().numpy().astype(np.int64).clip(max=65535)
  want = np.zeros(u.size, np.int64)
  np.add.at(want, inv, w)
  assert np.array_equal(c, np.minimum(want, 65535))          # summed counts, saturating like the frequency they feed


@pytest.mark.parametrize("D", [32, 5, 1])                    # the table's dim does not matter: no row is touched
@pytest.mark.parametrize("n", [1, 77, 2048, 2049, 50_000])
def test_unique_with_counts(ops, D, n):
  dev = torch.device("cuda", 0)
  gen = torch.Generator(device=dev).manual_seed(n * 7 + D)
  h = _table(ops, D)
  ids = torch.randint(-max(2, n // 3), max(2, n // 3), (n,), device=dev, generator=gen)
  if n > 3:
    ids[n // 2] = torch.iinfo(torch.int64).min               # the index's EMPTY sentinel is a legal id
    ids[0] = torch.iinfo(torch.int64).min
  _check_unique(ids, None, *ops.kv_unique(h, ids))
  cnt = torch.randint(0, 5, (n,), device=dev, generator=gen, dtype=torch.int32)
  cnt[0] = 70000                                             # one occurrence's count alone saturates
  _check_unique(ids, cnt, *ops.kv_unique(h, ids, cnt))
  u, c, inv, nu = ops.kv_unique(h, ids, sync=False)          # the count stays on the device
  k = int(nu.item())
  _check_unique(ids, None, u[:k], c[:k], inv)


def test_unique_int32_keys_and_heavy_hitter(ops):
  dev = torch.device("cuda", 0)
  gen = torch.Generator(device=dev).manual_seed(5)
  h = _table(ops, 8, key_dtype=torch.int32)
  ids = torch.randint(-1000, 1000, (300_000,), device=dev, generator=gen, dtype=torch.int32)
  ids[::3] = 7                                               # 100 000 occurrences of one id: its count saturates
  u, c, inv = ops.kv_unique(h, ids)
  _check_unique(ids, None, u, c, inv)
  assert int(c[u == 7].item()) == 65535


def test_unique_and_dedup_sum_over_the_old_limit(ops):
  """3 M ids in one call; the sums against float64."""
  dev = torch.device("cuda", 0)
  gen = torch.Generator(device=dev).manual_seed(9)
  D, n = 16, 3_000_000
  h = _table(ops, D)
  ids = (torch.rand(n, device=dev, generator=gen) ** 3 * 400_000).to(torch.int64) - 1000   # skewed: hot and cold ids
  grad = torch.randn(n, D, device=dev, generator=gen) * 1e-2
  u, c, inv = ops.kv_unique(h, ids)
  ref_u, ref_inv, ref_c = torch.unique(ids, return_inverse=True, return_counts=True)
  assert torch.equal(torch.sort(u).values, ref_u) and torch.equal(u[inv.long()], ids)
  assert torch.equal(c.long(), torch.clamp(ref_c, max=65535)[torch.searchsorted(ref_u, u)])
  u2, s, inv2 = ops.kv_dedup_segment_sum(h, ids, grad)
  assert torch.equal(torch.sort(u2).values, ref_u) and torch.equal(u2[inv2.long()], ids)
  exact = torch.zeros((u2.numel(), D), dtype=torch.float64, device=dev).index_add_(0, inv2.long(), grad.double())
  gabs = torch.zeros_like(exact).index_add_(0, inv2.long(), grad.double().abs())
  cnt = torch.bincount(inv2.long(), minlength=u2.numel()).double().unsqueeze(1)
  assert bool(((s.double() - exact).abs() <= cnt * 6e-8 * 1.01 * gabs + 1e-30).all())   # gamma_n * sum|g| for ANY order
  single = cnt[:, 0] == 1
  assert torch.equal(s[single], exact[single].float())       # singletons are copies
  from tfplus_amd import _lib
  with pytest.raises(_lib.KvError):
    ops.kv_unique(h, torch.zeros((1 << 23) + 1, dtype=torch.int64, device=dev))


@pytest.mark.parametrize("D", [4, 12, 64, 100, 256, 6])      # 6: a dim the entry-list kernels do not serve (the old pipeline)
def test_dedup_sum_dims_and_deterministic_mode(ops, D):
  dev = torch.device("cuda", 0)
  gen = torch.Generator(device=dev).manual_seed(D)
  h = _table(ops, D)
  n = 40_000
  ids = torch.randint(0, 3000, (n,), device=dev, generator=gen)
  grad = torch.randn(n, D, device=dev, generator=gen)
  u, s, inv = ops.kv_dedup_segment_sum(h, ids, grad)
  assert torch.equal(u[inv.long()], ids) and u.numel() == torch.unique(ids).numel()
  exact = torch.zeros((u.numel(), D), dtype=torch.float64, device=dev).index_add_(0, inv.long(), grad.double())
  torch.testing.assert_close(s.double(), exact, rtol=1e-5, atol=1e-5)
  ops.kv_set_deterministic(h, True)
  runs = []
  for _ in range(2):
    u, s, inv = ops.kv_dedup_segment_sum(h, ids, grad)
    order = torch.argsort(u)
    runs.append((u[order], s[order]))
  assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])   # bit-reproducible sums
