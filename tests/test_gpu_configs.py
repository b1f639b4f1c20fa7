"""BASELINE.json configs as parity cases.  configs[0] — the reference's own CPU-runnable case —
at its full size: one KvVariable, 100 k keys x dim 32, batches of 10 k int64 ids, GroupAdam steps, GPU against
the oracle: lookups bit-exact, frequency words / sizes exact, fp32 rows and optimizer state within 1e-6 relative
(north_star's tolerance).  configs[1] at full size is tests/test_gpu_parity.py::test_full_size_batch_properties;
configs[2]'s 26-table shape is tests/test_gpu_multi_table.py."""
import numpy as np
import pytest

from oracle import kv_oracle as ko

torch = pytest.importorskip("torch")
SEED = 20250211 + 1      # §8(d): seed = 20250211 + config#
DAY = 20000


@pytest.mark.gpu
def test_config0_100k_keys_10k_ids_group_adam_steps():
  if not torch.cuda.is_available():
    pytest.skip("needs a GPU")
  from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as ops
  rng = np.random.Generator(np.random.PCG64(SEED))
  K, N, D, STEPS = 100_000, 10_000, 32, 6
  table = rng.normal(-1.0, 1.0, (10000, D)).astype(np.float32)          # get_kv_variable's [10000, D] init table
  hv = ops.kv_variable([D], capacity_hint=K + N)
  hs = ops.kv_variable([3 * D], capacity_hint=K + N)
  ov = ko.OracleKv(D, 0, table, day=DAY, picker=1, seed=SEED)    # one thread: the reference's frequency bump of a repeated id is a racy read-modify-write under a shared lock
  os_ = ko.OracleKv(3 * D, 0, np.zeros((16, 3 * D), np.float32), day=DAY, picker=1, seed=SEED)
  for h, t in ((hv, table), (hs, np.zeros((16, 3 * D), np.float32))):
    ops.kv_set_clock_days(h, DAY); ops.kv_set_seed(h, SEED); ops.init_kv_variable_v2(h, t)
  keys = np.arange(K, dtype=np.int64)
  np.testing.assert_array_equal(ops.kv_variable_gather_or_insert_v2(hv, keys).cpu().numpy(), ov.gather_or_insert(keys))
  lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-8
  b1p, b2p = np.float32(b1), np.float32(b2)                              # beta powers start at beta (TF-core Adam)
  for step in range(STEPS):
    ids = rng.integers(0, K, N).astype(np.int64)
    grad = rng.normal(0.0, 1e-2, (N, D)).astype(np.float32)
    got = ops.kv_variable_gather_or_insert_v2(hv, ids).cpu().numpy()
    want = ov.gather_or_insert(ids)
    if step == 0:
      np.testing.assert_array_equal(got, want)                           # untouched rows: bit-exact
    else:
      np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-9)
    ops.kv_variable_group_sparse_apply_adam_v4(hv, hs, grad, ids, lr, b1p, b2p, b1, b2, eps, 0.0, 0.0, 0.0)
    u, s, _ = ko.dedup_segment_sum(ids, grad)                            # TF-core unique + unsorted_segment_sum
    ko.apply_group_adam(ov, os_, s, u, lr, b1p, b2p, b1, b2, eps)
    b1p, b2p = np.float32(b1p * np.float32(b1)), np.float32(b2p * np.float32(b2))   # multiplied after the apply (_finish)
  assert ops.kv_variable_size_v2(hv) == ov.size() == K
  assert ops.kv_variable_frequency(hv) == ov.sum_freq()
  assert ops.kv_variable_shape_v2(hs)[0] == os_.map_size()
  np.testing.assert_allclose(ops.kv_variable_gather_or_zeros_v2(hv, keys).cpu().numpy(), ov.gather_or_zeros(keys),
                             rtol=1e-6, atol=1e-9)
  # m | v | z of every key that was ever updated (atol: z is a difference of O(1) terms)
  np.testing.assert_allclose(ops.kv_variable_gather_or_zeros_v2(hs, keys).cpu().numpy(), os_.gather_or_zeros(keys),
                             rtol=1e-6, atol=1e-7)
  q = rng.integers(0, K, 2000)
  np.testing.assert_array_equal(ops.kv_variable_get_count_v2(hv, q).cpu().numpy(), ov.get_count(q))
