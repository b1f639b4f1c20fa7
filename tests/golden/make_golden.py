"""Generates tests/golden/*.npz — the known-answer vectors that pin the oracle.

The reference (tfplus on tensorflow-cpu 2.13) cannot be imported here, and its own
tests draw inputs from an unseeded np.random, so the fixtures restate WHAT those
tests assert: that one KvVariable optimizer step equals the closed-form TF-core
optimizer step (py_ut/tests/test_training_ops.py:418-473), and the constant /
counting results of py_ut/tests/test_kv_variable_ops.py:150-268.  The expected
values below are computed with plain numpy in float64 from the published TF-core
update rules (tf.compat.v1.train.AdamOptimizer / AdagradOptimizer sparse apply),
NOT with the oracle, so the oracle is checked against an independent statement.

Run:  python tests/golden/make_golden.py      (rewrites the .npz files in place)
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SEED = 20250211


def tf_adam_step(var, m, v, g, lr, b1, b2, eps, t):
  """tf.compat.v1.train.AdamOptimizer._apply_sparse_shared, step t (1-based), float64.

  beta1_power / beta2_power are float32 non-slot variables in TF-core (initial value beta,
  multiplied by beta in _finish), so their float32 rounding is part of the published rule:
  1 - beta2_power amplifies it to ~1e-5 relative in lr_t."""
  b1p, b2p = beta_powers(b1, b2, t)
  b1p, b2p = float(b1p), float(b2p)
  # hyper-parameters reach the kernels as float32 tensors (group_adam.py:203-212 casts);
  # 1 - float32(0.999) differs from 0.001 by 1.3e-5 relative, far above the test tolerance
  b1, b2, eps = float(np.float32(b1)), float(np.float32(b2)), float(np.float32(eps))
  lr_t = lr * np.sqrt(1 - b2p) / (1 - b1p)
  m = m * b1 + g * (1 - b1)
  v = v * b2 + g * g * (1 - b2)
  var = var - lr_t * m / (np.sqrt(v) + eps)
  return var, m, v


def beta_powers(b1, b2, t):
  p1, p2 = np.float32(b1), np.float32(b2)
  for _ in range(t - 1):
    p1, p2 = np.float32(p1 * np.float32(b1)), np.float32(p2 * np.float32(b2))
  return p1, p2


def tf_adagrad_step(var, acc, g, lr):
  """tf.compat.v1.train.AdagradOptimizer sparse apply, float64."""
  acc = acc + g * g
  var = var - lr * g / np.sqrt(acc)
  return var, acc


def tf_ftrl_v2_step(var, accum, linear, g, lr, l1, l2, l2_shrinkage, lr_power):
  """TF-core ResourceSparseApplyFtrlV2 (FtrlCompute with l2 shrinkage), float64."""
  g_s = g + 2.0 * l2_shrinkage * var
  new_accum = accum + g * g
  p = -lr_power
  linear = linear + g_s - (new_accum**p - accum**p) / lr * var
  quadratic = new_accum**p / lr + 2.0 * l2
  var = np.where(np.abs(linear) > l1, (np.sign(linear) * l1 - linear) / quadratic, 0.0)
  return var, new_accum, linear


def main():
  rng = np.random.Generator(np.random.PCG64(SEED))

  # A1: test_group_adam_v4_optimizer[_with_1embedding_dim] (test_training_ops.py:437-473)
  # var = ones (ones_initializer, :207-241), 10 ids 0..9, grad ~ U[0,1), lr .5, defaults
  # beta=(.9,.999) eps 1e-8, l1=l2=l21=0; asserted equal to TF Adam, allclose(atol=1e-8).
  for D in (64, 1):
    g = rng.random((10, D)).astype(np.float32)
    var, m, v = tf_adam_step(np.ones((10, D)), 0.0, 0.0, g.astype(np.float64), 0.5, 0.9, 0.999,
                             1e-8, 1)
    # second step with a fresh gradient: exercises the beta1 > beta1_power branch
    g2 = rng.random((10, D)).astype(np.float32)
    var2, m2, v2 = tf_adam_step(var, m, v, g2.astype(np.float64), 0.5, 0.9, 0.999, 1e-8, 2)
    np.savez(os.path.join(HERE, "A1_group_adam_v4_D%d.npz" % D), ids=np.arange(10, dtype=np.int64),
             grad=g, grad2=g2, expect_var=var.astype(np.float32), expect_m=m.astype(np.float32),
             expect_v=v.astype(np.float32), expect_var2=var2.astype(np.float32),
             expect_m2=m2.astype(np.float32), expect_v2=v2.astype(np.float32),
             beta_powers=np.array([beta_powers(0.9, 0.999, 1), beta_powers(0.9, 0.999, 2)],
                                  np.float32))

  # A2: test_adagrad_optimizer (test_training_ops.py:418-435): lr .5, accumulator init .1
  g = rng.random((10, 64)).astype(np.float32)
  var, acc = tf_adagrad_step(np.ones((10, 64)), np.full((10, 64), 0.1), g.astype(np.float64), 0.5)
  np.savez(os.path.join(HERE, "A2_adagrad.npz"), ids=np.arange(10, dtype=np.int64), grad=g,
           expect_var=var.astype(np.float32), expect_acc=acc.astype(np.float32))

  # G1: test_kv_variable_gather_v2 (test_kv_variable_ops.py:234-268): ones init table
  # [1024, 8]; GatherOrZeros on the empty table -> zeros[5,8]; GatherOrInsert -> ones[5,8].
  # G2: test_kv_variable_frequency (:150-189): enter_threshold=2, gather {0..4} -> sum_freq 0,
  # gather {2..6} -> 6, GatherOrZeros{0..4} leaves 6.
  np.savez(os.path.join(HERE, "G1G2_gather_frequency.npz"),
           ids0=np.arange(0, 5, dtype=np.int64), ids1=np.arange(2, 7, dtype=np.int64),
           expect_zeros=np.zeros((5, 8), np.float32), expect_ones=np.ones((5, 8), np.float32),
           expect_sum_freq=np.array([0, 6, 6], np.int64))

  # F1: KvVariableTest.TestKvStat (kernels/kv_variable_test.cc:359-382): freq word packing
  # lo16 = frequency (saturating at 65535), hi16 = day; MakeUint32FromUint16(hi, lo).
  # The test's vector is (day 65534, frequency 65535); 18303 is the day its comment names.
  np.savez(os.path.join(HERE, "F1_freq_word.npz"),
           hi=np.array([65534, 18303], np.uint32), lo=np.array([65535, 7], np.uint32),
           word=np.array([(65534 << 16) | 65535, (18303 << 16) | 7], np.uint32))

  # A4: test_kv_variable_sparse_apply_ftrl (test_training_ops.py:68-205): one FTRL-V2 step on 300
  # ids x dim 64, var = .03, accum = .1, linear = 0, lr = .01, l1 = l2 = l2_shrinkage = 0,
  # lr_power = -.5, grad ~ N(0, 1); asserted equal to TF's ResourceSparseApplyFtrlV2, atol 1e-8.
  # (The op under test there is the plain FTRL; KvVariableSparseGroupSparseApplyFtrlV2 with
  # l21 = 0 is the same update — training_ops.cc:532-801 — which is what this vector pins.)
  rng4 = np.random.Generator(np.random.PCG64(SEED + 4))      # own stream: earlier fixtures stay byte-identical
  g = rng4.standard_normal((300, 64)).astype(np.float32)
  var, acc, lin = tf_ftrl_v2_step(np.full((300, 64), np.float64(np.float32(0.03))), np.full((300, 64), np.float64(np.float32(0.1))),
                                  np.zeros((300, 64)), g.astype(np.float64), float(np.float32(0.01)), 0.0, 0.0, 0.0, -0.5)
  np.savez(os.path.join(HERE, "A4_ftrl_v2.npz"), ids=np.arange(300, dtype=np.int64), grad=g,
           expect_var=var.astype(np.float32), expect_accum=acc.astype(np.float32), expect_linear=lin.astype(np.float32))

  # I1: test_kv_variable_import_v2 (py_ut/tests/test_kv_variable_ops.py:345-435): enter_threshold = 1, keys
  # 0..4 with values k * ones[64], blacklist [7], frequency table keys 1..5 -> values 1..5 (uint16), imported
  # with first_n 3, 4, 6 in turn (each import clears the table) and exported with first_n = 6 after each.
  # The reference asserts the SHAPES of the six outputs: keys [5], values [5, D] every time; blacklist
  # 0 / 1 / 1 and frequency lists 5 / 6 / 6 entries (first_n 3 drops the blacklist and the frequency table,
  # kernels/kv_variable_ops.cc:806-822; the blacklisted key 7 is a key of the map, so it is in the frequency
  # export; frequency key 5 names no key and is dropped, dynamic_restore.hpp:232-246).
  D = 64
  np.savez(os.path.join(HERE, "I1_import_v2.npz"), keys=np.arange(5, dtype=np.int64),
           values=(np.arange(5, dtype=np.float32)[:, None] * np.ones((1, D), np.float32)),
           blacklist=np.array([7], np.int64), freq_keys=np.arange(1, 6, dtype=np.int64),
           freq_values=np.arange(1, 6, dtype=np.uint32), first_n=np.array([3, 4, 6], np.int64),
           expect_rows=np.array([5, 5, 5], np.int64), expect_blacklist=np.array([0, 1, 1], np.int64),
           expect_freq=np.array([5, 6, 6], np.int64), enter_threshold=np.array([1], np.int64))


if __name__ == "__main__":
  main()
