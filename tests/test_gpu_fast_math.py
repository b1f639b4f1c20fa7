"""kv_set_fast_math (include/kvhip.h:212; bound by gen_kv_variable_ops.kv_set_fast_math): the optimizers' row math
(kernels/training_ops.cc:7166-7195, :5895-5925, :1470-1482, :713-751) on the 1-ulp hardware sqrt / reciprocal
instructions instead of the IEEE sequences.  VERDICT r5 "What's weak" 3: exported, bound and untested.

What is pinned here:
  * on = 0 is the library default: a table that never called it and a table that called it with 0 hold the same bits;
  * on = 1 stays within a STATED tolerance of the oracle for all four optimizers — rtol 2e-6 / atol 1e-7 on the state:
    each sqrt / rcp / rsq is within 1 ulp (2^-23 relative) and an update chains at most four of them, so 4 * 1.2e-7 =
    4.8e-7 relative per operation chain plus the IEEE path's own 1e-6 bar.  The absolute term is the one the IEEE-path
    parity tests use (1e-7, rows of scale O(1)): an element whose update CANCELS (|x| orders below its row's scale) is
    bounded relative to the terms that cancelled, not to itself — round 6's first run of this test met one such element
    in 209 408 (|x| = 1.2e-5 in a row of scale 1, off by 2.0e-8 = 1.6e-3 of itself), the case kvhip.h documents;
    SparseGroupFtrl gets its own absolute term (4e-6 at the tests' lr = 0.05): its (sqrt(n) - sqrt(a)) / lr * x cancels
    by construction and multiplies the roots' ulps by |x| / lr — derived at FTRL_FAST_ATOL below;
  * on = 1 really changes the arithmetic (some element differs from the IEEE twin), so the test would notice a switch
    that does nothing;
  * a table in deterministic mode ignores it: same bits as the IEEE twin.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu

from test_gpu_parity import _np, _beta_pows, _assert_same_table  # noqa: E402
from test_gpu_unique_apply import _run, _oracle, _tables, _same_bits  # noqa: E402

FAST_RTOL, FAST_ATOL = 2e-6, 1e-7
# SparseGroupFtrl's linear term subtracts two square roots and divides by lr — z += g' - (sqrt(n) - sqrt(a)) / lr * x
# (training_ops.cc:726-733) — so a 1-ulp error of each root (2^-24 * sqrt(a) ~ 2e-8 at a ~ 0.1) reaches z multiplied by
# |x| / lr: with the tests' lr = 0.05 and |x| <= 4 that is 2 * 2e-8 * 4 / 0.05 = 3.2e-6 absolute, and x follows z.  The
# first run measured 2.3e-7 at most (1.3 % of the elements beyond 1e-7); the bound, not the luck, is what is stated.
FTRL_FAST_ATOL = 4e-6


def _atol(name):
  return FTRL_FAST_ATOL if name == "ftrl" else FAST_ATOL


@pytest.fixture(scope="module")
def ops():
  if not torch.cuda.is_available():
    pytest.skip("needs a GPU")
  from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as g
  return g


def _steps(ops, name, D, tabs, oracle=None, unique=False, nsteps=3, seed=7):
  """the same three steps (t = 1 and t >= 2 branches; repeated ids only on the plain path) on every table set of `tabs`"""
  rng = np.random.default_rng(seed + D)
  seen = []
  for t in range(nsteps):
    ids = rng.choice(5000, 1500, replace=False).astype(np.int64) - 300
    seen.append(ids)
    grad = rng.normal(0, 1e-2, (ids.size, D)).astype(np.float32)
    b1p, b2p = _beta_pows(t)
    kw = dict(lr=0.05, b1p=b1p, b2p=b2p)
    for hs in tabs:
      _run(ops, name, hs, grad, ids, unique, **kw)
    if oracle is not None:
      _oracle(name, oracle, grad, ids, **kw)
  return np.concatenate(seen)


@pytest.mark.parametrize("D", [8, 32, 64])
@pytest.mark.parametrize("name", ["adam4", "adam3", "adagrad", "ftrl"])
def test_fast_math_within_stated_tolerance_of_oracle(ops, name, D):
  hf, hi, os_ = _tables(ops, name, D)          # hf: fast math, hi: the IEEE default, os_: oracle
  ops.kv_set_fast_math(hf[0], True)
  keys = _steps(ops, name, D, [hf, hi], oracle=os_)
  for h, o in zip(hf, os_):
    _assert_same_table(ops, h, o, keys, rtol=FAST_RTOL, atol=_atol(name))
  # the switch does something: at least one state element differs from the IEEE twin
  differs = False
  uk = np.unique(keys)
  for a, b in zip(hf, hi):
    differs |= not np.array_equal(_np(ops.kv_variable_gather_or_zeros_v2(a, uk)), _np(ops.kv_variable_gather_or_zeros_v2(b, uk)))
  assert differs, "kv_set_fast_math(1) left every bit of the state as the IEEE path computes it"


@pytest.mark.parametrize("name", ["adam4", "adagrad", "ftrl"])
def test_fast_math_unique_path_within_tolerance(ops, name):
  """the one-launch unique-ids apply honours the switch too"""
  D = 32
  hf, hi, os_ = _tables(ops, name, D)
  ops.kv_set_fast_math(hf[0], True)
  keys = _steps(ops, name, D, [hf, hi], oracle=os_, unique=True)
  for h, o in zip(hf, os_):
    _assert_same_table(ops, h, o, keys, rtol=FAST_RTOL, atol=_atol(name))


@pytest.mark.parametrize("name", ["adam4", "adam3", "adagrad", "ftrl"])
def test_fast_math_off_is_the_default_bit_for_bit(ops, name):
  D = 32
  h0, hd, os_ = _tables(ops, name, D)          # h0: kv_set_fast_math(0) called, hd: never called
  ops.kv_set_fast_math(h0[0], False)
  keys = _steps(ops, name, D, [h0, hd], oracle=os_)
  for a, b in zip(h0, hd):
    _same_bits(ops, a, b, keys)
  for h, o in zip(hd, os_):
    _assert_same_table(ops, h, o, keys, rtol=1e-6, atol=1e-7)
  # switched on and off again: back to the IEEE sequences
  ops.kv_set_fast_math(h0[0], True)
  ops.kv_set_fast_math(h0[0], False)
  more = _steps(ops, name, D, [h0, hd], seed=99)
  for a, b in zip(h0, hd):
    _same_bits(ops, a, b, np.concatenate([keys, more]))


@pytest.mark.parametrize("name", ["adam4", "ftrl"])
def test_fast_math_is_ignored_in_deterministic_mode(ops, name):
  """kvhip.h: a table in deterministic mode always uses the IEEE sequences — whichever of the two switches came first"""
  D = 32
  hf, hi, _ = _tables(ops, name, D)
  ops.kv_set_fast_math(hf[0], True)            # fast math first, then deterministic ...
  for h in hf + hi:
    ops.kv_set_deterministic(h, True)
  keys = _steps(ops, name, D, [hf, hi])
  for a, b in zip(hf, hi):
    _same_bits(ops, a, b, keys)
  ops.kv_set_fast_math(hf[0], True)            # ... and asked for again while deterministic
  more = _steps(ops, name, D, [hf, hi], seed=55)
  for a, b in zip(hf, hi):
    _same_bits(ops, a, b, np.concatenate([keys, more]))
