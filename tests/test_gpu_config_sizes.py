"""BASELINE.json configs at THEIR sizes, on one MI355X, against the oracle:

  configs[1]  50 M-key KvVariable x dim 32, 1 M ids / batch Zipf(1.2), lookup + fused GroupAdam apply (batch token
              path, the one bench.py times): size-independent properties on the whole batch, and oracle parity on
              every key the batch touches (the oracle table holds exactly those keys: same init rule, same seed);
  configs[3]  one rank's share of the 8-GPU table: 125 M keys x dim 64 (+ m_v_linear 3 x 64), ~141 GB of HBM,
              1 M ids: the same properties and the oracle on a sample of the touched keys;
  configs[4]  256 KvVariables, dims {8, 16, 32, 64, 128}, GroupAdam + SparseGroupFtrl mixed, batched kv_multi_* ops
              per (optimizer, dim) group; one rank's share (32 tables, Criteo-like log-uniform cardinalities up to
              4e7) and all 256 small tables; the first table of every group against the oracle.

Tolerances (north_star): rows returned by lookups, row ids' effects, frequency words, sizes: bit-exact; fp32
optimizer state of ids that occur ONCE in the batch: 1e-6 relative; ids that repeat: the fp32 sum of their
gradients may be taken in another order than TF-core's, so those are bounded per element by the worst-case
rounding of an fp32 sum in any order, pushed through the update (the bound test_full_size_batch_properties
derives) — with two-signed gradients, i.e. including sums that cancel."""
import os
import sys

import numpy as np
import pytest

from oracle import kv_oracle as ko

torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)
DAY = 20000
SEED = 20250211


@pytest.fixture(scope="module")
def ops():
  if not torch.cuda.is_available():
    pytest.skip("needs a GPU")
  from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as g
  return g


def _need_hbm(gib):
  free, _ = torch.cuda.mem_get_info()
  if free < gib * (1 << 30):
    pytest.skip("needs %d GiB of free HBM, has %.0f" % (gib, free / (1 << 30)))


def _build(ops, K, D, table, seed, slot_mult=3):
  """var + slot tables pre-filled with keys splitmix64(1..K) as bench.py does (steady state)."""
  import bench
  dev = torch.device("cuda", 0)
  var = ops.kv_variable([D], capacity_hint=K + (4 << 20))
  slot = ops.kv_variable([slot_mult * D], capacity_hint=K + (4 << 20))
  for h, t in ((var, table), (slot, np.zeros((16, slot_mult * D), np.float32))):
    ops.kv_set_clock_days(h, DAY); ops.kv_set_seed(h, seed); ops.init_kv_variable_v2(h, t)
  CH = 1 << 21
  for i in range(0, K, CH):
    keys = bench.splitmix64(torch.arange(i + 1, min(i + CH, K) + 1, dtype=torch.int64, device=dev))
    ops.kv_variable_gather_or_insert_v2(var, keys)
    ops.kv_variable_gather_or_insert_v2(slot, keys)
  assert ops.kv_variable_shape_v2(var)[0] == K
  ops.kv_attach_slot(var, slot)
  return var, slot


def _adam_bound(gsum, gabs, cnt, alpha, omb1, omb2, eps32):
  """|x_gpu - x_exact| allowed when the fp32 gradient sum is taken in ANY order (see module docstring)."""
  dg = (cnt - 1).clamp(min=0) * 2.0 ** -24 * gabs
  gmin = (gsum.abs() - dg).clamp(min=0)
  slope = alpha * omb1 * eps32 / (np.sqrt(omb2) * gmin + eps32) ** 2
  return 1e-6 + slope * dg


def _lookup_apply_check(ops, var, slot, K, D, table, seed, sample_max, zipf=1.2, occurrence=False):
  """one training step of the configs[1] / configs[3] shape + every check; returns (unique ids, sampled)."""
  import bench
  dev = torch.device("cuda", 0)
  N = 1_000_000
  gen = torch.Generator(device=dev).manual_seed(SEED + 2)
  ids = bench.splitmix64(bench.Zipf(K, zipf, dev).sample(N, gen))
  grad = torch.randn(N, D, device=dev, generator=gen) * 1e-2            # two-signed: sums of repeated ids cancel
  # ---- lookup: size-independent properties -------------------------------------------------------------
  f0 = ops.kv_variable_frequency(var)
  out = ops.kv_variable_gather_or_insert_v2(var, ids)
  assert var.batch is not None                                           # the lookup named the batch
  uniq, inv = torch.unique(ids, return_inverse=True)
  cnt = torch.bincount(inv)
  first = torch.full((uniq.numel(),), N, device=dev, dtype=torch.int64).scatter_reduce(0, inv, torch.arange(N, device=dev), "amin")
  assert torch.equal(out, out[first][inv])                               # every occurrence of a key reads the same row
  assert torch.equal(ops.kv_variable_gather_or_zeros_v2(var, ids), out)  # training lookup == inference lookup
  assert ops.kv_variable_shape_v2(var)[0] == K                           # steady state: no key was new
  assert ops.kv_variable_frequency(var) - f0 == int(torch.clamp(cnt + 1, max=65535).sum() - cnt.numel())
  # ---- the oracle holds the touched keys (sampled when there are more than sample_max) ---------------------------
  pick = torch.arange(uniq.numel(), device=dev)
  if uniq.numel() > sample_max:
    pick = torch.randperm(uniq.numel(), device=dev, generator=gen)[:sample_max].sort().values
  insample = torch.zeros(uniq.numel(), dtype=torch.bool, device=dev)
  insample[pick] = True
  pos = torch.nonzero(insample[inv]).squeeze(1)                          # batch positions of sampled keys, in order
  skeys = uniq[pick].cpu().numpy()
  ov = ko.OracleKv(D, 0, table, day=DAY, picker=1, seed=seed)
  os_ = ko.OracleKv(slot.dim, 0, np.zeros((16, slot.dim), np.float32), day=DAY, picker=1, seed=seed)
  ov.gather_or_insert(skeys); os_.gather_or_insert(skeys)                # the pre-fill
  ids_s = ids[pos].cpu().numpy()
  np.testing.assert_array_equal(out[pos].cpu().numpy(), ov.gather_or_insert(ids_s))   # bit-exact
  got_cnt = ops.kv_variable_get_count_v2(var, skeys[:5000]).cpu().numpy()
  np.testing.assert_array_equal(got_cnt, ov.get_count(skeys[:5000]))    # frequency words: exact
  # ---- fused GroupAdam apply, batch token path -----------------------------------------------------------------
  tok_before = var.batch[0]
  ops.kv_variable_group_sparse_apply_adam_v4(var, slot, grad, ids, 1e-3, 0.9, 0.999, 0.9, 0.999, 1e-8, 0, 0, 0)
  assert var.batch[0] == tok_before
  u, sm, _ = ko.dedup_segment_sum(ids_s, grad[pos].cpu().numpy())        # TF-core unique + unsorted_segment_sum
  ko.apply_group_adam(ov, os_, sm, u, 1e-3, 0.9, 0.999, 0.9, 0.999, 1e-8)
  got = ops.kv_variable_gather_or_zeros_v2(var, uniq)
  gots = ops.kv_variable_gather_or_zeros_v2(slot, uniq)
  if occurrence:
    # kv_set_deterministic(var, 2): the repeats were added in TF-core's order — EVERY sampled key at the unique-id bar
    np.testing.assert_allclose(got[pick].cpu().numpy(), ov.gather_or_zeros(skeys), rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(gots[pick].cpu().numpy(), os_.gather_or_zeros(skeys), rtol=1e-6, atol=1e-9)
  # (1) ids that occur once: 1e-6 against the oracle, var and m | v | z
  once = (cnt[pick] == 1).cpu().numpy()
  assert once.sum() > 1000
  np.testing.assert_allclose(got[pick].cpu().numpy()[once], ov.gather_or_zeros(skeys)[once], rtol=1e-6, atol=1e-9)
  np.testing.assert_allclose(gots[pick].cpu().numpy()[once], os_.gather_or_zeros(skeys)[once], rtol=1e-6, atol=1e-9)
  # (2) every touched key: bounded against the exact (fp64) sum, per element
  gsum = torch.zeros((uniq.numel(), D), dtype=torch.float64, device=dev).index_add_(0, inv, grad.double())
  gabs = torch.zeros_like(gsum).index_add_(0, inv, grad.double().abs())
  omb1 = float(np.float32(1) - np.float32(0.9)); omb2 = float(np.float32(1) - np.float32(0.999))
  eps32 = float(np.float32(1e-8))
  alpha = float(np.float32(1e-3)) * np.sqrt(omb2) / omb1
  x0 = out[first].double()
  expect = x0 - alpha * (omb1 * gsum) / ((omb2 * gsum * gsum).sqrt() + eps32)
  bound = _adam_bound(gsum, gabs, cnt.double().unsqueeze(1), alpha, omb1, omb2, eps32)
  excess = ((got.double() - expect).abs() - bound).max().item()
  assert excess < 0, excess
  assert (bound > 2e-6).double().mean().item() < 1e-3                    # the amplified (cancelling) cases are rare
  # (3) slot rows exist for exactly the touched keys' set growth; slot frequency bumped once per key
  np.testing.assert_array_equal(ops.kv_variable_get_count_v2(slot, skeys[:5000]).cpu().numpy(), os_.get_count(skeys[:5000]))
  # a second identical lookup still returns one row per key (and the token of the first batch is stale now)
  out2 = ops.kv_variable_gather_or_insert_v2(var, ids)
  assert torch.equal(out2, out2[first][inv]) and torch.equal(out2[first], got)
  return uniq.numel(), pick.numel()


@pytest.mark.gpu
def test_config1_50M_keys_1M_zipf_ids(ops):
  _need_hbm(60)
  rng = np.random.Generator(np.random.PCG64(SEED + 2))
  D, K = 32, 50_000_000
  table = (rng.standard_normal((10000, D)) * 0.05).astype(np.float32)
  var, slot = _build(ops, K, D, table, seed=11)
  u, s = _lookup_apply_check(ops, var, slot, K, D, table, 11, sample_max=10 ** 9)   # every touched key
  assert 90_000 < u < 130_000 and s == u


@pytest.mark.gpu
def test_config1_occurrence_order_every_key(ops):
  """configs[1] at full size with the var in occurrence-order mode: every touched key — the one with 188 k rows of the batch
  included — is the oracle's to 1e-6 (the default mode above holds the repeated keys to the reorder bound instead)"""
  _need_hbm(60)
  rng = np.random.Generator(np.random.PCG64(SEED + 2))
  D, K = 32, 50_000_000
  table = (rng.standard_normal((10000, D)) * 0.05).astype(np.float32)
  var, slot = _build(ops, K, D, table, seed=11)
  ops.kv_set_deterministic(var, ops.KV_ORDER_OCCURRENCE)
  u, s = _lookup_apply_check(ops, var, slot, K, D, table, 11, sample_max=10 ** 9, occurrence=True)
  assert 90_000 < u < 130_000 and s == u


@pytest.mark.gpu
def test_low_skew_1M_ids_zipf03(ops):
  """the same step where nearly every id of the batch is distinct (Zipf 0.3: ~0.99 M keys per 1 M ids): the tile and
  partition passes see no repeats to fold, the partition pass runs with twice the partitions after the first batch"""
  _need_hbm(40)
  rng = np.random.Generator(np.random.PCG64(SEED + 3))
  D, K = 32, 20_000_000
  table = (rng.standard_normal((10000, D)) * 0.05).astype(np.float32)
  var, slot = _build(ops, K, D, table, seed=12)
  u, s = _lookup_apply_check(ops, var, slot, K, D, table, 12, sample_max=100_000, zipf=0.3)
  assert u > 950_000 and s == 100_000


@pytest.mark.gpu
def test_config3_one_rank_share_125M_keys_dim64(ops):
  _need_hbm(190)
  rng = np.random.Generator(np.random.PCG64(SEED + 4))
  D, K = 64, 125_000_000
  table = (rng.standard_normal((10000, D)) * 0.05).astype(np.float32)
  var, slot = _build(ops, K, D, table, seed=13)
  u, s = _lookup_apply_check(ops, var, slot, K, D, table, 13, sample_max=100_000)
  assert s == 100_000 and u > s


# ---------------------------------------------------------------------------------------------
# configs[4]
# ---------------------------------------------------------------------------------------------
DIMS = (8, 16, 32, 64, 128)


class _Table(object):
  def __init__(self, ops, D, opt, card, seed):
    rng = np.random.default_rng(seed)
    self.D, self.opt, self.card = D, opt, card
    self.table = rng.standard_normal((64, D)).astype(np.float32)
    self.h = ops.kv_variable([D])
    specs = [(3 * D, 0.0)] if opt == "adam" else [(D, 0.1), (D, 0.0)]
    self.slots = [ops.kv_variable([d]) for d, _ in specs]
    self.slot_tabs = [np.full((4, d), v, np.float32) for d, v in specs]
    for h, t in [(self.h, self.table)] + list(zip(self.slots, self.slot_tabs)):
      ops.kv_set_clock_days(h, DAY); ops.kv_set_seed(h, seed); ops.init_kv_variable_v2(h, t)
    self.seed = seed

  def oracle(self):
    ov = ko.OracleKv(self.D, 0, self.table, day=DAY, picker=1, seed=self.seed)
    sl = [ko.OracleKv(t.shape[1], 0, t, day=DAY, picker=1, seed=self.seed) for t in self.slot_tabs]
    return ov, sl


def _step_fn(opt, x, st, g, hp):
  """the reference's per-element update of a row with (summed) gradient g, float64 (training_ops.cc:7166-7195
  with l1 = l2 = l21 = 0 is Adam; :713-751 SparseGroupFtrl with l1 = l21 = l2_shrinkage = 0, lr_power = -0.5)."""
  if opt == "adam":
    m, v = st
    m2 = hp["b1"] * m + (1 - hp["b1"]) * g
    v2 = hp["b2"] * v + (1 - hp["b2"]) * g * g
    return x - hp["alpha"] * m2 / (np.sqrt(v2) + hp["eps"])
  a, z = st
  na = a + g * g
  z2 = z + g - (np.sqrt(na) - np.sqrt(a)) / hp["lr"] * x
  return -z2 / (np.sqrt(na) / hp["lr"] + 2 * hp["l2"])


def _reorder_bound(opt, x, st, gsum, dg, hp):
  """largest change of the updated row when the summed gradient moves anywhere inside [gsum - dg, gsum + dg]
  (the fp32 sum of a repeated id's gradients taken in any order): the update is piecewise monotone in g with
  its only kink at g = 0, so the end points (and 0 when inside) bracket it."""
  f0 = _step_fn(opt, x, st, gsum, hp)
  dev = np.maximum(np.abs(_step_fn(opt, x, st, gsum - dg, hp) - f0), np.abs(_step_fn(opt, x, st, gsum + dg, hp) - f0))
  inside = np.abs(gsum) <= dg
  dev = np.where(inside, np.maximum(dev, np.abs(_step_fn(opt, x, st, np.zeros_like(gsum), hp) - f0)), dev)
  # two fp32 evaluations at neighbouring g also round differently: a few ulps of the largest intermediate term,
  # carried to the row (Adam updates x in place; FTRL rebuilds it from the linear slot: x = -z / q)
  if opt == "adam":
    noise = 2.0 ** -19 * np.abs(x)
  else:
    a, z = st
    na = a + gsum * gsum
    c = (np.sqrt(na) - np.sqrt(a)) / hp["lr"]
    # ... and (sqrt(na) - sqrt(a)) / lr cancels: one ulp of sqrt(na) is a whole 2^-24 sqrt(na) / lr in c, times x
    noise = (2.0 ** -19 * (np.abs(z) + np.abs(gsum) + np.abs(c * x)) + 2.0 ** -22 * (np.sqrt(na) + np.sqrt(a)) / hp["lr"] * np.abs(x)) \
        / (np.sqrt(na) / hp["lr"] + 2 * hp["l2"])
  return 2.0 * dev + noise


def _run_config4(ops, ntables, ids_per_table, max_card, steps=2):
  rng = np.random.default_rng(SEED + 5)
  tabs = []
  for j in range(ntables):
    D = DIMS[j % len(DIMS)]
    opt = "adam" if (j // len(DIMS)) % 2 == 0 else "ftrl"                # half GroupAdam, half SparseGroupFtrl
    card = int(np.exp(rng.uniform(np.log(1e2), np.log(max_card))))       # Criteo-like: log-uniform cardinalities
    tabs.append(_Table(ops, D, opt, card, seed=100 + j))
  groups = {}
  for t in tabs:
    groups.setdefault((t.opt, t.D), []).append(t)
  assert len(groups) == 10
  refs = {k: g[0].oracle() for k, g in groups.items()}                   # first table of every (optimizer, dim) group
  seen = {k: {} for k in groups}                                         # key -> largest count in any step so far
  b1p, b2p = np.float32(0.9), np.float32(0.999)
  nclean = 0
  for step in range(steps):
    for (opt, D), g in groups.items():
      ids = [np.minimum(rng.zipf(1.2, ids_per_table), t.card).astype(np.int64) * 2654435761 % (t.card * 7 + 13) for t in g]
      grads = [rng.normal(0, 1e-2, (ids_per_table, D)).astype(np.float32) for _ in g]       # two-signed
      outs = ops.kv_multi_gather_or_insert([t.h for t in g], ids)
      ov, osl = refs[(opt, D)]
      want = ov.gather_or_insert(ids[0])
      if step == 0:
        np.testing.assert_array_equal(outs[0].cpu().numpy(), want)       # untouched rows: bit-exact
      u, sm, inv = ko.dedup_segment_sum(ids[0], grads[0])
      cnt = np.bincount(inv, minlength=u.size)
      x_before = ov.gather_or_zeros(u).astype(np.float64)
      if opt == "adam":
        mvz = osl[0].gather_or_zeros(u).astype(np.float64)               # zeros for keys the slot has not met
        st = (mvz[:, :D], mvz[:, D:2 * D])
        hp = {"b1": float(np.float32(0.9)), "b2": float(np.float32(0.999)), "eps": float(np.float32(1e-8)),
              "alpha": float(np.float32(1e-2)) * np.sqrt(1 - float(b2p)) / (1 - float(b1p))}
        ops.kv_multi_group_sparse_apply_adam([t.h for t in g], [t.slots[0] for t in g], grads, ids, 1e-2, b1p, b2p,
                                             0.9, 0.999, 1e-8, 0, 0, 0)
        ko.apply_group_adam(ov, osl[0], sm, u, 1e-2, b1p, b2p, 0.9, 0.999, 1e-8)
      else:
        acc = osl[0].gather_or_zeros(u).astype(np.float64)
        acc = np.where(osl[0].get_count(u)[:, None] > 0, acc, 0.1)       # a new accumulator row starts at 0.1
        st = (acc, osl[1].gather_or_zeros(u).astype(np.float64))
        hp = {"lr": float(np.float32(0.1)), "l2": float(np.float32(1e-3))}
        ops.kv_multi_sparse_group_sparse_apply_ftrl([t.h for t in g], [t.slots[0] for t in g], [t.slots[1] for t in g],
                                                    grads, ids, 0.1, 0.0, 1e-3, 0.0, 0.0, -0.5)
        ko.apply_sparse_group_ftrl(ov, osl[0], osl[1], sm, u, 0.1, 0.0, 1e-3, 0.0, 0.0, -0.5)
      # first table of the group against the oracle: key set and frequency exact
      h0 = g[0].h
      assert ops.kv_variable_shape_v2(h0)[0] == ov.map_size() and ops.kv_variable_frequency(h0) == ov.sum_freq()
      for hs, os_ in zip(g[0].slots, osl):
        assert ops.kv_variable_shape_v2(hs)[0] == os_.map_size() and ops.kv_variable_frequency(hs) == os_.sum_freq()
      got = ops.kv_variable_gather_or_zeros_v2(h0, u).cpu().numpy()
      exp = ov.gather_or_zeros(u)
      sn = seen[(opt, D)]
      clean = np.array([sn.get(int(k), 1) == 1 for k in u]) & (cnt == 1)  # never part of a reordered sum so far
      nclean += int(clean.sum())
      np.testing.assert_allclose(got[clean], exp[clean], rtol=1e-6, atol=1e-9)
      if step == 0:
        # ids that repeat: |fl(sum, any order) - sum| <= (cnt - 1) 2^-24 sum|g|, pushed through the update
        gabs = np.zeros((u.size, D)); np.add.at(gabs, inv, np.abs(grads[0]).astype(np.float64))
        dg = (cnt[:, None] - 1).clip(min=0) * 2.0 ** -24 * gabs
        bound = 1e-6 * np.abs(exp) + 1e-9 + _reorder_bound(opt, x_before, st, sm.astype(np.float64), dg, hp)
        bad = np.argwhere(np.abs(got - exp) > bound)
        if bad.size:
          i, e = bad[0]
          raise AssertionError("%s D=%d key %d elem %d cnt %d: got %.9g exp %.9g bound %.3g dg %.3g gsum %.9g x0 %.9g st %s" % (
              opt, D, u[i], e, cnt[i], got[i, e], exp[i, e], bound[i, e], dg[i, e], sm[i, e], x_before[i, e], [float(q[i, e]) for q in st]))
      for k, c in zip(u.tolist(), cnt.tolist()):
        sn[k] = max(sn.get(k, 1), c)
    b1p, b2p = np.float32(b1p * np.float32(0.9)), np.float32(b2p * np.float32(0.999))
  assert nclean > 100 * steps
  return sum(ops.kv_variable_shape_v2(t.h)[0] for t in tabs)


@pytest.mark.gpu
def test_config4_one_rank_share_32_tables(ops):
  keys = _run_config4(ops, 32, 32768, 4e7)
  assert keys > 32 * 100


@pytest.mark.gpu
def test_config4_all_256_tables(ops):
  keys = _run_config4(ops, 256, 4096, 4e7, steps=2)
  assert keys > 256 * 50


@pytest.mark.gpu
def test_apply_four_million_ids_in_one_call(ops):
  """one optimizer call takes more than 2^21 ids (the reference's op shards any N, training_ops.cc:7205-7208): 4 M ids
  over 300 k keys, the lookup's token and the apply's own index pass, against the oracle's sum-then-apply."""
  rng = np.random.default_rng(SEED + 9)
  D, N = 8, 4_000_000
  table = rng.standard_normal((64, D)).astype(np.float32)
  ids = (rng.zipf(1.3, N) % 300_000).astype(np.int64)
  sign = rng.choice([-1.0, 1.0], (1, D))
  grad = (rng.uniform(0.5, 1.5, (N, D)) * 1e-2 * sign).astype(np.float32)   # one-signed: no cancelling sums
  ov = ko.OracleKv(D, 0, table, day=DAY, picker=1, seed=5)
  os_ = ko.OracleKv(3 * D, 0, np.zeros((4, 3 * D), np.float32), day=DAY, picker=1, seed=5)
  want_rows = ov.gather_or_insert(ids)
  u, sm, _ = ko.dedup_segment_sum(ids, grad)
  ko.apply_group_adam(ov, os_, sm, u, 1e-2, 0.9, 0.999, 0.9, 0.999, 1e-8)
  exp = ov.gather_or_zeros(u)
  ids_t, grad_t = torch.from_numpy(ids).cuda(), torch.from_numpy(grad).cuda()
  for with_token in (True, False):
    hv = ops.kv_variable([D]); hs = ops.kv_variable([3 * D])
    for h, t in ((hv, table), (hs, np.zeros((4, 3 * D), np.float32))):
      ops.kv_set_clock_days(h, DAY); ops.kv_set_seed(h, 5); ops.init_kv_variable_v2(h, t)
    out = ops.kv_variable_gather_or_insert_v2(hv, ids_t)
    np.testing.assert_array_equal(out.cpu().numpy(), want_rows)
    assert ops.kv_variable_frequency(hv) == ov.sum_freq()
    ops.kv_variable_group_sparse_apply_adam_v4(hv, hs, grad_t, ids_t if with_token else ids_t.clone(), 1e-2, 0.9, 0.999, 0.9,
                                               0.999, 1e-8, 0, 0, 0)
    got = ops.kv_variable_gather_or_zeros_v2(hv, u).cpu().numpy()
    # sums of up to ~1e5 one-signed addends in another order: a few 1e-6 of the sum, pushed through Adam
    np.testing.assert_allclose(got, exp, rtol=2e-4, atol=2e-6)
    order = np.argsort(u)                                # u comes in the oracle's first-occurrence order
    once = np.empty(u.size, bool)
    once[order] = np.unique(ids, return_counts=True)[1] == 1
    assert once.sum() > 1000
    np.testing.assert_allclose(got[once], exp[once], rtol=1e-6, atol=1e-9)
