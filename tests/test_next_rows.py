"""scatter family, insert, import / export — oracle KATs on the CPU,
parity on the GPU."""
import os

import numpy as np
import pytest

from oracle import kv_oracle as ko

DAY = 20000
# KvVariableTest.ScatterUpdate (kernels/kv_variable_test.cc:226-357): keys 0..9, dim 64, random
# init table; assign 1 -> 1; add 1 -> 2; sub 1 -> 1; mul 2 -> 2; div 2 -> 1; min 2 -> 1; max 2 -> 2
S1_CHAIN = [(0, 1.0, 1.0), (1, 1.0, 2.0), (2, 1.0, 1.0), (3, 2.0, 2.0), (4, 2.0, 1.0), (5, 2.0, 1.0), (6, 2.0, 2.0)]


def test_S1_scatter_chain_oracle():
  rng = np.random.default_rng(0)
  kv = ko.OracleKv(64, 0, rng.random((1024, 64)).astype(np.float32), day=DAY)
  keys = np.arange(10)
  for op, upd, want in S1_CHAIN:
    kv.scatter_update(keys, np.full((10, 64), upd, np.float32), op)
    np.testing.assert_array_equal(kv.gather_or_zeros(keys), np.full((10, 64), want, np.float32))
  assert kv.map_size() == 10 and kv.sum_freq() == 10      # inserted with freq word 1, never bumped


def test_import_export_roundtrip_oracle():
  rng = np.random.default_rng(1)
  a = ko.OracleKv(8, 2, rng.standard_normal((64, 8)).astype(np.float32), day=DAY)
  a.gather_or_insert(rng.integers(0, 50, 400))
  s = ko.OracleKv(24, 0, np.zeros((4, 24), np.float32), day=DAY)
  g = (rng.standard_normal((50, 8)) * rng.uniform(1e-4, 3e-2, (50, 1))).astype(np.float32)
  ko.apply_group_adam(a, s, g, np.arange(50), 0.05, 0.9, 0.999, 0.9, 0.999, 1e-8, 1e-4, 1e-2, 4e-3)  # some rows blacklist
  k, v, bl, fk, fv = a.export(first_n=6)
  assert len(bl) > 0 and len(k) > 0
  b = ko.OracleKv(8, 2, np.zeros((4, 8), np.float32), day=DAY)
  b.import_(k, v, bl, fk, fv)
  k2, v2, bl2, fk2, fv2 = b.export(first_n=6)
  assert dict(zip(k, map(bytes, v))) == dict(zip(k2, map(bytes, v2)))
  assert sorted(bl) == sorted(bl2) and dict(zip(fk, fv)) == dict(zip(fk2, fv2))
  assert a.size() == b.size() and a.sum_freq() == b.sum_freq()


# ----------------------------------------------------------------------------------------------------
torch = pytest.importorskip("torch")


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


def _same(ops, h, o, keys):
  keys = np.unique(np.asarray(keys, np.int64))
  np.testing.assert_array_equal(ops.kv_variable_gather_or_zeros_v2(h, keys).cpu().numpy(), o.gather_or_zeros(keys))
  assert ops.kv_get_meta(h, keys) == [o.meta(int(k)) for k in keys]
  assert ops.kv_variable_shape_v2(h) == [o.map_size(), o.dim]
  assert ops.kv_variable_size_v2(h) == o.size() and ops.kv_variable_frequency(h) == o.sum_freq()


@pytest.mark.gpu
def test_S1_scatter_chain_gpu(ops):
  h, _ = _pair(ops, 64, rows=1024)
  keys = np.arange(10)
  fns = [ops.kv_variable_scatter_update_v2, ops.kv_variable_scatter_add_v2, ops.kv_variable_scatter_sub_v2,
         ops.kv_variable_scatter_mul_v2, ops.kv_variable_scatter_div_v2, ops.kv_variable_scatter_min_v2,
         ops.kv_variable_scatter_max_v2]
  for (op, upd, want), fn in zip(S1_CHAIN, fns):
    fn(h, keys, np.full((10, 64), upd, np.float32))
    assert bool((ops.kv_variable_gather_or_zeros_v2(h, keys) == want).all())


@pytest.mark.gpu
@pytest.mark.parametrize("D", [1, 8, 32, 50])
def test_scatter_and_insert_parity(ops, D):
  rng = np.random.default_rng(70 + D)
  h, o = _pair(ops, D, thr=2)
  warm = rng.integers(0, 300, 2000)
  ops.kv_variable_gather_or_insert_v2(h, warm)
  o.gather_or_insert(warm)
  seen = [warm]
  for op in range(7):
    ids = rng.choice(600, 250, replace=False).astype(np.int64)      # hits and fresh keys
    upd = rng.uniform(0.5, 2.0, (ids.size, D)).astype(np.float32)
    ops._scatter(op)(h, ids, upd)
    o.scatter_update(ids, upd, op)
    seen.append(ids)
    _same(ops, h, o, np.concatenate(seen))
  ids = rng.choice(900, 400, replace=False).astype(np.int64)
  vals = rng.standard_normal((ids.size, D)).astype(np.float32)
  vals[::7] = 0.0                                                   # all-zero rows -> under_threshold
  ops.kv_variable_insert_v2(h, ids, vals)
  o.insert(ids, vals)
  _same(ops, h, o, np.concatenate(seen + [ids]))


@pytest.mark.gpu
def test_import_export_parity(ops):
  rng = np.random.default_rng(5)
  D = 16
  hv, ov = _pair(ops, D, thr=2)
  hs, os_ = _pair(ops, 3 * D, seed=9)
  for t in (hs, os_):
    pass
  ids = rng.integers(0, 400, 3000)
  ops.kv_variable_gather_or_insert_v2(hv, ids)
  ov.gather_or_insert(ids)
  u = np.unique(ids)
  g = (rng.standard_normal((u.size, D)) * rng.uniform(1e-4, 3e-2, (u.size, 1))).astype(np.float32)
  hz = ops.kv_variable([3 * D]); ops.init_kv_variable_v2(hz, np.zeros((4, 3 * D), np.float32)); ops.kv_set_clock_days(hz, DAY)
  oz = ko.OracleKv(3 * D, 0, np.zeros((4, 3 * D), np.float32), day=DAY)
  ops.kv_variable_group_sparse_apply_adam_v4(hv, hz, g, u, 0.05, 0.9, 0.999, 0.9, 0.999, 1e-8, 1e-4, 1e-2, 5e-3)
  ko.apply_group_adam(ov, oz, g, u, 0.05, 0.9, 0.999, 0.9, 0.999, 1e-8, 1e-4, 1e-2, 5e-3)
  for first_n in (2, 3, 4, 6):
    gk, gv, gb, gfk, gfv = [x.cpu().numpy() for x in ops.kv_variable_export(hv, first_n=first_n)]
    ok, ovv, ob, ofk, ofv = ov.export(first_n=first_n)
    assert sorted(gk) == sorted(ok) and sorted(gb) == sorted(ob)
    assert dict(zip(gfk, gfv.view(np.uint32))) == dict(zip(ofk, ofv))
    got, exp = dict(zip(gk, gv)), dict(zip(ok, ovv))
    # x = u (1 - l21n / ||u||) / y: the norm's fp32 reduction order (sequential in the oracle, a shuffle tree on the
    # GPU) moves ||u|| by an ulp, which the subtraction amplifies by 1 / scale — tolerance per row = 1e-6 / scale,
    # as in test_gpu_parity.py::test_group_adam_parity_with_regularizers_and_blacklist; nothing else is allowed
    lr, l1s, l21n = np.float32(0.05), np.float32(1e-4 * 0.05), np.float32(5e-3 * 0.05) * np.sqrt(np.float32(D))
    z = oz.gather_or_zeros(ok)[:, 2 * D:]
    nrm = np.sqrt(((np.clip(z, -l1s, l1s) - z).astype(np.float64) ** 2).sum(1))
    scale = np.abs(1.0 - l21n / np.maximum(nrm, 1e-30))
    for k, sc in zip(ok, scale):
      if sc > 1e-4:      # rows closer than that to the blacklist threshold may legitimately flip
        np.testing.assert_allclose(got[k], exp[k], rtol=1e-6 / sc, atol=1e-9)
  assert len(gb) > 20 and len(gk) > 20                # both populations present
  # import what the GPU exported into a fresh GPU table and into a fresh oracle table
  h2 = ops.kv_variable([D], enter_threshold=2); ops.kv_set_clock_days(h2, DAY)
  o2 = ko.OracleKv(D, 2, np.zeros((4, D), np.float32), day=DAY)
  assert not ops.kv_variable_is_initialized_v2(h2)
  ops.kv_variable_import(h2, gk, gv, gb, gfk, gfv.view(np.uint32))
  o2.import_(gk, gv, gb, gfk, gfv.view(np.uint32))
  assert ops.kv_variable_is_initialized_v2(h2)
  keys = np.concatenate([gk, gb, np.array([10**6])])
  _same(ops, h2, o2, keys)
  k3, v3, b3, fk3, fv3 = [x.cpu().numpy() for x in ops.kv_variable_export(h2, first_n=6)]
  assert dict(zip(k3, map(bytes, v3))) == dict(zip(gk, map(bytes, gv))) and sorted(b3) == sorted(gb)
  # keys below enter_threshold are in neither keys nor blacklist, so their frequency words have no
  # key to land on (UpdateWithFn finds nothing, dynamic_restore.hpp:232-246): compare with the oracle
  _, _, _, ofk3, ofv3 = o2.export(first_n=6)
  assert dict(zip(fk3, fv3.view(np.uint32))) == dict(zip(ofk3, ofv3))
  assert set(fk3) <= set(gfk)
  # importing again replaces the content (table_->clear(), dynamic_restore.hpp:178)
  ops.kv_variable_import(h2, gk[:5], gv[:5])
  assert ops.kv_variable_shape_v2(h2) == [5, D]


@pytest.mark.gpu
def test_I1_import_v2_known_answer_gpu(ops, golden_dir):
  """The reference's own KAT for import / export (py_ut/tests/test_kv_variable_ops.py:345-435), on the GPU
  table and against the oracle for the contents the reference does not assert."""
  z = np.load(os.path.join(golden_dir, "I1_import_v2.npz"))
  D = z["values"].shape[1]
  thr = int(z["enter_threshold"][0])
  h = ops.kv_variable([D], enter_threshold=thr); ops.kv_set_clock_days(h, DAY)
  ops.init_kv_variable_v2(h, np.ones((1024, D), np.float32))
  o = ko.OracleKv(D, thr, np.ones((1024, D), np.float32), day=DAY)
  for i, first_n in enumerate(z["first_n"]):
    ops.kv_variable_import(h, z["keys"], z["values"], z["blacklist"], z["freq_keys"], z["freq_values"], first_n=int(first_n))
    o.import_(z["keys"], z["values"], z["blacklist"], z["freq_keys"], z["freq_values"], first_n=int(first_n))
    k, v, b, fk, fv = [x.cpu().numpy() for x in ops.kv_variable_export(h, first_n=6)]
    assert k.shape == (z["expect_rows"][i],) and v.shape == (z["expect_rows"][i], D)
    assert b.shape == (z["expect_blacklist"][i],) and fk.shape == fv.shape == (z["expect_freq"][i],)
    assert dict(zip(k, map(bytes, v))) == dict(zip(z["keys"], map(bytes, z["values"])))     # exact
    ok, ov, ob, ofk, ofv = o.export(first_n=6)
    assert sorted(b) == sorted(ob) and dict(zip(fk, fv.view(np.uint32))) == dict(zip(ofk, ofv))


@pytest.mark.gpu
@pytest.mark.parametrize("op", [1, 2])                     # ScatterAdd / ScatterSub
def test_scatter_add_sub_repeated_ids_accumulate(ops, op):
  """ScatterUpdate walks the indices one by one (kv_variable.h:616-734): repeated ids add up."""
  h, o = _pair(ops, 16)
  rng = np.random.default_rng(21)
  ids = rng.integers(-20, 20, 3000)                         # ~75 occurrences per id
  ops.kv_variable_gather_or_insert_v2(h, ids); o.gather_or_insert(ids)
  upd = rng.uniform(0.5, 1.5, (ids.size, 16)).astype(np.float32)
  (ops.kv_variable_scatter_add_v2 if op == 1 else ops.kv_variable_scatter_sub_v2)(h, ids, upd)
  o.scatter_update(ids, upd, op)
  q = np.arange(-20, 20)
  np.testing.assert_allclose(ops.kv_variable_gather_or_zeros_v2(h, q).cpu().numpy(), o.gather_or_zeros(q),
                             rtol=2e-6, atol=1e-6)
  assert ops.kv_variable_frequency(h) == o.sum_freq()


@pytest.mark.gpu
@pytest.mark.parametrize("op", [3, 4, 5, 6])               # ScatterMul / Div / Min / Max
def test_scatter_mul_div_min_max_repeated_ids_apply_every_occurrence(ops, op):
  """Every occurrence of a repeated id counts for every operation (kv_variable.h:616-734 walks the indices);
  min / max are exact whatever the order, mul / div multiply the same factors in another order."""
  h, o = _pair(ops, 16)
  rng = np.random.default_rng(23 + op)
  ids = rng.integers(-20, 20, 400)                          # ~10 occurrences per id
  ops.kv_variable_gather_or_insert_v2(h, ids); o.gather_or_insert(ids)
  upd = rng.uniform(0.9, 1.1, (ids.size, 16)).astype(np.float32)
  [None, None, None, ops.kv_variable_scatter_mul_v2, ops.kv_variable_scatter_div_v2, ops.kv_variable_scatter_min_v2,
   ops.kv_variable_scatter_max_v2][op](h, ids, upd)
  o.scatter_update(ids, upd, op)
  q = np.arange(-20, 20)
  got, exp = ops.kv_variable_gather_or_zeros_v2(h, q).cpu().numpy(), o.gather_or_zeros(q)
  if op >= 5:
    np.testing.assert_array_equal(got, exp)                 # min / max: bit-exact
  else:
    np.testing.assert_allclose(got, exp, rtol=2e-6, atol=0)  # ~10 roundings of a product, taken in another order
  assert ops.kv_variable_frequency(h) == o.sum_freq()


def test_delta_import_oracle_semantics():
  """DeltaImport (dynamic_restore.hpp:29-155): no clear, blacklist lifted on re-imported keys,
  blacklist marked or (first_n <= 3) deleted, frequency only on existing keys, delete_keys removed."""
  kv = ko.OracleKv(4, 0, np.zeros((4, 4), np.float32), day=DAY)
  kv.import_([1, 2, 3], np.ones((3, 4), np.float32), blacklist=[3, 9])
  assert sorted(kv.as_dict()) == [1, 2] and kv.map_size() == 4
  kv.import_delta([3, 7], np.full((2, 4), 5, np.float32), blacklist=[2], freq_keys=[1, 7, 100], freq_values=[(5 << 16) | 9, 11, 12],
                  delete_keys=[1, 55])
  d = kv.as_dict()
  assert sorted(d) == [3, 7] and float(d[3][0]) == 5.0       # 3 came back from the blacklist, 1 deleted, 2 blacklisted
  assert kv.meta(7)["freq"] == 11 and kv.meta(100) is None and kv.meta(2)["blacklist"] and kv.meta(9)["blacklist"]
  kv.import_delta([], np.zeros((0, 4), np.float32), blacklist=[9, 3], first_n=3)    # inference mode: blacklist keys vanish
  assert kv.meta(9) is None and kv.meta(3) is None and sorted(kv.as_dict()) == [7]


@pytest.mark.gpu
@pytest.mark.parametrize("first_n", [6, 3])
def test_delta_import_parity(ops, first_n):
  h, o = _pair(ops, 8, thr=2)
  rng = np.random.default_rng(31)
  ids = rng.integers(-60, 60, 500)
  ops.kv_variable_gather_or_insert_v2(h, ids); o.gather_or_insert(ids)
  s_h, s_o = _pair(ops, 24, seed=4)
  g = (rng.standard_normal((60, 8)) * rng.uniform(1e-4, 3e-2, (60, 1))).astype(np.float32)
  uids = np.arange(-30, 30)
  ops.kv_variable_group_sparse_apply_adam_v4(h, s_h, g, uids, 0.05, 0.9, 0.999, 0.9, 0.999, 1e-8, 1e-4, 1e-2, 4e-3)
  ko.apply_group_adam(o, s_o, g, uids, 0.05, 0.9, 0.999, 0.9, 0.999, 1e-8, 1e-4, 1e-2, 4e-3)          # some rows blacklist
  keys = rng.integers(-80, 80, 90); keys = np.unique(keys)
  vals = rng.standard_normal((keys.size, 8)).astype(np.float32)
  vals[::7] = 0.0                                                                                     # under_threshold rows
  bl = np.unique(rng.integers(-80, 120, 25)); bl = bl[~np.isin(bl, keys)]
  fk = np.unique(rng.integers(-100, 100, 40)); fv = rng.integers(1, 2**31, fk.size).astype(np.uint32)
  dk = np.unique(rng.integers(-100, 100, 30)); dk = dk[~np.isin(dk, keys)]
  ops.kv_variable_full_or_delta_import(h, keys, vals, bl, fk, fv, need_full_import=False, delete_keys=dk, first_n=first_n)
  o.import_delta(keys, vals, bl, fk, fv, dk, first_n=first_n)
  k2, v2, bl2, fk2, fv2 = ops.kv_variable_export(h, first_n=6)
  ok, ov, obl, ofk, ofv = o.export(6)
  assert dict(zip(k2.cpu().numpy().tolist(), map(bytes, v2.cpu().numpy()))) == dict(zip(ok.tolist(), map(bytes, ov)))
  assert sorted(bl2.cpu().numpy().tolist()) == sorted(obl.tolist())
  assert dict(zip(fk2.cpu().numpy().tolist(), fv2.cpu().numpy().view(np.uint32).tolist())) == dict(zip(ofk.tolist(), ofv.tolist()))
  assert ops.kv_variable_size_v2(h) == o.size() and ops.kv_variable_shape_v2(h)[0] == o.map_size()
  # a full import through the same op clears first
  ops.kv_variable_full_or_delta_import(h, keys[:5], vals[:5], need_full_import=True); o.import_(keys[:5], vals[:5])
  assert ops.kv_variable_shape_v2(h)[0] == o.map_size() == 5
  d = ops.kv_variable_full_or_delta_export(h, do_full_export=False)   # nothing tracked: an empty delta
  assert d[5] is False and d[0].numel() == 0 and d[6].numel() == 0
  assert ops.kv_variable_full_or_delta_export(h, True)[5] is True


@pytest.mark.gpu
def test_export_fill_refuses_buffers_sized_before_another_op(ops):
  """The two-phase calls hand the library buffers sized from an earlier count: a lookup in between (new rows) must
  make the fill fail instead of writing past them (kvhip.h: kv_export_count / kv_export_fill)."""
  import ctypes
  import torch
  from tfplus_amd import _lib
  h, _ = _pair(ops, 8)
  ops.kv_variable_gather_or_insert_v2(h, np.arange(100, dtype=np.int64))
  L = _lib.lib()
  st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
  cnt = (ctypes.c_int64 * 3)()
  _lib.check(L.kv_export_count(h.ptr, 2, cnt, st))
  assert cnt[0] == 100
  keys = torch.empty(100, dtype=torch.int64, device="cuda"); vals = torch.empty((100, 8), device="cuda")
  ops.kv_variable_gather_or_insert_v2(h, np.arange(100, 5000, dtype=np.int64))       # 4900 more rows
  rc = L.kv_export_fill(h.ptr, 2, keys.data_ptr(), vals.data_ptr(), None, None, None, st)
  assert rc == _lib.KV_FAILED_PRECONDITION
  assert "count again" in L.kv_last_error().decode()
  k, v = ops.read_kv_variable_op_v2(h)                                               # count + fill back to back
  assert k.numel() == 5000
  # the timed delete: a dry run sizes the key buffer
  n = ctypes.c_int64()
  _lib.check(L.kv_delete_with_timestamp(h.ptr, 0, 1, None, ctypes.byref(n), st))
  ops.kv_variable_gather_or_insert_v2(h, np.arange(5000, 5100, dtype=np.int64))
  out = torch.empty(max(int(n.value), 1), dtype=torch.int64, device="cuda")
  assert L.kv_delete_with_timestamp(h.ptr, 0, 0, out.data_ptr(), ctypes.byref(n), st) != 0
