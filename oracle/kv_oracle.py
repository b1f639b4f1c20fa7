"""ctypes front-end of the CPU oracle (oracle/kv_oracle.cc).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (tfplus_amd/) must never import it.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libkv_oracle.so")

_i64p = ctypes.POINTER(ctypes.c_int64)
_i32p = ctypes.POINTER(ctypes.c_int32)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_f32p = ctypes.POINTER(ctypes.c_float)


def build(force=False):
  src = os.path.join(_HERE, "kv_oracle.cc")
  if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
    subprocess.check_call(["make", "-C", _HERE, "-s"])
  return _SO


_lib = None


def lib():
  global _lib
  if _lib is not None:
    return _lib
  L = ctypes.CDLL(build())
  L.kvo_create.restype = ctypes.c_void_p
  L.kvo_create.argtypes = [ctypes.c_int, ctypes.c_int]
  L.kvo_destroy.argtypes = [ctypes.c_void_p]
  L.kvo_init_table.argtypes = [ctypes.c_void_p, _f32p, ctypes.c_int64]
  L.kvo_is_initialized.argtypes = [ctypes.c_void_p]
  L.kvo_set_picker.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64]
  L.kvo_set_day.argtypes = [ctypes.c_void_p, ctypes.c_int]
  L.kvo_srand.argtypes = [ctypes.c_uint]
  L.kvo_gather_or_insert.argtypes = [ctypes.c_void_p, _i64p, _i32p, ctypes.c_int64, _f32p,
                                     ctypes.c_int]
  L.kvo_gather_or_zeros.argtypes = [ctypes.c_void_p, _i64p, ctypes.c_int64, _f32p, ctypes.c_int]
  L.kvo_bulk_build.argtypes = [ctypes.c_void_p, _i64p, ctypes.c_int64, ctypes.c_int]
  L.kvo_apply_group_adam.restype = ctypes.c_int
  L.kvo_apply_group_adam.argtypes = [ctypes.c_void_p, ctypes.c_void_p, _f32p, _i64p,
                                     ctypes.c_int64] + [ctypes.c_float] * 9 + [ctypes.c_int,
                                                                              ctypes.c_int]
  L.kvo_apply_adagrad.restype = ctypes.c_int
  L.kvo_apply_adagrad.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, _f32p, _i64p,
                                  ctypes.c_int64, ctypes.c_int, ctypes.c_int]
  L.kvo_apply_sparse_group_ftrl.restype = ctypes.c_int
  L.kvo_apply_sparse_group_ftrl.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                            _f32p, _i64p, ctypes.c_int64] + [ctypes.c_float] * 6 + [
                                                ctypes.c_int]
  L.kvo_dedup_segment_sum.restype = ctypes.c_int64
  L.kvo_dedup_segment_sum.argtypes = [_i64p, _f32p, ctypes.c_int64, ctypes.c_int, _i64p, _f32p,
                                      _i32p]
  for f in ("kvo_size", "kvo_sum_freq", "kvo_map_size"):
    getattr(L, f).restype = ctypes.c_int64
    getattr(L, f).argtypes = [ctypes.c_void_p]
  L.kvo_export.argtypes = [ctypes.c_void_p, ctypes.c_int, _i64p, _i64p, _f32p, _i64p, _i64p, _u32p]
  L.kvo_scatter_update.argtypes = [ctypes.c_void_p, _i64p, _f32p, ctypes.c_int64, ctypes.c_int]
  L.kvo_insert.argtypes = [ctypes.c_void_p, _i64p, _f32p, ctypes.c_int64]
  L.kvo_import.argtypes = [ctypes.c_void_p, _i64p, _f32p, ctypes.c_int64, _i64p, ctypes.c_int64, _i64p,
                           _u32p, ctypes.c_int64]
  L.kvo_get_count.argtypes = [ctypes.c_void_p, _i64p, ctypes.c_int64, _i32p]
  L.kvo_get_timestamp.argtypes = [ctypes.c_void_p, _i64p, ctypes.c_int64, _u32p]
  L.kvo_after_export.argtypes = [ctypes.c_void_p, ctypes.c_int]
  L.kvo_set_delta_tracking.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
  L.kvo_export_delta.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _i64p, _i64p, _f32p, _i64p, _i64p, _u32p, _i64p]
  L.kvo_delete.restype = ctypes.c_int64
  L.kvo_delete.argtypes = [ctypes.c_void_p, _i64p, ctypes.c_int64]
  L.kvo_delete_with_timestamp.restype = ctypes.c_int64
  L.kvo_delete_with_timestamp.argtypes = [ctypes.c_void_p, ctypes.c_int, _i64p]
  L.kvo_import_delta.argtypes = [ctypes.c_void_p, _i64p, _f32p, ctypes.c_int64, _i64p, ctypes.c_int64, _i64p, _u32p,
                                 ctypes.c_int64, _i64p, ctypes.c_int64, ctypes.c_int]
  L.kvo_get_meta.restype = ctypes.c_int
  L.kvo_get_meta.argtypes = [ctypes.c_void_p, ctypes.c_int64, _u32p, ctypes.POINTER(ctypes.c_int),
                             ctypes.POINTER(ctypes.c_int)]
  L.kvo_murmur64a.restype = ctypes.c_uint64
  L.kvo_murmur64a.argtypes = [ctypes.c_int64]
  L.kvo_murmur64b.restype = ctypes.c_uint64
  L.kvo_murmur64b.argtypes = [ctypes.c_int64]
  _lib = L
  return L


def _p(a, t):
  return a.ctypes.data_as(t) if a is not None else None


def _ids(a):
  return np.ascontiguousarray(np.asarray(a).reshape(-1), dtype=np.int64)


def _f32(a):
  return np.ascontiguousarray(a, dtype=np.float32)


class OracleKv:
  """One reference-semantics KvVariable<int64, float> on the host."""

  def __init__(self, dim, enter_threshold=0, init_table=None, day=None, picker=1, seed=0,
               threads=1):
    self.dim = int(dim)
    self.threads = int(threads)
    self._h = lib().kvo_create(self.dim, int(enter_threshold))
    lib().kvo_set_picker(self._h, int(picker), int(seed))
    if day is not None:
      lib().kvo_set_day(self._h, int(day))
    if init_table is not None:
      self.init(init_table)

  def __del__(self):
    if getattr(self, "_h", None):
      lib().kvo_destroy(self._h)
      self._h = None

  def init(self, table):
    t = _f32(table)
    assert t.ndim == 2 and t.shape[1] == self.dim
    lib().kvo_init_table(self._h, _p(t, _f32p), t.shape[0])

  def is_initialized(self):
    return bool(lib().kvo_is_initialized(self._h))

  def gather_or_insert(self, ids, counts=None):
    shape = np.asarray(ids).shape
    ids = _ids(ids)
    c = None if counts is None else np.ascontiguousarray(np.asarray(counts).reshape(-1), np.int32)
    out = np.empty((ids.size, self.dim), np.float32)
    lib().kvo_gather_or_insert(self._h, _p(ids, _i64p), _p(c, _i32p), ids.size, _p(out, _f32p),
                               self.threads)
    return out.reshape(shape + (self.dim,))

  def bulk_build(self, ids):
    """The table state gather_or_insert(ids) leaves, built without the per-id lock traffic (one thread per hash
    segment): the UNTIMED set-up of bench.py's cpu_baseline.  With picker=0 the rows depend on std::rand() order."""
    ids = _ids(ids)
    lib().kvo_bulk_build(self._h, _p(ids, _i64p), ids.size, self.threads)

  def gather_or_zeros(self, ids):
    shape = np.asarray(ids).shape
    ids = _ids(ids)
    out = np.empty((ids.size, self.dim), np.float32)
    lib().kvo_gather_or_zeros(self._h, _p(ids, _i64p), ids.size, _p(out, _f32p), self.threads)
    return out.reshape(shape + (self.dim,))

  def scatter_update(self, ids, updates, op=0):
    i, u = _ids(ids), _f32(updates)
    lib().kvo_scatter_update(self._h, _p(i, _i64p), _p(u, _f32p), i.size, int(op))

  def insert(self, ids, values):
    i, v = _ids(ids), _f32(values)
    lib().kvo_insert(self._h, _p(i, _i64p), _p(v, _f32p), i.size)

  def import_(self, keys, values, blacklist=(), freq_keys=(), freq_values=(), first_n=6):
    # the op hands ImportValues empty tensors for the lists beyond first_n (kernels/kv_variable_ops.cc:806-822)
    if first_n <= 3:
      blacklist = ()
    if first_n <= 4:
      freq_keys, freq_values = (), ()
    k, v = _ids(keys), _f32(values)
    b, fk = _ids(blacklist), _ids(freq_keys)
    fv = np.ascontiguousarray(np.asarray(freq_values).reshape(-1), dtype=np.uint32)
    lib().kvo_import(self._h, _p(k, _i64p), _p(v, _f32p), k.size, _p(b, _i64p), b.size, _p(fk, _i64p),
                     _p(fv, _u32p), fk.size)

  def get_count(self, ids):
    i = _ids(ids)
    out = np.empty(i.size, np.int32)
    lib().kvo_get_count(self._h, _p(i, _i64p), i.size, _p(out, _i32p))
    return out.reshape(np.asarray(ids).shape)

  def get_timestamp(self, ids):
    i = _ids(ids)
    out = np.empty(i.size, np.uint32)
    lib().kvo_get_timestamp(self._h, _p(i, _i64p), i.size, _p(out, _u32p))
    return out.reshape(np.asarray(ids).shape)

  def delete(self, ids):
    i = _ids(ids)
    return int(lib().kvo_delete(self._h, _p(i, _i64p), i.size))

  def delete_with_timestamp(self, threshold):
    n = int(lib().kvo_delete_with_timestamp(self._h, int(threshold), None))
    out = np.empty(n, np.int64)
    if n:
      lib().kvo_delete_with_timestamp(self._h, int(threshold), _p(out, _i64p))
    return out

  def set_day(self, day):
    lib().kvo_set_day(self._h, int(day))

  def import_delta(self, keys, values, blacklist=(), freq_keys=(), freq_values=(), delete_keys=(), first_n=6):
    k, v = _ids(keys), _f32(values)
    b, fk, dk = _ids(blacklist), _ids(freq_keys), _ids(delete_keys)
    fv = np.ascontiguousarray(np.asarray(freq_values).reshape(-1), dtype=np.uint32)
    lib().kvo_import_delta(self._h, _p(k, _i64p), _p(v, _f32p), k.size, _p(b, _i64p), b.size, _p(fk, _i64p),
                           _p(fv, _u32p), fk.size, _p(dk, _i64p), dk.size, int(first_n))

  def size(self):
    return int(lib().kvo_size(self._h))

  def sum_freq(self):
    return int(lib().kvo_sum_freq(self._h))

  def map_size(self):
    return int(lib().kvo_map_size(self._h))

  def export(self, first_n=2):
    cnt = np.zeros(3, np.int64)
    lib().kvo_export(self._h, first_n, _p(cnt, _i64p), None, None, None, None, None)
    keys = np.empty(cnt[0], np.int64)
    vals = np.empty((cnt[0], self.dim), np.float32)
    bl = np.empty(cnt[1], np.int64)
    fk = np.empty(cnt[2], np.int64)
    fv = np.empty(cnt[2], np.uint32)
    lib().kvo_export(self._h, first_n, _p(cnt, _i64p), _p(keys, _i64p), _p(vals, _f32p),
                     _p(bl, _i64p), _p(fk, _i64p), _p(fv, _u32p))
    lib().kvo_after_export(self._h, first_n)
    return keys, vals, bl, fk, fv

  def set_delta_tracking(self, on=True, pred_on=False):
    lib().kvo_set_delta_tracking(self._h, int(on), int(pred_on))

  def export_delta(self, first_n=6):
    """DeltaExport (dynamic_save.hpp:198-451): (keys, values, blacklist, freq_keys, freq_values, delete_keys)."""
    cnt = np.zeros(4, np.int64)
    lib().kvo_export_delta(self._h, first_n, 0, _p(cnt, _i64p), None, None, None, None, None, None)
    keys = np.empty(cnt[0], np.int64)
    vals = np.empty((cnt[0], self.dim), np.float32)
    bl = np.empty(cnt[1], np.int64)
    fk = np.empty(cnt[2], np.int64)
    fv = np.empty(cnt[2], np.uint32)
    dk = np.empty(cnt[3], np.int64)
    lib().kvo_export_delta(self._h, first_n, 1, _p(cnt, _i64p), _p(keys, _i64p), _p(vals, _f32p), _p(bl, _i64p),
                           _p(fk, _i64p), _p(fv, _u32p), _p(dk, _i64p))
    return keys, vals, bl, fk, fv, dk

  def as_dict(self):
    k, v, *_ = self.export(2)
    return {int(a): b for a, b in zip(k, v)}

  def meta(self, key):
    f = ctypes.c_uint32()
    b = ctypes.c_int()
    u = ctypes.c_int()
    if not lib().kvo_get_meta(self._h, int(key), ctypes.byref(f), ctypes.byref(b), ctypes.byref(u)):
      return None
    return {"freq": f.value & 0xFFFF, "day": f.value >> 16, "blacklist": bool(b.value),
            "under_threshold": bool(u.value)}


def apply_group_adam(var, slot, grad, ids, lr, beta1_power, beta2_power, beta1, beta2, epsilon,
                     l1=0.0, l2=0.0, l21=0.0, version=4):
  g = _f32(grad)
  i = _ids(ids)
  rc = lib().kvo_apply_group_adam(var._h, slot._h, _p(g, _f32p), _p(i, _i64p), i.size, lr,
                                  beta1_power, beta2_power, beta1, beta2, epsilon, l1, l2, l21,
                                  version, var.threads)
  if rc:
    raise ValueError("kvo_apply_group_adam rc=%d" % rc)


def apply_adagrad(var, accum, lr, grad, ids, update_slots=True):
  g = _f32(grad)
  i = _ids(ids)
  rc = lib().kvo_apply_adagrad(var._h, accum._h, lr, _p(g, _f32p), _p(i, _i64p), i.size,
                               int(update_slots), var.threads)
  if rc:
    raise ValueError("kvo_apply_adagrad rc=%d" % rc)


def apply_sparse_group_ftrl(var, accum, linear, grad, ids, lr, l1, l2, l21, l2_shrinkage,
                            lr_power):
  g = _f32(grad)
  i = _ids(ids)
  rc = lib().kvo_apply_sparse_group_ftrl(var._h, accum._h, linear._h, _p(g, _f32p), _p(i, _i64p),
                                         i.size, lr, l1, l2, l21, l2_shrinkage, lr_power,
                                         var.threads)
  if rc:
    raise ValueError("kvo_apply_sparse_group_ftrl rc=%d" % rc)


def dedup_segment_sum(ids, grads):
  """tf.unique + tf.unsorted_segment_sum of TF-core's _deduplicate_indexed_slices."""
  i = _ids(ids)
  g = _f32(grads).reshape(i.size, -1)
  D = g.shape[1]
  uniq = np.empty(i.size, np.int64)
  summed = np.empty((i.size, D), np.float32)
  pos = np.empty(i.size, np.int32)
  U = lib().kvo_dedup_segment_sum(_p(i, _i64p), _p(g, _f32p), i.size, D, _p(uniq, _i64p),
                                  _p(summed, _f32p), _p(pos, _i32p))
  return uniq[:U].copy(), summed[:U].copy(), pos
