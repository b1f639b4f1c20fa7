"""Op-level host interface: one function per reference custom op, same names, argument
order and error behaviour as the wrappers `tf.load_op_library("_kv_variable_ops.so")`
generates in the reference (tfplus/kv_variable/python/ops/kv_variable_ops.py:74,
REGISTER_OPs in tfplus/kv_variable/ops/{kv_variable_ops,training_ops}.cc).

Tensors are torch tensors living in the GPU's HBM (host arrays are copied over); the
`table_handle` is a KvHandle around the opaque C pointer.  Every op runs on torch's current
HIP stream for the table's device, so it orders with surrounding torch work.  All compute
goes through libkvhip.so (include/kvhip.h); there is no CPU path.
"""
import ctypes

import numpy as np
import torch

from tfplus_amd import _lib

_TORCH_KEY = {torch.int64: _lib.KV_DT_INT64, torch.int32: _lib.KV_DT_INT32}


class KvHandle(object):
  """The DT_RESOURCE scalar of the reference (`table_handle`)."""

  def __init__(self, ptr, dim, key_dtype, value_dtype, device, enter_threshold, name):
    self.ptr = ptr
    self.dim = dim
    self.key_dtype = key_dtype
    self.value_dtype = value_dtype
    self.device = device
    self.enter_threshold = enter_threshold
    self.name = name
    self.batch = None   # (token, ids data_ptr, n, ids version) of the last training lookup

  def __del__(self):
    try:
      if self.ptr:
        _lib.lib().kv_destroy(self.ptr)
        self.ptr = None
    except Exception:  # interpreter shutdown
      pass


def _dev(handle):
  return torch.device("cuda", handle.device)


def _stream(handle):
  return ctypes.c_void_p(torch.cuda.current_stream(_dev(handle)).cuda_stream)


def _ids(handle, indices):
  t = torch.as_tensor(indices)
  if t.dtype != handle.key_dtype:
    t = t.to(handle.key_dtype)
  return t.to(_dev(handle)).contiguous()


def _remember_batch(handle, ids, token):
  """The lookup left the index of `ids` in the table's workspace (kv_gather_or_insert_tok).  An optimizer op
  that is handed the very same tensor OBJECT, unmodified since, passes the token on.  The tensor is kept
  alive here: an address comparison alone would mistake a new tensor in recycled memory for the old one."""
  handle.batch = (token, ids, ids._version) if token else None


def _token_for(handle, ids):
  b = handle.batch
  if b and b[1] is ids and b[2] == ids._version:
    return b[0]
  return 0


def _f32(handle, x):
  return torch.as_tensor(x, dtype=torch.float32).to(_dev(handle)).contiguous()


def _p(t):
  return ctypes.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else ctypes.c_void_p(0)


def _scalar(x):
  if isinstance(x, torch.Tensor):
    return float(x.detach().to(torch.float32).item())
  return float(np.float32(x))


# ---- lifecycle --------------------------------------------------------------------------------
def kv_variable(value_shape, key_dtype=torch.int64, value_dtype=torch.float32, container="",
                shared_name="", use_node_name_sharing=False, key_shape=(), enter_threshold=0,
                capacity_hint=0, device=None, name=None):
  """REGISTER_OP("KvVariable") ops/kv_variable_ops.cc:37-74 (and V2–V4 :78-201)."""
  if value_dtype != torch.float32:
    raise _lib.UnimplementedError("value_dtype %s: only float32 rows are supported" % value_dtype)
  if key_dtype not in _TORCH_KEY:
    raise _lib.InvalidArgumentError("key_dtype %s: int32 / int64 only" % key_dtype)
  dim = int(np.prod(list(value_shape))) if len(tuple(value_shape)) else 1
  if device is None:
    device = torch.cuda.current_device()
  device = torch.device(device).index if not isinstance(device, int) else device
  out = ctypes.c_void_p()
  _lib.check(_lib.lib().kv_create(_TORCH_KEY[key_dtype], _lib.KV_DT_FLOAT, dim, int(enter_threshold),
                                  int(capacity_hint), int(device), ctypes.byref(out)))
  return KvHandle(out.value, dim, key_dtype, value_dtype, int(device),
                  min(int(enter_threshold), 65535), shared_name or name or "kv_variable")


kv_variable_v2 = kv_variable
kv_variable_v3 = kv_variable
kv_variable_v4 = kv_variable


def init_kv_variable_v2(table_handle, random_initializer, name=None):
  """REGISTER_OP("InitKvVariableV2") ops/kv_variable_ops.cc:212-222."""
  t = _f32(table_handle, random_initializer)
  if t.dim() != 2 or t.shape[1] != table_handle.dim:
    raise _lib.InvalidArgumentError("random_initializer must be [rows, %d], got %s" %
                                    (table_handle.dim, tuple(t.shape)))
  _lib.check(_lib.lib().kv_init_table(table_handle.ptr, _p(t), t.shape[0], _stream(table_handle)))


def kv_variable_is_initialized_v2(table_handle, name=None):
  out = ctypes.c_int()
  _lib.check(_lib.lib().kv_is_initialized(table_handle.ptr, ctypes.byref(out)))
  return bool(out.value)


def destroy_kv_variable_op_v2(table_handle, ignore_lookup_error=True, name=None):
  if table_handle.ptr:
    _lib.check(_lib.lib().kv_destroy(table_handle.ptr))
    table_handle.ptr = None


def _i64_out(fn, handle):
  out = ctypes.c_int64()
  _lib.check(fn(handle.ptr, ctypes.byref(out), _stream(handle)))
  return out.value


def kv_variable_shape_v2(table_handle, out_type=torch.int64, name=None):
  """[number of keys in the map, dim] (kernels/kv_variable_ops.cc:159-177)."""
  return [_i64_out(_lib.lib().kv_map_size, table_handle), table_handle.dim]


kv_variable_shape = kv_variable_shape_v2


def kv_variable_size_v2(table_handle, T=torch.int64, name=None):
  return _i64_out(_lib.lib().kv_size, table_handle)


def kv_variable_frequency(table_handle, name=None):
  return _i64_out(_lib.lib().kv_sum_freq, table_handle)


# ---- lookup -------------------------------------------------------------------------------------
def _gather_out(table_handle, ids):
  return torch.empty(tuple(ids.shape) + (table_handle.dim,), dtype=torch.float32,
                     device=_dev(table_handle))


def kv_variable_gather_or_insert_v2(table_handle, indices, dtype=torch.float32, name=None):
  """REGISTER_OP("KvVariableGatherOrInsertV2") ops/kv_variable_ops.cc:310-320."""
  ids = _ids(table_handle, indices)
  out = _gather_out(table_handle, ids)
  tok = ctypes.c_uint64(0)
  _lib.check(_lib.lib().kv_gather_or_insert_tok(table_handle.ptr, _p(ids), None, ids.numel(), _p(out),
                                                ctypes.byref(tok), _stream(table_handle)))
  _remember_batch(table_handle, ids, tok.value)
  return out


def kv_variable_gather_or_insert_with_counts(table_handle, indices, counts, dtype=torch.float32,
                                             name=None):
  """REGISTER_OP("KvVariableGatherOrInsertWithCounts") ops/kv_variable_ops.cc:322-332."""
  ids = _ids(table_handle, indices)
  cnt = torch.as_tensor(counts)
  if cnt.dtype != torch.int32:
    # kv_variable.h:276-280
    raise _lib.InvalidArgumentError("KvVariable %s: increment count, counts dtype must be int32" %
                                    table_handle.name)
  if tuple(cnt.shape) != tuple(ids.shape):
    # kv_variable.h:268-274
    raise _lib.InvalidArgumentError(
        "KvVariable %s: increment count, indices shape %s does not match with counts shape %s" %
        (table_handle.name, tuple(ids.shape), tuple(cnt.shape)))
  cnt = cnt.to(_dev(table_handle)).contiguous()
  out = _gather_out(table_handle, ids)
  tok = ctypes.c_uint64(0)
  _lib.check(_lib.lib().kv_gather_or_insert_tok(table_handle.ptr, _p(ids), _p(cnt), ids.numel(), _p(out),
                                                ctypes.byref(tok), _stream(table_handle)))
  _remember_batch(table_handle, ids, tok.value)
  return out


def kv_variable_gather_or_zeros_v2(table_handle, indices, dtype=torch.float32, name=None):
  """REGISTER_OP("KvVariableGatherOrZerosV2") ops/kv_variable_ops.cc:285-295."""
  ids = _ids(table_handle, indices)
  out = _gather_out(table_handle, ids)
  _lib.check(_lib.lib().kv_gather_or_zeros(table_handle.ptr, _p(ids), ids.numel(), _p(out),
                                           _stream(table_handle)))
  return out


kv_variable_gather_v2 = kv_variable_gather_or_zeros_v2


# ---- optimizers ---------------------------------------------------------------------------------
def _grad_ids(var, grad, indices):
  ids = _ids(var, indices)
  if ids.dim() != 1:
    raise _lib.InvalidArgumentError("indices must be one-dimensional")
  g = _f32(var, grad)
  if g.dim() < 1 or g.shape[0] != ids.shape[0]:
    raise _lib.InvalidArgumentError("grad must be the same size as indices in the first dimension.")
  if g.numel() != ids.shape[0] * var.dim:
    raise _lib.InvalidArgumentError("var and grad must match in dimension 1")
  return g, ids


def _group_adam(version, var, m_v_linear, grad, indices, lr, beta1_power, beta2_power, beat1, beta2,
                epsilon, l1, l2, l21, use_locking, unique_indices=False):
  g, ids = _grad_ids(var, grad, indices)
  if unique_indices:   # the caller's promise (what TF-core's de-duplication guarantees in the reference's graph): kvhip.h
    _lib.check(_lib.lib().kv_apply_group_adam_unique(
        var.ptr, m_v_linear.ptr, _p(g), _p(ids), ids.numel(), _scalar(lr), _scalar(beta1_power),
        _scalar(beta2_power), _scalar(beat1), _scalar(beta2), _scalar(epsilon), _scalar(l1), _scalar(l2),
        _scalar(l21), version, _stream(var)))
    return
  _lib.check(_lib.lib().kv_apply_group_adam_tok(
      var.ptr, m_v_linear.ptr, _p(g), _p(ids), ids.numel(), _scalar(lr), _scalar(beta1_power),
      _scalar(beta2_power), _scalar(beat1), _scalar(beta2), _scalar(epsilon), _scalar(l1), _scalar(l2),
      _scalar(l21), version, _token_for(var, ids), _stream(var)))


def kv_variable_group_sparse_apply_adam_v4(var, m_v_linear, grad, indices, lr, beta1_power,
                                           beta2_power, beat1, beta2, epsilon, l1, l2, l21,
                                           use_locking=False, name=None, unique_indices=False):
  """REGISTER_OP("KvVariableGroupSparseApplyAdamV4") ops/training_ops.cc:1266-1285.
  unique_indices: the caller's promise that `indices` holds no id twice (kv_apply_group_adam_unique)."""
  _group_adam(4, var, m_v_linear, grad, indices, lr, beta1_power, beta2_power, beat1, beta2, epsilon,
              l1, l2, l21, use_locking, unique_indices)


def kv_variable_group_sparse_apply_adam_v3(var, m_v_linear, grad, indices, lr, beta1_power,
                                           beta2_power, beat1, beta2, epsilon, l1, l2, l21,
                                           use_locking=False, name=None, unique_indices=False):
  """REGISTER_OP("KvVariableGroupSparseApplyAdamV3") ops/training_ops.cc:1086-1105."""
  _group_adam(3, var, m_v_linear, grad, indices, lr, beta1_power, beta2_power, beat1, beta2, epsilon,
              l1, l2, l21, use_locking, unique_indices)


def kv_variable_sparse_apply_adagrad(var, accum, lr, grad, indices, use_locking=False,
                                     update_slots=True, name=None, unique_indices=False):
  """REGISTER_OP("KvVariableSparseApplyAdagrad") ops/training_ops.cc:214-226."""
  g, ids = _grad_ids(var, grad, indices)
  if unique_indices:
    _lib.check(_lib.lib().kv_apply_adagrad_unique(var.ptr, accum.ptr, _scalar(lr), _p(g), _p(ids), ids.numel(),
                                                  int(bool(update_slots)), _stream(var)))
    return
  _lib.check(_lib.lib().kv_apply_adagrad_tok(var.ptr, accum.ptr, _scalar(lr), _p(g), _p(ids), ids.numel(),
                                             int(bool(update_slots)), _token_for(var, ids), _stream(var)))


def kv_variable_sparse_group_sparse_apply_ftrl_v2(var, accum, linear, grad, indices, lr, l1, l2, l21,
                                                  l2_shrinkage, lr_power, use_locking=False,
                                                  name=None, unique_indices=False):
  """REGISTER_OP("KvVariableSparseGroupSparseApplyFtrlV2") ops/training_ops.cc:135-150."""
  g, ids = _grad_ids(var, grad, indices)
  if unique_indices:
    _lib.check(_lib.lib().kv_apply_sparse_group_ftrl_unique(
        var.ptr, accum.ptr, linear.ptr, _p(g), _p(ids), ids.numel(), _scalar(lr), _scalar(l1),
        _scalar(l2), _scalar(l21), _scalar(l2_shrinkage), _scalar(lr_power), _stream(var)))
    return
  _lib.check(_lib.lib().kv_apply_sparse_group_ftrl_tok(
      var.ptr, accum.ptr, linear.ptr, _p(g), _p(ids), ids.numel(), _scalar(lr), _scalar(l1),
      _scalar(l2), _scalar(l21), _scalar(l2_shrinkage), _scalar(lr_power), _token_for(var, ids), _stream(var)))


def kv_dedup_segment_sum(table_handle, indices, grad):
  """tf.unique + tf.unsorted_segment_sum (TF-core _deduplicate_indexed_slices) on the GPU.
  Returns (unique_ids [U], summed [U, dim], inverse [n])."""
  g, ids = _grad_ids(table_handle, grad, indices)
  n = ids.numel()
  dev = _dev(table_handle)
  uniq = torch.empty(n, dtype=torch.int64, device=dev)
  summed = torch.empty((n, table_handle.dim), dtype=torch.float32, device=dev)
  inv = torch.empty(n, dtype=torch.int32, device=dev)
  nu = ctypes.c_int64()
  _lib.check(_lib.lib().kv_dedup_segment_sum(table_handle.ptr, _p(ids), _p(g), n, _p(uniq), _p(summed),
                                             _p(inv), ctypes.byref(nu), _stream(table_handle)))
  return uniq[:nu.value], summed[:nu.value], inv


# ---- readback / export / import ---------------------------------------------------------------------
def kv_variable_export(table_handle, first_n=6, enable_cutoff=True, cutoff_value=1e-20, name=None):
  """REGISTER_OP("KvVariableExport") ops/kv_variable_ops.cc:360-390 -> ExportValues.
  Returns (keys, values, blacklist, freq_keys, freq_values); init_table is the caller's."""
  cnt = (ctypes.c_int64 * 3)()
  _lib.check(_lib.lib().kv_export_count(table_handle.ptr, int(first_n), cnt, _stream(table_handle)))
  dev = _dev(table_handle)
  keys = torch.empty(cnt[0], dtype=torch.int64, device=dev)
  vals = torch.empty((cnt[0], table_handle.dim), dtype=torch.float32, device=dev)
  bl = torch.empty(cnt[1], dtype=torch.int64, device=dev)
  fk = torch.empty(cnt[2], dtype=torch.int64, device=dev)
  fv = torch.empty(cnt[2], dtype=torch.int32, device=dev)
  _lib.check(_lib.lib().kv_export_fill(table_handle.ptr, int(first_n), _p(keys), _p(vals), _p(bl),
                                       _p(fk), _p(fv), _stream(table_handle)))
  return keys, vals, bl, fk, fv


def read_kv_variable_op_v2(table_handle, Tkeys=torch.int64, Tvalues=torch.float32, name=None):
  """REGISTER_OP("ReadKvVariableOpV2") -> ExportValues(first_n = 2): (keys, values)."""
  k, v, _, _, _ = kv_variable_export(table_handle, first_n=2)
  return k, v


def kv_variable_import(table_handle, keys, values, blacklist=None, freq_keys=None, freq_values=None,
                       first_n=6, name=None):
  """REGISTER_OP("KvVariableImport") ops/kv_variable_ops.cc:392-420 -> ImportValues.  The lists beyond
  first_n reach ImportValues as empty tensors (kernels/kv_variable_ops.cc:806-822)."""
  if first_n <= 3:
    blacklist = None
  if first_n <= 4:
    freq_keys, freq_values = None, None
  dev = _dev(table_handle)
  k = torch.as_tensor(keys, dtype=torch.int64).to(dev).contiguous()
  v = _f32(table_handle, values)
  if v.numel() != k.numel() * table_handle.dim:
    raise _lib.InvalidArgumentError("values must be [len(keys), %d]" % table_handle.dim)
  bl = None if blacklist is None else torch.as_tensor(blacklist, dtype=torch.int64).to(dev).contiguous()
  fk = None if freq_keys is None else torch.as_tensor(freq_keys, dtype=torch.int64).to(dev).contiguous()
  fv = None
  if freq_values is not None:
    fv = torch.as_tensor(np.asarray(freq_values).astype(np.uint32).view(np.int32)
                         if not isinstance(freq_values, torch.Tensor) else freq_values).to(dev).contiguous()
  _lib.check(_lib.lib().kv_import(table_handle.ptr, _p(k), _p(v), k.numel(), _p(bl),
                                  0 if bl is None else bl.numel(), _p(fk), _p(fv),
                                  0 if fk is None else fk.numel(), _stream(table_handle)))


def kv_variable_full_or_delta_import(table_handle, keys, values, blacklist=None, freq_keys=None, freq_values=None,
                                     need_full_import=True, delete_keys=None, first_n=6, name=None):
  """REGISTER_OP("KvVariableFullOrDeltaImport") ops/kv_variable_ops.cc:576-602: need_full_import ->
  ImportValues (the table is cleared first), otherwise DeltaImport (dynamic_restore.hpp:29-155)."""
  if need_full_import:
    return kv_variable_import(table_handle, keys, values, blacklist, freq_keys, freq_values, first_n)
  dev = _dev(table_handle)
  k = torch.as_tensor(keys, dtype=torch.int64).to(dev).contiguous()
  v = _f32(table_handle, values)
  if v.numel() != k.numel() * table_handle.dim:
    raise _lib.InvalidArgumentError("number of keys (%d) and number of values (%d) do not match" %
                                    (k.numel(), v.numel() // max(table_handle.dim, 1)))
  def i64(x):
    return None if x is None else torch.as_tensor(x, dtype=torch.int64).to(dev).contiguous()
  bl, fk, dk = i64(blacklist), i64(freq_keys), i64(delete_keys)
  fv = None
  if freq_values is not None:
    fv = torch.as_tensor(np.asarray(freq_values).astype(np.uint32).view(np.int32)
                         if not isinstance(freq_values, torch.Tensor) else freq_values).to(dev).contiguous()
  n = lambda t: 0 if t is None else t.numel()
  _lib.check(_lib.lib().kv_import_delta(table_handle.ptr, _p(k), _p(v), k.numel(), _p(bl), n(bl), _p(fk), _p(fv), n(fk),
                                        _p(dk), n(dk), int(first_n), _stream(table_handle)))


def kv_set_delta_tracking(table_handle, support_delta_export=True, support_prediction_delta_export=False):
  """The SUPPORT_DELTA_EXPORT / SUPPORT_PREDICTION_DELTA_EXPORT switches of the KvVariable constructor
  (kernels/kv_variable.h:100-111), per table instead of per process."""
  _lib.check(_lib.lib().kv_set_delta_tracking(table_handle.ptr, int(bool(support_delta_export)),
                                              int(bool(support_prediction_delta_export))))


def kv_variable_full_or_delta_export(table_handle, do_full_export=True, first_n=3, enable_cutoff=False, cutoff_value=0.0):
  """REGISTER_OP("KvVariableFullOrDeltaExport") ops/kv_variable_ops.cc:633-660 ->
  (keys, values, blacklist, freq_keys, freq_values, need_full_import, delete_keys).  do_full_export ->
  FullExport (need_full_import = True, no delete keys); otherwise DeltaExport (dynamic_save.hpp:198-451):
  only the keys touched since the last export (kv_set_delta_tracking), need_full_import = False."""
  if do_full_export:
    k, v, bl, fk, fv = kv_variable_export(table_handle, first_n=first_n)
    return k, v, bl, fk, fv, True, torch.empty(0, dtype=torch.int64, device=k.device)
  cnt = (ctypes.c_int64 * 4)()
  _lib.check(_lib.lib().kv_export_delta_count(table_handle.ptr, int(first_n), cnt, _stream(table_handle)))
  dev = _dev(table_handle)
  keys = torch.empty(cnt[0], dtype=torch.int64, device=dev)
  vals = torch.empty((cnt[0], table_handle.dim), dtype=torch.float32, device=dev)
  bl = torch.empty(cnt[1], dtype=torch.int64, device=dev)
  fk = torch.empty(cnt[2], dtype=torch.int64, device=dev)
  fv = torch.empty(cnt[2], dtype=torch.int32, device=dev)
  dk = torch.empty(cnt[3], dtype=torch.int64, device=dev)
  _lib.check(_lib.lib().kv_export_delta_fill(table_handle.ptr, int(first_n), _p(keys), _p(vals), _p(bl), _p(fk),
                                             _p(fv), _p(dk), _stream(table_handle)))
  return keys, vals, bl, fk, fv, False, dk


def kv_variable_insert_v2(table_handle, indices, values, name=None):
  """REGISTER_OP("KvVariableInsertV2") ops/kv_variable_ops.cc:334-347."""
  ids = _ids(table_handle, indices)
  v = _f32(table_handle, values)
  if v.numel() != ids.numel() * table_handle.dim:
    raise _lib.InvalidArgumentError("values must be indices.shape + [%d]" % table_handle.dim)
  _lib.check(_lib.lib().kv_insert(table_handle.ptr, _p(ids), _p(v), ids.numel(), _stream(table_handle)))


def _scatter(op):

  def fn(table_handle, indices, updates, name=None):
    ids = _ids(table_handle, indices)
    u = _f32(table_handle, updates)
    if u.numel() != ids.numel() * table_handle.dim:
      raise _lib.InvalidArgumentError("updates must be indices.shape + [%d]" % table_handle.dim)
    _lib.check(_lib.lib().kv_scatter_update(table_handle.ptr, _p(ids), _p(u), ids.numel(), op,
                                            _stream(table_handle)))

  return fn


# REGISTER_OP("KvVariableScatter*V2") ops/kv_variable_ops.cc:422-560
kv_variable_scatter_update_v2 = _scatter(0)
kv_variable_scatter_add_v2 = _scatter(1)
kv_variable_scatter_sub_v2 = _scatter(2)
kv_variable_scatter_mul_v2 = _scatter(3)
kv_variable_scatter_div_v2 = _scatter(4)
kv_variable_scatter_min_v2 = _scatter(5)
kv_variable_scatter_max_v2 = _scatter(6)


# ---- test hooks (no reference counterpart) ---------------------------------------------------------
def kv_set_clock_days(table_handle, day):
  _lib.check(_lib.lib().kv_set_clock_days(table_handle.ptr, int(day)))


def kv_set_seed(table_handle, seed):
  _lib.check(_lib.lib().kv_set_seed(table_handle.ptr, int(seed)))


def kv_get_meta(table_handle, indices):
  ids = torch.as_tensor(indices, dtype=torch.int64).to(_dev(table_handle)).contiguous().view(-1)
  fw = torch.empty(ids.numel(), dtype=torch.int32, device=ids.device)
  fl = torch.empty(ids.numel(), dtype=torch.uint8, device=ids.device)
  _lib.check(_lib.lib().kv_get_meta(table_handle.ptr, _p(ids), ids.numel(), _p(fw), _p(fl),
                                    _stream(table_handle)))
  fw = fw.cpu().numpy().view(np.uint32)
  fl = fl.cpu().numpy()
  return [None if not (b & 0x80) else {"freq": int(w & 0xFFFF), "day": int(w >> 16),
                                       "blacklist": bool(b & 1), "under_threshold": bool(b & 2)}
          for w, b in zip(fw, fl)]


def kv_reserve(table_handle, capacity):
  _lib.check(_lib.lib().kv_reserve(table_handle.ptr, int(capacity)))


PROF_KINDS = ("lookup_tile", "lookup_part", "lookup_order", "apply_index", "apply_sorted", "apply_span", "apply_tsum",
              "apply_unique", "apply_tile")   # include/kvhip.h KV_PROF_*


def kv_attach_slot(var, slot):
  """The var's index entries remember each key's row in `slot` (kvhip.h kv_attach_slot)."""
  _lib.check(_lib.lib().kv_attach_slot(var.ptr, slot.ptr, _stream(var)))


KV_ORDER_ARRIVAL, KV_ORDER_FIXED, KV_ORDER_OCCURRENCE = 0, 1, 2


def kv_set_deterministic(table_handle, on=True):
  """False / 0: the gradient rows of a repeated id are summed as they arrive; True / 1: in an order fixed by the input
  positions (bit-reproducible); 2 (KV_ORDER_OCCURRENCE): one by one in input order, the sum TF-core's unsorted_segment_sum
  takes on the CPU — bit for bit (include/kvhip.h kv_set_deterministic)."""
  _lib.check(_lib.lib().kv_set_deterministic(table_handle.ptr, int(on)))


def kv_set_fast_math(table_handle, on=True):
  """Opt-in: the optimizers' row math on the 1-ulp hardware sqrt / reciprocal instructions.  The LIBRARY default (never
  calling this, or on=False) is the IEEE sequences — bit-parity with the oracle from the same summed gradient; with
  on=True an element whose update cancels can leave rtol 1e-6 (kvhip.h kv_set_fast_math).  Ignored in deterministic mode."""
  _lib.check(_lib.lib().kv_set_fast_math(table_handle.ptr, int(bool(on))))


KV_STAT_MIRROR_APPLIES, KV_STAT_MIRROR_EPOCHS = 0, 1


def kv_get_stat(table_handle, which):
  """Counters of the table's own ops (kvhip.h kv_get_stat)."""
  v = ctypes.c_int64(0)
  _lib.check(_lib.lib().kv_get_stat(table_handle.ptr, int(which), ctypes.byref(v)))
  return int(v.value)


def kv_forget_stream(stream):
  """A torch stream the caller is about to drop: synchronised, then no table's next op refers to it (kvhip.h kv_forget_stream)."""
  _lib.check(_lib.lib().kv_forget_stream(ctypes.c_void_p(stream.cuda_stream)))


def kv_prepare_capture(table_handle, max_new_ids):
  """Refreshes the host's row-count bounds so that captured calls taking up to max_new_ids ids need no sync."""
  _lib.check(_lib.lib().kv_prepare_capture(table_handle.ptr, int(max_new_ids), _stream(table_handle)))


def kv_profile_enable(table_handle, max_launches):
  _lib.check(_lib.lib().kv_profile_enable(table_handle.ptr, int(max_launches)))


def kv_profile_select(table_handle, kinds=None):
  """Brackets only the named kernel kinds (None = all)."""
  mask = 0xFFFFFFFF if kinds is None else sum(1 << PROF_KINDS.index(k) for k in kinds)
  _lib.check(_lib.lib().kv_profile_select(table_handle.ptr, mask))


def kv_profile_sample(table_handle, every=1):
  """Bracket only every `every`-th launch of the selected kinds."""
  _lib.check(_lib.lib().kv_profile_sample(table_handle.ptr, int(every)))


def kv_profile_read(table_handle):
  """{kernel kind: (total ms, launches)} from the HIP events recorded since the last read."""
  n = len(PROF_KINDS)
  ms = (ctypes.c_double * n)()
  cnt = (ctypes.c_int64 * n)()
  _lib.check(_lib.lib().kv_profile_read(table_handle.ptr, ms, cnt, n))
  return {k: (ms[i], cnt[i]) for i, k in enumerate(PROF_KINDS)}


def kv_bucket_by_owner(table_handle, indices, world, n_dev=None, id_counts=None, with_payload=False, owner_rule=0):
  """Counting sort of the ids by owner rank floor_mod(id, world) on the GPU.
  Returns (ids grouped by owner, perm [n] int32 of input positions, counts [world] int64 on device).
  n_dev (1-element int64 device tensor): the list's real length when it is still on the device.
  with_payload: two more results — pairs [n, 2] int64 = (id, id_counts[i] or 1) in bucket order and
  pos [n] int32 = where input i went (the inverse of perm)."""
  ids = _ids(table_handle, indices).reshape(-1)
  dev = _dev(table_handle)
  out = torch.empty(ids.numel(), dtype=torch.int64, device=dev)
  perm = torch.empty(ids.numel(), dtype=torch.int32, device=dev)
  counts = torch.empty(int(world), dtype=torch.int64, device=dev)
  pairs = pos = cin = None
  if with_payload:
    pairs = torch.empty((ids.numel(), 2), dtype=torch.int64, device=dev)
    pos = torch.empty(ids.numel(), dtype=torch.int32, device=dev)
    cin = None if id_counts is None else id_counts.to(torch.int32).contiguous()
  _lib.check(_lib.lib().kv_bucket_by_owner(table_handle.ptr, _p(ids), ids.numel(), _p(n_dev), int(world), int(owner_rule), _p(out),
                                           _p(perm), ctypes.c_void_p(counts.data_ptr()), _p(cin), _p(pairs), _p(pos),
                                           _stream(table_handle)))
  return (out, perm, counts, pairs, pos) if with_payload else (out, perm, counts)


def kv_variable_get_count_v2(table_handle, indices):
  """KvVariableGetCountV2: int32 frequency of every id (0 when absent), shaped like indices."""
  ids = _ids(table_handle, indices)
  out = torch.empty(ids.shape, dtype=torch.int32, device=ids.device)
  _lib.check(_lib.lib().kv_get_count(table_handle.ptr, _p(ids), ids.numel(), _p(out), _stream(table_handle)))
  return out


def kv_variable_get_time_stamp(table_handle, indices):
  """KvVariableGetTimeStamp: day stamp of every id (today when absent); uint32 in the reference,
  carried as int64 here (torch has no uint32 arithmetic)."""
  ids = _ids(table_handle, indices)
  out = torch.empty(ids.shape, dtype=torch.int32, device=ids.device)
  _lib.check(_lib.lib().kv_get_timestamp(table_handle.ptr, _p(ids), ids.numel(), _p(out), _stream(table_handle)))
  return out.to(torch.int64) & 0xFFFFFFFF


def kv_variable_delete(table_handle, indices):
  """KvVariableDelete: removes the keys (absent ones are ignored)."""
  ids = _ids(table_handle, indices).reshape(-1)
  n = ctypes.c_int64()
  _lib.check(_lib.lib().kv_delete(table_handle.ptr, _p(ids), ids.numel(), ctypes.byref(n), _stream(table_handle)))
  return int(n.value)


def kv_variable_delete_with_timestamp(table_handle, threshold=7):
  """KvVariableDeleteWithTimestamp: removes keys last touched >= threshold days ago; returns them."""
  n = ctypes.c_int64()
  st = _stream(table_handle)
  _lib.check(_lib.lib().kv_delete_with_timestamp(table_handle.ptr, int(threshold), 1, None, ctypes.byref(n), st))
  keys = torch.empty(n.value, dtype=torch.int64, device=_dev(table_handle))
  if n.value:
    _lib.check(_lib.lib().kv_delete_with_timestamp(table_handle.ptr, int(threshold), 0, _p(keys), ctypes.byref(n), st))
    keys = keys[:n.value]
  return keys.to(table_handle.key_dtype) if table_handle.key_dtype != torch.int64 else keys


def batch_kv_variable_gather_or_zeros_v2(table_handles, indices):
  """BatchKvVariableGatherOrZerosV2: [GatherOrZeros(t, i) for t, i in zip(table_handles, indices)] with
  one kernel launch for all tables; each output is indices[i].shape + [dim_i]."""
  if len(table_handles) < 1 or len(table_handles) != len(indices):
    raise _lib.InvalidArgumentError("table_handles and indices must be equally long, N >= 1")
  n = len(table_handles)
  ids = [_ids(h, i) for h, i in zip(table_handles, indices)]
  outs = [torch.empty(tuple(i.shape) + (h.dim,), dtype=torch.float32, device=i.device) for h, i in zip(table_handles, ids)]
  hp = (ctypes.c_void_p * n)(*[h.ptr for h in table_handles])
  ip = (ctypes.c_void_p * n)(*[i.data_ptr() for i in ids])
  op = (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs])
  ns = (ctypes.c_int64 * n)(*[i.numel() for i in ids])
  _lib.check(_lib.lib().kv_batch_gather_or_zeros(n, hp, ip, ns, op, _stream(table_handles[0])))
  return outs


def _ptr_array(tensors):
  return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def kv_multi_gather_or_insert(table_handles, indices, counts=None):
  """[KvVariableGatherOrInsertV2 / ...WithCounts(t, i) for t, i in zip(...)] with three kernel launches
  for all tables (same dim and key dtype).  Each output is indices[i].shape + [dim]."""
  n = len(table_handles)
  if n < 1 or n != len(indices) or (counts is not None and len(counts) != n):
    raise _lib.InvalidArgumentError("table_handles, indices (and counts) must be equally long, N >= 1")
  ids = [_ids(h, i) for h, i in zip(table_handles, indices)]
  cnt = None if counts is None else [None if c is None else torch.as_tensor(c, dtype=torch.int32).to(i.device).reshape(-1).contiguous()
                                     for c, i in zip(counts, ids)]
  outs = [torch.empty(tuple(i.shape) + (h.dim,), dtype=torch.float32, device=i.device) for h, i in zip(table_handles, ids)]
  hp = (ctypes.c_void_p * n)(*[h.ptr for h in table_handles])
  ns = (ctypes.c_int64 * n)(*[i.numel() for i in ids])
  toks = (ctypes.c_uint64 * n)()
  _lib.check(_lib.lib().kv_multi_gather_or_insert_tok(n, hp, _ptr_array(ids), None if cnt is None else _ptr_array(cnt), ns,
                                                      _ptr_array(outs), toks, _stream(table_handles[0])))
  for h, i, t in zip(table_handles, ids, toks):   # the batched optimizer ops handed these very tensors skip their index pass
    _remember_batch(h, i, int(t))
  return outs


def _multi_tokens(var_handles, indices):
  """ids as the C ABI wants them (flat) and the batch tokens of a batched lookup over the very same tensor objects."""
  base = [_ids(h, i) for h, i in zip(var_handles, indices)]
  toks = (ctypes.c_uint64 * len(base))(*[_token_for(h, b) for h, b in zip(var_handles, base)])
  return [b.reshape(-1) for b in base], toks


def kv_multi_group_sparse_apply_adam(var_handles, m_v_linear_handles, grads, indices, lr, beta1_power, beta2_power,
                                     beat1, beta2, epsilon, l1, l2, l21, version=4, unique_indices=False):
  """KvVariableGroupSparseApplyAdamV4 (V3) on many (var, m_v_linear) pairs with two kernel launches (one with
  unique_indices: the caller's promise that no table's indices hold an id twice, kvhip.h kv_multi_apply_*_unique)."""
  n = len(var_handles)
  if n < 1 or not (n == len(m_v_linear_handles) == len(grads) == len(indices)):
    raise _lib.InvalidArgumentError("vars, slots, grads and indices must be equally long, N >= 1")
  ids, toks = _multi_tokens(var_handles, indices)
  gr = [_f32(h, g).reshape(-1, h.dim) for h, g in zip(var_handles, grads)]
  for g, i in zip(gr, ids):
    if g.shape[0] != i.numel():
      raise _lib.InvalidArgumentError("grad must be the same size as indices in the first dimension.")
  vp = (ctypes.c_void_p * n)(*[h.ptr for h in var_handles])
  sp = (ctypes.c_void_p * n)(*[h.ptr for h in m_v_linear_handles])
  ns = (ctypes.c_int64 * n)(*[i.numel() for i in ids])
  sc = [ctypes.c_float(_scalar(x)) for x in (lr, beta1_power, beta2_power, beat1, beta2, epsilon, l1, l2, l21)]
  if unique_indices:
    _lib.check(_lib.lib().kv_multi_apply_group_adam_unique(n, vp, sp, _ptr_array(gr), _ptr_array(ids), ns, *sc, int(version),
                                                           _stream(var_handles[0])))
    return
  _lib.check(_lib.lib().kv_multi_apply_group_adam_tok(n, vp, sp, _ptr_array(gr), _ptr_array(ids), ns, *sc, int(version),
                                                      toks, _stream(var_handles[0])))


def _multi_prep(var_handles, grads, indices):
  ids, toks = _multi_tokens(var_handles, indices)
  gr = [_f32(h, g).reshape(-1, h.dim) for h, g in zip(var_handles, grads)]
  for g, i in zip(gr, ids):
    if g.shape[0] != i.numel():
      raise _lib.InvalidArgumentError("grad must be the same size as indices in the first dimension.")
  n = len(var_handles)
  return ids, gr, (ctypes.c_int64 * n)(*[i.numel() for i in ids]), toks


def kv_multi_sparse_apply_adagrad(var_handles, accum_handles, lr, grads, indices, update_slots=True, unique_indices=False):
  """KvVariableSparseApplyAdagrad on many (var, accum) pairs with two kernel launches."""
  n = len(var_handles)
  if n < 1 or not (n == len(accum_handles) == len(grads) == len(indices)):
    raise _lib.InvalidArgumentError("vars, accums, grads and indices must be equally long, N >= 1")
  ids, gr, ns, toks = _multi_prep(var_handles, grads, indices)
  vp = (ctypes.c_void_p * n)(*[h.ptr for h in var_handles]); ap = (ctypes.c_void_p * n)(*[h.ptr for h in accum_handles])
  if unique_indices:
    _lib.check(_lib.lib().kv_multi_apply_adagrad_unique(n, vp, ap, ctypes.c_float(_scalar(lr)), _ptr_array(gr), _ptr_array(ids), ns,
                                                        int(bool(update_slots)), _stream(var_handles[0])))
    return
  _lib.check(_lib.lib().kv_multi_apply_adagrad_tok(n, vp, ap, ctypes.c_float(_scalar(lr)), _ptr_array(gr), _ptr_array(ids), ns,
                                                   int(bool(update_slots)), toks, _stream(var_handles[0])))


def kv_multi_sparse_group_sparse_apply_ftrl(var_handles, accum_handles, linear_handles, grads, indices, lr, l1, l2, l21,
                                            l2_shrinkage, lr_power, unique_indices=False):
  """KvVariableSparseGroupSparseApplyFtrlV2 on many (var, accum, linear) triples with two launches."""
  n = len(var_handles)
  if n < 1 or not (n == len(accum_handles) == len(linear_handles) == len(grads) == len(indices)):
    raise _lib.InvalidArgumentError("vars, accums, linears, grads and indices must be equally long, N >= 1")
  ids, gr, ns, toks = _multi_prep(var_handles, grads, indices)
  vp = (ctypes.c_void_p * n)(*[h.ptr for h in var_handles]); ap = (ctypes.c_void_p * n)(*[h.ptr for h in accum_handles])
  lp = (ctypes.c_void_p * n)(*[h.ptr for h in linear_handles])
  sc = [ctypes.c_float(_scalar(x)) for x in (lr, l1, l2, l21, l2_shrinkage, lr_power)]
  if unique_indices:
    _lib.check(_lib.lib().kv_multi_apply_sparse_group_ftrl_unique(n, vp, ap, lp, _ptr_array(gr), _ptr_array(ids), ns, *sc,
                                                                  _stream(var_handles[0])))
    return
  _lib.check(_lib.lib().kv_multi_apply_sparse_group_ftrl_tok(n, vp, ap, lp, _ptr_array(gr), _ptr_array(ids), ns, *sc,
                                                             toks, _stream(var_handles[0])))


_COMBINERS = {"sum": _lib.KV_COMBINER_SUM, "mean": _lib.KV_COMBINER_MEAN, "sqrtn": _lib.KV_COMBINER_SQRTN}


def kv_variable_lookup_sparse(table_handle, ids, segment_ids, weights, num_segments, combiner="mean",
                              count_occurrences=False):
  """embedding_lookup_sparse on one KvVariable in a single call: unique_with_counts ->
  GatherOrInsert[WithCounts] -> weighted segment sum / mean / sqrtn (embedding_ops.py:279-441).
  segment_ids ascending (sp_ids.indices[:, 0]); returns [num_segments, dim]."""
  if combiner not in _COMBINERS:
    raise ValueError("combiner must be one of 'mean', 'sqrtn' or 'sum'")
  ids = _ids(table_handle, ids).reshape(-1)
  dev = _dev(table_handle)
  seg = torch.as_tensor(segment_ids).to(dev).reshape(-1)
  if seg.dtype not in _TORCH_KEY:
    seg = seg.to(torch.int64)
  seg = seg.contiguous()
  if seg.numel() != ids.numel():
    raise _lib.InvalidArgumentError("segment_ids and ids must have the same length")
  w = None if weights is None else _f32(table_handle, weights).reshape(-1)
  if w is not None and w.numel() != ids.numel():
    raise _lib.InvalidArgumentError("sp_weights and sp_ids must have the same number of values")
  out = torch.empty((int(num_segments), table_handle.dim), dtype=torch.float32, device=dev)
  _lib.check(_lib.lib().kv_lookup_sparse(table_handle.ptr, _p(ids), _p(seg), _TORCH_KEY[seg.dtype], _p(w),
                                         ids.numel(), int(num_segments), _COMBINERS[combiner],
                                         int(bool(count_occurrences)), _p(out), _stream(table_handle)))
  return out


def kv_unsorted_segment_sum(table_handle, data, segment_ids, num_segments):
  """tf.unsorted_segment_sum on the GPU batch pipeline: [num_segments, dim] fp32."""
  d = _f32(table_handle, data).reshape(-1, table_handle.dim)
  seg = torch.as_tensor(segment_ids).to(_dev(table_handle)).reshape(-1).to(torch.int32).contiguous()
  if seg.numel() != d.shape[0]:
    raise _lib.InvalidArgumentError("segment_ids and data must have the same number of rows")
  out = torch.empty((int(num_segments), table_handle.dim), dtype=torch.float32, device=d.device)
  _lib.check(_lib.lib().kv_unsorted_segment_sum(table_handle.ptr, _p(seg), _p(d), seg.numel(), int(num_segments),
                                                _p(out), _stream(table_handle)))
  return out


def kv_take_rows(src, index, scatter=False, num_rows=None, index_outer=None):
  """out[i] = src[index[i]] (gather, default) or out[index[i]] = src[i] (scatter=True; index must be
  a permutation onto `num_rows` rows).  index_outer: out[i] = src[index[index_outer[i]]] (gather).
  Any 4-byte-multiple row type; indices int32 on the GPU."""
  src = src.contiguous()
  idx = index.to(torch.int32).contiguous()
  outer = None if index_outer is None else index_outer.to(torch.int32).contiguous()
  n = idx.numel() if outer is None else outer.numel()
  tail = tuple(src.shape[1:])
  row_bytes = src.element_size()
  for d in tail:
    row_bytes *= int(d)
  if scatter and src.shape[0] != n:
    raise _lib.InvalidArgumentError("kv_take_rows: scatter needs one index per source row")
  rows_out = (src.shape[0] if num_rows is None else int(num_rows)) if scatter else n
  out = torch.empty((rows_out,) + tail, dtype=src.dtype, device=src.device)
  stream = ctypes.c_void_p(torch.cuda.current_stream(src.device).cuda_stream)
  _lib.check(_lib.lib().kv_take_rows(src.device.index or 0, _p(src), _p(idx), _p(outer), n, row_bytes, int(bool(scatter)),
                                     _p(out), stream))
  return out


def kv_variable_gather_or_insert_pairs(table_handle, id_count_pairs):
  """KvVariableGatherOrInsertWithCounts fed with int64 (id, count) pairs [n, 2] (the sharded exchange payload)."""
  p = torch.as_tensor(id_count_pairs, dtype=torch.int64).to(_dev(table_handle)).contiguous()
  if p.dim() != 2 or p.shape[1] != 2:
    raise _lib.InvalidArgumentError("id_count_pairs must be [n, 2] int64")
  out = torch.empty((p.shape[0], table_handle.dim), dtype=torch.float32, device=p.device)
  _lib.check(_lib.lib().kv_gather_or_insert_pairs(table_handle.ptr, _p(p), p.shape[0], _p(out), _stream(table_handle)))
  return out


def kv_unique(table_handle, indices, counts=None, sync=True):
  """tf.unique_with_counts on the GPU: (unique ids [U], counts [U] int32, inverse [n] int32).
  sync=False leaves U on the device: returns (ids [n] of which the first U are valid, counts [n],
  inverse [n], U as a 1-element int64 device tensor) without waiting for the stream."""
  ids = _ids(table_handle, indices).reshape(-1)
  n = ids.numel()
  dev = _dev(table_handle)
  cnt = None if counts is None else torch.as_tensor(counts, dtype=torch.int32).to(dev).reshape(-1).contiguous()
  uniq = torch.empty(n, dtype=torch.int64, device=dev)
  ucnt = torch.empty(n, dtype=torch.int32, device=dev)
  inv = torch.empty(n, dtype=torch.int32, device=dev)
  if not sync:
    nu_dev = torch.empty(1, dtype=torch.int64, device=dev)
    _lib.check(_lib.lib().kv_unique(table_handle.ptr, _p(ids), _p(cnt), n, _p(uniq), _p(ucnt), _p(inv), None, _p(nu_dev),
                                    _stream(table_handle)))
    return uniq, ucnt, inv, nu_dev
  nu = ctypes.c_int64()
  _lib.check(_lib.lib().kv_unique(table_handle.ptr, _p(ids), _p(cnt), n, _p(uniq), _p(ucnt), _p(inv),
                                  ctypes.byref(nu), None, _stream(table_handle)))
  return uniq[:nu.value], ucnt[:nu.value], inv


# ---- sharded tables: the native path (kvhip.h kv_comm_* / kv_shard_*) -------------------------------------
KV_OWNER_HASH, KV_OWNER_MOD = 0, 1
OPT_GROUP_ADAM_V4, OPT_GROUP_ADAM_V3, OPT_ADAGRAD, OPT_SPARSE_GROUP_FTRL = 0, 1, 2, 3


class KvComm(object):
  """RCCL communicator of the library (grouped ncclSend / ncclRecv on its own stream)."""

  def __init__(self, world, rank, id128=None, device=0):
    self.ptr = ctypes.c_void_p()
    buf = None if id128 is None else ctypes.create_string_buffer(bytes(id128), 128)
    _lib.check(_lib.lib().kv_comm_create(int(world), int(rank), buf, int(device), ctypes.byref(self.ptr)))
    self.world, self.rank = world, rank

  def stream(self):
    """The communicator's stream as a torch stream: work queued there reaches the sharded ops without an event hop."""
    st = ctypes.c_void_p()
    _lib.check(_lib.lib().kv_comm_stream(self.ptr, ctypes.byref(st)))
    return torch.cuda.ExternalStream(st.value)

  def __del__(self):
    try:
      if self.ptr:
        _lib.lib().kv_comm_destroy(self.ptr)
        self.ptr = None
    except Exception:  # interpreter shutdown
      pass


class KvCommStaged(KvComm):
  """A communicator whose segments the CALLER moves (kvhip.h kv_comm_create_staged): the library synchronises its
  stream and calls back.  Default transport: a torch.distributed group on the HOST — device -> host ->
  all_to_all_single (gloo) -> device — for rehearsing the N > 1 ops where RCCL cannot run (ranks sharing one GPU).
  `exchange(send_ptr, recv_ptr, bytes_per_peer)` / `max_u32(value) -> value` replace it (with `world` and `rank`), e.g.
  ranks that are threads of one process.  Never a measurement path."""

  _XFN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64)
  _MFN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32))

  @staticmethod
  def raw(ptr, nbytes, dev):
    """`nbytes` of device memory at `ptr` as a flat uint8 tensor (zero copy)."""
    class _Raw(object):
      def __init__(self, p, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (p, False), "version": 2}
    return torch.as_tensor(_Raw(ptr, nbytes), device=dev)

  def __init__(self, device=0, group=None, world=None, rank=None, exchange=None, max_u32=None):
    dev = torch.device("cuda", device)
    if exchange is None:
      import torch.distributed as dist
      world, rank = dist.get_world_size(group), dist.get_rank(group)

      def exchange(send, recv, per_peer):
        n = int(per_peer) * world
        src = self.raw(send, n, dev).cpu()
        dst = torch.empty(n, dtype=torch.uint8)
        dist.all_to_all_single(dst, src, group=group)
        self.raw(recv, n, dev).copy_(dst)
        torch.cuda.synchronize(dev)

      def max_u32(value):
        t = torch.tensor([int(value)], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        return int(t.item())
    self.exchanges = 0

    def _x(_user, send, recv, per_peer):
      try:
        if int(per_peer) > 0:
          exchange(send, recv, int(per_peer))
        self.exchanges += 1
        return 0
      except Exception:   # an exception cannot cross the C frames: the op reports KV_INTERNAL
        import traceback
        traceback.print_exc()
        return 1

    def _m(_user, value):
      try:
        value[0] = int(max_u32(int(value[0])))
        return 0
      except Exception:
        import traceback
        traceback.print_exc()
        return 1

    self._cb = (self._XFN(_x), self._MFN(_m) if max_u32 is not None else None)   # kept alive as long as the communicator
    self.ptr = ctypes.c_void_p()
    _lib.check(_lib.lib().kv_comm_create_staged(int(world), int(rank), ctypes.cast(self._cb[0], ctypes.c_void_p),
                                                ctypes.cast(self._cb[1], ctypes.c_void_p) if self._cb[1] else None,
                                                None, int(device), ctypes.byref(self.ptr)))
    self.world, self.rank = world, rank


def kv_comm_unique_id():
  buf = ctypes.create_string_buffer(128)
  _lib.check(_lib.lib().kv_comm_unique_id(buf))
  return buf.raw


def kv_comm_from_torch_distributed(device, group=None):
  """One KvComm per rank of a torch.distributed group: rank 0's unique id travels through the group."""
  import torch.distributed as dist
  world, rank = dist.get_world_size(group), dist.get_rank(group)
  box = [kv_comm_unique_id() if rank == 0 else None]
  dist.broadcast_object_list(box, src=0, group=group)
  return KvComm(world, rank, box[0], device)


class KvShard(object):
  """This rank's side of a table sharded over `world` ranks (kvhip.h kv_shard_create)."""

  def __init__(self, table_handle, world, rank, owner_rule=KV_OWNER_HASH, max_ids=1 << 20, peer_capacity=0):
    self.table = table_handle
    self.ptr = ctypes.c_void_p()
    _lib.check(_lib.lib().kv_shard_create(table_handle.ptr, int(world), int(rank), int(owner_rule), int(max_ids),
                                          int(peer_capacity), ctypes.byref(self.ptr)))
    self.world, self.rank, self.owner_rule = world, rank, int(owner_rule)

  # A sharded table is saved as its ranks' own exports (kv_variable_export / full_or_delta_import on each rank's local
  # table) plus this record of WHO OWNS WHAT: rows restored under another world size or owner rule would sit on ranks
  # that never look them up (and be silently re-initialised there).  OWNER_RULE_VERSION names the arithmetic of the
  # "hash" rule — 2: (mix64(id) >> 32) % world, since round 3; 1 was mix64(id) % world.
  OWNER_RULE_VERSION = 2

  def manifest(self):
    """What a checkpoint of this shard must be restored under (store it next to the rank's export)."""
    return {"world": int(self.world), "rank": int(self.rank),
            "owner_rule": "mod" if self.owner_rule == KV_OWNER_MOD else "hash",
            "owner_rule_version": 1 if self.owner_rule == KV_OWNER_MOD else self.OWNER_RULE_VERSION}

  def check_manifest(self, m):
    """Raises ValueError when a checkpoint written under manifest `m` does not belong on this shard."""
    mine = self.manifest()
    bad = [k for k in mine if m.get(k) != mine[k]]
    if bad:
      raise ValueError("sharded checkpoint mismatch on %s: written under %r, this shard is %r — re-partition the rows "
                       "(read every rank's export, route the keys with sharded.owner_of) instead of importing them as they are"
                       % (", ".join(bad), {k: m.get(k) for k in mine}, mine))

  def __del__(self):
    try:
      if self.ptr:
        _lib.lib().kv_shard_destroy(self.ptr)
        self.ptr = None
    except Exception:  # interpreter shutdown
      pass

  PHASES = ("route", "exchange_ids", "serve", "exchange_rows", "finish", "presum", "exchange_grads", "apply")   # kvhip.h kv_shard_profile

  def profile(self, every=1):
    """Per-phase event timing of every `every`-th whole lookup / apply on the communicator's stream (0: off)."""
    _lib.check(_lib.lib().kv_shard_profile(self.ptr, int(every)))

  def profile_read(self):
    """{"phases_ms": {phase: mean ms per sampled step}, "samples": n, "rccl_ranks_seen", "peer_capacity", "grows", "overflows"}"""
    ms = (ctypes.c_double * len(self.PHASES))()
    cnt = (ctypes.c_int64 * len(self.PHASES))()
    info = (ctypes.c_int64 * 4)()
    _lib.check(_lib.lib().kv_shard_profile_read(self.ptr, ms, cnt, len(self.PHASES), info))
    return {"phases_ms": {p: (ms[i] / cnt[i] if cnt[i] else None) for i, p in enumerate(self.PHASES)},
            "samples": int(min(cnt)) if len(cnt) else 0, "rccl_ranks_seen": int(info[0]), "peer_capacity": int(info[1]),
            "grows": int(info[2]), "overflows": int(info[3])}

  def buffers(self):
    """The exchange buffers as flat uint8 tensors over the library's memory (zero copy): send_pairs, recv_pairs
    [world x pair_bytes], send_rows, recv_rows [world x row_bytes] — for a caller that moves the segments itself."""
    ptrs = [ctypes.c_void_p() for _ in range(4)]
    pb, rb = ctypes.c_int64(), ctypes.c_int64()
    _lib.check(_lib.lib().kv_shard_buffers(self.ptr, *[ctypes.byref(x) for x in ptrs], ctypes.byref(pb), ctypes.byref(rb)))

    class _Raw(object):
      def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}
    dev = _dev(self.table)
    names = ("send_pairs", "recv_pairs", "send_rows", "recv_rows")
    sizes = (pb.value, pb.value, rb.value, rb.value)
    out = {n: torch.as_tensor(_Raw(p.value, self.world * sz), device=dev) for n, p, sz in zip(names, ptrs, sizes)}
    out["pair_bytes_per_peer"], out["row_bytes_per_peer"] = pb.value, rb.value
    return out

  def set_lossless(self, on=True):
    """Ranks agree on the capacity before every exchange (one host round trip per lookup); nothing is ever dropped."""
    _lib.check(_lib.lib().kv_shard_set_lossless(self.ptr, int(bool(on))))

  @property
  def peer_capacity(self):
    return self.buffers()["pair_bytes_per_peer"] // 16 - 1

  # whole ops over a KvComm
  def lookup(self, comm, indices, join=True):
    ids = _ids(self.table, indices)
    out = _gather_out(self.table, ids)
    _lib.check(_lib.lib().kv_shard_lookup(self.ptr, comm.ptr, _p(ids), ids.numel(), _p(out), int(bool(join)), _stream(self.table)))
    self._keep = (ids, out)     # the shard's stream may still be reading / writing them
    return out

  def apply(self, comm, optimizer, slots, grad, hp, join=True):
    g = _f32(self.table, grad)
    arr = (ctypes.c_float * len(hp))(*[float(np.float32(x)) for x in hp])
    s0 = slots[0].ptr
    s1 = slots[1].ptr if len(slots) > 1 else None
    _lib.check(_lib.lib().kv_shard_apply(self.ptr, comm.ptr, int(optimizer), s0, s1, _p(g), arr, int(bool(join)), _stream(self.table)))
    self._keep_g = g

  def join(self):
    _lib.check(_lib.lib().kv_shard_join(self.ptr, _stream(self.table)))

  # the phases, for an exchange the caller provides
  def lookup_route(self, indices):
    ids = _ids(self.table, indices)
    self._ids = ids    # first: a late report of an earlier batch's overflow (raised below) leaves this batch routed
    _lib.check(_lib.lib().kv_shard_lookup_route(self.ptr, _p(ids), ids.numel(), _stream(self.table)))

  def lookup_serve(self):
    _lib.check(_lib.lib().kv_shard_lookup_serve(self.ptr, _stream(self.table)))

  def lookup_finish(self):
    out = _gather_out(self.table, self._ids)
    _lib.check(_lib.lib().kv_shard_lookup_finish(self.ptr, _p(out), _stream(self.table)))
    return out

  def apply_route(self, grad):
    g = _f32(self.table, grad)
    _lib.check(_lib.lib().kv_shard_apply_route(self.ptr, _p(g), _stream(self.table)))
    self._keep_g = g

  def apply_serve(self, optimizer, slots, hp):
    arr = (ctypes.c_float * len(hp))(*[float(np.float32(x)) for x in hp])
    s1 = slots[1].ptr if len(slots) > 1 else None
    _lib.check(_lib.lib().kv_shard_apply_serve(self.ptr, int(optimizer), slots[0].ptr, s1, arr, _stream(self.table)))


def kv_multi_shard_lookup(shards, comm, indices_list, join=True):
  """Sharded lookup of several tables in one step: two grouped exchanges whatever len(shards) is (kvhip.h
  kv_multi_shard_lookup).  Returns one [n_k, dim_k] tensor per table."""
  T = len(shards)
  ids = [_ids(sh.table, x) for sh, x in zip(shards, indices_list)]
  outs = [_gather_out(sh.table, i) for sh, i in zip(shards, ids)]
  arr = (ctypes.c_void_p * T)(*[sh.ptr for sh in shards])
  ip = (ctypes.c_void_p * T)(*[_p(i) for i in ids])
  nn = (ctypes.c_int64 * T)(*[i.numel() for i in ids])
  op = (ctypes.c_void_p * T)(*[_p(o) for o in outs])
  _lib.check(_lib.lib().kv_multi_shard_lookup(arr, T, comm.ptr, ip, nn, op, int(bool(join)), _stream(shards[0].table)))
  for sh, i, o in zip(shards, ids, outs):
    sh._keep = (i, o)
  return outs


def kv_multi_shard_apply(shards, comm, optimizer, slots_list, grads, hp, join=True):
  """The sharded optimizer apply of several tables: one grouped exchange (kvhip.h kv_multi_shard_apply).
  slots_list[k] = the slot table(s) of shards[k]."""
  T = len(shards)
  gs = [_f32(sh.table, g) for sh, g in zip(shards, grads)]
  arr = (ctypes.c_void_p * T)(*[sh.ptr for sh in shards])
  s0 = (ctypes.c_void_p * T)(*[sl[0].ptr for sl in slots_list])
  two = all(len(sl) > 1 for sl in slots_list)
  s1 = (ctypes.c_void_p * T)(*[sl[1].ptr for sl in slots_list]) if two else None
  gp = (ctypes.c_void_p * T)(*[_p(g) for g in gs])
  hpa = (ctypes.c_float * len(hp))(*[float(np.float32(x)) for x in hp])
  _lib.check(_lib.lib().kv_multi_shard_apply(arr, T, comm.ptr, int(optimizer), s0, s1, gp, hpa, int(bool(join)), _stream(shards[0].table)))
  for sh, g in zip(shards, gs):
    sh._keep_g = g


def kv_shard_agree_local(shards):
  """Lossless mode between shards of one process: True when the capacity was raised (route every shard again)."""
  arr = (ctypes.c_void_p * len(shards))(*[s.ptr for s in shards])
  again = ctypes.c_int32(0)
  _lib.check(_lib.lib().kv_shard_agree_local(arr, len(shards), ctypes.byref(again), _stream(shards[0].table)))
  return bool(again.value)


def kv_shard_exchange_local(shards, what):
  """Exchange between shards of one process on one device (what: 0 records, 1 rows)."""
  arr = (ctypes.c_void_p * len(shards))(*[s.ptr for s in shards])
  _lib.check(_lib.lib().kv_shard_exchange_local(arr, len(shards), int(what), _stream(shards[0].table)))
