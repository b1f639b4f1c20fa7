"""ctypes binding of libkvhip.so — the C ABI declared in include/kvhip.h.

There is no CPU fallback: if the HIP extension is not built, importing the ops raises.
"""
import ctypes
import os
import subprocess

# torch first: its wheel bundles libamdhip64.so (SONAME libamdhip64.so.7) and loads it by file
# name.  libkvhip.so needs "libamdhip64.so.7"; loaded after torch it binds to torch's copy.
# Loaded before torch, the system copy would come in as a SECOND HIP/HSA runtime in the
# process, and the later one sees no device.
import torch  # noqa: F401  (device memory, streams; also pins the HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.path.join(CSRC, "libkvhip.so")

KV_OK = 0
KV_INVALID_ARGUMENT = 3
KV_RESOURCE_EXHAUSTED = 8
KV_FAILED_PRECONDITION = 9
KV_UNIMPLEMENTED = 12
KV_INTERNAL = 13

KV_DT_FLOAT = 1
KV_DT_INT32 = 3
KV_DT_INT64 = 9
KV_DT_UINT64 = 23
KV_COMBINER_SUM, KV_COMBINER_MEAN, KV_COMBINER_SQRTN = 0, 1, 2

# every symbol include/kvhip.h declares (tests/test_abi.py checks the header against this)
_c = ctypes
_vp, _i64, _i32, _f = _c.c_void_p, _c.c_int64, _c.c_int, _c.c_float
SIGNATURES = {
    "kv_last_error": (_c.c_char_p, []),
    "kv_create": (_i32, [_i32, _i32, _i32, _i32, _i64, _i32, _c.POINTER(_vp)]),
    "kv_destroy": (_i32, [_vp]),
    "kv_reserve": (_i32, [_vp, _i64]),
    "kv_init_table": (_i32, [_vp, _vp, _i64, _vp]),
    "kv_is_initialized": (_i32, [_vp, _c.POINTER(_i32)]),
    "kv_set_clock_days": (_i32, [_vp, _i32]),
    "kv_set_seed": (_i32, [_vp, _c.c_uint64]),
    "kv_size": (_i32, [_vp, _c.POINTER(_i64), _vp]),
    "kv_map_size": (_i32, [_vp, _c.POINTER(_i64), _vp]),
    "kv_sum_freq": (_i32, [_vp, _c.POINTER(_i64), _vp]),
    "kv_get_meta": (_i32, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "kv_gather_or_insert": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "kv_gather_or_insert_tok": (_i32, [_vp, _vp, _vp, _i64, _vp, _c.POINTER(_c.c_uint64), _vp]),
    "kv_gather_or_zeros": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "kv_apply_group_adam": (_i32, [_vp, _vp, _vp, _vp, _i64] + [_f] * 9 + [_i32, _vp]),
    "kv_apply_adagrad": (_i32, [_vp, _vp, _f, _vp, _vp, _i64, _i32, _vp]),
    "kv_apply_sparse_group_ftrl": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64] + [_f] * 6 + [_vp]),
    "kv_apply_group_adam_unique": (_i32, [_vp, _vp, _vp, _vp, _i64] + [_f] * 9 + [_i32, _vp]),
    "kv_apply_adagrad_unique": (_i32, [_vp, _vp, _f, _vp, _vp, _i64, _i32, _vp]),
    "kv_apply_sparse_group_ftrl_unique": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64] + [_f] * 6 + [_vp]),
    "kv_apply_group_adam_tok": (_i32, [_vp, _vp, _vp, _vp, _i64] + [_f] * 9 + [_i32, _c.c_uint64, _vp]),
    "kv_apply_adagrad_tok": (_i32, [_vp, _vp, _f, _vp, _vp, _i64, _i32, _c.c_uint64, _vp]),
    "kv_apply_sparse_group_ftrl_tok": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64] + [_f] * 6 + [_c.c_uint64, _vp]),
    "kv_attach_slot": (_i32, [_vp, _vp, _vp]),
    "kv_set_deterministic": (_i32, [_vp, _i32]),
    "kv_set_fast_math": (_i32, [_vp, _i32]),
    "kv_get_stat": (_i32, [_vp, _i32, _c.POINTER(_i64)]),
    "kv_prepare_capture": (_i32, [_vp, _i64, _vp]),
    "kv_dedup_segment_sum": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _c.POINTER(_i64), _vp]),
    "kv_export_count": (_i32, [_vp, _i32, _c.POINTER(_i64), _vp]),
    "kv_export_fill": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "kv_set_delta_tracking": (_i32, [_vp, _i32, _i32]),
    "kv_export_delta_count": (_i32, [_vp, _i32, _c.POINTER(_i64), _vp]),
    "kv_export_delta_fill": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "kv_import": (_i32, [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp]),
    "kv_import_delta": (_i32, [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _i32, _vp]),
    "kv_insert": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "kv_scatter_update": (_i32, [_vp, _vp, _vp, _i64, _i32, _vp]),
    "kv_unique": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _c.POINTER(_i64), _vp, _vp]),
    "kv_bucket_by_owner": (_i32, [_vp, _vp, _i64, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "kv_comm_unique_id": (_i32, [_vp]),
    "kv_comm_create": (_i32, [_i32, _i32, _vp, _i32, _c.POINTER(_vp)]),
    "kv_comm_create_staged": (_i32, [_i32, _i32, _vp, _vp, _vp, _i32, _c.POINTER(_vp)]),
    "kv_comm_destroy": (_i32, [_vp]),
    "kv_forget_stream": (_i32, [_vp]),
    "kv_comm_stream": (_i32, [_vp, _c.POINTER(_vp)]),
    "kv_comm_all_to_all": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "kv_shard_create": (_i32, [_vp, _i32, _i32, _i32, _i64, _i64, _c.POINTER(_vp)]),
    "kv_shard_destroy": (_i32, [_vp]),
    "kv_shard_buffers": (_i32, [_vp] + [_c.POINTER(_vp)] * 4 + [_c.POINTER(_i64)] * 2),
    "kv_shard_lookup_route": (_i32, [_vp, _vp, _i64, _vp]),
    "kv_shard_lookup_serve": (_i32, [_vp, _vp]),
    "kv_shard_lookup_finish": (_i32, [_vp, _vp, _vp]),
    "kv_shard_apply_route": (_i32, [_vp, _vp, _vp]),
    "kv_shard_apply_serve": (_i32, [_vp, _i32, _vp, _vp, _c.POINTER(_f), _vp]),
    "kv_shard_lookup": (_i32, [_vp, _vp, _vp, _i64, _vp, _i32, _vp]),
    "kv_shard_apply": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp, _c.POINTER(_f), _i32, _vp]),
    "kv_shard_set_lossless": (_i32, [_vp, _i32]),
    "kv_shard_profile": (_i32, [_vp, _i32]),
    "kv_shard_profile_read": (_i32, [_vp, _c.POINTER(_c.c_double), _c.POINTER(_i64), _i32, _c.POINTER(_i64)]),
    "kv_shard_agree_local": (_i32, [_c.POINTER(_vp), _i32, _c.POINTER(_i32), _vp]),
    "kv_multi_shard_lookup": (_i32, [_c.POINTER(_vp), _i32, _vp, _c.POINTER(_vp), _c.POINTER(_c.c_int64), _c.POINTER(_vp), _i32, _vp]),
    "kv_multi_shard_apply": (_i32, [_c.POINTER(_vp), _i32, _vp, _i32, _c.POINTER(_vp), _c.POINTER(_vp), _c.POINTER(_vp),
                             _c.POINTER(_f), _i32, _vp]),
    "kv_shard_join": (_i32, [_vp, _vp]),
    "kv_shard_exchange_local": (_i32, [_c.POINTER(_vp), _i32, _i32, _vp]),
    "kv_get_count": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "kv_get_timestamp": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "kv_delete": (_i32, [_vp, _vp, _i64, _c.POINTER(_i64), _vp]),
    "kv_delete_with_timestamp": (_i32, [_vp, _i32, _i32, _vp, _c.POINTER(_i64), _vp]),
    "kv_batch_gather_or_zeros": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp]),
    "kv_multi_gather_or_insert": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "kv_multi_apply_group_adam": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp] + [_c.c_float] * 9 + [_i32, _vp]),
    "kv_multi_apply_adagrad": (_i32, [_i32, _vp, _vp, _c.c_float, _vp, _vp, _vp, _i32, _vp]),
    "kv_multi_apply_sparse_group_ftrl": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp] + [_c.c_float] * 6 + [_vp]),
    "kv_multi_gather_or_insert_tok": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "kv_multi_apply_group_adam_unique": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp] + [_c.c_float] * 9 + [_i32, _vp]),
    "kv_multi_apply_adagrad_unique": (_i32, [_i32, _vp, _vp, _c.c_float, _vp, _vp, _vp, _i32, _vp]),
    "kv_multi_apply_sparse_group_ftrl_unique": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp] + [_c.c_float] * 6 + [_vp]),
    "kv_multi_apply_group_adam_tok": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp] + [_c.c_float] * 9 + [_i32, _vp, _vp]),
    "kv_multi_apply_adagrad_tok": (_i32, [_i32, _vp, _vp, _c.c_float, _vp, _vp, _vp, _i32, _vp, _vp]),
    "kv_multi_apply_sparse_group_ftrl_tok": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp] + [_c.c_float] * 6 + [_vp, _vp]),
    "kv_lookup_sparse": (_i32, [_vp, _vp, _vp, _i32, _vp, _i64, _i64, _i32, _i32, _vp, _vp]),
    "kv_unsorted_segment_sum": (_i32, [_vp, _vp, _vp, _i64, _i64, _vp, _vp]),
    "kv_take_rows": (_i32, [_i32, _vp, _vp, _vp, _i64, _i64, _i32, _vp, _vp]),
    "kv_gather_or_insert_pairs": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "kv_profile_enable": (_i32, [_vp, _i32]),
    "kv_profile_select": (_i32, [_vp, _c.c_uint32]),
    "kv_profile_sample": (_i32, [_vp, _i32]),
    "kv_profile_read": (_i32, [_vp, _c.POINTER(_c.c_double), _c.POINTER(_i64), _i32]),
}


class KvError(RuntimeError):
  """Base of the status errors the C ABI reports (tensorflow error-code numbering)."""
  code = KV_INTERNAL


class InvalidArgumentError(KvError, ValueError):
  code = KV_INVALID_ARGUMENT


class FailedPreconditionError(KvError):
  code = KV_FAILED_PRECONDITION


class ResourceExhaustedError(KvError, MemoryError):
  code = KV_RESOURCE_EXHAUSTED


class UnimplementedError(KvError, NotImplementedError):
  code = KV_UNIMPLEMENTED


_BY_CODE = {c.code: c for c in (InvalidArgumentError, FailedPreconditionError,
                                ResourceExhaustedError, UnimplementedError)}


def build(force=False):
  """hipcc build of the extension, in-tree (tfplus_amd/csrc/libkvhip.so)."""
  srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
  srcs.append(os.path.join(_HERE, "..", "include", "kvhip.h"))
  stale = (not os.path.exists(SO_PATH)
           or os.path.getmtime(SO_PATH) < max(os.path.getmtime(f) for f in srcs))
  if force or stale:
    subprocess.check_call(["make", "-C", CSRC, "-s", "-j6"] + (["-B"] if force else []))
  return SO_PATH


_lib = None


def lib():
  global _lib
  if _lib is None:
    if not os.path.exists(SO_PATH):
      raise ImportError(
          "tfplus_amd: %s is missing — build it with `python -c 'import __graft_entry__ as g; "
          "g.build()'` or `make -C tfplus_amd/csrc`.  There is no CPU fallback." % SO_PATH)
    L = ctypes.CDLL(SO_PATH)
    for name, (res, args) in SIGNATURES.items():
      fn = getattr(L, name)
      fn.restype = res
      fn.argtypes = args
    _lib = L
  return _lib


def check(rc):
  if rc != KV_OK:
    msg = lib().kv_last_error().decode("utf-8", "replace")
    raise _BY_CODE.get(rc, KvError)(msg)
