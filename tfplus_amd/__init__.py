"""tfplus_amd — MI355X-native KvVariable hot path behind the tfplus op / Python surface.

Mirrors tfplus/__init__.py:22-29: `get_kv_variable`, `embedding_lookup[_sparse]`, the optimizers.
The ops need tfplus_amd/csrc/libkvhip.so (HIP, gfx950); they are imported lazily so that the
package itself imports on a machine without a GPU (the CPU test tier), but the first op call
without the extension raises — there is no CPU path.
"""

__all__ = ["get_kv_variable", "embedding_lookup", "embedding_lookup_sparse", "safe_embedding_lookup_sparse",
           "KvVariable", "GroupAdamOptimizer", "AdagradOptimizer", "SparseGroupFtrlOptimizer"]


def __getattr__(name):
  if name in ("get_kv_variable",):
    from tfplus_amd.kv_variable.python.ops import variable_scope as m
  elif name in ("embedding_lookup", "embedding_lookup_sparse", "safe_embedding_lookup_sparse"):
    from tfplus_amd.kv_variable.python.ops import embedding_ops as m
  elif name == "KvVariable":
    from tfplus_amd.kv_variable.python.ops import kv_variable_ops as m
  elif name in ("GroupAdamOptimizer", "AdagradOptimizer", "SparseGroupFtrlOptimizer"):
    from tfplus_amd.kv_variable.python import training as m
  else:
    raise AttributeError(name)
  return getattr(m, name)
