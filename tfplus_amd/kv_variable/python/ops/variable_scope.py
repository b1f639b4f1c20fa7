"""get_kv_variable — tfplus/kv_variable/python/ops/variable_scope.py:745-777 (store :129-447).

Same call shape as the reference: `get_kv_variable(name, embedding_dim, key_dtype, value_dtype,
initializer, trainable, partitioner, enter_threshold)`.  The init table is [10000, dim] drawn from
the initializer (variable_scope.py:229-231); with a partitioner every shard gets its own
[10000, dim] slice of a [10000 * shards, dim] draw (:293-296).
"""
import numpy as np
import torch

from tfplus_amd.kv_variable.python.ops.kv_variable_ops import KvVariable

INIT_TABLE_ROWS = 10000  # variable_scope.py:229-231


# ---- initializers, TF1 names: classes whose instances map shape -> float32 tensor.  Like in TF the
# class itself may be passed (`initializer=ones_initializer`), it is instantiated with defaults.
class ones_initializer(object):
  def __call__(self, shape):
    return torch.ones(shape, dtype=torch.float32)


class zeros_initializer(object):
  def __call__(self, shape):
    return torch.zeros(shape, dtype=torch.float32)


class constant_initializer(object):
  def __init__(self, value=0.0):
    self.value = float(value)

  def __call__(self, shape):
    return torch.full(shape, self.value, dtype=torch.float32)


class random_normal_initializer(object):
  def __init__(self, mean=0.0, stddev=1.0, seed=None):
    self.mean, self.stddev, self.seed = mean, stddev, seed

  def __call__(self, shape):
    g = torch.Generator().manual_seed(0 if self.seed is None else int(self.seed))
    return torch.randn(shape, generator=g) * self.stddev + self.mean


class random_uniform_initializer(object):
  def __init__(self, minval=0.0, maxval=1.0, seed=None):
    self.minval, self.maxval, self.seed = minval, maxval, seed

  def __call__(self, shape):
    g = torch.Generator().manual_seed(0 if self.seed is None else int(self.seed))
    return torch.rand(shape, generator=g) * (self.maxval - self.minval) + self.minval


class truncated_normal_initializer(object):
  def __init__(self, mean=0.0, stddev=1.0, seed=None):
    self.mean, self.stddev, self.seed = mean, stddev, seed

  def __call__(self, shape):
    g = torch.Generator().manual_seed(0 if self.seed is None else int(self.seed))
    t = torch.empty(shape)
    torch.nn.init.trunc_normal_(t, mean=self.mean, std=self.stddev, a=self.mean - 2 * self.stddev,
                                b=self.mean + 2 * self.stddev, generator=g)
    return t


def fixed_size_partitioner(num_shards, axis=0):
  """tf.compat.v1.fixed_size_partitioner: num_shards along axis 0."""
  def part(shape=None, dtype=None):
    return [int(num_shards)] + [1] * (len(shape) - 1 if shape is not None else 1)
  part.num_shards = int(num_shards)
  return part


class PartitionedKvVariable(list):
  """What the reference gets back from tf's PartitionedVariable: iterable over the shards."""

  def __init__(self, name, shards):
    super().__init__(shards)
    self.name = name

  @property
  def embedding_dim(self):
    return self[0].embedding_dim


class _KvVariableStore(object):
  """variable_scope.py:119-621 reduced to what the hot path needs: a name -> variable map."""

  def __init__(self):
    self._vars = {}
    self._partitioned_vars = {}

  def get_kv_variable(self, name, shape=None, key_dtype=torch.int64, dtype=torch.float32,
                      initializer=None, reuse=None, trainable=None, partitioner=None,
                      enter_threshold=0, capacity_hint=0, device=None):
    trainable = True if trainable is None else trainable
    if shape is None and not isinstance(initializer, (torch.Tensor, np.ndarray)):
      raise ValueError("embedding_dim must be specified when the initializer is not a tensor")
    dims = [int(shape)] if np.isscalar(shape) else (list(shape) if shape is not None else None)
    if partitioner is not None:
      if not callable(partitioner):
        raise ValueError("Partitioner must be callable, but received: %s" % partitioner)
      if name in self._vars:
        raise ValueError("A partitioner was provided, but an unpartitioned version of the variable "
                         "was found: %s." % name)
      if name in self._partitioned_vars:
        if reuse is False:
          raise ValueError("Partitioned variable with name %s already exists." % name)
        return self._partitioned_vars[name]
      if reuse is True:
        raise ValueError("PartitionedVariable %s does not exist, or was not created with "
                         "get_kv_variable(). Did you mean to set reuse=None?" % name)
      shards = int(partitioner(shape=[100000] + dims, dtype=dtype)[0])
      table = self._draw(initializer, [INIT_TABLE_ROWS * shards] + dims)
      parts = []
      for i in range(shards):
        sl = table[i * INIT_TABLE_ROWS:(i + 1) * INIT_TABLE_ROWS]
        parts.append(self._single("%s/part_%d" % (name, i), sl, key_dtype, dtype, trainable,
                                  enter_threshold, capacity_hint, device, reuse=None))
      pv = PartitionedKvVariable(name, parts)
      self._partitioned_vars[name] = pv
      return pv
    if reuse is True and name in self._partitioned_vars:
      return self._partitioned_vars[name]
    if "%s/part_0" % name in self._vars:
      raise ValueError("No partitioner was provided, but a partitioned version of the variable was "
                       "found: %s/part_0." % name)
    if name in self._vars:
      if reuse is False:
        raise ValueError("Variable %s already exists, disallowed. Did you mean to set reuse=True?" % name)
      return self._vars[name]
    if reuse is True:
      raise ValueError("Variable %s does not exist, or was not created with get_kv_variable()." % name)
    if isinstance(initializer, (torch.Tensor, np.ndarray)):
      table = torch.as_tensor(initializer, dtype=torch.float32)
    else:
      table = self._draw(initializer, [INIT_TABLE_ROWS] + dims)
    return self._single(name, table, key_dtype, dtype, trainable, enter_threshold, capacity_hint, device, reuse)

  @staticmethod
  def _draw(initializer, shape):
    if initializer is None:
      # tf.get_variable's default for float variables: glorot_uniform
      limit = float(np.sqrt(6.0 / (shape[0] + shape[-1])))
      initializer = random_uniform_initializer(-limit, limit)
    if isinstance(initializer, type):   # e.g. tf.compat.v1.ones_initializer passed as a class
      initializer = initializer()
    out = initializer(list(shape))
    return torch.as_tensor(out, dtype=torch.float32)

  def _single(self, name, table, key_dtype, dtype, trainable, enter_threshold, capacity_hint, device, reuse):
    v = KvVariable(initial_value=table, name=name, embedding_dim=table.shape[1], key_dtype=key_dtype,
                   value_dtype=dtype, trainable=trainable, enter_threshold=enter_threshold,
                   capacity_hint=capacity_hint, device=device)
    self._vars[name] = v
    return v


_STORE = _KvVariableStore()


def _get_default_kv_variable_store():
  return _STORE


def reset_default_store():
  """tf.reset_default_graph() for the variable store (tests)."""
  global _STORE
  _STORE = _KvVariableStore()


def get_kv_variable(name, embedding_dim=None, key_dtype=torch.int64, value_dtype=torch.float32,
                    initializer=None, regularizer=None, trainable=None, collections=None,
                    partitioner=None, constraint=None, enter_threshold=0, kv_options=None, reuse=None,
                    capacity_hint=0, device=None):
  """variable_scope.py:745-777.  `capacity_hint` / `device` are the only additions (HBM pre-sizing)."""
  return _get_default_kv_variable_store().get_kv_variable(
      name, shape=embedding_dim, key_dtype=key_dtype, dtype=value_dtype, initializer=initializer,
      reuse=reuse, trainable=trainable, partitioner=partitioner, enter_threshold=enter_threshold,
      capacity_hint=capacity_hint, device=device)
