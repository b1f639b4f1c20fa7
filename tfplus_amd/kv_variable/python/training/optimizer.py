"""The slice of TF-core's tf.compat.v1.train.Optimizer that the KvVariable optimizers rely on
(tensorflow-cpu 2.13, not under the reference tree): compute_gradients via autograd,
apply_gradients -> _create_slots / _prepare / per-variable sparse apply / _finish, and slot
creation through get_kv_variable (the reference patches slot_creator for that,
variable_scope.py:1027-1088: slot dim = dim * num_concat_opt_vars, trainable=False).

TF-core's _resource_apply_sparse_duplicate_indices first de-duplicates the IndexedSlices
(tf.unique + unsorted_segment_sum) and then calls _resource_apply_sparse; here the raw slices go
to the op, which does that step on the GPU (include/kvhip.h, "sparse optimizer apply").
"""
import torch

from tfplus_amd.kv_variable.python.ops import variable_scope
from tfplus_amd.kv_variable.python.ops.kv_variable_ops import IndexedSlices, KvVariable
from tfplus_amd.kv_variable.python.ops.variable_scope import PartitionedKvVariable


class Optimizer(object):

  def __init__(self, use_locking, name):
    if not name:
      raise ValueError("Must specify the optimizer name")
    self._use_locking = use_locking
    self._name = name
    self._slots = {}

  def get_name(self):
    return self._name

  # -- slots ---------------------------------------------------------------------------------------
  def _slot_dict(self, slot_name):
    return self._slots.setdefault(slot_name, {})

  def get_slot(self, var, name):
    ent = self._slots.get(name, {}).get(id(var))
    return ent[1] if ent is not None and ent[0] is var else None

  def get_slot_names(self):
    return sorted(self._slots)

  def _get_or_make_slot_with_value(self, var, value, slot_name, op_name):
    d = self._slot_dict(slot_name)
    # keyed by id(var) but the entry holds the variable itself: the id of a collected variable can never be taken
    # over by a new one, and a stale entry is never mistaken for the new variable's slot
    if id(var) not in d or d[id(var)][0] is not var:
      dim = var.embedding_dim * int(var.num_concat_opt_vars)
      d[id(var)] = (var, variable_scope.get_kv_variable(
          "%s/%s" % (var.name, op_name), embedding_dim=dim,
          initializer=variable_scope.constant_initializer(value), key_dtype=var.key_dtype,
          value_dtype=var.dtype, trainable=False, device=var.device.index))
    return d[id(var)][1]

  def _zeros_slot(self, var, slot_name, op_name):
    return self._get_or_make_slot_with_value(var, 0.0, slot_name, op_name)

  # -- the Optimizer protocol -----------------------------------------------------------------------
  def _create_slots(self, var_list):
    pass

  def _prepare(self):
    pass

  def _finish(self):
    pass

  def _resource_apply_sparse(self, grad, var, indices):
    raise NotImplementedError()

  @staticmethod
  def _flatten(var_list):
    out = []
    for v in var_list:
      out.extend(list(v) if isinstance(v, (PartitionedKvVariable, list, tuple)) else [v])
    return out

  def compute_gradients(self, loss, var_list=None):
    """Back-propagates `loss`; the gradient of every lookup arrives as IndexedSlices."""
    if var_list is None:
      raise ValueError("var_list is required (there is no global trainable-variables collection)")
    vs = self._flatten(var_list)
    for v in vs:
      v.pop_gradients()
    loss.backward()
    return [(v.pop_gradients(), v) for v in vs]

  def apply_gradients(self, grads_and_vars, global_step=None, name=None):
    grads_and_vars = [(g, v) for g, v in grads_and_vars if g is not None]
    if not grads_and_vars:
      raise ValueError("No variables provided.")
    for g, v in grads_and_vars:
      if not isinstance(v, KvVariable):
        raise TypeError("only KvVariables are handled here, got %r" % (v,))
      if not isinstance(g, IndexedSlices):
        raise TypeError("Gradient must be IndexedSlices for a KvVariable: %r" % (g,))
    self._create_slots([v for _, v in grads_and_vars])
    self._prepare()
    for g, v in grads_and_vars:
      self._resource_apply_sparse(g.values, v, g.indices)
    self._finish()

  def minimize(self, loss, var_list=None, global_step=None):
    self.apply_gradients(self.compute_gradients(loss, var_list), global_step)
