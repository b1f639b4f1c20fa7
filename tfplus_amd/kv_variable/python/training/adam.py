"""AdamOptimizer — tfplus/kv_variable/python/training/adam.py:32-163 over
tf.compat.v1.train.AdamOptimizer.  For KvVariables the moments live in ONE slot table "m_v" of
dim 2·D (num_concat_opt_vars = 2, adam.py:84-87) and the sparse step is built from the generic
ops (adam.py:93-155): gather(m_v) -> m, v update -> scatter_update(m_v) -> scatter_sub(var).
TF-core de-duplicates the IndexedSlices first (unique + unsorted_segment_sum), done here by
kv_dedup_segment_sum.  beta powers are fp32 non-slot variables multiplied after the apply
(_finish), exactly as in GroupAdamOptimizer."""
import numpy as np
import torch

from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops
from tfplus_amd.kv_variable.python.training.optimizer import Optimizer


class AdamOptimizer(Optimizer):

  def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8, use_locking=False,
               name="Adam", version=2):
    super(AdamOptimizer, self).__init__(use_locking, name)
    if version != 2:
      raise ValueError("Unknown version")          # separate m / v slots (version <= 1) are not carried over
    self._lr, self._beta1, self._beta2, self._epsilon = learning_rate, beta1, beta2, epsilon
    self._beta1_power = self._beta2_power = None

  def _get_beta_accumulators(self):
    return self._beta1_power, self._beta2_power

  def _create_slots(self, var_list):
    if self._beta1_power is None:
      self._beta1_power, self._beta2_power = np.float32(self._beta1), np.float32(self._beta2)
    for v in var_list:
      v.num_concat_opt_vars = 2
      self._zeros_slot(v, "m_v", self._name)

  def _resource_apply_sparse(self, grad, var, indices):
    D = var.embedding_dim
    ids, g, _ = gen_kv_variable_ops.kv_dedup_segment_sum(var.handle, indices, grad.reshape(-1, D))
    m_v = self.get_slot(var, "m_v")
    mv = gen_kv_variable_ops.kv_variable_gather_or_insert_v2(m_v.handle, ids)     # array_ops.gather on the slot
    b1, b2 = np.float32(self._beta1), np.float32(self._beta2)
    m = float(b1) * mv[:, :D] + g * float(np.float32(1) - b1)
    v = float(b2) * mv[:, D:] + (g * g) * float(np.float32(1) - b2)
    gen_kv_variable_ops.kv_variable_scatter_update_v2(m_v.handle, ids, torch.cat([m, v], 1))
    lr = np.float32(self._lr) * np.sqrt(np.float32(1) - self._beta2_power) / (np.float32(1) - self._beta1_power)
    upd = float(lr) * m / (float(np.float32(self._epsilon)) + torch.sqrt(v))
    return gen_kv_variable_ops.kv_variable_scatter_sub_v2(var.handle, ids, upd)

  def _finish(self):
    self._beta1_power = np.float32(self._beta1_power * np.float32(self._beta1))
    self._beta2_power = np.float32(self._beta2_power * np.float32(self._beta2))
