"""GroupAdamOptimizer — tfplus/kv_variable/python/training/group_adam.py:28-272.

Adam with group lasso, on KvVariables only.  One slot table `m_v_linear` of dim 3*D (group_adam.py:136-152);
version 4 (default) calls KvVariableGroupSparseApplyAdamV4, every other version KvVariableGroupSparseApplyAdamV3
(:199-232: with default kv_options that is where versions 1 and 2 end up in the reference too).  beta1_power / beta2_power start at beta and are
multiplied after the apply like TF-core's AdamOptimizer._finish, in float32.
"""
import numpy as np

from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops
from tfplus_amd.kv_variable.python.training.optimizer import Optimizer


class GroupAdamOptimizer(Optimizer):

  def __init__(self, learning_rate, initial_accumulator_value=0.0, beta1=0.9, beta2=0.999, epsilon=1e-8,
               l1_regularization_strength=0.0, l2_regularization_strength=0.0,
               l21_regularization_strength=0.0, use_locking=False, name="GroupAdam", accum_name=None,
               linear_name=None, version=4):
    super(GroupAdamOptimizer, self).__init__(use_locking, name)
    if initial_accumulator_value < 0.0:
      raise ValueError("initial_accumulator_value %f needs to be be positive or zero" % initial_accumulator_value)
    if l1_regularization_strength < 0.0:
      raise ValueError("l1_regularization_strength %f needs to be positive or zero" % l1_regularization_strength)
    if l2_regularization_strength < 0.0:
      raise ValueError("l2_regularization_strength %f needs to be positive or zero" % l2_regularization_strength)
    if l21_regularization_strength < 0.0:
      raise ValueError("l21_regularization_strength %f needs to be positive or zero" % l21_regularization_strength)
    # Versions <= 2 keep separate m / v / linear (/ accum) slot tables only for a KvVariable with non-default
    # kv_options (group_adam.py:141-170, 233-272); with the default options — the only ones this mirror has, there
    # is no SSD tier here — every version below 4 takes the fused m_v_linear table and the V3 op (:192-232).
    self._lr, self._beta1, self._beta2, self._epsilon = learning_rate, beta1, beta2, epsilon
    self._l1, self._l2, self._l21 = l1_regularization_strength, l2_regularization_strength, l21_regularization_strength
    self._linear_name = linear_name
    self._version = version
    self._beta1_power = self._beta2_power = None

  def _create_slots(self, var_list):
    if self._beta1_power is None:                       # _create_non_slot_variable(initial_value=beta)
      self._beta1_power = np.float32(self._beta1)
      self._beta2_power = np.float32(self._beta2)
    for v in var_list:
      v.num_concat_opt_vars = 3                         # group_adam.py:143
      self._zeros_slot(v, "m_v_linear", self._linear_name or (self._name + "_3"))

  def _get_beta_accumulators(self):
    return self._beta1_power, self._beta2_power

  def _resource_apply_sparse(self, grad, var, indices):
    slot = self.get_slot(var, "m_v_linear")
    fn = (gen_kv_variable_ops.kv_variable_group_sparse_apply_adam_v4 if self._version == 4 else
          gen_kv_variable_ops.kv_variable_group_sparse_apply_adam_v3)
    return fn(var.handle, slot.handle, grad, indices, self._lr, self._beta1_power, self._beta2_power,
              self._beta1, self._beta2, self._epsilon, self._l1, self._l2, self._l21, use_locking=False)

  def _finish(self):
    self._beta1_power = np.float32(self._beta1_power * np.float32(self._beta1))
    self._beta2_power = np.float32(self._beta2_power * np.float32(self._beta2))
