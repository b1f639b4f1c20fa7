"""GradientDescentOptimizer — tfplus/kv_variable/python/training/gradient_descent.py:24-33 over
tf.compat.v1.train.GradientDescentOptimizer: the sparse update is one
scatter_add(var, indices, -grad * learning_rate) on the raw (duplicate) indices."""
from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops
from tfplus_amd.kv_variable.python.training.optimizer import Optimizer


class GradientDescentOptimizer(Optimizer):

  def __init__(self, learning_rate, use_locking=False, name="GradientDescent"):
    super(GradientDescentOptimizer, self).__init__(use_locking, name)
    self._learning_rate = learning_rate

  def _resource_apply_sparse(self, grad, var, indices):
    # _resource_apply_sparse_duplicate_indices in the reference: no de-duplication, the scatter adds
    # every occurrence
    return gen_kv_variable_ops.kv_variable_scatter_add_v2(var.handle, indices, -grad * self._learning_rate)
