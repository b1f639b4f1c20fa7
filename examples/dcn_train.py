"""BASELINE.json configs[2]: the reference's example/dcn model shape (example/dcn/train.py:45-330)
with every sparse feature in a GPU KvVariable, fp32, synthetic Criteo-shaped data.

26 categorical features (hash buckets and embedding dims of train.py:45-101: eighteen tables of
dim 64, eight of dim 128), 13 continuous ones, a 2-layer cross network and a 1024-512-256 MLP.
The sparse side goes through the batched ops — one kv_multi_gather_or_insert and one
kv_multi_apply_group_adam per embedding dim — so a step issues 2 x (2 + 2) sparse launches
instead of 26 x 5; the dense tower is plain torch (it is not part of the rebuilt hot path).

  python examples/dcn_train.py --steps 200 --batch_size 2048
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as ops  # noqa: E402

HASH_BUCKET_SIZES = [2500, 2000, 300000, 250000, 1000, 100, 20000, 4000, 20, 100000, 10000, 250000, 40000, 100,
                     100, 200000, 50, 10000, 4000, 20, 250000, 100, 100, 250000, 400, 100000]
EMBEDDING_DIMENSIONS = [64, 64, 128, 128, 64, 64, 64, 64, 64, 128, 64, 128, 64, 64, 64, 128, 64, 64, 64, 64, 128, 64,
                        64, 128, 64, 128]


class SparseFeatures(object):
  """The 26 embedding tables, grouped by dim so each group is one batched launch."""

  def __init__(self, device, lr, seed=2021):
    g = torch.Generator(device=device).manual_seed(seed)
    self.groups = {}
    for i, (buckets, dim) in enumerate(zip(HASH_BUCKET_SIZES, EMBEDDING_DIMENSIONS)):
      var = ops.kv_variable([dim], capacity_hint=buckets, shared_name="embedding_weight_%d" % (i + 1))
      # keras RandomNormal(-1, 1) init table of 10000 rows (variable_scope.py:229-231)
      ops.init_kv_variable_v2(var, torch.randn(10000, dim, device=device, generator=g) - 1.0)
      slot = ops.kv_variable([3 * dim], capacity_hint=buckets)
      ops.init_kv_variable_v2(slot, torch.zeros(16, 3 * dim, device=device))
      self.groups.setdefault(dim, []).append((i, var, slot))
    self.lr = lr
    self.b1p, self.b2p = np.float32(0.9), np.float32(0.999)

  def lookup(self, cat_ids):
    """cat_ids [B, 26] int64 -> list of 26 leaf tensors [B, dim] that collect their gradients."""
    outs = [None] * 26
    self._cols = (cat_ids, {})   # the id columns as handed to the lookup: the apply passes the SAME tensors, so the
    for dim, members in self.groups.items():   # batched optimizer op takes over the lookup's index (batch tokens)
      cols = [cat_ids[:, m[0]].contiguous() for m in members]
      self._cols[1][dim] = cols
      rows = ops.kv_multi_gather_or_insert([m[1] for m in members], cols)
      for m, r in zip(members, rows):
        outs[m[0]] = r.requires_grad_(True)
    return outs

  def apply(self, cat_ids, leaves):
    kept = getattr(self, "_cols", (None, {}))
    for dim, members in self.groups.items():
      cols = kept[1][dim] if kept[0] is cat_ids else [cat_ids[:, m[0]].contiguous() for m in members]
      ops.kv_multi_group_sparse_apply_adam([m[1] for m in members], [m[2] for m in members],
                                           [leaves[m[0]].grad for m in members], cols, self.lr, self.b1p, self.b2p,
                                           0.9, 0.999, 1e-8, 0.0, 0.0, 0.0, version=4)
    self.b1p, self.b2p = np.float32(self.b1p * np.float32(0.9)), np.float32(self.b2p * np.float32(0.999))


class DenseTower(torch.nn.Module):
  """Cross network (train.py:178-203) + MLP (:157-176) + the summed logits (:330-345)."""

  def __init__(self, width, hidden=(1024, 512, 256), cross_layers=2):   # _cross_net(layer_num=2)
    super().__init__()
    self.cross_w = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(width) * 0.01) for _ in range(cross_layers)])
    self.cross_b = torch.nn.ParameterList([torch.nn.Parameter(torch.zeros(width)) for _ in range(cross_layers)])
    layers, d = [], width
    for h in hidden:
      layers += [torch.nn.Linear(d, h), torch.nn.ReLU()]
      d = h
    self.mlp = torch.nn.Sequential(*layers)
    self.dnn_logit = torch.nn.Linear(d, 1)
    self.cross_logit = torch.nn.Linear(width, 1)

  def forward(self, x0):
    x = x0
    for w, b in zip(self.cross_w, self.cross_b):
      x = x0 * (x * w).sum(1, keepdim=True) + b + x
    return (self.dnn_logit(self.mlp(x0)) + self.cross_logit(x)).squeeze(1)


def synthetic_batch(gen, batch, device):
  """Criteo-shaped: Zipf-ish categorical ids inside each feature's hash range, 13 floats, a label
  that depends on a few features so the loss can fall."""
  u = torch.rand(batch, 26, device=device, generator=gen)
  buckets = torch.tensor(HASH_BUCKET_SIZES, device=device, dtype=torch.float32)
  cat = (buckets * u ** 3).to(torch.int64)                     # skewed towards small ids
  cont = torch.rand(batch, 13, device=device, generator=gen)
  logit = (cat[:, 5] % 2).float() * 2 - 1 + (cont[:, 0] - 0.5) * 2 + ((cat[:, 8] % 3) == 0).float()
  label = (torch.rand(batch, device=device, generator=gen) < torch.sigmoid(logit)).float()
  return cat, cont, label


def main(argv=None):
  ap = argparse.ArgumentParser()
  ap.add_argument("--steps", type=int, default=200)
  ap.add_argument("--batch_size", type=int, default=2048)       # train.py:740
  ap.add_argument("--learning_rate", type=float, default=0.01)  # train.py:777 (sparse tables)
  ap.add_argument("--dense_learning_rate", type=float, default=0.001)
  ap.add_argument("--seed", type=int, default=2021)
  args = ap.parse_args(argv)
  dev = torch.device("cuda", 0)
  torch.manual_seed(args.seed)
  gen = torch.Generator(device=dev).manual_seed(args.seed)
  sparse = SparseFeatures(dev, args.learning_rate, args.seed)
  tower = DenseTower(sum(EMBEDDING_DIMENSIONS) + 13).to(dev)
  dense_opt = torch.optim.Adam(tower.parameters(), lr=args.dense_learning_rate)
  losses, t_sparse = [], 0.0
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  t_half = t0
  for step in range(args.steps):
    if step == args.steps // 2:     # the second half is the steady state (tables sized, GEMM kernels chosen)
      torch.cuda.synchronize()
      t_half = time.perf_counter()
    cat, cont, label = synthetic_batch(gen, args.batch_size, dev)
    leaves = sparse.lookup(cat)
    logits = tower(torch.cat(leaves + [cont], 1))
    loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, label)
    dense_opt.zero_grad(set_to_none=True)
    loss.backward()
    dense_opt.step()
    sparse.apply(cat, leaves)
    if step % 20 == 0 or step == args.steps - 1:
      losses.append(float(loss.detach()))
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  keys = sum(int(ops.kv_variable_shape_v2(m[1])[0]) for ms in sparse.groups.values() for m in ms)
  n2 = args.steps - args.steps // 2
  dt2 = time.perf_counter() - t_half
  print("steps %d  batch %d  loss %.4f -> %.4f  %.2f ms/step over all steps, %.2f ms/step over the last %d  "
        "(%.0f examples/s)  %d keys in 26 tables" % (
            args.steps, args.batch_size, losses[0], losses[-1], dt / args.steps * 1e3, dt2 / n2 * 1e3, n2,
            n2 * args.batch_size / dt2, keys))
  return losses


if __name__ == "__main__":
  main()
