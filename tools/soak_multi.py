"""Randomised soak of the batched multi-table ops (kv_multi_gather_or_insert, kv_multi_*_apply_*) against the
single-table ops: python tools/soak_multi.py FIRST LAST (on the GPU box).  Per seed: a random number of tables, dim,
optimizer, batch sizes (empty tables, single ids, batches that span many tiles), id span (heavy repeats .. mostly
distinct) and sign pattern; 3 steps of lookup + apply, deterministic mode — both paths run the same kernels, so rows
returned, keys, values, frequencies and sizes must be bit-identical."""
import os, sys, traceback
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as ops
DAY = 20000


def dump(h):
  k, v = ops.read_kv_variable_op_v2(h)
  o = torch.argsort(k)
  return k[o], v[o], ops.kv_variable_frequency(h), ops.kv_variable_size_v2(h)


def run(seed):
  rng = np.random.default_rng(seed)
  T = int(rng.integers(2, 13)); D = int(rng.choice([4, 8, 16, 32, 64])); opt = str(rng.choice(["adam", "adagrad", "ftrl"]))
  span = int(rng.choice([30, 500, 20000, 10 ** 9]))
  sizes = [int(rng.choice([0, 1, 7, 300, 2048, 2049, 5000, 9000, 40000], p=[.05, .05, .1, .2, .2, .1, .15, .1, .05])) for _ in range(T)]
  slot_vals = {"adam": [None], "adagrad": [0.1], "ftrl": [0.1, 0.0]}[opt]

  def make():
    out = []
    for j in range(T):
      hs = []
      for si, val in enumerate([None] + slot_vals):
        dim = 3 * D if (opt == "adam" and si == 1) else D
        h = ops.kv_variable([dim])
        ops.kv_set_clock_days(h, DAY); ops.kv_set_seed(h, 11 + j)
        if si == 0:
          tab = np.random.default_rng(seed * 100 + j).standard_normal((32, D)).astype(np.float32)
        else:
          tab = np.full((4, dim), 0.0 if val is None else val, np.float32)
        ops.init_kv_variable_v2(h, tab)
        ops.kv_set_deterministic(h, True)
        hs.append(h)
      out.append(hs)
    return out
  A, B = make(), make()
  b1p, b2p = np.float32(0.9), np.float32(0.999)
  for step in range(3):
    ids = [rng.integers(-span, span, n) for n in sizes]
    grads = [(rng.standard_normal((i.size, D)) * 1e-2).astype(np.float32) for i in ids]
    if seed % 2 == 0:   # device tensors: the batched lookup's tokens reach the batched optimizer op (no second index pass)
      ids = [torch.from_numpy(i).cuda() for i in ids]
    outs = ops.kv_multi_gather_or_insert([a[0] for a in A], ids)
    for j in range(T):
      want = ops.kv_variable_gather_or_insert_v2(B[j][0], ids[j])
      assert torch.equal(outs[j], want), ("lookup", step, j)
    if opt == "adam":
      ops.kv_multi_group_sparse_apply_adam([a[0] for a in A], [a[1] for a in A], grads, ids, 1e-2, b1p, b2p, 0.9, 0.999, 1e-8, 0, 0, 0)
      for b, g, i in zip(B, grads, ids):
        ops.kv_variable_group_sparse_apply_adam_v4(b[0], b[1], g, i, 1e-2, b1p, b2p, 0.9, 0.999, 1e-8, 0, 0, 0)
      b1p, b2p = np.float32(b1p * np.float32(0.9)), np.float32(b2p * np.float32(0.999))
    elif opt == "adagrad":
      ops.kv_multi_sparse_apply_adagrad([a[0] for a in A], [a[1] for a in A], 0.05, grads, ids)
      for b, g, i in zip(B, grads, ids):
        ops.kv_variable_sparse_apply_adagrad(b[0], b[1], 0.05, g, i, use_locking=True)
    else:
      ops.kv_multi_sparse_group_sparse_apply_ftrl([a[0] for a in A], [a[1] for a in A], [a[2] for a in A], grads, ids,
                                                  0.05, 1e-3, 1e-3, 1e-4, 0.0, -0.5)
      for b, g, i in zip(B, grads, ids):
        ops.kv_variable_sparse_group_sparse_apply_ftrl_v2(b[0], b[1], b[2], g, i, 0.05, 1e-3, 1e-3, 1e-4, 0.0, -0.5)
  for j, (a, b) in enumerate(zip(A, B)):
    for ha, hb in zip(a, b):
      ka, va, fa, sa = dump(ha)
      kb, vb, fb, sb = dump(hb)
      assert torch.equal(ka, kb) and fa == fb and sa == sb, ("keys", j)
      assert torch.equal(va, vb), ("values", j)
  return "T %d dim %d %s span %d sizes %s" % (T, D, opt, span, sizes)


first, last = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, last):
  try:
    print("seed %d ok  %s" % (seed, run(seed)), flush=True)
  except Exception:
    bad += 1
    print("seed %d FAILED" % seed, flush=True)
    traceback.print_exc()
sys.exit(1 if bad else 0)
