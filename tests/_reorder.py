"""Per-element bound for a GroupAdam (l1 = l2 = l21 = 0) row update whose gradient sum was taken in ANOTHER fp32
order than the checker's: instead of a blanket rtol, the largest change of the updated element when the summed
gradient moves anywhere inside [gsum - dg, gsum + dg], dg = (cnt - 1) 2^-24 sum|g| (every fp32 order of the same
addends stays inside that interval around the exact sum).  Shared by the tests that compare two summation orders."""
import numpy as np


def adam_step(x, m, v, z, g, hp):
  """KvVariableGroupSparseApplyAdamV4 with l1 = l2 = l21 = 0 (training_ops.cc:7166-7195), float64: the linear slot z
  carries alpha * m - d; with beta1_power following the steps this telescopes to Adam, with a constant one (what some
  tests pass) it does not — so the slot is part of the state here."""
  mn = hp["b1"] * m + (1 - hp["b1"]) * g
  vn = hp["b2"] * v + (1 - hp["b2"]) * g * g
  s = np.sqrt(vn)
  d = (s - np.sqrt(v)) * x if hp["b1"] > hp["b1p"] else (s + hp["eps"]) * x
  zn = z + (hp["alpha"] * mn - d)
  return -zn / (s + hp["eps"])


def adam_eval_err(x, m, v, z, g, hp):
  """What fp32 evaluation of adam_step can be off by: x_new = -z_new / (s + eps) and z_new is a sum of terms that
  cancel (z, alpha m_new, d), each rounded to fp32 — a few ulps of the LARGEST of them, divided by s + eps, which
  for a small row can be far more than an ulp of x_new itself."""
  mn = hp["b1"] * m + (1 - hp["b1"]) * g
  vn = hp["b2"] * v + (1 - hp["b2"]) * g * g
  s = np.sqrt(vn)
  d = (s - np.sqrt(v)) * x if hp["b1"] > hp["b1p"] else (s + hp["eps"]) * x
  # ... and m_new itself is a sum of two terms of opposite sign when the gradient turns against the momentum
  mterms = hp["alpha"] * (np.abs(hp["b1"] * m) + np.abs((1 - hp["b1"]) * g))
  return 2.0 ** -22 * (np.abs(z) + mterms + np.abs(d)) / (s + hp["eps"])


def adam_hp(lr, b1p, b2p, b1=0.9, b2=0.999, eps=1e-8):
  b1p, b2p = float(np.float32(b1p)), float(np.float32(b2p))
  return {"alpha": lr * np.sqrt(1.0 - b2p) / (1.0 - b1p), "b1": float(np.float32(b1)), "b2": float(np.float32(b2)),
          "eps": eps, "b1p": b1p}


def adam_reorder_check(x0, m0, v0, z0, ids, grads, x1, hp, what=""):
  """x0 / m0 / v0 / z0: rows of the distinct ids (np.unique order) BEFORE the step, x1: after.  Raises when an element of
  x1 is farther from the float64 update than the reorder bound + 1e-6 relative."""
  u, inv, cnt = np.unique(ids, return_inverse=True, return_counts=True)
  D = grads.shape[1]
  g64 = grads.astype(np.float64)
  gsum = np.zeros((u.size, D)); np.add.at(gsum, inv, g64)
  gabs = np.zeros((u.size, D)); np.add.at(gabs, inv, np.abs(g64))
  dg = (cnt[:, None] - 1).clip(min=0) * 2.0 ** -24 * gabs
  x0 = x0.astype(np.float64); m0 = m0.astype(np.float64); v0 = v0.astype(np.float64); z0 = z0.astype(np.float64)
  f0 = adam_step(x0, m0, v0, z0, gsum, hp)
  dev = np.maximum(np.abs(adam_step(x0, m0, v0, z0, gsum - dg, hp) - f0), np.abs(adam_step(x0, m0, v0, z0, gsum + dg, hp) - f0))
  inside = np.abs(gsum) <= dg
  dev = np.where(inside, np.maximum(dev, np.abs(adam_step(x0, m0, v0, z0, np.zeros_like(gsum), hp) - f0)), dev)
  # the fp32 evaluation of the update itself: a few ulps of |x| and of the step
  bound = 2.0 * dev + 2.0 ** -19 * (np.abs(x0) + np.abs(f0 - x0)) + 1e-6 * np.abs(f0) + 1e-9 + adam_eval_err(x0, m0, v0, z0, gsum, hp)
  bad = np.argwhere(np.abs(x1.astype(np.float64) - f0) > bound)
  if bad.size:
    i, e = bad[0]
    raise AssertionError("%s key %d elem %d cnt %d: got %.9g exp %.9g bound %.3g dg %.3g gsum %.9g x0 %.9g m0 %.9g v0 %.9g z0 %.9g" % (
        what, u[i], e, cnt[i], x1[i, e], f0[i, e], bound[i, e], dg[i, e], gsum[i, e], x0[i, e], m0[i, e], v0[i, e], z0[i, e]))
  return u
