"""Calibration: what the box's HBM actually sustains for write-only, read-only and copy streams."""
import torch, time
dev = torch.device("cuda", 0)
def T(fn, n=20):
  fn(); torch.cuda.synchronize()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(n): fn()
  e.record(); torch.cuda.synchronize()
  return s.elapsed_time(e) / n * 1e-3
for mb in (128, 512, 2048):
  n = mb * 1024 * 1024 // 4
  a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
  t = T(lambda: a.fill_(1.0)); print("fill  %5d MB: %6.1f us  %.2f TB/s write" % (mb, t * 1e6, n * 4 / t / 1e12))
  t = T(lambda: b.copy_(a)); print("copy  %5d MB: %6.1f us  %.2f TB/s (r+w)" % (mb, t * 1e6, 2 * n * 4 / t / 1e12))
  t = T(lambda: a.sum()); print("sum   %5d MB: %6.1f us  %.2f TB/s read" % (mb, t * 1e6, n * 4 / t / 1e12))
