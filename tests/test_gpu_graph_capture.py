"""The read-only serving op (KvVariableGatherOrZerosV2) inside a HIP graph: one kernel, no host-side
table bookkeeping, so a captured launch can be replayed on new ids written into the same buffer.
(The training ops are not capturable: capacity accounting and growth run on the host, DESIGN.md.)"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")


@pytest.mark.gpu
@pytest.mark.parametrize("D", [8, 32, 100])
def test_gather_or_zeros_replays_in_a_graph(D):
  if not torch.cuda.is_available():
    pytest.skip("needs a GPU")
  from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as ops
  dev = torch.device("cuda", 0)
  rng = np.random.default_rng(5)
  h = ops.kv_variable([D])
  ops.init_kv_variable_v2(h, np.zeros((4, D), np.float32))
  keys = torch.arange(0, 4000, 2, dtype=torch.int64, device=dev)          # even keys only
  vals = torch.from_numpy(rng.standard_normal((keys.numel(), D)).astype(np.float32)).to(dev)
  ops.kv_variable_insert_v2(h, keys, vals)

  N = 5000
  ids = torch.zeros(N, dtype=torch.int64, device=dev)
  side = torch.cuda.Stream()
  side.wait_stream(torch.cuda.current_stream())
  with torch.cuda.stream(side):                                            # warm-up outside the capture
    ops.kv_variable_gather_or_zeros_v2(h, ids)
  torch.cuda.current_stream().wait_stream(side)
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g):
    out = ops.kv_variable_gather_or_zeros_v2(h, ids)
  for rep in range(3):
    new = torch.from_numpy(rng.integers(-100, 4100, N)).to(dev)
    ids.copy_(new)
    g.replay()
    torch.cuda.synchronize()
    want = torch.zeros(N, D, device=dev)
    hit = (new >= 0) & (new < 4000) & (new % 2 == 0)
    want[hit] = vals[(new[hit] // 2)]
    assert torch.equal(out, want)
    assert torch.equal(out, ops.kv_variable_gather_or_zeros_v2(h, new))
  # rows updated between replays are seen by the next replay (the graph holds no copy of the table)
  vals2 = vals + 1.0
  ops.kv_variable_insert_v2(h, keys, vals2)
  g.replay()
  torch.cuda.synchronize()
  want = torch.zeros(N, D, device=dev)
  want[hit] = vals2[(new[hit] // 2)]
  assert torch.equal(out, want)
