"""HIP graphs over the ops.  The read-only serving op (KvVariableGatherOrZerosV2) is one kernel with no host-side table
bookkeeping: a captured launch replays on new ids written into the same buffer.  The training ops keep their capacity
accounting on the host: they are capturable behind kv_prepare_capture (which settles it for a stated number of ids), and
a replay re-runs the captured launches with the captured arguments."""
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


@pytest.mark.gpu
@pytest.mark.parametrize("unique", [False, True])
def test_training_step_replays_in_a_graph(unique):
  """A captured optimizer step behind kv_prepare_capture, replayed three times, against the same three steps issued
  eagerly on a twin table.  unique: the op as an unchanged graph calls it (kv_apply_*_unique) — under capture it runs
  the batch pipeline (its duplicate guard's launch serial lives on the host), the same bits as the one-launch path."""
  if not torch.cuda.is_available():
    pytest.skip("needs a GPU")
  from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as ops
  dev = torch.device("cuda", 0)
  gen = torch.Generator(device=dev).manual_seed(3)
  D, n = 32, 30_000
  ids = torch.randperm(200_000, device=dev, generator=gen)[:n] - 500          # unique ids
  grad = torch.randn(n, D, device=dev, generator=gen) * 1e-2
  init = torch.randn(64, D, device=dev, generator=gen)
  hp = (1e-2, 0.9, 0.999, 0.9, 0.999, 1e-8, 0.0, 0.0, 0.0)

  def pair():
    v = ops.kv_variable([D], capacity_hint=4 * n)
    s = ops.kv_variable([3 * D], capacity_hint=4 * n)
    ops.kv_set_seed(v, 7)
    ops.init_kv_variable_v2(v, init)
    ops.init_kv_variable_v2(s, torch.zeros(4, 3 * D, device=dev))
    ops.kv_variable_gather_or_insert_v2(v, ids)
    # warm-up outside the capture with the PLAIN op: slot rows exist, hints are set, and the batch workspace has its size
    # (the unique form alone never builds one, and a captured call cannot: it is refused with FAILED_PRECONDITION)
    ops.kv_variable_group_sparse_apply_adam_v4(v, s, grad, ids, *hp)
    return v, s

  (gv, gs), (ev, es) = pair(), pair()
  torch.cuda.synchronize()
  if unique:   # a table that has only ever seen the one-launch form has no batch workspace: the captured call says so, cleanly
    from tfplus_amd import _lib
    cv = ops.kv_variable([D], capacity_hint=4 * n)
    cs = ops.kv_variable([3 * D], capacity_hint=4 * n)
    ops.init_kv_variable_v2(cv, init)
    ops.init_kv_variable_v2(cs, torch.zeros(4, 3 * D, device=dev))
    ops.kv_variable_group_sparse_apply_adam_v4(cv, cs, grad, ids, *hp, unique_indices=True)
    torch.cuda.synchronize()
    for h in (cv, cs):
      ops.kv_prepare_capture(h, 4 * n)
    s0 = torch.cuda.Stream()
    s0.wait_stream(torch.cuda.current_stream())
    g0 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g0, stream=s0):                      # (the refusal comes before anything is queued: the capture survives)
      with pytest.raises(_lib.FailedPreconditionError):
        ops.kv_variable_group_sparse_apply_adam_v4(cv, cs, grad, ids, *hp, unique_indices=True)
  for h in (gv, gs):
    ops.kv_prepare_capture(h, 4 * n)
  side = torch.cuda.Stream()
  side.wait_stream(torch.cuda.current_stream())
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g, stream=side):
    ops.kv_variable_group_sparse_apply_adam_v4(gv, gs, grad, ids, *hp, unique_indices=unique)
  for _ in range(3):
    g.replay()
    ops.kv_variable_group_sparse_apply_adam_v4(ev, es, grad, ids, *hp, unique_indices=unique)
  torch.cuda.synchronize()
  assert torch.equal(ops.kv_variable_gather_or_zeros_v2(gv, ids), ops.kv_variable_gather_or_zeros_v2(ev, ids))
  assert torch.equal(ops.kv_variable_gather_or_zeros_v2(gs, ids), ops.kv_variable_gather_or_zeros_v2(es, ids))
  assert ops.kv_variable_frequency(gs) == ops.kv_variable_frequency(es)
  assert ops.kv_variable_size_v2(gv) == n                                      # (a call after the replays: no error word is up)
