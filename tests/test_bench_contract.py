"""bench.py's pieces that need no GPU: the `exchange` object of a sharded line at N = 8 (the driver's largest run; no
8-GPU node is available to the tests), parsed dry."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
  spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
  m = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(m)
  return m


def test_exchange_block_of_the_eight_gpu_line():
  b = _bench()
  D, cap, distinct = 32, 18000, 109000
  x = b.exchange_block(8, cap, D, distinct, lossless=False, staged=False)
  assert x["exchanges_per_step"] == 3 and x["peer_capacity_records"] == cap and x["lossless"] is False
  # seven peers, one fixed segment each per exchange: (id, count) records out, rows back, gradient rows out
  assert x["wire_bytes_per_rank_per_step"] == 7 * (cap + 1) * (16 + 4 * D + 4 * D)
  assert x["payload_bytes_per_rank_per_step_estimate"] == int(7 / 8 * distinct * (16 + 8 * D))
  assert x["payload_bytes_per_rank_per_step_estimate"] <= x["wire_bytes_per_rank_per_step"]
  assert "RCCL" in x["transport"] and "stays in place" in x["transport"]
  # a world of one puts nothing on the wire; the rehearsal transport says what it is
  assert b.exchange_block(1, cap, D, distinct, True, False)["wire_bytes_per_rank_per_step"] == 0
  assert "staged" in b.exchange_block(2, cap, D, distinct, True, True)["transport"]
