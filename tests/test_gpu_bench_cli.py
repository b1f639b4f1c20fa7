"""bench.py's contract: the default single-GPU line carries every required key, and the N > 1
control flow runs (two ranks on one GPU, collectives staged through gloo — a debugging mode of
bench.py; real runs use RCCL with one rank per GPU)."""
import json
import os
import socket
import subprocess
import sys

import pytest

torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"}


def _line(out):
  rows = [l for l in out.splitlines() if l.startswith("{")]
  assert len(rows) == 1, out[-2000:]
  return json.loads(rows[0])


@pytest.mark.gpu
def test_bench_single_gpu_line():
  if not torch.cuda.is_available():
    pytest.skip("needs a GPU")
  r = subprocess.run([sys.executable, "bench.py", "--steps", "5", "--warmup", "2", "--keys", "2000000", "--cpu-keys", "100000",
                      "--cpu-steps", "1"], cwd=ROOT, capture_output=True, text=True, timeout=600)
  assert r.returncode == 0, r.stderr[-2000:]
  d = _line(r.stdout)
  assert REQUIRED <= set(d) and d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["value"] > 0
  rf = d["roofline"]
  assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
  cb = d["cpu_baseline"]
  assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
  # what the drop-in runs, next to the headline: op boundary, no token, staged through pinned host memory; other skews
  assert d["op_boundary"]["ms"] > 0 and d["no_token"]["ms_per_step"] > 0 and d["staged"]["ms_per_step"] > 0
  assert [x["zipf"] for x in d["skew_sweep"]] == [0.3, 0.8, 1.2] and "repeated_id_tolerance" in d
  assert all(0 < x["lookup_frac"] < 1 and x["lookup_rows_ready_frac"] >= x["lookup_frac"] * 0.9 for x in d["skew_sweep"])
  # round 6: the unchanged graph's step and the sharded mechanism at world 1 are on the driver's line
  ug = d["unchanged_graph"]
  assert ug["ms_per_step"] > 0 and set(ug["parts_ms"]) == {"lookup_complete", "dedup_segment_sum", "apply_unique"}
  sw = d["sharded_world1"]
  assert "error" not in sw, sw
  assert sw["ms_per_step"] > 0 and set(sw["phases_ms"]) == set(PHASES) and sw["overflowed_batches"] == 0
  assert sw["phases_sum_ms"] >= 0.8 * sw["ms_per_step"]          # the phases account for the step (markers cost time: >= )
  oo = d["occurrence_order"]
  assert oo["ms_per_step"] > d["ms_per_step"] and oo["ids_per_s"] > 0      # the reference's bits cost a chain per key
  ph = cb["phases_s"]
  assert set(ph) == {"lookup", "dedup_1_thread", "apply"} and all(v > 0 for v in ph.values())
  assert str(cb["cores"]) in cb["threads_sweep_ids_per_s"] and cb["cores"] <= cb["host_cores"]


@pytest.mark.gpu
def test_bench_two_ranks_control_flow():
  if not torch.cuda.is_available():
    pytest.skip("needs a GPU")
  s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
  env = dict(os.environ, KV_BENCH_ONE_GPU="1")
  r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                      "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1",
                      "--keys", "1000000", "--batch", "200000"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
  assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
  d = _line(r.stdout)
  assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 400000 and d["value"] > 0
  _check_phases(d, 2)


PHASES = ("route", "exchange_ids", "serve", "exchange_rows", "finish", "presum", "exchange_grads", "apply")


def _check_phases(d, world):
  """the sharded line says where its step goes: every phase timed, the handshake's rank count, the wire rate per exchange"""
  assert set(d["phases_ms"]) == set(PHASES) and all(d["phases_ms"][p] is not None and d["phases_ms"][p] >= 0 for p in PHASES)
  assert d["phases_samples"] >= 1 and d["rccl_ranks_seen"] == world
  assert abs(d["phases_sum_ms"] - sum(d["phases_ms"].values())) < 1e-9
  w = d["exchange"]["wire_GBps"]
  assert set(w) == {"exchange_ids", "exchange_rows", "exchange_grads"}
  if world > 1:
    assert all(v is not None and v > 0 for v in w.values())


@pytest.mark.gpu
def test_forced_sharded_world_of_one_accounts_for_its_step():
  """`--force-sharded` at one rank runs the N > 1 ops through RCCL itself; the phases of the whole ops add up to the step
  (the event markers cost stream time: the sum may run over ms_per_step, not under it)"""
  if not torch.cuda.is_available():
    pytest.skip("needs a GPU")
  r = subprocess.run([sys.executable, "bench.py", "--force-sharded", "--steps", "8", "--warmup", "2", "--keys", "2000000",
                      "--batch", "400000", "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=600)
  assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
  d = _line(r.stdout)
  _check_phases(d, 1)
  assert 0.9 * d["ms_per_step"] <= d["phases_sum_ms"] <= 1.35 * d["ms_per_step"], (d["ms_per_step"], d["phases_ms"])


@pytest.mark.gpu
def test_bench_launches_its_own_ranks():
  """`python bench.py --gpus 2` without a launcher: the parent starts the ranks as child processes before it touches
  the GPU and relays rank 0's line (how the driver may call it)."""
  if not torch.cuda.is_available():
    pytest.skip("needs a GPU")
  env = dict(os.environ, KV_BENCH_ONE_GPU="1")
  for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
    env.pop(k, None)
  r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--keys", "1000000",
                      "--batch", "200000"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
  assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
  d = _line(r.stdout)
  assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 400000 and d["value"] > 0
  assert d["exchange"]["wire_bytes_per_rank_per_step"] > 0


def _two_ranks(extra, timeout=600):
  env = dict(os.environ, KV_BENCH_ONE_GPU="1")
  for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
    env.pop(k, None)
  return subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--keys", "1000000",
                         "--batch", "200000"] + extra, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.gpu
def test_bench_two_ranks_lossless_agreement():
  """The library's default mode on the N > 1 path: before every exchange the ranks agree on the capacity (here through
  the staged communicator's max callback, over RCCL an all-reduce) — the same whole ops the real run calls."""
  if not torch.cuda.is_available():
    pytest.skip("needs a GPU")
  r = _two_ranks(["--lossless"])
  assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
  d = _line(r.stdout)
  assert d["n_gpus"] == 2 and d["exchange"]["lossless"] is True and d["value"] > 0


@pytest.mark.gpu
def test_bench_two_ranks_that_disagree_on_capacity_fail_fast():
  """Ranks created with different peer_capacity must not hang in an exchange of mismatched sizes nor read each other's
  padding: the first sharded op verifies {world, rank, capacity, dim, owner rule} across the ranks and every rank
  fails with FAILED_PRECONDITION (kvhip.hip shard_verify)."""
  if not torch.cuda.is_available():
    pytest.skip("needs a GPU")
  r = _two_ranks(["--debug-capacity-skew", "7"], timeout=300)
  assert r.returncode != 0
  assert "peer_capacity" in r.stderr and "FailedPrecondition" in r.stderr, r.stderr[-2000:]
  assert not [l for l in r.stdout.splitlines() if l.startswith("{")]   # no line from a run that did not happen
