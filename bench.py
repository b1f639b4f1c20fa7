#!/usr/bin/env python3
"""bench.py — KvVariable hot path on MI355X: embedding_lookup (GatherOrInsert) + fused sparse
GroupAdam-V4 apply over one batch of int64 ids, BASELINE.json config 2:
  50M-key KvVariable x dim 32, 1M ids/batch Zipf(1.2), fp32.

One "step" = one lookup of the batch + one fused apply of the batch's gradients (dedup +
segment-sum + row update).  Inputs (ids, grads) are resident in HBM before the timed region.

  python bench.py [--gpus N --steps K --warmup W]           (N > 1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the dominant
kernel (HIP events on the op's own stream, kv_profile_*) and `cpu_baseline` (the CPU oracle —
a port of the reference algorithm — timed on this host on a bounded sample).
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 20250211 + 2
SAMPLE_EVERY = int(os.environ.get("KV_BENCH_SAMPLE_EVERY", "4"))   # the dominant kernel is bracketed by events on every 4th step of the timed region
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)


def splitmix64(x):
  """int64 tensor -> int64 tensor; keys = splitmix64(rank) spreads ranks over the int64 space."""
  def c(v):  # two's-complement constant
    return v - (1 << 64) if v >= (1 << 63) else v
  x = x + c(0x9E3779B97F4A7C15)
  # logical shifts on int64: mask off the sign-extended bits
  def lsr(v, s):
    return (v >> s) & ((1 << (64 - s)) - 1)
  x = (x ^ lsr(x, 30)) * c(0xBF58476D1CE4E5B9)
  x = (x ^ lsr(x, 27)) * c(0x94D049BB133111EB)
  return x ^ lsr(x, 31)


class Zipf(object):
  """Exact inverse-CDF Zipf(s) over ranks 1..K: tabulated head + Euler-Maclaurin tail."""

  def __init__(self, K, s, device, head=1 << 20):
    self.K, self.s = K, s
    M = min(head, K)
    self.M = M
    w = torch.arange(1, M + 1, dtype=torch.float64, device=device)**(-s)
    self.cdf = torch.cumsum(w, 0)
    self.head_mass = float(self.cdf[-1])
    a = 1.0 - s
    self.tail = lambda x: (x**a) / a  # antiderivative of x^-s
    self.tail_mass = float(self.tail(K + 0.5) - self.tail(M + 0.5)) if K > M else 0.0
    self.total = self.head_mass + self.tail_mass

  def sample(self, n, gen):
    u = torch.rand(n, dtype=torch.float64, device=self.cdf.device, generator=gen) * self.total
    r = torch.searchsorted(self.cdf, u.clamp(max=self.head_mass * (1 - 1e-15))) + 1
    if self.K > self.M:
      a = 1.0 - self.s
      t = (u - self.head_mass).clamp(min=0) + (self.M + 0.5)**a / a
      rt = torch.clamp(torch.round((t * a)**(1.0 / a)), self.M + 1, self.K).to(torch.int64)
      r = torch.where(u >= self.head_mass, rt, r)
    return r.to(torch.int64)


def cpu_baseline(args, D):
  """The oracle (a port of the reference's CPU algorithm: 1031-segment unordered_map, per-row heap
  buffers, rw spin locks, Shard-style contiguous blocks) on this host's cores, on a bounded sample:
  a table with --cpu-keys keys (instead of 50M) and the same 1M-id Zipf(1.2) batch shape."""
  from oracle import kv_oracle as ko
  cores = os.cpu_count() or 1
  K = args.cpu_keys
  rng = np.random.default_rng(SEED)
  table = (rng.standard_normal((10000, D)) * 0.05).astype(np.float32)
  var = ko.OracleKv(D, 0, table, day=20000, picker=1, seed=1, threads=cores)
  slot = ko.OracleKv(3 * D, 0, np.zeros((16, 3 * D), np.float32), day=20000, picker=1, seed=1, threads=cores)
  z = Zipf(K, args.zipf, torch.device("cpu"))
  g = torch.Generator().manual_seed(SEED)
  keys_all = splitmix64(torch.arange(1, K + 1, dtype=torch.int64)).numpy()
  t0 = time.perf_counter()
  for i in range(0, K, 1 << 20):
    var.gather_or_insert(keys_all[i:i + (1 << 20)])
  build_s = time.perf_counter() - t0
  N = args.batch
  steps = args.cpu_steps
  times = []
  for k in range(steps + 1):
    ids = splitmix64(z.sample(N, g)).numpy()
    grad = rng.normal(0, 1e-2, (N, D)).astype(np.float32)
    t0 = time.perf_counter()
    var.gather_or_insert(ids)
    u, s, _ = ko.dedup_segment_sum(ids, grad)          # TF-core unique + unsorted_segment_sum (1 thread)
    ko.apply_group_adam(var, slot, s, u, 1e-3, 0.9, 0.999, 0.9, 0.999, 1e-8)
    times.append(time.perf_counter() - t0)
  t = float(np.median(times[1:]))                      # first step inserts the slot rows
  # the same step on ONE thread (SURVEY.md §8d asks for both): one more batch, warm tables
  var.threads = 1
  ids = splitmix64(z.sample(N, g)).numpy()
  grad = rng.normal(0, 1e-2, (N, D)).astype(np.float32)
  t0 = time.perf_counter()
  var.gather_or_insert(ids)
  u, s, _ = ko.dedup_segment_sum(ids, grad)
  ko.apply_group_adam(var, slot, s, u, 1e-3, 0.9, 0.999, 0.9, 0.999, 1e-8)
  t1 = time.perf_counter() - t0
  return {"value": N / t, "unit": "ids/s", "cores": cores, "kind": "port", "value_1_thread": N / t1,
          "sample": "oracle/kv_oracle.cc, %d threads, %d-key table (not 50M), %d steps of %d Zipf(%.1f) ids: "
                    "lookup + tf.unique/segment_sum + GroupAdamV4; median %.3f s/step; table build %.1f s"
                    % (cores, K, steps, N, args.zipf, t, build_s)}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=50)
  ap.add_argument("--warmup", type=int, default=5)
  ap.add_argument("--keys", type=int, default=50_000_000)
  ap.add_argument("--batch", type=int, default=1_000_000)
  ap.add_argument("--dim", type=int, default=32)
  ap.add_argument("--zipf", type=float, default=1.2)
  ap.add_argument("--pool", type=int, default=8, help="distinct pre-generated batches cycled through")
  ap.add_argument("--cpu-keys", type=int, default=2_000_000)
  ap.add_argument("--cpu-steps", type=int, default=3)
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--no-kernel-events", action="store_true",
                  help="diagnostic: do not bracket kernels with HIP events in the timed region (no roofline)")
  ap.add_argument("--no-token", action="store_true",
                  help="diagnostic: the apply rebuilds the batch index instead of taking over the lookup's")
  ap.add_argument("--deterministic", action="store_true",
                  help="diagnostic: the tables' deterministic reduction mode (kv_set_deterministic)")
  ap.add_argument("--overlap", action="store_true",
                  help="diagnostic: overlap mode (kv_set_overlap) without graph capture: the forks and joins are event hops")
  ap.add_argument("--graph", action="store_true",
                  help="the timed steps replay HIP graphs captured in overlap mode (one graph per pooled batch)")
  ap.add_argument("--force-sharded", action="store_true",
                  help="run the all_to_all exchange path even with one rank (exercises the N > 1 code on one GPU)")
  args = ap.parse_args()

  rank = int(os.environ.get("RANK", "0"))
  world = int(os.environ.get("WORLD_SIZE", "1"))
  local = int(os.environ.get("LOCAL_RANK", "0"))
  if world != args.gpus:
    if world == 1 and args.gpus > 1:
      sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
  # KV_BENCH_ONE_GPU=1 (debugging only, never a measurement): every rank shares cuda:0 and the
  # collectives are staged through gloo on the host, so the N > 1 control flow of this file can be
  # exercised on a 1-GPU box (RCCL refuses two ranks on one device)
  one_gpu_debug = os.environ.get("KV_BENCH_ONE_GPU") == "1" and world > 1
  if one_gpu_debug:
    local = 0
  ndev = torch.cuda.device_count()
  if ndev > 0:
    local %= ndev          # a launcher that masks devices per rank leaves one visible GPU: index 0
  torch.cuda.set_device(local)
  dev = torch.device("cuda", local)
  shard_path = world > 1 or args.force_sharded
  if shard_path:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if one_gpu_debug:
      dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
      dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

  from tfplus_amd import _lib
  from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as ops
  from tfplus_amd.kv_variable.python.ops.sharded import owner_of as _owner_of
  L = _lib.lib()

  D, N = args.dim, args.batch
  # weak scaling: every rank owns ~args.keys keys of a table of args.keys * world keys, sharded by
  # floor_mod(key, world) (the reference's partition rule), and feeds its own batch of N ids
  K = args.keys * world
  K_local = args.keys if world == 1 else int(args.keys * 1.05) + 1024
  gen = torch.Generator(device=dev).manual_seed(SEED + rank)
  table = (torch.randn(10000, D, device=dev, generator=torch.Generator(device=dev).manual_seed(SEED)) * 0.05)
  var = ops.kv_variable([D], capacity_hint=K_local + 24 * N, device=local)
  slot = ops.kv_variable([3 * D], capacity_hint=K_local + 24 * N, device=local)
  ops.init_kv_variable_v2(var, table)
  ops.init_kv_variable_v2(slot, torch.zeros(16, 3 * D, device=dev))
  stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

  # ---- pre-insert K keys (ids = splitmix64(rank)); optimizer state rows too: steady state ----
  CH = 1 << 22
  buf = torch.empty((CH, 3 * D), dtype=torch.float32, device=dev)
  owned = 0
  for i in range(0, K, CH):
    r = torch.arange(i + 1, min(i + CH, K) + 1, dtype=torch.int64, device=dev)
    keys = splitmix64(r)
    if world > 1:
      keys = keys[_owner_of(keys, world) == rank].contiguous()      # hashed ownership: (mix64(id) >> 32) % world
    owned += keys.numel()
    if keys.numel() == 0:
      continue
    _lib.check(L.kv_gather_or_insert(var.ptr, keys.data_ptr(), None, keys.numel(), buf.data_ptr(), stream))
    _lib.check(L.kv_gather_or_insert(slot.ptr, keys.data_ptr(), None, keys.numel(), buf.data_ptr(), stream))
  torch.cuda.synchronize()
  del buf
  assert ops.kv_variable_shape_v2(var)[0] == owned, (ops.kv_variable_shape_v2(var), owned)
  # steady state of a trained table: every key the optimizer has applied has its slot row remembered in
  # the var's index entry (what GroupAdamOptimizer's first apply per key leaves behind)
  ops.kv_attach_slot(var, slot)
  if args.deterministic:
    ops.kv_set_deterministic(var, True); ops.kv_set_deterministic(slot, True)

  # ---- synthetic batches, resident in HBM ----
  z = Zipf(K, args.zipf, dev)
  pool = []
  for p in range(args.pool):
    ids = splitmix64(z.sample(N, gen))
    grad = torch.randn(N, D, device=dev, generator=gen) * 1e-2
    U = int(torch.unique(ids).numel())
    # entries of the batch = distinct (2048-id tile, id) pairs; S1 of them occur once in their tile (their gradient
    # row goes straight to the apply, the others' rows are summed per tile first)
    tl = torch.arange(N, device=dev, dtype=torch.int64) // 2048
    _, ec = torch.unique(torch.stack([tl, ids], 1), dim=0, return_counts=True)
    pool.append((ids, grad, U, int(ec.numel()), int((ec == 1).sum())))
    del tl, ec
  out = torch.empty((N, D), dtype=torch.float32, device=dev)
  U_mean = float(np.mean([p[2] for p in pool]))
  E_mean = float(np.mean([p[3] for p in pool]))
  S1_mean = float(np.mean([p[4] for p in pool]))

  state = {"b1p": np.float32(0.9), "b2p": np.float32(0.999)}

  native_shard = shard_path and not one_gpu_debug

  def peer_capacity():
    """The records one rank may send one owner per batch — a deployment constant: both ends of every send / recv
    must agree on it without talking, so it is the SAME number on every rank (max over ranks here).  Sized from
    the workload: the most distinct ids any pooled batch holds, shared evenly by the hash, plus a quarter."""
    m = torch.tensor([max(p[2] for p in pool)], dtype=torch.int64, device="cpu" if (one_gpu_debug or world == 1) else dev)
    if world > 1:
      dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return int(int(m.item()) / world * 1.25) + 1024
  if native_shard:
    # the production path: kvhip.h kv_shard_* over a kv_comm (RCCL grouped send / recv on its own stream; a world of
    # one still goes through RCCL here so that --force-sharded prices the whole mechanism)
    comm = ops.kv_comm_from_torch_distributed(local) if world > 1 else ops.KvComm(1, 0, ops.kv_comm_unique_id(), local)
    cap = peer_capacity()
    shard = ops.KvShard(var, world, rank, ops.KV_OWNER_HASH, max_ids=N, peer_capacity=cap)
    hp_t = ctypes.c_float * 9
    # this step has no dense tower to overlap with: queue it on the communicator's own stream, so the ops fork and
    # join nothing (with a tower on another stream each op pays one event hop in and one out, hidden behind it)
    comm_stream = comm.stream()
    torch.cuda.synchronize()
    torch.cuda.set_stream(comm_stream)
    stream = ctypes.c_void_p(comm_stream.cuda_stream)
  elif shard_path:
    # debugging mode (KV_BENCH_ONE_GPU=1): the same native phases, every rank on cuda:0, the fixed-size segments
    # moved between the processes through gloo on the host (RCCL refuses two ranks on one device)
    cap = peer_capacity()
    shard = ops.KvShard(var, world, rank, ops.KV_OWNER_HASH, max_ids=N, peer_capacity=cap)
    hp_t = ctypes.c_float * 9
    bufs = shard.buffers()

    def xchg(what):
      send, recv = (bufs["send_pairs"], bufs["recv_pairs"]) if what == 0 else (bufs["send_rows"], bufs["recv_rows"])
      torch.cuda.synchronize()
      o = torch.empty(send.numel(), dtype=torch.uint8)
      dist.all_to_all_single(o, send.cpu())
      recv.copy_(o)

  def teardown():
    """The library's communicator goes first (every rank past its last exchange), then torch's group."""
    torch.cuda.synchronize()
    if world > 1:
      dist.barrier()
    torch.cuda.set_stream(torch.cuda.default_stream(dev))   # the communicator's stream is about to go
    held = [x for x in ("shard", "comm") if x in live]
    for x in held:
      obj = live.pop(x)
      if x == "comm":
        _lib.check(L.kv_comm_destroy(obj.ptr)); obj.ptr = None
      else:
        _lib.check(L.kv_shard_destroy(obj.ptr)); obj.ptr = None
    dist.destroy_process_group()

  live = {}
  if shard_path:
    live["shard"] = shard
    if native_shard:
      live["comm"] = comm

  if args.overlap and not shard_path:
    ops.kv_set_overlap(var, True)

  def step(k):
    ids, grad = pool[k % len(pool)][:2]
    if not shard_path:
      st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)   # (the capture stream inside a graph capture)
      # the lookup names the batch; the apply of the same ids takes its index over (kvhip.h: batch token)
      tok = ctypes.c_uint64(0)
      _lib.check(L.kv_gather_or_insert_tok(var.ptr, ids.data_ptr(), None, N, out.data_ptr(), ctypes.byref(tok), st))
      _lib.check(L.kv_apply_group_adam_tok(var.ptr, slot.ptr, grad.data_ptr(), ids.data_ptr(), N, 1e-3,
                                           float(state["b1p"]), float(state["b2p"]), 0.9, 0.999, 1e-8, 0.0, 0.0,
                                           0.0, 4, tok.value if not args.no_token else 0, st))
    elif native_shard:
      # ids -> owners (grouped send / recv over xGMI) -> rows back; summed gradients -> owners -> fused apply
      _lib.check(L.kv_shard_lookup(shard.ptr, comm.ptr, ids.data_ptr(), N, out.data_ptr(), 1, stream))
      hp = hp_t(1e-3, float(state["b1p"]), float(state["b2p"]), 0.9, 0.999, 1e-8, 0.0, 0.0, 0.0)
      _lib.check(L.kv_shard_apply(shard.ptr, comm.ptr, 0, slot.ptr, None, grad.data_ptr(), hp, 1, stream))
    else:
      _lib.check(L.kv_shard_lookup_route(shard.ptr, ids.data_ptr(), N, stream)); xchg(0)
      _lib.check(L.kv_shard_lookup_serve(shard.ptr, stream)); xchg(1)
      _lib.check(L.kv_shard_lookup_finish(shard.ptr, out.data_ptr(), stream))
      _lib.check(L.kv_shard_apply_route(shard.ptr, grad.data_ptr(), stream)); xchg(1)
      hp = hp_t(1e-3, float(state["b1p"]), float(state["b2p"]), 0.9, 0.999, 1e-8, 0.0, 0.0, 0.0)
      _lib.check(L.kv_shard_apply_serve(shard.ptr, 0, slot.ptr, None, hp, stream))
    state["b1p"] = np.float32(state["b1p"] * np.float32(0.9))      # TF-core Adam _finish
    state["b2p"] = np.float32(state["b2p"] * np.float32(0.999))

  def barrier():
    torch.cuda.synchronize()
    if world > 1:
      dist.barrier()
      torch.cuda.synchronize()

  # A hipEvent pair around a kernel costs a few microseconds of stream time (five pairs per step
  # are ~16 % of this step), so only ONE kernel — the slowest one, found during the warm-up steps,
  # where every kernel is bracketed — carries events inside the timed region; that is the launch
  # `roofline` is computed from.  The per-kernel table (`kernels_ms`) comes from `steps` further
  # fully bracketed steps after the clock has stopped.
  ops.kv_profile_enable(var, 16 * args.warmup + 8)
  for k in range(args.warmup):
    step(k)
  torch.cuda.synchronize()
  warm = ops.kv_profile_read(var)
  dom = max(warm, key=lambda k: warm[k][0] / max(warm[k][1], 1)) if args.warmup > 0 else "apply_sorted"
  graphs = None
  if args.graph and not shard_path:
    # one graph per pooled batch, captured in overlap mode: the lookup's rows beside its tile pass, the partition pass
    # beside the apply's tile sums — graph edges instead of event hops, no launch gaps.  (The Adam powers b1p / b2p
    # are arguments by value: a replay uses the ones of its capture.)
    ops.kv_profile_enable(var, 0)
    torch.cuda.synchronize()
    for h in (var, slot):
      ops.kv_prepare_capture(h, (len(pool) + 2) * N)
    graphs = []
    for p in range(len(pool)):
      g = torch.cuda.CUDAGraph()
      with torch.cuda.graph(g):
        step(p)
      graphs.append(g)
    for g in graphs:           # untimed: every graph once
      g.replay()
    torch.cuda.synchronize()
  ops.kv_profile_enable(var, 0 if (args.no_kernel_events or graphs) else args.steps + 8)
  ops.kv_profile_select(var, [dom])
  ops.kv_profile_sample(var, SAMPLE_EVERY)     # a pair of event markers costs ~4 us of stream time per launch
  barrier()
  t0 = time.perf_counter()
  if graphs:
    for k in range(args.steps):
      graphs[(args.warmup + k) % len(graphs)].replay()
  else:
    for k in range(args.steps):
      step(args.warmup + k)
  torch.cuda.synchronize()
  t1 = time.perf_counter()
  dt = t1 - t0
  if world > 1:
    tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if one_gpu_debug else dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
  timed = ops.kv_profile_read(var)
  ops.kv_profile_select(var, None)
  ops.kv_profile_sample(var, 1)
  prof = timed
  if not args.no_kernel_events:
    ops.kv_profile_enable(var, 8 * args.steps + 8)
    for k in range(args.steps):
      step(args.warmup + args.steps + k)
    torch.cuda.synchronize()
    prof = ops.kv_profile_read(var)
  ops.kv_profile_enable(var, 0)

  ms_per_step = dt / args.steps * 1e3
  value = N * world / (dt / args.steps)
  if args.no_kernel_events or graphs:
    if rank == 0:
      print(json.dumps({"ms_per_step": ms_per_step, "value": value, "note": "diagnostic run without kernel events"}))
    if shard_path:
      teardown()
    return

  # ---- roofline of the dominant kernel: algorithmic bytes per launch / mean launch time ----
  # SURVEY.md §8(d) per-step figures, split over the kernels of each op by which kernel moves the
  # bytes: ids are read by the tile pass; the index probe and the output rows belong to
  # the partition pass (its gather blocks copy the rows); the apply reads every gradient row once and
  # reads + writes the optimizer state of every unique key in k_apply_sorted.
  Ub = U_mean
  fused = prof["apply_tsum"][1] > 0     # the entry-list pipeline ran (kv_fused.h)
  if fused:
    alg = {
        "lookup_tile": N * 8 + Ub * (16 + 4 * D) + N * 4 * D,   # k_ltile: ids, one probe + one row per key, output rows
        "lookup_part": 0,                                     # k_part2: row records only (not in SURVEY 8d's figure)
        "lookup_order": 0,
        "apply_index": N * 8 + Ub * 16,
        "apply_tsum": (N - S1_mean) * 4 * D,                   # k_tsum: gradient rows of ids repeated inside their tile
        "apply_sorted": S1_mean * 4 * D + Ub * (4 * 4 * D) + Ub * 4 * 4 * D,   # k_apply: the other gradient rows + state r/w
        "apply_span": 0,
    }
  else:
    alg = {
        "lookup_tile": N * 8,
        "lookup_part": Ub * 16,
        "lookup_order": Ub * 4 * D + N * 4 * D + N * 4,
        "apply_index": N * 8 + Ub * 16,
        "apply_tsum": 0,
        "apply_sorted": N * 4 * D + Ub * (4 * 4 * D) + Ub * 4 * 4 * D,
        "apply_span": 0,
    }
  kern = {k: (ms / max(c, 1)) for k, (ms, c) in prof.items()}
  dom_ms = timed[dom][0] / max(timed[dom][1], 1)      # the dominant kernel, inside the timed region
  achieved = alg[dom] / (dom_ms * 1e-3) / 1e9
  lookup_ms = kern["lookup_tile"] + kern["lookup_part"] + kern["lookup_order"]
  apply_ms = kern["apply_index"] + kern["apply_sorted"] + kern["apply_span"] + kern["apply_tsum"]
  lookup_bytes = N * (8 + 4 * D) + Ub * (16 + 4 * D)
  apply_bytes = N * (8 + 4 * D) + Ub * (16 + 4 * 4 * D) + Ub * 4 * 4 * D

  # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process, so the
  # figure is the one collected by scripts_prof.sh (rocprofv3 --pmc passes of this same command) and
  # committed under profiles/; null when that file is absent or was taken on another workload.
  traffic, traffic_src = None, None
  try:
    tj = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r02_traffic.json")))
    if tj.get("workload") == [K, N, D, args.zipf] and dom in tj["kernels"]:
      traffic = tj["kernels"][dom]["hbm_bytes"]
      traffic_src = "profiles/r02_traffic.json: " + tj["source"]
  except (OSError, ValueError, KeyError):
    pass

  res = {
      "metric": "lookups+GroupAdam-applies/sec and HBM GB/s, 1M int64 ids x dim32",
      "value": value,
      "unit": "ids/s (each id: 1 embedding lookup + 1 fused GroupAdam-V4 apply)",
      "n_gpus": world,
      "steps": args.steps,
      "warmup": args.warmup,
      "ms_per_step": ms_per_step,
      "higher_is_better": True,
      "scaling": "weak",
      "vs_baseline": None,
      "dtype": "f32",
      "data": "synthetic",
      "config": {"workload": "configs[1]: %dM-key KvVariable x dim%d per GPU, %d ids/batch per GPU Zipf(%.1f), "
                             "lookup + sparse GroupAdam apply" % (args.keys // 1_000_000, D, N, args.zipf),
                 "keys": K, "dim": D, "batch": N, "zipf": args.zipf, "global_batch": N * world, "unique_per_batch": Ub,
                 "tile_entries_per_batch": E_mean, "tile_entries_single": S1_mean,
                 "parallelism": ("table sharded over %d GPUs by (mix64(id) >> 32) %% G, fixed-capacity id/row/grad exchange, "
                                 "grouped ncclSend/ncclRecv over RCCL" % world) if world > 1 else "single GPU"},
      "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                   "algorithmic_bytes_per_launch": alg[dom], "avg_launch_ms": dom_ms,
                   "launches_timed": int(timed[dom][1]),
                   "measured": "hipEvent pairs on the op's stream around every %dth launch of this kernel inside "
                               "the timed region" % SAMPLE_EVERY},
      "kernels_ms": kern,
      "kernels_ms_measured": "%d further steps after the timed region with every kernel bracketed" % args.steps,
      "ops": {"lookup": {"gpu_ms": lookup_ms, "algorithmic_bytes": lookup_bytes,
                         "GBps": lookup_bytes / (lookup_ms * 1e-3) / 1e9,
                         "frac_of_peak": lookup_bytes / (lookup_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "lookups_per_s": N / (lookup_ms * 1e-3)},
              "group_adam_apply": {"gpu_ms": apply_ms, "algorithmic_bytes": apply_bytes,
                                   "GBps": apply_bytes / (apply_ms * 1e-3) / 1e9,
                                   "frac_of_peak": apply_bytes / (apply_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "applies_per_s": N / (apply_ms * 1e-3),
                                   "unique_applies_per_s": Ub / (apply_ms * 1e-3)}},
  }
  if shard_path:
    # On the sharded path every rank runs the same kernels on what the exchange hands it (unique
    # ids per source rank; sizes change with every batch), so the per-launch byte model above does
    # not describe those launches: the kernels are characterised by the N = 1 line, this line
    # carries the measured launch time only.
    res["roofline"].update({"achieved": None, "frac": None, "traffic": None, "traffic_source": None,
                            "algorithmic_bytes_per_launch": None,
                            "note": "sharded path: kernel byte model is reported on the single-GPU line "
                                    "(launch sizes here depend on the exchanged unique ids)"})
    res.pop("ops", None)
    # xGMI traffic is reported apart from HBM: three fixed-size exchanges per step, every rank sends
    # every peer one segment of peer_capacity + 1 records — (id, count) pairs, rows back, summed gradient rows
    seg = cap + 1
    res["exchange"] = {"exchanges_per_step": 3, "peer_capacity_records": cap,
                       "wire_bytes_per_rank_per_step": (world - 1) * seg * (16 + 2 * 4 * D),
                       "payload_bytes_per_rank_per_step_estimate": int((world - 1) / world * Ub * (16 + 2 * 4 * D)),
                       "transport": "grouped ncclSend / ncclRecv (RCCL) on the communicator's stream; a rank's own "
                                    "segment is a device copy" if native_shard else "debug: host-staged through gloo"}
  if rank == 0 and world == 1 and not args.no_cpu_baseline:
    res["cpu_baseline"] = cpu_baseline(args, D)
  if rank == 0:
    print(json.dumps(res), flush=True)
  if shard_path:
    teardown()


if __name__ == "__main__":
  main()
