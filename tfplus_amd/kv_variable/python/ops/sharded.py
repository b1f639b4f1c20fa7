"""Hash-sharded KvVariable across the GPUs of one node: one table shard per rank, ids and rows /
gradients exchanged with all_to_all over RCCL (xGMI is point to point, so each peer pair uses its
own link; no ring collective and no all-reduce anywhere on this path).

The reference has no communication layer: multi-device there is TF1 PS/worker placement of
partitioned variables with `ids % num_shards` (python/ops/embedding_ops.py:115-204,
kernels/utility.h:90-107).  Ownership here is hashed by default ((mix64(id) >> 32) % world: raw Criteo-style ids
are not uniform mod 8); the reference's floor-mod rule is the "mod" option, under which a checkpoint
partitioned by the reference maps shard-for-shard onto ranks.

This module is the torch.distributed statement of the exchange (variable-size all_to_all; runs on CPU / gloo in
the tests).  The production path is native: gen_kv_variable_ops.KvShard / KvComm over kvhip.h's kv_shard_* and
kv_comm_* — fixed-capacity segments, no size exchange, no host synchronisation, RCCL on a stream of its own.

Per lookup and rank:  local unique-with-counts -> bucket the unique ids by owner -> all_to_all of
bucket sizes (the one host sync) -> all_to_all of (id, occurrence count) pairs -> owner-side
GatherOrInsertWithCounts on its shard -> all_to_all of rows back -> expand straight from the
exchange order to the input order.  The apply of the same batch sums gradients into that exchange
order (unsorted_segment_sum by position) and sends them: one all_to_all, one fused optimizer call
at the owner, which still holds the ids it served.

The exchange is written against torch.distributed only (works on CPU/gloo for the tests and on
GPU/RCCL in production); the shard itself is any object with the KvVariable lookup/apply calls.
"""
import torch
import torch.distributed as dist


class Routing(object):
  """Where each local id goes and how to undo it."""
  __slots__ = ("perm", "send_counts", "recv_counts", "n_local", "n_recv", "bucketed_ids", "pairs", "pos")

  def __init__(self, perm, send_counts, recv_counts):
    self.perm, self.send_counts, self.recv_counts = perm, send_counts, recv_counts
    self.bucketed_ids = None
    self.pairs = self.pos = None     # optional extras of the GPU bucket kernel (see route)
    self.n_local = int(perm.numel())
    self.n_recv = int(sum(recv_counts))


def _mix64(x):
  """the library's mix64 (kv_device.h) on int64 tensors: wrapping multiplies, logical shifts."""
  def lsr(v, s):
    return (v >> s) & ((1 << (64 - s)) - 1)
  def c(v):
    return v - (1 << 64) if v >= (1 << 63) else v
  x = x ^ lsr(x, 33)
  x = x * c(0xff51afd7ed558ccd)
  x = x ^ lsr(x, 33)
  x = x * c(0xc4ceb9fe1a85ec53)
  return x ^ lsr(x, 33)


def owner_of(ids, world, rule="hash"):
  """Owner rank of every id.  "hash" (default): (mix64(id) >> 32) % world — balanced whatever the ids
  look like (kvhip.h KV_OWNER_HASH).  "mod": floor-mod (negative ids included), the reference's ModKeyImpl
  (utility.h:90-107), for checkpoints partitioned by it."""
  if rule == "mod":
    return torch.remainder(ids, world)
  h = _mix64(ids.to(torch.int64))
  # the high half of the hash (the table index takes its home slot from the low bits), as an unsigned 32-bit value
  return torch.remainder((h >> 32) & 0xFFFFFFFF, world)


def route(ids, group=None, bucket_fn=None, known_counts=None, n_dev=None, id_counts=None, rule="hash"):
  """Buckets a flat id tensor by owner rank and exchanges the bucket sizes.  `bucket_fn(ids, world
  [, n_dev])` -> (bucketed ids, perm, counts) is the GPU counting sort (kv_bucket_by_owner); without
  it the same thing is done with torch ops (CPU tests).  known_counts = (send, recv) python lists
  skips the size exchange and its host sync.  n_dev (1-element int64 device tensor): only the first
  n_dev[0] ids are real and that number is not on the host yet — it comes back with the bucket sizes
  in the SAME device->host copy, so the lookup has one sync instead of two; returns (routing, n)."""
  world = dist.get_world_size(group)
  bucketed = pairs = pos = None
  if bucket_fn is not None:
    res = bucket_fn(ids, world) if n_dev is None else bucket_fn(ids, world, n_dev, id_counts)
    bucketed, perm, send = res[:3]
    if len(res) == 5:                 # (id, count) pairs in bucket order + inverse of perm, from the same kernel
      pairs, pos = res[3], res[4]
  else:
    own = owner_of(ids, world, rule)
    perm = torch.argsort(own, stable=True)
    send = torch.bincount(own, minlength=world).to(torch.int64)
  if known_counts is not None:
    rt = Routing(perm, list(known_counts[0]), list(known_counts[1]))
    rt.bucketed_ids = bucketed
    return rt
  recv = torch.empty_like(send)
  dist.all_to_all_single(recv, send, group=group)
  if n_dev is not None:
    host = torch.cat([send, recv, n_dev.reshape(1)]).tolist()    # the one device -> host sync of a lookup
    n = int(host[-1])
    rt = Routing(perm[:n], [int(x) for x in host[:world]], [int(x) for x in host[world:2 * world]])
    rt.bucketed_ids = None if bucketed is None else bucketed[:n]
    rt.pairs = None if pairs is None else pairs[:n]
    rt.pos = pos
    return rt, n
  both = torch.stack([send, recv]).tolist()      # one device -> host sync for both count vectors
  rt = Routing(perm, [int(x) for x in both[0]], [int(x) for x in both[1]])
  rt.bucketed_ids = bucketed
  return rt


def _take(payload, index, take_fn, scatter=False):
  if take_fn is not None:
    return take_fn(payload, index, scatter)
  if not scatter:
    return payload.index_select(0, index.to(torch.int64)).contiguous()
  out = torch.empty_like(payload)
  out.index_copy_(0, index.to(torch.int64), payload)
  return out


def exchange(rt, payload, reverse=False, group=None, presorted=None, take_fn=None, unpermute=True):
  """all_to_all of per-id rows.  Forward: `payload` is in local order (or `presorted` already in
  exchange order = grouped by owner), the result is what this rank must serve (grouped by source
  rank).  reverse=True: `payload` is in served order, the result is back in exchange order, or in
  local order when unpermute.  take_fn(rows, index, scatter) is the GPU row permutation
  (kv_take_rows); torch indexing otherwise."""
  if not reverse:
    src = presorted if presorted is not None else _take(payload, rt.perm, take_fn)
    out = torch.empty((rt.n_recv,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    dist.all_to_all_single(out, src.contiguous(), output_split_sizes=rt.recv_counts, input_split_sizes=rt.send_counts,
                           group=group)
    return out
  tail = tuple(payload.shape[1:])
  back = torch.empty((rt.n_local,) + tail, dtype=payload.dtype, device=payload.device)
  dist.all_to_all_single(back, payload.contiguous(), output_split_sizes=rt.send_counts,
                         input_split_sizes=rt.recv_counts, group=group)
  return _take(back, rt.perm, take_fn, scatter=True) if unpermute else back


class ShardedKvVariable(object):
  """One logical KvVariable whose rows live on the rank that owns `id mod world`.

  Both directions de-duplicate locally BEFORE the exchange (a Zipf batch shrinks ~9x): lookups send
  unique ids with their occurrence counts (so the owner's frequency words still count every
  occurrence) and expand the returned rows locally; applies send one summed gradient row per
  unique id (tf.unique + unsorted_segment_sum, the TF-core step, done per rank) and the owner's
  fused apply sums once more across ranks.

  Collectives per step: lookup = bucket sizes, (ids, counts) packed in one payload, rows back;
  apply of the batch that was just looked up = summed gradients only — the owner still holds the
  ids it served and the sender still knows where each id sits in the exchange order, so neither
  ids nor sizes travel again.  Every rank must make the same calls in the same order (SPMD); the
  short backward path is taken when `ids` is the very tensor the last lookup saw, so all ranks
  must either reuse their tensor or not.

  unique_fn(ids, counts) -> (uniq, counts, inverse), segsum_fn(ids, grad) -> (uniq, summed),
  index_sum_fn(grad, index, num) -> [num, D] (unsorted_segment_sum), bucket_fn and take_fn are the
  GPU kernels (gen_kv_variable_ops.kv_unique / kv_dedup_segment_sum / kv_unsorted_segment_sum /
  kv_bucket_by_owner / kv_take_rows).  Without them torch ops do the same (CPU tests)."""

  def __init__(self, shard, group=None, bucket_fn=None, unique_fn=None, segsum_fn=None, take_fn=None,
               index_sum_fn=None, unique_async_fn=None, owner_rule="hash"):
    self.owner_rule = owner_rule   # "hash" (default) or "mod"; a bucket_fn must implement the same rule
    self.shard = shard          # the rank-local table (KvVariable, or any stand-in with the same calls)
    self.group = group
    self.bucket_fn, self.unique_fn, self.segsum_fn = bucket_fn, unique_fn, segsum_fn
    self.take_fn, self.index_sum_fn = take_fn, index_sum_fn
    # unique_async_fn(ids, counts) -> (uniq [n], counts [n], inverse, U on the device): no host sync
    # (gen_kv_variable_ops.kv_unique(sync=False)); bucket_fn must then accept (ids, world, n_dev)
    self.unique_async_fn = unique_async_fn
    self.world = dist.get_world_size(group)
    self.rank = dist.get_rank(group)
    self._last = None           # what the last lookup left behind for its backward pass

  def _unique(self, flat, counts):
    if self.unique_fn is not None:
      return self.unique_fn(flat, counts)
    uniq, inv = torch.unique(flat, return_inverse=True)
    c = torch.ones_like(flat, dtype=torch.int64) if counts is None else counts.reshape(-1).to(torch.int64).clamp(max=65535)
    ucnt = torch.zeros(uniq.numel(), dtype=torch.int64, device=flat.device).index_add_(0, inv, c)
    return uniq, ucnt.clamp(max=65535).to(torch.int32), inv

  def _segsum(self, flat, grad):
    if self.segsum_fn is not None:
      u, s = self.segsum_fn(flat, grad)[:2]
      return u, s
    uniq, inv = torch.unique(flat, return_inverse=True)
    summed = torch.zeros((uniq.numel(), grad.shape[1]), dtype=grad.dtype, device=grad.device).index_add_(0, inv, grad)
    return uniq, summed

  def _index_sum(self, grad, index, num):
    if self.index_sum_fn is not None:
      return self.index_sum_fn(grad, index, num)
    return torch.zeros((num, grad.shape[1]), dtype=grad.dtype, device=grad.device).index_add_(0, index.to(torch.int64), grad)

  def lookup(self, ids, counts=None):
    """embedding_lookup over the sharded table; returns rows in the order of `ids`."""
    flat = ids.reshape(-1)
    if self.unique_async_fn is not None and self.bucket_fn is not None and flat.numel() > 0:
      uniq, ucnt, inv, nu_dev = self.unique_async_fn(flat, counts)
      rt, U = route(uniq, self.group, self.bucket_fn, n_dev=nu_dev, id_counts=ucnt, rule=self.owner_rule)
      uniq, ucnt = uniq[:U], ucnt[:U]
    else:
      uniq, ucnt, inv = self._unique(flat, counts)
      U = int(uniq.numel())
      rt = route(uniq, self.group, self.bucket_fn, rule=self.owner_rule)
    # one payload for ids and their occurrence counts
    if rt.pairs is not None:
      payload = rt.pairs
    else:
      bids = rt.bucketed_ids if rt.bucketed_ids is not None else _take(uniq, rt.perm, self.take_fn)
      payload = torch.stack([bids.to(torch.int64), _take(ucnt, rt.perm, self.take_fn).to(torch.int64)], 1)
    got = exchange(rt, None, group=self.group, presorted=payload)
    served = got[:, 0].contiguous().to(uniq.dtype)       # kept for the backward pass of this batch
    if hasattr(self.shard, "sparse_read_pairs"):          # the shard takes the (id, count) payload as it arrived
      rows = self.shard.sparse_read_pairs(got)
    else:
      rows = self.shard.sparse_read_with_counts(served, got[:, 1].to(torch.int32))
    back = exchange(rt, rows, reverse=True, group=self.group, unpermute=False)       # in exchange order
    # where each input id sits in the exchange order: pos[perm[j]] = j, then through the inverse
    pos = rt.pos if rt.pos is not None else \
        _take(torch.arange(U, dtype=torch.int32, device=flat.device), rt.perm, self.take_fn, scatter=True)
    where = _take(pos, inv, self.take_fn)
    self._last = (ids, ids._version, rt, where, U, served)
    out = _take(back, where, self.take_fn)
    return out.reshape(tuple(ids.shape) + tuple(out.shape[1:]))

  def apply_gradients(self, apply_fn, grad, ids):
    """Sends one summed gradient row per unique id to its owner; each owner runs
    apply_fn(shard, grad, ids) once — its fused dedup + segment-sum + row update."""
    flat = ids.reshape(-1)
    g2 = grad.reshape(flat.numel(), -1)
    last = self._last
    if last is not None and last[0] is ids and last[1] == ids._version:
      # backward of the last lookup: sum straight into the exchange order it established
      _, _, rt, where, U, served = last
      g = exchange(rt, None, group=self.group, presorted=self._index_sum(g2, where, U))
      apply_fn(self.shard, g, served)
      return
    uniq, summed = self._segsum(flat, g2)
    rt = route(uniq, self.group, self.bucket_fn, rule=self.owner_rule)
    served = exchange(rt, uniq, group=self.group, presorted=rt.bucketed_ids, take_fn=self.take_fn)
    g = exchange(rt, summed, group=self.group, take_fn=self.take_fn)
    apply_fn(self.shard, g, served)
