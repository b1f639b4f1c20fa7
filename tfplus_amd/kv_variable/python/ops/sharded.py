"""Hash-sharded KvVariable across the GPUs of one node: one table shard per rank, ids and rows /
gradients exchanged with all_to_all over RCCL (xGMI is point to point, so each peer pair uses its
own link; no ring collective and no all-reduce anywhere on this path).

The reference has no communication layer: multi-device there is TF1 PS/worker placement of
partitioned variables with `ids % num_shards` (python/ops/embedding_ops.py:115-204,
kernels/utility.h:90-107).  The same floor-mod rule decides ownership here, so a checkpoint
partitioned by the reference maps shard-for-shard onto ranks.

Per lookup and rank:  local unique-with-counts -> bucket the unique ids by owner -> all_to_all
(counts) -> all_to_all(ids, occurrence counts) -> owner-side GatherOrInsertWithCounts on its shard
-> all_to_all(rows) back -> un-permute -> expand to the input order.  The apply mirrors it with
(unique ids, locally summed grads) to the owner and one fused optimizer call there.

The exchange is written against torch.distributed only (works on CPU/gloo for the tests and on
GPU/RCCL in production); the shard itself is any object with the KvVariable lookup/apply calls.
"""
import torch
import torch.distributed as dist


class Routing(object):
  """Where each local id goes and how to undo it."""
  __slots__ = ("perm", "send_counts", "recv_counts", "n_local", "n_recv", "bucketed_ids")

  def __init__(self, perm, send_counts, recv_counts):
    self.perm, self.send_counts, self.recv_counts = perm, send_counts, recv_counts
    self.bucketed_ids = None
    self.n_local = int(perm.numel())
    self.n_recv = int(sum(recv_counts))


def owner_of(ids, world):
  """floor-mod ownership (negative ids included), the reference's ModKeyImpl (utility.h:90-107)."""
  return torch.remainder(ids, world)


def route(ids, group=None, bucket_fn=None, known_counts=None):
  """Buckets a flat id tensor by owner rank and exchanges the bucket sizes.  `bucket_fn(ids, world)`
  -> (bucketed ids, perm, counts) is the GPU counting sort (kv_bucket_by_owner); without it the
  same thing is done with torch ops (CPU tests).  known_counts = (send, recv) python lists skips
  the size exchange and its host sync: valid when `ids` is a re-ordering of a set routed before
  (the backward pass of a lookup), since bucket sizes depend only on the set."""
  world = dist.get_world_size(group)
  bucketed = None
  if bucket_fn is not None:
    bucketed, perm, send = bucket_fn(ids, world)
  else:
    own = owner_of(ids, world)
    perm = torch.argsort(own, stable=True)
    send = torch.bincount(own, minlength=world).to(torch.int64)
  if known_counts is not None:
    rt = Routing(perm, list(known_counts[0]), list(known_counts[1]))
    rt.bucketed_ids = bucketed
    return rt
  recv = torch.empty_like(send)
  dist.all_to_all_single(recv, send, group=group)
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


def exchange(rt, payload, reverse=False, group=None, presorted=None, take_fn=None):
  """all_to_all of per-id rows.  Forward: `payload` is in local order, the result is what this
  rank must serve (grouped by source rank).  reverse=True: `payload` is in served order, the
  result is back in local order.  take_fn(rows, index, scatter) is the GPU row permutation
  (kv_take_rows); torch indexing otherwise."""
  tail = tuple(payload.shape[1:])
  if not reverse:
    src = presorted if presorted is not None else _take(payload, rt.perm, take_fn)
    out = torch.empty((rt.n_recv,) + tail, dtype=payload.dtype, device=payload.device)
    dist.all_to_all_single(out, src, output_split_sizes=rt.recv_counts, input_split_sizes=rt.send_counts,
                           group=group)
    return out
  back = torch.empty((rt.n_local,) + tail, dtype=payload.dtype, device=payload.device)
  dist.all_to_all_single(back, payload.contiguous(), output_split_sizes=rt.send_counts,
                         input_split_sizes=rt.recv_counts, group=group)
  return _take(back, rt.perm, take_fn, scatter=True)


class ShardedKvVariable(object):
  """One logical KvVariable whose rows live on the rank that owns `id mod world`.

  Both directions de-duplicate locally BEFORE the exchange (a Zipf batch shrinks ~9x): lookups send
  unique ids with their occurrence counts (so the owner's frequency words still count every
  occurrence) and expand the returned rows locally; applies send one summed gradient row per
  unique id (tf.unique + unsorted_segment_sum, the TF-core step, done per rank) and the owner's
  fused apply sums once more across ranks.

  unique_fn(ids, counts) -> (uniq, counts, inverse) and segsum_fn(ids, grad) -> (uniq, summed)
  are the GPU kernels (gen_kv_variable_ops.kv_unique / kv_dedup_segment_sum); bucket_fn is
  kv_bucket_by_owner.  Without them torch ops do the same (CPU tests)."""

  def __init__(self, shard, group=None, bucket_fn=None, unique_fn=None, segsum_fn=None, take_fn=None):
    self.shard = shard          # the rank-local table (KvVariable, or any stand-in with the same calls)
    self.group = group
    self.bucket_fn, self.unique_fn, self.segsum_fn = bucket_fn, unique_fn, segsum_fn
    self.take_fn = take_fn
    self._last = None           # (ids tensor, its version, send counts, recv counts) of the last lookup
    self.world = dist.get_world_size(group)
    self.rank = dist.get_rank(group)

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

  def lookup(self, ids, counts=None):
    """embedding_lookup over the sharded table; returns rows in the order of `ids`."""
    flat = ids.reshape(-1)
    uniq, ucnt, inv = self._unique(flat, counts)
    rt = route(uniq, self.group, self.bucket_fn)
    self._last = (ids, ids._version, rt.send_counts, rt.recv_counts)
    served = exchange(rt, uniq, group=self.group, presorted=rt.bucketed_ids)
    sc = exchange(rt, ucnt, group=self.group, take_fn=self.take_fn)
    rows = self.shard.sparse_read_with_counts(served, sc)
    urows = exchange(rt, rows, reverse=True, group=self.group, take_fn=self.take_fn)
    out = _take(urows, inv, self.take_fn)
    return out.reshape(tuple(ids.shape) + tuple(out.shape[1:]))

  def apply_gradients(self, apply_fn, grad, ids):
    """Sends (unique ids, locally summed grads) to the owners; each owner runs
    apply_fn(shard, grad, ids) once — its fused dedup + segment-sum + row update."""
    flat = ids.reshape(-1)
    uniq, summed = self._segsum(flat, grad.reshape(flat.numel(), -1))
    known = None
    if self._last is not None and self._last[0] is ids and self._last[1] == ids._version:
      known = self._last[2:]      # same id set as the forward pass: same bucket sizes, no size exchange
    rt = route(uniq, self.group, self.bucket_fn, known_counts=known)
    served = exchange(rt, uniq, group=self.group, presorted=rt.bucketed_ids, take_fn=self.take_fn)
    g = exchange(rt, summed, group=self.group, take_fn=self.take_fn)
    apply_fn(self.shard, g, served)
