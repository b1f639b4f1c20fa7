"""An in-loop speed of light for the apply: tools/apply_sol.hip's kernel — a FREE batch index, every
gradient row streamed once into its key's sum, every key's state read and written once — run inside the training loop,
right behind the product's k_ltile, on stand-in state arrays of the table's own footprint (var slab, slot slab, both
record arrays; row of a key = its Zipf rank, as in bench.py's table).  Next to it, the same loop with the product's apply.
  python tools/apply_sol.py [keys]      -> profiles/r06_apply_sol.txt
"""
import ctypes, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from tfplus_amd import _lib
from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as ops

so = os.path.join(ROOT, "build", "apply_sol.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so,
                       os.path.join(ROOT, "tools", "apply_sol.hip")])
S = ctypes.CDLL(so)
dev = torch.device("cuda", 0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
N, D, HOTMIN, LEAF = 1_000_000, 32, 16, 256
gen = torch.Generator(device=dev).manual_seed(bench.SEED)
L = _lib.lib()

# ---- the product's table (the lookup in front of the calibrated kernel is the real k_ltile) -------------------------
var = ops.kv_variable([D], capacity_hint=K + 24 * N)
slot = ops.kv_variable([3 * D], capacity_hint=K + 24 * N)
ops.init_kv_variable_v2(var, torch.randn(10000, D, device=dev) * 0.05)
ops.init_kv_variable_v2(slot, torch.zeros(16, 3 * D, device=dev))
buf = torch.empty((1 << 22, 3 * D), device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for i in range(0, K, 1 << 22):
  keys = bench.splitmix64(torch.arange(i + 1, min(i + (1 << 22), K) + 1, device=dev))
  _lib.check(L.kv_gather_or_insert(var.ptr, keys.data_ptr(), None, keys.numel(), buf.data_ptr(), st))
  _lib.check(L.kv_gather_or_insert(slot.ptr, keys.data_ptr(), None, keys.numel(), buf.data_ptr(), st))
torch.cuda.synchronize()
del buf
ops.kv_attach_slot(var, slot)

# ---- stand-in state of the same footprint ----------------------------------------------------------------------------
R = K + 1
vrows = torch.full((R, D), 0.01, device=dev)
srows = torch.zeros((R, 3 * D), device=dev)
vmeta = torch.zeros((R, 4), dtype=torch.int32, device=dev)
smeta = torch.zeros((R, 4), dtype=torch.int32, device=dev)

# ---- batches and their free index --------------------------------------------------------------------------------------
z = bench.Zipf(K, 1.2, dev)
pool = []
for _ in range(8):
  ranks = z.sample(N, gen)
  ids = bench.splitmix64(ranks)
  grad = torch.randn(N, D, device=dev, generator=gen) * 1e-2
  u, inv, cnt = torch.unique(ranks, return_inverse=True, return_counts=True)
  kpos = torch.argsort(inv, stable=True).to(torch.int32)
  start = (torch.cumsum(cnt, 0) - cnt).to(torch.int64)
  hotm = cnt > HOTMIN
  # cold keys, most positions first (the lane groups of a wave get keys of equal length)
  ck = torch.nonzero(~hotm).flatten()
  ck = ck[torch.argsort(cnt[ck], descending=True, stable=True)]
  cold = torch.stack([start[ck], cnt[ck], u[ck], kpos[start[ck]].to(torch.int64)], 1).to(torch.int32).contiguous()
  # hot keys: leaves of LEAF positions, groups of at most 64 siblings, at most two levels (tools/apply_sol.hip)
  hk = torch.nonzero(hotm).flatten()
  h_start, h_cnt, h_row = start[hk].tolist(), cnt[hk].tolist(), u[hk].tolist()
  NONE = 0xFFFFFFFF
  leaves, groups, nslots = [], [], 0
  for s0, c0, r0 in zip(h_start, h_cnt, h_row):
    nl = (c0 + LEAF - 1) // LEAF
    if nl == 1:
      groups.append((0, 1, NONE, r0)); leaves.append((s0, c0, len(groups) - 1, 0)); continue
    if nl <= 64:
      groups.append((nslots, nl, NONE, r0)); gi = len(groups) - 1
      for j in range(nl):
        leaves.append((s0 + j * LEAF, min(LEAF, c0 - j * LEAF), gi, nslots + j))
      nslots += nl
      continue
    n1 = (nl + 63) // 64
    groups.append((nslots, n1, NONE, r0)); top = len(groups) - 1
    top_first = nslots; nslots += n1
    for q in range(n1):
      ns = min(64, nl - q * 64)
      groups.append((nslots, ns, top, top_first + q)); gq = len(groups) - 1
      for j in range(ns):
        jj = q * 64 + j
        leaves.append((s0 + jj * LEAF, min(LEAF, c0 - jj * LEAF), gq, nslots + j))
      nslots += ns
  # the longest leaves first (a wave's items are dealt round-robin)
  leaves.sort(key=lambda t: -t[1])
  i32 = lambda rows: torch.tensor(rows, dtype=torch.int64, device=dev).to(torch.int32).contiguous()
  pool.append(dict(ids=ids, grad=grad, kpos=kpos.contiguous(), cold=cold, leaf=i32(leaves), group=i32(groups), nslots=nslots,
                   U=int(u.numel()), nhotkeys=int(hk.numel()), hotpos=int(cnt[hk].sum()), u=u, inv=inv))
hpart = torch.zeros((max(p["nslots"] for p in pool) + 64, D), device=dev)
gcnt = torch.zeros(max(p["group"].shape[0] for p in pool) + 1, dtype=torch.int32, device=dev)
out = torch.empty((N, D), device=dev)
U = sum(p["U"] for p in pool) / len(pool)
print("K = %d, N = %d, D = %d: %.0f distinct keys per batch, %d hot keys (> %d positions) holding %d positions in %d leaves of <= %d, %d tree groups"
      % (K, N, D, U, pool[0]["nhotkeys"], HOTMIN, pool[0]["hotpos"], pool[0]["leaf"].shape[0], LEAF, pool[0]["group"].shape[0]))


def sol(p, grid, rb, state, b1p=0.5, b2p=0.9):
  rc = S.sol_apply(ctypes.c_void_p(p["grad"].data_ptr()), ctypes.c_void_p(p["kpos"].data_ptr()), ctypes.c_void_p(p["leaf"].data_ptr()),
                   p["leaf"].shape[0], ctypes.c_void_p(p["group"].data_ptr()), ctypes.c_void_p(p["cold"].data_ptr()), p["cold"].shape[0],
                   ctypes.c_void_p(vrows.data_ptr()), ctypes.c_void_p(srows.data_ptr()), ctypes.c_void_p(vmeta.data_ptr()),
                   ctypes.c_void_p(smeta.data_ptr()), ctypes.c_void_p(hpart.data_ptr()), ctypes.c_void_p(gcnt.data_ptr()), ctypes.c_float(1e-3), ctypes.c_float(b1p),
                   ctypes.c_float(b2p), 20000, grid, rb, state, ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
  assert rc == 0, rc


# ---- the kernel computes what it claims: m after one step on zero state = 0.1 * the key's summed gradient ------------------
p = pool[0]
sol(p, 2048, 4, 1)
torch.cuda.synchronize()
want = torch.zeros((p["U"], D), device=dev).index_add_(0, p["inv"], p["grad"]) * 0.1
got = srows[p["u"], :D]
err = ((got - want).abs().max() / want.abs().max()).item()
print("check: slot m after one step against 0.1 * index_add of the gradient: max error %.2e of the largest element" % err)
assert err < 1e-4, err
assert int(gcnt.sum()) == 0
srows.zero_(); vrows.fill_(0.01)


def ev():
  return torch.cuda.Event(enable_timing=True)


def lookup(p):
  tok = ctypes.c_uint64(0)
  _lib.check(L.kv_gather_or_insert_tok(var.ptr, p["ids"].data_ptr(), None, N, out.data_ptr(), ctypes.byref(tok), st))
  return tok.value


def loop(kind, steps=40, **kw):
  """steps of { product token lookup (k_ltile; the pass the previous lookup left pending is settled in front of it when no
  product apply took it over) ; apply }: events around the apply only"""
  t = 0.0
  for s in range(steps + 6):
    p = pool[s % len(pool)]
    tok = lookup(p)
    a, b = ev(), ev()
    a.record()
    if kind == "product":
      _lib.check(L.kv_apply_group_adam_tok(var.ptr, slot.ptr, p["grad"].data_ptr(), p["ids"].data_ptr(), N, 1e-3, 0.5, 0.9, 0.9, 0.999,
                                           1e-8, 0.0, 0.0, 0.0, 4, tok, st))
    else:
      sol(p, **kw)
    b.record()
    torch.cuda.synchronize()
    if s >= 6:
      t += a.elapsed_time(b)
  return t / steps * 1e3


alg = N * (8 + 4 * D) + U * (16 + 4 * 4 * D) + U * 4 * 4 * D
print("apply, events around it, inside { token lookup ; apply } steps (a synchronisation per step); algorithmic bytes %.1f MB" % (alg / 1e6))
us = loop("product")
print("  product apply (k_tsum + k_papply, batch token)                %6.1f us   %.3f of 8 TB/s" % (us, alg / us / 8e6))
best = None
for grid in (1024, 2048, 4096):
  for rb in (2, 4, 8):
    us = loop("sol", grid=grid, rb=rb, state=1)
    print("  speed of light: free index, grid %4d x 256, %d rows in flight  %6.1f us   %.3f of 8 TB/s" % (grid, rb, us, alg / us / 8e6))
    best = min(best or us, us)
for grid in (2048,):
  for rb in (4, 8):
    us = loop("sol", grid=grid, rb=rb, state=0)
    print("  the gradient stream alone (sums dropped), grid %4d, %d in flight %6.1f us   (%.0f MB at %.2f TB/s)"
          % (grid, rb, us, N * 4 * D / 1e6, N * 4 * D / us / 1e6))
for grid in (2048, 4096):
  us = loop("sol", grid=grid, rb=4, state=3)
  print("  the state read-modify-write alone (no gradient row read), grid %4d  %6.1f us   (%.0f MB at %.2f TB/s)"
        % (grid, us, U * (32 + 2 * 4 * 4 * D) / 1e6, U * (32 + 2 * 4 * 4 * D) / us / 1e6))
print("speed of light of the apply inside the loop: %.1f us" % best)
