// kvhip.hip — MI355X (gfx950) KvVariable: HBM hash table + row slab, lookup and fused
// sparse optimizer kernels, and the C ABI of include/kvhip.h.
//
// Layout in HBM (per table):
//   index    Entry[cap+1]   16 B {int64 key, u32 row, u32 slot-row hint}, open addressing, linear
//                           probing, cap = 2^k >= 2 * rows (load <= 0.5).  Entry[cap] is the
//                           home of the one key that equals the EMPTY sentinel.
//   chunks   row slab in chunks of 2^cb rows: rows[r][dim] fp32 and one 32-byte record unit per row: RowMeta 16 B
//            {int64 key, u32 freq = (day << 16) | saturating u16 frequency, u8 flags (bit0 blacklist, bit1
//            under_threshold, bit2 under_threshold stale, bit3 released by Delete), u8 delta-list bits,
//            u16 stamp (serial of the last unique-ids apply that updated the row)} + SlotMirror 16 B (var tables of
//            a (var, slot) pair: a write-back copy of the slot row's frequency word and flags; mirror_* below).  Row ids are
//            dense (bump allocated; rows released by Delete are recycled from a device free
//            list), row 0 is a permanent all-zero row (misses / nothing).
//   workspace per-batch index (ent_key / ent_a / ent_b / ent_base / ent_rec, toff, mrow, epart, order; the
//            sorted-position kernels' lists in the same buffers): plain stores only, rewritten by every
//            op — nothing to clean.
//
// Kernel pipelines (DESIGN.md section 3 has the byte accounting):
//   entry-list kernels (kv_fused.h, kv_papply.h, kv_uapply.h; dims that are multiples of 4 up to 256):
//     lookup : k_ltile            tile pass (LDS dedup, winner probes / inserts, entries by hash partition) + output rows
//              k_part2            the lookup's bookkeeping alone (deferred when a batch token is handed out)
//     apply  : k_tsum + k_papply  tile sums, then partition pass + bookkeeping + fused update in one launch
//              k_ltsum + k_papply the same for ids the table has not indexed (no token)
//              k_uapply           ids promised unique + pre-summed rows: one launch
//   sorted-position kernels (kv_kernels.h; every other dim, kv_unique / dedup / scatter / marks / sparse lookup):
//     k_tile, k_part_keys<MODE>, k_gather<ORDER> / k_order, k_apply<OPT>, k_apply_fin<OPT>
//   sharded: kv_comm_* (RCCL by dlopen, grouped send / recv), kv_shard_* (route / serve / finish phases)
//   many tables in one launch: the *_multi entry points (grid.y = table)
//
// Reference semantics restated per function with file:line (relative to the tfplus tree).

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iterator>
#include <ctime>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>
#include <unordered_set>

#include "../../include/kvhip.h"

// k_apply_sorted / k_apply_span dispatch, compiled in two other translation units (kv_apply_launch.h)
extern "C" int kvp_launch_apply_a(int mode, int opt, const void* wd, const void* pa, void* stream, const void* md, int ntab,
                                  unsigned nchunks, int span);
extern "C" int kvp_launch_apply_b(int mode, int opt, const void* wd, const void* pa, void* stream, const void* md, int ntab,
                                  unsigned nchunks, int span);

extern "C" int kvp_launch_tsum(const void* td, const void* wd, const float* grad, void* stream, const void* md, int ntab);
// k_papply dispatch (kv_papply.h), two more translation units: a = GroupAdam V4 / V3, b = Adagrad / SparseGroupFtrl
// k_ltsum dispatch (kv_fused.h: tile pass + tile sums), instantiated next to k_tsum; ids_kind 0 int64, 1 int32
extern "C" int kvp_launch_ltsum(const void* td, const void* wd, const void* ids, int ids_kind, long long n, int det,
                                const float* grad, void* stream);
extern "C" int kvp_launch_papply_a(int opt, const void* wd, const void* pa, int mode, void* stream, const void* md = nullptr, int ntab = 0);
extern "C" int kvp_launch_papply_b(int opt, const void* wd, const void* pa, int mode, void* stream, const void* md = nullptr, int ntab = 0);
extern "C" int kvp_launch_papply_ud(const void* wd, const void* pa, int mode, void* stream, const void* md = nullptr, int ntab = 0);   // PA_UNIQUE / PA_DEDUP
// k_uapply dispatch (kv_uapply.h: the apply on unique ids), instantiated next to k_papply
extern "C" int kvp_launch_uapply_a(int opt, const void* pa, const void* ids, int ids32, long long n, void* stream, const void* md = nullptr,
                                   int ntab = 0);
extern "C" int kvp_launch_uapply_b(int opt, const void* pa, const void* ids, int ids32, long long n, void* stream, const void* md = nullptr,
                                   int ntab = 0);

namespace {

#include "kv_device.h"
#include "kv_kernels.h"
#include "kv_fused.h"
#include "kv_papply.h"

// ------------------------------------------------------------------------------------------
// maintenance kernels
// ------------------------------------------------------------------------------------------
__global__ void k_fill_entries(Entry* e, unsigned long long count) {
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (unsigned long long)gridDim.x * blockDim.x) {
    Entry v; v.key = EMPTY_KEY; v.row = 0; v.hint = 0;
    *reinterpret_cast<uint4*>(&e[i]) = *reinterpret_cast<uint4*>(&v);
  }
}
__global__ void k_fill_i64(long long* p, long long v, unsigned long long count) {
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (unsigned long long)gridDim.x * blockDim.x) p[i] = v;
}
// int64 list -> int32 list in place (one block: element i is read before any thread can overwrite
// it, because writes land at half the byte offset and the loop is barrier-stepped)
__global__ void k_narrow_keys(long long* keys, long long n) {
  int* out = reinterpret_cast<int*>(keys);
  for (long long base = 0; base < n; base += blockDim.x) {
    const long long i = base + threadIdx.x;
    const long long v = i < n ? keys[i] : 0;
    __syncthreads();
    if (i < n) out[i] = (int)v;
    __syncthreads();
  }
}

// re-insert rows [1, next_row) into a fresh index; the slot-row hints travel from the old index (told)
__global__ void k_rehash(TableDev t, TableDev told, unsigned nrows) {
  for (unsigned r = 1 + blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += gridDim.x * blockDim.x) {
    if (*flags_ptr(t, r) & FLAG_FREE) continue;  // released by Delete: no index entry
    const long long key = *key_ptr(t, r);
    unsigned hint = 0;
    if (told.entries) { const Entry* oe = table_entry_of(told, key); if (oe) hint = oe->hint; }
    if (key == EMPTY_KEY) {
      Entry* s = &t.entries[t.mask + 1];
      s->key = 0; s->row = r; s->hint = hint;
      continue;
    }
    unsigned long long p = mix64((unsigned long long)key) & t.mask;
    for (;;) {
      unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&t.entries[p].key),
                                         (unsigned long long)EMPTY_KEY, (unsigned long long)key);
      if (old == (unsigned long long)EMPTY_KEY) { t.entries[p].row = r; t.entries[p].hint = hint; break; }
      p = (p + 1) & t.mask;
    }
  }
}
// kv_uapply.h: the 16-bit launch serial wrapped — every row's stamp back to "none"
__global__ void k_clear_stamps(TableDev t, unsigned nrows) {
  for (unsigned r = blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += gridDim.x * blockDim.x) meta_ptr(t, r)->stamp = 0;
}
// forget every slot-row hint (the attached slot table changed or was cleared)
__global__ void k_clear_hints(Entry* e, unsigned long long count) {
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (unsigned long long)gridDim.x * blockDim.x) e[i].hint = 0;
}
// kv_attach_slot: every key of the var learns its row in the slot table (one pass over the var's rows) and, mirrors != 0,
// takes a clean copy of that row's frequency word and flags into its own record line (kv_device.h SlotMirror)
__global__ void k_link_hints(TableDev tv, TableDev ts, unsigned nrows, unsigned epoch, int mirrors) {
  for (unsigned r = 1 + blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += gridDim.x * blockDim.x) {
    if (mirrors) mirror_ptr(tv, r)->state = (unsigned char)MIRROR_INVALID;
    if (*flags_ptr(tv, r) & FLAG_FREE) continue;
    const long long key = *key_ptr(tv, r);
    const unsigned sr = table_find(ts, key);
    if (!sr) continue;
    Entry* e = table_entry_of(tv, key);
    if (e && load_entry(e).row == r) {
      e->hint = sr;
      if (mirrors) {
        const uint2 sm = load_freq_flags(ts, sr);
        SlotMirror nm;
        nm.srow = sr; nm.freq = sm.x; nm.flags = (unsigned char)(sm.y & 0xFFu); nm.state = (unsigned char)MIRROR_CLEAN;
        nm.epoch = (unsigned short)epoch; nm.pad = 0u;
        *mirror_ptr(tv, r) = nm;
      }
    }
  }
}
// the end of a mirror epoch: what the lean applies wrote into the var rows' mirrors goes back into the slot table's own
// records (a dirty mirror of THIS epoch names a live slot row: nothing else has touched the slot table since it was made)
__global__ void k_flush_mirrors(TableDev tv, TableDev ts, unsigned epoch) {
  const unsigned nrows = min(tv.counters[0], tv.max_rows);
  for (unsigned r = 1 + blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += gridDim.x * blockDim.x) {
    SlotMirror* mp = mirror_ptr(tv, r);
    const SlotMirror m = *mp;
    if (m.state != MIRROR_DIRTY || m.epoch != (unsigned short)epoch || m.srow == 0u || m.srow >= ts.max_rows) continue;
    RowMeta* sm = meta_ptr(ts, m.srow);
    sm->freq = m.freq;
    sm->flags = m.flags;
    mp->state = (unsigned char)MIRROR_CLEAN;
  }
}
// the 16-bit epoch wrapped: every mirror back to "none"
__global__ void k_clear_mirrors(TableDev tv) {
  const unsigned nrows = min(tv.counters[0], tv.max_rows);
  for (unsigned r = blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += gridDim.x * blockDim.x)
    mirror_ptr(tv, r)->state = (unsigned char)MIRROR_INVALID;
}

// size() / sum_freq() kv_variable.h:139-175 ; out[0] = size, out[1] = sum_freq
__global__ void k_stats(TableDev t, unsigned nrows, unsigned long long* out) {
  unsigned long long c = 0, f = 0;
  for (unsigned r = 1 + blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += gridDim.x * blockDim.x) {
    const unsigned fl = *flags_ptr(t, r);
    const unsigned fr = *freq_ptr(t, r) & 0xFFFFu;
    if (!(fl & (FLAG_BLACK | FLAG_FREE)) && fr >= t.enter_threshold) { c += 1; f += fr; }
  }
  for (int o = 32; o > 0; o >>= 1) { c += __shfl_xor(c, o); f += __shfl_xor(f, o); }
  if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], c); atomicAdd(&out[1], f); }
}

template <typename IdT>
__global__ void k_get_meta(TableDev t, const IdT* ids, long long n, unsigned* fw, unsigned char* fl) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const unsigned r = table_find(t, load_id(ids, (size_t)i));
    fw[i] = r ? *freq_ptr(t, r) : 0u;
    fl[i] = r ? (unsigned char)(*flags_ptr(t, r) | 0x80u) : 0;
  }
}

// GetCount kv_variable.h:503-524 (absent -> 0, else the low 16 bits) and GetTimeStamp :526-561
// (absent -> today, else the high 16 bits = day stamp of the last training lookup)
template <typename IdT>
__global__ void k_get_count_ts(TableDev t, const IdT* ids, long long n, int what, unsigned today, unsigned* out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const unsigned r = table_find(t, load_id(ids, (size_t)i));
    const unsigned fw = r ? *freq_ptr(t, r) : 0u;
    out[i] = what == 0 ? (r ? (fw & 0xFFFFu) : 0u) : (r ? (fw >> 16) : today);
  }
}

// releases one key: its index entry becomes a tombstone, its row goes to the free list
// (TableManager::DeleteKey table_manager.h:405-416: Evict + erase).  A key listed twice is
// released once (the second probe finds the tombstone).
__device__ __forceinline__ bool release_key(const TableDev& t, long long key, unsigned* free_rows) {
  Entry* slot;
  if (key == EMPTY_KEY) {
    slot = &t.entries[t.mask + 1];
    if (load_entry(slot).key != 0) return false;
  } else {
    unsigned long long p = mix64((unsigned long long)key) & t.mask;
    for (;;) {
      slot = &t.entries[p];
      const Entry e = load_entry(slot);
      if (e.key == key) break;
      if (e.key == EMPTY_KEY) return false;
      p = (p + 1) & t.mask;
    }
  }
  const unsigned r = atomicExch(&slot->row, ROW_TOMB);   // duplicates of the key race here: one wins
  if (r == ROW_TOMB || r == 0u) return false;
  *flags_ptr(t, r) = (unsigned char)FLAG_FREE;
  free_rows[atomicAdd(&t.counters[2], 1u)] = r;
  return true;
}
template <typename IdT>
__global__ void k_delete(TableDev t, const IdT* ids, long long n, unsigned* free_rows, unsigned long long* cnt) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    if (release_key(t, load_id(ids, (size_t)i), free_rows)) atomicAdd(&cnt[0], 1ull);
}
// DeleteWithTimestamp kv_variable.h:757-789: keys whose day stamp is > 0 and at least `threshold`
// days old.  fill == 0 only counts; fill == 1 releases them and lists their keys.
__global__ void k_delete_by_time(TableDev t, unsigned nrows, unsigned today, unsigned threshold, int fill,
                                 unsigned* free_rows, unsigned long long* cnt, long long* out_keys) {
  for (unsigned r = 1 + blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += gridDim.x * blockDim.x) {
    if (*flags_ptr(t, r) & FLAG_FREE) continue;
    const unsigned kt = *freq_ptr(t, r) >> 16;
    if (kt == 0 || (int)today - (int)kt < (int)threshold) continue;
    const long long key = *key_ptr(t, r);
    if (!fill) { atomicAdd(&cnt[0], 1ull); continue; }
    if (release_key(t, key, free_rows)) out_keys[atomicAdd(&cnt[0], 1ull)] = key;
  }
}

// ExportValues dynamic_save.hpp:47-195.  cnt[0..2] = rows, blacklist, freq.  fill != 0 writes.
__global__ void k_export(TableDev t, unsigned nrows, int first_n, int fill, unsigned long long* cnt,
                         long long* keys, float* values, long long* blacklist, long long* fkeys,
                         unsigned* fvals) {
  const int D = t.dim;
  for (unsigned r = 1 + blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += gridDim.x * blockDim.x) {
    const unsigned fl = *flags_ptr(t, r);
    const unsigned fw = *freq_ptr(t, r);
    const long long key = *key_ptr(t, r);
    if (fl & FLAG_FREE) continue;
    if (fl & FLAG_BLACK) {
      if (first_n > 3) {
        unsigned long long p = atomicAdd(&cnt[1], 1ull);
        if (fill && blacklist) blacklist[p] = key;
      }
    } else if ((first_n <= 3 || (fw & 0xFFFFu) >= t.enter_threshold) && !(fl & FLAG_UNDER)) {
      unsigned long long p = atomicAdd(&cnt[0], 1ull);
      if (fill) {
        keys[p] = key;
        const float* row = row_ptr(t, r);
        for (int e = 0; e < D; ++e) values[p * D + e] = row[e];
      }
    }
    if (first_n > 4) {
      unsigned long long p = atomicAdd(&cnt[2], 1ull);
      if (fill && fkeys) { fkeys[p] = key; fvals[p] = fw; }
    }
  }
}

// DeltaExport dynamic_save.hpp:198-451 over the rows whose delta bytes are set (train list, plus the
// prediction list when first_n <= 3).  cnt[0] = update rows, [1] = blacklisted keys, [2] = all delta rows.
// Order per key as in :231-248: low frequency -> only in the frequency list; blacklisted -> black list
// (the caller hands the delete list as `black` when first_n <= 3, :345-351); else key + row.
__global__ void k_export_delta(TableDev t, unsigned nrows, int first_n, int fill, unsigned long long* cnt,
                               long long* keys, float* values, long long* black, long long* fkeys,
                               unsigned* fvals) {
  const int D = t.dim;
  for (unsigned r = 1 + blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += gridDim.x * blockDim.x) {
    const RowMeta m = *meta_ptr(t, r);
    if (m.flags & FLAG_FREE) continue;
    if (!((m.delta & DELTA_TRAIN) || (first_n <= 3 && (m.delta & DELTA_PRED)))) continue;
    if (first_n > 4) {  // ExportFrequencyDelta kv_variable.h:937-957: the whole 32-bit word
      unsigned long long p = atomicAdd(&cnt[2], 1ull);
      if (fill && fkeys) { fkeys[p] = m.key; fvals[p] = m.freq; }
    }
    if ((m.freq & 0xFFFFu) < t.enter_threshold) continue;
    if (m.flags & FLAG_BLACK) {
      unsigned long long p = atomicAdd(&cnt[1], 1ull);
      if (fill && black) black[p] = m.key;
      continue;
    }
    unsigned long long p = atomicAdd(&cnt[0], 1ull);
    if (fill) {
      keys[p] = m.key;
      const float* row = row_ptr(t, r);
      for (int e = 0; e < D; ++e) values[p * D + e] = row[e];
    }
  }
}
// keys recorded by Delete: one that has a row again is a live member of the list (its row carries the
// byte from here on), one without stays a "deleted" member.  which: 0 train list, 1 prediction list
__global__ void k_delta_resolve(TableDev t, const long long* keys, long long n, int which, unsigned char* present) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const unsigned r = table_find(t, keys[i]);
    present[i] = r ? 1 : 0;
    if (r) meta_ptr(t, r)->delta |= (unsigned char)(which == 0 ? DELTA_TRAIN : DELTA_PRED);
  }
}
// end of an export (dynamic_save.hpp:179-192, 432-443).  mode 0 (training export): the train list moves
// to the prediction list (if kept) and empties; mode 1 (prediction export): the prediction list empties
__global__ void k_delta_clear(TableDev t, unsigned nrows, int mode, int keep_pred) {
  for (unsigned r = 1 + blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += gridDim.x * blockDim.x) {
    RowMeta* m = meta_ptr(t, r);
    const unsigned d = m->delta;
    if (mode == 1) { if (d & DELTA_PRED) m->delta = (unsigned char)(d & ~DELTA_PRED); continue; }
    if (d & DELTA_TRAIN) m->delta = (unsigned char)((d & ~DELTA_TRAIN) | (keep_pred ? DELTA_PRED : 0u));
  }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      return fail(_e == hipErrorOutOfMemory ? KV_RESOURCE_EXHAUSTED : KV_INTERNAL,            \
                  "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

struct Workspace {
  long long cap_n = 0;       // ids (multiple of TILE)
  unsigned capP = 0;         // partitions toff was sized for
  long long* ent_key = nullptr;
  unsigned* ent_a = nullptr;
  unsigned* ent_b = nullptr;
  unsigned* ent_base = nullptr;
  unsigned* ent_rec = nullptr;
  unsigned* toff = nullptr;
  unsigned* slot_rank = nullptr;
  unsigned* order = nullptr;
  uint4* coldlist = nullptr;   // [cap_n][2]
  uint4* hotlist = nullptr;    // [cap_n][2]
  uint4* litem = nullptr;      // [cap_n]
  uint4* items = nullptr;      // [cap_n]
  uint4* pmeta = nullptr;      // [capP]
  float* hpart = nullptr;      // [chunk_cap(cap_n)][dim]
  long long hpart_elems = 0;
  unsigned* ctr = nullptr;
  unsigned* mcount = nullptr;  // entry-list pipeline: [cap_n / TILE]
  unsigned short* pos_ent = nullptr;   // [pos_cap] sharded route: every position's entry number in its tile
  long long pos_cap = 0;
  float* epart = nullptr;      // [cap_n / 2][dim] tile sums
  long long epart_elems = 0;
  long long* scat_keys = nullptr;  // kv_scatter_update on repeated ids: de-duplicated ids and combined updates
  float* scat_sum = nullptr;
  long long scat_cap = 0;          // rows
  unsigned* seg_off = nullptr;   // kv_lookup_sparse: CSR offsets [seg_cap + 1]
  long long seg_cap = 0;
  unsigned long long* dbg = nullptr;
};

}  // namespace

struct kv_table {
  // every op that may change which rows exist (or what the delta lists hold) advances op_serial; the two-phase
  // calls (count, then fill into buffers the caller sized from the counts) refuse to fill once it has moved on
  uint64_t op_serial = 1, export_serial = 0, delta_serial = 0, expire_serial = 0;
  int device = 0;
  int key_dtype = KV_DT_INT64;
  int dim = 0;
  unsigned enter_threshold = 0;
  unsigned long long seed = 0;
  int fixed_day = -1;
  // index
  Entry* entries = nullptr;
  unsigned long long cap = 0;
  // slab
  int chunk_bits = 16;
  std::vector<Chunk> chunks;
  Chunk* d_chunks = nullptr;
  unsigned* d_counters = nullptr;  // [0] next_row [1] error [2] rows on the free list
  unsigned* free_rows = nullptr;   // rows released by Delete (device stack, rows_cap entries)
  unsigned long long free_cap = 0;
  long long free_known = 0;        // free-list length at the last sync (> 0: inserts pop from it)
  unsigned long long idx_ub = 0;   // upper bound of claimed index entries (live keys + tombstones)
  // exact claimed entries at a sync = idx_base + (next_row - bump_base) + free-list pops since the
  // last index rebuild, pops = pushes_since - (free_now - free_base)   (revivals make it an upper bound)
  unsigned long long idx_base = 0, bump_base = 1, pushes_since = 0;
  long long free_base = 0;
  unsigned long long rows_cap = 0;  // chunks.size() << chunk_bits
  unsigned long long rows_ub = 1;   // upper bound of next_row
  // init table
  float* init_table = nullptr;
  long long init_rows = 0;
  bool initialized = false;
  bool init_placeholder = false;   // init_table is the zero row an import put there, not a real init table
  Workspace ws;
  // the batch index the workspace holds: `batch_serial` names it (0 = none); an optimizer apply handed the
  // same token takes the index over instead of rebuilding it
  uint64_t batch_serial = 0;
  long long batch_n = 0;
  bool fused_index = false;        // the index is the tiles' entries (kv_fused.h: an apply of that batch goes through k_papply),
                                   // not a sorted position list (kv_kernels.h)
  long long batch_n_prev = 0;      // ids of the previous entry-list index pass (the distinct-count hint belongs to that size)
  unsigned index_P = 0;            // partitions of the entry-list index the workspace holds
  // A training lookup that hands out a batch token returns when its rows are written; its partition pass (frequency
  // words, rows of new keys, the batch's key records and entry list) is PENDING: the optimizer apply of that batch
  // runs it in front of its own kernels, any other op on the table runs it first thing (settle).  Same stream order
  // as before, the rows just do not wait for it.
  bool part_pending = false;
  unsigned char pend_wd[sizeof(WsDev)], pend_pa[sizeof(PartArgs)];
  // Slot mirrors (kv_device.h SlotMirror; mirror_* below): a var table paired with ONE slot table keeps, next to each row's
  // record, a write-back copy of the slot row's frequency word and flags; the lean apply works on the copy alone.
  long long stat_mirror_applies = 0;            // kv_get_stat
  std::atomic<long long> stat_mirror_epochs{0}; // ... (an epoch of a var's mirrors may be ended under the slot table's lock)
  // both tables of a mirror pair hold the pair's device views as the last lean apply (or the pairing) saw them — written with
  // BOTH locks held.  An op that ends the epoch holds ONE of the two locks: it flushes through the copy in the table it
  // holds and never reads the other table's host state (whose owner may be growing it on another thread).  What a dirty
  // copy names — a var row and a slot row of the chunk-0 slabs — is inside these views whatever happened to the tables since.
  TableDev mview_var{}, mview_slot{};
  kv_table* mirror_slot = nullptr;          // var side: the slot table its mirrors stand for
  std::atomic<unsigned> mirror_epoch{1};    // var side: generation of the copies (16 bits on the device)
  std::atomic<bool> mirror_dirty{false};    // var side: a lean apply has written mirrors since the last flush
  kv_table* mirror_var = nullptr;           // slot side: the var that holds this table's mirrors
  bool mirror_banned = false;               // either side: the table is used under stream capture (kv_prepare_capture): no mirrors, ever
  bool mirror_shared = false;               // slot side: a second var attached it — no mirrors for this table any more
  unsigned uniq_serial = 0;        // stamp of the table's last kv_apply_*_unique launch (kv_uapply.h; wraps at 65535: stamps cleared)
  bool deterministic = false;      // kv_set_deterministic
  bool occurrence_order = false;   // kv_set_deterministic(h, 2): a repeated id's gradient rows are added one by one in input order
                                   // (the sorted-position pipeline with one chain per key; implies deterministic)
  std::atomic<int> shard_refs{0};  // kv_shard handles built on this table
  bool fast_math = false;          // kv_set_fast_math: the optimizers' sqrt / division on v_sqrt_f32 / v_rcp_f32 (1 ulp) —
                                   // never in deterministic mode, which keeps the IEEE sequences
  uint64_t uid = 0;                // unique over the process: names the attached slot table safely
  uint64_t slot_uid = 0;           // uid of the slot table the index entries' hints refer to (0 = none)
  uint64_t slot_gen = 0;           // that table's `gen` when the hints were valid
  uint64_t gen = 0;                // bumped when the table is cleared (import): hints into it die
  unsigned* err_host = nullptr;    // pinned: the device error flag, copied back after every batch op
  unsigned* cnt_host = nullptr;    // pinned: where the synchronous ops (kv_dedup_segment_sum, kv_unique) read their count back
  hipStream_t last_stream = nullptr;  // stream of the table's last op; a different stream first waits for it
  bool has_last = false;
  hipEvent_t last_done = nullptr;
  // delta lists (SUPPORT_DELTA_EXPORT / SUPPORT_PREDICTION_DELTA_EXPORT, kv_variable.h:100-111): live keys
  // carry a byte in their RowMeta; keys recorded by Delete have no row and wait here
  bool track_delta = false, track_pred = false;
  std::vector<long long> del_train, del_pred;
  unsigned long long* d_stat = nullptr;  // [4]
  std::mutex mu;
  unsigned* route_hist = nullptr;  // kv_bucket_by_owner scratch
  size_t route_hist_cap = 0;
  // optional per-kernel timing (kv_profile_*): event pairs recorded on the op's stream
  bool prof = false;
  unsigned prof_mask = 0xFFFFFFFFu;
  std::vector<hipEvent_t> ev;
  std::vector<int> ev_kind;
  size_t ev_used = 0;
  int prof_every = 1;                // bracket every prof_every-th launch of a kind (kv_profile_sample)
  unsigned prof_seq[KV_PROF_KINDS] = {};
};

namespace {

unsigned long long pow2ceil(unsigned long long x) {
  unsigned long long p = 1;
  while (p < x) p <<= 1;
  return p;
}
int ilog2(unsigned long long x) {
  int l = 0;
  while ((1ull << l) < x) ++l;
  return l;
}
int nblocks(long long work, int per_block, int cap = 4096) {
  long long b = (work + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

struct DeviceGuard {
  int prev = 0;
  explicit DeviceGuard(int d) { hipGetDevice(&prev); if (prev != d) hipSetDevice(d); cur = d; }
  ~DeviceGuard() { if (prev != cur) hipSetDevice(prev); }
  int cur;
};

TableDev dev_view(const kv_table* t) {
  TableDev d;
  d.c0 = t->chunks.empty() ? Chunk{} : t->chunks[0];
  d.entries = t->entries;
  d.mask = t->cap - 1;
  d.chunks = t->d_chunks;
  d.chunk_bits = t->chunk_bits;
  d.counters = t->d_counters;
  d.free_rows = t->free_known > 0 ? t->free_rows : nullptr;
  d.max_rows = (unsigned)std::min<unsigned long long>(t->rows_cap, 0x7FFFFFFFull);
  d.init_table = t->init_table;
  d.init_rows = (unsigned)t->init_rows;
  d.dim = t->dim;
  d.enter_threshold = t->enter_threshold;
  d.seed = t->seed;
  d.track_delta = t->track_delta ? 1u : 0u;
  d.err_host = t->err_host;   // hipHostMallocMapped: the same address on the device
  d.single = t->chunks.size() <= 1 ? 1u : 0u;
  return d;
}

int add_chunk(kv_table* t, hipStream_t s) {
  if (t->chunks.size() >= (size_t)MAX_CHUNKS) return fail(KV_RESOURCE_EXHAUSTED, "row slab: too many chunks");
  const size_t R = (size_t)1 << t->chunk_bits;
  Chunk c{};
  HIP_TRY(hipMalloc(&c.rows, R * t->dim * sizeof(float)));
  HIP_TRY(hipMalloc(&c.meta, R * META_STRIDE * sizeof(RowMeta)));   // (record + slot mirror per row: kv_device.h)
  // every mirror unit starts INVALID (state 0): k_flush_mirrors walks all rows below next_row, also those no apply has met
  HIP_TRY(hipMemsetAsync(c.meta, 0, R * META_STRIDE * sizeof(RowMeta), s));
  if (t->chunks.empty()) {
    // row 0: the permanent zero row
    HIP_TRY(hipMemsetAsync(c.rows, 0, (size_t)t->dim * sizeof(float), s));
    HIP_TRY(hipMemsetAsync(c.meta, 0, META_STRIDE * sizeof(RowMeta), s));
  }
  t->chunks.push_back(c);
  HIP_TRY(hipMemcpyAsync(t->d_chunks + (t->chunks.size() - 1), &t->chunks.back(), sizeof(Chunk),
                         hipMemcpyHostToDevice, s));
  t->rows_cap = (unsigned long long)t->chunks.size() << t->chunk_bits;
  return KV_OK;
}

int build_index(kv_table* t, unsigned long long newcap, unsigned nrows, hipStream_t s) {
  Entry* ne = nullptr;
  HIP_TRY(hipMalloc(&ne, (newcap + 1) * sizeof(Entry)));
  k_fill_entries<<<nblocks((long long)newcap + 1, TB, 8192), TB, 0, s>>>(ne, newcap + 1);
  Entry* old = t->entries;
  TableDev told = dev_view(t);   // the old index: its slot-row hints are carried over
  if (!old) told.entries = nullptr;
  t->entries = ne;
  t->cap = newcap;
  if (nrows > 1) {
    k_rehash<<<nblocks(nrows, TB), TB, 0, s>>>(dev_view(t), told, nrows);
  }
  HIP_TRY(hipStreamSynchronize(s));
  if (old) HIP_TRY(hipFree(old));
  return KV_OK;
}

// the device error flag was found set at a synchronous point: report once, then clear it so the table
// stays usable (the batch that raised it had no effect beyond rows it may have inserted)
int flagged_error(kv_table* t, unsigned code, hipStream_t s) {
  hipMemsetAsync(t->d_counters + 1, 0, sizeof(unsigned), s);
  hipStreamSynchronize(s);
  if (t->err_host) *reinterpret_cast<volatile unsigned*>(t->err_host) = 0u;
  t->batch_serial = 0;
  if (code == 4)
    return fail(KV_INVALID_ARGUMENT, "kv_apply_*_unique: the ids of an earlier call were NOT unique (an id was listed twice): that "
                                     "batch was not applied as the reference applies repeated ids; pass such batches to "
                                     "kv_apply_* (which sums repeated ids) instead");
  return fail(KV_INTERNAL, code == 2 ? "a hash partition received more than 65535 entries in one batch "
                                       "(key set crafted against the partition hash); that batch was not applied"
                                     : "row slab overflow detected on device");
}

// make room for `extra` more keys (worst case: every id of the batch is new).  rows_ub bounds the
// bump allocator, idx_ub the claimed index entries (a key inserted into a row taken from the free
// list claims a new entry while the deleted key's tombstone stays until the next rebuild).
int ensure_capacity(kv_table* t, long long extra, hipStream_t s) {
  unsigned long long need = t->rows_ub + (unsigned long long)extra;
  unsigned long long need_idx = t->idx_ub + (unsigned long long)extra;
  if (need > t->rows_cap || need_idx * 2 > t->cap) {
    // refresh the exact counts before deciding to grow
    unsigned c[3];
    HIP_TRY(hipMemcpyAsync(c, t->d_counters, sizeof c, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (c[1]) return flagged_error(t, c[1], s);
    const long long freed = std::max(0, (int)c[2]);
    t->rows_ub = c[0];
    t->free_known = freed;
    t->idx_ub = t->idx_base + (c[0] - t->bump_base) +
                (unsigned long long)std::max<long long>(0, (long long)t->pushes_since - (freed - t->free_base));
    need = t->rows_ub + (unsigned long long)extra;
    need_idx = t->idx_ub + (unsigned long long)extra;
    if (need >= 0x7FFFFFFFull) return fail(KV_RESOURCE_EXHAUSTED, "more than 2^31 rows in one table");
    while (need > t->rows_cap) {
      int rc = add_chunk(t, s);
      if (rc) return rc;
    }
    if (need_idx * 2 > t->cap) {
      // rebuild from the live rows: tombstones vanish, so the live count decides the size
      const unsigned long long live = t->rows_ub - 1 - (unsigned long long)freed;
      unsigned long long nc = pow2ceil(std::max<unsigned long long>((live + (unsigned long long)extra) * 2, 1024));
      if (nc < t->cap && (live + (unsigned long long)extra) * 4 > t->cap) nc = t->cap;  // no shrink thrash
      int rc = build_index(t, nc, (unsigned)t->rows_ub, s);
      if (rc) return rc;
      t->idx_ub = t->idx_base = live;
      t->bump_base = t->rows_ub;
      t->free_base = freed;
      t->pushes_since = 0;
    }
  }
  t->rows_ub += (unsigned long long)extra;
  t->idx_ub += (unsigned long long)extra;
  return KV_OK;
}

// partitions for a batch of n ids: a power of two <= MAX_P.  Up to 1 M ids: min(1024, n / 32) —
// small batches are cut fine (~32 ids per block) so a 2048-id op still spreads over 64 CUs, large ones
// get 1024 blocks = one resident wave of blocks.  Above 1 M ids: ~1024 ids per partition (2 M ids ->
// 2048 blocks in two waves; with 1024 the hot partitions overflow the LDS entry lists and split:
// measured 331 us vs 2 x 70)
// `many_distinct`: the table's recent batches held mostly distinct ids (err_host[1], below): twice the partitions,
// so that a partition's distinct keys still fit the partition block's LDS hash in one round (1 M nearly distinct ids
// over 1024 partitions are ~960 keys each against 768 slots: every block split its keys and read its entries three
// times)
unsigned pick_partitions(long long n, bool many_distinct = false) {
  if (many_distinct && n >= (1ll << 18)) {
    const unsigned P0 = pick_partitions(n, false);
    return P0 < (unsigned)MAX_P ? P0 * 2u : P0;
  }
  static const long long forced = [] { const char* e = getenv("KV_FORCE_P"); return e ? atoll(e) : 0ll; }();
  if (forced > 0) return (unsigned)forced;  // diagnostic A/B only (tools/)
  unsigned long long want = std::min<unsigned long long>(1024, (unsigned long long)((n + 31) / 32));
  want = std::max<unsigned long long>(want, (unsigned long long)((n + 1023) / 1024));
  unsigned P = 1;
  while (P < want && P < (unsigned)MAX_P) P <<= 1;
  return P;
}

// the entry-list pipeline's partition count when nothing is known about the batch: about 384 ids per partition block
// (k_part2 works on a tile's distinct keys, not on positions: fewer, fatter partitions than the sorted-position
// pipeline wants), 64 .. 1024; never more than pick_partitions(n), so the workspace of either rule holds it
long long forced_P() {   // KV_FORCE_P: diagnostic A/B only (tools/)
  static const long long forced = [] { const char* e = getenv("KV_FORCE_P"); return e ? atoll(e) : 0ll; }();
  return forced;
}
unsigned fused_default_P(long long n) {
  if (forced_P() > 0) return (unsigned)forced_P();
  unsigned P = 64;
  while ((long long)P * 384 < n && P < 1024u) P <<= 1;
  return std::min(P, pick_partitions(n, false));
}

// upper bounds of a batch of n ids: hot keys (more than LCOLD occurrences each) and their chunks
size_t chunk_cap(long long n) { return (size_t)(n / HC + n / (LCOLD + 1) + 4); }

// (re)allocation that leaves the old buffer in place when the new one cannot be had
template <typename T>
int regrow(T** p, size_t count) {
  T* q = nullptr;
  HIP_TRY(hipMalloc(&q, count * sizeof(T)));
  if (*p) hipFree(*p);
  *p = q;
  return KV_OK;
}

static bool stream_is_capturing(hipStream_t s) {
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
}
// the workspace grows behind a stream synchronisation: under a stream capture that is refused BEFORE anything is queued
// (a failed synchronisation would invalidate the caller's capture)
static int ws_sync(hipStream_t s) {
  if (stream_is_capturing(s))
    return fail(KV_FAILED_PRECONDITION, "the table's batch workspace has to grow for this call, which needs a stream synchronisation: "
                                        "run the op once with this batch length outside the stream capture first");
  HIP_TRY(hipStreamSynchronize(s));
  return KV_OK;
}
int ensure_workspace(kv_table* t, long long n, bool need_part, hipStream_t s) {
  Workspace& w = t->ws;
  const unsigned P = pick_partitions(n, true);
  int rc;
  if (n > w.cap_n || P > w.capP) {
    if ((rc = ws_sync(s))) return rc;
    long long cap = std::max<long long>(n, TILE);
    if (w.cap_n) cap = std::max<long long>(cap, std::min<long long>(w.cap_n * 2, 1ll << 30));
    cap = (cap + TILE - 1) / TILE * TILE;
    const unsigned capP = std::max(pick_partitions(cap, true), P);
    const size_t nt = (size_t)(cap / TILE);
    // every buffer is replaced only once its successor exists; a failure leaves the old sizes in force
    w.cap_n = 0; w.capP = 0; w.hpart_elems = 0; w.epart_elems = 0;
    t->batch_serial = 0;
    if ((rc = regrow(&w.ent_key, (size_t)cap)) || (rc = regrow(&w.ent_a, (size_t)cap)) ||
        (rc = regrow(&w.ent_b, (size_t)cap)) || (rc = regrow(&w.ent_base, (size_t)cap)) || (rc = regrow(&w.ent_rec, (size_t)cap)) ||
        (rc = regrow(&w.toff, nt * (capP + 1))) || (rc = regrow(&w.slot_rank, (size_t)cap)) ||
        (rc = regrow(&w.order, (size_t)cap + 1)) || (rc = regrow(&w.coldlist, 2 * (size_t)cap)) ||
        (rc = regrow(&w.hotlist, 2 * (size_t)cap)) || (rc = regrow(&w.litem, (size_t)cap)) ||
        (rc = regrow(&w.items, (size_t)cap)) || (rc = regrow(&w.pmeta, (size_t)capP + 1)) ||
        (rc = regrow(&w.mcount, nt + 1)))
      return rc;
    if (!w.ctr) {   // zeroed: the first tile pass publishes ctr[5] (the previous pass's distinct keys) as a hint
      HIP_TRY(hipMalloc(&w.ctr, 8 * sizeof(unsigned)));
      HIP_TRY(hipMemset(w.ctr, 0, 8 * sizeof(unsigned)));
    }
#ifdef KV_STAMPS
    if ((rc = regrow(&w.dbg, (size_t)16384 * 16))) return rc;
    hipMemset(w.dbg, 0, (size_t)16384 * 16 * 8);
#endif
    w.cap_n = cap;
    w.capP = capP;
  }
  const long long pe = (long long)chunk_cap(w.cap_n) * (long long)t->dim;
  if (need_part && w.hpart_elems < pe) {
    if ((rc = ws_sync(s))) return rc;
    w.hpart_elems = 0;
    if ((rc = regrow(&w.hpart, (size_t)pe))) return rc;
    w.hpart_elems = pe;
  }
  // the tile sums: [tiles of THIS batch][TILE / 2][dim] — sized by the batch, not by the workspace's doubled capacity
  // (half a row per id: 512 MB for a 1 M-id batch at dim 256), grown by half when a longer batch comes
  const long long ee = ((n + TILE - 1) / TILE) * (long long)(TILE / 2) * (long long)t->dim;
  if (need_part && w.epart_elems < ee) {
    if ((rc = ws_sync(s))) return rc;
    const long long want = std::max(ee, std::min((w.cap_n / 2) * (long long)t->dim, w.epart_elems + w.epart_elems / 2));
    w.epart_elems = 0;
    if ((rc = regrow(&w.epart, (size_t)want))) return rc;
    w.epart_elems = want;
  }
  return KV_OK;
}

bool fused_ok(int D);
bool fused_off();
bool fused_tab(const kv_table* t);

// The sharded owner ops (kv_shard_lookup_serve / kv_shard_apply_serve) read a rank's OWN exchange segment where it was
// written: records [lo, lo + len) of the buffers the calling thread's next op on `table` reads come from `ids` / `grad`
// (the send buffers) instead.  Per thread, so another thread's op on the same table sees nothing of it.
struct SelfSegment { const kv_table* table = nullptr; unsigned lo = 0, len = 0; const void* ids = nullptr; const float* grad = nullptr; };
static thread_local bool tl_unique = false;          // kv_apply_*_unique: the caller promises unique ids (apply_common takes the one-launch path)
// The duplicate guard of the one-launch path stamps rows with a launch serial that lives on the HOST: a captured launch
// would be replayed with the serial it was captured with and find its own stamps.  Under stream capture the unique forms
// therefore run the batch pipeline (which needs no promise; same results, bit for bit): stream_is_capturing().
static thread_local bool tl_require_reuse = false;   // the batched sharded apply: the tables must still hold their lookups' indexes
static thread_local std::vector<SelfSegment> tl_selfs;   // (empty outside the sharded owner ops; several tables in the batched ones)
struct SelfScope {
  size_t mark;
  SelfScope() : mark(tl_selfs.size()) {}
  SelfScope(const kv_table* t, bool on, unsigned lo, unsigned len, const void* ids, const float* grad) : mark(tl_selfs.size()) {
    add(t, on, lo, len, ids, grad);
  }
  void add(const kv_table* t, bool on, unsigned lo, unsigned len, const void* ids, const float* grad) {
    if (on) tl_selfs.push_back(SelfSegment{t, lo, len, ids, grad});
  }
  ~SelfScope() { tl_selfs.resize(mark); }
};

WsDev ws_view(kv_table* t, long long n) {
  Workspace& w = t->ws;
  WsDev d;
  d.ent_key = w.ent_key; d.ent_a = w.ent_a; d.ent_b = w.ent_b; d.ent_base = w.ent_base; d.ent_rec = w.ent_rec;
  d.toff = w.toff;
  d.slot_rank = w.slot_rank;
  d.order = w.order;
  d.coldlist = w.coldlist;
  d.hotlist = w.hotlist;
  d.litem = w.litem;
  d.items = w.items;
  d.pmeta = w.pmeta;
  d.hpart = w.hpart;
  d.ctr = w.ctr;
  d.ntiles = (unsigned)((n + TILE - 1) / TILE);
  d.P = pick_partitions(n);
  d.pshift = 64 - ilog2(d.P);
  d.seg_cap = 0;
  d.row_map = nullptr;
  d.zero_counts = nullptr;
  d.dbg = w.dbg;
  // entry-list pipeline: mrow lives in slot_rank's storage
  d.mrow = w.slot_rank;
  d.mcount = w.mcount;
  d.epart = w.epart;
  d.pos_ent = nullptr;
  d.self_lo = d.self_len = 0; d.ids_self = nullptr; d.grad_self = nullptr;
  for (const SelfSegment& x : tl_selfs)
    if (x.table == t) { d.self_lo = x.lo; d.self_len = x.len; d.ids_self = x.ids; d.grad_self = x.grad; }
  return d;
}

// brackets one kernel launch with a pair of events when profiling is on
struct ProfScope {
  kv_table* t;
  hipStream_t s;
  bool on;
  ProfScope(kv_table* t_, int kind, hipStream_t s_) : t(t_), s(s_), on(false) {
    if (t->prof && ((t->prof_mask >> kind) & 1u) && t->ev_used + 2 <= t->ev.size() &&
        (t->prof_every <= 1 || (t->prof_seq[kind]++ % (unsigned)t->prof_every) == 0u)) {
      on = true;
      t->ev_kind[t->ev_used / 2] = kind;
      hipEventRecord(t->ev[t->ev_used], s);
    }
  }
  ~ProfScope() {
    if (on) {
      hipEventRecord(t->ev[t->ev_used + 1], s);
      t->ev_used += 2;
    }
  }
};

// the optimizers' row math on the hardware's 1-ulp sqrt / reciprocal (kv_device.h kv_sqrt / kv_div)?
bool fast_math_on(const kv_table* t) { return t->fast_math && !t->deterministic; }
// PartArgs::det: 0 arrival order, 1 an order fixed by the input positions, 2 occurrence order (one chain per key; kv_kernels.h)
int det_mode(const kv_table* t) { return t->occurrence_order ? 2 : t->deterministic ? 1 : 0; }

unsigned today(const kv_table* t) {
  if (t->fixed_day >= 0) return (unsigned)t->fixed_day & 0xFFFFu;
  return (unsigned)(std::time(nullptr) / (3600 * 24)) & 0xFFFFu;  // utility.cc:38-40
}

// tile pass.  FIRST: scatter / mark flavour (one input position per key instead of the counts).
// ids_kind: -1 = the table's key dtype, 0 int64, 1 int32, 2 (id, count) int64 pairs (lookups only).
// md != nullptr: the same launch over `ntab` tables (grid.y), arguments from the descriptor array
// md, grid.x = gx (the largest table's tile count)
template <bool FIRST>
void launch_tile(kv_table* t, const WsDev& wd, const void* ids, const int* counts, long long n, hipStream_t s,
                 int ids_kind = -1, const MultiDesc* md = nullptr, int ntab = 0, unsigned gx = 0) {
  if (ids_kind < 0) ids_kind = t->key_dtype == KV_DT_INT32 ? 1 : 0;
  const int grid = (int)wd.ntiles;
  const size_t sh = tile_smem_bytes(FIRST);
  const int det = det_mode(t);
#define KV_TILE(IDT)                                                                     \
  do {                                                                                   \
    if (md) k_tile_multi<FIRST, IDT><<<dim3(gx, (unsigned)ntab), TBT, sh, s>>>(md);       \
    else k_tile<FIRST, IDT><<<grid, TBT, sh, s>>>(wd, (const IDT*)ids, counts, n, det);  \
  } while (0)
  if (ids_kind == 2) {
    if constexpr (!FIRST) KV_TILE(IdCount);
  } else if (ids_kind == 1) KV_TILE(int);
  else KV_TILE(long long);
#undef KV_TILE
}

// out[i] = rows[row of ids[i]] after the lookup index passes; md: many tables in one launch.
// order: the same kernel also builds the sorted position list (the training lookup's third and last kernel)
void launch_gather(const TableDev& td, const WsDev& wd, float* op, long long m, hipStream_t s,
                   const MultiDesc* md = nullptr, int ntab = 0, bool order = false) {
  const int D = td.dim;
  const int q = (D % 4 == 0) ? D / 4 : 0;
  const bool vec = q > 0 && (q & (q - 1)) == 0 && q <= TB;
  const long long rows_per_block = vec ? TB / q : 1;
  constexpr int gcap = 8192;  // one 64-row step per wave at 1M rows: residency, not a loop, hides the hops
  const int grid = vec ? nblocks(m, q <= 64 ? TB : (int)rows_per_block, gcap) : nblocks(m * D, TB, 4096);
#define KV_GATHER(VQ)                                                                        \
  do {                                                                                       \
    if (md) k_gather_multi<VQ><<<dim3((unsigned)grid, (unsigned)ntab), TB, 0, s>>>(md);       \
    else if (order) k_gather<VQ, true><<<grid + ITEM_BLOCKS, TB, 0, s>>>(td, wd, op, m);      \
    else k_gather<VQ, false><<<grid, TB, 0, s>>>(td, wd, op, m);                              \
  } while (0)
  switch (vec ? q : 0) {
    case 1: KV_GATHER(1); break;
    case 2: KV_GATHER(2); break;
    case 4: KV_GATHER(4); break;
    case 8: KV_GATHER(8); break;
    case 16: KV_GATHER(16); break;
    case 32: KV_GATHER(32); break;
    case 64: KV_GATHER(64); break;
    case 128: KV_GATHER(128); break;
    case 256: KV_GATHER(256); break;
    default: KV_GATHER(0); break;
  }
#undef KV_GATHER
}

// partition pass.  multi (md != nullptr): wd carries the LARGEST ntiles / P of the batch of tables (LDS
// sizing, grid.x); instantiated for MODE_LOOKUP and MODE_APPLYIDX
template <int MODE>
void launch_part_keys(const WsDev& wd, const PartArgs& pa, hipStream_t s, const MultiDesc* md = nullptr, int ntab = 0) {
  const int grid = (int)wd.P;
  const size_t sh = (size_t)wd.ntiles * 4 + 32;
  if constexpr (MODE == MODE_LOOKUP || MODE == MODE_APPLYIDX) {
    if (md) {
      k_part_keys_multi<MODE><<<dim3((unsigned)grid, (unsigned)ntab), TBK, sh, s>>>(md);
      return;
    }
  }
  k_part_keys<MODE><<<grid, TBK, sh, s>>>(wd, pa);
}
// sorted position list of the batch (the training lookup builds it in its gather kernel instead)
void launch_order(const TableDev& td, const WsDev& wd, long long n, hipStream_t s,
                  const MultiDesc* md = nullptr, int ntab = 0) {
  const int grid = nblocks(n, TB, 4096) + ITEM_BLOCKS;   // ITEM_BLOCKS blocks in front build the item directory only
  if (md) k_order_multi<<<dim3((unsigned)grid, (unsigned)ntab), TB, 0, s>>>(md);
  else k_order<<<grid, TB, 0, s>>>(td, wd, n);
}

// ---- the entry-list pipeline (kv_fused.h) ----
// ids per index pass: positions and epart rows are 30-bit fields of the entry list's words, a partition block takes
// up to 65535 entries; 2^23 ids (4096 tiles) stay well inside both
constexpr long long FUSED_MAX_N = 1ll << 23;
// dims it serves: every multiple of 4 up to 256 (rows of dim / 4 float4; a row's lane group is the next power of two,
// the lanes past the row's end masked: dims 12, 20, 100 ... run the same kernels as 16, 32, 128)
bool fused_off() {
  static const bool off = [] { const char* e = getenv("KV_NO_FUSED"); return e && atoi(e) != 0; }();   // A/B against the sorted-position pipeline
  return off;
}
bool fused_ok(int D) {
  if (fused_off() || (D & 3) != 0) return false;
  const int q = D / 4;
  return q >= 1 && q <= 64;
}
// ... and tables: one in occurrence-order mode takes the sorted-position pipeline for every op, like a dim the entry-list
// kernels do not serve (its sums are one chain per key there; the entry lists sum tile by tile)
bool fused_tab(const kv_table* t) { return fused_ok(t->dim) && !t->occurrence_order; }
// lanes per row of the row-copy kernels: dim / 4 rounded up to a power of two
int row_lanes(int D) { return (int)pow2ceil((unsigned long long)std::max(1, D / 4)); }
bool pow2_rows(int D) { return (D & 3) == 0 && row_lanes(D) == D / 4; }
// tile pass: dedup, index probes / inserts, entries, tile-local order and (out != nullptr) the output rows
// md != nullptr: `ntab` tables in one launch (grid.y), arguments from the descriptor array; multi_rows: with rows
void launch_ltile(kv_table* t, const TableDev& td, const WsDev& wd, const void* ids, const int* counts, long long n, float* out,
                  hipStream_t s, int ids_kind = -1, const MultiDesc* md = nullptr, int ntab = 0, bool multi_rows = false) {
  if (ids_kind < 0) ids_kind = t->key_dtype == KV_DT_INT32 ? 1 : 0;
  const int grid = (int)wd.ntiles;
  const size_t sh = ltile_smem_bytes();
  const int det = det_mode(t);
  const int q = row_lanes(td.dim);
#define KV_LT2(IDT, VQ)                                                                                     \
  do {                                                                                                      \
    if (md && multi_rows) k_ltile_multi<IDT, VQ, true><<<dim3((unsigned)grid, (unsigned)ntab), TBT, sh, s>>>(md); \
    else if (md) k_ltile_multi<IDT, 1, false><<<dim3((unsigned)grid, (unsigned)ntab), TBT, sh, s>>>(md);     \
    else if (out) k_ltile<IDT, VQ, true><<<grid, TBT, sh, s>>>(td, wd, (const IDT*)ids, counts, n, det, out); \
    else k_ltile<IDT, 1, false><<<grid, TBT, sh, s>>>(td, wd, (const IDT*)ids, counts, n, det, nullptr);     \
  } while (0)
#define KV_LT(IDT)                                                              \
  do {                                                                          \
    switch (q) {                                                                \
      case 1: KV_LT2(IDT, 1); break;   case 2: KV_LT2(IDT, 2); break;           \
      case 4: KV_LT2(IDT, 4); break;   case 8: KV_LT2(IDT, 8); break;           \
      case 16: KV_LT2(IDT, 16); break; case 32: KV_LT2(IDT, 32); break;         \
      default: KV_LT2(IDT, 64); break;                                          \
    }                                                                           \
  } while (0)
  if (ids_kind == 2) KV_LT(IdCount);
  else if (ids_kind == 1) KV_LT(int);
  else KV_LT(long long);
#undef KV_LT
#undef KV_LT2
}
// the table-less tile pass of the sharded route (int64 ids): entries, mrow, every position's entry number
void launch_ltile_notable(kv_table* t, const TableDev& td, const WsDev& wd, const void* ids, long long n, hipStream_t s,
                          const int* counts = nullptr, bool int32_ids = false) {
  const int det = det_mode(t);
  if (int32_ids) k_ltile<int, 1, false, true><<<(int)wd.ntiles, TBT, ltile_smem_bytes(), s>>>(td, wd, (const int*)ids, counts, n, det, nullptr);
  else k_ltile<long long, 1, false, true><<<(int)wd.ntiles, TBT, ltile_smem_bytes(), s>>>(td, wd, (const long long*)ids, counts, n, det, nullptr);
}
// the bookkeeping of a training lookup that no apply takes over (k_part2); md: `ntab` tables in one launch
void launch_part2(const WsDev& wd, const PartArgs& pa, hipStream_t s, const MultiDesc* md = nullptr, int ntab = 0) {
  if (md) k_part2_multi<<<dim3(wd.P, (unsigned)ntab), TBK, (size_t)wd.ntiles * 4 + 32, s>>>(md);
  else k_part2<<<(int)wd.P, TBK, (size_t)wd.ntiles * 4 + 32, s>>>(wd, pa);
}

// segmented fold over the sorted positions + fused update (k_apply_sorted), then the keys that cross chunk
// boundaries (k_apply_span).  pa.n = ids of the batch (multi: nmax = the largest table's batch)
template <int MODE, int OPT>
int launch_apply(kv_table* prof_t, const WsDev& wd, const PartArgs& pa, long long nmax, hipStream_t s,
                 const MultiDesc* md = nullptr, int ntab = 0, bool skip_fin = false) {
  const int D = pa.tv.dim;
  // waves stride over the items (hot chunks, then cold batches of 64 / LPR keys); 8 blocks of 4 waves per CU
  // is everything the chip holds at once, fewer for small batches
  constexpr int gmax = 2048;
  const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>(gmax, (nmax / 2 + chunk_cap(nmax)) / 4 + 1));
  const unsigned gfin = (unsigned)std::max<long long>(1, std::min<long long>(256, nmax / 4096 + 1));   // each block reads its share of the items at once
  auto fn = (MODE == MODE_APPLY && (OPT == OPT_ADAM_V4 || OPT == OPT_ADAM_V3)) ? kvp_launch_apply_a : kvp_launch_apply_b;
  int rc;
  if (pa.det == 2 && !md) {
    // occurrence order: the hot keys' chains first (k_occ_sum: one block per key, the sums to hpart), k_apply reads them
    const int fop = MODE == MODE_APPLY ? KV_SCATTER_ADD : pa.fold_op;
    const unsigned og = (unsigned)std::max<long long>(1, std::min<long long>(2048, nmax / 256 + 1));
    const int nc = (D + 63) / 64;
    // (two stages + positions: above the 64 KB a launch gets without asking — per device, so asked at every launch)
#define KV_OCC(NC_)                                                                                                          \
    do {                                                                                                                       \
      HIP_TRY(hipFuncSetAttribute((const void*)k_occ_sum<NC_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)occ_smem_bytes())); \
      k_occ_sum<NC_><<<og, OCC_TB, occ_smem_bytes(), s>>>(wd, pa, fop);                                                        \
    } while (0)
    if (nc <= 1) KV_OCC(1);
    else if (nc <= 2) KV_OCC(2);
    else if (nc <= 4) KV_OCC(4);
    else if (nc <= 8) KV_OCC(8);
    else KV_OCC(16);
#undef KV_OCC
  }
  {
    ProfScope ps(prof_t, KV_PROF_APPLY_SORTED, s);
    rc = fn(MODE, OPT, &wd, &pa, (void*)s, md, ntab, grid, 0);
  }
  if (rc == KV_OK && !skip_fin) {
    ProfScope ps(prof_t, KV_PROF_APPLY_SPAN, s);
    rc = fn(MODE, OPT, &wd, &pa, (void*)s, md, ntab, gfin, 1);
  }
  if (rc == KV_UNIMPLEMENTED)
    return md ? fail(KV_UNIMPLEMENTED, "batched launch: embedding dim %d (multiples of 4 only)", D)
              : fail(KV_UNIMPLEMENTED, "embedding dim %d not supported by the fused kernels "
                     "(multiples of 4 up to 1024, any dim up to 256)", D);
  if (rc) return fail(rc, "apply pass: no kernel for mode %d / optimizer %d", MODE, OPT);
  return KV_OK;
}

bool dim_supported(int D) { return (D & 3) == 0 ? D <= 1024 : D <= 256; }

int check_table(kv_handle_t h) {
  if (!h) return fail(KV_INVALID_ARGUMENT, "null table handle");
  return KV_OK;
}

// A batch the index pass gave up on (a hash partition with more than 65535 entries, or keys that no
// sub-hash separates) raises the device flag AND this pinned host word; every later kernel of that op saw
// the flag and did nothing.  The next call on the table reports it — no synchronisation on the good path.
int report_deferred_error(kv_table* t, hipStream_t s) {
  if (!t->err_host || *reinterpret_cast<volatile unsigned*>(t->err_host) == 0u) return KV_OK;
  const unsigned code = *reinterpret_cast<volatile unsigned*>(t->err_host);
  hipStreamSynchronize(s);
  return flagged_error(t, code, s);
}

// Ops of one table run in the order they were issued, whatever their streams: the per-table workspace
// and the table itself are shared by every op (the reference's table locks cover execution, not just
// enqueue, training_ops.cc:96-184).  Same stream as the last op: nothing to do.  Another stream: it first
// waits for everything the previous stream had been given.
// launches a lookup's pending partition pass (see kv_table::part_pending) on stream s
int flush_part(kv_table* t, hipStream_t s);
// ---- slot mirrors: the host side -------------------------------------------------------------------------------------
// Invariant: the slot table's own records are up to date for every key whose var-row mirror is not (valid in the current
// epoch AND dirty).  Mirrors are written only by the lean apply of (var, slot) — k_papply / k_uapply with use_mirror — and
// read only by it.  EVERY other op that enters either table first ends the epoch (mirror_end_epoch: flush the dirty
// copies, one kernel over the var's rows, then epoch + 1 — which invalidates every copy at once), so it sees, and may
// change, the authoritative records; the keys' next lean apply finds no valid mirror, takes the general path once and
// leaves a fresh clean copy (finish_key).  The ops that keep the epoch name their tables in tl_mirror_keep while they
// enter: the training / inference lookups on the var (they touch var records and rows only), the GroupAdam and Adagrad
// applies on (var, slot), single and batched, and the ops that only borrow a table's workspace.  KV_NO_MIRROR=1: never (A/B).
static thread_local kv_table* tl_mirror_keep[2] = {nullptr, nullptr};   // [0]: kept in its VAR role, [1]: kept in its SLOT role
static thread_local kv_table* const* tl_mirror_keep_vars = nullptr;      // ... a batched op's tables in their VAR role
static thread_local int tl_mirror_keep_nvars = 0;
static thread_local kv_table* const* tl_mirror_keep_slots = nullptr;     // ... a batched apply's first slot tables, in their SLOT role
static thread_local int tl_mirror_keep_nslots = 0;
struct MirrorKeep {   // (scopes nest: the previous names come back)
  kv_table* p0; kv_table* p1; kv_table* const* pv; int pn; kv_table* const* ps; int psn;
  void save() {
    p0 = tl_mirror_keep[0]; p1 = tl_mirror_keep[1]; pv = tl_mirror_keep_vars; pn = tl_mirror_keep_nvars;
    ps = tl_mirror_keep_slots; psn = tl_mirror_keep_nslots;
  }
  explicit MirrorKeep(kv_table* as_var, kv_table* as_slot = nullptr) { save(); tl_mirror_keep[0] = as_var; tl_mirror_keep[1] = as_slot; }
  MirrorKeep(kv_table* const* vars, int n, kv_table* const* slots = nullptr, int nslots = 0) {
    save(); tl_mirror_keep_vars = vars; tl_mirror_keep_nvars = n; tl_mirror_keep_slots = slots; tl_mirror_keep_nslots = nslots;
  }
  ~MirrorKeep() {
    tl_mirror_keep[0] = p0; tl_mirror_keep[1] = p1; tl_mirror_keep_vars = pv; tl_mirror_keep_nvars = pn;
    tl_mirror_keep_slots = ps; tl_mirror_keep_nslots = psn;
  }
};
static bool mirror_kept_as_var(const kv_table* t) {
  if (t == tl_mirror_keep[0]) return true;
  for (int i = 0; i < tl_mirror_keep_nvars; ++i)
    if (tl_mirror_keep_vars[i] == t) return true;
  return false;
}
static bool mirror_kept_as_slot(const kv_table* t) {
  if (t == tl_mirror_keep[1]) return true;
  for (int i = 0; i < tl_mirror_keep_nslots; ++i)
    if (tl_mirror_keep_slots[i] == t) return true;
  return false;
}
bool mirror_off() {
  static const bool off = [] { const char* e = getenv("KV_NO_MIRROR"); return e && e[0] == '1'; }();
  return off;
}
// held: the table of the pair whose lock the caller holds (the var itself, or its slot table): the views come from there
void mirror_end_epoch(kv_table* var, hipStream_t s, const kv_table* held) {
  if (!var->mirror_slot) return;
  var->stat_mirror_epochs.fetch_add(1);
  const TableDev& tv = held->mview_var;
  const long long rows = (long long)tv.max_rows + 1;
  if (var->mirror_dirty.exchange(false))
    k_flush_mirrors<<<nblocks(rows, TB, 8192), TB, 0, s>>>(tv, held->mview_slot, var->mirror_epoch.load() & 0xFFFFu);
  if ((var->mirror_epoch.fetch_add(1) + 1u) > 0xFFFFu) {
    k_clear_mirrors<<<nblocks(rows, TB, 8192), TB, 0, s>>>(tv);
    var->mirror_epoch.store(1);
  }
}
void mirror_unpair(kv_table* var, hipStream_t s, const kv_table* held) {
  if (!var->mirror_slot) return;
  mirror_end_epoch(var, s, held);
  var->mirror_slot->mirror_var = nullptr;
  var->mirror_slot = nullptr;
}
// (both locks held)
static void mirror_snapshot(kv_table* v, kv_table* sl) {
  v->mview_var = sl->mview_var = dev_view(v);
  v->mview_slot = sl->mview_slot = dev_view(sl);
}
// the entry hook of hand_over / join_side.  Under a stream capture the flush would be RECORDED, not run, and a replay would
// carry the epoch of its capture: a table that is captured gives up its mirrors beforehand (kv_prepare_capture ->
// mirror_ban); an op that would have to end an epoch inside a capture is refused.
int mirror_on_entry(kv_table* t, hipStream_t s) {
  if (!t->mirror_var && !t->mirror_slot) return KV_OK;
  const bool end_slot = t->mirror_var && !mirror_kept_as_slot(t), end_var = t->mirror_slot && !mirror_kept_as_var(t);
  if ((end_slot || end_var) && stream_is_capturing(s))
    return fail(KV_FAILED_PRECONDITION, "this table is half of a (var, slot) pair whose optimizer applies keep the slot records' "
                                        "frequency words in the var's rows between ops; call kv_prepare_capture on it (outside the "
                                        "capture) before capturing ops on it");
  if (end_slot) mirror_end_epoch(t->mirror_var, s, t);   // t is a slot table: its records are about to be read or written
  if (end_var) mirror_end_epoch(t, s, t);                // t is a var: its rows may be released, moved or read back
  return KV_OK;
}
// kv_prepare_capture: the table's ops are about to be captured and replayed — no host code runs at a replay, so nothing
// could flush or re-validate a mirror: the pair is dissolved now (the dirty copies go back) and never forms again
void mirror_ban(kv_table* t, hipStream_t s) {
  if (t->mirror_slot) mirror_unpair(t, s, t);
  if (t->mirror_var) mirror_unpair(t->mirror_var, s, t);
  t->mirror_banned = true;
}
// (var, slot) become a mirror pair — or stay / become unpaired when the slot table already serves another var
bool mirror_pair(kv_table* v, kv_table* sl, hipStream_t s) {
  if (mirror_off() || sl->mirror_shared || v->mirror_banned || sl->mirror_banned) return false;
  if (v->mirror_slot == sl && sl->mirror_var == v) return true;
  if (sl->mirror_var && sl->mirror_var != v) {   // a second var on one slot table: no mirrors for it at all
    mirror_unpair(sl->mirror_var, s, sl);
    sl->mirror_shared = true;
    return false;
  }
  if (v->mirror_slot && v->mirror_slot != sl) mirror_unpair(v, s, v);
  v->mirror_slot = sl;
  sl->mirror_var = v;
  mirror_snapshot(v, sl);
  mirror_end_epoch(v, s, v);   // a fresh epoch: whatever bytes the rows' mirror units hold are void
  return true;
}

// Does this apply of (v, s0) work on the var rows' slot mirrors?  lean: the launch is k_papply / k_uapply (their lean
// update is the only code that reads or writes a mirror).  Otherwise the apply reads and writes the slot table's own
// records: the epoch ends first (the caller entered both tables under MirrorKeep, so nothing has ended it yet).
static int mirror_decide_rt(int opt, kv_table* v, kv_table* s0, PartArgs& pa, bool lean, hipStream_t s);
template <int OPT>
static int mirror_decide(kv_table* v, kv_table* s0, PartArgs& pa, bool lean, hipStream_t s) { return mirror_decide_rt(OPT, v, s0, pa, lean, s); }
static int mirror_decide_rt(int opt, kv_table* v, kv_table* s0, PartArgs& pa, bool lean, hipStream_t s) {
  pa.use_mirror = 0; pa.mirror_epoch = 0u;
  // (a captured apply of a pair that still has mirrors: their flush would be recorded, not run — see mirror_on_entry)
  if ((v->mirror_slot || s0->mirror_var) && stream_is_capturing(s))
    return fail(KV_FAILED_PRECONDITION, "optimizer apply under stream capture on a (var, slot) pair with live slot mirrors: call "
                                        "kv_prepare_capture on both tables (outside the capture) first");
  const bool eligible = opt != OPT_FTRL && lean && pa.use_hints != 0 && pa.tv.single != 0u && pa.ts0.single != 0u &&
                        !v->track_delta && !s0->track_delta && !stream_is_capturing(s) && mirror_pair(v, s0, s);
  if (eligible) {
    pa.use_mirror = 1;
    pa.mirror_epoch = v->mirror_epoch.load() & 0xFFFFu;
    mirror_snapshot(v, s0);   // what this apply's dirty copies name is inside these views
    v->mirror_dirty.store(true);
    ++v->stat_mirror_applies;
  } else {
    if (v->mirror_slot) mirror_end_epoch(v, s, v);
    if (s0->mirror_var && s0->mirror_var != v) mirror_end_epoch(s0->mirror_var, s, s0);
  }
  return KV_OK;
}

// mutates == false: a read-only op (the inference gathers): ordered like any other op of the table — behind the table's
// last op whatever its stream, and the next op behind it — but it does not move op_serial (a two-phase export may go on)
// settle == false: the caller is the optimizer apply that takes the table's pending partition pass over
int hand_over(kv_table* t, hipStream_t s, bool settle = true, bool mutates = true) {
  if (t->part_pending && settle) {   // (an apply that takes the batch over runs it itself, behind the stream hand-over below)
    int rc;
    if (t->has_last && t->last_stream != s) {
      HIP_TRY(hipEventRecord(t->last_done, t->last_stream));
      HIP_TRY(hipStreamWaitEvent(s, t->last_done, 0));
      t->last_stream = s;
    }
    if ((rc = flush_part(t, s))) return rc;
  }
  if (t->has_last && t->last_stream != s) {
    HIP_TRY(hipEventRecord(t->last_done, t->last_stream));
    HIP_TRY(hipStreamWaitEvent(s, t->last_done, 0));
  }
  t->last_stream = s;
  t->has_last = true;
  if (mutates) ++t->op_serial;
  return mirror_on_entry(t, s);
}

// ops that read a table without the full hand_over (no workspace, no row-set change): the last lookup's pending
// partition pass (it may still have rows to initialise) is settled first
int join_side(kv_table* t, hipStream_t s) {
  if (t->part_pending) {
    if (t->has_last && t->last_stream != s) {
      HIP_TRY(hipEventRecord(t->last_done, t->last_stream));
      HIP_TRY(hipStreamWaitEvent(s, t->last_done, 0));
      t->last_stream = s;
    }
    const int rc = flush_part(t, s);
    if (rc) return rc;
  }
  // An epoch of slot mirrors that ends here flushes copies the last lean apply wrote — on the stream of t's last op (an apply
  // enters both tables; anything later on t has ended the epoch already): the flush must run behind it.  (A flush that
  // overtook the apply would miss its copies, and the epoch number that ends with it would orphan them for good.)
  if ((t->mirror_var || t->mirror_slot) && t->has_last && t->last_stream != s) {
    HIP_TRY(hipEventRecord(t->last_done, t->last_stream));
    HIP_TRY(hipStreamWaitEvent(s, t->last_done, 0));
    t->last_stream = s;
  }
  return mirror_on_entry(t, s);
}

// locks tables in address order like MaybeLockVariableInputMutexesInOrder (training_ops.cc:96-184)
struct MultiLock {
  std::vector<kv_table*> ts;
  explicit MultiLock(std::initializer_list<kv_table*> l) : MultiLock(std::vector<kv_table*>(l)) {}
  explicit MultiLock(std::vector<kv_table*> l) : ts(std::move(l)) {
    std::sort(ts.begin(), ts.end());
    ts.erase(std::unique(ts.begin(), ts.end()), ts.end());
    for (auto* t : ts) t->mu.lock();
  }
  ~MultiLock() { for (auto it = ts.rbegin(); it != ts.rend(); ++it) (*it)->mu.unlock(); }
  // later: this table's pending partition pass is taken over by the caller (the optimizer apply of that batch)
  int enter(hipStream_t s, kv_table* later = nullptr) {
    int rc;
    for (auto* t : ts)
      if ((rc = report_deferred_error(t, s)) || (rc = hand_over(t, s, t != later))) return rc;
    return KV_OK;
  }
};
// the same for ops on one table (the caller holds t->mu)
int enter_op(kv_table* t, hipStream_t s) {
  int rc;
  if ((rc = report_deferred_error(t, s))) return rc;
  return hand_over(t, s);
}

// The index of a batch (kv_kernels.h): tile pass, partition pass, sorted position list.
//   MODE_LOOKUP   with out != nullptr: the training lookup (rows copied by the kernel that builds the list)
//   MODE_APPLYIDX the optimizer meets the ids first (FindOrInsertUnsafe on the var table)
//   MODE_UNIQUE   no table: dense unique indices (pa.out_keys / direct_rows)
template <int MODE>
void index_pass(kv_table* t, const WsDev& wd, const PartArgs& pa, const void* ids, const int* counts, long long n,
                int ids_kind, float* out, hipStream_t s, bool file_order = true) {
  t->fused_index = false;
  {
    ProfScope ps(t, MODE == MODE_LOOKUP ? KV_PROF_LOOKUP_TILE : KV_PROF_INDEX, s);
    launch_tile<false>(t, wd, ids, counts, n, s, ids_kind);
  }
  {
    ProfScope ps(t, MODE == MODE_LOOKUP ? KV_PROF_LOOKUP_PART : KV_PROF_INDEX, s);
    launch_part_keys<MODE>(wd, pa, s);
  }
  ProfScope ps(t, MODE == MODE_LOOKUP ? KV_PROF_LOOKUP_ORDER : KV_PROF_INDEX, s);
  // file_order == false: a lookup nobody will follow with an apply of the same batch (no token asked for): the
  // plain gather, 36 instead of 48 us at configs[1]
  if (MODE == MODE_LOOKUP && out) launch_gather(pa.tv, wd, out, n, s, nullptr, 0, file_order);
  else launch_order(pa.tv, wd, n, s);
}

int flush_part(kv_table* t, hipStream_t s) {
  if (!t->part_pending) return KV_OK;
  t->part_pending = false;
  WsDev wd; PartArgs pa;
  std::memcpy(&wd, t->pend_wd, sizeof wd);
  std::memcpy(&pa, t->pend_pa, sizeof pa);
  ProfScope ps(t, KV_PROF_LOOKUP_PART, s);
  // (k_part2 reads the tiles' entries alone — also behind a sharded owner lookup whose own segment stayed in the send
  //  buffers; an apply that still comes with the batch's token runs k_papply PA_NONE over the same entries)
  launch_part2(wd, pa, s);
  HIP_TRY(hipGetLastError());
  return KV_OK;
}

// partitions of an entry-list index pass over n ids (wd.P, wd.pshift; remembered in t->index_P)
void choose_partitions(kv_table* t, WsDev& wd, long long n) {
  {
    // the distinct ids of the batch before the last one (the tile pass hands the partition pass's count to the host
    // through a pinned word, no synchronisation): mostly distinct ids -> twice the partitions
    const unsigned u_prev = t->err_host ? reinterpret_cast<volatile unsigned*>(t->err_host)[1] : 0u;
    // (not in deterministic mode: the partitioning decides the order of a tile's entries, hence of the additions —
    // there it depends on the batch alone)
    const bool hinted = !t->deterministic && t->batch_n_prev == n && u_prev > 0u && forced_P() <= 0;
    t->batch_n_prev = n;
    if (hinted) {
      // about 384 distinct keys per partition block (the LDS hash of k_part2 holds 768 before a partition splits),
      // at most 2048 entries: 109 k keys of 1 M ids (Zipf 1.2) -> 512 partitions, 773 k (Zipf 0.8) -> 2048.
      // Measured at 1 M ids: 512 against 1024 partitions is -2.5 us per step at Zipf 1.2 and +40 us at Zipf 0.8.
      // With k_papply (kv_papply.h) the partition block also applies its keys' updates: two blocks of eight waves per
      // CU, about 256 keys per block (109 k keys -> 512 partitions, one resident generation: 60 us; four blocks of four
      // waves with 128 keys each 64.7 us, one block of sixteen waves with 512 keys 75 us; profiles/r04_tools_output.txt).
      // (a hint is only a hint: never more distinct keys than ids, never more partitions than the workspace was sized for)
      const unsigned long long u = std::min<unsigned long long>(u_prev, (unsigned long long)n);
      const unsigned long long per = 256ull;
      const unsigned long long want = std::max<unsigned long long>((u + per - 1ull) / per, (unsigned long long)((n + 2047) / 2048));
      const unsigned pmax = std::min<unsigned>((unsigned)MAX_P, std::max(64u, t->ws.capP));
      unsigned P = 64;
      while (P < want && P < pmax) P <<= 1;
      wd.P = P;
    } else {
      wd.P = fused_default_P(n);
    }
    wd.pshift = 64 - ilog2(wd.P);
    t->index_P = wd.P;
  }
}

// The training lookup on the entry-list kernels (kv_fused.h): the tile pass with the output rows; then the lookup's
// bookkeeping (k_part2) — or, defer_part: it stays PENDING for the optimizer apply of this batch (k_papply completes it
// in the same pass as the update) or for whatever op the table sees next (flush_part)
int fused_lookup_pass(kv_table* t, WsDev& wd, const PartArgs& pa, const void* ids, const int* counts, long long n,
                      int ids_kind, float* out, hipStream_t s, bool defer_part) {
  t->fused_index = true;
  choose_partitions(t, wd, n);
  {
    ProfScope ps(t, KV_PROF_LOOKUP_TILE, s);
    launch_ltile(t, pa.tv, wd, ids, counts, n, out, s, ids_kind);
  }
  if (defer_part) {   // the rows are out: the partition pass waits for the table's next op
    std::memcpy(t->pend_wd, &wd, sizeof wd);
    std::memcpy(t->pend_pa, &pa, sizeof pa);
    t->part_pending = true;
    return KV_OK;
  }
  ProfScope ps(t, KV_PROF_LOOKUP_PART, s);
  launch_part2(wd, pa, s);
  return KV_OK;
}
// ... and the optimizer apply over the tiles' entries: the tile sums of the repeated ids (k_tsum; tile_ids != nullptr: the
// batch's tile pass has not run yet and runs in the same launch, k_ltsum), then partition pass + update in one launch
// (k_papply, kv_papply.h: pa_mode = PA_LOOKUP / PA_APPLYIDX / PA_NONE)
template <int OPT>
int fused_apply(kv_table* v, WsDev& wd, PartArgs& pa, long long n, hipStream_t s, int pa_mode, const void* tile_ids = nullptr) {
  pa.epart = wd.epart;
  if (tile_ids) {
    ProfScope ps(v, KV_PROF_APPLY_TILE, s);
    const int rc = kvp_launch_ltsum(&pa.tv, &wd, tile_ids, v->key_dtype == KV_DT_INT32 ? 1 : 0, n, v->deterministic ? 1 : 0,
                                    pa.grad, (void*)s);
    if (rc) return fail(rc, "tile pass + tile sums: no kernel for dim %d", pa.tv.dim);
  } else {
    ProfScope ps(v, KV_PROF_APPLY_TSUM, s);
    const int rc = kvp_launch_tsum(&pa.tv, &wd, pa.grad, (void*)s, nullptr, 0);
    if (rc) return fail(rc, "tile sums: no kernel for dim %d", pa.tv.dim);
  }
  ProfScope ps(v, KV_PROF_APPLY_SORTED, s);
  const int rc = (OPT == OPT_ADAM_V4 || OPT == OPT_ADAM_V3) ? kvp_launch_papply_a(OPT, &wd, &pa, pa_mode, (void*)s)
                                                            : kvp_launch_papply_b(OPT, &wd, &pa, pa_mode, (void*)s);
  if (rc) return fail(rc, "partition + apply pass: no kernel for dim %d", pa.tv.dim);
  return KV_OK;
}

// Every live table: a stream the LIBRARY owns (a communicator's) is retired from the tables that last ran on it before it
// is destroyed — hand_over would otherwise record an event on a dead stream at the table's next op (found in round 6:
// bench.py's sharded_world1 sub-record destroys its communicator, the next lookup on another stream crashed in
// hipEventRecord).  A stream the CALLER owns must outlive the table's next op, or the caller synchronises it first and
// calls kv_forget_stream.
std::mutex g_tables_mu;
std::unordered_set<kv_table*> g_tables;
void retire_stream(hipStream_t dead) {   // `dead` is drained (the caller synchronised it)
  std::lock_guard<std::mutex> l(g_tables_mu);
  for (kv_table* t : g_tables) {
    std::lock_guard<std::mutex> lt(t->mu);
    if (t->has_last && t->last_stream == dead) { t->has_last = false; t->last_stream = nullptr; }
  }
}

std::atomic<uint64_t> g_serial{0};   // batch tokens
std::atomic<uint64_t> g_uid{0};

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
// keys recorded by Delete while the table tracks deltas (kv_variable.h:747,772): no row carries them
static int record_deleted(kv_table* t, const void* ids, int64_t n, bool int32_ids, hipStream_t s) {
  if (!t->track_delta || n <= 0) return KV_OK;
  const size_t base = t->del_train.size();
  t->del_train.resize(base + (size_t)n);
  if (int32_ids) {
    std::vector<int> tmp((size_t)n);
    HIP_TRY(hipMemcpyAsync(tmp.data(), ids, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (int64_t i = 0; i < n; ++i) t->del_train[base + (size_t)i] = tmp[(size_t)i];
  } else {
    HIP_TRY(hipMemcpyAsync(t->del_train.data() + base, ids, (size_t)n * sizeof(long long), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  return KV_OK;
}

// one recorded list: sorted, unique; keys that have a row again hand their membership to that row
static int delta_resolve(kv_table* t, std::vector<long long>* list, int which, hipStream_t s) {
  std::sort(list->begin(), list->end());
  list->erase(std::unique(list->begin(), list->end()), list->end());
  const size_t n = list->size();
  if (n == 0) return KV_OK;
  long long* dk = nullptr;
  unsigned char* dp = nullptr;
  HIP_TRY(hipMalloc(&dk, n * sizeof(long long)));
  if (hipMalloc(&dp, n) != hipSuccess) { hipFree(dk); return fail(KV_RESOURCE_EXHAUSTED, "delta export scratch"); }
  std::vector<unsigned char> present(n);
  hipError_t e = hipMemcpyAsync(dk, list->data(), n * sizeof(long long), hipMemcpyHostToDevice, s);
  if (e == hipSuccess) {
    k_delta_resolve<<<nblocks((long long)n, TB, 4096), TB, 0, s>>>(dev_view(t), dk, (long long)n, which, dp);
    e = hipMemcpyAsync(present.data(), dp, n, hipMemcpyDeviceToHost, s);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  hipFree(dk); hipFree(dp);
  if (e != hipSuccess) return fail(KV_INTERNAL, "delta export: %s", hipGetErrorString(e));
  size_t o = 0;
  for (size_t i = 0; i < n; ++i)
    if (!present[i]) (*list)[o++] = (*list)[i];
  list->resize(o);
  return KV_OK;
}

// all_delta minus the keys that have rows (dynamic_save.hpp:213-228): the recorded deletions still absent
static int delta_prepare(kv_table* t, int first_n, hipStream_t s, std::vector<long long>* absent) {
  int rc;
  if ((rc = delta_resolve(t, &t->del_train, 0, s))) return rc;
  *absent = t->del_train;
  if (first_n <= 3) {
    if ((rc = delta_resolve(t, &t->del_pred, 1, s))) return rc;
    std::vector<long long> u;
    std::set_union(t->del_train.begin(), t->del_train.end(), t->del_pred.begin(), t->del_pred.end(), std::back_inserter(u));
    absent->swap(u);
  }
  return KV_OK;
}

static int delta_after_export(kv_table* t, int first_n, unsigned nrows, hipStream_t s) {
  if (!t->track_delta && !t->track_pred && t->del_train.empty() && t->del_pred.empty()) return KV_OK;
  const int mode = first_n <= 3 ? 1 : 0;
  k_delta_clear<<<nblocks(nrows, TB, 2048), TB, 0, s>>>(dev_view(t), nrows, mode, t->track_pred ? 1 : 0);
  HIP_TRY(hipGetLastError());
  if (mode == 1) {
    t->del_pred.clear();
  } else {
    if (t->track_pred) t->del_pred.insert(t->del_pred.end(), t->del_train.begin(), t->del_train.end());
    t->del_train.clear();
  }
  return KV_OK;
}

extern "C" {

const char* kv_last_error(void) { return g_err.c_str(); }

int kv_create(int key_dtype, int value_dtype, int dim, int enter_threshold, int64_t capacity_hint,
              int device, kv_handle_t* out) {
  if (!out) return fail(KV_INVALID_ARGUMENT, "out is null");
  if (key_dtype != KV_DT_INT64 && key_dtype != KV_DT_INT32 && key_dtype != KV_DT_UINT64)
    return fail(KV_INVALID_ARGUMENT, "key_dtype %d: only int32/int64/uint64 (kv_variable_ops.cc:149-156)", key_dtype);
  if (value_dtype != KV_DT_FLOAT)
    return fail(KV_UNIMPLEMENTED, "value_dtype %d: only float has optimizer kernels (training_ops.cc:7232)", value_dtype);
  if (dim <= 0) return fail(KV_INVALID_ARGUMENT, "Inner dimension should be greater than zero.");
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(KV_INVALID_ARGUMENT, "device %d out of range (%d GPUs)", device, ndev);
  DeviceGuard dg(device);
  kv_table* t = new kv_table();
  t->device = device;
  t->key_dtype = key_dtype;
  t->dim = dim;
  t->enter_threshold = (unsigned)(unsigned short)std::min<int>(enter_threshold, 65535);  // SaturateMaxFrequency
  unsigned long long hint = capacity_hint > 0 ? (unsigned long long)capacity_hint + 1 : 0;
  t->chunk_bits = std::max(16, std::min(30, ilog2(std::max<unsigned long long>(hint, 1))));
  hipStream_t s = nullptr;
  int rc = KV_OK;
  do {
    if (hipMalloc(&t->d_chunks, MAX_CHUNKS * sizeof(Chunk)) != hipSuccess ||
        hipMalloc(&t->d_counters, 8 * sizeof(unsigned)) != hipSuccess ||
        hipMalloc(&t->d_stat, 4 * sizeof(unsigned long long)) != hipSuccess) {
      rc = fail(KV_RESOURCE_EXHAUSTED, "hipMalloc of table header failed");
      break;
    }
    if (hipHostMalloc(&t->err_host, 4 * sizeof(unsigned), hipHostMallocMapped) != hipSuccess ||
        hipEventCreateWithFlags(&t->last_done, hipEventDisableTiming) != hipSuccess) {
      rc = fail(KV_RESOURCE_EXHAUSTED, "table header: pinned word / event");
      break;
    }
    t->err_host[0] = t->err_host[1] = t->err_host[2] = t->err_host[3] = 0;
    t->uid = ++g_uid;
    unsigned init[8] = {1, 0, 0, 0, 0, 0, 0, 0};  // next_row = 1 (row 0 is the zero row)
    if (hipMemcpy(t->d_counters, init, sizeof init, hipMemcpyHostToDevice) != hipSuccess) {
      rc = fail(KV_INTERNAL, "hipMemcpy failed");
      break;
    }
    if ((rc = add_chunk(t, s))) break;
    if ((rc = build_index(t, pow2ceil(std::max<unsigned long long>(2 * t->rows_cap, 1024)), 1, s))) break;
  } while (0);
  if (rc) { kv_destroy(t); return rc; }
  { std::lock_guard<std::mutex> l(g_tables_mu); g_tables.insert(t); }
  *out = t;
  return KV_OK;
}

int kv_destroy(kv_handle_t t) {
  if (!t) return KV_OK;
  { std::lock_guard<std::mutex> l(g_tables_mu); g_tables.erase(t); }
  {   // slot mirrors: a var hands its dirty copies back before it goes; a slot table takes its var's pairing with it
    DeviceGuard dgm(t->device);
    if (t->mirror_slot) mirror_unpair(t, nullptr, t);
    if (t->mirror_var) { t->mirror_var->mirror_dirty.store(false); t->mirror_var->mirror_slot = nullptr; t->mirror_var = nullptr; }
  }
  DeviceGuard dg(t->device);
  hipDeviceSynchronize();
  for (auto& c : t->chunks) { hipFree(c.rows); hipFree(c.meta); }
  hipFree(t->entries); hipFree(t->d_chunks); hipFree(t->d_counters); hipFree(t->d_stat); hipFree(t->free_rows);
  hipFree(t->init_table);
  hipFree(t->route_hist);
  for (auto e : t->ev) hipEventDestroy(e);
  Workspace& w = t->ws;
  hipFree(w.ent_key); hipFree(w.ent_a); hipFree(w.ent_b); hipFree(w.ent_base); hipFree(w.ent_rec); hipFree(w.toff); hipFree(w.slot_rank);
  hipFree(w.order); hipFree(w.coldlist); hipFree(w.hotlist); hipFree(w.litem); hipFree(w.items); hipFree(w.pmeta); hipFree(w.hpart);
  hipFree(w.mcount); hipFree(w.epart); hipFree(w.pos_ent);
  hipFree(w.ctr); hipFree(w.dbg); hipFree(w.scat_keys); hipFree(w.scat_sum); hipFree(w.seg_off);
  if (t->err_host) hipHostFree(t->err_host);
  if (t->cnt_host) hipHostFree(t->cnt_host);
  if (t->last_done) hipEventDestroy(t->last_done);
  delete t;
  return KV_OK;
}

// A lookup's deferred passes hold a snapshot of the table's arrays (pend_pa): whatever moves or frees those arrays, or
// changes what the passes would compute (seed, deterministic order), first lets them run — on the stream of the
// table's last op — and waits for them.
static int settle_pending(kv_table* t) {
  if (!t->part_pending) return KV_OK;
  hipStream_t s = t->has_last ? t->last_stream : nullptr;
  int rc;
  if ((rc = join_side(t, s))) return rc;
  HIP_TRY(hipStreamSynchronize(s));
  return KV_OK;
}

int kv_reserve(kv_handle_t t, int64_t capacity) {
  int rc;
  if ((rc = check_table(t))) return rc;
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  if ((rc = settle_pending(t))) return rc;
  unsigned long long save = t->rows_ub, save_idx = t->idx_ub;
  long long extra = capacity + 1 - (long long)t->rows_ub;
  if (extra <= 0) return KV_OK;
  rc = ensure_capacity(t, extra, nullptr);
  t->rows_ub = std::min(save, t->rows_ub);  // reserve does not consume the bounds
  t->idx_ub = std::min(save_idx, t->idx_ub);
  return rc;
}

int kv_init_table(kv_handle_t t, const float* table, int64_t rows, kv_stream_t stream) {
  int rc;
  if ((rc = check_table(t))) return rc;
  if (!table || rows <= 0) return fail(KV_INVALID_ARGUMENT, "random_initializer must be a non-empty [rows, dim] matrix");
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  // "re-initialization ignored" once a table is set (random_init_table_.NumElements() > 0, kv_variable.h:188-193);
  // the zero row an import leaves in place of a missing init table is not one
  if (t->initialized && t->init_table && !t->init_placeholder) return KV_OK;
  if ((rc = settle_pending(t))) return rc;   // (a pending pass reads the placeholder the next line frees)
  if (t->init_table) { HIP_TRY(hipStreamSynchronize((hipStream_t)stream)); hipFree(t->init_table); t->init_table = nullptr; }
  HIP_TRY(hipMalloc(&t->init_table, (size_t)rows * t->dim * sizeof(float)));
  HIP_TRY(hipMemcpyAsync(t->init_table, table, (size_t)rows * t->dim * sizeof(float),
                         hipMemcpyDeviceToDevice, (hipStream_t)stream));
  t->init_rows = rows;
  t->init_placeholder = false;
  t->initialized = true;
  return KV_OK;
}

int kv_is_initialized(kv_handle_t t, int* out) {
  int rc;
  if ((rc = check_table(t))) return rc;
  *out = t->initialized ? 1 : 0;
  return KV_OK;
}

int kv_set_clock_days(kv_handle_t t, int day) {
  int rc;
  if ((rc = check_table(t))) return rc;
  t->fixed_day = day;
  ++t->op_serial;   // what a timed delete would release depends on the day
  return KV_OK;
}
int kv_set_seed(kv_handle_t t, uint64_t seed) {
  int rc;
  if ((rc = check_table(t))) return rc;
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  if ((rc = settle_pending(t))) return rc;   // rows a pending pass initialises follow the seed the lookup answered with
  t->seed = seed;
  return KV_OK;
}

static int stats(kv_handle_t t, hipStream_t s, unsigned long long out[2], unsigned* nrows_out) {
  { const int jr = join_side(t, s); if (jr) return jr; }
  unsigned c[3];
  HIP_TRY(hipMemcpyAsync(c, t->d_counters, sizeof c, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (c[1]) return flagged_error(t, c[1], s);
  t->rows_ub = c[0];
  t->free_known = std::max(0, (int)c[2]);
  if (nrows_out) *nrows_out = c[0];
  if (out) {
    HIP_TRY(hipMemsetAsync(t->d_stat, 0, 4 * sizeof(unsigned long long), s));
    k_stats<<<nblocks(c[0], TB, 2048), TB, 0, s>>>(dev_view(t), c[0], t->d_stat);
    HIP_TRY(hipMemcpyAsync(out, t->d_stat, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  return KV_OK;
}

int kv_size(kv_handle_t t, int64_t* out, kv_stream_t stream) {
  int rc;
  if ((rc = check_table(t))) return rc;
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  unsigned long long o[2];
  if ((rc = stats(t, (hipStream_t)stream, o, nullptr))) return rc;
  *out = (int64_t)o[0];
  return KV_OK;
}
int kv_sum_freq(kv_handle_t t, int64_t* out, kv_stream_t stream) {
  int rc;
  if ((rc = check_table(t))) return rc;
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  unsigned long long o[2];
  if ((rc = stats(t, (hipStream_t)stream, o, nullptr))) return rc;
  *out = (int64_t)o[1];
  return KV_OK;
}
int kv_map_size(kv_handle_t t, int64_t* out, kv_stream_t stream) {
  int rc;
  if ((rc = check_table(t))) return rc;
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  unsigned nrows = 1;
  if ((rc = stats(t, (hipStream_t)stream, nullptr, &nrows))) return rc;
  *out = (int64_t)nrows - 1 - t->free_known;
  return KV_OK;
}

int kv_get_meta(kv_handle_t t, const int64_t* ids, int64_t n, uint32_t* fw, uint8_t* fl, kv_stream_t stream) {
  int rc;
  if ((rc = check_table(t))) return rc;
  if (n <= 0) return KV_OK;
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  if ((rc = join_side(t, (hipStream_t)stream))) return rc;
  k_get_meta<long long><<<nblocks(n, TB), TB, 0, (hipStream_t)stream>>>(dev_view(t), (const long long*)ids, n, fw, fl);
  HIP_TRY(hipGetLastError());
  return KV_OK;
}

static int gather_or_insert_impl(kv_handle_t t, const void* ids, const int32_t* counts, int64_t n, float* out,
                                 kv_stream_t stream, int pairs, kv_batch_token_t* token, unsigned seg_cap = 0);

int kv_gather_or_insert(kv_handle_t t, const void* ids, const int32_t* counts, int64_t n, float* out,
                        kv_stream_t stream) {
  return gather_or_insert_impl(t, ids, counts, n, out, stream, 0, nullptr);
}
int kv_gather_or_insert_tok(kv_handle_t t, const void* ids, const int32_t* counts, int64_t n, float* out,
                            kv_batch_token_t* token, kv_stream_t stream) {
  if (token) *token = 0;
  return gather_or_insert_impl(t, ids, counts, n, out, stream, 0, token);
}
int kv_gather_or_insert_pairs(kv_handle_t t, const int64_t* id_count_pairs, int64_t n, float* out,
                              kv_stream_t stream) {
  if (t && t->key_dtype == KV_DT_INT32) return fail(KV_INVALID_ARGUMENT, "id/count pairs carry int64 ids");
  return gather_or_insert_impl(t, id_count_pairs, nullptr, n, out, stream, 1, nullptr);
}

static int gather_or_insert_impl(kv_handle_t t, const void* ids, const int32_t* counts, int64_t n, float* out,
                                 kv_stream_t stream, int pairs, kv_batch_token_t* token, unsigned seg_cap) {
  int rc;
  if ((rc = check_table(t))) return rc;
  if (n == 0) return KV_OK;  // kv_variable_ops.cc:530-532
  if (n < 0 || n > (1ll << 30)) return fail(KV_INVALID_ARGUMENT, "indices: bad length %lld", (long long)n);
  if (!ids || !out) return fail(KV_INVALID_ARGUMENT, "indices / output pointer is null");
  if (!t->initialized)
    return fail(KV_FAILED_PRECONDITION, "Failed to use uninitialized variables: KvVariable init table not set");
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  MirrorKeep mk(t);   // a training lookup touches the var's rows and records, never a slot record or a mirror
  hipStream_t s = (hipStream_t)stream;
  if ((rc = enter_op(t, s))) return rc;
  // any batch length: chunks of 2^21 ids are looked up one after another (same semantics as one
  // pass: the frequency adds saturate identically and rows are inserted by the first chunk)
  // the entry-list pipeline indexes a batch of up to FUSED_MAX_N ids in one pass; the sorted-position one 2^21
  const long long CHK = fused_tab(t) ? FUSED_MAX_N : (1ll << 21);
  const size_t idsz = pairs ? 16 : (t->key_dtype == KV_DT_INT32 ? 4 : 8);
  t->batch_serial = 0;
  for (long long off = 0; off < n; off += CHK) {
    const long long m = std::min(CHK, (long long)n - off);
    const void* idp = (const char*)ids + (size_t)off * idsz;
    const int32_t* cp = counts ? counts + off : nullptr;
    float* op = out + (size_t)off * t->dim;
    if ((rc = ensure_capacity(t, m, s))) return rc;
    if ((rc = ensure_workspace(t, m, false, s))) return rc;
    const TableDev td = dev_view(t);
    WsDev wd = ws_view(t, m);
    wd.seg_cap = seg_cap;
    PartArgs pa{};
    pa.tv = td; pa.ts0 = td; pa.ts1 = td;
    pa.day = today(t);
    pa.det = det_mode(t);
    pa.n = m;
    const bool defer_part = token != nullptr && n <= CHK;   // a token is asked for: an apply of this batch follows
    if (fused_tab(t)) { if ((rc = fused_lookup_pass(t, wd, pa, idp, cp, m, pairs ? 2 : -1, op, s, defer_part))) return rc; }
    else index_pass<MODE_LOOKUP>(t, wd, pa, idp, cp, m, pairs ? 2 : -1, op, s, token != nullptr && n <= CHK);
  }
  HIP_TRY(hipGetLastError());
  if (token && n <= CHK) {   // the workspace now holds the index of exactly this batch, positions filed
    t->batch_serial = ++g_serial;
    t->batch_n = n;
    if (token) *token = t->batch_serial;
  }
  return KV_OK;
}

int kv_lookup_sparse(kv_handle_t t, const void* ids, const void* segment_ids, int segment_dtype,
                     const float* weights, int64_t n, int64_t num_segments, int combiner, int count_occurrences,
                     float* out, kv_stream_t stream) {
  int rc;
  if ((rc = check_table(t))) return rc;
  if (combiner < KV_COMBINER_SUM || combiner > KV_COMBINER_SQRTN)
    return fail(KV_INVALID_ARGUMENT, "combiner must be one of 'mean', 'sqrtn' or 'sum'");  // embedding_ops.py:345
  if (segment_dtype != KV_DT_INT32 && segment_dtype != KV_DT_INT64)
    return fail(KV_INVALID_ARGUMENT, "segment ids must be int32 or int64");
  const bool fused = fused_tab(t);   // (dim is fixed at creation: readable without the lock)
  if (n < 0 || n > (fused ? FUSED_MAX_N : (1ll << 21)))
    return fail(KV_INVALID_ARGUMENT, "sp_ids: %lld values (at most 2^%d per call)", (long long)n, fused ? 23 : 21);
  if (num_segments < 0 || num_segments > (1ll << 31) - 2) return fail(KV_INVALID_ARGUMENT, "bad num_segments");
  if (num_segments == 0) return KV_OK;
  if (!out || (n > 0 && (!ids || !segment_ids))) return fail(KV_INVALID_ARGUMENT, "ids / segment ids / output pointer is null");
  if (!t->initialized)
    return fail(KV_FAILED_PRECONDITION, "Failed to use uninitialized variables: KvVariable init table not set");
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  MirrorKeep mk(t);   // the table's own rows and records only
  hipStream_t s = (hipStream_t)stream;
  if ((rc = enter_op(t, s))) return rc;
  const int D = t->dim;
  if (n == 0) {  // every segment is empty
    HIP_TRY(hipMemsetAsync(out, 0, (size_t)num_segments * D * sizeof(float), s));
    return KV_OK;
  }
  if ((rc = ensure_capacity(t, n, s))) return rc;
  if ((rc = ensure_workspace(t, n, false, s))) return rc;
  Workspace& ws = t->ws;
  if (ws.seg_cap < num_segments) {
    if ((rc = ws_sync(s))) return rc;
    const long long want = std::max<long long>(num_segments, ws.seg_cap * 2);
    ws.seg_cap = 0;
    if ((rc = regrow(&ws.seg_off, (size_t)(want + 1)))) return rc;
    ws.seg_cap = want;
  }
  if (fused && ws.pos_cap < n) {   // every position's entry in its tile (k_ltile files it for the combiner)
    if ((rc = ws_sync(s))) return rc;
    ws.pos_cap = 0;
    if ((rc = regrow(&ws.pos_ent, (size_t)std::max<long long>(n, ws.cap_n)))) return rc;
    ws.pos_cap = std::max<long long>(n, ws.cap_n);
  }
  const TableDev td = dev_view(t);
  WsDev wd = ws_view(t, n);
  PartArgs pa{};
  pa.tv = td; pa.ts0 = td; pa.ts1 = td;
  pa.day = today(t);
  pa.count_once = count_occurrences ? 0 : 1;
  pa.det = det_mode(t);
  pa.n = n;
  t->batch_serial = 0;
  if (fused) {
    // the entry-list kernels: tile pass without rows (entries, every position's entry), the lookup's bookkeeping (which
    // also publishes the rows of new keys), then the combiner reads position -> entry -> row
    wd.pos_ent = ws.pos_ent;
    if ((rc = fused_lookup_pass(t, wd, pa, ids, nullptr, n, -1, nullptr, s, false))) return rc;
  } else {
    {
      ProfScope ps(t, KV_PROF_LOOKUP_TILE, s);
      launch_tile<false>(t, wd, ids, nullptr, n, s);
    }
    ProfScope ps(t, KV_PROF_LOOKUP_PART, s);
    launch_part_keys<MODE_LOOKUP>(wd, pa, s);
  }
  ProfScope ps_gather(t, KV_PROF_LOOKUP_ORDER, s);
  if (segment_dtype == KV_DT_INT32)
    k_seg_offsets<int><<<nblocks(n + 1, TB, 2048), TB, 0, s>>>((const int*)segment_ids, n, num_segments, ws.seg_off);
  else
    k_seg_offsets<long long><<<nblocks(n + 1, TB, 2048), TB, 0, s>>>((const long long*)segment_ids, n, num_segments, ws.seg_off);
  if (fused) {
    const int ql = row_lanes(D);
    const int grid = nblocks(num_segments * ql, TB, 8192);
#define KV_SCE(VQ) k_seg_combine_e<VQ><<<grid, TB, 0, s>>>(td, ws.pos_ent, wd.ent_b, wd.ent_key, ws.seg_off, weights, num_segments, combiner, out)
    switch (ql) {
      case 1: KV_SCE(1); break;   case 2: KV_SCE(2); break;   case 4: KV_SCE(4); break;   case 8: KV_SCE(8); break;
      case 16: KV_SCE(16); break; case 32: KV_SCE(32); break; default: KV_SCE(64); break;
    }
#undef KV_SCE
    HIP_TRY(hipGetLastError());
    return KV_OK;
  }
  const int q = (D % 4 == 0) ? D / 4 : 0;
  const bool vec = q > 0 && (q & (q - 1)) == 0 && q <= 64;
  const int grid = nblocks(num_segments * (vec ? q : 1), TB, 8192);
  switch (vec ? q : 0) {
    case 1: k_seg_combine<1><<<grid, TB, 0, s>>>(td, wd, ws.seg_off, weights, num_segments, combiner, out); break;
    case 2: k_seg_combine<2><<<grid, TB, 0, s>>>(td, wd, ws.seg_off, weights, num_segments, combiner, out); break;
    case 4: k_seg_combine<4><<<grid, TB, 0, s>>>(td, wd, ws.seg_off, weights, num_segments, combiner, out); break;
    case 8: k_seg_combine<8><<<grid, TB, 0, s>>>(td, wd, ws.seg_off, weights, num_segments, combiner, out); break;
    case 16: k_seg_combine<16><<<grid, TB, 0, s>>>(td, wd, ws.seg_off, weights, num_segments, combiner, out); break;
    case 32: k_seg_combine<32><<<grid, TB, 0, s>>>(td, wd, ws.seg_off, weights, num_segments, combiner, out); break;
    case 64: k_seg_combine<64><<<grid, TB, 0, s>>>(td, wd, ws.seg_off, weights, num_segments, combiner, out); break;
    default: k_seg_combine<0><<<grid, TB, 0, s>>>(td, wd, ws.seg_off, weights, num_segments, combiner, out); break;
  }
  HIP_TRY(hipGetLastError());
  return KV_OK;
}

int kv_gather_or_zeros(kv_handle_t t, const void* ids, int64_t n, float* out, kv_stream_t stream) {
  int rc;
  if ((rc = check_table(t))) return rc;
  if (!t->initialized)   // FindOrZeros -> CheckInitializedInternal (kv_variable.h:242)
    return fail(KV_FAILED_PRECONDITION, "Failed to use uninitialized variables: KvVariable init table not set");
  if (n == 0) return KV_OK;
  if (n < 0 || !ids || !out) return fail(KV_INVALID_ARGUMENT, "indices / output pointer is null");
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  hipStream_t s = (hipStream_t)stream;
  MirrorKeep mk(t);   // (the inference gather reads rows and flags of THIS table: as a var it leaves the mirrors alone)
  if ((rc = hand_over(t, s, true, false))) return rc;   // a read: behind the table's last op on whatever stream, no serial bump
  const TableDev td = dev_view(t);
  const int q = t->dim / 4;
  const bool wave_shaped = (t->dim & 3) == 0 && q >= 1 && q <= 64 && (q & (q - 1)) == 0;
  const int gw = nblocks(n, TB, 8192);  // a 64-id step per wave at 1 M ids: residency hides the hops
#define KV_GOZ(IDT, VQ) k_gather_or_zeros_w<IDT, VQ><<<gw, TB, 0, s>>>(td, (const IDT*)ids, out, n)
#define KV_GOZ_ALL(IDT)                                                                        \
  switch (q) {                                                                                 \
    case 1: KV_GOZ(IDT, 1); break;   case 2: KV_GOZ(IDT, 2); break;   case 4: KV_GOZ(IDT, 4); break;    \
    case 8: KV_GOZ(IDT, 8); break;   case 16: KV_GOZ(IDT, 16); break; case 32: KV_GOZ(IDT, 32); break;  \
    default: KV_GOZ(IDT, 64); break;                                                           \
  }
  if (wave_shaped) {
    if (t->key_dtype == KV_DT_INT32) { KV_GOZ_ALL(int) } else { KV_GOZ_ALL(long long) }
  } else if (t->key_dtype == KV_DT_INT32) {
    k_gather_or_zeros<int><<<nblocks(n, TB / 8, 8192), TB, 0, s>>>(td, (const int*)ids, out, n);
  } else {
    k_gather_or_zeros<long long><<<nblocks(n, TB / 8, 8192), TB, 0, s>>>(td, (const long long*)ids, out, n);
  }
#undef KV_GOZ_ALL
#undef KV_GOZ
  HIP_TRY(hipGetLastError());
  return KV_OK;
}

// descriptor staging for the batched launches: one pinned host buffer + device buffer per device
// and descriptor kind; the next upload waits until the previous launch has consumed the buffer
namespace {
struct StageSlot {
  char* host = nullptr;   // pinned
  char* dev = nullptr;
  size_t cap = 0;
  hipEvent_t consumed = nullptr;
};
struct BatchStage {       // a small ring, so the host can prepare call k+1 while call k still runs
  std::mutex mu;
  StageSlot slot[4];
  unsigned cursor = 0;
};
BatchStage g_stage[64][2];   // [device][0 = inference gather, 1 = training ops]

// returns with st.mu HELD (released by StageRelease after `consumed` is recorded on the stream)
int stage_acquire(BatchStage& st, size_t bytes, StageSlot** out) {
  st.mu.lock();
  StageSlot& sl = st.slot[st.cursor++ & 3u];
  if (sl.consumed && hipEventSynchronize(sl.consumed) != hipSuccess) {
    st.mu.unlock();
    return fail(KV_INTERNAL, "descriptor staging: event sync failed");
  }
  if (sl.cap < bytes) {
    if (sl.host) hipHostFree(sl.host);
    if (sl.dev) hipFree(sl.dev);
    sl.host = sl.dev = nullptr;
    sl.cap = std::max<size_t>(bytes, 64 * 1024);
    if (hipHostMalloc(&sl.host, sl.cap) != hipSuccess || hipMalloc(&sl.dev, sl.cap) != hipSuccess ||
        (!sl.consumed && hipEventCreateWithFlags(&sl.consumed, hipEventDisableTiming) != hipSuccess)) {
      sl.cap = 0;
      st.mu.unlock();
      return fail(KV_RESOURCE_EXHAUSTED, "descriptor staging: allocation failed");
    }
  }
  *out = &sl;
  return KV_OK;
}
struct StageRelease {   // unlocks (and marks the slot busy until the stream gets there) on scope exit
  BatchStage& st; StageSlot* sl; hipStream_t s; bool launched = false;
  ~StageRelease() { if (launched) hipEventRecord(sl->consumed, s); st.mu.unlock(); }
};

int check_same_shape(int num_tables, const kv_handle_t* tables, const char* what) {
  int rc;
  if (num_tables < 1) return fail(KV_INVALID_ARGUMENT, "N must be >= 1");
  if (!tables) return fail(KV_INVALID_ARGUMENT, "null argument array");
  for (int i = 0; i < num_tables; ++i) {
    if ((rc = check_table(tables[i]))) return rc;
    if (tables[i]->device != tables[0]->device) return fail(KV_INVALID_ARGUMENT, "%s live on different devices", what);
  }
  if (tables[0]->device < 0 || tables[0]->device >= 64) return fail(KV_INVALID_ARGUMENT, "device index");
  return KV_OK;
}
}  // namespace

int kv_batch_gather_or_zeros(int num_tables, const kv_handle_t* tables, const void* const* ids,
                             const int64_t* ns, float* const* outs, kv_stream_t stream) {
  int rc;
  if ((rc = check_same_shape(num_tables, tables, "tables"))) return rc;  // Attr("N: int >= 1")
  if (!ids || !ns || !outs) return fail(KV_INVALID_ARGUMENT, "null argument array");
  for (int i = 0; i < num_tables; ++i) {
    if (ns[i] < 0 || (ns[i] > 0 && (!ids[i] || !outs[i]))) return fail(KV_INVALID_ARGUMENT, "indices / output pointer is null");
    if (!tables[i]->initialized)   // FindOrZeros -> CheckInitializedInternal (kv_variable.h:242)
      return fail(KV_FAILED_PRECONDITION, "Failed to use uninitialized variables: KvVariable init table not set");
  }
  const int device = tables[0]->device;
  DeviceGuard dg(device);
  hipStream_t s = (hipStream_t)stream;
  MultiLock lock(std::vector<kv_table*>(tables, tables + num_tables));
  MirrorKeep mk(tables, num_tables);
  // every table is read on the op's stream: behind whatever its own last op queued on another stream (an optimizer
  // apply that has not finished), and its next op behind this read
  for (kv_table* tb : lock.ts)
    if ((rc = report_deferred_error(tb, s)) || (rc = hand_over(tb, s, true, false))) return rc;
  BatchStage& st = g_stage[device][0];
  StageSlot* sl = nullptr;
  if ((rc = stage_acquire(st, (size_t)num_tables * sizeof(BatchGatherDesc), &sl))) return rc;
  StageRelease rel{st, sl, s};
  BatchGatherDesc* hd = reinterpret_cast<BatchGatherDesc*>(sl->host);
  long long nmax = 0;
  for (int i = 0; i < num_tables; ++i) {
    BatchGatherDesc& d = hd[i];
    d.t = dev_view(tables[i]);
    d.ids = ids[i];
    d.out = outs[i];
    d.n = ns[i];
    d.ids_int32 = tables[i]->key_dtype == KV_DT_INT32;
    nmax = std::max<long long>(nmax, ns[i]);
  }
  if (nmax == 0) return KV_OK;
  HIP_TRY(hipMemcpyAsync(sl->dev, sl->host, (size_t)num_tables * sizeof(BatchGatherDesc), hipMemcpyHostToDevice, s));
  rel.launched = true;
  dim3 grid((unsigned)nblocks(nmax, TB / 8, 2048), (unsigned)num_tables);
  k_batch_gather_or_zeros<<<grid, TB, 0, s>>>(reinterpret_cast<const BatchGatherDesc*>(sl->dev));
  HIP_TRY(hipGetLastError());
  return KV_OK;
}

static bool claim_slot(kv_table* v, kv_table* sl, hipStream_t s);

// ---- many tables, one launch per pipeline stage (26-feature CTR step: 5 launches, not 130) ------
// All tables: same device, same dim, same key dtype; each batch <= 2^21 ids.
static int multi_common(int num_tables, const kv_handle_t* tables, const void* const* ids, const int64_t* ns) {
  int rc;
  if ((rc = check_same_shape(num_tables, tables, "tables"))) return rc;
  if (!ids || !ns) return fail(KV_INVALID_ARGUMENT, "null argument array");
  for (int i = 0; i < num_tables; ++i) {
    if (tables[i]->dim != tables[0]->dim || tables[i]->key_dtype != tables[0]->key_dtype)
      return fail(KV_INVALID_ARGUMENT, "batched op: tables must share dim and key dtype (group them by shape)");
    if (tables[i]->occurrence_order)
      return fail(KV_UNIMPLEMENTED, "batched op: a table in occurrence-order mode (kv_set_deterministic(h, 2)) takes the per-table ops");
    // (the entry-list kernels index up to FUSED_MAX_N ids per table and call, like the single-table ops; other dims 2^21)
    if (ns[i] < 0 || ns[i] > (fused_tab(tables[0]) ? FUSED_MAX_N : (1ll << 21)))
      return fail(KV_INVALID_ARGUMENT, "indices: bad length %lld", (long long)ns[i]);
    if (ns[i] > 0 && !ids[i]) return fail(KV_INVALID_ARGUMENT, "indices pointer is null");
    if (!tables[i]->initialized)
      return fail(KV_FAILED_PRECONDITION, "Failed to use uninitialized variables: KvVariable init table not set");
    for (int j = 0; j < i; ++j)
      if (tables[j] == tables[i]) return fail(KV_INVALID_ARGUMENT, "batched op: table listed twice");
  }
  return KV_OK;
}

int kv_multi_gather_or_insert(int num_tables, const kv_handle_t* tables, const void* const* ids,
                              const int32_t* const* counts, const int64_t* ns, float* const* outs,
                              kv_stream_t stream) {
  return kv_multi_gather_or_insert_tok(num_tables, tables, ids, counts, ns, outs, nullptr, stream);
}

// ids_kind 2 + seg_caps: the (id, count) records of the sharded owner lookups, in fixed-capacity segments (kv_multi_shard_lookup)
static int multi_lookup_impl(int num_tables, const kv_handle_t* tables, const void* const* ids,
                             const int32_t* const* counts, const int64_t* ns, float* const* outs,
                             kv_batch_token_t* tokens, kv_stream_t stream, int ids_kind, const unsigned* seg_caps);
int kv_multi_gather_or_insert_tok(int num_tables, const kv_handle_t* tables, const void* const* ids,
                                  const int32_t* const* counts, const int64_t* ns, float* const* outs,
                                  kv_batch_token_t* tokens, kv_stream_t stream) {
  return multi_lookup_impl(num_tables, tables, ids, counts, ns, outs, tokens, stream, -1, nullptr);
}
static int multi_lookup_impl(int num_tables, const kv_handle_t* tables, const void* const* ids,
                             const int32_t* const* counts, const int64_t* ns, float* const* outs,
                             kv_batch_token_t* tokens, kv_stream_t stream, int ids_kind, const unsigned* seg_caps) {
  int rc;
  if (tokens && num_tables > 0) std::memset(tokens, 0, (size_t)num_tables * sizeof(kv_batch_token_t));
  if ((rc = multi_common(num_tables, tables, ids, ns))) return rc;
  if (!outs) return fail(KV_INVALID_ARGUMENT, "null argument array");
  const int device = tables[0]->device;
  DeviceGuard dg(device);
  hipStream_t s = (hipStream_t)stream;
  MultiLock lock(std::vector<kv_table*>(tables, tables + num_tables));
  MirrorKeep mk(tables, num_tables);   // lookups: the tables' own rows and records only
  if ((rc = lock.enter(s))) return rc;
  long long nmax = 0;
  for (int i = 0; i < num_tables; ++i) {
    tables[i]->batch_serial = 0;
    if (ns[i] > 0 && !outs[i]) return fail(KV_INVALID_ARGUMENT, "output pointer is null");
    if ((rc = ensure_capacity(tables[i], ns[i], s))) return rc;
    if ((rc = ensure_workspace(tables[i], std::max<long long>(ns[i], 1), false, s))) return rc;
    nmax = std::max<long long>(nmax, ns[i]);
  }
  if (nmax == 0) return KV_OK;
  BatchStage& st = g_stage[device][1];
  StageSlot* sl = nullptr;
  if ((rc = stage_acquire(st, (size_t)num_tables * sizeof(MultiDesc), &sl))) return rc;
  StageRelease rel{st, sl, s};
  MultiDesc* hd = reinterpret_cast<MultiDesc*>(sl->host);
  WsDev wmax{};
  for (int i = 0; i < num_tables; ++i) {
    MultiDesc& d = hd[i];
    std::memset(&d, 0, sizeof d);
    d.w = ws_view(tables[i], std::max<long long>(ns[i], 1));
    if (seg_caps) d.w.seg_cap = seg_caps[i];
    if (fused_tab(tables[i])) { d.w.P = fused_default_P(std::max<long long>(ns[i], 1)); d.w.pshift = 64 - ilog2(d.w.P); }
    d.a.tv = dev_view(tables[i]); d.a.ts0 = d.a.tv; d.a.ts1 = d.a.tv;
    d.a.day = today(tables[i]);
    d.a.det = tables[i]->deterministic ? 1 : 0;
    d.a.n = ns[i];
    d.ids = ids[i];
    d.counts = counts ? counts[i] : nullptr;
    d.out = outs[i];
    d.n = ns[i];
    if (ns[i] == 0) d.w.ntiles = 0;
    wmax.ntiles = std::max(wmax.ntiles, d.w.ntiles);
    wmax.P = std::max(wmax.P, d.w.P);
  }
  HIP_TRY(hipMemcpyAsync(sl->dev, sl->host, (size_t)num_tables * sizeof(MultiDesc), hipMemcpyHostToDevice, s));
  rel.launched = true;
  const MultiDesc* md = reinterpret_cast<const MultiDesc*>(sl->dev);
  kv_table* t0 = tables[0];
  if (fused_tab(t0)) {
    for (int i = 0; i < num_tables; ++i) tables[i]->fused_index = true;
    launch_ltile(t0, hd[0].a.tv, wmax, nullptr, nullptr, nmax, nullptr, s, ids_kind, md, num_tables, true);
    // tokens asked for: an optimizer apply of these batches follows — every table's partition pass stays pending
    // (kv_multi_apply_*_tok completes it inside k_papply_multi; any other op on a table settles that table first)
    const bool defer = tokens != nullptr;
    if (!defer) launch_part2(wmax, hd[0].a, s, md, num_tables);
    if (tokens)   // every table's workspace now holds the index of exactly its batch (kv_multi_apply_*_tok takes it over)
      for (int i = 0; i < num_tables; ++i) {
        if (ns[i] <= 0) continue;
        tables[i]->batch_serial = ++g_serial;
        tables[i]->batch_n = ns[i];
        tables[i]->index_P = hd[i].w.P;
        tokens[i] = tables[i]->batch_serial;
        if (defer) {
          std::memcpy(tables[i]->pend_wd, &hd[i].w, sizeof(WsDev));
          std::memcpy(tables[i]->pend_pa, &hd[i].a, sizeof(PartArgs));
          tables[i]->part_pending = true;
        }
      }
  } else {
    for (int i = 0; i < num_tables; ++i) tables[i]->fused_index = false;
    launch_tile<false>(t0, wmax, nullptr, nullptr, nmax, s, -1, md, num_tables, wmax.ntiles);
    launch_part_keys<MODE_LOOKUP>(wmax, hd[0].a, s, md, num_tables);
    launch_gather(hd[0].a.tv, wmax, nullptr, nmax, s, md, num_tables);
  }
  HIP_TRY(hipGetLastError());
  return KV_OK;
}

// shared body of the batched optimizer ops: opt = OPT_*; slots1 only for FTRL (linear); slot_mult =
// slot dim / var dim
static int multi_apply_common(int num_tables, const kv_handle_t* vars, const kv_handle_t* slots0,
                              const kv_handle_t* slots1, int slot_mult, const float* const* grads,
                              const void* const* ids, const int64_t* ns, const OptArgs& a, int opt,
                              kv_stream_t stream, const kv_batch_token_t* tokens = nullptr) {
  int rc;
  if ((rc = multi_common(num_tables, vars, ids, ns))) return rc;
  if ((rc = check_same_shape(num_tables, slots0, "slot tables"))) return rc;
  if (slots1 && (rc = check_same_shape(num_tables, slots1, "slot tables"))) return rc;
  if (!grads) return fail(KV_INVALID_ARGUMENT, "null argument array");
  const int D = vars[0]->dim;
  if ((D & 3) != 0 || !dim_supported(D))
    return fail(KV_UNIMPLEMENTED, "batched optimizer op: embedding dim %d (multiples of 4 up to 1024)", D);
  std::vector<kv_table*> all;
  for (int i = 0; i < num_tables; ++i) {
    for (const kv_handle_t* sl : {slots0, slots1}) {
      if (!sl) continue;
      if (!sl[i]->initialized) return fail(KV_FAILED_PRECONDITION, "Failed to use uninitialized variables: optimizer slot");
      if (sl[i]->dim != slot_mult * D || sl[i]->device != vars[0]->device || sl[i]->key_dtype != vars[0]->key_dtype)
        return fail(KV_INVALID_ARGUMENT, "var and slot do not have matching shapes (slot dim must be %d x var dim, same device / key dtype)", slot_mult);
      all.push_back(sl[i]);
    }
    if (ns[i] > 0 && !grads[i]) return fail(KV_INVALID_ARGUMENT, "grad pointer is null");
    all.push_back(vars[i]);
  }
  {
    std::vector<kv_table*> u(all);
    std::sort(u.begin(), u.end());
    if (std::adjacent_find(u.begin(), u.end()) != u.end())
      return fail(KV_INVALID_ARGUMENT, "batched op: a table is listed twice (var or slot)");
  }
  const int device = vars[0]->device;
  DeviceGuard dg(device);
  hipStream_t s = (hipStream_t)stream;
  MultiLock lock(all);
  // GroupAdam / Adagrad over pairs (var_i, slot_i): the lean update works on the var rows' slot mirrors (mirror_decide per
  // table below); FTRL reads and writes the slot tables' own records: its entry ends the tables' epochs
  MirrorKeep mk(vars, opt != OPT_FTRL ? num_tables : 0, slots0, opt != OPT_FTRL ? num_tables : 0);
  if (tl_unique && fused_ok(D) && !stream_is_capturing(s)) {
    // The caller promises that no table's ids hold an id twice (kv_multi_apply_*_unique; kv_uapply.h): ONE launch for all
    // tables, one lane group per id (grid.y = table).  Pending lookup passes are settled first.
    long long nmax = 0;
    for (kv_table* tb : lock.ts)
      if ((rc = report_deferred_error(tb, s)) || (rc = hand_over(tb, s))) return rc;
    for (int i = 0; i < num_tables; ++i) {
      if ((rc = ensure_capacity(vars[i], ns[i], s)) || (rc = ensure_capacity(slots0[i], ns[i], s)) ||
          (slots1 && (rc = ensure_capacity(slots1[i], ns[i], s))))
        return rc;
      nmax = std::max<long long>(nmax, ns[i]);
    }
    if (nmax == 0) return KV_OK;
    BatchStage& st = g_stage[device][1];
    StageSlot* sl = nullptr;
    if ((rc = stage_acquire(st, (size_t)num_tables * sizeof(MultiDesc), &sl))) return rc;
    StageRelease rel{st, sl, s};
    MultiDesc* hd = reinterpret_cast<MultiDesc*>(sl->host);
    for (int i = 0; i < num_tables; ++i) {
      kv_table* v = vars[i];
      if (v->uniq_serial >= 65535u) {   // the 16-bit stamp wraps: every row back to "none"
        k_clear_stamps<<<nblocks((long long)v->rows_ub, TB, 4096), TB, 0, s>>>(dev_view(v), (unsigned)v->rows_ub);
        v->uniq_serial = 0;
      }
      MultiDesc& d = hd[i];
      std::memset(&d, 0, sizeof d);
      d.a.tv = dev_view(v); d.a.ts0 = dev_view(slots0[i]); d.a.ts1 = slots1 ? dev_view(slots1[i]) : d.a.ts0;
      d.a.opt = a; d.a.grad = grads[i]; d.a.day = today(v);
      d.a.opt.fast = fast_math_on(v) ? 1 : 0;
      d.a.n = ns[i];
      d.a.use_hints = claim_slot(v, slots0[i], s) ? 1 : 0;
      if ((rc = mirror_decide_rt(opt, v, slots0[i], d.a, true, s))) return rc;
      d.a.uniq_serial = ns[i] > 0 ? ++v->uniq_serial : 0u;
      d.ids = ids[i];
      d.n = ns[i];
    }
    HIP_TRY(hipMemcpyAsync(sl->dev, sl->host, (size_t)num_tables * sizeof(MultiDesc), hipMemcpyHostToDevice, s));
    rel.launched = true;
    const int ids32 = vars[0]->key_dtype == KV_DT_INT32 ? 1 : 0;
    ProfScope ps(vars[0], KV_PROF_APPLY_UNIQUE, s);
    rc = (opt == OPT_ADAM_V4 || opt == OPT_ADAM_V3) ? kvp_launch_uapply_a(opt, &hd[0].a, nullptr, ids32, nmax, (void*)s, sl->dev, num_tables)
                                                    : kvp_launch_uapply_b(opt, &hd[0].a, nullptr, ids32, nmax, (void*)s, sl->dev, num_tables);
    if (rc) return fail(rc, "batched unique apply: no kernel for dim %d", D);
    HIP_TRY(hipGetLastError());
    return KV_OK;
  }
  // The entry-list kernels (fused_ok): every table still holds the tiles' entries of its batch (kv_multi_gather_or_insert_tok)
  // and every token matches -> k_papply_multi over them: PA_LOOKUP when the lookups' partition passes are still pending (it
  // completes them together with the update), PA_NONE when they have been settled since.  One stale token and all tables
  // are indexed again (PA_APPLYIDX: one launch either way); pending passes that are not taken over are settled on entry.
  const bool fz = fused_ok(D);
  bool reuse = tokens != nullptr && fz;       // every table holds its batch's entries
  bool pa_reuse = reuse;                      // ... and its partition pass is still pending
  for (int i = 0; i < num_tables && reuse; ++i)
    if (ns[i] > 0) {
      const bool held = tokens[i] != 0 && tokens[i] == vars[i]->batch_serial && ns[i] == vars[i]->batch_n && vars[i]->fused_index;
      if (!held) reuse = false;
      if (!held || !vars[i]->part_pending) pa_reuse = false;
    }
  if (!reuse) pa_reuse = false;
  for (kv_table* tb : lock.ts) {
    bool keep = false;
    if (pa_reuse)
      for (int i = 0; i < num_tables; ++i) keep = keep || (vars[i] == tb && ns[i] > 0);
    if ((rc = report_deferred_error(tb, s)) || (rc = hand_over(tb, s, !keep))) return rc;
  }
  // (a table whose pass was pending while another's was not: hand_over has just settled it — the batch's entries stay valid)
  long long nmax = 0;
  if (tl_require_reuse && !reuse)
    return fail(KV_FAILED_PRECONDITION, "batched sharded apply: another op used a table since this batch's lookup");
  for (int i = 0; i < num_tables; ++i) {
    if (!reuse) vars[i]->batch_serial = 0;
    if (!reuse && (rc = ensure_capacity(vars[i], ns[i], s))) return rc;
    if ((rc = ensure_capacity(slots0[i], ns[i], s))) return rc;
    if (slots1 && (rc = ensure_capacity(slots1[i], ns[i], s))) return rc;
    if ((rc = ensure_workspace(vars[i], std::max<long long>(ns[i], 1), true, s))) return rc;
    nmax = std::max<long long>(nmax, ns[i]);
  }
  if (nmax == 0) return KV_OK;
  BatchStage& st = g_stage[device][1];
  StageSlot* sl = nullptr;
  if ((rc = stage_acquire(st, (size_t)num_tables * sizeof(MultiDesc), &sl))) return rc;
  StageRelease rel{st, sl, s};
  MultiDesc* hd = reinterpret_cast<MultiDesc*>(sl->host);
  WsDev wmax{};
  for (int i = 0; i < num_tables; ++i) {
    MultiDesc& d = hd[i];
    std::memset(&d, 0, sizeof d);
    d.w = ws_view(vars[i], std::max<long long>(ns[i], 1));
    d.a.tv = dev_view(vars[i]); d.a.ts0 = dev_view(slots0[i]); d.a.ts1 = slots1 ? dev_view(slots1[i]) : d.a.ts0;
    if (fz) { d.a.epart = d.w.epart; d.w.P = fused_default_P(std::max<long long>(ns[i], 1)); d.w.pshift = 64 - ilog2(d.w.P); }
    d.a.opt = a; d.a.grad = grads[i]; d.a.day = today(vars[i]);
    d.a.opt.fast = fast_math_on(vars[i]) ? 1 : 0;
    d.a.det = vars[i]->deterministic ? 1 : 0;
    d.a.n = ns[i];
    d.a.use_hints = claim_slot(vars[i], slots0[i], s) ? 1 : 0;
    if ((rc = mirror_decide_rt(opt, vars[i], slots0[i], d.a, fz, s))) return rc;   // (fz: k_papply_multi; else the sorted-position kernels, no mirrors)
    d.ids = ids[i];
    d.n = ns[i];
    if (ns[i] == 0) d.w.ntiles = 0;
    if (reuse && ns[i] > 0 && vars[i]->index_P) { d.w.P = vars[i]->index_P; d.w.pshift = 64 - ilog2(d.w.P); }   // the lookup's partitioning
    d.a.day_lk = d.a.day;
    if (pa_reuse && ns[i] > 0) {   // the pending lookup's own day stamp and counting rule
      PartArgs pend;
      std::memcpy(&pend, vars[i]->pend_pa, sizeof pend);
      d.a.day_lk = pend.day; d.a.count_once = pend.count_once;
      vars[i]->part_pending = false;
    }
    wmax.ntiles = std::max(wmax.ntiles, d.w.ntiles);
    wmax.P = std::max(wmax.P, d.w.P);
  }
  HIP_TRY(hipMemcpyAsync(sl->dev, sl->host, (size_t)num_tables * sizeof(MultiDesc), hipMemcpyHostToDevice, s));
  rel.launched = true;
  const MultiDesc* md = reinterpret_cast<const MultiDesc*>(sl->dev);
  if (fz) {
    // partition pass + update in one launch (k_papply_multi) behind the tile sums; an optimizer that meets the ids first
    // runs the tile pass of all tables in front (PA_APPLYIDX)
    int pa_mode = pa_reuse ? PA_LOOKUP : PA_NONE;
    if (!reuse) {
      for (int i = 0; i < num_tables; ++i) vars[i]->fused_index = true;
      launch_ltile(vars[0], hd[0].a.tv, wmax, nullptr, nullptr, nmax, nullptr, s, -1, md, num_tables, false);
      pa_mode = PA_APPLYIDX;
      for (int i = 0; i < num_tables; ++i)
        if (ns[i] > 0) { vars[i]->batch_serial = ++g_serial; vars[i]->batch_n = ns[i]; vars[i]->index_P = hd[i].w.P; }
    }
    if ((rc = kvp_launch_tsum(&hd[0].a.tv, &wmax, nullptr, (void*)s, md, num_tables)))
      return fail(rc, "tile sums: no kernel for dim %d", D);
    rc = (opt == OPT_ADAM_V4 || opt == OPT_ADAM_V3) ? kvp_launch_papply_a(opt, &wmax, &hd[0].a, pa_mode, (void*)s, md, num_tables)
                                                    : kvp_launch_papply_b(opt, &wmax, &hd[0].a, pa_mode, (void*)s, md, num_tables);
    if (rc) return fail(rc, "partition + apply pass: no kernel for dim %d", D);
    HIP_TRY(hipGetLastError());
    return KV_OK;
  }
  for (int i = 0; i < num_tables; ++i) vars[i]->fused_index = false;
  launch_tile<false>(vars[0], wmax, nullptr, nullptr, nmax, s, -1, md, num_tables, wmax.ntiles);
  launch_part_keys<MODE_APPLYIDX>(wmax, hd[0].a, s, md, num_tables);
  launch_order(hd[0].a.tv, wmax, nmax, s, md, num_tables);
  switch (opt) {
    case OPT_ADAM_V4: rc = launch_apply<MODE_APPLY, OPT_ADAM_V4>(vars[0], wmax, hd[0].a, nmax, s, md, num_tables); break;
    case OPT_ADAM_V3: rc = launch_apply<MODE_APPLY, OPT_ADAM_V3>(vars[0], wmax, hd[0].a, nmax, s, md, num_tables); break;
    case OPT_ADAGRAD: rc = launch_apply<MODE_APPLY, OPT_ADAGRAD>(vars[0], wmax, hd[0].a, nmax, s, md, num_tables); break;
    default: rc = launch_apply<MODE_APPLY, OPT_FTRL>(vars[0], wmax, hd[0].a, nmax, s, md, num_tables); break;
  }
  if (rc) return rc;
  HIP_TRY(hipGetLastError());
  return KV_OK;
}

int kv_multi_apply_group_adam(int num_tables, const kv_handle_t* vars, const kv_handle_t* slots,
                              const float* const* grads, const void* const* ids, const int64_t* ns, float lr,
                              float b1p, float b2p, float b1, float b2, float eps, float l1, float l2, float l21,
                              int version, kv_stream_t stream) {
  return kv_multi_apply_group_adam_tok(num_tables, vars, slots, grads, ids, ns, lr, b1p, b2p, b1, b2, eps, l1, l2, l21, version,
                                       nullptr, stream);
}

int kv_multi_apply_group_adam_tok(int num_tables, const kv_handle_t* vars, const kv_handle_t* slots,
                                  const float* const* grads, const void* const* ids, const int64_t* ns, float lr,
                                  float b1p, float b2p, float b1, float b2, float eps, float l1, float l2, float l21,
                                  int version, const kv_batch_token_t* tokens, kv_stream_t stream) {
  if (version != 3 && version != 4) return fail(KV_INVALID_ARGUMENT, "GroupAdam version %d: 3 or 4", version);
  if (!(lr > 0.f)) return fail(KV_INVALID_ARGUMENT, "lr is not a positive scalar: %g", lr);
  if (!(l1 >= 0.f)) return fail(KV_INVALID_ARGUMENT, "l1 regularization strength is not a non-negative scalar: %g", l1);
  if (!(l2 >= 0.f)) return fail(KV_INVALID_ARGUMENT, "l2 regularization strength is not a non-negative scalar: %g", l2);
  if (!(l21 >= 0.f)) return fail(KV_INVALID_ARGUMENT, "l21 regularization strength is not a non-negative scalar: %g", l21);
  if (num_tables < 1 || !vars || !vars[0]) return fail(KV_INVALID_ARGUMENT, "N must be >= 1");
  OptArgs a{};
  a.lr = lr; a.b1p = b1p; a.b2p = b2p; a.b1 = b1; a.b2 = b2; a.eps = eps;
  if (version == 4) {  // training_ops.cc:7111-7120
    a.l1 = l1 * lr; a.l2 = l2 * lr; a.l21 = l21 * lr;
    a.alpha = lr * std::sqrt(1.f - b2p) / (1.f - b1p);
  } else {             // :5840-5849
    a.l1 = l1; a.l2 = l2; a.l21 = l21;
    a.alpha = std::sqrt(1.f - b2p) / (1.f - b1p);
  }
  a.l21_norm = a.l21 * std::sqrt((float)vars[0]->dim);
  return multi_apply_common(num_tables, vars, slots, nullptr, 3, grads, ids, ns, a,
                            version == 4 ? OPT_ADAM_V4 : OPT_ADAM_V3, stream, tokens);
}

int kv_multi_apply_adagrad(int num_tables, const kv_handle_t* vars, const kv_handle_t* accums, float lr,
                           const float* const* grads, const void* const* ids, const int64_t* ns,
                           int update_slots, kv_stream_t stream) {
  return kv_multi_apply_adagrad_tok(num_tables, vars, accums, lr, grads, ids, ns, update_slots, nullptr, stream);
}

int kv_multi_apply_adagrad_tok(int num_tables, const kv_handle_t* vars, const kv_handle_t* accums, float lr,
                               const float* const* grads, const void* const* ids, const int64_t* ns,
                               int update_slots, const kv_batch_token_t* tokens, kv_stream_t stream) {
  OptArgs a{};
  a.lr = lr; a.update_slots = update_slots;
  return multi_apply_common(num_tables, vars, accums, nullptr, 1, grads, ids, ns, a, OPT_ADAGRAD, stream, tokens);
}

int kv_multi_apply_sparse_group_ftrl(int num_tables, const kv_handle_t* vars, const kv_handle_t* accums,
                                     const kv_handle_t* linears, const float* const* grads,
                                     const void* const* ids, const int64_t* ns, float lr, float l1, float l2,
                                     float l21, float l2s, float lr_power, kv_stream_t stream) {
  return kv_multi_apply_sparse_group_ftrl_tok(num_tables, vars, accums, linears, grads, ids, ns, lr, l1, l2, l21, l2s, lr_power,
                                              nullptr, stream);
}

int kv_multi_apply_sparse_group_ftrl_tok(int num_tables, const kv_handle_t* vars, const kv_handle_t* accums,
                                         const kv_handle_t* linears, const float* const* grads,
                                         const void* const* ids, const int64_t* ns, float lr, float l1, float l2,
                                         float l21, float l2s, float lr_power, const kv_batch_token_t* tokens,
                                         kv_stream_t stream) {
  if (!(lr > 0.f)) return fail(KV_INVALID_ARGUMENT, "lr is not a positive scalar: %g", lr);
  if (!(l1 >= 0.f)) return fail(KV_INVALID_ARGUMENT, "l1 regularization strength is not a non-negative scalar: %g", l1);
  if (!(l2 >= 0.f)) return fail(KV_INVALID_ARGUMENT, "l2 regularization strength is not a non-negative scalar: %g", l2);
  if (!(l21 >= 0.f)) return fail(KV_INVALID_ARGUMENT, "l21 regularization strength is not a non-negative scalar: %g", l21);
  if (!(lr_power <= 0.f)) return fail(KV_INVALID_ARGUMENT, "lr_power is not a non-positive scalar: %g", lr_power);
  if (!(l2s >= 0.f)) return fail(KV_INVALID_ARGUMENT, "l2 shrinkage regularization strength is not a non-negative scalar: %g", l2s);
  if (num_tables < 1 || !vars || !vars[0] || !linears) return fail(KV_INVALID_ARGUMENT, "N must be >= 1");
  OptArgs a{};
  a.lr = lr; a.l1 = l1; a.l2 = l2; a.l21 = l21; a.l2s = l2s; a.lr_power = lr_power;
  a.l21_norm = l21 * std::sqrt((float)vars[0]->dim);  // training_ops.cc:728
  return multi_apply_common(num_tables, vars, accums, linears, 1, grads, ids, ns, a, OPT_FTRL, stream, tokens);
}

}  // extern "C"

static bool claim_slot(kv_table* v, kv_table* sl, hipStream_t s);

// shared body of the optimizer ops.  `token` names the batch index a lookup left in the var's workspace
// (kv_gather_or_insert_tok): the same ids, so the index pass is skipped.  The caller holds the locks.
template <int OPT>
static int apply_common(kv_table* v, kv_table* s0, kv_table* s1, const float* grad, const void* ids, int64_t n,
                        const OptArgs& a, kv_batch_token_t token, hipStream_t s) {
  const long long nmax = fused_tab(v) ? FUSED_MAX_N : (1ll << 21);
  if (n < 0 || n > nmax)
    return fail(n < 0 ? KV_INVALID_ARGUMENT : KV_UNIMPLEMENTED,
                "indices: %lld ids in one optimizer call (limit %lld for this embedding dim; split the batch)", (long long)n, nmax);
  if (n > 0 && (!grad || !ids)) return fail(KV_INVALID_ARGUMENT, "grad / indices pointer is null");
  if (!dim_supported(v->dim))
    return fail(KV_UNIMPLEMENTED, "embedding dim %d not supported by the fused kernels", v->dim);
  int rc;
  if (tl_unique && fused_ok(v->dim) && !stream_is_capturing(s)) {
    // The caller promises unique ids (kv_apply_*_unique; kv_uapply.h): one launch, one lane group per id.  A pending
    // partition pass was settled by the caller's hand_over (no token is given).  Dims the kernel does not serve take the
    // batch pipeline below, which needs no promise.
    if ((rc = ensure_capacity(v, n, s)) || (rc = ensure_capacity(s0, n, s)) || (s1 && (rc = ensure_capacity(s1, n, s)))) return rc;
    if (v->uniq_serial >= 65535u) {   // the 16-bit stamp wraps: every row back to "none" (once per 65535 launches)
      k_clear_stamps<<<nblocks((long long)v->rows_ub, TB, 4096), TB, 0, s>>>(dev_view(v), (unsigned)v->rows_ub);
      v->uniq_serial = 0;
    }
    PartArgs pa{};
    pa.tv = dev_view(v); pa.ts0 = dev_view(s0); pa.ts1 = s1 ? dev_view(s1) : pa.ts0;
    pa.opt = a; pa.grad = grad; pa.day = today(v);
    pa.opt.fast = fast_math_on(v) ? 1 : 0;
    pa.n = n;
    pa.use_hints = claim_slot(v, s0, s) ? 1 : 0;
    if ((rc = mirror_decide<OPT>(v, s0, pa, true, s))) return rc;
    pa.uniq_serial = ++v->uniq_serial;
    ProfScope ps(v, KV_PROF_APPLY_UNIQUE, s);
    rc = (OPT == OPT_ADAM_V4 || OPT == OPT_ADAM_V3) ? kvp_launch_uapply_a(OPT, &pa, ids, v->key_dtype == KV_DT_INT32 ? 1 : 0, n, (void*)s)
                                                    : kvp_launch_uapply_b(OPT, &pa, ids, v->key_dtype == KV_DT_INT32 ? 1 : 0, n, (void*)s);
    if (rc) return fail(rc, "unique apply: no kernel for dim %d", v->dim);
    HIP_TRY(hipGetLastError());
    return KV_OK;
  }
  const bool reuse = token != 0 && token == v->batch_serial && n == v->batch_n;
  // The entry-list kernels serve this dim (pa_route): the tile sums, then k_papply — the partition pass and the update in
  // one launch — in the mode the batch's state asks for:
  //   PA_LOOKUP    the token names the lookup whose partition pass is still pending: k_papply completes its bookkeeping too
  //   PA_NONE      the token names a batch whose bookkeeping is done (a second optimizer on the token; a pass another op settled)
  //   PA_APPLYIDX  no (valid) token: the optimizer meets the ids first — the tile pass runs with the tile sums (k_ltsum)
  const bool pa_route = fused_tab(v);
  int pa_mode = -1;
  PartArgs pend{};
  const void* tile_ids = nullptr;   // != nullptr: the batch's tile pass runs in front of the apply (k_ltsum)
  if (v->part_pending) {
    if (reuse && pa_route && v->fused_index) {
      std::memcpy(&pend, v->pend_pa, sizeof pend);
      v->part_pending = false;
      pa_mode = PA_LOOKUP;
    } else if ((rc = flush_part(v, s))) {
      return rc;
    }
  }
  if (!reuse && (rc = ensure_capacity(v, n, s))) return rc;
  if ((rc = ensure_capacity(s0, n, s))) return rc;
  if (s1 && (rc = ensure_capacity(s1, n, s))) return rc;
  if ((rc = ensure_workspace(v, n, true, s))) return rc;
  WsDev wd = ws_view(v, n);
  PartArgs pa{};
  pa.tv = dev_view(v); pa.ts0 = dev_view(s0); pa.ts1 = s1 ? dev_view(s1) : pa.ts0;
  pa.opt = a; pa.grad = grad; pa.day = today(v);
  pa.opt.fast = fast_math_on(v) ? 1 : 0;
  pa.det = det_mode(v);
  pa.n = n;
  pa.use_hints = claim_slot(v, s0, s) ? 1 : 0;
  pa.day_lk = pa.day;
  if (pa_mode == PA_LOOKUP) { pa.day_lk = pend.day; pa.count_once = pend.count_once; }
  if (!reuse) {
    v->batch_serial = 0;
    if (pa_route) {   // tile pass + tile sums in one launch, then partition pass + update in one launch
      v->fused_index = true;
      choose_partitions(v, wd, n);
      tile_ids = ids;
      pa_mode = PA_APPLYIDX;
    } else {
      index_pass<MODE_APPLYIDX>(v, wd, pa, ids, nullptr, n, -1, nullptr, s);
    }
    v->batch_serial = ++g_serial;   // the index stays valid for this batch (e.g. a second optimizer on the same ids)
    v->batch_n = n;
  } else if (pa_mode < 0 && v->fused_index) {
    pa_mode = PA_NONE;   // the tiles' entries of a batch whose bookkeeping is done
  }
  if (v->fused_index && reuse && v->index_P) { wd.P = v->index_P; wd.pshift = 64 - ilog2(wd.P); }   // the lookup's partitioning
  if ((rc = mirror_decide<OPT>(v, s0, pa, v->fused_index, s))) return rc;
  if (v->fused_index) rc = fused_apply<OPT>(v, wd, pa, n, s, pa_mode, tile_ids);
  else rc = launch_apply<MODE_APPLY, OPT>(v, wd, pa, n, s);
  if (rc) return rc;
  HIP_TRY(hipGetLastError());
  return KV_OK;
}

extern "C" {

int kv_apply_group_adam_tok(kv_handle_t v, kv_handle_t mvl, const float* grad, const void* ids, int64_t n,
                            float lr, float b1p, float b2p, float b1, float b2, float eps, float l1,
                            float l2, float l21, int version, kv_batch_token_t token, kv_stream_t stream) {
  int rc;
  if ((rc = check_table(v)) || (rc = check_table(mvl))) return rc;
  if (version != 3 && version != 4) return fail(KV_INVALID_ARGUMENT, "GroupAdam version %d: 3 or 4", version);
  // order and wording of training_ops.cc:7001-7103
  if (!v->initialized || !mvl->initialized)
    return fail(KV_FAILED_PRECONDITION, "Failed to use uninitialized variables: %s", !v->initialized ? "var" : "m_v_linear");
  if (!(lr > 0.f)) return fail(KV_INVALID_ARGUMENT, "lr is not a positive scalar: %g", lr);
  if (!(l1 >= 0.f)) return fail(KV_INVALID_ARGUMENT, "l1 regularization strength is not a non-negative scalar: %g", l1);
  if (!(l2 >= 0.f)) return fail(KV_INVALID_ARGUMENT, "l2 regularization strength is not a non-negative scalar: %g", l2);
  if (!(l21 >= 0.f)) return fail(KV_INVALID_ARGUMENT, "l21 regularization strength is not a non-negative scalar: %g", l21);
  if (mvl->dim != 3 * v->dim)
    return fail(KV_INVALID_ARGUMENT, "kv_variable and linear do not have the same shape [%d] [%d] (m_v_linear must be 3x)", v->dim, mvl->dim);
  if (v->device != mvl->device) return fail(KV_INVALID_ARGUMENT, "var and slot live on different devices");
  if (v == mvl) return fail(KV_INVALID_ARGUMENT, "var and m_v_linear are the same table");
  if (n == 0) return KV_OK;
  DeviceGuard dg(v->device);
  MultiLock lk({v, mvl});
  MirrorKeep mk(v, mvl);   // (apply_common decides whether this apply works on the mirrors: mirror_decide)
  hipStream_t s = (hipStream_t)stream;
  if ((rc = lk.enter(s, token != 0 && token == v->batch_serial ? v : nullptr))) return rc;
  OptArgs a{};
  a.lr = lr; a.b1p = b1p; a.b2p = b2p; a.b1 = b1; a.b2 = b2; a.eps = eps;
  if (version == 4) {  // :7111-7120
    a.l1 = l1 * lr; a.l2 = l2 * lr; a.l21 = l21 * lr;
    a.alpha = lr * std::sqrt(1.f - b2p) / (1.f - b1p);
  } else {             // :5840-5849
    a.l1 = l1; a.l2 = l2; a.l21 = l21;
    a.alpha = std::sqrt(1.f - b2p) / (1.f - b1p);
  }
  a.l21_norm = a.l21 * std::sqrt((float)v->dim);
  return version == 4 ? apply_common<OPT_ADAM_V4>(v, mvl, nullptr, grad, ids, n, a, token, s)
                      : apply_common<OPT_ADAM_V3>(v, mvl, nullptr, grad, ids, n, a, token, s);
}
int kv_apply_group_adam(kv_handle_t v, kv_handle_t mvl, const float* grad, const void* ids, int64_t n,
                        float lr, float b1p, float b2p, float b1, float b2, float eps, float l1,
                        float l2, float l21, int version, kv_stream_t stream) {
  return kv_apply_group_adam_tok(v, mvl, grad, ids, n, lr, b1p, b2p, b1, b2, eps, l1, l2, l21, version, 0, stream);
}

int kv_apply_adagrad_tok(kv_handle_t v, kv_handle_t acc, float lr, const float* grad, const void* ids,
                         int64_t n, int update_slots, kv_batch_token_t token, kv_stream_t stream) {
  int rc;
  if ((rc = check_table(v)) || (rc = check_table(acc))) return rc;
  if (!v->initialized || !acc->initialized)
    return fail(KV_FAILED_PRECONDITION, "Attempting to use uninitialized variables: %s", !v->initialized ? "var" : "accum");
  if (acc->dim != v->dim) return fail(KV_INVALID_ARGUMENT, "var and accum do not have the same shape [%d] [%d]", v->dim, acc->dim);
  if (v->device != acc->device || v == acc) return fail(KV_INVALID_ARGUMENT, "var and accum must be distinct tables on one device");
  if (n == 0) return KV_OK;
  DeviceGuard dg(v->device);
  MultiLock lk({v, acc});
  MirrorKeep mk(v, acc);   // (mirror_decide in apply_common)
  hipStream_t s = (hipStream_t)stream;
  if ((rc = lk.enter(s, token != 0 && token == v->batch_serial ? v : nullptr))) return rc;
  OptArgs a{};
  a.lr = lr; a.update_slots = update_slots;
  return apply_common<OPT_ADAGRAD>(v, acc, nullptr, grad, ids, n, a, token, s);
}
int kv_apply_adagrad(kv_handle_t v, kv_handle_t acc, float lr, const float* grad, const void* ids,
                     int64_t n, int update_slots, kv_stream_t stream) {
  return kv_apply_adagrad_tok(v, acc, lr, grad, ids, n, update_slots, 0, stream);
}

int kv_apply_sparse_group_ftrl_tok(kv_handle_t v, kv_handle_t acc, kv_handle_t lin, const float* grad,
                                   const void* ids, int64_t n, float lr, float l1, float l2, float l21,
                                   float l2s, float lr_power, kv_batch_token_t token, kv_stream_t stream) {
  int rc;
  if ((rc = check_table(v)) || (rc = check_table(acc)) || (rc = check_table(lin))) return rc;
  if (!v->initialized || !acc->initialized || !lin->initialized)
    return fail(KV_FAILED_PRECONDITION, "Failed to use uninitialized variables");
  if (!(lr > 0.f)) return fail(KV_INVALID_ARGUMENT, "lr is not a positive scalar: %g", lr);
  if (!(l1 >= 0.f)) return fail(KV_INVALID_ARGUMENT, "l1 regularization strength is not a non-negative scalar: %g", l1);
  if (!(l2 >= 0.f)) return fail(KV_INVALID_ARGUMENT, "l2 regularization strength is not a non-negative scalar: %g", l2);
  if (!(l21 >= 0.f)) return fail(KV_INVALID_ARGUMENT, "l21 regularization strength is not a non-negative scalar: %g", l21);
  if (!(lr_power <= 0.f)) return fail(KV_INVALID_ARGUMENT, "lr_power is not a non-positive scalar: %g", lr_power);
  if (!(l2s >= 0.f)) return fail(KV_INVALID_ARGUMENT, "l2 shrinkage regularization strength is not a non-negative scalar: %g", l2s);
  if (acc->dim != v->dim) return fail(KV_INVALID_ARGUMENT, "kv_varaible and accum do not have the same shape [%d] [%d]", v->dim, acc->dim);
  if (lin->dim != v->dim) return fail(KV_INVALID_ARGUMENT, "kv_variable and linear do not have the same shape [%d] [%d]", v->dim, lin->dim);
  if (v->device != acc->device || v->device != lin->device || v == acc || v == lin || acc == lin)
    return fail(KV_INVALID_ARGUMENT, "var, accum and linear must be distinct tables on one device");
  if (n == 0) return KV_OK;
  DeviceGuard dg(v->device);
  MultiLock lk({v, acc, lin});
  hipStream_t s = (hipStream_t)stream;
  if ((rc = lk.enter(s, token != 0 && token == v->batch_serial ? v : nullptr))) return rc;
  OptArgs a{};
  a.lr = lr; a.l1 = l1; a.l2 = l2; a.l21 = l21; a.l2s = l2s; a.lr_power = lr_power;
  a.l21_norm = l21 * std::sqrt((float)v->dim);  // :728
  return apply_common<OPT_FTRL>(v, acc, lin, grad, ids, n, a, token, s);
}
int kv_apply_sparse_group_ftrl(kv_handle_t v, kv_handle_t acc, kv_handle_t lin, const float* grad,
                               const void* ids, int64_t n, float lr, float l1, float l2, float l21,
                               float l2s, float lr_power, kv_stream_t stream) {
  return kv_apply_sparse_group_ftrl_tok(v, acc, lin, grad, ids, n, lr, l1, l2, l21, l2s, lr_power, 0, stream);
}

// The same ops with the caller's promise that `ids` holds no id twice — what the reference's ops receive in an unchanged
// TF graph (TF-core de-duplicates the IndexedSlices in front of them, variable_scope.py:1096-1106): kv_uapply.h
struct UniqueScope { UniqueScope() { tl_unique = true; } ~UniqueScope() { tl_unique = false; } };
int kv_apply_group_adam_unique(kv_handle_t v, kv_handle_t mvl, const float* grad, const void* ids, int64_t n,
                               float lr, float b1p, float b2p, float b1, float b2, float eps, float l1,
                               float l2, float l21, int version, kv_stream_t stream) {
  UniqueScope u;
  return kv_apply_group_adam_tok(v, mvl, grad, ids, n, lr, b1p, b2p, b1, b2, eps, l1, l2, l21, version, 0, stream);
}
int kv_multi_apply_group_adam_unique(int num_tables, const kv_handle_t* vars, const kv_handle_t* slots,
                                     const float* const* grads, const void* const* ids, const int64_t* ns, float lr,
                                     float b1p, float b2p, float b1, float b2, float eps, float l1, float l2, float l21,
                                     int version, kv_stream_t stream) {
  UniqueScope u;
  return kv_multi_apply_group_adam_tok(num_tables, vars, slots, grads, ids, ns, lr, b1p, b2p, b1, b2, eps, l1, l2, l21, version,
                                       nullptr, stream);
}
int kv_multi_apply_adagrad_unique(int num_tables, const kv_handle_t* vars, const kv_handle_t* accums, float lr,
                                  const float* const* grads, const void* const* ids, const int64_t* ns, int update_slots,
                                  kv_stream_t stream) {
  UniqueScope u;
  return kv_multi_apply_adagrad_tok(num_tables, vars, accums, lr, grads, ids, ns, update_slots, nullptr, stream);
}
int kv_multi_apply_sparse_group_ftrl_unique(int num_tables, const kv_handle_t* vars, const kv_handle_t* accums,
                                            const kv_handle_t* linears, const float* const* grads, const void* const* ids,
                                            const int64_t* ns, float lr, float l1, float l2, float l21, float l2s,
                                            float lr_power, kv_stream_t stream) {
  UniqueScope u;
  return kv_multi_apply_sparse_group_ftrl_tok(num_tables, vars, accums, linears, grads, ids, ns, lr, l1, l2, l21, l2s, lr_power,
                                              nullptr, stream);
}
int kv_apply_adagrad_unique(kv_handle_t v, kv_handle_t acc, float lr, const float* grad, const void* ids,
                            int64_t n, int update_slots, kv_stream_t stream) {
  UniqueScope u;
  return kv_apply_adagrad_tok(v, acc, lr, grad, ids, n, update_slots, 0, stream);
}
int kv_apply_sparse_group_ftrl_unique(kv_handle_t v, kv_handle_t acc, kv_handle_t lin, const float* grad,
                                      const void* ids, int64_t n, float lr, float l1, float l2, float l21,
                                      float l2s, float lr_power, kv_stream_t stream) {
  UniqueScope u;
  return kv_apply_sparse_group_ftrl_tok(v, acc, lin, grad, ids, n, lr, l1, l2, l21, l2s, lr_power, 0, stream);
}

// The slot table whose rows the var's index entries remember (Entry::hint): the first slot-0 table an
// optimizer uses with the var, or the one kv_attach_slot names.  Hints of a table that was cleared since
// (import) mean nothing any more and are forgotten; another table simply goes without hints.
static bool claim_slot(kv_table* v, kv_table* sl, hipStream_t s) {
  if (v->slot_uid == sl->uid && v->slot_gen == sl->gen) return true;
  if (v->slot_uid != 0 && v->slot_uid != sl->uid) return false;
  if (v->slot_uid == sl->uid)   // same table, cleared since
    k_clear_hints<<<nblocks((long long)v->cap + 1, TB, 8192), TB, 0, s>>>(v->entries, v->cap + 1);
  v->slot_uid = sl->uid;
  v->slot_gen = sl->gen;
  return true;
}

int kv_attach_slot(kv_handle_t v, kv_handle_t sl, kv_stream_t stream) {
  int rc;
  if ((rc = check_table(v)) || (rc = check_table(sl))) return rc;
  if (v == sl || v->device != sl->device || v->key_dtype != sl->key_dtype)
    return fail(KV_INVALID_ARGUMENT, "kv_attach_slot: var and slot must be distinct tables on one device with one key dtype");
  DeviceGuard dg(v->device);
  MultiLock lk({v, sl});
  hipStream_t s = (hipStream_t)stream;
  if ((rc = lk.enter(s))) return rc;
  unsigned nrows = 1;
  if ((rc = stats(v, s, nullptr, &nrows))) return rc;
  if (v->slot_uid != 0 && (v->slot_uid != sl->uid || v->slot_gen != sl->gen))
    k_clear_hints<<<nblocks((long long)v->cap + 1, TB, 8192), TB, 0, s>>>(v->entries, v->cap + 1);
  v->slot_uid = sl->uid;
  v->slot_gen = sl->gen;
  v->batch_serial = 0;
  // (the entry above ended any running epoch of either table; a pair of single-chunk tables gets its mirrors filled here)
  const bool mir = v->chunks.size() == 1 && sl->chunks.size() == 1 && !v->track_delta && !sl->track_delta && mirror_pair(v, sl, s);
  if (nrows > 1)
    k_link_hints<<<nblocks(nrows, TB, 8192), TB, 0, s>>>(dev_view(v), dev_view(sl), nrows, v->mirror_epoch.load() & 0xFFFFu, mir ? 1 : 0);
  HIP_TRY(hipGetLastError());
  return KV_OK;
}

int kv_set_deterministic(kv_handle_t t, int on) {
  int rc;
  if ((rc = check_table(t))) return rc;
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  if ((rc = settle_pending(t))) return rc;
  if (on < 0 || on > 2) return fail(KV_INVALID_ARGUMENT, "kv_set_deterministic: on = %d (0, 1 or 2)", on);
  if (on == 2 && t->shard_refs.load() > 0)
    return fail(KV_UNIMPLEMENTED, "kv_set_deterministic(h, 2): the table serves a kv_shard (occurrence order is a single-table notion)");
  t->deterministic = on != 0;
  t->occurrence_order = on == 2;
  t->batch_serial = 0;      // the index a lookup left was built by the other pipeline's rules
  t->fused_index = false;
  return KV_OK;
}

int kv_set_fast_math(kv_handle_t t, int on) {
  int rc;
  if ((rc = check_table(t))) return rc;
  std::lock_guard<std::mutex> l(t->mu);
  t->fast_math = on != 0;
  return KV_OK;
}

int kv_get_stat(kv_handle_t t, int which, int64_t* value) {
  int rc;
  if ((rc = check_table(t))) return rc;
  if (!value) return fail(KV_INVALID_ARGUMENT, "kv_get_stat: null value");
  std::lock_guard<std::mutex> l(t->mu);
  if (which == KV_STAT_MIRROR_APPLIES) { *value = t->stat_mirror_applies; return KV_OK; }
  if (which == KV_STAT_MIRROR_EPOCHS) { *value = t->stat_mirror_epochs.load(); return KV_OK; }
  return fail(KV_INVALID_ARGUMENT, "kv_get_stat: unknown counter %d", which);
}

int kv_prepare_capture(kv_handle_t t, int64_t max_new_ids, kv_stream_t stream) {
  int rc;
  if ((rc = check_table(t))) return rc;
  if (max_new_ids < 0) return fail(KV_INVALID_ARGUMENT, "max_new_ids < 0");
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  hipStream_t s = (hipStream_t)stream;
  if ((rc = enter_op(t, s))) return rc;
  mirror_ban(t, s);   // (a captured apply replays without host code: no slot mirrors for this table from here on)
  unsigned c[3];
  HIP_TRY(hipMemcpyAsync(c, t->d_counters, sizeof c, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (c[1]) return flagged_error(t, c[1], s);
  const long long freed = std::max(0, (int)c[2]);
  t->rows_ub = c[0];
  t->free_known = freed;
  t->idx_ub = t->idx_base + (c[0] - t->bump_base) +
              (unsigned long long)std::max<long long>(0, (long long)t->pushes_since - (freed - t->free_base));
  // make room now (this may grow the table), then give the room back: the captured calls take it piece by piece
  const unsigned long long r0 = t->rows_ub, i0 = t->idx_ub;
  if ((rc = ensure_capacity(t, max_new_ids, s))) return rc;
  t->rows_ub = r0; t->idx_ub = i0;
  HIP_TRY(hipStreamSynchronize(s));
  return KV_OK;
}

// inverse[i] = the dense number of position i's id: position -> its entry in its tile -> the number k_papply PA_UNIQUE gave it
__global__ void __launch_bounds__(TB) k_inverse_e(const unsigned short* __restrict__ pos_ent, const unsigned* __restrict__ ent_b,
                                                  long long n, int* __restrict__ inverse) {
  for (long long i = (long long)blockIdx.x * TB + threadIdx.x; i < n; i += (long long)gridDim.x * TB)
    inverse[i] = (int)ent_b[(size_t)(i / TILE) * TILE + pos_ent[i]];
}

// tf.unique_with_counts on the entry-list kernels (any dim: no row is touched): a table-less tile pass (entries, every
// position's entry) and k_papply PA_UNIQUE with dense numbers — uniq / uniq_counts written, every entry learns its id's
// number, the count in wd.ctr[0].  The table's mutex is held by the caller.
// the table-less tile pass of the distinct-id ops: entries, every position's entry number (pos_ent)
static int unique_tile_pass(kv_table* t, WsDev& wd, PartArgs& pa, const void* ids, const int* counts, long long n, hipStream_t s) {
  Workspace& ws = t->ws;
  int rc;
  if (ws.pos_cap < n) {
    if ((rc = ws_sync(s))) return rc;   // (refused under a stream capture before anything is queued, like ensure_workspace)
    ws.pos_cap = 0;
    if ((rc = regrow(&ws.pos_ent, (size_t)std::max<long long>(n, ws.cap_n)))) return rc;
    ws.pos_cap = std::max<long long>(n, ws.cap_n);
  }
  t->fused_index = true;
  choose_partitions(t, wd, n);
  wd.pos_ent = ws.pos_ent;
  launch_ltile_notable(t, pa.tv, wd, ids, n, s, counts, t->key_dtype == KV_DT_INT32);
  return KV_OK;
}
static int fused_unique_pass(kv_table* t, WsDev& wd, PartArgs& pa, const void* ids, const int* counts, long long n, hipStream_t s) {
  int rc;
  if ((rc = unique_tile_pass(t, wd, pa, ids, counts, n, s))) return rc;
  // (k_papply_uniq: numbering only, nothing of the row geometry is touched — one kernel whatever the table's dim)
  if ((rc = kvp_launch_papply_ud(&wd, &pa, PA_UNIQUE, (void*)s))) return fail(rc, "unique: no kernel");
  return KV_OK;
}

// the count a synchronous op returns: device word -> pinned host word -> the caller (a copy into pageable memory goes through
// the runtime's staging path: measured against this in bench.py's unchanged_graph record)
static int read_count(kv_table* t, const unsigned* dev, hipStream_t s, unsigned* out) {
  if (!t->cnt_host) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&t->cnt_host), 64, hipHostMallocDefault));
  HIP_TRY(hipMemcpyAsync(t->cnt_host, dev, sizeof(unsigned), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  *out = *reinterpret_cast<volatile unsigned*>(t->cnt_host);
  return KV_OK;
}

// tf.unique + unsorted_segment_sum on the batch pipeline; the table's mutex is held by the caller.
// fold_op: how the rows of one id combine (KV_SCATTER_ADD = sum, MUL = product, MIN, MAX)
static int dedup_locked(kv_table* t, const void* ids, const float* grad, int64_t n, int64_t* uniq,
                        float* summed, int32_t* inverse, int64_t* num_unique, int fold_op, hipStream_t s) {
  int rc;
  if ((rc = ensure_workspace(t, n, true, s))) return rc;
  WsDev wd = ws_view(t, n);
  t->batch_serial = 0;
  PartArgs pa{};
  pa.tv = dev_view(t); pa.ts0 = pa.tv; pa.ts1 = pa.tv;
  pa.grad = grad;
  pa.out_keys = (long long*)uniq;
  pa.out_sum = summed;
  pa.fold_op = fold_op;
  pa.det = det_mode(t);
  pa.n = n;
  if (fold_op == KV_SCATTER_ADD && fused_tab(t)) {
    // the entry-list kernels: distinct ids numbered (fused_unique_pass), tile sums of the rows of ids repeated inside their
    // tile (k_tsum), the per-id sums over the tiles' entries straight to summed[number] (k_papply PA_DEDUP)
    // (round 6: ONE partition pass — k_papply PA_DEDUP numbers the ids it sums, dd_number — where PA_UNIQUE's numbering pass
    //  ran in front of it: the table-less tile pass, the tile sums, the pass)
    if ((rc = unique_tile_pass(t, wd, pa, ids, nullptr, n, s))) return rc;
    pa.out_map = nullptr;
    pa.dd_number = 1;
    pa.epart = wd.epart;
    pa.day_lk = pa.day;
    if ((rc = kvp_launch_tsum(&pa.tv, &wd, grad, (void*)s, nullptr, 0))) return fail(rc, "tile sums: no kernel for dim %d", t->dim);
    if ((rc = kvp_launch_papply_ud(&wd, &pa, PA_DEDUP, (void*)s))) return fail(rc, "per-id sums: no kernel for dim %d", t->dim);
    if (inverse) k_inverse_e<<<nblocks(n, TB, 2048), TB, 0, s>>>(t->ws.pos_ent, wd.ent_b, n, inverse);
  } else {
    index_pass<MODE_UNIQUE>(t, wd, pa, ids, nullptr, n, -1, nullptr, s);
    if ((rc = launch_apply<MODE_DEDUP, OPT_ADAGRAD>(t, wd, pa, n, s))) return rc;
    if (inverse) k_dedup_inverse<<<nblocks(n, TB, 2048), TB, 0, s>>>(wd, n, inverse);
  }
  unsigned U = 0;
  if ((rc = read_count(t, wd.ctr, s, &U))) return rc;
  *num_unique = U;
  return KV_OK;
}

int kv_dedup_segment_sum(kv_handle_t t, const void* ids, const float* grad, int64_t n, int64_t* uniq,
                         float* summed, int32_t* inverse, int64_t* num_unique, kv_stream_t stream) {
  int rc;
  if ((rc = check_table(t))) return rc;
  if (!num_unique) return fail(KV_INVALID_ARGUMENT, "num_unique is null");
  *num_unique = 0;
  if (n == 0) return KV_OK;
  if (n < 0 || !ids || !grad || !uniq || !summed) return fail(KV_INVALID_ARGUMENT, "bad arguments");
  if (n > (fused_tab(t) ? FUSED_MAX_N : (1ll << 21)))
    return fail(KV_UNIMPLEMENTED, "%lld ids in one call (limit 2^%d)", (long long)n, fused_tab(t) ? 23 : 21);
  if (!dim_supported(t->dim)) return fail(KV_UNIMPLEMENTED, "embedding dim %d not supported", t->dim);
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  MirrorKeep mk(t, t);   // `t` lends its workspace: neither its rows nor any record is touched
  if ((rc = enter_op(t, (hipStream_t)stream))) return rc;
  return dedup_locked(t, ids, grad, n, uniq, summed, inverse, num_unique, KV_SCATTER_ADD, (hipStream_t)stream);
}

int kv_unsorted_segment_sum(kv_handle_t t, const int32_t* segment_ids, const float* data, int64_t n,
                            int64_t num_segments, float* out, kv_stream_t stream) {
  int rc;
  if ((rc = check_table(t))) return rc;
  if (n < 0 || num_segments < 0 || num_segments > 0x7FFFFFFFll || (n > 0 && (!segment_ids || !data)) ||
      (num_segments > 0 && !out))
    return fail(KV_INVALID_ARGUMENT, "bad arguments");
  if (n > (fused_tab(t) ? FUSED_MAX_N : (1ll << 21)))
    return fail(KV_UNIMPLEMENTED, "%lld rows in one call (limit 2^%d)", (long long)n, fused_tab(t) ? 23 : 21);
  if (!dim_supported(t->dim)) return fail(KV_UNIMPLEMENTED, "embedding dim %d not supported", t->dim);
  if (num_segments == 0) return KV_OK;
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  MirrorKeep mk(t, t);   // `t` lends its workspace: neither its rows nor any record is touched
  hipStream_t s = (hipStream_t)stream;
  if ((rc = enter_op(t, s))) return rc;
  HIP_TRY(hipMemsetAsync(out, 0, (size_t)num_segments * t->dim * sizeof(float), s));  // segments nobody names
  if (n == 0) return KV_OK;
  if ((rc = ensure_workspace(t, n, true, s))) return rc;
  WsDev wd = ws_view(t, n);
  t->batch_serial = 0;
  PartArgs pa{};
  pa.tv = dev_view(t); pa.ts0 = pa.tv; pa.ts1 = pa.tv;
  pa.grad = data;
  pa.out_sum = out;
  pa.direct_rows = num_segments;
  pa.fold_op = KV_SCATTER_ADD;
  pa.det = det_mode(t);
  pa.n = n;
  if (fused_tab(t)) {
    // the entry-list kernels: the segment ids de-duplicated per tile (no numbering: an id IS its output row), the tile
    // sums, the per-id sums over the tiles' entries straight to out[id]
    t->fused_index = true;
    choose_partitions(t, wd, n);
    launch_ltile_notable(t, pa.tv, wd, segment_ids, n, s, nullptr, true);
    pa.epart = wd.epart;
    pa.day_lk = pa.day;
    if ((rc = kvp_launch_tsum(&pa.tv, &wd, data, (void*)s, nullptr, 0))) return fail(rc, "tile sums: no kernel for dim %d", t->dim);
    if ((rc = kvp_launch_papply_ud(&wd, &pa, PA_DEDUP, (void*)s))) return fail(rc, "segment sums: no kernel for dim %d", t->dim);
    HIP_TRY(hipGetLastError());
    return KV_OK;
  }
  index_pass<MODE_UNIQUE>(t, wd, pa, segment_ids, nullptr, n, 1, nullptr, s);
  if ((rc = launch_apply<MODE_DEDUP, OPT_ADAGRAD>(t, wd, pa, n, s))) return rc;
  HIP_TRY(hipGetLastError());
  return KV_OK;
}

int kv_unique(kv_handle_t t, const void* ids, const int32_t* counts, int64_t n, int64_t* uniq, int32_t* uniq_counts,
              int32_t* inverse, int64_t* num_unique, int64_t* num_unique_dev, kv_stream_t stream) {
  int rc;
  if ((rc = check_table(t))) return rc;
  if (!num_unique && !num_unique_dev) return fail(KV_INVALID_ARGUMENT, "num_unique and num_unique_dev are both null");
  if (num_unique) *num_unique = 0;
  if (n == 0) {
    if (num_unique_dev) HIP_TRY(hipMemsetAsync(num_unique_dev, 0, sizeof(int64_t), (hipStream_t)stream));
    return KV_OK;
  }
  if (n < 0 || !ids || !uniq) return fail(KV_INVALID_ARGUMENT, "bad arguments");
  if (n > (fused_off() ? (1ll << 21) : FUSED_MAX_N)) return fail(KV_UNIMPLEMENTED, "%lld ids in one call (limit 2^%d)", (long long)n, fused_off() ? 21 : 23);
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  MirrorKeep mk(t, t);   // `t` lends its workspace: neither its rows nor any record is touched
  hipStream_t s = (hipStream_t)stream;
  if ((rc = enter_op(t, s))) return rc;
  if ((rc = ensure_workspace(t, n, false, s))) return rc;
  WsDev wd = ws_view(t, n);
  t->batch_serial = 0;
  PartArgs pa{};
  pa.tv = dev_view(t); pa.ts0 = pa.tv; pa.ts1 = pa.tv;
  pa.out_keys = (long long*)uniq;
  pa.out_counts = uniq_counts;
  pa.det = det_mode(t);
  pa.n = n;
  if (!fused_off()) {   // the entry-list kernels, whatever the table's dim (no row is touched)
    if ((rc = fused_unique_pass(t, wd, pa, ids, counts, n, s))) return rc;
    if (inverse) k_inverse_e<<<nblocks(n, TB, 2048), TB, 0, s>>>(t->ws.pos_ent, wd.ent_b, n, inverse);
  } else {
    launch_tile<false>(t, wd, ids, counts, n, s);   // ent_a = occurrences | saturating count per tile (and zeroes wd.ctr)
    launch_part_keys<MODE_UNIQUE>(wd, pa, s);
    if (inverse) k_dedup_inverse<<<nblocks(n, TB, 2048), TB, 0, s>>>(wd, n, inverse);
  }
  if (num_unique_dev) k_store_count<<<1, 1, 0, s>>>(wd.ctr, (long long*)num_unique_dev);
  HIP_TRY(hipGetLastError());
  if (num_unique) {   // synchronous form
    unsigned U = 0;
    int rc2;
    if ((rc2 = read_count(t, wd.ctr, s, &U))) return rc2;
    *num_unique = U;
  }
  return KV_OK;
}

int kv_bucket_by_owner(kv_handle_t t, const void* ids, int64_t n, const int64_t* n_dev, int world, int owner_rule, int64_t* out_ids,
                       int32_t* perm, int64_t* counts_dev, const int32_t* id_counts, int64_t* pairs_out,
                       int32_t* pos_out, kv_stream_t stream) {
  int rc;
  if ((rc = check_table(t))) return rc;
  if (world < 1 || world > MAXW) return fail(KV_INVALID_ARGUMENT, "world %d: 1..%d ranks", world, MAXW);
  if (owner_rule != KV_OWNER_HASH && owner_rule != KV_OWNER_MOD) return fail(KV_INVALID_ARGUMENT, "owner_rule %d", owner_rule);
  if (n < 0 || n > (1ll << 31) - 1 || (n > 0 && (!ids || !out_ids || !perm)) || !counts_dev)
    return fail(KV_INVALID_ARGUMENT, "bad arguments");
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  MirrorKeep mk(t, t);   // `t` lends its workspace: neither its rows nor any record is touched
  hipStream_t s = (hipStream_t)stream;
  const unsigned ntiles = (unsigned)((n + RT - 1) / RT);
  const size_t need = (size_t)std::max(1u, ntiles) * world;
  if (t->route_hist_cap < need) {
    HIP_TRY(hipStreamSynchronize(s));
    hipFree(t->route_hist);
    HIP_TRY(hipMalloc(&t->route_hist, need * sizeof(unsigned)));
    t->route_hist_cap = need;
  }
  if (n == 0) {
    HIP_TRY(hipMemsetAsync(counts_dev, 0, (size_t)world * sizeof(long long), s));
    return KV_OK;
  }
  if (t->key_dtype == KV_DT_INT32) {
    k_owner_hist<int><<<ntiles, TB, 0, s>>>((const int*)ids, n, world, owner_rule, ntiles, t->route_hist, (const long long*)n_dev);
    k_owner_scan<<<1, 1024, 0, s>>>(t->route_hist, ntiles * world, ntiles, world, (long long*)counts_dev);
    k_owner_scatter<int><<<ntiles, TB, 0, s>>>((const int*)ids, n, world, owner_rule, ntiles, t->route_hist, (long long*)out_ids, perm,
                                               (const long long*)n_dev, id_counts, (long long*)pairs_out, pos_out);
  } else {
    k_owner_hist<long long><<<ntiles, TB, 0, s>>>((const long long*)ids, n, world, owner_rule, ntiles, t->route_hist,
                                                  (const long long*)n_dev);
    k_owner_scan<<<1, 1024, 0, s>>>(t->route_hist, ntiles * world, ntiles, world, (long long*)counts_dev);
    k_owner_scatter<long long><<<ntiles, TB, 0, s>>>((const long long*)ids, n, world, owner_rule, ntiles, t->route_hist,
                                                     (long long*)out_ids, perm, (const long long*)n_dev, id_counts,
                                                     (long long*)pairs_out, pos_out);
  }
  HIP_TRY(hipGetLastError());
  return KV_OK;
}

// the free-list stack must hold every row of the slab
static int ensure_free_list(kv_table* t, hipStream_t s) {
  if (t->free_cap >= t->rows_cap) return KV_OK;
  unsigned* nf = nullptr;
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipMalloc(&nf, (size_t)t->rows_cap * sizeof(unsigned)));
  if (t->free_rows) {
    HIP_TRY(hipMemcpy(nf, t->free_rows, (size_t)t->free_cap * sizeof(unsigned), hipMemcpyDeviceToDevice));
    hipFree(t->free_rows);
  }
  t->free_rows = nf;
  t->free_cap = t->rows_cap;
  return KV_OK;
}

// after a kernel that pushed rows: read the counters, account the pushes (synchronous)
static int after_release(kv_table* t, hipStream_t s, unsigned long long* released) {
  unsigned c[3];
  unsigned long long n = 0;
  HIP_TRY(hipMemcpyAsync(&n, t->d_stat, sizeof n, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(c, t->d_counters, sizeof c, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  t->pushes_since += n;
  t->free_known = std::max(0, (int)c[2]);
  t->rows_ub = c[0];
  *released = n;
  return KV_OK;
}

static int delete_locked(kv_table* t, const void* ids, int64_t n, int64_t* num_deleted, hipStream_t s);

int kv_delete(kv_handle_t t, const void* ids, int64_t n, int64_t* num_deleted, kv_stream_t stream) {
  int rc;
  if ((rc = check_table(t))) return rc;
  if (num_deleted) *num_deleted = 0;
  if (n < 0 || (n > 0 && !ids)) return fail(KV_INVALID_ARGUMENT, "indices pointer is null");
  if (!t->initialized)
    return fail(KV_FAILED_PRECONDITION, "Failed to use uninitialized variables: KvVariable init table not set");
  if (n == 0) return KV_OK;
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  if ((rc = record_deleted(t, ids, n, t->key_dtype == KV_DT_INT32, (hipStream_t)stream))) return rc;
  return delete_locked(t, ids, n, num_deleted, (hipStream_t)stream);
}

static int delete_locked(kv_table* t, const void* ids, int64_t n, int64_t* num_deleted, hipStream_t s) {
  int rc;
  if ((rc = enter_op(t, s))) return rc;
  t->batch_serial = 0;   // rows are released: an index of a batch that held them is void (its token goes stale)
  if ((rc = ensure_free_list(t, s))) return rc;
  HIP_TRY(hipMemsetAsync(t->d_stat, 0, 4 * sizeof(unsigned long long), s));
  const TableDev td = dev_view(t);
  if (t->key_dtype == KV_DT_INT32)
    k_delete<int><<<nblocks(n, TB, 4096), TB, 0, s>>>(td, (const int*)ids, n, t->free_rows, t->d_stat);
  else
    k_delete<long long><<<nblocks(n, TB, 4096), TB, 0, s>>>(td, (const long long*)ids, n, t->free_rows, t->d_stat);
  HIP_TRY(hipGetLastError());
  unsigned long long rel = 0;
  if ((rc = after_release(t, s, &rel))) return rc;
  if (num_deleted) *num_deleted = (int64_t)rel;
  return KV_OK;
}

int kv_delete_with_timestamp(kv_handle_t t, int threshold, int dry_run, int64_t* out_keys, int64_t* count,
                             kv_stream_t stream) {
  int rc;
  if ((rc = check_table(t))) return rc;
  if (!count || (!dry_run && !out_keys)) return fail(KV_INVALID_ARGUMENT, "count / delete_keys pointer is null");
  if (!t->initialized)
    return fail(KV_FAILED_PRECONDITION, "Failed to use uninitialized variables: KvVariable init table not set");
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  hipStream_t s = (hipStream_t)stream;
  if (!dry_run) {   // out_keys was sized by a dry run: nothing may have touched the table (or its clock) since
    if (t->expire_serial != t->op_serial)
      return fail(KV_FAILED_PRECONDITION, "the table was used between the dry run and kv_delete_with_timestamp: the key "
                                          "buffer sized from the count may be too small; count again");
    ++t->op_serial;
    t->batch_serial = 0;   // rows are released: an index of a batch that held them is void
  }
  unsigned nrows = 1;
  if ((rc = stats(t, s, nullptr, &nrows))) return rc;
  if (!dry_run && (rc = ensure_free_list(t, s))) return rc;
  HIP_TRY(hipMemsetAsync(t->d_stat, 0, 4 * sizeof(unsigned long long), s));
  const unsigned thr = (unsigned)(threshold & 0xFFFF);  // static_cast<uint16_t>(threshold), kv_variable.h:771
  k_delete_by_time<<<nblocks(nrows, TB, 4096), TB, 0, s>>>(dev_view(t), nrows, today(t), thr, dry_run ? 0 : 1,
                                                        t->free_rows, t->d_stat, (long long*)out_keys);
  HIP_TRY(hipGetLastError());
  unsigned long long rel = 0;
  if (dry_run) {
    HIP_TRY(hipMemcpyAsync(&rel, t->d_stat, sizeof rel, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
  } else if ((rc = after_release(t, s, &rel))) {
    return rc;
  }
  *count = (int64_t)rel;
  if (dry_run) t->expire_serial = t->op_serial;
  if (!dry_run && (rc = record_deleted(t, out_keys, (int64_t)rel, false, s))) return rc;
  return KV_OK;
}

static int count_or_ts(kv_handle_t t, const void* ids, int64_t n, int what, uint32_t* out, kv_stream_t stream) {
  int rc;
  if ((rc = check_table(t))) return rc;
  if (n < 0 || (n > 0 && (!ids || !out))) return fail(KV_INVALID_ARGUMENT, "indices / output pointer is null");
  if (!t->initialized)
    return fail(KV_FAILED_PRECONDITION, "Failed to use uninitialized variables: KvVariable init table not set");
  if (n == 0) return KV_OK;
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  hipStream_t s = (hipStream_t)stream;
  if ((rc = join_side(t, s))) return rc;
  if (t->key_dtype == KV_DT_INT32)
    k_get_count_ts<int><<<nblocks(n, TB, 4096), TB, 0, s>>>(dev_view(t), (const int*)ids, n, what, today(t), out);
  else
    k_get_count_ts<long long><<<nblocks(n, TB, 4096), TB, 0, s>>>(dev_view(t), (const long long*)ids, n, what, today(t), out);
  HIP_TRY(hipGetLastError());
  return KV_OK;
}
int kv_get_count(kv_handle_t t, const void* ids, int64_t n, int32_t* counts, kv_stream_t stream) {
  return count_or_ts(t, ids, n, 0, (uint32_t*)counts, stream);
}
int kv_get_timestamp(kv_handle_t t, const void* ids, int64_t n, uint32_t* days, kv_stream_t stream) {
  return count_or_ts(t, ids, n, 1, days, stream);
}

int kv_take_rows(int device, const void* src, const int32_t* index, const int32_t* index_outer, int64_t n,
                 int64_t row_bytes, int scatter, void* out, kv_stream_t stream) {
  if (index_outer && scatter) return fail(KV_INVALID_ARGUMENT, "kv_take_rows: the two-level index is gather only");
  if (n < 0 || row_bytes <= 0 || row_bytes % 4 || (n > 0 && (!src || !index || !out)))
    return fail(KV_INVALID_ARGUMENT, "kv_take_rows: n %lld, row_bytes %lld (a positive multiple of 4)",
                (long long)n, (long long)row_bytes);
  if (n == 0) return KV_OK;
  DeviceGuard dg(device);
  hipStream_t s = (hipStream_t)stream;
  const bool wide = row_bytes % 16 == 0 && ((uintptr_t)src % 16 == 0) && ((uintptr_t)out % 16 == 0);
  const unsigned nu = (unsigned)(row_bytes / (wide ? 16 : 4));
  const int sh = (nu & (nu - 1)) == 0 ? ilog2(nu) : -1;
  const int grid = nblocks(n * nu, TB * 4, 8192);
  if (wide) {
    if (scatter) k_take_rows<float4, 1><<<grid, TB, 0, s>>>((const float4*)src, index, n, nu, sh, (float4*)out);
    else k_take_rows<float4, 0><<<grid, TB, 0, s>>>((const float4*)src, index, n, nu, sh, (float4*)out, index_outer);
  } else {
    if (scatter) k_take_rows<float, 1><<<grid, TB, 0, s>>>((const float*)src, index, n, nu, sh, (float*)out);
    else k_take_rows<float, 0><<<grid, TB, 0, s>>>((const float*)src, index, n, nu, sh, (float*)out, index_outer);
  }
  HIP_TRY(hipGetLastError());
  return KV_OK;
}

int kv_profile_enable(kv_handle_t t, int max_launches) {
  int rc;
  if ((rc = check_table(t))) return rc;
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  for (auto e : t->ev) hipEventDestroy(e);
  t->ev.clear();
  t->ev_kind.clear();
  t->ev_used = 0;
  t->prof = max_launches > 0;
  for (int i = 0; i < 2 * max_launches; ++i) {
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    t->ev.push_back(e);
  }
  t->ev_kind.assign((size_t)std::max(max_launches, 0), 0);
  return KV_OK;
}

int kv_profile_select(kv_handle_t t, unsigned kind_mask) {
  int rc;
  if ((rc = check_table(t))) return rc;
  std::lock_guard<std::mutex> l(t->mu);
  t->prof_mask = kind_mask;
  return KV_OK;
}

int kv_profile_sample(kv_handle_t t, int every) {
  int rc;
  if ((rc = check_table(t))) return rc;
  if (every < 1) return fail(KV_INVALID_ARGUMENT, "kv_profile_sample: every %d", every);
  std::lock_guard<std::mutex> l(t->mu);
  t->prof_every = every;
  for (unsigned& q : t->prof_seq) q = 0;
  return KV_OK;
}

int kv_profile_read(kv_handle_t t, double* ms_sum, int64_t* launches, int n_kinds) {
  int rc;
  if ((rc = check_table(t))) return rc;
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  for (int k = 0; k < n_kinds; ++k) { ms_sum[k] = 0; launches[k] = 0; }
  for (size_t i = 0; i + 1 < t->ev_used; i += 2) {
    HIP_TRY(hipEventSynchronize(t->ev[i + 1]));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, t->ev[i], t->ev[i + 1]));
    const int k = t->ev_kind[i / 2];
    if (k < n_kinds) { ms_sum[k] += ms; launches[k] += 1; }
  }
  t->ev_used = 0;
  return KV_OK;
}

#ifdef KV_STAMPS
int kv_debug_read_stamps(kv_handle_t t, unsigned long long* out, int64_t nblocks_) {
  DeviceGuard dg(t->device);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, t->ws.dbg, (size_t)std::min<int64_t>(nblocks_, 16384) * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return KV_OK;
}
#endif

int kv_export_count(kv_handle_t t, int first_n, int64_t* counts, kv_stream_t stream) {
  int rc;
  if ((rc = check_table(t))) return rc;
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  hipStream_t s = (hipStream_t)stream;
  unsigned nrows = 1;
  if ((rc = stats(t, s, nullptr, &nrows))) return rc;
  HIP_TRY(hipMemsetAsync(t->d_stat, 0, 4 * sizeof(unsigned long long), s));
  k_export<<<nblocks(nrows, TB, 2048), TB, 0, s>>>(dev_view(t), nrows, first_n, 0, t->d_stat, nullptr,
                                                   nullptr, nullptr, nullptr, nullptr);
  unsigned long long c[3];
  HIP_TRY(hipMemcpyAsync(c, t->d_stat, sizeof c, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  t->export_serial = t->op_serial;
  counts[0] = (int64_t)c[0]; counts[1] = (int64_t)c[1]; counts[2] = (int64_t)c[2];
  return KV_OK;
}

int kv_export_fill(kv_handle_t t, int first_n, int64_t* keys, float* values, int64_t* blacklist,
                   int64_t* fkeys, uint32_t* fvals, kv_stream_t stream) {
  int rc;
  if ((rc = check_table(t))) return rc;
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  hipStream_t s = (hipStream_t)stream;
  if (t->export_serial != t->op_serial)
    return fail(KV_FAILED_PRECONDITION, "the table was used between kv_export_count and kv_export_fill: the buffers sized "
                                        "from the counts may be too small; count again");
  ++t->op_serial;   // a fill ends the export (delta lists handed on): the next fill needs a new count
  unsigned nrows = 1;
  if ((rc = stats(t, s, nullptr, &nrows))) return rc;
  HIP_TRY(hipMemsetAsync(t->d_stat, 0, 4 * sizeof(unsigned long long), s));
  k_export<<<nblocks(nrows, TB, 2048), TB, 0, s>>>(dev_view(t), nrows, first_n, 1, t->d_stat,
                                                   (long long*)keys, values, (long long*)blacklist,
                                                   (long long*)fkeys, fvals);
  HIP_TRY(hipGetLastError());
  if (first_n > 2 && (rc = delta_after_export(t, first_n, nrows, s))) return rc;  // dynamic_save.hpp:179-192
  return KV_OK;
}

int kv_set_delta_tracking(kv_handle_t t, int support_delta_export, int support_prediction_delta_export) {
  int rc;
  if ((rc = check_table(t))) return rc;
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  if ((rc = settle_pending(t))) return rc;
  t->track_delta = support_delta_export != 0;
  t->track_pred = support_prediction_delta_export != 0;
  return KV_OK;
}

int kv_export_delta_count(kv_handle_t t, int first_n, int64_t* counts, kv_stream_t stream) {
  int rc;
  if ((rc = check_table(t))) return rc;
  if (!counts) return fail(KV_INVALID_ARGUMENT, "counts pointer is null");
  if (!t->initialized)
    return fail(KV_FAILED_PRECONDITION, "Failed to use uninitialized variables: KvVariable init table not set");
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  hipStream_t s = (hipStream_t)stream;
  unsigned nrows = 1;
  if ((rc = stats(t, s, nullptr, &nrows))) return rc;
  std::vector<long long> absent;
  if ((rc = delta_prepare(t, first_n, s, &absent))) return rc;
  HIP_TRY(hipMemsetAsync(t->d_stat, 0, 4 * sizeof(unsigned long long), s));
  k_export_delta<<<nblocks(nrows, TB, 2048), TB, 0, s>>>(dev_view(t), nrows, first_n, 0, t->d_stat, nullptr,
                                                         nullptr, nullptr, nullptr, nullptr);
  unsigned long long c[3];
  HIP_TRY(hipMemcpyAsync(c, t->d_stat, sizeof c, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  counts[0] = (int64_t)c[0];
  counts[1] = first_n > 3 ? (int64_t)c[1] : 0;
  counts[2] = first_n > 4 ? (int64_t)(c[2] + absent.size()) : 0;
  counts[3] = (int64_t)absent.size() + (first_n > 3 ? 0 : (int64_t)c[1]);
  t->delta_serial = t->op_serial;
  return KV_OK;
}

int kv_export_delta_fill(kv_handle_t t, int first_n, int64_t* keys, float* values, int64_t* blacklist,
                         int64_t* fkeys, uint32_t* fvals, int64_t* delete_keys, kv_stream_t stream) {
  int rc;
  if ((rc = check_table(t))) return rc;
  if (!t->initialized)
    return fail(KV_FAILED_PRECONDITION, "Failed to use uninitialized variables: KvVariable init table not set");
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  hipStream_t s = (hipStream_t)stream;
  if (t->delta_serial != t->op_serial)
    return fail(KV_FAILED_PRECONDITION, "the table was used between kv_export_delta_count and kv_export_delta_fill: the "
                                        "buffers sized from the counts may be too small; count again");
  ++t->op_serial;
  unsigned nrows = 1;
  if ((rc = stats(t, s, nullptr, &nrows))) return rc;
  std::vector<long long> absent;
  if ((rc = delta_prepare(t, first_n, s, &absent))) return rc;
  HIP_TRY(hipMemsetAsync(t->d_stat, 0, 4 * sizeof(unsigned long long), s));
  // prediction exports move the blacklisted keys to the delete list (dynamic_save.hpp:345-351)
  k_export_delta<<<nblocks(nrows, TB, 2048), TB, 0, s>>>(dev_view(t), nrows, first_n, 1, t->d_stat, (long long*)keys,
                                                         values, (long long*)(first_n > 3 ? blacklist : delete_keys),
                                                         (long long*)fkeys, fvals);
  HIP_TRY(hipGetLastError());
  unsigned long long c[3];
  HIP_TRY(hipMemcpyAsync(c, t->d_stat, sizeof c, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (!absent.empty()) {  // keys without a row: deleted (:233-236), frequency 0 (kv_variable.h:950)
    if (!delete_keys) return fail(KV_INVALID_ARGUMENT, "delete_keys pointer is null");
    const size_t off = first_n > 3 ? 0 : (size_t)c[1];
    HIP_TRY(hipMemcpyAsync(delete_keys + off, absent.data(), absent.size() * sizeof(long long), hipMemcpyHostToDevice, s));
    if (first_n > 4 && fkeys && fvals) {
      HIP_TRY(hipMemcpyAsync(fkeys + c[2], absent.data(), absent.size() * sizeof(long long), hipMemcpyHostToDevice, s));
      HIP_TRY(hipMemsetAsync(fvals + c[2], 0, absent.size() * sizeof(uint32_t), s));
    }
    HIP_TRY(hipStreamSynchronize(s));  // `absent` is the copy source
  }
  return delta_after_export(t, first_n, nrows, s);
}

// insert / scatter / import marks: tile pass (dedup) -> partition pass on the unique keys
static int scatter_like(kv_handle_t t, const void* ids, const float* vals, int64_t n, int op,
                        int is_insert, int mark, const unsigned* fvals, hipStream_t s) {
  int rc;
  if (n == 0) return KV_OK;
  if (n < 0 || !ids || (!vals && mark < 0)) return fail(KV_INVALID_ARGUMENT, "bad arguments");
  if (!t->initialized && !is_insert && mark < 0)
    return fail(KV_FAILED_PRECONDITION, "Failed to use uninitialized variables: KvVariable init table not set");
  const long long CH = 1ll << 21;
  const size_t idsz = t->key_dtype == KV_DT_INT32 ? 4 : 8;
  for (long long off = 0; off < n; off += CH) {
    const long long m = std::min(CH, (long long)n - off);
    if ((rc = ensure_capacity(t, m, s))) return rc;
    if ((rc = ensure_workspace(t, m, false, s))) return rc;
    const WsDev wd = ws_view(t, m);
    PartArgs pa{};
    pa.tv = dev_view(t);
    if (!t->initialized) {
      // InsertOrUpdate / import never consult the init table; new rows start from the zero row
      pa.tv.init_table = t->chunks[0].rows;
      pa.tv.init_rows = 1;
    }
    pa.ts0 = pa.tv; pa.ts1 = pa.tv;
    pa.grad = vals ? vals + (size_t)off * t->dim : nullptr;
    pa.scatter_op = op; pa.is_insert = is_insert;
    pa.mark_what = mark; pa.fvals = fvals ? fvals + off : nullptr;
    t->batch_serial = 0;
    launch_tile<true>(t, wd, (const char*)ids + (size_t)off * idsz, nullptr, m, s);
    if (mark >= 0) launch_part_keys<MODE_MARK>(wd, pa, s);
    else launch_part_keys<MODE_SCATTER>(wd, pa, s);
  }
  HIP_TRY(hipGetLastError());
  return KV_OK;
}

int kv_insert(kv_handle_t t, const void* ids, const float* values, int64_t n, kv_stream_t stream) {
  int rc;
  if ((rc = check_table(t))) return rc;
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  if ((rc = enter_op(t, (hipStream_t)stream))) return rc;
  return scatter_like(t, ids, values, n, KV_SCATTER_ASSIGN, 1, -1, nullptr, (hipStream_t)stream);
}

int kv_scatter_update(kv_handle_t t, const void* ids, const float* updates, int64_t n, int op,
                      kv_stream_t stream) {
  int rc;
  if ((rc = check_table(t))) return rc;
  if (op < KV_SCATTER_ASSIGN || op > KV_SCATTER_MAX) return fail(KV_INVALID_ARGUMENT, "unsupported update operation %d", op);
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  hipStream_t s = (hipStream_t)stream;
  if ((rc = enter_op(t, s))) return rc;
  if (op != KV_SCATTER_ASSIGN && n > 1 && ids && updates && dim_supported(t->dim)) {
    // ScatterUpdate applies every occurrence of an id in turn (kv_variable.h:616-734): the update rows of a
    // repeated id are combined first (sum for add / sub, product for mul / div, min, max — one row per
    // distinct id), then applied once.  Chunks of 2^21 ids one after another: occurrences in a later
    // chunk meet the row the earlier chunk left, as in the reference's sequential order.
    const int fold = (op == KV_SCATTER_ADD || op == KV_SCATTER_SUB) ? KV_SCATTER_ADD
                   : (op == KV_SCATTER_MUL || op == KV_SCATTER_DIV) ? KV_SCATTER_MUL : op;
    const long long CHK = 1ll << 21;
    const size_t idsz = t->key_dtype == KV_DT_INT32 ? 4 : 8;
    Workspace& w = t->ws;
    const long long want = std::min<long long>(n, CHK);
    if (w.scat_cap < want) {
      HIP_TRY(hipStreamSynchronize(s));
      const long long cap = std::max<long long>(want, std::min<long long>(w.scat_cap * 2, CHK));
      w.scat_cap = 0;
      if ((rc = regrow(&w.scat_keys, (size_t)cap)) || (rc = regrow(&w.scat_sum, (size_t)cap * t->dim))) return rc;
      w.scat_cap = cap;
    }
    for (long long off = 0; off < n; off += CHK) {
      const long long m = std::min(CHK, (long long)n - off);
      int64_t U = 0;
      if ((rc = dedup_locked(t, (const char*)ids + (size_t)off * idsz, updates + (size_t)off * t->dim, m,
                             (int64_t*)w.scat_keys, w.scat_sum, nullptr, &U, fold, s)))
        return rc;
      if (t->key_dtype == KV_DT_INT32 && U > 0)   // the unique list is int64; the table's ops take its own key type
        k_narrow_keys<<<1, 1024, 0, s>>>(w.scat_keys, U);
      if ((rc = scatter_like(t, w.scat_keys, w.scat_sum, U, op, 0, -1, nullptr, s))) return rc;
    }
    return KV_OK;
  }
  // assign: one of the occurrences of a repeated id stays (the reference's result depends on its thread
  // interleaving there); dims outside the fused kernels' range take this path for every operation
  return scatter_like(t, ids, updates, n, op, 0, -1, nullptr, s);
}

// placeholder init table for tables that are marked initialised by an import
static int ensure_init_placeholder(kv_table* t, hipStream_t s) {
  if (!t->init_table) {
    // the import marks the variable initialised (dynamic_restore.hpp:249-255).  The checkpoint's
    // init table is the caller's to pass through kv_init_table; without one, keys inserted later
    // start from a one-row zero table instead of dereferencing nothing.
    HIP_TRY(hipMalloc(&t->init_table, (size_t)t->dim * sizeof(float)));
    HIP_TRY(hipMemsetAsync(t->init_table, 0, (size_t)t->dim * sizeof(float), s));
    t->init_rows = 1;
    t->init_placeholder = true;
  }
  return KV_OK;
}

int kv_import_delta(kv_handle_t t, const int64_t* keys, const float* values, int64_t n, const int64_t* blacklist,
                    int64_t n_black, const int64_t* fkeys, const uint32_t* fvals, int64_t n_freq,
                    const int64_t* delete_keys, int64_t n_delete, int first_n, kv_stream_t stream) {
  int rc;
  if ((rc = check_table(t))) return rc;
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  hipStream_t s = (hipStream_t)stream;
  if (t->key_dtype == KV_DT_INT32) return fail(KV_UNIMPLEMENTED, "import with int32 keys");
  if ((rc = enter_op(t, s))) return rc;
  // Stage 1 (dynamic_restore.hpp:58-77): insert or overwrite, lift the blacklist, re-evaluate under_threshold
  if ((rc = scatter_like(t, keys, values, n, KV_SCATTER_ASSIGN, 3, -1, nullptr, s))) return rc;
  // Stage 2 (:92-112): first_n > 3 marks the blacklist, otherwise (inference load) those keys are removed
  if (n_black > 0) {
    if (first_n > 3) {
      if ((rc = scatter_like(t, blacklist, nullptr, n_black, 0, 1, 0, nullptr, s))) return rc;
    } else if ((rc = delete_locked(t, blacklist, n_black, nullptr, s))) {
      return rc;
    }
  }
  // Stage 3/4 (:114-135): frequency words of keys that exist
  if (n_freq > 0 && (rc = scatter_like(t, fkeys, nullptr, n_freq, 0, 1, 1, fvals, s))) return rc;
  // Stage 5 (:137-145): keys deleted since the checkpoint this delta follows
  if (n_delete > 0 && (rc = delete_locked(t, delete_keys, n_delete, nullptr, s))) return rc;
  if ((rc = ensure_init_placeholder(t, s))) return rc;
  t->initialized = true;
  HIP_TRY(hipGetLastError());
  return KV_OK;
}

int kv_import(kv_handle_t t, const int64_t* keys, const float* values, int64_t n, const int64_t* blacklist,
              int64_t n_black, const int64_t* fkeys, const uint32_t* fvals, int64_t n_freq,
              kv_stream_t stream) {
  int rc;
  if ((rc = check_table(t))) return rc;
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  hipStream_t s = (hipStream_t)stream;
  if (t->key_dtype == KV_DT_INT32) return fail(KV_UNIMPLEMENTED, "import with int32 keys");
  if ((rc = enter_op(t, s))) return rc;
  // clear(): dynamic_restore.hpp:60-62
  HIP_TRY(hipStreamSynchronize(s));
  t->gen += 1;            // row ids start over: slot-row hints into this table are void
  t->slot_uid = 0;        // and the fresh index carries none of its own
  t->batch_serial = 0;
  unsigned init[3] = {1, 0, 0};
  HIP_TRY(hipMemcpy(t->d_counters, init, sizeof init, hipMemcpyHostToDevice));  // stack source: synchronous
  k_fill_entries<<<nblocks((long long)t->cap + 1, TB, 8192), TB, 0, s>>>(t->entries, t->cap + 1);
  t->rows_ub = 1;
  t->idx_ub = t->idx_base = 0; t->bump_base = 1; t->pushes_since = 0; t->free_base = 0; t->free_known = 0;
  t->del_train.clear(); t->del_pred.clear();  // dynamic_restore.hpp:258-259 (the rows start over, and so do their bytes)
  if ((rc = scatter_like(t, keys, values, n, KV_SCATTER_ASSIGN, 2, -1, nullptr, s))) return rc;
  if (n_black > 0 && (rc = scatter_like(t, blacklist, nullptr, n_black, 0, 1, 0, nullptr, s))) return rc;
  if (n_freq > 0 && (rc = scatter_like(t, fkeys, nullptr, n_freq, 0, 1, 1, fvals, s))) return rc;
  if ((rc = ensure_init_placeholder(t, s))) return rc;
  t->initialized = true;
  HIP_TRY(hipGetLastError());
  return KV_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// sharded tables: hashed ownership, fixed-capacity exchange, RCCL over xGMI
// ------------------------------------------------------------------------------------------
namespace {
// RCCL is reached through dlopen: the copy the process already holds (PyTorch bundles one under the same soname)
// is reused, so there is one RCCL per process; nothing links against it when the table is not sharded
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;   // lossless mode only
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};
RcclApi* rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    // PyTorch's wheel carries its RCCL under the soname librccl.so, ROCm's own is librccl.so.1: whichever the process
    // already holds wins, so that there is one RCCL (one set of xGMI channels) per process
    for (const char* name : {"librccl.so", "librccl.so.1"})
      if ((api.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!api.lib)
      for (const char* name : {"librccl.so.1", "librccl.so"})
        if ((api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!api.lib) return;
    auto sym = [&](const char* n) { return dlsym(api.lib, n); };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
    api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
    api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
    api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.GroupStart && api.GroupEnd && api.Send && api.Recv;
  });
  return &api;
}
#define NCCL_TRY(expr)                                                                                      \
  do {                                                                                                      \
    ncclResult_t _r = (expr);                                                                               \
    if (_r != ncclSuccess)                                                                                  \
      return fail(KV_INTERNAL, "%s failed: %s", #expr, rccl()->GetErrorString ? rccl()->GetErrorString(_r) : "?"); \
  } while (0)
}  // namespace

struct kv_comm {
  ncclComm_t comm = nullptr;
  int world = 1, rank = 0, device = 0;
  hipStream_t stream = nullptr;      // the collectives' own stream
  hipEvent_t ev_in = nullptr, ev_out = nullptr;
  // host-staged transport (kv_comm_create_staged): the caller's callbacks move the segments
  kv_comm_exchange_fn xfn = nullptr;
  kv_comm_max_fn mfn = nullptr;
  void* user = nullptr;
};
// the segments really travel (RCCL or the caller's transport); else: a world of one whose exchange is a device copy
static inline bool wired(const kv_comm* c) { return c->comm != nullptr || c->xfn != nullptr; }

struct kv_shard {
  kv_table* table = nullptr;         // this rank's share of the rows
  kv_table* route = nullptr;         // owns the workspace of the local unique index (no rows of its own)
  int world = 1, rank = 0, rule = KV_OWNER_HASH;
  long long max_ids = 0;
  unsigned C = 0;                    // (id, count) records per peer segment, header not counted
  long long* uniq = nullptr;         // [max_ids]
  int* ucnt = nullptr;               // [max_ids]
  int* slot_of = nullptr;            // [max_ids] unique id -> record in the send buffer
  long long* send_pairs = nullptr;   // [world][C + 1][2]
  long long* recv_pairs = nullptr;
  float* send_rows = nullptr;        // [world][C + 1][dim]
  float* recv_rows = nullptr;
  long long* counts = nullptr;       // [world]
  unsigned* hist = nullptr;          // [world][tiles]
  unsigned* gcount = nullptr;        // [MAXW + 1] k_owner_route_fixed's counters (zero between launches)
  unsigned* overflow = nullptr;      // pinned, mapped: a segment was too small for a batch
  unsigned long long overflows = 0;  // batches reported so far
  unsigned* need = nullptr;          // device [2]: the largest segment the last routed batch wanted; the same over all ranks
  unsigned* need_host = nullptr;     // pinned copy of need[1]
  bool lossless = true;              // ranks agree on the capacity before every exchange (the default: nothing can be lost);
                                     // kv_shard_set_lossless(shard, 0) opts into the synchronisation-free mode
  unsigned long long grows = 0;      // times the capacity was raised
  long long n_last = 0;              // ids of the batch whose index `route` holds
  bool ordered = false;              // ... and whether its positions are filed (order, work items) yet
  bool self_in_place = false;        // this rank's own segments are read from the send buffers (no device copy in the exchange)
  bool route_fused = false;          // ... and whether it is an entry-list index (k_ltile<NOTABLE> + k_papply PA_UNIQUE), not a sorted position list
  const kv_comm* verified = nullptr; // the communicator whose ranks were seen to agree on world / capacity / dim
  uint64_t route_token = 0;
  kv_batch_token_t serve_token = 0;
  hipEvent_t ev_fork = nullptr, ev_done = nullptr;
  int ranks_seen = 0;                // ranks whose handshake record arrived in the first exchange (shard_verify)
  // per-phase timing of the whole ops (kv_shard_profile): events on the communicator's stream at the phase boundaries of
  // every prof_every-th step; one sample in flight, collected by the next sampled call or by the read
  int prof_every = 0;
  unsigned prof_seq_l = 0, prof_seq_a = 0;
  hipEvent_t pev[10] = {};           // lookup: [0..5] = 5 phases; apply: [6..9] = 3 phases
  bool pend_l = false, pend_a = false;
  double psum[KV_SHARD_PHASES] = {};
  long long pcnt[KV_SHARD_PHASES] = {};
};

namespace {
// phase timing of the whole sharded ops (kv_shard_profile): see kv_shard
void shard_prof_collect(kv_shard* sh, bool lookup) {
  bool& pend = lookup ? sh->pend_l : sh->pend_a;
  if (!pend) return;
  const int e0 = lookup ? 0 : 6, np = lookup ? 5 : 3, p0 = lookup ? 0 : 5;
  if (hipEventSynchronize(sh->pev[e0 + np]) == hipSuccess)
    for (int i = 0; i < np; ++i) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, sh->pev[e0 + i], sh->pev[e0 + i + 1]) == hipSuccess) { sh->psum[p0 + i] += ms; sh->pcnt[p0 + i] += 1; }
    }
  pend = false;
}
// does this call carry markers?  (collects the previous sample first: its events are about to be reused)
bool shard_prof_begin(kv_shard* sh, bool lookup) {
  if (sh->prof_every <= 0) return false;
  unsigned& seq = lookup ? sh->prof_seq_l : sh->prof_seq_a;
  if ((seq++ % (unsigned)sh->prof_every) != 0u) return false;
  shard_prof_collect(sh, lookup);
  for (hipEvent_t& e : sh->pev)
    if (!e && hipEventCreate(&e) != hipSuccess) return false;
  return true;
}
inline void shard_prof_mark(kv_shard* sh, bool on, int ev, hipStream_t w) { if (on) hipEventRecord(sh->pev[ev], w); }

void shard_free_buffers(kv_shard* sh) {
  hipFree(sh->send_pairs); hipFree(sh->recv_pairs); hipFree(sh->send_rows); hipFree(sh->recv_rows);
  sh->send_pairs = sh->recv_pairs = nullptr; sh->send_rows = sh->recv_rows = nullptr;
}
int shard_alloc_buffers(kv_shard* sh, unsigned C) {
  const size_t rec = (size_t)sh->world * (C + 1);
  if (rec > (1ull << 21)) return fail(KV_INVALID_ARGUMENT, "peer_capacity %u x world %d exceeds 2^21 records per exchange", C, sh->world);
  shard_free_buffers(sh);
  HIP_TRY(hipMalloc(&sh->send_pairs, rec * 16));
  HIP_TRY(hipMalloc(&sh->recv_pairs, rec * 16));
  HIP_TRY(hipMalloc(&sh->send_rows, rec * sh->table->dim * sizeof(float)));
  HIP_TRY(hipMalloc(&sh->recv_rows, rec * sh->table->dim * sizeof(float)));
  HIP_TRY(hipMemset(sh->send_pairs, 0, rec * 16));
  HIP_TRY(hipMemset(sh->recv_pairs, 0, rec * 16));
  sh->C = C;
  return KV_OK;
}
}  // namespace

extern "C" {

int kv_comm_unique_id(void* id128) {
  if (!id128) return fail(KV_INVALID_ARGUMENT, "id128 is null");
  if (!rccl()->ok) return fail(KV_UNIMPLEMENTED, "librccl.so.1 could not be loaded");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId");
  NCCL_TRY(rccl()->GetUniqueId(reinterpret_cast<ncclUniqueId*>(id128)));
  return KV_OK;
}

int kv_comm_create(int world, int rank, const void* id128, int device, kv_comm_t* out) {
  if (!out || world < 1 || world > MAXW || rank < 0 || rank >= world) return fail(KV_INVALID_ARGUMENT, "kv_comm_create: world %d rank %d", world, rank);
  DeviceGuard dg(device);
  kv_comm* c = new kv_comm();
  c->world = world; c->rank = rank; c->device = device;
  int rc = KV_OK;
  do {
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming) != hipSuccess) {
      rc = fail(KV_INTERNAL, "kv_comm_create: stream / events");
      break;
    }
    if (world > 1 || id128) {   // a world of one needs no RCCL (the exchange is a device copy) unless asked for
      if (!id128) { rc = fail(KV_INVALID_ARGUMENT, "kv_comm_create: unique id is null"); break; }
      if (!rccl()->ok) { rc = fail(KV_UNIMPLEMENTED, "librccl.so.1 could not be loaded"); break; }
      ncclUniqueId id;
      std::memcpy(&id, id128, sizeof id);
      ncclResult_t r = rccl()->CommInitRank(&c->comm, world, id, rank);
      if (r != ncclSuccess) { rc = fail(KV_INTERNAL, "ncclCommInitRank: %s", rccl()->GetErrorString ? rccl()->GetErrorString(r) : "?"); break; }
    }
  } while (0);
  if (rc) { kv_comm_destroy(c); return rc; }
  *out = c;
  return KV_OK;
}

int kv_comm_create_staged(int world, int rank, kv_comm_exchange_fn exchange, kv_comm_max_fn max_u32, void* user, int device,
                          kv_comm_t* out) {
  if (!out || !exchange || world < 1 || world > MAXW || rank < 0 || rank >= world)
    return fail(KV_INVALID_ARGUMENT, "kv_comm_create_staged: world %d rank %d", world, rank);
  DeviceGuard dg(device);
  kv_comm* c = new kv_comm();
  c->world = world; c->rank = rank; c->device = device;
  c->xfn = exchange; c->mfn = max_u32; c->user = user;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming) != hipSuccess) {
    kv_comm_destroy(c);
    return fail(KV_INTERNAL, "kv_comm_create_staged: stream / events");
  }
  *out = c;
  return KV_OK;
}

int kv_comm_stream(kv_comm_t c, kv_stream_t* stream) {
  if (!c || !stream) return fail(KV_INVALID_ARGUMENT, "kv_comm_stream: null argument");
  *stream = (kv_stream_t)c->stream;
  return KV_OK;
}

int kv_forget_stream(kv_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (hipStreamSynchronize(s) != hipSuccess) return fail(KV_INVALID_ARGUMENT, "kv_forget_stream: the stream cannot be synchronised");
  retire_stream(s);
  return KV_OK;
}

int kv_comm_destroy(kv_comm_t c) {
  if (!c) return KV_OK;
  DeviceGuard dg(c->device);
  if (c->stream) { hipStreamSynchronize(c->stream); retire_stream(c->stream); }   // no table keeps the stream as its last one
  if (c->comm && rccl()->ok) rccl()->CommDestroy(c->comm);
  if (c->ev_in) hipEventDestroy(c->ev_in);
  if (c->ev_out) hipEventDestroy(c->ev_out);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
  return KV_OK;
}

// bytes_per_peer bytes to / from every rank: grouped ncclSend / ncclRecv (xGMI is point to point: one pair per
// link), on the communicator's own stream, behind everything `stream` was given and in front of what it gets next.
// nseg buffers (the tables of a multi-table step) go out in ONE group: RCCL runs a group's sends and receives as one
// launch, so 40 tables cost one exchange, not 40.
// skip_self[k] (may be null): segment k of this rank is read from its send buffer by the kernels themselves
static int comm_exchange(kv_comm* c, int nseg, const void* const* sends, void* const* recvs, const int64_t* bytes_per_peer, hipStream_t s,
                         const char* skip_self = nullptr) {
  if (c->xfn) {   // the caller's transport: everything queued so far has run, then the segments move on the host's clock
    HIP_TRY(hipStreamSynchronize(s));
    for (int k = 0; k < nseg; ++k) {
      const int r = c->xfn(c->user, sends[k], recvs[k], bytes_per_peer[k]);
      if (r) return fail(KV_INTERNAL, "the staged exchange failed (callback returned %d)", r);
    }
    return KV_OK;
  }
  if (!c->comm) {   // world of one without RCCL
    for (int k = 0; k < nseg; ++k)
      if (!(skip_self && skip_self[k]))
        HIP_TRY(hipMemcpyAsync(recvs[k], sends[k], (size_t)bytes_per_peer[k], hipMemcpyDeviceToDevice, s));
    return KV_OK;
  }
  const bool hop = s != c->stream;   // the sharded ops run on the communicator's stream themselves: no hop
  if (hop) {
    HIP_TRY(hipEventRecord(c->ev_in, s));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_in, 0));
  }
  // this rank's own segment never meets RCCL: it stays in place (skip_self: the sharded whole ops read it in the send
  // buffer) or is a device copy at HBM speed, queued in front of the group
  // (KV_COMM_SELF_VIA_RCCL=1, a test switch: the self segment goes through ncclSend / ncclRecv like a peer's, which
  // lets a single GPU exercise the grouped send / recv code)
  static const bool self_rccl = [] { const char* e = getenv("KV_COMM_SELF_VIA_RCCL"); return e && e[0] == '1'; }();
  if (!self_rccl)
    for (int k = 0; k < nseg; ++k)
      if (!(skip_self && skip_self[k]))
        HIP_TRY(hipMemcpyAsync((char*)recvs[k] + (size_t)c->rank * bytes_per_peer[k], (const char*)sends[k] + (size_t)c->rank * bytes_per_peer[k],
                             (size_t)bytes_per_peer[k], hipMemcpyDeviceToDevice, c->stream));
  const bool grouped = c->world > 1 || self_rccl;
  if (grouped) NCCL_TRY(rccl()->GroupStart());
  ncclResult_t bad = ncclSuccess;   // a failed Send / Recv must not leave the group open on this communicator
  for (int p = 0; p < c->world && bad == ncclSuccess; ++p) {
    if (p == c->rank && !self_rccl) continue;
    for (int k = 0; k < nseg && bad == ncclSuccess; ++k) {   // peer-major: both ends of a pair post the tables in the same order
      const size_t b = (size_t)bytes_per_peer[k];
      bad = rccl()->Send((const char*)sends[k] + (size_t)p * b, b, ncclChar, p, c->comm, c->stream);
      if (bad == ncclSuccess) bad = rccl()->Recv((char*)recvs[k] + (size_t)p * b, b, ncclChar, p, c->comm, c->stream);
    }
  }
  if (grouped) {
    const ncclResult_t e = rccl()->GroupEnd();
    if (bad == ncclSuccess) bad = e;
  }
  if (bad != ncclSuccess)
    return fail(KV_INTERNAL, "grouped send / recv failed: %s", rccl()->GetErrorString ? rccl()->GetErrorString(bad) : "?");
  if (hop) {
    HIP_TRY(hipEventRecord(c->ev_out, c->stream));
    HIP_TRY(hipStreamWaitEvent(s, c->ev_out, 0));
  }
  return KV_OK;
}
int kv_comm_all_to_all(kv_comm_t c, const void* send, void* recv, int64_t bytes_per_peer, kv_stream_t stream) {
  if (!c || !send || !recv || bytes_per_peer < 0) return fail(KV_INVALID_ARGUMENT, "kv_comm_all_to_all: bad arguments");
  DeviceGuard dg(c->device);
  return comm_exchange(c, 1, &send, &recv, &bytes_per_peer, (hipStream_t)stream);
}

int kv_shard_create(kv_handle_t local_table, int world, int rank, int owner_rule, int64_t max_ids, int64_t peer_capacity,
                    kv_shard_t* out) {
  int rc;
  if ((rc = check_table(local_table))) return rc;
  if (!out || world < 1 || world > MAXW || rank < 0 || rank >= world) return fail(KV_INVALID_ARGUMENT, "kv_shard_create: world %d rank %d", world, rank);
  if (owner_rule != KV_OWNER_HASH && owner_rule != KV_OWNER_MOD) return fail(KV_INVALID_ARGUMENT, "owner_rule %d", owner_rule);
  if (max_ids < 1 || max_ids > (1ll << 21)) return fail(KV_INVALID_ARGUMENT, "max_ids %lld: 1 .. 2^21 ids per sharded batch", (long long)max_ids);
  if (local_table->key_dtype == KV_DT_INT32) return fail(KV_UNIMPLEMENTED, "sharded tables carry int64 ids");
  if ((local_table->dim & 3) != 0) return fail(KV_UNIMPLEMENTED, "sharded tables: dim %d (multiples of 4)", local_table->dim);
  if (fused_off())   // (the owners' serve side exists on the entry-list kernels alone)
    return fail(KV_UNIMPLEMENTED, "KV_NO_FUSED=1 is an A/B switch of the single-table ops: no sharded tables under it");
  if (local_table->occurrence_order)
    return fail(KV_UNIMPLEMENTED, "sharded tables: occurrence-order mode is a single-table notion (senders pre-sum, owners add the senders' sums in rank order)");
  DeviceGuard dg(local_table->device);
  kv_shard* sh = new kv_shard();
  sh->table = local_table; sh->world = world; sh->rank = rank; sh->rule = owner_rule; sh->max_ids = max_ids;
  local_table->shard_refs.fetch_add(1);
  do {
    if ((rc = kv_create(KV_DT_INT64, KV_DT_FLOAT, local_table->dim, 0, 0, local_table->device, &sh->route))) break;
    // default capacity: twice an even share of max_ids distinct ids (hashed ownership spreads them evenly), at least 1024
    long long C = peer_capacity > 0 ? peer_capacity : std::max<long long>(1024, 2 * ((max_ids + world - 1) / world));
    C = std::min<long long>(C, max_ids);
    const unsigned ntr = (unsigned)((max_ids + RT - 1) / RT);
    if (hipMalloc(&sh->uniq, (size_t)max_ids * 8) != hipSuccess || hipMalloc(&sh->ucnt, (size_t)max_ids * 4) != hipSuccess ||
        hipMalloc(&sh->slot_of, (size_t)max_ids * 4) != hipSuccess || hipMalloc(&sh->counts, (size_t)world * 8) != hipSuccess ||
        hipMalloc(&sh->hist, (size_t)ntr * world * 4) != hipSuccess ||
        hipMalloc(&sh->gcount, (MAXW + 1) * 4) != hipSuccess || hipMemset(sh->gcount, 0, (MAXW + 1) * 4) != hipSuccess ||
        hipHostMalloc(&sh->overflow, 2 * sizeof(unsigned), hipHostMallocMapped) != hipSuccess ||
        hipMalloc(&sh->need, 2 * sizeof(unsigned)) != hipSuccess || hipMemset(sh->need, 0, 2 * sizeof(unsigned)) != hipSuccess ||
        hipHostMalloc(&sh->need_host, sizeof(unsigned), hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&sh->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&sh->ev_done, hipEventDisableTiming) != hipSuccess) {
      rc = fail(KV_RESOURCE_EXHAUSTED, "kv_shard_create: allocation failed");
      break;
    }
    sh->overflow[0] = 0; sh->overflow[1] = 0;   // [1]: the last routed batch's distinct ids (k_seg_headers_take)
    if ((rc = shard_alloc_buffers(sh, (unsigned)C))) break;
  } while (0);
  if (rc) { kv_shard_destroy(sh); return rc; }
  *out = sh;
  return KV_OK;
}

int kv_shard_destroy(kv_shard_t sh) {
  if (!sh) return KV_OK;
  DeviceGuard dg(sh->table->device);
  sh->table->shard_refs.fetch_sub(1);
  hipDeviceSynchronize();
  shard_free_buffers(sh);
  hipFree(sh->uniq); hipFree(sh->ucnt); hipFree(sh->slot_of); hipFree(sh->counts); hipFree(sh->hist); hipFree(sh->gcount);
  if (sh->overflow) hipHostFree(sh->overflow);
  if (sh->need) hipFree(sh->need);
  if (sh->need_host) hipHostFree(sh->need_host);
  if (sh->ev_fork) hipEventDestroy(sh->ev_fork);
  if (sh->ev_done) hipEventDestroy(sh->ev_done);
  for (hipEvent_t e : sh->pev) if (e) hipEventDestroy(e);
  if (sh->route) kv_destroy(sh->route);
  delete sh;
  return KV_OK;
}

int kv_shard_buffers(kv_shard_t sh, void** send_pairs, void** recv_pairs, void** send_rows, void** recv_rows,
                     int64_t* pair_bytes_per_peer, int64_t* row_bytes_per_peer) {
  if (!sh) return fail(KV_INVALID_ARGUMENT, "null shard");
  if (send_pairs) *send_pairs = sh->send_pairs;
  if (recv_pairs) *recv_pairs = sh->recv_pairs;
  if (send_rows) *send_rows = sh->send_rows;
  if (recv_rows) *recv_rows = sh->recv_rows;
  if (pair_bytes_per_peer) *pair_bytes_per_peer = (int64_t)(sh->C + 1) * 16;
  if (row_bytes_per_peer) *row_bytes_per_peer = (int64_t)(sh->C + 1) * sh->table->dim * (int64_t)sizeof(float);
  return KV_OK;
}

// A batch that sent one owner more than peer_capacity distinct ids raised the pinned flag (the surplus read zeros).
// It is reported by the first call that ENDS after the flag landed — after that call has queued all its work, so a
// rank that reports keeps step with its peers (an early return would leave them waiting in the exchange).  The
// capacity is not changed here: it must change on every rank at once.
// (`seen` is the flag as the call found it when it STARTED: a batch's own overflow is reported by the next call, never
// by itself — which call reports does not depend on how fast the kernels ran.)
static unsigned shard_take_flag(kv_shard* sh) {
  return __atomic_exchange_n(sh->overflow, 0u, __ATOMIC_RELAXED);
}
static int shard_late_report(kv_shard* sh, unsigned seen) {
  if (!seen) return KV_OK;
  ++sh->overflows;
  return fail(KV_RESOURCE_EXHAUSTED, "an earlier sharded batch sent one owner more than peer_capacity (%u) distinct ids: the surplus "
                                     "ids read zeros and their gradients were dropped (this call itself was queued in full); "
                                     "create the shards with a larger peer_capacity on every rank", sh->C);
}

// The whole-op entry points (kv_shard_lookup / kv_shard_apply / kv_multi_shard_*) do the exchanges themselves, so they
// leave this rank's own segments where they are: every kernel that reads a receive buffer downstream (the owner lookup's
// tile pass, the finish, the owner apply's tile sums and k_papply) takes records [rank * (C + 1), +C + 1) from the send
// buffer.  Only the entry-list kernels know how (other dims: the segment is copied like any peer's).
static bool shard_can_stay(const kv_shard* sh) {
  return fused_ok(sh->table->dim) && (long long)sh->world * (sh->C + 1) <= FUSED_MAX_N;
}

// ids -> local unique ids with counts -> the owners' segments of the send buffer.  4 launches, no host sync.
static int lookup_route_impl(kv_shard_t sh, const void* ids, int64_t n, kv_stream_t stream) {
  if (!sh || (n > 0 && !ids) || n < 0 || n > sh->max_ids) return fail(KV_INVALID_ARGUMENT, "kv_shard_lookup_route: n %lld (max %lld)", (long long)n, sh ? sh->max_ids : 0ll);
  DeviceGuard dg(sh->table->device);
  hipStream_t s = (hipStream_t)stream;
  int rc;
  kv_table* rt = sh->route;
  std::lock_guard<std::mutex> l(rt->mu);
  if ((rc = hand_over(rt, s))) return rc;   // the phases of one shard keep their order whatever streams they are given
  rt->deterministic = sh->table->deterministic;   // the route index (tile pass, position order) follows the table's mode
  sh->n_last = n;
  sh->route_token = 0;
  if (n == 0) {
    HIP_TRY(hipMemsetAsync(sh->counts, 0, (size_t)sh->world * 8, s));
    k_seg_headers<<<1, MAXW, 0, s>>>(sh->counts, sh->world, sh->C, sh->send_pairs, sh->need);
    return KV_OK;
  }
  if ((rc = ensure_workspace(rt, n, true, s))) return rc;
  WsDev wd = ws_view(rt, n);
  // The route's index on the entry-list kernels (dims they do not serve: the sorted-position kernels):
  // a table-less tile pass (k_ltile<NOTABLE>: entries, mrow, every position's entry number) and k_papply in PA_UNIQUE
  // mode (the distinct ids numbered, uniq / ucnt written, every entry learns its id's number).  The finish then reads
  // position -> entry -> number -> record, the gradient pre-sum is k_tsum + k_papply PA_DEDUP.
  const bool fused_route = fused_ok(rt->dim);
  sh->route_fused = fused_route;
  PartArgs pa{};
  pa.tv = dev_view(rt); pa.ts0 = pa.tv; pa.ts1 = pa.tv;
  pa.out_keys = sh->uniq;
  pa.out_counts = sh->ucnt;
  pa.sparse_unique = 1;   // unique numbers with gaps: no counter to serialise on
  pa.det = sh->table->deterministic ? 1 : 0;
  pa.n = n;
  if (fused_route) {
    Workspace& ws = rt->ws;
    if (ws.pos_cap < n) {
      HIP_TRY(hipStreamSynchronize(s));
      ws.pos_cap = 0;
      if ((rc = regrow(&ws.pos_ent, (size_t)std::max<long long>(n, ws.cap_n)))) return rc;
      ws.pos_cap = std::max<long long>(n, ws.cap_n);
    }
    rt->fused_index = true;
    choose_partitions(rt, wd, n);   // (the distinct-id hint of the previous route of this length: k_papply publishes it)
    WsDev wz = wd;
    wz.zero_counts = sh->ucnt;
    wz.pos_ent = ws.pos_ent;
    launch_ltile_notable(rt, pa.tv, wz, ids, n, s);
    pa.day_lk = pa.day;
    if (!pa.det) {   // the numbered ids go straight to their owners' segments (the deterministic mode keeps the ordered scatter below)
      pa.route_world = sh->world; pa.route_rule = sh->rule; pa.route_C = sh->C;
      pa.route_seg = sh->send_pairs; pa.route_slot_of = sh->slot_of; pa.route_overflow = sh->overflow; pa.route_gcount = sh->gcount;
      pa.route_need = sh->need; pa.route_uhint = sh->overflow + 1;   // (the launch's last block writes the headers: no k_seg_headers_take)
    }
    if ((rc = kvp_launch_papply_ud(&wd, &pa, PA_UNIQUE, (void*)s))) return fail(rc, "route: no kernel for dim %d", rt->dim);
  } else {
    {
      // a skewed batch (the previous one of this length held at most n / 8 distinct ids; pinned word, no
      // synchronisation): half the partitions, as in the unsharded index pass (fused_index_pass)
      const unsigned u_prev = reinterpret_cast<volatile unsigned*>(sh->overflow)[1];
      const bool few = !rt->deterministic && rt->batch_n_prev == n && u_prev > 0u && (long long)u_prev * 8 <= n && n >= (1ll << 18);
      rt->batch_n_prev = n;
      if (few) { wd.P = std::max(64u, wd.P / 2u); wd.pshift = 64 - ilog2(wd.P); }
      rt->index_P = wd.P;
    }
    rt->fused_index = false;
    // tile + partition passes only: kv_shard_lookup_finish's gather files the positions (order, work items) on its way
    WsDev wz = wd;
    wz.zero_counts = sh->ucnt;   // sparse unique numbers: the tile pass clears the counts, the partition pass sets the real ones
    launch_tile<false>(rt, wz, ids, nullptr, n, s, 0);
    launch_part_keys<MODE_UNIQUE>(wd, pa, s);
  }
  rt->batch_serial = ++g_serial;
  rt->batch_n = n;
  sh->route_token = rt->batch_serial;
  sh->ordered = false;
  const unsigned ntr = (unsigned)((n + RT - 1) / RT);
  if (!pa.det) {
    if (!fused_route)
      k_owner_route_fixed<<<ntr, TB, 0, s>>>(sh->uniq, sh->ucnt, (long long)n, sh->world, sh->rule, sh->C, sh->send_pairs, sh->slot_of,
                                             sh->overflow, sh->gcount);
    if (!fused_route) k_seg_headers_take<<<1, MAXW, 0, s>>>(sh->gcount, sh->world, sh->C, sh->send_pairs, sh->need, sh->overflow + 1);
    HIP_TRY(hipGetLastError());
    return KV_OK;
  }
  k_owner_hist_u32<<<ntr, TB, 0, s>>>(sh->uniq, sh->ucnt, (long long)n, sh->world, sh->rule, ntr, sh->hist);
  k_owner_scan<<<1, 1024, 0, s>>>(sh->hist, ntr * sh->world, ntr, sh->world, sh->counts);
  k_seg_headers<<<1, MAXW, 0, s>>>(sh->counts, sh->world, sh->C, sh->send_pairs, sh->need);
  k_owner_scatter_fixed<<<ntr, TB, 0, s>>>(sh->uniq, sh->ucnt, (long long)n, sh->world, sh->rule, ntr, sh->hist, sh->C, sh->send_pairs,
                                           sh->slot_of, sh->overflow);
  HIP_TRY(hipGetLastError());
  return KV_OK;
}

int kv_shard_lookup_route(kv_shard_t sh, const void* ids, int64_t n, kv_stream_t stream) {
  int rc;
  const unsigned seen = sh ? shard_take_flag(sh) : 0u;
  if (sh) sh->self_in_place = false;   // the caller makes the exchanges: all segments, its own included, arrive in the receive buffers
  if ((rc = lookup_route_impl(sh, ids, n, stream))) return rc;
  return shard_late_report(sh, seen);
}

// the owner's half: the ids the peers sent (recv_pairs) are looked up in this rank's table — frequency words count
// every occurrence — and their rows go to send_rows, record for record.  3 launches.
int kv_shard_lookup_serve(kv_shard_t sh, kv_stream_t stream) {
  if (!sh) return fail(KV_INVALID_ARGUMENT, "null shard");
  const int64_t nrec = (int64_t)sh->world * (sh->C + 1);
  sh->serve_token = 0;
  SelfScope self(sh->table, sh->self_in_place, (unsigned)sh->rank * (sh->C + 1), sh->C + 1, sh->send_pairs, nullptr);
  return gather_or_insert_impl(sh->table, sh->recv_pairs, nullptr, nrec, sh->send_rows, stream, 1, &sh->serve_token, sh->C + 1);
}

// out[i] = the row that came back for ids[i].  1 launch.
int kv_shard_lookup_finish(kv_shard_t sh, float* out, kv_stream_t stream) {
  if (!sh || (sh->n_last > 0 && !out)) return fail(KV_INVALID_ARGUMENT, "kv_shard_lookup_finish: output pointer is null");
  if (sh->n_last == 0) return KV_OK;
  DeviceGuard dg(sh->table->device);
  kv_table* rt = sh->route;
  std::lock_guard<std::mutex> l(rt->mu);
  if (rt->batch_serial != sh->route_token || sh->route_token == 0) return fail(KV_FAILED_PRECONDITION, "kv_shard_lookup_finish without kv_shard_lookup_route");
  { int rc; if ((rc = hand_over(rt, (hipStream_t)stream))) return rc; }
  // the training lookup's gather (k_gather<ORDER>) over the rows that came back: position -> entry -> dense unique
  // index -> the record its id was sent in; the same pass files the positions for the gradient sum to come
  if (sh->route_fused) {
    const int q = row_lanes(rt->dim);
    const int grid = nblocks(sh->n_last, TB, 8192);
    hipStream_t st = (hipStream_t)stream;
    const unsigned slo = (unsigned)sh->rank * (sh->C + 1), slen = sh->self_in_place ? sh->C + 1 : 0u;
#define KV_SF(VQ) k_shard_finish<VQ><<<grid, TB, 0, st>>>(rt->ws.pos_ent, rt->ws.ent_b, sh->slot_of, sh->recv_rows, out, sh->n_last, rt->dim, \
                                                          sh->send_rows, slo, slen)
    switch (q) {
      case 1: KV_SF(1); break;   case 2: KV_SF(2); break;   case 4: KV_SF(4); break;   case 8: KV_SF(8); break;
      case 16: KV_SF(16); break; case 32: KV_SF(32); break; default: KV_SF(64); break;
    }
#undef KV_SF
    HIP_TRY(hipGetLastError());
    return KV_OK;
  }
  WsDev wd = ws_view(rt, sh->n_last);
  if (rt->index_P) { wd.P = rt->index_P; wd.pshift = 64 - ilog2(wd.P); }   // the route's partitioning
  wd.row_map = sh->slot_of;
  TableDev rows = dev_view(rt);
  rows.c0.rows = sh->recv_rows;
  rows.chunk_bits = 31;
  launch_gather(rows, wd, out, sh->n_last, (hipStream_t)stream, nullptr, 0, true);
  sh->ordered = true;
  HIP_TRY(hipGetLastError());
  return KV_OK;
}

// backward: the gradient rows of the batch just looked up are summed per distinct id straight into the records their
// ids were sent in (the route index is still there: no ids, no sizes, no sync).  2 launches.
int kv_shard_apply_route(kv_shard_t sh, const float* grad, kv_stream_t stream) {
  if (!sh || (sh->n_last > 0 && !grad)) return fail(KV_INVALID_ARGUMENT, "kv_shard_apply_route: grad pointer is null");
  if (sh->n_last == 0) return KV_OK;
  DeviceGuard dg(sh->table->device);
  kv_table* rt = sh->route;
  std::lock_guard<std::mutex> l(rt->mu);
  if (rt->batch_serial != sh->route_token || sh->route_token == 0)
    return fail(KV_FAILED_PRECONDITION, "kv_shard_apply_route: the batch's lookup must come first (kv_shard_lookup_route)");
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if ((rc = hand_over(rt, s))) return rc;
  if ((rc = ensure_workspace(rt, sh->n_last, true, s))) return rc;
  WsDev wd = ws_view(rt, sh->n_last);
  if (rt->index_P) { wd.P = rt->index_P; wd.pshift = 64 - ilog2(wd.P); }   // the route's partitioning
  if (sh->route_fused) {
    // the tile sums of the ids repeated inside their tile, then the per-id sums straight into the records the ids were
    // sent in (k_papply PA_DEDUP over the route's entries)
    PartArgs pa{};
    pa.tv = dev_view(rt); pa.ts0 = pa.tv; pa.ts1 = pa.tv;
    pa.grad = grad;
    pa.out_sum = sh->send_rows;
    pa.out_map = sh->slot_of;
    pa.det = rt->deterministic ? 1 : 0;
    pa.n = sh->n_last;
    pa.epart = wd.epart;
    pa.day_lk = pa.day;
    if ((rc = kvp_launch_tsum(&pa.tv, &wd, grad, (void*)s, nullptr, 0))) return fail(rc, "tile sums: no kernel for dim %d", rt->dim);
    if ((rc = kvp_launch_papply_ud(&wd, &pa, PA_DEDUP, (void*)s))) return fail(rc, "gradient pre-sum: no kernel for dim %d", rt->dim);
    HIP_TRY(hipGetLastError());
    return KV_OK;
  }
  if (!sh->ordered) {   // a gradient for a batch whose rows were never fetched (kv_shard_lookup_finish skipped)
    launch_order(dev_view(rt), wd, sh->n_last, s);
    sh->ordered = true;
  }
  PartArgs pa{};
  pa.tv = dev_view(rt); pa.ts0 = pa.tv; pa.ts1 = pa.tv;
  pa.grad = grad;
  pa.out_sum = sh->send_rows;
  pa.out_map = sh->slot_of;
  pa.fold_op = KV_SCATTER_ADD;
  pa.det = rt->deterministic ? 1 : 0;
  pa.n = sh->n_last;
  return launch_apply<MODE_DEDUP, OPT_ADAGRAD>(rt, wd, pa, sh->n_last, s);
}

// the owner's half: recv_rows holds the peers' summed gradients, record for record as the lookup served them; the
// fused apply takes the index that lookup left in the table's workspace.  2 launches.
// optimizer: 0 GroupAdam V4, 1 GroupAdam V3 (hp = lr, beta1_power, beta2_power, beta1, beta2, epsilon, l1, l2, l21),
// 2 Adagrad (hp = lr, update_slots), 3 SparseGroupFtrl (hp = lr, l1, l2, l21, l2_shrinkage, lr_power; slot1 = linear)
int kv_shard_apply_serve(kv_shard_t sh, int optimizer, kv_handle_t slot0, kv_handle_t slot1, const float* hp, kv_stream_t stream) {
  if (!sh || !hp) return fail(KV_INVALID_ARGUMENT, "kv_shard_apply_serve: bad arguments");
  if (sh->serve_token == 0 || sh->serve_token != sh->table->batch_serial)
    return fail(KV_FAILED_PRECONDITION, "kv_shard_apply_serve: another op used the table since this batch's lookup "
                                        "(the sharded apply takes over the lookup's index)");
  const int64_t nrec = (int64_t)sh->world * (sh->C + 1);
  SelfScope self(sh->table, sh->self_in_place, (unsigned)sh->rank * (sh->C + 1), sh->C + 1, sh->send_pairs, sh->send_rows);
  switch (optimizer) {
    case 0: case 1:
      return kv_apply_group_adam_tok(sh->table, slot0, sh->recv_rows, sh->recv_pairs, nrec, hp[0], hp[1], hp[2], hp[3], hp[4], hp[5],
                                     hp[6], hp[7], hp[8], optimizer == 0 ? 4 : 3, sh->serve_token, stream);
    case 2:
      return kv_apply_adagrad_tok(sh->table, slot0, hp[0], sh->recv_rows, sh->recv_pairs, nrec, hp[1] != 0.f, sh->serve_token, stream);
    case 3:
      return kv_apply_sparse_group_ftrl_tok(sh->table, slot0, slot1, sh->recv_rows, sh->recv_pairs, nrec, hp[0], hp[1], hp[2], hp[3],
                                            hp[4], hp[5], sh->serve_token, stream);
    default:
      return fail(KV_INVALID_ARGUMENT, "kv_shard_apply_serve: optimizer %d", optimizer);
  }
}

// whole ops: forked from `stream` onto the shard's own stream (the caller's stream is free for the dense tower);
// join != 0 makes `stream` wait for the result right away, else kv_shard_join does when the caller needs it
int kv_shard_join(kv_shard_t sh, kv_stream_t stream) {
  if (!sh) return fail(KV_INVALID_ARGUMENT, "null shard");
  DeviceGuard dg(sh->table->device);
  HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, sh->ev_done, 0));
  return KV_OK;
}
static int shard_exchange_one(kv_comm* comm, const void* send, void* recv, int64_t bytes_per_peer, hipStream_t w, char stay) {
  return comm_exchange(comm, 1, &send, &recv, &bytes_per_peer, w, &stay);
}
static int shard_fork(kv_shard* sh, hipStream_t s, hipStream_t work) {
  if (s == work) return KV_OK;   // the caller works on the communicator's stream itself: one queue, no hops
  HIP_TRY(hipEventRecord(sh->ev_fork, s));
  HIP_TRY(hipStreamWaitEvent(work, sh->ev_fork, 0));
  return KV_OK;
}
static int shard_done(kv_shard* sh, hipStream_t s, hipStream_t work, int join) {
  HIP_TRY(hipEventRecord(sh->ev_done, work));
  if (join && s != work) HIP_TRY(hipStreamWaitEvent(s, sh->ev_done, 0));
  return KV_OK;
}

// Once per (shard, communicator): every rank tells every other its {world, rank, capacity, dim, owner rule}.  The
// exchange has no size negotiation, so ranks that disagree would otherwise hang in RCCL or read each other's padding.
static int shard_verify(kv_shard* sh, kv_comm* comm) {
  if (sh->verified == comm) return KV_OK;
  const int W = sh->world;
  std::vector<long long> mine((size_t)W * 4), theirs((size_t)W * 4, 0);
  for (int p = 0; p < W; ++p) {
    mine[4 * p] = ((long long)W << 32) | (unsigned)sh->rank;
    mine[4 * p + 1] = sh->C; mine[4 * p + 2] = sh->table->dim; mine[4 * p + 3] = sh->rule;
  }
  long long *ds = nullptr, *dr = nullptr;
  HIP_TRY(hipMalloc(&ds, mine.size() * 8));
  if (hipMalloc(&dr, mine.size() * 8) != hipSuccess) { hipFree(ds); return fail(KV_RESOURCE_EXHAUSTED, "kv_shard: out of memory"); }
  int rc = KV_OK;
  do {
    if (hipMemcpy(ds, mine.data(), mine.size() * 8, hipMemcpyHostToDevice) != hipSuccess) { rc = fail(KV_INTERNAL, "kv_shard: copy"); break; }
    if ((rc = kv_comm_all_to_all(comm, ds, dr, wired(comm) ? 32 : 32 * W, comm->stream))) break;
    if (hipStreamSynchronize(comm->stream) != hipSuccess || hipMemcpy(theirs.data(), dr, theirs.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) {
      rc = fail(KV_INTERNAL, "kv_shard: the first exchange failed");
      break;
    }
    for (int p = 0; p < W && !rc; ++p) {
      if ((theirs[4 * p] >> 32) != W || (int)(theirs[4 * p] & 0xFFFFFFFF) != (wired(comm) ? p : sh->rank))
        rc = fail(KV_FAILED_PRECONDITION, "kv_shard: rank %d of the communicator is not shard %d of a world of %d", p, p, W);
      else if (theirs[4 * p + 1] != (long long)sh->C || theirs[4 * p + 2] != sh->table->dim || theirs[4 * p + 3] != sh->rule)
        rc = fail(KV_FAILED_PRECONDITION, "kv_shard: rank %d was created with peer_capacity %lld, dim %lld, owner rule %lld; this "
                                          "rank with %u, %d, %d — they must be equal on every rank",
                  p, theirs[4 * p + 1], theirs[4 * p + 2], theirs[4 * p + 3], sh->C, sh->table->dim, sh->rule);
    }
  } while (0);
  hipFree(ds); hipFree(dr);
  if (!rc) { sh->verified = comm; sh->ranks_seen = W; }
  return rc;
}

// Lossless mode (kv_shard_set_lossless): before anything is exchanged every rank learns the largest segment ANY rank's
// route wanted (one 4-byte all-reduce per table, one stream synchronisation for all tables) and, when that exceeds the
// capacity, every rank raises its capacity to the same new value — they all computed it from the same number — and
// routes the batch again.  grown[k] != 0: shards[k] must be routed again.  The price is the host round trip per lookup
// that the default mode avoids; nothing is ever dropped.
static int shard_agree_many(const kv_shard_t* shards, int ntab, kv_comm* comm, hipStream_t w, char* grown) {
  bool any = false;
  for (int k = 0; k < ntab; ++k) { grown[k] = 0; any = any || shards[k]->lossless; }
  if (!any) return KV_OK;
  if (comm->comm) {
    if (!rccl()->AllReduce) return fail(KV_UNIMPLEMENTED, "lossless sharding needs ncclAllReduce");
    NCCL_TRY(rccl()->GroupStart());
    ncclResult_t bad = ncclSuccess;
    for (int k = 0; k < ntab && bad == ncclSuccess; ++k)
      if (shards[k]->lossless)
        bad = rccl()->AllReduce(shards[k]->need, shards[k]->need + 1, 1, ncclUint32, ncclMax, comm->comm, w);
    const ncclResult_t e = rccl()->GroupEnd();
    if (bad == ncclSuccess) bad = e;
    if (bad != ncclSuccess) return fail(KV_INTERNAL, "ncclAllReduce failed: %s", rccl()->GetErrorString ? rccl()->GetErrorString(bad) : "?");
  }
  for (int k = 0; k < ntab; ++k)
    if (shards[k]->lossless)
      HIP_TRY(hipMemcpyAsync(shards[k]->need_host, shards[k]->need + (comm->comm ? 1 : 0), sizeof(unsigned), hipMemcpyDeviceToHost, w));
  HIP_TRY(hipStreamSynchronize(w));
  if (comm->xfn && comm->world > 1) {   // the caller's transport: the maximum over the ranks, table by table
    if (!comm->mfn) return fail(KV_UNIMPLEMENTED, "lossless sharding over a staged communicator needs its max_u32 callback");
    for (int k = 0; k < ntab; ++k)
      if (shards[k]->lossless) {
        uint32_t v = *shards[k]->need_host;
        const int r = comm->mfn(comm->user, &v);
        if (r) return fail(KV_INTERNAL, "the staged agreement failed (callback returned %d)", r);
        *shards[k]->need_host = v;
      }
  }
  for (int k = 0; k < ntab; ++k) {
    kv_shard* sh = shards[k];
    if (!sh->lossless) continue;
    const unsigned need = *sh->need_host;
    if (need <= sh->C) continue;
    const long long newC = std::min<long long>(sh->max_ids, (long long)need + need / 4 + 64);
    int rc;
    if ((rc = shard_alloc_buffers(sh, (unsigned)newC))) return rc;
    *reinterpret_cast<volatile unsigned*>(sh->overflow) = 0;   // raised by the attempt that is now repeated
    ++sh->grows;
    grown[k] = 1;
  }
  return KV_OK;
}

int kv_shard_profile(kv_shard_t sh, int every) {
  if (!sh) return fail(KV_INVALID_ARGUMENT, "null shard");
  DeviceGuard dg(sh->table->device);
  shard_prof_collect(sh, true); shard_prof_collect(sh, false);
  sh->prof_every = every > 0 ? every : 0;
  sh->prof_seq_l = sh->prof_seq_a = 0;
  for (int i = 0; i < KV_SHARD_PHASES; ++i) { sh->psum[i] = 0.0; sh->pcnt[i] = 0; }
  return KV_OK;
}
int kv_shard_profile_read(kv_shard_t sh, double* ms_sum, int64_t* samples, int n_phases, int64_t* info) {
  if (!sh || !ms_sum || !samples || n_phases < KV_SHARD_PHASES) return fail(KV_INVALID_ARGUMENT, "kv_shard_profile_read: bad arguments");
  DeviceGuard dg(sh->table->device);
  shard_prof_collect(sh, true); shard_prof_collect(sh, false);
  for (int i = 0; i < KV_SHARD_PHASES; ++i) { ms_sum[i] = sh->psum[i]; samples[i] = sh->pcnt[i]; }
  if (info) { info[0] = sh->ranks_seen; info[1] = (int64_t)sh->C; info[2] = (int64_t)sh->grows; info[3] = (int64_t)sh->overflows; }
  return KV_OK;
}

int kv_shard_set_lossless(kv_shard_t sh, int on) {
  if (!sh) return fail(KV_INVALID_ARGUMENT, "null shard");
  sh->lossless = on != 0;
  return KV_OK;
}

// the same agreement between shards that live in ONE process (see kv_shard_exchange_local): call it after every
// shard's kv_shard_lookup_route; *rerouted != 0 means the capacity was raised on all of them and the routes must run again
int kv_shard_agree_local(const kv_shard_t* shards, int world, int* rerouted, kv_stream_t stream) {
  if (!shards || world < 1 || !rerouted) return fail(KV_INVALID_ARGUMENT, "kv_shard_agree_local: bad arguments");
  *rerouted = 0;
  for (int p = 0; p < world; ++p)
    if (!shards[p] || shards[p]->world != world || shards[p]->C != shards[0]->C)
      return fail(KV_INVALID_ARGUMENT, "kv_shard_agree_local: shards differ in world / capacity");
  DeviceGuard dg(shards[0]->table->device);
  hipStream_t s = (hipStream_t)stream;
  unsigned need = 0;
  for (int p = 0; p < world; ++p)
    HIP_TRY(hipMemcpyAsync(shards[p]->need_host, shards[p]->need, sizeof(unsigned), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  for (int p = 0; p < world; ++p) need = std::max(need, *shards[p]->need_host);
  if (need <= shards[0]->C) return KV_OK;
  const long long newC = std::min<long long>(shards[0]->max_ids, (long long)need + need / 4 + 64);
  for (int p = 0; p < world; ++p) {
    int rc;
    if ((rc = shard_alloc_buffers(shards[p], (unsigned)newC))) return rc;
    *reinterpret_cast<volatile unsigned*>(shards[p]->overflow) = 0;
    ++shards[p]->grows;
  }
  *rerouted = 1;
  return KV_OK;
}

int kv_shard_lookup(kv_shard_t sh, kv_comm_t comm, const void* ids, int64_t n, float* out, int join, kv_stream_t stream) {
  if (!sh || !comm || comm->world != sh->world) return fail(KV_INVALID_ARGUMENT, "kv_shard_lookup: shard / communicator mismatch");
  DeviceGuard dg(sh->table->device);
  hipStream_t s = (hipStream_t)stream;
  int rc;
  // The lossless mode's agreement reads a device word on the host: one stream synchronisation, not allowed inside a stream
  // capture.  Refused HERE, before the communicator's stream is forked into the capture and before anything is queued (a
  // refusal behind the fork would leave the side stream unjoined and invalidate the caller's capture).
  if (sh->lossless && (stream_is_capturing(s) || stream_is_capturing(comm->stream)))
    return fail(KV_FAILED_PRECONDITION, "kv_shard_lookup under stream capture: the lossless mode (the default) synchronises once per "
                                        "lookup; capture sharded steps with kv_shard_set_lossless(shard, 0) and a peer_capacity "
                                        "sized for the workload");
  if ((rc = shard_verify(sh, comm))) return rc;
  const unsigned seen = shard_take_flag(sh);
  hipStream_t w = comm->stream;   // phases and exchanges in one queue: no event hop between a kernel and its exchange
  if ((rc = shard_fork(sh, s, w))) return rc;
  int64_t pb = (int64_t)(sh->C + 1) * 16, rb = (int64_t)(sh->C + 1) * sh->table->dim * (int64_t)sizeof(float);
  // A failure of THIS rank's phase (out of memory, a bad argument) must not leave the peers waiting in a grouped recv:
  // the exchanges are queued all the same — void headers for a failed route, zero rows for a failed serve — and the
  // first error is returned once everything is queued.  Only a failed exchange itself returns at once.
  int first = KV_OK;
  std::string first_msg;
  auto note = [&](int r) { if (r && !first) { first = r; first_msg = kv_last_error(); } return r; };
  sh->self_in_place = shard_can_stay(sh);
  char stay = sh->self_in_place ? 1 : 0;
  const bool pm = shard_prof_begin(sh, true);
  shard_prof_mark(sh, pm, 0, w);
  if (note(lookup_route_impl(sh, ids, n, w))) {
    sh->n_last = 0; sh->route_token = 0;
    HIP_TRY(hipMemsetAsync(sh->counts, 0, (size_t)sh->world * 8, w));
    k_seg_headers<<<1, MAXW, 0, w>>>(sh->counts, sh->world, sh->C, sh->send_pairs, sh->need);
  }
  if (sh->lossless) {   // every rank takes part in the agreement, whatever its own route did
    char grown = 0;
    if ((rc = shard_agree_many(&sh, 1, comm, w, &grown))) return rc;
    if (grown) {
      pb = (int64_t)(sh->C + 1) * 16; rb = (int64_t)(sh->C + 1) * sh->table->dim * (int64_t)sizeof(float);
      stay = (sh->self_in_place = shard_can_stay(sh)) ? 1 : 0;   // (the record count changed)
      if (!first && note(lookup_route_impl(sh, ids, n, w))) { sh->n_last = 0; sh->route_token = 0; }
      if (first) {   // this rank's batch was refused: its segments are void in the new buffers too
        HIP_TRY(hipMemsetAsync(sh->counts, 0, (size_t)sh->world * 8, w));
        k_seg_headers<<<1, MAXW, 0, w>>>(sh->counts, sh->world, sh->C, sh->send_pairs, sh->need);
      }
    }
  }
  shard_prof_mark(sh, pm, 1, w);
  if ((rc = shard_exchange_one(comm, sh->send_pairs, sh->recv_pairs, wired(comm) ? pb : pb * sh->world, w, stay))) return rc;
  shard_prof_mark(sh, pm, 2, w);
  if (note(kv_shard_lookup_serve(sh, w)))
    HIP_TRY(hipMemsetAsync(sh->send_rows, 0, (size_t)rb * sh->world, w));
  shard_prof_mark(sh, pm, 3, w);
  if ((rc = shard_exchange_one(comm, sh->send_rows, sh->recv_rows, wired(comm) ? rb : rb * sh->world, w, stay))) return rc;
  shard_prof_mark(sh, pm, 4, w);
  if (!first) note(kv_shard_lookup_finish(sh, out, w));
  shard_prof_mark(sh, pm, 5, w);
  if (pm) sh->pend_l = true;
  if ((rc = shard_done(sh, s, w, join))) return rc;
  if (first) return fail(first, "%s (this rank's exchanges were queued all the same)", first_msg.c_str());
  return shard_late_report(sh, seen);
}

int kv_shard_apply(kv_shard_t sh, kv_comm_t comm, int optimizer, kv_handle_t slot0, kv_handle_t slot1, const float* grad,
                   const float* hp, int join, kv_stream_t stream) {
  if (!sh || !comm || comm->world != sh->world) return fail(KV_INVALID_ARGUMENT, "kv_shard_apply: shard / communicator mismatch");
  DeviceGuard dg(sh->table->device);
  hipStream_t s = (hipStream_t)stream;
  int rc;
  hipStream_t w = comm->stream;
  if ((rc = shard_fork(sh, s, w))) return rc;
  const int64_t rb = (int64_t)(sh->C + 1) * sh->table->dim * (int64_t)sizeof(float);
  // as in kv_shard_lookup: a rank whose route phase failed still takes part in the exchange (zero gradient rows)
  int first = KV_OK;
  std::string first_msg;
  const bool pm = shard_prof_begin(sh, false);
  shard_prof_mark(sh, pm, 6, w);
  if ((first = kv_shard_apply_route(sh, grad, w))) {
    first_msg = kv_last_error();
    HIP_TRY(hipMemsetAsync(sh->send_rows, 0, (size_t)rb * sh->world, w));
  }
  shard_prof_mark(sh, pm, 7, w);
  if ((rc = shard_exchange_one(comm, sh->send_rows, sh->recv_rows, wired(comm) ? rb : rb * sh->world, w, sh->self_in_place ? 1 : 0))) return rc;
  shard_prof_mark(sh, pm, 8, w);
  rc = kv_shard_apply_serve(sh, optimizer, slot0, slot1, hp, w);
  shard_prof_mark(sh, pm, 9, w);
  if (pm) sh->pend_a = true;
  if (rc && !first) { first = rc; first_msg = kv_last_error(); }
  if ((rc = shard_done(sh, s, w, join))) return rc;
  if (first) return fail(first, "%s (this rank's exchange was queued all the same)", first_msg.c_str());
  return KV_OK;
}

// Several sharded tables in one step (the 40 embedding tables of a DCN): every table's route phase, then ONE grouped
// exchange carrying all their segments, every table's serve phase, ONE exchange back, every table's finish — two
// exchanges per lookup and one per apply whatever the number of tables.  Same results as the per-table ops.
static int multi_shard_check(const kv_shard_t* shards, int ntab, kv_comm_t comm, const char* what) {
  if (!shards || ntab < 1 || ntab > 4096 || !comm) return fail(KV_INVALID_ARGUMENT, "%s: bad arguments", what);
  for (int k = 0; k < ntab; ++k) {
    if (!shards[k] || shards[k]->world != comm->world || shards[k]->table->device != shards[0]->table->device)
      return fail(KV_INVALID_ARGUMENT, "%s: shard %d / communicator mismatch", what, k);
    for (int j = 0; j < k; ++j)
      if (shards[j] == shards[k]) return fail(KV_INVALID_ARGUMENT, "%s: shard %d is listed twice", what, k);
  }
  return KV_OK;
}

// ---- several sharded tables, one launch per phase -----------------------------------------------------
// The route-side phases of kv_multi_shard_* run on the shards' private route tables; a table takes part in the batched
// launches when its route is the entry-list one (not the deterministic mode, a dim the kernels serve, a batch that is
// not empty); the others go through the per-table functions as before.
static bool shard_batchable(const kv_shard* sh, int64_t n) {
  return fused_ok(sh->table->dim) && !sh->table->deterministic && n > 0 && n <= sh->max_ids;
}

// route of the tables todo[0..m): the table-less tile pass of all of them (grid.y = table), then k_papply_multi in
// PA_UNIQUE mode — numbering, the owners' segments, and each table's headers by its own last block.  2 launches.
static int multi_route_impl(const kv_shard_t* shards, const int* todo, int m, const void* const* ids, const int64_t* n, hipStream_t s) {
  int rc;
  const int device = shards[todo[0]]->table->device;
  std::vector<kv_table*> rts;
  for (int j = 0; j < m; ++j) rts.push_back(shards[todo[j]]->route);
  MultiLock lock(rts);
  for (int j = 0; j < m; ++j) {
    kv_shard* sh = shards[todo[j]];
    kv_table* rt = sh->route;
    if ((rc = hand_over(rt, s))) return rc;
    rt->deterministic = false;
    sh->n_last = n[todo[j]];
    sh->route_token = 0;
    sh->route_fused = true;
    if ((rc = ensure_workspace(rt, sh->n_last, true, s))) return rc;
    Workspace& ws = rt->ws;
    if (ws.pos_cap < sh->n_last) {
      HIP_TRY(hipStreamSynchronize(s));
      ws.pos_cap = 0;
      if ((rc = regrow(&ws.pos_ent, (size_t)std::max<long long>(sh->n_last, ws.cap_n)))) return rc;
      ws.pos_cap = std::max<long long>(sh->n_last, ws.cap_n);
    }
  }
  BatchStage& st = g_stage[device][1];
  StageSlot* sl = nullptr;
  if ((rc = stage_acquire(st, (size_t)m * sizeof(MultiDesc), &sl))) return rc;
  StageRelease rel{st, sl, s};
  MultiDesc* hd = reinterpret_cast<MultiDesc*>(sl->host);
  WsDev wmax{};
  for (int j = 0; j < m; ++j) {
    kv_shard* sh = shards[todo[j]];
    kv_table* rt = sh->route;
    MultiDesc& d = hd[j];
    std::memset(&d, 0, sizeof d);
    d.w = ws_view(rt, sh->n_last);
    rt->fused_index = true;
    choose_partitions(rt, d.w, sh->n_last);
    d.w.zero_counts = sh->ucnt;
    d.w.pos_ent = rt->ws.pos_ent;
    PartArgs& pa = d.a;
    pa.tv = dev_view(rt); pa.ts0 = pa.tv; pa.ts1 = pa.tv;
    pa.out_keys = sh->uniq; pa.out_counts = sh->ucnt;
    pa.sparse_unique = 1;
    pa.det = 0;
    pa.n = sh->n_last;
    pa.day_lk = pa.day;
    pa.route_world = sh->world; pa.route_rule = sh->rule; pa.route_C = sh->C;
    pa.route_seg = sh->send_pairs; pa.route_slot_of = sh->slot_of; pa.route_overflow = sh->overflow; pa.route_gcount = sh->gcount;
    pa.route_need = sh->need; pa.route_uhint = sh->overflow + 1;
    d.ids = ids[todo[j]];
    d.n = sh->n_last;
    wmax.ntiles = std::max(wmax.ntiles, d.w.ntiles);
    wmax.P = std::max(wmax.P, d.w.P);
  }
  HIP_TRY(hipMemcpyAsync(sl->dev, sl->host, (size_t)m * sizeof(MultiDesc), hipMemcpyHostToDevice, s));
  rel.launched = true;
  const MultiDesc* md = reinterpret_cast<const MultiDesc*>(sl->dev);
  k_ltile_multi_notable<<<dim3(wmax.ntiles, (unsigned)m), TBT, ltile_smem_bytes(), s>>>(md);
  // (PA_UNIQUE never reaches the code that depends on the row geometry: one variant serves every dim)
  PartArgs p0 = hd[0].a;
  p0.tv.dim = 4;
  if ((rc = kvp_launch_papply_ud(&wmax, &p0, PA_UNIQUE, (void*)s, md, m))) return fail(rc, "route: no kernel");
  for (int j = 0; j < m; ++j) {
    kv_shard* sh = shards[todo[j]];
    kv_table* rt = sh->route;
    rt->batch_serial = ++g_serial;
    rt->batch_n = sh->n_last;
    sh->route_token = rt->batch_serial;
    sh->ordered = false;
  }
  HIP_TRY(hipGetLastError());
  return KV_OK;
}

// finish of the tables todo[0..m) — all of one row geometry (row_lanes(dim)): 1 launch
static int multi_finish_impl(const kv_shard_t* shards, const int* todo, int m, float* const* outs, hipStream_t s) {
  int rc;
  const int device = shards[todo[0]]->table->device;
  std::vector<kv_table*> rts;
  for (int j = 0; j < m; ++j) rts.push_back(shards[todo[j]]->route);
  MultiLock lock(rts);
  BatchStage& st = g_stage[device][1];
  StageSlot* sl = nullptr;
  if ((rc = stage_acquire(st, (size_t)m * sizeof(FinishDesc), &sl))) return rc;
  StageRelease rel{st, sl, s};
  FinishDesc* hd = reinterpret_cast<FinishDesc*>(sl->host);
  long long nmax = 0;
  for (int j = 0; j < m; ++j) {
    kv_shard* sh = shards[todo[j]];
    kv_table* rt = sh->route;
    if (rt->batch_serial != sh->route_token || sh->route_token == 0) return fail(KV_FAILED_PRECONDITION, "kv_multi_shard_lookup: finish without route");
    if (!outs[todo[j]]) return fail(KV_INVALID_ARGUMENT, "kv_multi_shard_lookup: output pointer is null");
    if ((rc = hand_over(rt, s))) return rc;
    FinishDesc& d = hd[j];
    d.pos_ent = rt->ws.pos_ent; d.ent_u = rt->ws.ent_b; d.slot_of = sh->slot_of; d.rows = sh->recv_rows; d.out = outs[todo[j]];
    d.n = sh->n_last; d.rows_self = sh->send_rows;
    d.self_lo = (unsigned)sh->rank * (sh->C + 1); d.self_len = sh->self_in_place ? sh->C + 1 : 0u;
    d.dim = rt->dim; d.pad = 0;
    nmax = std::max<long long>(nmax, sh->n_last);
  }
  HIP_TRY(hipMemcpyAsync(sl->dev, sl->host, (size_t)m * sizeof(FinishDesc), hipMemcpyHostToDevice, s));
  rel.launched = true;
  const FinishDesc* md = reinterpret_cast<const FinishDesc*>(sl->dev);
  const dim3 grid((unsigned)nblocks(nmax, TB, 8192), (unsigned)m);
#define KV_SFM(VQ) k_shard_finish_multi<VQ><<<grid, TB, 0, s>>>(md)
  switch (row_lanes(shards[todo[0]]->table->dim)) {
    case 1: KV_SFM(1); break;   case 2: KV_SFM(2); break;   case 4: KV_SFM(4); break;   case 8: KV_SFM(8); break;
    case 16: KV_SFM(16); break; case 32: KV_SFM(32); break; default: KV_SFM(64); break;
  }
#undef KV_SFM
  HIP_TRY(hipGetLastError());
  return KV_OK;
}

// gradient pre-sum of the tables todo[0..m) — all of one dim: the tile sums, then k_papply_multi PA_DEDUP.  2 launches.
static int multi_presum_impl(const kv_shard_t* shards, const int* todo, int m, const float* const* grads, hipStream_t s) {
  int rc;
  const int device = shards[todo[0]]->table->device;
  std::vector<kv_table*> rts;
  for (int j = 0; j < m; ++j) rts.push_back(shards[todo[j]]->route);
  MultiLock lock(rts);
  for (int j = 0; j < m; ++j) {
    kv_shard* sh = shards[todo[j]];
    kv_table* rt = sh->route;
    if (rt->batch_serial != sh->route_token || sh->route_token == 0)
      return fail(KV_FAILED_PRECONDITION, "kv_multi_shard_apply: the batch's lookup must come first");
    if (!grads[todo[j]]) return fail(KV_INVALID_ARGUMENT, "kv_multi_shard_apply: grad pointer is null");
    if ((rc = hand_over(rt, s))) return rc;
    if ((rc = ensure_workspace(rt, sh->n_last, true, s))) return rc;
  }
  BatchStage& st = g_stage[device][1];
  StageSlot* sl = nullptr;
  if ((rc = stage_acquire(st, (size_t)m * sizeof(MultiDesc), &sl))) return rc;
  StageRelease rel{st, sl, s};
  MultiDesc* hd = reinterpret_cast<MultiDesc*>(sl->host);
  WsDev wmax{};
  long long nmax = 0;
  for (int j = 0; j < m; ++j) {
    kv_shard* sh = shards[todo[j]];
    kv_table* rt = sh->route;
    MultiDesc& d = hd[j];
    std::memset(&d, 0, sizeof d);
    d.w = ws_view(rt, sh->n_last);
    if (rt->index_P) { d.w.P = rt->index_P; d.w.pshift = 64 - ilog2(d.w.P); }
    PartArgs& pa = d.a;
    pa.tv = dev_view(rt); pa.ts0 = pa.tv; pa.ts1 = pa.tv;
    pa.grad = grads[todo[j]];
    pa.out_sum = sh->send_rows;
    pa.out_map = sh->slot_of;
    pa.det = 0;
    pa.n = sh->n_last;
    pa.epart = d.w.epart;
    pa.day_lk = pa.day;
    d.n = sh->n_last;
    wmax.ntiles = std::max(wmax.ntiles, d.w.ntiles);
    wmax.P = std::max(wmax.P, d.w.P);
    nmax = std::max<long long>(nmax, sh->n_last);
  }
  HIP_TRY(hipMemcpyAsync(sl->dev, sl->host, (size_t)m * sizeof(MultiDesc), hipMemcpyHostToDevice, s));
  rel.launched = true;
  const MultiDesc* md = reinterpret_cast<const MultiDesc*>(sl->dev);
  if ((rc = kvp_launch_tsum(&hd[0].a.tv, &wmax, nullptr, (void*)s, md, m))) return fail(rc, "tile sums: no kernel for dim %d", hd[0].a.tv.dim);
  if ((rc = kvp_launch_papply_ud(&wmax, &hd[0].a, PA_DEDUP, (void*)s, md, m))) return fail(rc, "gradient pre-sum: no kernel for dim %d", hd[0].a.tv.dim);
  HIP_TRY(hipGetLastError());
  return KV_OK;
}

int kv_multi_shard_lookup(const kv_shard_t* shards, int ntab, kv_comm_t comm, const void* const* ids, const int64_t* n,
                          float* const* outs, int join, kv_stream_t stream) {
  int rc;
  if ((rc = multi_shard_check(shards, ntab, comm, "kv_multi_shard_lookup"))) return rc;
  if (!ids || !n || !outs) return fail(KV_INVALID_ARGUMENT, "kv_multi_shard_lookup: null argument list");
  DeviceGuard dg(shards[0]->table->device);
  hipStream_t s = (hipStream_t)stream, w = comm->stream;
  {   // as in kv_shard_lookup: the lossless agreement's synchronisation is refused before the fork, not behind it
    bool lossless = false;
    for (int k = 0; k < ntab; ++k) lossless = lossless || shards[k]->lossless;
    if (lossless && (stream_is_capturing(s) || stream_is_capturing(w)))
      return fail(KV_FAILED_PRECONDITION, "kv_multi_shard_lookup under stream capture: the lossless mode (the default) synchronises once "
                                          "per lookup; capture sharded steps with kv_shard_set_lossless(shard, 0)");
  }
  for (int k = 0; k < ntab; ++k)
    if ((rc = shard_verify(shards[k], comm))) return rc;
  std::vector<unsigned> seen(ntab);
  for (int k = 0; k < ntab; ++k) seen[k] = shard_take_flag(shards[k]);
  if ((rc = shard_fork(shards[0], s, w))) return rc;
  int first = KV_OK;
  std::string first_msg;
  auto note = [&](int r) { if (r && !first) { first = r; first_msg = kv_last_error(); } return r; };
  std::vector<const void*> sp(ntab), sr(ntab);
  std::vector<void*> rp(ntab), rr(ntab);
  std::vector<int64_t> pb(ntab), rb(ntab);
  std::vector<char> routed(ntab, 1), stay(ntab, 0);
  for (int k = 0; k < ntab; ++k) stay[k] = (shards[k]->self_in_place = shard_can_stay(shards[k])) ? 1 : 0;
  auto buffers = [&](int k) {
    kv_shard* sh = shards[k];
    const int64_t mul = wired(comm) ? 1 : sh->world;
    pb[k] = (int64_t)(sh->C + 1) * 16 * mul;
    rb[k] = (int64_t)(sh->C + 1) * sh->table->dim * (int64_t)sizeof(float) * mul;
    sp[k] = sh->send_pairs; rp[k] = sh->recv_pairs; sr[k] = sh->send_rows; rr[k] = sh->recv_rows;
  };
  auto route = [&](int k) -> int {
    kv_shard* sh = shards[k];
    if (routed[k] && note(lookup_route_impl(sh, ids[k], n[k], w))) routed[k] = 0;
    if (!routed[k]) {   // as in kv_shard_lookup: void headers, the peers are not left waiting
      sh->n_last = 0; sh->route_token = 0;
      HIP_TRY(hipMemsetAsync(sh->counts, 0, (size_t)sh->world * 8, w));
      k_seg_headers<<<1, MAXW, 0, w>>>(sh->counts, sh->world, sh->C, sh->send_pairs, sh->need);
    }
    return KV_OK;
  };
  {
    // the tables whose route is the entry-list one go together: 2 launches for all of them
    std::vector<int> todo;
    std::vector<char> batched(ntab, 0);
    for (int k = 0; k < ntab; ++k)
        if (shard_batchable(shards[k], n[k])) todo.push_back(k);
    if (todo.size() >= 2) {
      for (int k : todo) batched[k] = 1;
      if (note(multi_route_impl(shards, todo.data(), (int)todo.size(), ids, n, w)))
        for (int k : todo) routed[k] = 0;   // (void headers below: the peers are not left waiting)
    }
    for (int k = 0; k < ntab; ++k) {
      buffers(k);
      if (batched[k] && routed[k]) continue;
      if ((rc = route(k))) return rc;
    }
  }
  {   // lossless tables: one agreement for all of them (see shard_agree_many), then the grown ones are routed again
    std::vector<char> grown(ntab, 0);
    bool any_lossless = false;
    for (int k = 0; k < ntab; ++k) any_lossless = any_lossless || shards[k]->lossless;
    if ((rc = shard_agree_many(shards, ntab, comm, w, grown.data()))) return rc;
    for (int k = 0; k < ntab; ++k)
      if (grown[k]) {
        stay[k] = (shards[k]->self_in_place = shard_can_stay(shards[k])) ? 1 : 0;
        buffers(k);
        if ((rc = route(k))) return rc;
      }
  }
  if ((rc = comm_exchange(comm, ntab, sp.data(), rp.data(), pb.data(), w, stay.data()))) return rc;
  {
    // the owners' lookups: the tables of one dim in one batched lookup over their receive buffers (tile pass of all of
    // them in one launch; the partition passes stay pending for the batched apply)
    std::vector<char> served(ntab, 0);
    std::vector<int> dims;
    auto can = [&](int k) {
      const kv_shard* sh = shards[k];
      return fused_ok(sh->table->dim) && sh->table->key_dtype == KV_DT_INT64 && (long long)sh->world * (sh->C + 1) <= (1ll << 21);
    };
    for (int k = 0; k < ntab; ++k)
        if (can(k)) dims.push_back(shards[k]->table->dim);
    std::sort(dims.begin(), dims.end());
    dims.erase(std::unique(dims.begin(), dims.end()), dims.end());
    for (int D : dims) {
      std::vector<int> grp;
      for (int k = 0; k < ntab; ++k)
        if (can(k) && shards[k]->table->dim == D) grp.push_back(k);
      if (grp.size() < 2) continue;
      const int m = (int)grp.size();
      std::vector<kv_handle_t> tb(m);
      std::vector<const void*> ip(m);
      std::vector<int64_t> nn(m);
      std::vector<float*> op(m);
      std::vector<kv_batch_token_t> tok(m, 0);
      std::vector<unsigned> caps(m);
      SelfScope self;
      for (int j = 0; j < m; ++j) {
        kv_shard* sh = shards[grp[j]];
        tb[j] = sh->table; ip[j] = sh->recv_pairs; nn[j] = (int64_t)sh->world * (sh->C + 1); op[j] = sh->send_rows; caps[j] = sh->C + 1;
        sh->serve_token = 0;
        self.add(sh->table, sh->self_in_place, (unsigned)sh->rank * (sh->C + 1), sh->C + 1, sh->send_pairs, nullptr);
      }
      if (note(multi_lookup_impl(m, tb.data(), ip.data(), nullptr, nn.data(), op.data(), tok.data(), w, 2, caps.data()))) {
        for (int k : grp) HIP_TRY(hipMemsetAsync(shards[k]->send_rows, 0, (size_t)rb[k] * (wired(comm) ? shards[k]->world : 1), w));
      } else {
        for (int j = 0; j < m; ++j) shards[grp[j]]->serve_token = tok[j];
      }
      for (int k : grp) served[k] = 1;
    }
    for (int k = 0; k < ntab; ++k)
      if (!served[k] && note(kv_shard_lookup_serve(shards[k], w)))
        HIP_TRY(hipMemsetAsync(shards[k]->send_rows, 0, (size_t)rb[k] * (wired(comm) ? shards[k]->world : 1), w));
  }
  if ((rc = comm_exchange(comm, ntab, sr.data(), rr.data(), rb.data(), w, stay.data()))) return rc;
  {
    // finish: the tables of one row geometry whose route is an entry-list index in one launch
    std::vector<char> done(ntab, 0);
    for (int q : {1, 2, 4, 8, 16, 32, 64}) {
        std::vector<int> grp;
        for (int k = 0; k < ntab; ++k)
          if (routed[k] && shards[k]->n_last > 0 && shards[k]->route_fused && row_lanes(shards[k]->table->dim) == q) grp.push_back(k);
        if (grp.size() < 2) continue;
        if (!note(multi_finish_impl(shards, grp.data(), (int)grp.size(), outs, w)))
          for (int k : grp) done[k] = 1;
      }
    for (int k = 0; k < ntab; ++k)
      if (routed[k] && !done[k]) note(kv_shard_lookup_finish(shards[k], outs[k], w));
  }
  if ((rc = shard_done(shards[0], s, w, join))) return rc;
  if (first) return fail(first, "%s (this rank's exchanges were queued all the same)", first_msg.c_str());
  for (int k = 0; k < ntab; ++k)
    if ((rc = shard_late_report(shards[k], seen[k]))) return rc;
  return KV_OK;
}

// slot0 / slot1: one handle per table (slot1 only for SparseGroupFtrl, else nullptr); hp as in kv_shard_apply_serve
int kv_multi_shard_apply(const kv_shard_t* shards, int ntab, kv_comm_t comm, int optimizer, const kv_handle_t* slot0,
                         const kv_handle_t* slot1, const float* const* grads, const float* hp, int join, kv_stream_t stream) {
  int rc;
  if ((rc = multi_shard_check(shards, ntab, comm, "kv_multi_shard_apply"))) return rc;
  if (!slot0 || !grads || !hp) return fail(KV_INVALID_ARGUMENT, "kv_multi_shard_apply: null argument list");
  DeviceGuard dg(shards[0]->table->device);
  hipStream_t s = (hipStream_t)stream, w = comm->stream;
  if ((rc = shard_fork(shards[0], s, w))) return rc;
  int first = KV_OK;
  std::string first_msg;
  auto note = [&](int r) { if (r && !first) { first = r; first_msg = kv_last_error(); } return r; };
  std::vector<const void*> sr(ntab);
  std::vector<void*> rr(ntab);
  std::vector<int64_t> rb(ntab);
  std::vector<char> stay(ntab, 0);
  std::vector<char> summed(ntab, 0);
  {
    // gradient pre-sum: the tables of one dim whose route is an entry-list index in two launches
    std::vector<int> dims;
    for (int k = 0; k < ntab; ++k)
      if (shards[k]->n_last > 0 && shards[k]->route_fused && !shards[k]->table->deterministic) dims.push_back(shards[k]->table->dim);
    std::sort(dims.begin(), dims.end());
    dims.erase(std::unique(dims.begin(), dims.end()), dims.end());
    for (int D : dims) {
      std::vector<int> grp;
      for (int k = 0; k < ntab; ++k)
        if (shards[k]->n_last > 0 && shards[k]->route_fused && !shards[k]->table->deterministic && shards[k]->table->dim == D) grp.push_back(k);
      if (grp.size() < 2) continue;
      if (!note(multi_presum_impl(shards, grp.data(), (int)grp.size(), grads, w)))
        for (int k : grp) summed[k] = 1;
      else
        for (int k : grp) summed[k] = 2;   // failed: zero gradient rows below
    }
  }
  for (int k = 0; k < ntab; ++k) {
    kv_shard* sh = shards[k];
    stay[k] = sh->self_in_place ? 1 : 0;
    rb[k] = (int64_t)(sh->C + 1) * sh->table->dim * (int64_t)sizeof(float) * (wired(comm) ? 1 : sh->world);
    sr[k] = sh->send_rows; rr[k] = sh->recv_rows;
    if (summed[k] == 1) continue;
    if (summed[k] == 2 || note(kv_shard_apply_route(sh, grads[k], w)))
      HIP_TRY(hipMemsetAsync(sh->send_rows, 0, (size_t)rb[k] * (wired(comm) ? sh->world : 1), w));
  }
  if ((rc = comm_exchange(comm, ntab, sr.data(), rr.data(), rb.data(), w, stay.data()))) return rc;
  {
    // the owners' applies: the tables of one dim whose serve lookups left their batch index in one batched apply
    std::vector<char> applied(ntab, 0);
    std::vector<int> dims;
    auto can = [&](int k) {
      const kv_shard* sh = shards[k];
      return fused_ok(sh->table->dim) && sh->table->key_dtype == KV_DT_INT64 &&
             (long long)sh->world * (sh->C + 1) <= (1ll << 21) && sh->serve_token != 0 && sh->serve_token == sh->table->batch_serial &&
             sh->table->fused_index && slot0[k] != nullptr && (optimizer != 3 || (slot1 && slot1[k]));
    };
    if (optimizer >= 0 && optimizer <= 3)
      for (int k = 0; k < ntab; ++k)
        if (can(k)) dims.push_back(shards[k]->table->dim);
    std::sort(dims.begin(), dims.end());
    dims.erase(std::unique(dims.begin(), dims.end()), dims.end());
    for (int D : dims) {
      std::vector<int> grp;
      for (int k = 0; k < ntab; ++k)
        if (can(k) && shards[k]->table->dim == D) grp.push_back(k);
      if (grp.size() < 2) continue;
      const int m = (int)grp.size();
      std::vector<kv_handle_t> vs(m), s0(m), s1(m);
      std::vector<const float*> gp(m);
      std::vector<const void*> ip(m);
      std::vector<int64_t> nn(m);
      std::vector<kv_batch_token_t> tok(m);
      SelfScope self;
      for (int j = 0; j < m; ++j) {
        kv_shard* sh = shards[grp[j]];
        vs[j] = sh->table; s0[j] = slot0[grp[j]]; s1[j] = slot1 ? slot1[grp[j]] : nullptr;
        gp[j] = sh->recv_rows; ip[j] = sh->recv_pairs; nn[j] = (int64_t)sh->world * (sh->C + 1); tok[j] = sh->serve_token;
        self.add(sh->table, sh->self_in_place, (unsigned)sh->rank * (sh->C + 1), sh->C + 1, sh->send_pairs, sh->send_rows);
      }
      tl_require_reuse = true;   // (the ids are (id, count) records: an apply that rebuilt its index from them would read them as plain ids)
      int r;
      switch (optimizer) {
        case 0: case 1:
          r = kv_multi_apply_group_adam_tok(m, vs.data(), s0.data(), gp.data(), ip.data(), nn.data(), hp[0], hp[1], hp[2], hp[3], hp[4], hp[5],
                                            hp[6], hp[7], hp[8], optimizer == 0 ? 4 : 3, tok.data(), w);
          break;
        case 2:
          r = kv_multi_apply_adagrad_tok(m, vs.data(), s0.data(), hp[0], gp.data(), ip.data(), nn.data(), hp[1] != 0.f, tok.data(), w);
          break;
        default:
          r = kv_multi_apply_sparse_group_ftrl_tok(m, vs.data(), s0.data(), s1.data(), gp.data(), ip.data(), nn.data(), hp[0], hp[1], hp[2],
                                                   hp[3], hp[4], hp[5], tok.data(), w);
          break;
      }
      tl_require_reuse = false;
      note(r);
      for (int k : grp) applied[k] = 1;
    }
    for (int k = 0; k < ntab; ++k)
      if (!applied[k]) note(kv_shard_apply_serve(shards[k], optimizer, slot0[k], slot1 ? slot1[k] : nullptr, hp, w));
  }
  if ((rc = shard_done(shards[0], s, w, join))) return rc;
  if (first) return fail(first, "%s (this rank's exchange was queued all the same)", first_msg.c_str());
  return KV_OK;
}

// The exchange between shards that live in ONE process (all on one device): segment r of shard p's send buffer to
// segment p of shard r's receive buffer.  what: 0 = the (id, count) records, 1 = the rows.  This is how the
// phases are exercised with several ranks on a single GPU (RCCL refuses two ranks on one device).
int kv_shard_exchange_local(const kv_shard_t* shards, int world, int what, kv_stream_t stream) {
  if (!shards || world < 1) return fail(KV_INVALID_ARGUMENT, "kv_shard_exchange_local: bad arguments");
  for (int p = 0; p < world; ++p)
    if (!shards[p] || shards[p]->world != world || shards[p]->C != shards[0]->C || shards[p]->table->dim != shards[0]->table->dim)
      return fail(KV_INVALID_ARGUMENT, "kv_shard_exchange_local: shards differ in world / capacity / dim");
  DeviceGuard dg(shards[0]->table->device);
  const size_t rec = shards[0]->C + 1;
  const size_t bytes = rec * (what == 0 ? 16 : (size_t)shards[0]->table->dim * sizeof(float));
  for (int p = 0; p < world; ++p)
    for (int r = 0; r < world; ++r) {
      const char* src = what == 0 ? (const char*)shards[p]->send_pairs : (const char*)shards[p]->send_rows;
      char* dst = what == 0 ? (char*)shards[r]->recv_pairs : (char*)shards[r]->recv_rows;
      HIP_TRY(hipMemcpyAsync(dst + (size_t)p * bytes, src + (size_t)r * bytes, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    }
  return KV_OK;
}

}  // extern "C"
