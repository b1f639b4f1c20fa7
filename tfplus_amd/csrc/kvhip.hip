// kvhip.hip — MI355X (gfx950) KvVariable: HBM hash table + row slab, lookup and fused
// sparse optimizer kernels, and the C ABI of include/kvhip.h.
//
// Layout in HBM (per table):
//   index    Entry[cap+1]   16 B {int64 key, u32 row, u32 pad}, open addressing, linear
//                           probing, cap = 2^k >= 2 * rows (load <= 0.5).  Entry[cap] is the
//                           home of the one key that equals the EMPTY sentinel.
//   chunks   row slab in chunks of 2^cb rows: rows[r][dim] fp32, freq[r] u32
//            ((day << 16) | saturating u16 frequency), flags[r] u8 (bit0 blacklist,
//            bit1 under_threshold), keys[r] int64.  Row ids are dense (bump allocated),
//            row 0 is a permanent all-zero row (misses / nothing).
//   scratch  per-batch dedup hash skeys[S+1], smeta[S+1] {count, unique idx}, srow[S+1];
//            self-cleaning (the last kernel of every op restores EMPTY / 0).
//
// Kernel pipeline (DESIGN.md has the byte accounting):
//   lookup : k_dedup_find  (LDS-staged tile dedup -> batch scratch hash -> one owner lane per
//                           unique key probes / inserts the table)
//            k_gather      (coalesced 16 B/lane row gather + per-unique frequency/flag finalize)
//   apply  : k_dedup_find  (same, optimizer-side FindOrInsertUnsafe semantics)
//            k_accumulate  (LDS pre-reduction of repeated ids, fp32 atomics for the spill)
//            k_apply<OPT>  (slot-table probes + fused row update, LPR lanes per row)
//
// Reference semantics restated per function with file:line (relative to the tfplus tree).

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/kvhip.h"

namespace {

// ------------------------------------------------------------------------------------------
// constants
// ------------------------------------------------------------------------------------------
constexpr long long EMPTY_KEY = (long long)0x8000000000000000ULL;
constexpr unsigned FLAG_BLACK = 1u;   // EmbeddingValue::in_black_   (embedding_value.h:225)
constexpr unsigned FLAG_UNDER = 2u;   // EmbeddingValue::under_threshold_
constexpr float CUTOFF = 1.0e-20f;    // DEFAULT_CUTOFF_VALUE (kv_variable_interface.h:55)
constexpr unsigned ROW_FILTERED = 0x80000000u;  // urow bit: var frequency < enter_threshold
constexpr unsigned ROW_MASK = 0x7FFFFFFFu;

constexpr int TB = 256;          // threads per block everywhere
constexpr int IPT = 4;           // ids per thread in the tile kernels
constexpr int TILE = TB * IPT;   // ids per tile (1024)
constexpr int LS = 2 * TILE;     // LDS hash slots per tile (load <= 0.5)
constexpr int MAX_CHUNKS = 1024;

enum Mode { MODE_LOOKUP = 0, MODE_APPLY = 1, MODE_DEDUP = 2, MODE_SCATTER = 3 };
enum Opt { OPT_ADAM_V4 = 0, OPT_ADAM_V3 = 1, OPT_ADAGRAD = 2, OPT_FTRL = 3 };

struct __attribute__((aligned(16))) Entry {
  long long key;
  unsigned row;
  unsigned pad;
};

struct Chunk {
  float* rows;
  unsigned* freq;
  unsigned char* flags;
  long long* keys;
};

// device view of one table; passed to kernels by value
struct TableDev {
  Entry* entries;
  unsigned long long mask;  // cap - 1; entries[cap] = sentinel-key home
  Chunk* chunks;
  int chunk_bits;
  unsigned* counters;  // [0] next_row  [1] error flag (row overflow)
  unsigned max_rows;
  const float* init_table;
  unsigned init_rows;
  int dim;
  unsigned enter_threshold;
  unsigned long long seed;
};

// device view of the per-batch workspace
struct WsDev {
  long long* skeys;            // [S+1]
  uint2* smeta;                // [S+1] x = summed count, y = unique index
  unsigned* srow;              // [S+1] var row id of the key (lookup gather)
  unsigned long long smask;    // S - 1
  int sshift;                  // 64 - log2(S)
  unsigned* sslot_of_id;       // [n]
  long long* ukey;             // [n] per unique
  unsigned* urow;              // [n] var row id | ROW_FILTERED
  unsigned* usslot;            // [n]
  unsigned* ufirst;            // [n] one input position of the key (singletons: THE position)
  unsigned* ctr;               // this op's counters: [0] = U (dense unique count).  Bumped ONCE per
                               // tile: a returning atomic on one word saturates at ~88 ops/us
  unsigned* ctr_next;          // next op's counters (zeroed by this op's last kernel)
  float* gacc;                 // [n, dim] accumulators of repeated ids (kept all-zero between ops)
  unsigned long long* dbg;     // diagnostic build only (-DKV_STAMPS): per-block phase stamps
};

// In-kernel phase stamps for the diagnostic build (never in the product .so): thread 0 of each
// block stores s_memtime at phase boundaries into a buffer nothing else reads.
#ifdef KV_STAMPS
#define KV_STAMP(slot) do { if (threadIdx.x == 0) w.dbg[(size_t)blockIdx.x * 16 + (slot)] = clock64(); } while (0)
#else
#define KV_STAMP(slot) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}
// same picker as oracle/kv_oracle.cc (splitmix64 finaliser) — see kv_set_seed
__device__ __forceinline__ unsigned long long pick64(unsigned long long x) {
  x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
  x ^= x >> 27; x *= 0x94d049bb133111ebULL;
  x ^= x >> 31;
  return x;
}

__device__ __forceinline__ float* row_ptr(const TableDev& t, unsigned r) {
  const Chunk& c = t.chunks[r >> t.chunk_bits];
  return c.rows + (size_t)(r & ((1u << t.chunk_bits) - 1)) * t.dim;
}
__device__ __forceinline__ unsigned* freq_ptr(const TableDev& t, unsigned r) {
  return t.chunks[r >> t.chunk_bits].freq + (r & ((1u << t.chunk_bits) - 1));
}
__device__ __forceinline__ unsigned char* flags_ptr(const TableDev& t, unsigned r) {
  return t.chunks[r >> t.chunk_bits].flags + (r & ((1u << t.chunk_bits) - 1));
}
__device__ __forceinline__ long long* key_ptr(const TableDev& t, unsigned r) {
  return t.chunks[r >> t.chunk_bits].keys + (r & ((1u << t.chunk_bits) - 1));
}

__device__ __forceinline__ Entry load_entry(const Entry* e) {
  const uint4 v = *reinterpret_cast<const uint4*>(e);
  Entry r;
  r.key = (long long)(((unsigned long long)v.y << 32) | v.x);
  r.row = v.z;
  r.pad = v.w;
  return r;
}

// read-only probe; 0 = absent (row 0 is the zero row)
__device__ __forceinline__ unsigned table_find(const TableDev& t, long long key) {
  if (key == EMPTY_KEY) {
    Entry e = load_entry(&t.entries[t.mask + 1]);
    return e.key == 0 ? e.row : 0u;
  }
  unsigned long long p = mix64((unsigned long long)key) & t.mask;
  for (;;) {
    Entry e = load_entry(&t.entries[p]);
    if (e.key == key) return e.row;
    if (e.key == EMPTY_KEY) return 0u;
    p = (p + 1) & t.mask;
  }
}

// Find or insert.  The caller is the ONLY lane of the launch that handles `key` (batch
// dedup guarantees it), so a freshly claimed entry is never read by anyone else before the
// kernel ends; other keys racing for the same empty entry are settled by the 64-bit CAS.
// Returns the row id; *inserted tells whether it was allocated now.  Returns 0 and raises
// counters[1] when the slab is full (the host pre-sizes, so this is a bug trap).
__device__ __forceinline__ unsigned table_find_or_insert(const TableDev& t, long long key,
                                                        bool* inserted) {
  *inserted = false;
  Entry* slot;
  long long stored;
  if (key == EMPTY_KEY) {
    slot = &t.entries[t.mask + 1];
    stored = 0;  // the sentinel's home holds 0 when occupied
    Entry e = load_entry(slot);
    if (e.key == stored) return e.row;
  } else {
    stored = key;
    unsigned long long p = mix64((unsigned long long)key) & t.mask;
    for (;;) {
      slot = &t.entries[p];
      Entry e = load_entry(slot);
      if (e.key == key) return e.row;
      if (e.key == EMPTY_KEY) {
        unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&slot->key),
                                           (unsigned long long)EMPTY_KEY, (unsigned long long)key);
        if (old == (unsigned long long)EMPTY_KEY) goto claimed;
        // another key took it between our load and the CAS: keep probing
      }
      p = (p + 1) & t.mask;
    }
  }
  {
    unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&slot->key),
                                       (unsigned long long)EMPTY_KEY, (unsigned long long)stored);
    if (old != (unsigned long long)EMPTY_KEY) return load_entry(slot).row;  // cannot happen (single owner)
  }
claimed:
  unsigned r = atomicAdd(&t.counters[0], 1u);
  if (r >= t.max_rows) {
    atomicExch(&t.counters[1], 1u);
    slot->row = 0;
    return 0u;
  }
  slot->row = r;
  *key_ptr(t, r) = key;
  *inserted = true;
  return r;
}

// kv_variable.h:889-898 GenerateRandomInitialValue: row = 0.5 * (T[r1] + T[r2]).  The
// reference draws r1, r2 from std::rand(); here they are a hash of (key, seed) so a run is
// reproducible.  Executed by `lanes` cooperating lanes (lane = 0..lanes-1).  Returns
// whether this lane saw any |x| >= CUTOFF.
__device__ __forceinline__ bool init_row_coop(const TableDev& t, long long key, float* dst,
                                              int lane, int lanes) {
  unsigned long long h = pick64((unsigned long long)key ^ (t.seed * 0x9E3779B97F4A7C15ULL));
  const float* a = t.init_table + (size_t)((unsigned)h % t.init_rows) * t.dim;
  const float* b = t.init_table + (size_t)((unsigned)(h >> 32) % t.init_rows) * t.dim;
  bool big = false;
  for (int e = lane; e < t.dim; e += lanes) {
    float v = (a[e] + b[e]) * 0.5f;
    dst[e] = v;
    big |= fabsf(v) >= CUTOFF;
  }
  return big;
}

template <typename IdT>
__device__ __forceinline__ long long load_id(const IdT* ids, size_t i) {
  return (long long)ids[i];
}

// ------------------------------------------------------------------------------------------
// K1: tile dedup in LDS -> batch scratch hash -> one owner per unique key touches the table
// ------------------------------------------------------------------------------------------
// Restates, per unique key:
//   MODE_LOOKUP  KvVariable::FindOrInsertLocally        kv_variable.h:287-380
//                TableManager::FindOrInsertWithDifferentFn table_manager.h:167-190
//                (frequency / under_threshold are finalised in k_gather with the batch count)
//   MODE_APPLY   KvVariable::FindOrInsertUnsafe(filter_out != nullptr)  kv_variable.h:382-408
//                + RemoveBlacklistUnsafe table_manager.h:359-372
//   MODE_SCATTER find-or-insert as ScatterUpdate's insert_func      kv_variable.h:698-715
//   MODE_DEDUP   tf.unique only (no table)
template <int MODE, typename IdT>
__global__ void __launch_bounds__(TB) k_dedup_find(TableDev t, WsDev w, const IdT* __restrict__ ids,
                                                   const int* __restrict__ counts, long long n,
                                                   unsigned day) {
  __shared__ long long lkeys[LS + 1];
  __shared__ unsigned lcnt[LS + 1];   // tile count, later re-used as the key's scratch slot
  __shared__ unsigned lfirst[LS + 1];
  __shared__ unsigned short lwork[TILE + 1];
  __shared__ unsigned short lown[TILE + 1];
  __shared__ unsigned lnew[TILE + 1];
  __shared__ unsigned lnwork, lnown, lnnew, lsent, lubase;

  const int tid = threadIdx.x;
  const long long base = (long long)blockIdx.x * TILE;
  KV_STAMP(0);

  for (int s = tid; s <= LS; s += TB) {
    lkeys[s] = EMPTY_KEY;
    lcnt[s] = 0;
  }
  if (tid == 0) { lnwork = 0; lnown = 0; lnnew = 0; lsent = 0; }
  __syncthreads();

  // ---- phase 1: LDS hash insert of this tile's ids --------------------------------------
  unsigned tslot[IPT];
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    const long long i = base + (long long)k * TB + tid;
    tslot[k] = 0xFFFFFFFFu;
    if (i < n) {
      const long long key = load_id(ids, (size_t)i);
      unsigned c = 1;
      if (MODE == MODE_LOOKUP && counts != nullptr) {
        // SaturateMaxFrequency(int32) -> uint16 (utility.h:57-59)
        int ci = counts[i];
        c = (unsigned)(unsigned short)(ci < 65535 ? ci : 65535);
      }
      unsigned h;
      if (key == EMPTY_KEY) {
        h = LS;
        if (atomicCAS(&lsent, 0u, 1u) == 0u) lfirst[LS] = (unsigned)i;
      } else {
        h = (unsigned)(mix64((unsigned long long)key) >> 40) & (LS - 1);
        for (;;) {
          unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&lkeys[h]),
                                             (unsigned long long)EMPTY_KEY, (unsigned long long)key);
          if (old == (unsigned long long)EMPTY_KEY) { lfirst[h] = (unsigned)i; break; }
          if (old == (unsigned long long)key) break;
          h = (h + 1) & (LS - 1);
        }
      }
      atomicAdd(&lcnt[h], c);
      tslot[k] = h;
    }
  }
  __syncthreads();
  KV_STAMP(1);

  // ---- phase 2a: compact the occupied LDS slots into a work list -------------------------
  for (int s = tid; s < LS; s += TB) {
    if (lkeys[s] != EMPTY_KEY) lwork[atomicAdd(&lnwork, 1u)] = (unsigned short)s;
  }
  if (tid == 0 && lsent) lwork[atomicAdd(&lnwork, 1u)] = (unsigned short)LS;
  __syncthreads();
  KV_STAMP(2);

  // ---- phase 2b: one lane per tile-unique key: batch scratch insert; the batch-wide first
  //      inserter owns the key.  Load before CAS: for a heavy hitter every tile but the first
  //      finds the key with a plain (cacheable) load instead of a serialised returning atomic;
  //      a stale EMPTY from another XCD's L2 only costs the CAS it would have done anyway.
  const unsigned nwork = lnwork;
  for (unsigned wi = tid; wi < nwork; wi += TB) {
    const unsigned s = lwork[wi];
    const long long key = (s == LS) ? EMPTY_KEY : lkeys[s];
    unsigned cnt = lcnt[s];
    if (cnt > 65535u) cnt = 65535u;  // saturating add is order independent: clamp early
    unsigned long long g;
    bool owner = false;
    if (s == LS) {
      g = w.smask + 1;
      if (w.skeys[g] == EMPTY_KEY)
        owner = atomicCAS(reinterpret_cast<unsigned long long*>(&w.skeys[g]),
                          (unsigned long long)EMPTY_KEY, 0ULL) == (unsigned long long)EMPTY_KEY;
    } else {
      g = (mix64((unsigned long long)key) * 0x9E3779B97F4A7C15ULL) >> w.sshift;
      for (;;) {
        long long cur = w.skeys[g];
        if (cur == EMPTY_KEY) {
          cur = (long long)atomicCAS(reinterpret_cast<unsigned long long*>(&w.skeys[g]),
                                     (unsigned long long)EMPTY_KEY, (unsigned long long)key);
          if (cur == EMPTY_KEY) { owner = true; break; }
        }
        if (cur == key) break;
        g = (g + 1) & w.smask;
      }
    }
    atomicAdd(&w.smeta[g].x, cnt);
    lcnt[s] = (unsigned)g;  // phase 4 reads it back as the scratch slot of the key
    if (owner) lown[atomicAdd(&lnown, 1u)] = (unsigned short)s;
  }
  __syncthreads();
  KV_STAMP(3);

  // ---- phase 2c: owners do the table work; dense unique index = tile base + j ---------------
  const unsigned nown = lnown;
  if (tid == 0) lubase = nown ? atomicAdd(&w.ctr[0], nown) : 0u;
  __syncthreads();
  KV_STAMP(4);
  const unsigned ubase = lubase;
  for (unsigned j = tid; j < nown; j += TB) {
    const unsigned s = lown[j];
    const long long key = (s == LS) ? EMPTY_KEY : lkeys[s];
    const unsigned g = lcnt[s];
    const unsigned u = ubase + j;
    w.smeta[g].y = u;
    w.ukey[u] = key;
    w.usslot[u] = g;
    w.ufirst[u] = lfirst[s];
    if (MODE == MODE_DEDUP) continue;
    bool inserted;
    unsigned r = table_find_or_insert(t, key, &inserted);
    unsigned tag = r;
    if (inserted) {
      lnew[atomicAdd(&lnnew, 1u)] = r;
      // lookup: count is added in k_gather; optimizer-side insert keeps EmbeddingValue's
      // constructor value freq_val = 1 with day 0 (table_manager.h:94, kv_variable.h:384-399)
      *freq_ptr(t, r) = (MODE == MODE_LOOKUP) ? 0u : 1u;
      *flags_ptr(t, r) = 0;
    } else if (MODE == MODE_APPLY && r != 0) {
      const unsigned f = *freq_ptr(t, r);
      const bool filtered = (f & 0xFFFFu) < t.enter_threshold;  // HasLowFrequency kv_variable.h:910
      if (filtered) {
        tag |= ROW_FILTERED;
      } else {
        unsigned char* fl = flags_ptr(t, r);
        // RemoveBlacklistUnsafe: fresh zero row (ours is already zero), under_threshold = true
        if (*fl & FLAG_BLACK) *fl = FLAG_UNDER;
      }
    }
    w.urow[u] = tag;
    w.srow[g] = r;
  }
  __syncthreads();
  KV_STAMP(5);

  // ---- phase 3: cooperative init of the rows this tile inserted ---------------------------
  if (MODE != MODE_DEDUP) {
    const unsigned nnew = lnnew;
    for (unsigned j = tid >> 3; j < nnew; j += TB / 8) {
      const unsigned r = lnew[j];
      const long long key = *key_ptr(t, r);
      bool big = init_row_coop(t, key, row_ptr(t, r), tid & 7, 8);
      if (MODE != MODE_LOOKUP) {
        // UpdateUnderThreshold (kv_variable.h:837-861); lookup recomputes it in k_gather
        unsigned long long m = __ballot(big);
        const int lane = tid & 63;
        const bool any = ((m >> (lane & ~7)) & 0xFFull) != 0;
        if ((tid & 7) == 0) *flags_ptr(t, r) = any ? 0 : FLAG_UNDER;
      }
    }
  }

  // ---- phase 4: every input position learns its key's scratch slot ------------------------
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    const long long i = base + (long long)k * TB + tid;
    if (i < n) w.sslot_of_id[i] = lcnt[tslot[k]];
  }
  KV_STAMP(6);
  if (threadIdx.x == 0) { KV_STAMP(7); }
#ifdef KV_STAMPS
  if (threadIdx.x == 0) { w.dbg[(size_t)blockIdx.x * 16 + 8] = lnwork; w.dbg[(size_t)blockIdx.x * 16 + 9] = lnown; }
#endif
}

// ------------------------------------------------------------------------------------------
// K2: gather + per-unique finalize (lookup)
// ------------------------------------------------------------------------------------------
// finalize restates find_func / insert_func of kv_variable.h:320-363:
//   freq.lo16 = sat_add(lo16, batch count), freq.hi16 = today, UpdateUnderThreshold.
__device__ __forceinline__ void finalize_unique(const TableDev& t, const WsDev& w, unsigned u,
                                                int lane8, unsigned day) {
  const unsigned r = w.urow[u] & ROW_MASK;
  const unsigned g = w.usslot[u];
  const float* row = row_ptr(t, r);
  bool big = false;
  for (int e = lane8; e < t.dim; e += 8) big |= fabsf(row[e]) >= CUTOFF;
  unsigned long long m = __ballot(big);
  const int lane = threadIdx.x & 63;
  const bool any = ((m >> (lane & ~7)) & 0xFFull) != 0;
  if (lane8 == 0 && r != 0) {
    const unsigned cnt = w.smeta[g].x;
    unsigned* fp = freq_ptr(t, r);
    unsigned lo = (*fp & 0xFFFFu) + (cnt > 65535u ? 65535u : cnt);
    if (lo > 65535u) lo = 65535u;
    *fp = (day << 16) | lo;
    unsigned char* fl = flags_ptr(t, r);
    const unsigned black = *fl & FLAG_BLACK;
    *fl = (unsigned char)(black ? (FLAG_BLACK | FLAG_UNDER) : (any ? 0u : FLAG_UNDER));
  }
  if (lane8 == 0) {  // self-clean the scratch entry
    w.smeta[g] = make_uint2(0u, 0u);
    w.skeys[g] = EMPTY_KEY;
  }
}

// VQ = float4 vectors per row (dim / 4) when > 0 (power of two); VQ = 0 -> generic dim
template <int VQ>
__global__ void __launch_bounds__(TB) k_gather(TableDev t, WsDev w, float* __restrict__ out,
                                               long long n, unsigned day, int gather_blocks) {
  // every block first finalizes a slice of the unique keys (8 lanes each), then gathers
  {
    const unsigned U = w.ctr[0];
    const int lane8 = threadIdx.x & 7;
    for (unsigned u0 = blockIdx.x * (TB / 8); u0 < U; u0 += gridDim.x * (TB / 8)) {
      const unsigned u = u0 + (threadIdx.x >> 3);
      if (u < U) finalize_unique(t, w, u, lane8, day);
    }
    if (blockIdx.x == 0 && threadIdx.x < 8) w.ctr_next[threadIdx.x] = 0;
  }
  const long long gb = (long long)blockIdx.x;
  if constexpr (VQ > 0) {
    constexpr int RPB = TB / VQ;  // rows per block per step
    const int v = threadIdx.x % VQ;
    const long long r0 = gb * RPB + threadIdx.x / VQ;
    const long long stride = (long long)gather_blocks * RPB;
    constexpr int UNR = 4;
    for (long long i = r0; i < n; i += stride * UNR) {
      unsigned sl[UNR], rr[UNR];
      float4 val[UNR];
#pragma unroll
      for (int k = 0; k < UNR; ++k) {
        const long long ii = i + k * stride;
        sl[k] = ii < n ? w.sslot_of_id[ii] : 0u;
      }
#pragma unroll
      for (int k = 0; k < UNR; ++k) {
        const long long ii = i + k * stride;
        rr[k] = ii < n ? w.srow[sl[k]] : 0u;
      }
#pragma unroll
      for (int k = 0; k < UNR; ++k)
        val[k] = reinterpret_cast<const float4*>(row_ptr(t, rr[k]))[v];
#pragma unroll
      for (int k = 0; k < UNR; ++k) {
        const long long ii = i + k * stride;
        if (ii < n) reinterpret_cast<float4*>(out + (size_t)ii * (VQ * 4))[v] = val[k];
      }
    }
  } else {
    const int D = t.dim;
    const long long total = n * D;
    for (long long x = gb * TB + threadIdx.x; x < total; x += (long long)gather_blocks * TB) {
      const long long i = x / D;
      const int e = (int)(x - i * D);
      out[x] = row_ptr(t, w.srow[w.sslot_of_id[i]])[e];
    }
  }
}

// KvVariableGatherOrZeros: read-only, no dedup needed (no writes, repeated keys hit cache).
// FindOrZeros kv_variable.h:239-254 / BatchGetWithFn table_manager.h:112-154.
template <typename IdT>
__global__ void __launch_bounds__(TB) k_gather_or_zeros(TableDev t, const IdT* __restrict__ ids,
                                                        float* __restrict__ out, long long n) {
  const int D = t.dim;
  const int lane8 = threadIdx.x & 7;
  for (long long i = (long long)blockIdx.x * (TB / 8) + (threadIdx.x >> 3); i < n;
       i += (long long)gridDim.x * (TB / 8)) {
    const unsigned r = table_find(t, load_id(ids, (size_t)i));
    const float* row = row_ptr(t, r);  // blacklisted rows are stored as zeros; row 0 is zeros
    float* o = out + (size_t)i * D;
    if ((D & 3) == 0) {
      for (int q = lane8; q < (D >> 2); q += 8)
        reinterpret_cast<float4*>(o)[q] = reinterpret_cast<const float4*>(row)[q];
    } else {
      for (int e = lane8; e < D; e += 8) o[e] = row[e];
    }
  }
}

// ------------------------------------------------------------------------------------------
// B2: sum the gradient rows of repeated ids (tf.unsorted_segment_sum of TF-core's
// _deduplicate_indexed_slices) into gacc[unique idx].  Keys that occur once in the batch
// are skipped here: k_apply reads their gradient row in place.
// ------------------------------------------------------------------------------------------
// Structure (per tile of TILE ids): counting sort of the tile's repeated-id rows by key (integer
// LDS atomics only) -> each 8-lane group folds a chunk of ACC_CHUNK consecutive sorted rows in
// registers (independent 16-byte loads, one flush per key change) -> flushes go to gacc with
// fp32 global atomics, except keys that are hot inside the tile (> HOT_MIN rows), whose chunk
// partials meet in an LDS accumulator first so the tile issues one global row-add per key.
// (LDS float atomics per ROW were the bottleneck of the first version: SQ_LDS_IDX_ACTIVE 90M.)
constexpr int ACC_CHUNK = 16;
constexpr int HOT_MIN = 2 * ACC_CHUNK;
constexpr int HOT_ROWS = TILE / HOT_MIN;  // at most this many keys can exceed HOT_MIN rows per tile

// VPL = float4 vectors per lane per row (8 lanes per row): dim <= 32 * VPL, dim % 4 == 0.
// VPL = 0: any dim, scalar lanes, straight global atomics (small / odd dims; not a hot path).
template <int VPL>
__global__ void __launch_bounds__(TB) k_accumulate(WsDev w, const float* __restrict__ grad,
                                                   long long n, int D) {
  const int tid = threadIdx.x;
  const long long base = (long long)blockIdx.x * TILE;
  const int lane8 = tid & 7;
  if constexpr (VPL == 0) {
    for (int j = tid >> 3; j < TILE; j += TB / 8) {
      const long long i = base + j;
      if (i >= n) break;
      const uint2 m = w.smeta[w.sslot_of_id[i]];
      if (m.x < 2u) continue;
      float* dst = w.gacc + (size_t)m.y * D;
      const float* g = grad + (size_t)i * D;
      for (int e = lane8; e < D; e += 8) atomicAdd(&dst[e], g[e]);
    }
  } else {
    __shared__ unsigned lkey[LS];             // unique idx + 1 (0 = empty)
    __shared__ unsigned lcnt[LS];             // rows of the key in this tile, then its offset
    __shared__ unsigned char lhot[LS];        // LDS accumulator of the key (0xFF = none)
    __shared__ unsigned short perm[TILE];     // tile rows grouped by key
    __shared__ unsigned pkey[TILE];           // unique idx of each sorted entry
    __shared__ unsigned char phot[TILE];
    __shared__ unsigned hot_u[HOT_ROWS];
    __shared__ unsigned wtot[TB / 64];
    __shared__ unsigned lnhot, lM;
    extern __shared__ float hacc[];           // [HOT_ROWS][D + 1]
    const int DP = D + 1;
    const int NV = D >> 2;                    // float4 vectors per row

    for (int s = tid; s < LS; s += TB) { lkey[s] = 0; lcnt[s] = 0; lhot[s] = 0xFF; }
    if (tid == 0) lnhot = 0;
    __syncthreads();

    // phase 1: group this tile's repeated-id rows by key
    unsigned myh[IPT], myrank[IPT];
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
      const long long i = base + (long long)k * TB + tid;
      myh[k] = 0xFFFFFFFFu;
      myrank[k] = 0;
      if (i < n) {
        const uint2 m = w.smeta[w.sslot_of_id[i]];
        if (m.x >= 2u) {
          unsigned h = (m.y * 0x9E3779B1u) >> 21;  // 11 bits
          for (;;) {
            const unsigned old = atomicCAS(&lkey[h], 0u, m.y + 1u);
            if (old == 0u || old == m.y + 1u) break;
            h = (h + 1) & (LS - 1);
          }
          myrank[k] = atomicAdd(&lcnt[h], 1u);
          myh[k] = h;
        }
      }
    }
    __syncthreads();

    // phase 2: exclusive scan of the per-key counts (8 consecutive slots per thread)
    {
      unsigned c[LS / TB];
      unsigned tsum = 0;
#pragma unroll
      for (int q = 0; q < LS / TB; ++q) { c[q] = lcnt[tid * (LS / TB) + q]; tsum += c[q]; }
      unsigned incl = tsum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = __shfl_up(incl, o);
        if ((tid & 63) >= o) incl += v;
      }
      if ((tid & 63) == 63) wtot[tid >> 6] = incl;
      __syncthreads();
      unsigned run = incl - tsum;
      for (int wv = 0; wv < (tid >> 6); ++wv) run += wtot[wv];
#pragma unroll
      for (int q = 0; q < LS / TB; ++q) {
        const int sl = tid * (LS / TB) + q;
        lcnt[sl] = run;
        run += c[q];
        if (c[q] > (unsigned)HOT_MIN) {
          const unsigned a = atomicAdd(&lnhot, 1u);  // < HOT_ROWS by construction
          lhot[sl] = (unsigned char)a;
          hot_u[a] = lkey[sl] - 1u;
        }
      }
      if (tid == TB - 1) lM = run;
    }
    __syncthreads();
    const unsigned nhot = lnhot;
    for (unsigned x = tid; x < nhot * (unsigned)DP; x += TB) hacc[x] = 0.f;

    // phase 3: scatter rows to their sorted position
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
      if (myh[k] != 0xFFFFFFFFu) {
        const unsigned pos = lcnt[myh[k]] + myrank[k];
        perm[pos] = (unsigned short)(k * TB + tid);
        pkey[pos] = lkey[myh[k]] - 1u;
        phot[pos] = lhot[myh[k]];
      }
    }
    __syncthreads();

    // phase 4: fold chunks of sorted rows in registers
    const unsigned M = lM;
    constexpr int RB = 8 / VPL;  // rows loaded together (8 float4 in flight per lane)
    for (unsigned e0 = (tid >> 3) * ACC_CHUNK; e0 < M; e0 += (TB / 8) * ACC_CHUNK) {
      const unsigned e1 = min(e0 + (unsigned)ACC_CHUNK, M);
      unsigned cur = 0xFFFFFFFFu;
      unsigned curhot = 0xFF;
      float4 acc[VPL];
      auto flush = [&]() {
        if (cur == 0xFFFFFFFFu) return;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
          const int q = lane8 + 8 * v;
          if (q < NV) {
            if (curhot != 0xFF) {
              float* dst = hacc + (size_t)curhot * DP + 4 * q;
              atomicAdd(&dst[0], acc[v].x); atomicAdd(&dst[1], acc[v].y);
              atomicAdd(&dst[2], acc[v].z); atomicAdd(&dst[3], acc[v].w);
            } else {
              float* dst = w.gacc + (size_t)cur * D + 4 * q;
              atomicAdd(&dst[0], acc[v].x); atomicAdd(&dst[1], acc[v].y);
              atomicAdd(&dst[2], acc[v].z); atomicAdd(&dst[3], acc[v].w);
            }
          }
        }
      };
      for (unsigned eb = e0; eb < e1; eb += RB) {
        float4 val[RB][VPL];
        unsigned ku[RB], kh[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          const unsigned e = eb + r;
          ku[r] = 0xFFFFFFFFu;
          kh[r] = 0xFF;
          if (e < e1) {
            ku[r] = pkey[e];
            kh[r] = phot[e];
            const float4* g4 = reinterpret_cast<const float4*>(grad + (size_t)(base + perm[e]) * D);
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
              const int q = lane8 + 8 * v;
              val[r][v] = q < NV ? g4[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
          }
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          if (ku[r] == 0xFFFFFFFFu) continue;
          if (ku[r] != cur) {
            flush();
            cur = ku[r];
            curhot = kh[r];
#pragma unroll
            for (int v = 0; v < VPL; ++v) acc[v] = val[r][v];
          } else {
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
              acc[v].x += val[r][v].x; acc[v].y += val[r][v].y;
              acc[v].z += val[r][v].z; acc[v].w += val[r][v].w;
            }
          }
        }
      }
      flush();
    }
    __syncthreads();

    // phase 5: one global row-add per tile-hot key
    for (unsigned a = tid >> 3; a < nhot; a += TB / 8) {
      float* dst = w.gacc + (size_t)hot_u[a] * D;
      const float* src = hacc + (size_t)a * DP;
      for (int e = lane8; e < D; e += 8) atomicAdd(&dst[e], src[e]);
    }
  }
}

void launch_accumulate(const WsDev& wd, const float* grad, long long n, int D, hipStream_t s) {
  const int grid = (int)((n + TILE - 1) / TILE);
  const size_t sh = (size_t)HOT_ROWS * (D + 1) * sizeof(float);
  if ((D & 3) == 0 && D <= 32) k_accumulate<1><<<grid, TB, sh, s>>>(wd, grad, n, D);
  else if ((D & 3) == 0 && D <= 64) k_accumulate<2><<<grid, TB, sh, s>>>(wd, grad, n, D);
  else if ((D & 3) == 0 && D <= 128) k_accumulate<4><<<grid, TB, sh, s>>>(wd, grad, n, D);
  else if ((D & 3) == 0 && D <= 256) k_accumulate<8><<<grid, TB, sh, s>>>(wd, grad, n, D);
  else k_accumulate<0><<<grid, TB, 0, s>>>(wd, grad, n, D);
}

// ------------------------------------------------------------------------------------------
// B3: fused optimizer row update, LPR lanes per unique key
// ------------------------------------------------------------------------------------------
struct OptArgs {
  float lr, b1p, b2p, b1, b2, eps, l1, l2, l21, l2s, lr_power;
  float alpha, l21_norm;  // host-precomputed in fp32 exactly as the reference does
  int update_slots;
};

// optimizer-side slot-table access: FindOrInsertUnsafe(filter_out == nullptr), kv_variable.h:382-416.
// Called by the group leader only.  New rows get freq word 1 (day 0); hits AddFrequency(1, today).
__device__ __forceinline__ unsigned slot_find_or_insert(const TableDev& t, long long key,
                                                       unsigned day, bool* inserted) {
  unsigned r = table_find_or_insert(t, key, inserted);
  if (r == 0) return 0;
  unsigned* fp = freq_ptr(t, r);
  if (*inserted) {
    *fp = 1u;
  } else {
    unsigned lo = (*fp & 0xFFFFu) + 1u;
    if (lo > 65535u) lo = 65535u;
    *fp = (day << 16) | lo;
  }
  return r;
}

// V-wide row access (V = 4: one 16-byte access per lane; V = 1: scalar)
template <int V>
__device__ __forceinline__ void ldv(const float* p, float (&o)[V]) {
  if (V == 4) {
    const float4 t4 = *reinterpret_cast<const float4*>(p);
    o[0] = t4.x; o[1 % V] = t4.y; o[2 % V] = t4.z; o[3 % V] = t4.w;
  } else {
    o[0] = p[0];
  }
}
template <int V>
__device__ __forceinline__ void stv(float* p, const float (&o)[V]) {
  if (V == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1 % V], o[2 % V], o[3 % V]);
  } else {
    p[0] = o[0];
  }
}
// slot row element block: existing row, or the init rule 0.5 * (T[r1] + T[r2]) for a new key
template <int V>
__device__ __forceinline__ void ldslot(const float* row, const float* ia, const float* ib, bool isnew,
                                       int e, float (&o)[V]) {
  if (isnew) {
    float a[V], b[V];
    ldv<V>(ia + e, a);
    ldv<V>(ib + e, b);
#pragma unroll
    for (int c = 0; c < V; ++c) o[c] = (a[c] + b[c]) * 0.5f;
  } else {
    ldv<V>(row + e, o);
  }
}

template <int W>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, W);
  return v;
}
template <int W>
__device__ __forceinline__ bool group_any(bool p) {
  const unsigned long long m = __ballot(p);
  if (W >= 64) return m != 0;
  const int lane = threadIdx.x & 63;
  const unsigned long long gm = (W >= 64) ? ~0ull : ((1ull << W) - 1ull);
  return ((m >> (lane & ~(W - 1))) & gm) != 0;
}

// V = elements per lane-vector (4 or 1), LPR = lanes per row (power of two <= 64),
// K = vectors per lane.  Element e of the row lives at lane (e / V) % LPR, step (e / V) / LPR.
template <int OPT, int V, int LPR, int K>
__global__ void __launch_bounds__(TB) k_apply(TableDev tv, TableDev ts0, TableDev ts1, WsDev w,
                                              const float* __restrict__ grad, OptArgs a,
                                              unsigned day) {
  const int D = tv.dim;
  const int lane = threadIdx.x % LPR;
  constexpr unsigned GPB = TB / LPR;       // keys per block per step
  const unsigned U = w.ctr[0];
  for (unsigned u0 = blockIdx.x * GPB; u0 < U; u0 += gridDim.x * GPB) {
    const unsigned u = u0 + threadIdx.x / LPR;
    const bool live = u < U;
    unsigned tag = live ? w.urow[u] : ROW_FILTERED;
    const unsigned g = live ? w.usslot[u] : 0u;
    const bool multi = live && w.smeta[g].x >= 2u;
    const long long key = live ? w.ukey[u] : 0;
    const bool skip = (tag & ROW_FILTERED) || (tag & ROW_MASK) == 0u;  // training_ops.cc:7150-7152
    const unsigned rv = tag & ROW_MASK;

    // gradient of this key: in place for singletons, the accumulator for repeated ids
    const float* gsrc = multi ? (w.gacc + (size_t)u * D)
                              : (grad + (size_t)(live ? w.ufirst[u] : 0u) * D);
    float gv[K][V];
    const float zeros[V] = {};
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int e0 = (lane + k * LPR) * V;
#pragma unroll
      for (int c = 0; c < V; ++c) gv[k][c] = 0.f;
      if (live && e0 < D) {
        ldv<V>(gsrc + e0, gv[k]);
        if (multi) stv<V>(w.gacc + (size_t)u * D + e0, zeros);  // keep gacc all-zero between ops
      }
    }
    if (live && lane == 0) {  // self-clean the scratch entry of this key
      w.smeta[g] = make_uint2(0u, 0u);
      w.skeys[g] = EMPTY_KEY;
    }

    // slot tables (leader probes, group shares the row id)
    unsigned r0 = 0, r1 = 0;
    bool new0 = false, new1 = false;
    if (!skip && lane == 0) {
      // FTRL probes linear (ts1) before accum (ts0): training_ops.cc:701-704
      if (OPT == OPT_FTRL) r1 = slot_find_or_insert(ts1, key, day, &new1);
      r0 = slot_find_or_insert(ts0, key, day, &new0);
    }
    r0 = __shfl(r0, 0, LPR);
    new0 = __shfl((int)new0, 0, LPR) != 0;
    if (OPT == OPT_FTRL) {
      r1 = __shfl(r1, 0, LPR);
      new1 = __shfl((int)new1, 0, LPR) != 0;
    }
    const bool act = !skip && r0 != 0 && (OPT != OPT_FTRL || r1 != 0);

    float* xrow = row_ptr(tv, act ? rv : 0u);
    float* s0row = row_ptr(ts0, act ? r0 : 0u);
    float* s1row = (OPT == OPT_FTRL) ? row_ptr(ts1, act ? r1 : 0u) : nullptr;

    // new slot rows are initialised in registers with the slot table's init rule
    const float *ia0 = nullptr, *ib0 = nullptr, *ia1 = nullptr, *ib1 = nullptr;
    if (act && new0) {
      unsigned long long h = pick64((unsigned long long)key ^ (ts0.seed * 0x9E3779B97F4A7C15ULL));
      ia0 = ts0.init_table + (size_t)((unsigned)h % ts0.init_rows) * ts0.dim;
      ib0 = ts0.init_table + (size_t)((unsigned)(h >> 32) % ts0.init_rows) * ts0.dim;
    }
    if (OPT == OPT_FTRL && act && new1) {
      unsigned long long h = pick64((unsigned long long)key ^ (ts1.seed * 0x9E3779B97F4A7C15ULL));
      ia1 = ts1.init_table + (size_t)((unsigned)h % ts1.init_rows) * ts1.dim;
      ib1 = ts1.init_table + (size_t)((unsigned)(h >> 32) % ts1.init_rows) * ts1.dim;
    }
    if (OPT == OPT_ADAM_V4 || OPT == OPT_ADAM_V3) {
      // training_ops.cc:7166-7195 (V4) / :5895-5925 (V3); slot row = [m | v | z]
      float m[K][V], nv[K][V], sq[K][V], z[K][V], uu[K][V];
      float part = 0.f;
      const float omb1 = 1.f - a.b1, omb2 = 1.f - a.b2;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int e0 = (lane + k * LPR) * V;
        float xo[V], mo[V], vo[V], zo[V];
#pragma unroll
        for (int c = 0; c < V; ++c) xo[c] = mo[c] = vo[c] = zo[c] = 0.f;
        const bool valid = act && e0 < D;
        if (valid) {
          ldv<V>(xrow + e0, xo);
          ldslot<V>(s0row, ia0, ib0, new0, e0, mo);
          ldslot<V>(s0row, ia0, ib0, new0, e0 + D, vo);
          ldslot<V>(s0row, ia0, ib0, new0, e0 + 2 * D, zo);
        }
#pragma unroll
        for (int c = 0; c < V; ++c) {
          const float gg = gv[k][c];
          const float mn = a.b1 * mo[c] + omb1 * gg;
          const float vn = a.b2 * vo[c] + omb2 * (gg * gg);
          const float s = sqrtf(vn);
          float d;
          if (OPT == OPT_ADAM_V4) {
            d = (a.b1 > a.b1p) ? (s - sqrtf(vo[c])) * xo[c] : (s + a.eps) * xo[c];
          } else {
            d = (a.b1 > a.b1p) ? (s - sqrtf(vo[c])) / a.lr * xo[c]
                               : (s - sqrtf(vo[c]) + a.eps) / a.lr * xo[c];
          }
          const float zn = zo[c] + (a.alpha * mn - d);
          const float adj = fmaxf(fminf(zn, a.l1), -a.l1);
          const float uv = adj - zn;
          m[k][c] = mn; nv[k][c] = vn; sq[k][c] = s; z[k][c] = zn; uu[k][c] = uv;
          if (valid) part += uv * uv;
        }
      }
      const float norm = sqrtf(group_sum<LPR>(part));
      const bool upd = norm > a.l21_norm;
      const float scale = 1.f - a.l21_norm / norm;
      const float two_l2 = 2.f * a.l2;
      bool big = false, sbig = false;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int e0 = (lane + k * LPR) * V;
        if (act && e0 < D) {
          float xn[V];
#pragma unroll
          for (int c = 0; c < V; ++c) {
            xn[c] = 0.f;  // blacklist: the row reads as zeros (table_manager.h:335-357)
            if (upd) {
              const float y = (OPT == OPT_ADAM_V4) ? (sq[k][c] + a.eps) + two_l2
                                                   : (sq[k][c] + a.eps) / a.lr + two_l2;
              xn[c] = uu[k][c] * scale / y;
            }
            big |= fabsf(xn[c]) >= CUTOFF;
            sbig |= fabsf(m[k][c]) >= CUTOFF || fabsf(nv[k][c]) >= CUTOFF || fabsf(z[k][c]) >= CUTOFF;
          }
          stv<V>(xrow + e0, xn);
          stv<V>(s0row + e0, m[k]);
          stv<V>(s0row + e0 + D, nv[k]);
          stv<V>(s0row + e0 + 2 * D, z[k]);
        }
      }
      const bool anyx = group_any<LPR>(big), anys = group_any<LPR>(sbig);
      if (act && lane == 0) {
        // CoverUpdateUnsafe -> UpdateUnderThreshold, or MarkBlacklistUnsafe (:7187-7195)
        *flags_ptr(tv, rv) = (unsigned char)(upd ? (anyx ? 0u : FLAG_UNDER) : (FLAG_BLACK | FLAG_UNDER));
        *flags_ptr(ts0, r0) = (unsigned char)(anys ? 0u : FLAG_UNDER);
      }
    } else if (OPT == OPT_ADAGRAD) {
      // training_ops.cc:1470-1482.  No CoverUpdate: flags of existing rows are left alone.
      bool sbig = false;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int e0 = (lane + k * LPR) * V;
        if (act && e0 < D) {
          float xo[V], acc[V];
          ldv<V>(xrow + e0, xo);
          ldslot<V>(s0row, ia0, ib0, new0, e0, acc);
#pragma unroll
          for (int c = 0; c < V; ++c) {
            const float gg = gv[k][c];
            sbig |= fabsf(acc[c]) >= CUTOFF;
            if (a.update_slots) acc[c] = acc[c] + gg * gg;
            xo[c] = (D > 1) ? xo[c] - (a.lr * gg) * (1.f / sqrtf(acc[c]))
                            : xo[c] - (a.lr * gg) / sqrtf(acc[c]);
          }
          stv<V>(xrow + e0, xo);
          stv<V>(s0row + e0, acc);
        }
      }
      const bool anys = group_any<LPR>(sbig);
      if (act && new0 && lane == 0) *flags_ptr(ts0, r0) = (unsigned char)(anys ? 0u : FLAG_UNDER);
    } else {
      // OPT_FTRL: training_ops.cc:713-751 with has_l2_shrinkage; ts0 = accum, ts1 = linear
      float x[K][V], ac[K][V], z[K][V], uu[K][V];
      float part = 0.f;
      const bool half = a.lr_power == -0.5f;
      const float two_l2s = 2.f * a.l2s;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int e0 = (lane + k * LPR) * V;
        float zo[V];
#pragma unroll
        for (int c = 0; c < V; ++c) x[k][c] = ac[k][c] = zo[c] = 0.f;
        const bool valid = act && e0 < D;
        if (valid) {
          ldv<V>(xrow + e0, x[k]);
          ldslot<V>(s0row, ia0, ib0, new0, e0, ac[k]);
          ldslot<V>(s1row, ia1, ib1, new1, e0, zo);
        }
#pragma unroll
        for (int c = 0; c < V; ++c) {
          const float xo = x[k][c], ao = ac[k][c];
          const float gs = gv[k][c] + two_l2s * xo;
          const float na = ao + gs * gs;
          const float pn = half ? sqrtf(na) : powf(na, -a.lr_power);
          const float po = half ? sqrtf(ao) : powf(ao, -a.lr_power);
          const float zn = zo[c] + (gs - (pn - po) / a.lr * xo);
          const float adj = fmaxf(fminf(zn, a.l1), -a.l1);
          const float uv = adj - zn;
          z[k][c] = zn; uu[k][c] = uv;
          if (valid) part += uv * uv;
        }
      }
      const float norm = sqrtf(group_sum<LPR>(part));
      const bool upd = norm > a.l21_norm;
      const float scale = 1.f - (a.l21_norm / norm);
      const float two_l2 = 2.f * a.l2;
      bool big = false, abig = false, zbig = false;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int e0 = (lane + k * LPR) * V;
        if (act && e0 < D) {
          float xn[V], an[V];
#pragma unroll
          for (int c = 0; c < V; ++c) {
            const float xo = x[k][c];
            const float gs = gv[k][c] + two_l2s * xo;
            const float na = ac[k][c] + gs * gs;
            const float pn = half ? sqrtf(na) : powf(na, -a.lr_power);
            xn[c] = 0.f;
            if (upd) xn[c] = uu[k][c] * scale / (pn / a.lr + two_l2);
            // accum += grad_to_use.square() re-evaluates the lazy expression with the updated
            // var (:747); on the blacklist branch the reference reads a freed row — we keep
            // the pre-blacklist value like oracle/kv_oracle.cc
            const float xa = upd ? xn[c] : xo;
            const float gs2 = gv[k][c] + two_l2s * xa;
            an[c] = ac[k][c] + gs2 * gs2;
            big |= fabsf(xn[c]) >= CUTOFF;
            abig |= fabsf(an[c]) >= CUTOFF;
            zbig |= fabsf(z[k][c]) >= CUTOFF;
          }
          stv<V>(xrow + e0, xn);
          stv<V>(s0row + e0, an);
          stv<V>(s1row + e0, z[k]);
        }
      }
      const bool anyx = group_any<LPR>(big), anya = group_any<LPR>(abig), anyz = group_any<LPR>(zbig);
      if (act && lane == 0) {
        *flags_ptr(tv, rv) = (unsigned char)(upd ? (anyx ? 0u : FLAG_UNDER) : (FLAG_BLACK | FLAG_UNDER));
        *flags_ptr(ts0, r0) = (unsigned char)(anya ? 0u : FLAG_UNDER);
        *flags_ptr(ts1, r1) = (unsigned char)(anyz ? 0u : FLAG_UNDER);
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < 8) w.ctr_next[threadIdx.x] = 0;
}

// ------------------------------------------------------------------------------------------
// maintenance kernels
// ------------------------------------------------------------------------------------------
__global__ void k_fill_entries(Entry* e, unsigned long long count) {
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (unsigned long long)gridDim.x * blockDim.x) {
    Entry v; v.key = EMPTY_KEY; v.row = 0; v.pad = 0;
    *reinterpret_cast<uint4*>(&e[i]) = *reinterpret_cast<uint4*>(&v);
  }
}
__global__ void k_fill_i64(long long* p, long long v, unsigned long long count) {
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (unsigned long long)gridDim.x * blockDim.x) p[i] = v;
}
// re-insert rows [1, next_row) into a fresh index
__global__ void k_rehash(TableDev t, unsigned nrows) {
  for (unsigned r = 1 + blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += gridDim.x * blockDim.x) {
    const long long key = *key_ptr(t, r);
    if (key == EMPTY_KEY) {
      Entry* s = &t.entries[t.mask + 1];
      s->key = 0; s->row = r;
      continue;
    }
    unsigned long long p = mix64((unsigned long long)key) & t.mask;
    for (;;) {
      unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&t.entries[p].key),
                                         (unsigned long long)EMPTY_KEY, (unsigned long long)key);
      if (old == (unsigned long long)EMPTY_KEY) { t.entries[p].row = r; break; }
      p = (p + 1) & t.mask;
    }
  }
}

// size() / sum_freq() kv_variable.h:139-175 ; out[0] = size, out[1] = sum_freq
__global__ void k_stats(TableDev t, unsigned nrows, unsigned long long* out) {
  unsigned long long c = 0, f = 0;
  for (unsigned r = 1 + blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += gridDim.x * blockDim.x) {
    const unsigned fl = *flags_ptr(t, r);
    const unsigned fr = *freq_ptr(t, r) & 0xFFFFu;
    if (!(fl & FLAG_BLACK) && fr >= t.enter_threshold) { c += 1; f += fr; }
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

// ExportValues dynamic_save.hpp:47-195.  cnt[0..2] = rows, blacklist, freq.  fill != 0 writes.
__global__ void k_export(TableDev t, unsigned nrows, int first_n, int fill, unsigned long long* cnt,
                         long long* keys, float* values, long long* blacklist, long long* fkeys,
                         unsigned* fvals) {
  const int D = t.dim;
  for (unsigned r = 1 + blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += gridDim.x * blockDim.x) {
    const unsigned fl = *flags_ptr(t, r);
    const unsigned fw = *freq_ptr(t, r);
    const long long key = *key_ptr(t, r);
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

// scatter / insert on the deduplicated unique list (ids must be unique per call)
// ScatterUpdate kv_variable.h:616-734 ; InsertOrUpdate kv_variable.h:423-485
__global__ void __launch_bounds__(TB) k_scatter(TableDev t, WsDev w, const float* __restrict__ upd,
                                                int op, int is_insert) {
  const int D = t.dim;
  const int lane8 = threadIdx.x & 7;
  const unsigned U = w.ctr[0];
  for (unsigned u0 = blockIdx.x * (TB / 8); u0 < U; u0 += gridDim.x * (TB / 8)) {
    const unsigned u = u0 + (threadIdx.x >> 3);
    bool big = false;
    unsigned r = 0;
    bool touch = false;
    if (u < U) {
      r = w.urow[u] & ROW_MASK;
      const unsigned g = w.usslot[u];
      const float* src = upd + (size_t)w.ufirst[u] * D;
      const unsigned fl = r ? *flags_ptr(t, r) : FLAG_BLACK;
      // scatter leaves blacklisted rows alone (:690); insert overwrites and keeps flags' blacklist
      touch = r != 0 && (is_insert || !(fl & FLAG_BLACK));
      if (touch) {
        float* row = row_ptr(t, r);
        for (int e = lane8; e < D; e += 8) {
          const float l = row[e], v = src[e];
          float o;
          switch (op) {
            case KV_SCATTER_ADD: o = l + v; break;
            case KV_SCATTER_SUB: o = l - v; break;
            case KV_SCATTER_MUL: o = l * v; break;
            case KV_SCATTER_DIV: o = l / v; break;
            case KV_SCATTER_MIN: o = fminf(l, v); break;
            case KV_SCATTER_MAX: o = fmaxf(l, v); break;
            default: o = v;
          }
          row[e] = o;
          big |= fabsf(o) >= CUTOFF;
        }
      }
      if (lane8 == 0) {
        w.smeta[g] = make_uint2(0u, 0u);
        w.skeys[g] = EMPTY_KEY;
      }
    }
    unsigned long long m = __ballot(big);
    const int lane = threadIdx.x & 63;
    const bool any = ((m >> (lane & ~7)) & 0xFFull) != 0;
    if (touch && lane8 == 0) {
      unsigned char* fp = flags_ptr(t, r);
      *fp = (unsigned char)((*fp & FLAG_BLACK) ? (FLAG_BLACK | FLAG_UNDER) : (any ? 0u : FLAG_UNDER));
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < 8) w.ctr_next[threadIdx.x] = 0;
}

// kv_dedup_segment_sum output stage
__global__ void __launch_bounds__(TB) k_dedup_emit(WsDev w, const float* __restrict__ grad, int D,
                                                   long long* uniq, float* summed) {
  const unsigned U = w.ctr[0];
  const int lane8 = threadIdx.x & 7;
  for (unsigned u = blockIdx.x * (TB / 8) + (threadIdx.x >> 3); u < U; u += gridDim.x * (TB / 8)) {
    const unsigned g = w.usslot[u];
    const bool multi = w.smeta[g].x >= 2u;
    const float* src = multi ? w.gacc + (size_t)u * D : grad + (size_t)w.ufirst[u] * D;
    for (int e = lane8; e < D; e += 8) {
      summed[(size_t)u * D + e] = src[e];
      if (multi) w.gacc[(size_t)u * D + e] = 0.f;
    }
    if (lane8 == 0) uniq[u] = w.ukey[u];
  }
}
__global__ void k_dedup_inverse(WsDev w, long long n, int* inverse) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    inverse[i] = (int)w.smeta[w.sslot_of_id[i]].y;
}
__global__ void k_dedup_clean(WsDev w) {
  const unsigned U = w.ctr[0];
  for (unsigned u = blockIdx.x * blockDim.x + threadIdx.x; u < U; u += gridDim.x * blockDim.x) {
    const unsigned g = w.usslot[u];
    w.smeta[g] = make_uint2(0u, 0u);
    w.skeys[g] = EMPTY_KEY;
  }
  if (blockIdx.x == 0 && threadIdx.x < 8) w.ctr_next[threadIdx.x] = 0;
}

// import helpers: blacklist marks / frequency words on existing-or-new keys (unique list)
__global__ void k_import_mark(TableDev t, WsDev w, int what, const unsigned* fvals) {
  const unsigned U = w.ctr[0];
  for (unsigned u = blockIdx.x * blockDim.x + threadIdx.x; u < U; u += gridDim.x * blockDim.x) {
    const unsigned r = w.urow[u] & ROW_MASK;
    const unsigned g = w.usslot[u];
    if (r) {
      if (what == 0) {  // blacklist: zero row, flags
        float* row = row_ptr(t, r);
        for (int e = 0; e < t.dim; ++e) row[e] = 0.f;
        *flags_ptr(t, r) = (unsigned char)(FLAG_BLACK | FLAG_UNDER);
      } else {
        *freq_ptr(t, r) = fvals[w.ufirst[u]];
      }
    }
    w.smeta[g] = make_uint2(0u, 0u);
    w.skeys[g] = EMPTY_KEY;
  }
  if (blockIdx.x == 0 && threadIdx.x < 8) w.ctr_next[threadIdx.x] = 0;
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
  long long cap_n = 0;
  unsigned long long S = 0;
  long long* skeys = nullptr;
  uint2* smeta = nullptr;
  unsigned* srow = nullptr;
  unsigned* sslot_of_id = nullptr;
  long long* ukey = nullptr;
  unsigned* urow = nullptr;
  unsigned* usslot = nullptr;
  unsigned* ufirst = nullptr;
  unsigned* ctr = nullptr;  // [2][8]
  unsigned long long* dbg = nullptr;
  unsigned long long seq = 0;
  float* gacc = nullptr;
  long long gacc_elems = 0;
};

}  // namespace

struct kv_table {
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
  unsigned* d_counters = nullptr;  // [0] next_row [1] error
  unsigned long long rows_cap = 0;  // chunks.size() << chunk_bits
  unsigned long long rows_ub = 1;   // upper bound of next_row
  // init table
  float* init_table = nullptr;
  long long init_rows = 0;
  bool initialized = false;
  Workspace ws;
  unsigned long long* d_stat = nullptr;  // [4]
  std::mutex mu;
  // optional per-kernel timing (kv_profile_*): event pairs recorded on the op's stream
  bool prof = false;
  std::vector<hipEvent_t> ev;
  std::vector<int> ev_kind;
  size_t ev_used = 0;
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
  d.entries = t->entries;
  d.mask = t->cap - 1;
  d.chunks = t->d_chunks;
  d.chunk_bits = t->chunk_bits;
  d.counters = t->d_counters;
  d.max_rows = (unsigned)std::min<unsigned long long>(t->rows_cap, 0x7FFFFFFFull);
  d.init_table = t->init_table;
  d.init_rows = (unsigned)t->init_rows;
  d.dim = t->dim;
  d.enter_threshold = t->enter_threshold;
  d.seed = t->seed;
  return d;
}

int add_chunk(kv_table* t, hipStream_t s) {
  if (t->chunks.size() >= (size_t)MAX_CHUNKS) return fail(KV_RESOURCE_EXHAUSTED, "row slab: too many chunks");
  const size_t R = (size_t)1 << t->chunk_bits;
  Chunk c{};
  HIP_TRY(hipMalloc(&c.rows, R * t->dim * sizeof(float)));
  HIP_TRY(hipMalloc(&c.freq, R * sizeof(unsigned)));
  HIP_TRY(hipMalloc(&c.flags, R));
  HIP_TRY(hipMalloc(&c.keys, R * sizeof(long long)));
  if (t->chunks.empty()) {
    // row 0: the permanent zero row
    HIP_TRY(hipMemsetAsync(c.rows, 0, (size_t)t->dim * sizeof(float), s));
    HIP_TRY(hipMemsetAsync(c.freq, 0, sizeof(unsigned), s));
    HIP_TRY(hipMemsetAsync(c.flags, 0, 1, s));
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
  t->entries = ne;
  t->cap = newcap;
  if (nrows > 1) {
    k_rehash<<<nblocks(nrows, TB), TB, 0, s>>>(dev_view(t), nrows);
  }
  HIP_TRY(hipStreamSynchronize(s));
  if (old) HIP_TRY(hipFree(old));
  return KV_OK;
}

// make room for `extra` more keys (worst case: every id of the batch is new)
int ensure_capacity(kv_table* t, long long extra, hipStream_t s) {
  unsigned long long need = t->rows_ub + (unsigned long long)extra;
  if (need > t->rows_cap || need * 2 > t->cap) {
    // refresh the exact row count before deciding to grow
    unsigned c[2];
    HIP_TRY(hipMemcpyAsync(c, t->d_counters, sizeof c, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (c[1]) return fail(KV_INTERNAL, "row slab overflow detected on device");
    t->rows_ub = c[0];
    need = t->rows_ub + (unsigned long long)extra;
    if (need >= 0x7FFFFFFFull) return fail(KV_RESOURCE_EXHAUSTED, "more than 2^31 rows in one table");
    while (need > t->rows_cap) {
      int rc = add_chunk(t, s);
      if (rc) return rc;
    }
    if (need * 2 > t->cap) {
      unsigned long long nc = pow2ceil(std::max(need * 2, t->cap * 2));
      int rc = build_index(t, nc, (unsigned)t->rows_ub, s);
      if (rc) return rc;
    }
  }
  t->rows_ub += (unsigned long long)extra;
  return KV_OK;
}

int ensure_workspace(kv_table* t, long long n, bool need_gacc, hipStream_t s) {
  Workspace& w = t->ws;
  if (n > w.cap_n) {
    HIP_TRY(hipStreamSynchronize(s));
    long long cap = std::max<long long>(n, 1024);
    if (w.cap_n) cap = std::max<long long>(cap, w.cap_n * 2);
    hipFree(w.skeys); hipFree(w.smeta); hipFree(w.srow); hipFree(w.sslot_of_id);
    hipFree(w.ukey); hipFree(w.urow); hipFree(w.usslot); hipFree(w.ufirst); hipFree(w.gacc);
    w.gacc = nullptr; w.gacc_elems = 0;
    w.S = pow2ceil((unsigned long long)cap * 2);
    HIP_TRY(hipMalloc(&w.skeys, (w.S + 1) * sizeof(long long)));
    HIP_TRY(hipMalloc(&w.smeta, (w.S + 1) * sizeof(uint2)));
    HIP_TRY(hipMalloc(&w.srow, (w.S + 1) * sizeof(unsigned)));
    HIP_TRY(hipMalloc(&w.sslot_of_id, cap * sizeof(unsigned)));
    HIP_TRY(hipMalloc(&w.ukey, cap * sizeof(long long)));
    HIP_TRY(hipMalloc(&w.urow, cap * sizeof(unsigned)));
    HIP_TRY(hipMalloc(&w.usslot, cap * sizeof(unsigned)));
    HIP_TRY(hipMalloc(&w.ufirst, cap * sizeof(unsigned)));
    k_fill_i64<<<nblocks((long long)w.S + 1, TB, 8192), TB, 0, s>>>(w.skeys, EMPTY_KEY, w.S + 1);
    HIP_TRY(hipMemsetAsync(w.smeta, 0, (w.S + 1) * sizeof(uint2), s));
    HIP_TRY(hipMemsetAsync(w.srow, 0, (w.S + 1) * sizeof(unsigned), s));
    if (!w.ctr) {
      HIP_TRY(hipMalloc(&w.ctr, 16 * sizeof(unsigned)));
      HIP_TRY(hipMemsetAsync(w.ctr, 0, 16 * sizeof(unsigned), s));
    }
#ifdef KV_STAMPS
    hipFree(w.dbg);
    HIP_TRY(hipMalloc(&w.dbg, (size_t)(cap / TILE + 1) * 16 * sizeof(unsigned long long)));
#endif
    w.cap_n = cap;
  }
  if (need_gacc && w.gacc_elems < w.cap_n * t->dim) {
    HIP_TRY(hipStreamSynchronize(s));
    hipFree(w.gacc);
    w.gacc_elems = w.cap_n * t->dim;
    HIP_TRY(hipMalloc(&w.gacc, (size_t)w.gacc_elems * sizeof(float)));
    HIP_TRY(hipMemsetAsync(w.gacc, 0, (size_t)w.gacc_elems * sizeof(float), s));
  }
  return KV_OK;
}

WsDev ws_view(kv_table* t, long long n) {
  Workspace& w = t->ws;
  WsDev d;
  d.skeys = w.skeys; d.smeta = w.smeta; d.srow = w.srow;
  d.smask = w.S - 1;
  d.sshift = 64 - ilog2(w.S);
  d.sslot_of_id = w.sslot_of_id;
  d.ukey = w.ukey; d.urow = w.urow; d.usslot = w.usslot; d.ufirst = w.ufirst;
  d.ctr = w.ctr + 8 * (w.seq & 1);
  d.ctr_next = w.ctr + 8 * ((w.seq + 1) & 1);
  d.gacc = w.gacc;
  d.dbg = w.dbg;
  w.seq++;
  return d;
}

// brackets one kernel launch with a pair of events when profiling is on
struct ProfScope {
  kv_table* t;
  hipStream_t s;
  bool on;
  ProfScope(kv_table* t_, int kind, hipStream_t s_) : t(t_), s(s_), on(false) {
    if (t->prof && t->ev_used + 2 <= t->ev.size()) {
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

unsigned today(const kv_table* t) {
  if (t->fixed_day >= 0) return (unsigned)t->fixed_day & 0xFFFFu;
  return (unsigned)(std::time(nullptr) / (3600 * 24)) & 0xFFFFu;  // utility.cc:38-40
}

template <int MODE>
void launch_dedup(kv_table* t, const TableDev& td, const WsDev& wd, const void* ids, const int* counts,
                  long long n, unsigned day, hipStream_t s) {
  const int grid = (int)((n + TILE - 1) / TILE);
  if (t->key_dtype == KV_DT_INT32)
    k_dedup_find<MODE, int><<<grid, TB, 0, s>>>(td, wd, (const int*)ids, counts, n, day);
  else
    k_dedup_find<MODE, long long><<<grid, TB, 0, s>>>(td, wd, (const long long*)ids, counts, n, day);
}

int check_table(kv_handle_t h) {
  if (!h) return fail(KV_INVALID_ARGUMENT, "null table handle");
  return KV_OK;
}

// locks tables in address order like MaybeLockVariableInputMutexesInOrder (training_ops.cc:96-184)
struct MultiLock {
  std::vector<kv_table*> ts;
  explicit MultiLock(std::initializer_list<kv_table*> l) : ts(l) {
    std::sort(ts.begin(), ts.end());
    ts.erase(std::unique(ts.begin(), ts.end()), ts.end());
    for (auto* t : ts) t->mu.lock();
  }
  ~MultiLock() { for (auto it = ts.rbegin(); it != ts.rend(); ++it) (*it)->mu.unlock(); }
};

// dispatch k_apply on the row geometry: D % 4 == 0 -> float4 lanes, else scalar lanes
template <int OPT>
int launch_apply(const TableDev& tv, const TableDev& t0, const TableDev& t1, const WsDev& wd,
                 const float* grad, const OptArgs& a, unsigned day, long long n, hipStream_t s) {
  const int D = tv.dim;
#define KV_LAUNCH(V, LPR, K)                                                                  \
  do {                                                                                        \
    const int gpb = TB / (LPR);                                                               \
    k_apply<OPT, V, LPR, K><<<nblocks(n, gpb, 2048), TB, 0, s>>>(tv, t0, t1, wd, grad, a, day); \
    return KV_OK;                                                                             \
  } while (0)
  if ((D & 3) == 0) {
    const int q = D / 4;
    if (q <= 1) KV_LAUNCH(4, 1, 1);
    if (q <= 2) KV_LAUNCH(4, 2, 1);
    if (q <= 4) KV_LAUNCH(4, 4, 1);
    if (q <= 8) KV_LAUNCH(4, 8, 1);
    if (q <= 16) KV_LAUNCH(4, 16, 1);
    if (q <= 32) KV_LAUNCH(4, 32, 1);
    if (q <= 64) KV_LAUNCH(4, 64, 1);
    if (q <= 128) KV_LAUNCH(4, 64, 2);
    if (q <= 256) KV_LAUNCH(4, 64, 4);
  } else {
    if (D <= 1) KV_LAUNCH(1, 1, 1);
    if (D <= 2) KV_LAUNCH(1, 2, 1);
    if (D <= 4) KV_LAUNCH(1, 4, 1);
    if (D <= 8) KV_LAUNCH(1, 8, 1);
    if (D <= 16) KV_LAUNCH(1, 16, 1);
    if (D <= 32) KV_LAUNCH(1, 32, 1);
    if (D <= 64) KV_LAUNCH(1, 64, 1);
    if (D <= 128) KV_LAUNCH(1, 64, 2);
    if (D <= 256) KV_LAUNCH(1, 64, 4);
  }
#undef KV_LAUNCH
  return fail(KV_UNIMPLEMENTED, "embedding dim %d not supported by the fused apply kernels "
              "(multiples of 4 up to 1024, any dim up to 256)", D);
}

// shared front half of every optimizer op: validation common to all, capacity, dedup, accumulate
int apply_prologue(kv_table* v, std::initializer_list<kv_table*> slots, const float* grad,
                   const void* ids, long long n, hipStream_t s, TableDev* tv, WsDev* wd, unsigned* day) {
  if (n < 0 || n > (1ll << 30)) return fail(KV_INVALID_ARGUMENT, "indices: bad length %lld", n);
  if (n > 0 && (!grad || !ids)) return fail(KV_INVALID_ARGUMENT, "grad / indices pointer is null");
  int rc;
  if ((rc = ensure_capacity(v, n, s))) return rc;
  for (auto* sl : slots)
    if ((rc = ensure_capacity(sl, n, s))) return rc;
  if ((rc = ensure_workspace(v, n, true, s))) return rc;
  *tv = dev_view(v);
  *wd = ws_view(v, n);
  *day = today(v);
  {
    ProfScope ps(v, KV_PROF_APPLY_DEDUP_FIND, s);
    launch_dedup<MODE_APPLY>(v, *tv, *wd, ids, nullptr, n, *day, s);
  }
  {
    ProfScope ps(v, KV_PROF_APPLY_ACCUMULATE, s);
    launch_accumulate(*wd, grad, n, v->dim, s);
  }
  return KV_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
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
    unsigned init[8] = {1, 0, 0, 0, 0, 0, 0, 0};  // next_row = 1 (row 0 is the zero row)
    if (hipMemcpy(t->d_counters, init, sizeof init, hipMemcpyHostToDevice) != hipSuccess) {
      rc = fail(KV_INTERNAL, "hipMemcpy failed");
      break;
    }
    if ((rc = add_chunk(t, s))) break;
    if ((rc = build_index(t, pow2ceil(std::max<unsigned long long>(2 * t->rows_cap, 1024)), 1, s))) break;
  } while (0);
  if (rc) { kv_destroy(t); return rc; }
  *out = t;
  return KV_OK;
}

int kv_destroy(kv_handle_t t) {
  if (!t) return KV_OK;
  DeviceGuard dg(t->device);
  hipDeviceSynchronize();
  for (auto& c : t->chunks) { hipFree(c.rows); hipFree(c.freq); hipFree(c.flags); hipFree(c.keys); }
  hipFree(t->entries); hipFree(t->d_chunks); hipFree(t->d_counters); hipFree(t->d_stat);
  hipFree(t->init_table);
  for (auto e : t->ev) hipEventDestroy(e);
  Workspace& w = t->ws;
  hipFree(w.skeys); hipFree(w.smeta); hipFree(w.srow); hipFree(w.sslot_of_id); hipFree(w.ukey);
  hipFree(w.urow); hipFree(w.usslot); hipFree(w.ufirst); hipFree(w.ctr); hipFree(w.gacc);
  delete t;
  return KV_OK;
}

int kv_reserve(kv_handle_t t, int64_t capacity) {
  int rc;
  if ((rc = check_table(t))) return rc;
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  unsigned long long save = t->rows_ub;
  long long extra = capacity + 1 - (long long)t->rows_ub;
  if (extra <= 0) return KV_OK;
  rc = ensure_capacity(t, extra, nullptr);
  t->rows_ub = std::min(save, t->rows_ub);  // reserve does not consume the bound
  return rc;
}

int kv_init_table(kv_handle_t t, const float* table, int64_t rows, kv_stream_t stream) {
  int rc;
  if ((rc = check_table(t))) return rc;
  if (!table || rows <= 0) return fail(KV_INVALID_ARGUMENT, "random_initializer must be a non-empty [rows, dim] matrix");
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  if (t->initialized) return KV_OK;  // "re-initialization ignored" kv_variable.h:188-193
  HIP_TRY(hipMalloc(&t->init_table, (size_t)rows * t->dim * sizeof(float)));
  HIP_TRY(hipMemcpyAsync(t->init_table, table, (size_t)rows * t->dim * sizeof(float),
                         hipMemcpyDeviceToDevice, (hipStream_t)stream));
  t->init_rows = rows;
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
  return KV_OK;
}
int kv_set_seed(kv_handle_t t, uint64_t seed) {
  int rc;
  if ((rc = check_table(t))) return rc;
  t->seed = seed;
  return KV_OK;
}

static int stats(kv_handle_t t, hipStream_t s, unsigned long long out[2], unsigned* nrows_out) {
  unsigned c[2];
  HIP_TRY(hipMemcpyAsync(c, t->d_counters, sizeof c, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (c[1]) return fail(KV_INTERNAL, "row slab overflow detected on device");
  t->rows_ub = c[0];
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
  *out = (int64_t)nrows - 1;
  return KV_OK;
}

int kv_get_meta(kv_handle_t t, const int64_t* ids, int64_t n, uint32_t* fw, uint8_t* fl, kv_stream_t stream) {
  int rc;
  if ((rc = check_table(t))) return rc;
  if (n <= 0) return KV_OK;
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  k_get_meta<long long><<<nblocks(n, TB), TB, 0, (hipStream_t)stream>>>(dev_view(t), (const long long*)ids, n, fw, fl);
  HIP_TRY(hipGetLastError());
  return KV_OK;
}

int kv_gather_or_insert(kv_handle_t t, const void* ids, const int32_t* counts, int64_t n, float* out,
                        kv_stream_t stream) {
  int rc;
  if ((rc = check_table(t))) return rc;
  if (n == 0) return KV_OK;  // kv_variable_ops.cc:530-532
  if (n < 0 || n > (1ll << 30)) return fail(KV_INVALID_ARGUMENT, "indices: bad length %lld", (long long)n);
  if (!ids || !out) return fail(KV_INVALID_ARGUMENT, "indices / output pointer is null");
  if (!t->initialized)
    return fail(KV_FAILED_PRECONDITION, "Failed to use uninitialized variables: KvVariable init table not set");
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  hipStream_t s = (hipStream_t)stream;
  if ((rc = ensure_capacity(t, n, s))) return rc;
  if ((rc = ensure_workspace(t, n, false, s))) return rc;
  const TableDev td = dev_view(t);
  const WsDev wd = ws_view(t, n);
  const unsigned day = today(t);
  {
    ProfScope ps(t, KV_PROF_LOOKUP_DEDUP_FIND, s);
    launch_dedup<MODE_LOOKUP>(t, td, wd, ids, counts, n, day, s);
  }
  ProfScope ps_gather(t, KV_PROF_LOOKUP_GATHER, s);
  const int D = t->dim;
  const int q = (D % 4 == 0) ? D / 4 : 0;
  const bool vec = q > 0 && (q & (q - 1)) == 0 && q <= TB;
  const long long rows_per_block = vec ? TB / q : 1;
  int gb = vec ? nblocks((n + 3) / 4, (int)rows_per_block, 4096) : nblocks(n * D, TB, 4096);
  const int grid = gb;
  switch (vec ? q : 0) {
    case 1: k_gather<1><<<grid, TB, 0, s>>>(td, wd, out, n, day, gb); break;
    case 2: k_gather<2><<<grid, TB, 0, s>>>(td, wd, out, n, day, gb); break;
    case 4: k_gather<4><<<grid, TB, 0, s>>>(td, wd, out, n, day, gb); break;
    case 8: k_gather<8><<<grid, TB, 0, s>>>(td, wd, out, n, day, gb); break;
    case 16: k_gather<16><<<grid, TB, 0, s>>>(td, wd, out, n, day, gb); break;
    case 32: k_gather<32><<<grid, TB, 0, s>>>(td, wd, out, n, day, gb); break;
    case 64: k_gather<64><<<grid, TB, 0, s>>>(td, wd, out, n, day, gb); break;
    case 128: k_gather<128><<<grid, TB, 0, s>>>(td, wd, out, n, day, gb); break;
    case 256: k_gather<256><<<grid, TB, 0, s>>>(td, wd, out, n, day, gb); break;
    default: k_gather<0><<<grid, TB, 0, s>>>(td, wd, out, n, day, gb); break;
  }
  HIP_TRY(hipGetLastError());
  return KV_OK;
}

int kv_gather_or_zeros(kv_handle_t t, const void* ids, int64_t n, float* out, kv_stream_t stream) {
  int rc;
  if ((rc = check_table(t))) return rc;
  if (n == 0) return KV_OK;
  if (n < 0 || !ids || !out) return fail(KV_INVALID_ARGUMENT, "indices / output pointer is null");
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  hipStream_t s = (hipStream_t)stream;
  const TableDev td = dev_view(t);
  if (t->key_dtype == KV_DT_INT32)
    k_gather_or_zeros<int><<<nblocks(n, TB / 8, 8192), TB, 0, s>>>(td, (const int*)ids, out, n);
  else
    k_gather_or_zeros<long long><<<nblocks(n, TB / 8, 8192), TB, 0, s>>>(td, (const long long*)ids, out, n);
  HIP_TRY(hipGetLastError());
  return KV_OK;
}

int kv_apply_group_adam(kv_handle_t v, kv_handle_t mvl, const float* grad, const void* ids, int64_t n,
                        float lr, float b1p, float b2p, float b1, float b2, float eps, float l1,
                        float l2, float l21, int version, kv_stream_t stream) {
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
  hipStream_t s = (hipStream_t)stream;
  TableDev tv; WsDev wd; unsigned day;
  if ((rc = apply_prologue(v, {mvl}, grad, ids, n, s, &tv, &wd, &day))) return rc;
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
  const TableDev ts = dev_view(mvl);
  {
    ProfScope ps(v, KV_PROF_APPLY_UPDATE, s);
    rc = version == 4 ? launch_apply<OPT_ADAM_V4>(tv, ts, ts, wd, grad, a, day, n, s)
                      : launch_apply<OPT_ADAM_V3>(tv, ts, ts, wd, grad, a, day, n, s);
  }
  if (rc) return rc;
  HIP_TRY(hipGetLastError());
  return KV_OK;
}

int kv_apply_adagrad(kv_handle_t v, kv_handle_t acc, float lr, const float* grad, const void* ids,
                     int64_t n, int update_slots, kv_stream_t stream) {
  int rc;
  if ((rc = check_table(v)) || (rc = check_table(acc))) return rc;
  if (!v->initialized || !acc->initialized)
    return fail(KV_FAILED_PRECONDITION, "Attempting to use uninitialized variables: %s", !v->initialized ? "var" : "accum");
  if (acc->dim != v->dim) return fail(KV_INVALID_ARGUMENT, "var and accum do not have the same shape [%d] [%d]", v->dim, acc->dim);
  if (v->device != acc->device || v == acc) return fail(KV_INVALID_ARGUMENT, "var and accum must be distinct tables on one device");
  if (n == 0) return KV_OK;
  DeviceGuard dg(v->device);
  MultiLock lk({v, acc});
  hipStream_t s = (hipStream_t)stream;
  TableDev tv; WsDev wd; unsigned day;
  if ((rc = apply_prologue(v, {acc}, grad, ids, n, s, &tv, &wd, &day))) return rc;
  OptArgs a{};
  a.lr = lr; a.update_slots = update_slots;
  const TableDev ts = dev_view(acc);
  {
    ProfScope ps(v, KV_PROF_APPLY_UPDATE, s);
    rc = launch_apply<OPT_ADAGRAD>(tv, ts, ts, wd, grad, a, day, n, s);
  }
  if (rc) return rc;
  HIP_TRY(hipGetLastError());
  return KV_OK;
}

int kv_apply_sparse_group_ftrl(kv_handle_t v, kv_handle_t acc, kv_handle_t lin, const float* grad,
                               const void* ids, int64_t n, float lr, float l1, float l2, float l21,
                               float l2s, float lr_power, kv_stream_t stream) {
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
  TableDev tv; WsDev wd; unsigned day;
  if ((rc = apply_prologue(v, {acc, lin}, grad, ids, n, s, &tv, &wd, &day))) return rc;
  OptArgs a{};
  a.lr = lr; a.l1 = l1; a.l2 = l2; a.l21 = l21; a.l2s = l2s; a.lr_power = lr_power;
  a.l21_norm = l21 * std::sqrt((float)v->dim);  // :728
  {
    ProfScope ps(v, KV_PROF_APPLY_UPDATE, s);
    rc = launch_apply<OPT_FTRL>(tv, dev_view(acc), dev_view(lin), wd, grad, a, day, n, s);
  }
  if (rc) return rc;
  HIP_TRY(hipGetLastError());
  return KV_OK;
}

int kv_dedup_segment_sum(kv_handle_t t, const void* ids, const float* grad, int64_t n, int64_t* uniq,
                         float* summed, int32_t* inverse, int64_t* num_unique, kv_stream_t stream) {
  int rc;
  if ((rc = check_table(t))) return rc;
  if (!num_unique) return fail(KV_INVALID_ARGUMENT, "num_unique is null");
  *num_unique = 0;
  if (n == 0) return KV_OK;
  if (n < 0 || n > (1ll << 30) || !ids || !grad || !uniq || !summed) return fail(KV_INVALID_ARGUMENT, "bad arguments");
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  hipStream_t s = (hipStream_t)stream;
  if ((rc = ensure_workspace(t, n, true, s))) return rc;
  const TableDev td = dev_view(t);
  const WsDev wd = ws_view(t, n);
  launch_dedup<MODE_DEDUP>(t, td, wd, ids, nullptr, n, 0, s);
  launch_accumulate(wd, grad, n, t->dim, s);
  k_dedup_emit<<<nblocks(n, TB / 8, 2048), TB, 0, s>>>(wd, grad, t->dim, (long long*)uniq, summed);
  if (inverse) k_dedup_inverse<<<nblocks(n, TB, 2048), TB, 0, s>>>(wd, n, inverse);
  unsigned U = 0;
  HIP_TRY(hipMemcpyAsync(&U, wd.ctr, sizeof U, hipMemcpyDeviceToHost, s));
  k_dedup_clean<<<nblocks(n, TB, 1024), TB, 0, s>>>(wd);
  HIP_TRY(hipStreamSynchronize(s));
  *num_unique = (int64_t)U;
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
  HIP_TRY(hipMemcpy(out, t->ws.dbg, (size_t)nblocks_ * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
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
  unsigned nrows = 1;
  if ((rc = stats(t, s, nullptr, &nrows))) return rc;
  HIP_TRY(hipMemsetAsync(t->d_stat, 0, 4 * sizeof(unsigned long long), s));
  k_export<<<nblocks(nrows, TB, 2048), TB, 0, s>>>(dev_view(t), nrows, first_n, 1, t->d_stat,
                                                   (long long*)keys, values, (long long*)blacklist,
                                                   (long long*)fkeys, fvals);
  HIP_TRY(hipGetLastError());
  return KV_OK;
}

// insert / scatter / import share: dedup (MODE_SCATTER) -> k_scatter on the unique list
static int scatter_like(kv_handle_t t, const void* ids, const float* vals, int64_t n, int op,
                        int is_insert, hipStream_t s) {
  int rc;
  if (n == 0) return KV_OK;
  if (n < 0 || n > (1ll << 30) || !ids || !vals) return fail(KV_INVALID_ARGUMENT, "bad arguments");
  if (!t->initialized && !is_insert)
    return fail(KV_FAILED_PRECONDITION, "Failed to use uninitialized variables: KvVariable init table not set");
  if ((rc = ensure_capacity(t, n, s))) return rc;
  if ((rc = ensure_workspace(t, n, false, s))) return rc;
  const TableDev td = dev_view(t);
  const WsDev wd = ws_view(t, n);
  if (is_insert && !t->initialized) {
    // InsertOrUpdate never consults the init table; give new rows a defined value source
    TableDev td2 = td;
    td2.init_table = t->chunks[0].rows;  // the zero row
    td2.init_rows = 1;
    launch_dedup<MODE_SCATTER>(t, td2, wd, ids, nullptr, n, 0, s);
  } else {
    launch_dedup<MODE_SCATTER>(t, td, wd, ids, nullptr, n, 0, s);
  }
  k_scatter<<<nblocks(n, TB / 8, 2048), TB, 0, s>>>(td, wd, vals, op, is_insert);
  HIP_TRY(hipGetLastError());
  return KV_OK;
}

int kv_insert(kv_handle_t t, const void* ids, const float* values, int64_t n, kv_stream_t stream) {
  int rc;
  if ((rc = check_table(t))) return rc;
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  return scatter_like(t, ids, values, n, KV_SCATTER_ASSIGN, 1, (hipStream_t)stream);
}

int kv_scatter_update(kv_handle_t t, const void* ids, const float* updates, int64_t n, int op,
                      kv_stream_t stream) {
  int rc;
  if ((rc = check_table(t))) return rc;
  if (op < KV_SCATTER_ASSIGN || op > KV_SCATTER_MAX) return fail(KV_INVALID_ARGUMENT, "unsupported update operation %d", op);
  DeviceGuard dg(t->device);
  std::lock_guard<std::mutex> l(t->mu);
  return scatter_like(t, ids, updates, n, op, 0, (hipStream_t)stream);
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
  // clear(): dynamic_restore.hpp:60-62
  HIP_TRY(hipStreamSynchronize(s));
  unsigned init[2] = {1, 0};
  HIP_TRY(hipMemcpyAsync(t->d_counters, init, sizeof init, hipMemcpyHostToDevice, s));
  k_fill_entries<<<nblocks((long long)t->cap + 1, TB, 8192), TB, 0, s>>>(t->entries, t->cap + 1);
  t->rows_ub = 1;
  if ((rc = scatter_like(t, keys, values, n, KV_SCATTER_ASSIGN, 1, s))) return rc;
  if (n_black > 0) {
    if ((rc = ensure_capacity(t, n_black, s))) return rc;
    if ((rc = ensure_workspace(t, n_black, false, s))) return rc;
    TableDev td = dev_view(t);
    if (!t->initialized) { td.init_table = t->chunks[0].rows; td.init_rows = 1; }
    const WsDev wd = ws_view(t, n_black);
    launch_dedup<MODE_SCATTER>(t, td, wd, blacklist, nullptr, n_black, 0, s);
    k_import_mark<<<nblocks(n_black, TB, 1024), TB, 0, s>>>(td, wd, 0, nullptr);
  }
  if (n_freq > 0) {
    if ((rc = ensure_capacity(t, n_freq, s))) return rc;
    if ((rc = ensure_workspace(t, n_freq, false, s))) return rc;
    TableDev td = dev_view(t);
    if (!t->initialized) { td.init_table = t->chunks[0].rows; td.init_rows = 1; }
    const WsDev wd = ws_view(t, n_freq);
    launch_dedup<MODE_SCATTER>(t, td, wd, fkeys, nullptr, n_freq, 0, s);
    k_import_mark<<<nblocks(n_freq, TB, 1024), TB, 0, s>>>(td, wd, 1, fvals);
  }
  HIP_TRY(hipGetLastError());
  return KV_OK;
}

}  // extern "C"
