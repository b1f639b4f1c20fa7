// kv_device.h — device-side types and helpers of libkvhip (included by kvhip.hip only).
#pragma once

// ------------------------------------------------------------------------------------------
// constants
// ------------------------------------------------------------------------------------------
constexpr long long EMPTY_KEY = (long long)0x8000000000000000ULL;
constexpr unsigned FLAG_BLACK = 1u;   // EmbeddingValue::in_black_   (embedding_value.h:225)
constexpr unsigned FLAG_UNDER = 2u;   // EmbeddingValue::under_threshold_
constexpr unsigned FLAG_DIRTY = 4u;   // row changed since under_threshold was computed (Adagrad: no CoverUpdate)
constexpr float CUTOFF = 1.0e-20f;    // DEFAULT_CUTOFF_VALUE (kv_variable_interface.h:55)
constexpr unsigned FLAG_FREE = 8u;    // row released by Delete (table_manager.h:405-416): skipped by scans, reusable
constexpr unsigned ROW_TOMB = 0xFFFFFFFFu;  // index entry of a deleted key (kept so probe chains stay intact;
                                            // the same key revives it, an index rebuild drops it)
constexpr unsigned ROW_FILTERED = 0x80000000u;  // tag bit: var frequency < enter_threshold
constexpr unsigned ROW_MASK = 0x7FFFFFFFu;
constexpr unsigned HEAD_BIT = 0x80000000u;      // sorted position list / ent_base: first position of its key
constexpr unsigned BASE_MASK = 0x3FFFFFFFu;
constexpr int RANK_SHIFT = 21;                  // slot_rank: entry index (21 bits, n <= 2^21) | in-tile occurrence rank << 21
constexpr unsigned SLOT_MASK = (1u << RANK_SHIFT) - 1u;
constexpr int LCOLD = 16;   // a key with at most this many occurrences in the batch is "cold": one lane group sums its
                            // rows and updates it; a "hot" key's rows are summed in chunks by whole waves
constexpr int HC = 128;     // rows per hot chunk
// ---- the entry-list pipeline (kv_fused.h) ----
constexpr unsigned EP_TAG = 0x40000000u;    // entry list: the source is row (word & ~EP_TAG) of epart, not an input position
constexpr unsigned NEW_BIT = 0x80000000u;   // ent_b row word: the key was inserted by this batch (row part 0: by another tile, row not seen yet)
constexpr unsigned HINT_NEW = 0xFFFFFFFFu;  // Entry::hint of a key inserted by the batch in flight (the partition pass resets it)
constexpr int HC2 = 512;    // entries per hot chunk of the entry-list apply: a key has at most one entry per tile, so up to
                            // 1 M ids no key spans chunks and k_apply_fin is not launched

constexpr int TB = 256;          // threads per block of the gather / maintenance kernels
#ifndef KV_TBT
#define KV_TBT 512   // (A/B: 256 = 1024-id tiles, four tile blocks per CU)
#endif
constexpr int TBT = KV_TBT;  // threads per block of the tile kernel
constexpr int IPT = 4;           // ids per thread in the tile kernel
constexpr int TILE = TBT * IPT;  // ids per tile (2048): a key present in every tile contributes
                                 // N / 2048 entries to its partition, which keeps the partition
                                 // blocks' LDS lists small enough for 4 blocks per CU
constexpr int LS = 2 * TILE;     // LDS hash slots per tile (load <= 0.5)
constexpr int MAX_P = 2048;      // partitions (power of two); 1024 up to 1 M ids, 2048 for 2 M
constexpr int MAX_CHUNKS = 32768;   // 2^16-row chunks (no capacity hint) still reach the 2^31-row limit

// index-pass modes (k_part_keys) and fold modes (k_apply_sorted)
enum Mode { MODE_LOOKUP = 0, MODE_APPLY = 1, MODE_DEDUP = 2, MODE_SCATTER = 3, MODE_MARK = 4, MODE_UNIQUE = 5,
            MODE_APPLYIDX = 6 };
enum Opt { OPT_ADAM_V4 = 0, OPT_ADAM_V3 = 1, OPT_ADAGRAD = 2, OPT_FTRL = 3 };

struct __attribute__((aligned(16))) Entry {
  long long key;
  unsigned row;
  unsigned hint;   // row of this key in the table's attached optimizer slot table (0 = not known yet); a
                   // hint is validated against the slot row's own key before use, so a stale one only costs a probe
};

// per-row metadata next to each other (one 16-byte record, so a key's frequency word and flags
// arrive with one memory transaction and an insert writes one line): embedding_value.h:225-235
struct RowMeta {
  long long key;
  unsigned freq;          // (day << 16) | saturating u16 count
  unsigned char flags;    // FLAG_*
  unsigned char delta;    // DELTA_TRAIN: key is in train_deltalist_ (kv_variable.h:870; set only while the table tracks
                          // deltas); DELTA_PRED: in prediction_deltalist_ (:871)
  unsigned short stamp;   // serial of the last kv_apply_*_unique launch that updated the row (kv_uapply.h: how an id that
                          // breaks the caller's promise of unique ids is caught); 0 = none
};
constexpr unsigned DELTA_TRAIN = 1u, DELTA_PRED = 2u;
static_assert(sizeof(RowMeta) == 16, "RowMeta layout");

// A row's record array holds META_STRIDE 16-byte units per row: [0] the RowMeta, [1] a SlotMirror — in the SAME 32-byte
// sector, so the two are one line to read and one request to write.  Every read miss on this chip is a 128-byte line
// (profiles/r06_fetch_calibration.txt): the optimizer apply used to read two lines per key for two 16-byte records — the
// var's and the hinted slot row's.  The mirror is a write-back copy of what the apply needs of the SLOT row's record (its
// frequency word and flags), kept next to the VAR row's record: while it is valid the apply neither reads nor writes the
// slot table's own record.  `srow` names the slot row it stands for (must equal the index entry's hint), `epoch` the
// generation of the pairing (the host bumps it — one integer — whenever anything but a mirror apply may have read or
// written the slot table's records, after flushing the dirty mirrors back: kvhip.hip mirror_*), `state` 0 invalid / 1 clean /
// 2 dirty.  Only var tables of an attached (var, slot) pair use their mirrors; the units exist in every table.
constexpr int META_STRIDE = 2;
struct SlotMirror {
  unsigned srow;          // the slot row this stands for
  unsigned freq;          // its frequency word (day << 16 | saturating count)
  unsigned char flags;    // its FLAG_* byte
  unsigned char state;    // MIRROR_*
  unsigned short epoch;   // pairing generation (PartArgs::mirror_epoch)
  unsigned pad;
};
static_assert(sizeof(SlotMirror) == 16, "SlotMirror layout");
constexpr unsigned MIRROR_INVALID = 0u, MIRROR_CLEAN = 1u, MIRROR_DIRTY = 2u;

struct Chunk {
  float* rows;
  RowMeta* meta;
};

// device view of one table; passed to kernels by value
struct TableDev {
  Chunk c0;                 // chunk 0 by value: the common single-chunk table needs no table hop
  Entry* entries;
  unsigned long long mask;  // cap - 1; entries[cap] = sentinel-key home
  Chunk* chunks;
  int chunk_bits;
  unsigned* counters;  // [0] next_row  [1] error flag (row overflow)  [2] (int) rows on the free list
  const unsigned* free_rows;  // rows released by Delete, popped by inserts (nullptr: none known)
  unsigned max_rows;
  const float* init_table;
  unsigned init_rows;
  int dim;
  unsigned enter_threshold;
  unsigned long long seed;
  unsigned track_delta;     // NeedDeltaInfo() kv_variable.h:816: touched keys are remembered for DeltaExport
  unsigned* err_host;       // pinned host copy of counters[1] (the host reads it without a synchronisation)
  unsigned single;          // the slab is chunk 0 alone (every pre-sized table): row addresses need no chunk-table hop —
                            // a UNIFORM test, so the loads behind it are not fenced by a per-lane branch
};

// device view of the per-batch workspace (kv_kernels.h explains the pipeline)
struct WsDev {
  long long* ent_key;      // [ntiles * TILE] tile t's deduplicated entries, sorted by partition
  unsigned* ent_a;         // index modes: occurrences of the key in the tile (low 16) | saturating frequency
                           // count of the tile (high 16); scatter / mark: one input position of the key
  unsigned* ent_b;         // OUT of the partition pass: var row id of the key (unique: dense index)
  unsigned* ent_base;      // OUT of the partition pass: where the entry's positions start in the sorted position
                           // list | HEAD_BIT (first entry of its key)
  unsigned* ent_rec;       // OUT of the partition pass, first entry of a key only: the key's record (list index,
                           // bit 31: hot list) — k_order files the key's first input position there
  unsigned* toff;          // [ntiles][P + 1] partition boundaries inside each tile: entry prefix (low 16) |
                           // position prefix (high 16)
  unsigned* slot_rank;     // [n] entry index of every input position | its rank among the key's occurrences
                           // in the tile << RANK_SHIFT
  unsigned* order;         // [n + 1] input positions sorted by key (a key's occurrences are contiguous), first
                           // one tagged HEAD_BIT; order[n] = HEAD_BIT
  uint4* coldlist;         // [n][2] KeyRec of the cold keys: {key lo, key hi, row, slot-row hint} {start, count, first position, -}
  uint4* hotlist;          // [n][2] the hot keys: ... {start, count, first chunk (in the partition), rows per chunk}
                           // (both indexed by the partition's first sorted position + the key's number in it)
  uint4* litem;            // [n] the partitions' work items, partition p's at [its first sorted position ...)
  uint4* pmeta;            // [P] per partition: {items, hot chunks, first sorted position, cold keys}
  uint4* items;            // [n] the dense work item directory (k_gather<ORDER> / k_order build it): hot chunk
                           // {hot list index | HEAD_BIT, chunk in the key, chunk number in the batch, -}, cold
                           // batch {first cold list index, keys, -, -}
  float* hpart;            // [hot chunks][dim] partial sums of the keys that have more than one chunk
  unsigned* ctr;           // [8] op counters, zeroed by the tile pass: [0] unique count (kv_unique / kv_dedup_segment_sum),
                           // [2] work items, [3] hot chunks
  unsigned ntiles, P;
  int pshift;              // 64 - log2(P)
  unsigned seg_cap;        // (id, count) input in fixed-capacity exchange segments of this many records (0: plain list)
  int* zero_counts;        // k_tile clears this [n] array on its way (the sharded route's sparse unique counts); else null
  const int* row_map;      // k_gather: rows are read at row_map[ent_b] (sharded lookup: the exchange buffer's records); else null
  unsigned long long* dbg; // diagnostic build only (-DKV_STAMPS): per-block phase stamps
  // ---- the entry-list pipeline (kv_fused.h).  It reuses ent_b (row word), ent_base (slot-row hint), ent_rec (the
  //      entry's source: its input position, or EP_TAG | epart row), order (the entry list) and:
  unsigned* mrow;          // [n] per tile, the rows of its entries with more than one occurrence, entry by entry and in
                           // rank order: tile-local position (bits 0..10) | the entry's epart row in the tile (bits 11..20)
                           // | bit 31: first row of its entry  (in slot_rank's storage)
  unsigned* mcount;        // [ntiles] rows in mrow (low 16) | entries they belong to (high 16)
  float* epart;            // [ntiles][TILE / 2][dim] their gradient sums (k_tsum)
  // ---- sharded path: a rank's OWN segment of an exchange is not copied from the send to the receive buffer: the
  //      kernels that read a receive buffer read positions [self_lo, self_lo + self_len) from the send buffer instead
  unsigned self_lo, self_len;   // (length 0: no such range; one unsigned compare: pos - self_lo < self_len)
  const void* ids_self;         // k_ltile<IdCount>: the (id, count) records of that range (send_pairs)
  const float* grad_self;       // k_tsum / k_papply: its gradient rows (send_rows)
  unsigned short* pos_ent; // [n] k_ltile: every input position's entry number in its tile (sharded route: the finish reads
                           // position -> entry -> record); nullptr: not filed
};

// In-kernel phase stamps for the diagnostic build (never in the product .so): thread 0 of each
// block stores s_memtime at phase boundaries into a buffer nothing else reads.
#ifdef KV_STAMPS
#define KV_STAMP(slot) do { if (threadIdx.x == 0) w.dbg[(size_t)blockIdx.x * 16 + (slot)] = wall_clock64(); } while (0)
#define KV_STAMPP(slot) do { if (threadIdx.x == 0) w.dbg[(size_t)(blockIdx.x + 4096) * 16 + (slot)] = wall_clock64(); } while (0)
// where the block runs: HW_ID (wave / simd / cu / sh / se) and XCC_ID
#define KV_STAMP_HW(slot) do { if (threadIdx.x == 0) w.dbg[(size_t)blockIdx.x * 16 + (slot)] = \
    ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492); } while (0)
#define KV_STAMPPV(slot, v) do { if (threadIdx.x == 0) w.dbg[(size_t)(blockIdx.x + 4096) * 16 + (slot)] = (v); } while (0)
#define KV_STAMPA(slot) do { if (threadIdx.x == 0) w.dbg[(size_t)(blockIdx.x + 8192) * 16 + (slot)] = wall_clock64(); } while (0)
#define KV_STAMPV(slot, v) do { if (threadIdx.x == 0) w.dbg[(size_t)(blockIdx.x + 8192) * 16 + (slot)] = (v); } while (0)
#define KV_STAMPT(slot) do { if (threadIdx.x == 0) w.dbg[(size_t)(blockIdx.x + 2048) * 16 + (slot)] = wall_clock64(); } while (0)
#else
#define KV_STAMP(slot) do { } while (0)
#define KV_STAMPP(slot) do { } while (0)
#define KV_STAMP_HW(slot) do { } while (0)
#define KV_STAMPPV(slot, v) do { } while (0)
#define KV_STAMPA(slot) do { } while (0)
#define KV_STAMPV(slot, v) do { } while (0)
#define KV_STAMPT(slot) do { } while (0)
#endif

// Blocks of `threads` threads of kernel `k` that are resident at once on the CURRENT device (CUs x occupancy, at most
// `cap_per_cu` per CU: above 96 SGPRs the hardware admits 6 blocks of 256 threads, one fewer than the API says), cached
// PER DEVICE: a process that drives GPUs with different CU counts or partition modes gets each device's own number
// (a function-local static kept the first caller's device).  `Tag` makes the cache one per call site.
template <typename Tag, typename Kernel>
static inline int resident_blocks(Kernel k, int threads, int cap_per_cu, int fallback_per_cu) {
  constexpr int MAXDEV = 64;
  static int cache[MAXDEV];   // 0 = not asked yet (benign race: every thread computes the same number)
  int dev = 0;
  hipGetDevice(&dev);
  const bool slot = dev >= 0 && dev < MAXDEV;
  if (slot) { const int c = __atomic_load_n(&cache[dev], __ATOMIC_RELAXED); if (c > 0) return c; }
  int nb = 0, cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, threads, 0) != hipSuccess || nb < 1) nb = fallback_per_cu;
  if (nb > cap_per_cu) nb = cap_per_cu;
  const int r = nb * (cus > 0 ? cus : 256);
  if (slot) __atomic_store_n(&cache[dev], r, __ATOMIC_RELAXED);
  return r;
}

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
  if (t.single || (r >> t.chunk_bits) == 0) return t.c0.rows + (size_t)r * t.dim;
  const Chunk& c = t.chunks[r >> t.chunk_bits];
  return c.rows + (size_t)(r & ((1u << t.chunk_bits) - 1)) * t.dim;
}
// the whole slab is chunk 0 (every pre-sized table): rows are c0.rows + r * dim, no chunk-table hop and no branch in
// front of the load — a loop that asks this once keeps all its row loads in flight together
__device__ __forceinline__ bool single_chunk(const TableDev& t) { return t.single != 0u; }
__device__ __forceinline__ RowMeta* meta_ptr(const TableDev& t, unsigned r) {
  if (t.single || (r >> t.chunk_bits) == 0) return t.c0.meta + (size_t)r * META_STRIDE;
  return t.chunks[r >> t.chunk_bits].meta + (size_t)(r & ((1u << t.chunk_bits) - 1)) * META_STRIDE;
}
__device__ __forceinline__ SlotMirror* mirror_ptr(const TableDev& t, unsigned r) {
  return reinterpret_cast<SlotMirror*>(meta_ptr(t, r) + 1);
}
__device__ __forceinline__ unsigned* freq_ptr(const TableDev& t, unsigned r) { return &meta_ptr(t, r)->freq; }
__device__ __forceinline__ unsigned char* flags_ptr(const TableDev& t, unsigned r) { return &meta_ptr(t, r)->flags; }
__device__ __forceinline__ long long* key_ptr(const TableDev& t, unsigned r) { return &meta_ptr(t, r)->key; }
// train_deltalist_.insert(key) (kv_variable.h:316,451,685; MarkAsDeltaListElements :791-799): one byte in the
// row's own RowMeta record, a plain store by the key's single owner thread
__device__ __forceinline__ void mark_delta(const TableDev& t, unsigned r) {
  if (t.track_delta) meta_ptr(t, r)->delta |= (unsigned char)DELTA_TRAIN;
}
// frequency word and flags of a row with ONE 8-byte load: .x = freq word, .y & 0xFF = flags
__device__ __forceinline__ uint2 load_freq_flags(const TableDev& t, unsigned r) {
  return *reinterpret_cast<const uint2*>(&meta_ptr(t, r)->freq);
}

__device__ __forceinline__ Entry load_entry(const Entry* e) {
  const uint4 v = *reinterpret_cast<const uint4*>(e);
  Entry r;
  r.key = (long long)(((unsigned long long)v.y << 32) | v.x);
  r.row = v.z;
  r.hint = v.w;
  return r;
}

// read-only probe; 0 = absent (row 0 is the zero row)
__device__ __forceinline__ unsigned table_find(const TableDev& t, long long key) {
  if (key == EMPTY_KEY) {
    Entry e = load_entry(&t.entries[t.mask + 1]);
    return (e.key == 0 && e.row != ROW_TOMB) ? e.row : 0u;
  }
  unsigned long long p = mix64((unsigned long long)key) & t.mask;
  for (;;) {
    Entry e = load_entry(&t.entries[p]);
    if (e.key == key) return e.row != ROW_TOMB ? e.row : 0u;
    if (e.key == EMPTY_KEY) return 0u;
    p = (p + 1) & t.mask;
  }
}

// read-only probe continuing from an entry the caller already loaded (so the first loads of
// several tables can be in flight together)
__device__ __forceinline__ unsigned table_find_from(const TableDev& t, long long key, unsigned long long p,
                                                   Entry e, unsigned* hint = nullptr) {
  if (key == EMPTY_KEY) {
    if (hint) *hint = e.hint;
    return (e.key == 0 && e.row != ROW_TOMB) ? e.row : 0u;
  }
  for (;;) {
    if (e.key == key) {
      if (hint) *hint = e.hint;
      return e.row != ROW_TOMB ? e.row : 0u;
    }
    if (e.key == EMPTY_KEY) return 0u;
    p = (p + 1) & t.mask;
    e = load_entry(&t.entries[p]);
  }
}
// index entry of a key that is present (nullptr when absent): where its slot-row hint lives
__device__ __forceinline__ Entry* table_entry_of(const TableDev& t, long long key) {
  if (key == EMPTY_KEY) {
    Entry* s = &t.entries[t.mask + 1];
    return load_entry(s).key == 0 ? s : nullptr;
  }
  unsigned long long p = mix64((unsigned long long)key) & t.mask;
  for (;;) {
    Entry* s = &t.entries[p];
    const Entry e = load_entry(s);
    if (e.key == key) return s;
    if (e.key == EMPTY_KEY) return nullptr;
    p = (p + 1) & t.mask;
  }
}
__device__ __forceinline__ unsigned long long home_of(const TableDev& t, long long key, unsigned long long h) {
  return key == EMPTY_KEY ? t.mask + 1 : (h & t.mask);
}

// a batch that cannot be processed (see report_deferred_error in kvhip.hip): device flag for the kernels
// that follow in the same op, pinned host word for the next call
__device__ __forceinline__ void raise_error(const TableDev& t, unsigned code) {
  atomicExch(&t.counters[1], code);
  if (t.err_host) __hip_atomic_store(t.err_host, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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
    if (e.key == stored) {
      if (e.row != ROW_TOMB) return e.row;
      goto claimed;  // deleted earlier: the entry is still this key's, give it a row again
    }
  } else {
    stored = key;
    unsigned long long p = mix64((unsigned long long)key) & t.mask;
    for (;;) {
      slot = &t.entries[p];
      Entry e = load_entry(slot);
      if (e.key == key) {
        if (e.row != ROW_TOMB) return e.row;
        goto claimed;
      }
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
  unsigned r = 0;
  bool have = false;
  if (t.free_rows) {  // rows released by Delete first (pops only: Delete never runs beside an insert)
    const int f = atomicSub(reinterpret_cast<int*>(&t.counters[2]), 1);
    if (f > 0) { r = t.free_rows[f - 1]; have = true; }
    else atomicAdd(reinterpret_cast<int*>(&t.counters[2]), 1);
  }
  if (!have) {
    r = atomicAdd(&t.counters[0], 1u);
    if (r >= t.max_rows) {
      raise_error(t, 1u);
      slot->row = 0;
      return 0u;
    }
  }
  slot->row = r;
  *key_ptr(t, r) = key;
  { RowMeta* nm = meta_ptr(t, r); nm->delta = 0; nm->stamp = 0; }  // fresh (or recycled) row: in no delta list yet
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

// partition that owns a key (top bits of the hash; table probes use the low bits)
__device__ __forceinline__ unsigned part_of(long long key, int pshift) {
  return pshift >= 64 ? 0u : (unsigned)(mix64((unsigned long long)key) >> pshift);
}

// exclusive scan of one value per thread across the block (NW = waves per block);
// wtot: LDS scratch [NW + 1]; returns the exclusive prefix, *total = block sum.
// A barrier that orders LDS only.  __syncthreads() also waits for the wave's outstanding GLOBAL loads (its fence
// drains vmcnt), which exposes the latency of loads requested ahead of time (index probes, id loads) at the first
// barrier behind them.  Where every hand-over between the waves goes through LDS, this one lets them stay in flight.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LDSONLY: barriers that leave global loads in flight (lds_barrier)
template <int NW, bool LDSONLY = false>
__device__ __forceinline__ unsigned block_excl_scan(unsigned v, unsigned* wtot, unsigned* total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned x = __shfl_up(incl, o);
    if (lane >= o) incl += x;
  }
  if (LDSONLY) lds_barrier(); else __syncthreads();  // wtot may still be read from a previous scan
  if (lane == 63) wtot[wv] = incl;
  if (LDSONLY) lds_barrier(); else __syncthreads();
  unsigned base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const unsigned x = wtot[i];
    if (i < wv) base += x;
    tot += x;
  }
  *total = tot;
  return base + incl - v;
}

template <typename IdT>
__device__ __forceinline__ long long load_id(const IdT* ids, size_t i) {
  return (long long)ids[i];
}
// (id, occurrence count) pairs as the sharded exchange delivers them (kv_bucket_by_owner pairs_out)
struct IdCount { long long id, count; };
template <>
__device__ __forceinline__ long long load_id<IdCount>(const IdCount* ids, size_t i) { return ids[i].id; }


struct OptArgs {
  float lr, b1p, b2p, b1, b2, eps, l1, l2, l21, l2s, lr_power;
  float alpha, l21_norm;  // host-precomputed in fp32 exactly as the reference does
  int update_slots;
  int fast;               // row math on the hardware's 1-ulp v_sqrt_f32 / v_rcp_f32 (kv_set_fast_math; 0: IEEE sequences)
};

// sqrt and division of the row math.  fast: one v_sqrt_f32 / v_rcp_f32 (1 ulp each: the update stays within north_star's
// 1e-6 relative of the reference's correctly rounded Eigen arithmetic); else the IEEE sequences (bit-for-bit the
// oracle's results: the deterministic mode and every table that asks for it keep them).  `fast` is uniform over the launch.
#ifdef KV_FASTM_CONST
#define KV_FASTM(a) (KV_FASTM_CONST != 0)
#else
#define KV_FASTM(a) ((a).fast != 0)
#endif
__device__ __forceinline__ float kv_sqrt(float x, bool fast) { return fast ? __builtin_amdgcn_sqrtf(x) : sqrtf(x); }
__device__ __forceinline__ float kv_div(float x, float y, bool fast) { return fast ? x * __builtin_amdgcn_rcpf(y) : x / y; }

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
// same, for data this launch reads exactly once (gradient rows): streaming load, so the table rows
// and the partition lists keep the L2 / infinity cache
template <int V>
__device__ __forceinline__ void ldv_stream(const float* p, float (&o)[V]) {
#pragma unroll
  for (int c = 0; c < V; ++c) o[c] = __builtin_nontemporal_load(p + c);
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


// One unique key's fused optimizer update, executed by LPR cooperating lanes (lane = 0..LPR-1).
// `tag` = var row id | ROW_FILTERED; gv = the key's summed gradient, element e at lane
// (e / V) % LPR, step (e / V) / LPR.  All LPR lanes of every group in the wave must call it
// (shuffles inside); `live` masks groups without a key.  r0 / r1 (+ new flags) are the slot-table
// rows the group leader probed (probe_for_apply).
// Restates the per-id body of KvVariableGroupSparseApplyAdamV4Op / V3Op / SparseApplyAdagradOp /
// SparseGroupSparseApplyFtrlOp (training_ops.cc:7142-7197, 5871-5927, 1455-1486, 684-763).
// State rows the caller already holds (requested together with the gradient rows, so the update needs no
// second round trip): x = the var row; s[0..2] = the first slot table's row in blocks of dim floats
// (GroupAdam m | v | z; Adagrad / FTRL accum in s[0]).  have_x / have_s say which of them are valid.
template <int V, int K>
struct PreRows {
  float x[K][V], s[3][K][V];
};
// The row math and the stores of one key's update, on values the caller already holds: xin = the var row;
// sin[0..2] = GroupAdam m | v | z, Adagrad accum, FTRL accum | linear.  Element e of a row lives at lane
// (e / V) % LPR, step (e / V) / LPR.  All LPR lanes of every group of the wave call it (shuffles inside); `act`
// masks groups without an update.  fvp / f0p / f1p = the flag bytes of the var / first / second slot row.
// Restates the per-id body of KvVariableGroupSparseApplyAdamV4Op / V3Op / SparseApplyAdagradOp /
// SparseGroupSparseApplyFtrlOp (training_ops.cc:7142-7197, 5871-5927, 1455-1486, 684-763).
template <int OPT, int V, int LPR, int K>
__device__ __forceinline__ void opt_core(float* xrow, float* s0row, float* s1row, unsigned char* fvp, unsigned char* f0p,
                                         unsigned char* f1p, bool act, bool new0, int D, const float (&gv)[K][V],
                                         const OptArgs& a, int lane, const float (&xin)[K][V], const float (&sin)[3][K][V]) {
  if (OPT == OPT_ADAM_V4 || OPT == OPT_ADAM_V3) {
    // training_ops.cc:7166-7195 (V4) / :5895-5925 (V3); slot row = [m | v | z]
    float m[K][V], nv[K][V], sq[K][V], z[K][V], uu[K][V];
    float part = 0.f;
    const float omb1 = 1.f - a.b1, omb2 = 1.f - a.b2;
    const bool fm = KV_FASTM(a);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int e0 = (lane + k * LPR) * V;
      const bool valid = act && e0 < D;
#pragma unroll
      for (int c = 0; c < V; ++c) {
        const float xo = valid ? xin[k][c] : 0.f, mo = valid ? sin[0][k][c] : 0.f;
        const float vo = valid ? sin[1][k][c] : 0.f, zo = valid ? sin[2][k][c] : 0.f;
        const float gg = gv[k][c];
        const float mn = a.b1 * mo + omb1 * gg;
        const float vn = a.b2 * vo + omb2 * (gg * gg);
        const float s = kv_sqrt(vn, fm);
        float d;
        if (OPT == OPT_ADAM_V4) {
          d = (a.b1 > a.b1p) ? (s - kv_sqrt(vo, fm)) * xo : (s + a.eps) * xo;
        } else {
          d = (a.b1 > a.b1p) ? kv_div(s - kv_sqrt(vo, fm), a.lr, fm) * xo
                             : kv_div(s - kv_sqrt(vo, fm) + a.eps, a.lr, fm) * xo;
        }
        const float zn = zo + (a.alpha * mn - d);
        const float adj = fmaxf(fminf(zn, a.l1), -a.l1);
        const float uv = adj - zn;
        m[k][c] = mn; nv[k][c] = vn; sq[k][c] = s; z[k][c] = zn; uu[k][c] = uv;
        if (valid) part += uv * uv;
      }
    }
    const float norm = kv_sqrt(group_sum<LPR>(part), fm);
    const bool upd = norm > a.l21_norm;
    const float scale = 1.f - kv_div(a.l21_norm, norm, fm);
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
                                                 : kv_div(sq[k][c] + a.eps, a.lr, fm) + two_l2;
            xn[c] = kv_div(uu[k][c] * scale, y, fm);
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
      *fvp = (unsigned char)(upd ? (anyx ? 0u : FLAG_UNDER) : (FLAG_BLACK | FLAG_UNDER));
      *f0p = (unsigned char)(anys ? 0u : FLAG_UNDER);
    }
  } else if (OPT == OPT_ADAGRAD) {
    // training_ops.cc:1470-1482.  No CoverUpdate: flags of existing rows are left alone.
    bool sbig = false;
    const bool fm = KV_FASTM(a);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int e0 = (lane + k * LPR) * V;
      if (act && e0 < D) {
        float xo[V], acc[V];
#pragma unroll
        for (int c = 0; c < V; ++c) {
          xo[c] = xin[k][c]; acc[c] = sin[0][k][c];
          const float gg = gv[k][c];
          sbig |= fabsf(acc[c]) >= CUTOFF;
          if (a.update_slots) acc[c] = acc[c] + gg * gg;
          if (fm) xo[c] = xo[c] - (a.lr * gg) * __builtin_amdgcn_rsqf(acc[c]);
          else xo[c] = (D > 1) ? xo[c] - (a.lr * gg) * (1.f / sqrtf(acc[c]))
                               : xo[c] - (a.lr * gg) / sqrtf(acc[c]);
        }
        stv<V>(xrow + e0, xo);
        stv<V>(s0row + e0, acc);
      }
    }
    const bool anys = group_any<LPR>(sbig);
    if (act && lane == 0) {
      // the reference does not refresh under_threshold here; a later lookup does (FLAG_DIRTY)
      *fvp |= (unsigned char)FLAG_DIRTY;
      if (new0) *f0p = (unsigned char)((anys ? 0u : FLAG_UNDER) | (a.update_slots ? FLAG_DIRTY : 0u));
      else if (a.update_slots) *f0p |= (unsigned char)FLAG_DIRTY;
    }
  } else {
    // OPT_FTRL: training_ops.cc:713-751 with has_l2_shrinkage; slot 0 = accum, slot 1 = linear
    float x[K][V], ac[K][V], z[K][V], uu[K][V];
    float part = 0.f;
    const bool half = a.lr_power == -0.5f;
    const float two_l2s = 2.f * a.l2s;
    const bool fm = KV_FASTM(a);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int e0 = (lane + k * LPR) * V;
      const bool valid = act && e0 < D;
#pragma unroll
      for (int c = 0; c < V; ++c) {
        x[k][c] = valid ? xin[k][c] : 0.f; ac[k][c] = valid ? sin[0][k][c] : 0.f;
        const float zo = valid ? sin[1][k][c] : 0.f;
        const float xo = x[k][c], ao = ac[k][c];
        const float gs = gv[k][c] + two_l2s * xo;
        const float na = ao + gs * gs;
        const float pn = half ? kv_sqrt(na, fm) : powf(na, -a.lr_power);
        const float po = half ? kv_sqrt(ao, fm) : powf(ao, -a.lr_power);
        const float zn = zo + (gs - kv_div(pn - po, a.lr, fm) * xo);
        const float adj = fmaxf(fminf(zn, a.l1), -a.l1);
        const float uv = adj - zn;
        z[k][c] = zn; uu[k][c] = uv;
        if (valid) part += uv * uv;
      }
    }
    const float norm = kv_sqrt(group_sum<LPR>(part), fm);
    const bool upd = norm > a.l21_norm;
    const float scale = 1.f - kv_div(a.l21_norm, norm, fm);
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
          const float pn = half ? kv_sqrt(na, fm) : powf(na, -a.lr_power);
          xn[c] = 0.f;
          if (upd) xn[c] = kv_div(uu[k][c] * scale, kv_div(pn, a.lr, fm) + two_l2, fm);
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
      *fvp = (unsigned char)(upd ? (anyx ? 0u : FLAG_UNDER) : (FLAG_BLACK | FLAG_UNDER));
      *f0p = (unsigned char)(anya ? 0u : FLAG_UNDER);
      *f1p = (unsigned char)(anyz ? 0u : FLAG_UNDER);
    }
  }
}

// One unique key's fused optimizer update through the tables: resolves the row pointers, loads (or initialises, for
// a slot row inserted now) the state rows the caller does not hold yet, then opt_core.
template <int OPT, int V, int LPR, int K>
__device__ __forceinline__ void opt_update_row(const TableDev& tv, const TableDev& ts0,
                                               const TableDev& ts1, long long key, unsigned tag,
                                               unsigned r0, bool new0, unsigned r1, bool new1,
                                               bool live, const float (&gv)[K][V], const OptArgs& a,
                                               int lane, const PreRows<V, K>* pre = nullptr, bool have_x = false,
                                               bool have_s = false) {
  const int D = tv.dim;
  const bool skip = !live || (tag & ROW_FILTERED) || (tag & ROW_MASK) == 0u;  // training_ops.cc:7150-7152
  const unsigned rv = tag & ROW_MASK;
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
  float xin[K][V], sin[3][K][V];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int e0 = (lane + k * LPR) * V;
#pragma unroll
    for (int c = 0; c < V; ++c) xin[k][c] = sin[0][k][c] = sin[1][k][c] = sin[2][k][c] = 0.f;
    if (!(act && e0 < D)) continue;
    if (pre && have_x) {
#pragma unroll
      for (int c = 0; c < V; ++c) xin[k][c] = pre->x[k][c];
    } else {
      ldv<V>(xrow + e0, xin[k]);
    }
    if (pre && have_s && !new0) {
      constexpr int NS0 = (OPT == OPT_ADAM_V4 || OPT == OPT_ADAM_V3) ? 3 : 1;
#pragma unroll
      for (int b3 = 0; b3 < NS0; ++b3)
#pragma unroll
        for (int c = 0; c < V; ++c) sin[b3][k][c] = pre->s[b3][k][c];
    } else {
      ldslot<V>(s0row, ia0, ib0, new0, e0, sin[0][k]);
      if (OPT == OPT_ADAM_V4 || OPT == OPT_ADAM_V3) {
        ldslot<V>(s0row, ia0, ib0, new0, e0 + D, sin[1][k]);
        ldslot<V>(s0row, ia0, ib0, new0, e0 + 2 * D, sin[2][k]);
      }
    }
    if (OPT == OPT_FTRL) ldslot<V>(s1row, ia1, ib1, new1, e0, sin[1][k]);
  }
  opt_core<OPT, V, LPR, K>(xrow, s0row, s1row, flags_ptr(tv, act ? rv : 0u), flags_ptr(ts0, act ? r0 : 0u),
                           OPT == OPT_FTRL ? flags_ptr(ts1, act ? r1 : 0u) : nullptr, act, new0, D, gv, a, lane, xin, sin);
}
