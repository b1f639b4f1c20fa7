// kv_kernels.h — the batch pipeline (included by kvhip.hip and the kv_apply_*.hip units).
//
// Why this shape.  A device-scope atomic on ONE address costs ~40 ns on MI355X and same-address
// atomics serialise; a Zipf(1.2) batch of 1 M ids has ~100 keys that occur in (nearly) every tile, so
// a global scratch hash or fp32 atomic accumulation spends its time queueing on those addresses
// (measured in round 1).  This pipeline has NO global atomics on the data path.  It is built around
// ONE index of the batch — the input positions sorted by key — that the training lookup builds while
// the output rows are being copied, and that the optimizer apply of the same batch then consumes:
//
//   k_tile        one block (TBT threads) per TILE input positions: LDS hash dedup of the tile; the
//                 tile's unique keys ("entries") are counting-sorted by the key's hash partition and
//                 written with plain coalesced stores (ent_*), partition boundaries in toff[tile][0..P]
//                 (entries and positions).  Every input position learns its entry and its rank among the
//                 key's occurrences in the tile (slot_rank).
//   k_part_keys   one block per partition p: takes partition p's entries from EVERY tile, so it sees
//                 all occurrences of its keys: exact counts, exclusive ownership of the table rows (find /
//                 insert / frequency / flags).  It also lays the partition's keys out in the sorted
//                 position list: partition p starts at sum over tiles of toff[t][p].positions (no scan
//                 kernel, no atomics), a key at the block-scan prefix of the occurrence counts, an entry at
//                 its key's start plus the occurrences of the key's entries before it (ent_base); the key's
//                 record {key, row, slot-row hint, start, count} goes to the cold or the hot key list.
//   k_gather<ORDER> / k_order   order[ent_base[entry] + rank] = position, and (training lookup) the output
//                 rows: out[i] = rows[ent_b[entry]], one wave per 64 rows.
//   k_apply       wave-granular segmented sum of the gradient rows (a key's occurrences are contiguous)
//                 fused with the optimizer row update: hot keys in chunks of HC rows per wave, cold keys
//                 one per lane group (see there);
//   k_apply_fin   the hot keys that have more than one chunk: chunk sums added in chunk order, update.
//
// Every kernel body is a __device__ function with two entry points: one table (arguments by
// value) and many tables in one launch (grid.y = table, arguments from a MultiDesc array).
//
// Summation order of repeated ids: position-sorted inside a tile only in deterministic mode; see DESIGN.md.
#pragma once

// ------------------------------------------------------------------------------------------
// k_tile
// ------------------------------------------------------------------------------------------
// FIRST = false (index modes): ent_a = occurrences of the key in the tile | min(sum of per-occurrence
//                              counts, 65535) << 16                      (kv_variable.h:320-350)
// FIRST = true (scatter / mark): ent_a = one input position of the key in the tile
struct TileSmem {
  long long* lkeys;        // [LS + 1]   (slot LS: the key that equals EMPTY_KEY)
  unsigned* lcnt;          // [LS + 1]   occurrences of the slot's key
  unsigned short* lfirst;  // [LS + 1]   (FIRST) tile-local position of one occurrence
  unsigned short* lpos;    // [LS + 1]   entry position of the slot's key
  unsigned short* lwork;   // [TILE + 1] occupied slots
  unsigned* hist;          // [MAX_P + 1] per partition: entries (low 16) | positions (high 16); with
                           //            per-occurrence counts later reused as the entries' frequency sums
  unsigned* wtot;          // [8]
};
static_assert(MAX_P + 1 >= TILE + 1, "hist doubles as the per-entry frequency sums");

__host__ __device__ inline size_t tile_smem_bytes(bool first) {
  size_t b = (size_t)(LS + 1) * 8 + 16;        // lkeys
  b += (size_t)(LS + 1) * 4 + 16;              // lcnt
  b += (size_t)(LS + 1) * 2 + 16;              // lpos
  b += (size_t)(TILE + 1) * 2 + 16;            // lwork
  b += (size_t)(MAX_P + 1) * 4 + 16;           // hist
  b += 64;                                     // wtot
  if (first) b += (size_t)(LS + 1) * 2 + 16;   // lfirst
  return b;
}

template <bool FIRST>
__device__ __forceinline__ TileSmem carve_tile(char* base) {
  TileSmem s;
  auto take = [&](size_t bytes) { char* p = base; base += (bytes + 15) & ~(size_t)15; return p; };
  s.lkeys = reinterpret_cast<long long*>(take((size_t)(LS + 1) * 8));
  s.lcnt = reinterpret_cast<unsigned*>(take((size_t)(LS + 1) * 4));
  s.lpos = reinterpret_cast<unsigned short*>(take((size_t)(LS + 1) * 2));
  s.lwork = reinterpret_cast<unsigned short*>(take((size_t)(TILE + 1) * 2));
  s.hist = reinterpret_cast<unsigned*>(take((size_t)(MAX_P + 1) * 4));
  s.wtot = reinterpret_cast<unsigned*>(take(64));
  s.lfirst = FIRST ? reinterpret_cast<unsigned short*>(take((size_t)(LS + 1) * 2)) : nullptr;
  return s;
}

// det != 0: a position's rank among its key's occurrences in the tile follows the input order (the
// deterministic reduction mode); otherwise it is the arrival order of the LDS atomics.
template <bool FIRST, typename IdT>
__device__ __forceinline__ void tile_body(const WsDev& w, const IdT* __restrict__ ids,
                                          const int* __restrict__ counts, long long n, int det) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  TileSmem sm = carve_tile<FIRST>(smem_raw);
  __shared__ unsigned lnwork, lsent;

  const int tid = threadIdx.x;
  const unsigned tile = blockIdx.x;
  const long long base = (long long)tile * TILE;
  const unsigned P = w.P;
  constexpr bool PAIRS = std::is_same<IdT, IdCount>::value;
  const bool has_counts = !FIRST && (PAIRS || counts != nullptr);
  KV_STAMP(0);

  // the tile's ids (and counts) are requested first: their HBM latency runs under the LDS clearing
  long long kreg[IPT];
  unsigned creg[IPT];
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    const long long i = base + (long long)k * TBT + tid;
    kreg[k] = 0; creg[k] = 1;
    if (i < n) {
      kreg[k] = load_id(ids, (size_t)i);
      if constexpr (PAIRS) {
        const long long ci = ids[i].count;             // counts travel with the ids
        creg[k] = (unsigned)(unsigned short)(ci < 65535 ? ci : 65535);
      } else if (!FIRST && counts != nullptr) {
        // SaturateMaxFrequency(int32) -> uint16 (utility.h:57-59)
        const int ci = counts[i];
        creg[k] = (unsigned)(unsigned short)(ci < 65535 ? ci : 65535);
      }
    }
  }
  for (int s = tid; s <= LS; s += TBT) {
    sm.lkeys[s] = EMPTY_KEY;
    sm.lcnt[s] = 0;
  }
  for (unsigned p = tid; p <= P; p += TBT) sm.hist[p] = 0;
  if (tid == 0) { lnwork = 0; lsent = 0; }
  if (tile == 0 && tid < 8) w.ctr[tid] = 0;   // the partition pass counts into them
  if (w.zero_counts) {   // sparse unique numbers (sharded route): a count of 0 = "names no key"
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
      const long long i = base + (long long)k * TBT + tid;
      if (i < n) w.zero_counts[i] = 0;
    }
  }
  __syncthreads();

  // ---- phase 1: LDS hash insert of this tile's ids --------------------------------------
  unsigned tslot[IPT], myrank[IPT];
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    const long long i = base + (long long)k * TBT + tid;
    tslot[k] = 0xFFFFFFFFu;
    myrank[k] = 0;
    bool there = i < n;
    if constexpr (PAIRS) {
      // fixed-capacity exchange segments (k_owner_scatter_fixed): record 0 of a segment is its header {records, -},
      // records past the header's count are stale bytes of earlier batches; a count of 0 marks a record void too
      if (there && creg[k] == 0u) there = false;
      if (there && w.seg_cap) {
        const long long r = i % w.seg_cap;
        there = r >= 1 && r <= ids[i - r].id;
      }
    }
    if (there) {
      const long long key = kreg[k];
      unsigned h;
      if (key == EMPTY_KEY) {
        h = LS;
        if (atomicCAS(&lsent, 0u, 1u) == 0u && FIRST) sm.lfirst[LS] = (unsigned short)(k * TBT + tid);
      } else {
        h = (unsigned)(mix64((unsigned long long)key) >> 40) & (LS - 1);
        for (;;) {
          const unsigned long long old =
              atomicCAS(reinterpret_cast<unsigned long long*>(&sm.lkeys[h]),
                        (unsigned long long)EMPTY_KEY, (unsigned long long)key);
          if (old == (unsigned long long)EMPTY_KEY) {
            if (FIRST) sm.lfirst[h] = (unsigned short)(k * TBT + tid);
            break;
          }
          if (old == (unsigned long long)key) break;
          h = (h + 1) & (LS - 1);
        }
      }
      myrank[k] = atomicAdd(&sm.lcnt[h], 1u);
      tslot[k] = h;
    }
  }
  __syncthreads();
  KV_STAMP(1);

  // ---- phase 2: compact the occupied slots into a work list ------------------------------
  for (int s = tid; s < LS; s += TBT)
    if (sm.lkeys[s] != EMPTY_KEY) sm.lwork[atomicAdd(&lnwork, 1u)] = (unsigned short)s;
  if (tid == 0 && lsent) sm.lwork[atomicAdd(&lnwork, 1u)] = (unsigned short)LS;
  __syncthreads();
  const unsigned nwork = lnwork;

  // ---- phase 3: counting sort of the tile's unique keys by owning partition ---------------
  constexpr int WPT = (TILE + 1 + TBT - 1) / TBT;  // work items per thread (5)
  unsigned wp[WPT], wr[WPT];
#pragma unroll
  for (int q = 0; q < WPT; ++q) {
    const unsigned wi = tid + q * TBT;
    wp[q] = 0; wr[q] = 0;
    if (wi < nwork) {
      const unsigned s = sm.lwork[wi];
      const long long key = (s == LS) ? EMPTY_KEY : sm.lkeys[s];
      wp[q] = part_of(key, w.pshift);
      // one entry and lcnt positions for the partition: both halves of the word in one atomic
      wr[q] = atomicAdd(&sm.hist[wp[q]], 1u | (sm.lcnt[s] << 16)) & 0xFFFFu;
    }
  }
  __syncthreads();
  {
    const unsigned per = (P + TBT - 1) / TBT;
    const unsigned p0 = tid * per, p1 = min(p0 + per, P);
    unsigned sum = 0;
    for (unsigned p = p0; p < p1; ++p) sum += sm.hist[p];
    unsigned tot;
    unsigned run = block_excl_scan<TBT / 64>(sum, sm.wtot, &tot);
    for (unsigned p = p0; p < p1; ++p) { const unsigned c = sm.hist[p]; sm.hist[p] = run; run += c; }
    if (tid == 0) sm.hist[P] = tot;
  }
  __syncthreads();
  unsigned* toff = w.toff + (size_t)tile * (P + 1);
  for (unsigned p = tid; p <= P; p += TBT) toff[p] = sm.hist[p];
  unsigned wpos[WPT];
  unsigned short wslot[WPT];
#pragma unroll
  for (int q = 0; q < WPT; ++q) {
    wpos[q] = 0xFFFFFFFFu; wslot[q] = 0;
    const unsigned wi = tid + q * TBT;
    if (wi < nwork) {
      const unsigned s = sm.lwork[wi];
      const long long key = (s == LS) ? EMPTY_KEY : sm.lkeys[s];
      const unsigned pos = (sm.hist[wp[q]] & 0xFFFFu) + wr[q];
      const size_t e = (size_t)tile * TILE + pos;
      wpos[q] = pos; wslot[q] = (unsigned short)s;
      sm.lpos[s] = (unsigned short)pos;
      w.ent_key[e] = key;
      if (FIRST) {
        w.ent_a[e] = (unsigned)base + sm.lfirst[s];
      } else if (!has_counts) {
        const unsigned c = sm.lcnt[s];   // <= TILE: the frequency count equals the occurrences
        w.ent_a[e] = c | (c << 16);
      }
    }
  }
  __syncthreads();
  KV_STAMP(2);

  // ---- phase 4: every input position learns its key's entry and its rank in the tile -----------
  if (det) {
    // rank = occurrences of the key at smaller input positions.  Positions are taken in input order: round k
    // holds positions k * TBT .. + TBT - 1, one per thread, waves in order.  Inside a wave the lanes that hold
    // the same entry find each other with one ballot per bit of the entry number (a match-any); across waves
    // and rounds a running count per entry (in lwork's place: the entries are written, lwork is dead) is read
    // and advanced wave by wave.
    unsigned short* run = sm.lwork;
    for (int e = tid; e <= TILE; e += TBT) run[e] = 0;
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
      const int lp = k * TBT + tid;
      const bool valid = base + lp < n && tslot[k] != 0xFFFFFFFFu;
      const unsigned e = valid ? (unsigned)sm.lpos[tslot[k]] : 0u;
      unsigned long long mask = __ballot(valid);
#pragma unroll
      for (int b = 0; b < 11; ++b) {   // entry numbers are below TILE = 2^11
        const bool bit = (e >> b) & 1u;
        const unsigned long long bal = __ballot(bit);
        mask &= bit ? bal : ~bal;
      }
      const unsigned rw = (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
      const unsigned cnt = (unsigned)__popcll(mask);
      unsigned before = 0;
      for (int wv = 0; wv < TBT / 64; ++wv) {   // block-uniform
        if (wave == wv && valid) {
          before = run[e];
          if (rw == 0u) run[e] = (unsigned short)(before + cnt);   // the group's first lane advances the count
        }
        __syncthreads();
      }
      myrank[k] = valid ? before + rw : 0u;
    }
  }
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    const long long i = base + (long long)k * TBT + tid;
    if (i < n) w.slot_rank[i] = tslot[k] == 0xFFFFFFFFu ? 0xFFFFFFFFu : ((tile * TILE + sm.lpos[tslot[k]]) | (myrank[k] << RANK_SHIFT));
  }
  // ---- per-occurrence counts: frequency sum per entry (hist is dead: reused as lfreq) -------------
  if (has_counts) {
    __syncthreads();
    for (unsigned e = tid; e <= (unsigned)TILE; e += TBT) sm.hist[e] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < IPT; ++k)
      if (tslot[k] != 0xFFFFFFFFu) atomicAdd(&sm.hist[sm.lpos[tslot[k]]], creg[k]);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < WPT; ++q) {
      if (wpos[q] == 0xFFFFFFFFu) continue;
      const unsigned s = wslot[q];
      const unsigned f = sm.hist[wpos[q]];
      // saturating add is order independent: clamp early
      w.ent_a[(size_t)tile * TILE + wpos[q]] = sm.lcnt[s] | ((f > 65535u ? 65535u : f) << 16);
    }
  }
  KV_STAMP(3);
}

// One op on one table (arguments by value) or the same op on many tables in one launch
// (blockIdx.y = table; arguments from a descriptor array in device memory, see MultiDesc below).
struct MultiDesc;
template <bool FIRST, typename IdT>
__global__ void __launch_bounds__(TBT) k_tile(WsDev w, const IdT* __restrict__ ids,
                                             const int* __restrict__ counts, long long n, int det) {
  tile_body<FIRST, IdT>(w, ids, counts, n, det);
}

// ------------------------------------------------------------------------------------------
// partition pass
// ------------------------------------------------------------------------------------------
struct PartArgs {
  TableDev tv, ts0, ts1;      // var table; optimizer slot tables (apply)
  OptArgs opt;
  const float* grad;          // apply / dedup: input gradient rows; scatter: update rows
  unsigned day;
  int scatter_op, is_insert;  // MODE_SCATTER
  int mark_what;              // MODE_MARK: 0 = blacklist, 1 = frequency words (in fvals)
  const unsigned* fvals;
  long long* out_keys;        // MODE_UNIQUE
  float* out_sum;             // MODE_DEDUP fold: out_sum[row] = the key's sum
  const int* out_map;         // ... or out_sum[out_map[row]] when given
  int* out_counts;            // MODE_UNIQUE: occurrences (saturating) of each unique key
  int count_once;             // MODE_LOOKUP: frequency += 1 per unique key instead of per occurrence
  long long direct_rows;      // MODE_UNIQUE, > 0: keys ARE output row indices in [0, direct_rows)
                              // (tf.unsorted_segment_sum): no key list, no counter
  int use_hints;              // apply: ts0 is tv's attached slot table (Entry::hint names ts0's rows)
  int fold_op;                // MODE_DEDUP fold: KV_SCATTER_ADD (sum) / MUL (product) / MIN / MAX / ASSIGN (last)
  int det;                    // deterministic reduction mode
  int sparse_unique;          // MODE_UNIQUE: unique numbers = sorted position of the partition + local number (with gaps)
  long long n;                // ids in the batch
  const float* epart;         // entry-list pipeline: list words tagged EP_TAG name rows of this array (tile sums), else of grad
  unsigned day_lk;            // k_papply (kv_papply.h), PA_LOOKUP: the day stamp of the lookup whose bookkeeping it completes
  // k_papply PA_UNIQUE with route_world > 0 (sharded lookup route): every distinct id goes straight to its owner's segment
  int route_world, route_rule;     // owner_rank(id, world, rule)
  unsigned route_C;                // records per segment (header not counted)
  long long* route_seg;            // [world][C + 1][2] (id, count) records
  int* route_slot_of;              // [number] the record the id went to (0: no room)
  unsigned* route_overflow;        // pinned flag: a segment was too small
  unsigned* route_gcount;          // [MAXW + 1] records per owner so far; [MAXW]: blocks of the launch that are done
  unsigned* route_need;            // != nullptr: the LAST block of the launch writes the segments' headers, the largest segment
                                   // wanted (here) and clears the counters — what k_seg_headers_take does in its own launch
  unsigned* route_uhint;           // pinned host word (may be null): the batch's distinct ids
  unsigned uniq_serial;            // k_uapply (kv_uapply.h): this launch's stamp (1 .. 65535)
  int use_mirror;                  // the lean update reads / writes the slot row's frequency word and flags in the var row's
  unsigned mirror_epoch;           // SlotMirror (kv_device.h) when it is valid for this epoch; the host flushes (kvhip.hip mirror_*)
  int dd_number;                   // k_papply PA_DEDUP: the pass numbers the distinct ids itself (dense, ctr[0]; out_keys[number] = id) —
                                   // kv_dedup_segment_sum in one partition pass instead of PA_UNIQUE's and then this one
};

// round r of R keeps the keys whose sub-hash selects it (R = 1: everything)
__device__ __forceinline__ bool in_round(long long key, unsigned R, unsigned round) {
  return R == 1 || ((mix64((unsigned long long)key) >> 20) & (R - 1)) == round;
}

// LDS hash of the partition's unique keys: 64-bit CAS on the key, slot HSL = the EMPTY_KEY key.
// Returns the slot; *first = this call inserted the key.
template <int HSL>
__device__ __forceinline__ unsigned lds_key_slot(long long* hkey, unsigned* sent, long long key,
                                                 bool insert, bool* first) {
  *first = false;
  if (key == EMPTY_KEY) {
    if (insert) *first = atomicCAS(sent, 0u, 1u) == 0u;
    return HSL;
  }
  unsigned h = (unsigned)(mix64((unsigned long long)key) >> 8) & (HSL - 1);
  for (;;) {
    if (insert) {
      const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&hkey[h]),
                                               (unsigned long long)EMPTY_KEY, (unsigned long long)key);
      if (old == (unsigned long long)EMPTY_KEY) { *first = true; return h; }
      if (old == (unsigned long long)key) return h;
    } else if (hkey[h] == key) {
      return h;
    }
    h = (h + 1) & (HSL - 1);
  }
}

// Partition of a workgroup.  Workgroups go round-robin over the 8 XCDs (workgroup b runs on XCD
// b % 8), each with its own L2, and a partition's entries are short runs (~1 entry per tile) inside
// lines that hold the runs of ~20 neighbouring partitions: with p = b every XCD ends up fetching
// nearly every line of the entry arrays (8x the bytes).  So an XCD takes a CONTIGUOUS range of
// partitions; P/8 = 128 partitions are resident together on its 32 CUs, so each line is fetched once.
__device__ __forceinline__ unsigned xcd_partition(unsigned b, unsigned P) {
  return P >= 8u ? (b & 7u) * (P >> 3) + (b >> 3) : b;
}

// ---- k_part_keys ---------------------------------------------------------------------------------
// Streams the partition's entries twice and keeps only the unique keys in LDS, so a key that
// occurs in every tile costs nothing extra.  256 threads, ~30 KB LDS, so every block of a
// 1024-partition launch is resident at once (4 per CU needs < 40 KB).
constexpr int TBK = 256;

// per-partition segment directory in LDS: tpre[t] = entries of tiles < t, tstart[t] = first entry of
// tile t's segment.  Entry x of the partition (0 <= x < E) lives at tile t = last tpre[t] <= x.
// Filled by seg_directory(); lets every thread take entries x = tid, tid + T, ... whatever the number
// of tiles (one tile with 1000 entries is as parallel as 1000 tiles with one entry).
// *pbase = input positions that belong to partitions < p = where partition p starts in the sorted list.
template <int T, int NW>
__device__ __forceinline__ unsigned seg_directory(const WsDev& w, unsigned p, unsigned short* tpre,
                                                  unsigned short* tstart, unsigned* wtot, unsigned* pbase) {
  const unsigned NT = w.ntiles, P = w.P;
  const unsigned per = (NT + T - 1) / T;
  const unsigned t0 = min(NT, threadIdx.x * per), t1 = min(NT, t0 + per);
  unsigned sum = 0, psum = 0;
  for (unsigned t = t0; t < t1; ++t) {
    const unsigned* to = w.toff + (size_t)t * (P + 1) + p;
    const unsigned a = to[0], b = to[1];
    const unsigned s0 = a & 0xFFFFu, s1 = b & 0xFFFFu;
    tstart[t] = (unsigned short)s0;
    tpre[t] = (unsigned short)(s1 - s0);  // length for now
    sum += s1 - s0;
    psum += a >> 16;
  }
  unsigned E, PB;
  block_excl_scan<NW>(psum, wtot, &PB);
  unsigned run = block_excl_scan<NW>(sum, wtot, &E);
  for (unsigned t = t0; t < t1; ++t) { const unsigned len = tpre[t]; tpre[t] = (unsigned short)min(run, 65535u); run += len; }
  __syncthreads();
  *pbase = PB;
  return E;
}
__device__ __forceinline__ size_t seg_entry(const unsigned short* tpre, const unsigned short* tstart,
                                            unsigned NT, unsigned x) {
  unsigned lo = 0, hi = NT;  // last t with tpre[t] <= x
  while (hi - lo > 1) {
    const unsigned mid = (lo + hi) >> 1;
    if (tpre[mid] <= x) lo = mid; else hi = mid;
  }
  return (size_t)lo * TILE + tstart[lo] + (x - tpre[lo]);
}

// lanes that share one row in the apply kernels, by dim (kv_apply_launch.h dispatches on the same table);
// 64 / lanes keys ride side by side in a wave = the cold batch the partition pass forms
__host__ __device__ inline int apply_lanes(int D) {
  if ((D & 3) == 0) {
    const int q = D / 4;
    return q <= 1 ? 1 : q <= 2 ? 2 : q <= 4 ? 4 : q <= 16 ? 8 : q <= 32 ? 16 : 64;
  }
  return D <= 1 ? 1 : D <= 2 ? 2 : D <= 4 ? 4 : D <= 8 ? 8 : D <= 16 ? 16 : D <= 32 ? 32 : 64;
}

// chunk geometry of a hot key: HC rows per chunk — about what a batch of cold keys costs, so every work item
// of the apply weighs the same and a round-robin hand-out is balanced.  Occurrence order (det == 2): one chunk, the
// whole key — a sum taken one row at a time in list order is one chain.
__host__ __device__ inline unsigned chunk_rows(unsigned cnt, int det) { return det == 2 ? cnt : (unsigned)HC; }

// HSK_: LDS hash slots (the lookup that runs beside the probing gather takes 512 to leave LDS for the
// gather blocks; more unique keys than 3/4 of the slots split into sub-hash classes either way)
template <int MODE, int HSK_ = 1024, int EB_ = 8>
__device__ __forceinline__ void part_keys_body(const WsDev& w, const PartArgs& a) {
  constexpr int HSK = HSK_;
  // unique keys per class: the check `lnu >= UCAPK` in pass 1 races with up to TBK inserts, so the LDS hash
  // can receive UCAPK + TBK keys — which must still be fewer than its slots (or the probe loop never ends)
  constexpr int UCAPK = HSK - TBK;
  static_assert(UCAPK > 0 && UCAPK + TBK <= HSK, "LDS key hash would overflow");
  constexpr bool ORD = (MODE == MODE_LOOKUP || MODE == MODE_UNIQUE || MODE == MODE_APPLYIDX);
  constexpr bool CNT = (MODE == MODE_LOOKUP || MODE == MODE_UNIQUE);
  __shared__ long long hkey[HSK + 1];
  __shared__ unsigned hval[HSK + 1];   // lookup / unique: summed frequency count; scatter / mark: an input position
  __shared__ unsigned hrow[HSK + 1];   // row id of the key (bit 31: inserted now); unique: dense index
  __shared__ unsigned hocc[ORD ? HSK + 1 : 1];   // occurrences of the key, then its start in the sorted list
  __shared__ unsigned hrun[ORD ? HSK + 1 : 1];   // positions handed to the key's entries so far (pass 2)
  __shared__ unsigned short lnew[UCAPK + 8];  // slots whose row was inserted now / needs a row scan
  __shared__ unsigned short ulist[UCAPK + 8]; // slots of the unique keys, in order of first sight
  __shared__ unsigned lnu, lsent, lnnew, lpcur;
  __shared__ unsigned wtot[8];
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  unsigned short* tpre = reinterpret_cast<unsigned short*>(smem_raw);
  unsigned short* tstart = tpre + w.ntiles;

  const int tid = threadIdx.x;
  const unsigned p = xcd_partition(blockIdx.x, w.P);
  const unsigned NT = w.ntiles;
  const int D = a.tv.dim;
  KV_STAMPP(0);
  unsigned pbase;
  const unsigned E = seg_directory<TBK, TBK / 64>(w, p, tpre, tstart, wtot, &pbase);
  // The partition's key records and work items live in ITS stretch of the arrays — index pbase + local number
  // (a partition has at least as many positions as keys, chunks or items), so nothing is counted through global
  // atomics; pmeta[p] tells the next kernel how many there are and where.
  __shared__ unsigned lcold, lhot, lchunk, lnbig;
  __shared__ unsigned lbig[16][3];   // keys of the class with more than 16 chunks: {hot list index, first chunk, chunks}
  if (ORD && E == 0) { if (tid == 0) w.pmeta[p] = make_uint4(0u, 0u, pbase, 0u); }
  if (E == 0) return;
  const bool wide = E > 65535u;  // needs a key set crafted against the partition hash; guarded, not handled

  // work list of (R, round) sub-hash classes; an overflowing class is split in two and each
  // class is processed exactly once (block-uniform control flow)
  __shared__ unsigned stkR[24], stkr[24];
  __shared__ int sp;
  if (tid == 0) { stkR[0] = 1; stkr[0] = 0; sp = 1; lpcur = pbase; lcold = 0; lhot = 0; lchunk = 0; }
  __syncthreads();
  if (wide) {  // never silent: every later kernel of this op sees the flag and does nothing; the next call reports it
    if (tid == 0) { raise_error(a.tv, 2u); if (ORD) w.pmeta[p] = make_uint4(0u, 0u, pbase, 0u); }
    return;
  }
  while (sp > 0) {
    const unsigned R = stkR[sp - 1], round = stkr[sp - 1];
    __syncthreads();
    if (tid == 0) --sp;
    for (int s = tid; s <= HSK; s += TBK) {
      hkey[s] = EMPTY_KEY; hval[s] = 0;
      if constexpr (ORD) { hocc[s] = 0; hrun[s] = 0; }
    }
    if (tid == 0) { lnu = 0; lsent = 0; lnnew = 0; lnbig = 0; }
    __syncthreads();
    // ---- pass 1: unique keys of the partition + their summed counts ---------------------------
    // entries are taken EB per thread at a time with all their global loads in flight together (a
    // partition that holds a key present in every tile has ~5x the median number of entries); the
    // slot and position of the first EB entries stay in registers for pass 2
    constexpr int EB = EB_;   // entries per thread in flight (the lean variant beside the gather takes 2: registers)
    unsigned cge[EB];
    unsigned short cslot[EB], cocc[EB];
    const bool cached = (R == 1 && E <= (unsigned)(EB * TBK));
    for (unsigned x0 = 0; x0 < E; x0 += EB * TBK) {
      unsigned ge[EB];
      long long key[EB];
      unsigned ea[EB];
#pragma unroll
      for (int k = 0; k < EB; ++k) {
        const unsigned x = x0 + k * TBK + tid;
        ge[k] = x < E ? (unsigned)seg_entry(tpre, tstart, NT, x) : 0xFFFFFFFFu;
      }
#pragma unroll
      for (int k = 0; k < EB; ++k) {
        key[k] = 0; ea[k] = 0;
        if (ge[k] != 0xFFFFFFFFu) { key[k] = w.ent_key[ge[k]]; ea[k] = w.ent_a[ge[k]]; }
      }
#pragma unroll
      for (int k = 0; k < EB; ++k) {
        if (x0 == 0) { cge[k] = ge[k]; cslot[k] = 0; cocc[k] = 0; }
        if (ge[k] == 0xFFFFFFFFu || !in_round(key[k], R, round)) continue;
        if (lnu >= (unsigned)UCAPK) continue;  // overflow: this class is split below
        bool first;
        const unsigned h = lds_key_slot<HSK>(hkey, &lsent, key[k], true, &first);
        if (first) {
          const unsigned u = atomicAdd(&lnu, 1u);
          if (u < (unsigned)UCAPK) ulist[u] = (unsigned short)h;
        }
        if (CNT) atomicAdd(&hval[h], ea[k] >> 16);
        else if (!ORD && first) hval[h] = ea[k];
        if constexpr (ORD) atomicAdd(&hocc[h], ea[k] & 0xFFFFu);
        if (x0 == 0) { cslot[k] = (unsigned short)h; cocc[k] = (unsigned short)(ea[k] & 0xFFFFu); }
      }
    }
    __syncthreads();
    if (lnu >= (unsigned)UCAPK) {  // block-uniform: too many keys -> split the class, nothing applied yet
      __syncthreads();
      if (tid == 0) {
        if (sp + 2 <= 24) {
          stkR[sp] = 2 * R; stkr[sp] = round; ++sp;
          stkR[sp] = 2 * R; stkr[sp] = round + R; ++sp;
        } else {
          raise_error(a.tv, 2u);   // keys that no sub-hash separates: reported, not applied
        }
      }
      __syncthreads();
      continue;
    }
    KV_STAMPP(1);
    const unsigned nu = lnu;

    // ---- the class's keys in the sorted position list: start = partition start + keys of the classes
    //      before + occurrences of the keys before it in this class (block scan over the unique list);
    //      the same scan numbers the cold and the hot keys and the hot keys' chunks.  Thread tid owns
    //      keys tid * PERU .. + PERU - 1 of the unique list from here on.
    constexpr int PERU = (UCAPK + TBK - 1) / TBK;
    unsigned kst[PERU], kcnt[PERU], krank[PERU], kchunk[PERU];
    if constexpr (ORD) {
      unsigned sum = 0, ch = 0, nchs = 0;
#pragma unroll
      for (int q = 0; q < PERU; ++q) {
        const unsigned u = tid * PERU + q;
        kcnt[q] = u < nu ? hocc[ulist[u]] : 0u;
        sum += kcnt[q];
        if (u < nu) {
          if (kcnt[q] <= (unsigned)LCOLD) ch += 1u;
          else { ch += 1u << 16; nchs += (kcnt[q] + chunk_rows(kcnt[q], a.det) - 1u) / chunk_rows(kcnt[q], a.det); }
        }
      }
      const unsigned cur = lpcur;
      unsigned tot, chtot, ntot;
      unsigned run = cur + block_excl_scan<TBK / 64>(sum, wtot, &tot);
      unsigned chrun = block_excl_scan<TBK / 64>(ch, wtot, &chtot);
      unsigned nrun = block_excl_scan<TBK / 64>(nchs, wtot, &ntot);
      const unsigned c0 = lcold, h0 = lhot, k0 = lchunk;   // the classes before this one
      const unsigned cbase = pbase + c0, hbase = pbase + h0, kbase = k0;   // kbase: the partition's own chunk numbers
#pragma unroll
      for (int q = 0; q < PERU; ++q) {
        const unsigned u = tid * PERU + q;
        kst[q] = run; krank[q] = 0; kchunk[q] = 0;
        if (u < nu) {
          hocc[ulist[u]] = run; run += kcnt[q];
          if (kcnt[q] <= (unsigned)LCOLD) { krank[q] = cbase + (chrun & 0xFFFFu); chrun += 1u; }
          else {
            krank[q] = hbase + (chrun >> 16); chrun += 1u << 16;
            kchunk[q] = kbase + nrun; nrun += (kcnt[q] + chunk_rows(kcnt[q], a.det) - 1u) / chunk_rows(kcnt[q], a.det);
          }
        }
      }
      __syncthreads();
      if (tid == 0) { lpcur = cur + tot; lcold = c0 + (chtot & 0xFFFFu); lhot = h0 + (chtot >> 16); lchunk = k0 + ntot; }
    }
    // the key's record for the apply: cold or hot list, and the hot key's chunk items
    auto put_rec = [&](int q, long long key, unsigned roww, unsigned hint) {
      const uint4 ra = make_uint4((unsigned)key, (unsigned)((unsigned long long)key >> 32), roww, hint);
      if (kcnt[q] <= (unsigned)LCOLD) {
        w.coldlist[2 * (size_t)krank[q]] = ra;
        w.coldlist[2 * (size_t)krank[q] + 1] = make_uint4(kst[q], kcnt[q], 0u, 0u);
      } else {
        const unsigned cr = chunk_rows(kcnt[q], a.det);
        w.hotlist[2 * (size_t)krank[q]] = ra;
        w.hotlist[2 * (size_t)krank[q] + 1] = make_uint4(kst[q], kcnt[q], kchunk[q], cr);
        const unsigned nch = (kcnt[q] + cr - 1u) / cr;
        // one work item per chunk: {hot list index | HEAD_BIT, chunk in the key, the partition's chunk number};
        // a key with many chunks leaves them to the whole block (lbig, below)
        if (nch <= 16u) {
          for (unsigned i = 0; i < nch; ++i) w.litem[pbase + kchunk[q] + i] = make_uint4(krank[q] | HEAD_BIT, i, kchunk[q] + i, 0u);
        } else {
          const unsigned b = atomicAdd(&lnbig, 1u);
          if (b < 16u) { lbig[b][0] = krank[q]; lbig[b][1] = kchunk[q]; lbig[b][2] = nch; }
          else for (unsigned i = 0; i < nch; ++i) w.litem[pbase + kchunk[q] + i] = make_uint4(krank[q] | HEAD_BIT, i, kchunk[q] + i, 0u);
        }
      }
    };

    // ---- owner work: one thread per unique key ------------------------------------------------
    if constexpr (MODE == MODE_UNIQUE) {
      // tf.unique_with_counts: dense index = block base (ONE global atomic per block) + local rank;
      // direct_rows: the keys are the output rows themselves (tf.unsorted_segment_sum)
      // ... or, sparse_unique (the sharded route: nobody needs the list to be dense), the partition's first
      // sorted position + the keys of its earlier classes: no atomic at all (1024 returning atomics on one
      // address cost this kernel 9 of its 29 us); the gaps keep the count 0 the caller cleared the array to
      __shared__ unsigned lbase;
      if (a.direct_rows == 0) {
        if (tid == 0) lbase = a.sparse_unique ? pbase + lcold + lhot - nu : atomicAdd(&w.ctr[0], nu);
        __syncthreads();
      }
#pragma unroll
      for (int q = 0; q < PERU; ++q) {
        const unsigned u = tid * PERU + q;
        if (u >= nu) continue;
        const unsigned s = ulist[u];
        const long long key = (s == HSK) ? EMPTY_KEY : hkey[s];
        unsigned dense;
        if (a.direct_rows > 0) {
          dense = (key >= 0 && key < a.direct_rows) ? (unsigned)key : ROW_MASK;   // out of range: dropped
        } else {
          dense = lbase + u;
          a.out_keys[dense] = key;
          if (a.out_counts) a.out_counts[dense] = (int)(hval[s] > 65535u ? 65535u : hval[s]);
        }
        hrow[s] = dense;
        put_rec(q, key, dense, 0u);
      }
      __syncthreads();
    }
    // one thread per unique key, all keys of the partition at once: probe, then frequency word and
    // flags with a single load (RowMeta).  A thread that owns several keys (more than 256 distinct
    // keys in the partition: low-skew batches) takes them OB at a time with their probes in flight
    // together, so it pays the two dependent hops once per batch, not once per key.
    constexpr int OB = PERU;
    if (MODE != MODE_UNIQUE) {
      unsigned sl[OB], r[OB], hint[OB];
      long long key[OB];
      unsigned long long pp[OB];
      Entry en[OB];
      bool isnew[OB];
#pragma unroll
      for (int k = 0; k < OB; ++k) {
        const unsigned u = tid * PERU + k;
        sl[k] = 0xFFFFFFFFu; r[k] = 0; isnew[k] = false; key[k] = 0; pp[k] = 0; hint[k] = 0;
        if (u < nu) {
          sl[k] = ulist[u];
          key[k] = (sl[k] == HSK) ? EMPTY_KEY : hkey[sl[k]];
          pp[k] = home_of(a.tv, key[k], mix64((unsigned long long)key[k]));
          en[k] = load_entry(&a.tv.entries[pp[k]]);
        }
      }
#pragma unroll
      for (int k = 0; k < OB; ++k) {
        if (sl[k] == 0xFFFFFFFFu) continue;
        // warm path: a read-only probe; the insert (atomics, row allocation) is a cold branch.
        // import frequency words only touch keys that exist (dynamic_restore.hpp:232-246)
        r[k] = table_find_from(a.tv, key[k], pp[k], en[k], &hint[k]);
        if (__builtin_expect(r[k] == 0u, 0) && !(MODE == MODE_MARK && a.mark_what == 1)) {
          r[k] = table_find_or_insert(a.tv, key[k], &isnew[k]);
          if (isnew[k]) hint[k] = 0;
        }
        hrow[sl[k]] = r[k] | (isnew[k] ? 0x80000000u : 0u);
        // bit 31 of the row word: the OPTIMIZER's index pass inserted the key (such a row is neither filtered
        // by enter_threshold nor counted, FindOrInsertWithFnUnsafe returns false: table_manager.h:192-204)
        if constexpr (ORD) put_rec(k, key[k], r[k] | ((MODE == MODE_APPLYIDX && isnew[k]) ? 0x80000000u : 0u), hint[k]);
      }
      uint2 m[OB];
#pragma unroll
      for (int k = 0; k < OB; ++k) {
        m[k] = make_uint2(0u, (unsigned)FLAG_DIRTY);
        if (MODE == MODE_LOOKUP && sl[k] != 0xFFFFFFFFu && r[k] != 0u && !isnew[k]) m[k] = load_freq_flags(a.tv, r[k]);
      }
#pragma unroll
      for (int k = 0; k < OB; ++k) {
        if (sl[k] == 0xFFFFFFFFu || r[k] == 0u) continue;
        const unsigned s = sl[k];
        RowMeta* mp = meta_ptr(a.tv, r[k]);
        // FindOrInsert / ScatterUpdate / InsertOrUpdate remember every key they see (kv_variable.h:316,451,685);
        // the import paths (is_insert 2, 3) and the blacklist / frequency marks do not
        if (MODE == MODE_LOOKUP || (MODE == MODE_SCATTER && a.is_insert < 2)) mark_delta(a.tv, r[k]);
        if (MODE == MODE_LOOKUP) {
          // find_func / insert_func (kv_variable.h:320-363): lo16 = sat_add(lo16, batch count),
          // hi16 = today; UpdateUnderThreshold only has work to do when the row changed since the
          // flag was computed (FLAG_DIRTY) or the row is new — every other writer keeps it current
          const unsigned cnt = a.count_once ? 1u : hval[s];
          unsigned lo = (m[k].x & 0xFFFFu) + (cnt > 65535u ? 65535u : cnt);
          if (lo > 65535u) lo = 65535u;
          mp->freq = (a.day << 16) | lo;
          if (isnew[k]) mp->flags = (unsigned char)FLAG_DIRTY;
          if (m[k].y & FLAG_DIRTY) lnew[atomicAdd(&lnnew, 1u)] = (unsigned short)(s | (isnew[k] ? 0x8000u : 0u));
        } else if (MODE == MODE_APPLYIDX) {
          // FindOrInsertUnsafe (kv_variable.h:382-416): a key the optimizer meets first gets a row from the
          // init rule, frequency word 1 (EmbeddingValue ctor, table_manager.h:94); existing rows are not touched
          if (isnew[k]) {
            mp->freq = 1u; mp->flags = 0;
            lnew[atomicAdd(&lnnew, 1u)] = (unsigned short)(s | 0x8000u);
          }
        } else {
          if (isnew[k]) { mp->freq = 1u; mp->flags = 0; }  // EmbeddingValue ctor: freq_val 1, day 0 (table_manager.h:94)
          lnew[atomicAdd(&lnnew, 1u)] = (unsigned short)(s | (isnew[k] ? 0x8000u : 0u));
        }
      }
    }
    if constexpr (ORD) {
      __syncthreads();
      const unsigned nbig = min(lnbig, 16u);
      for (unsigned b = 0; b < nbig; ++b)   // block-uniform
        for (unsigned i = tid; i < lbig[b][2]; i += TBK)
          w.litem[pbase + lbig[b][1] + i] = make_uint4(lbig[b][0] | HEAD_BIT, i, lbig[b][1] + i, 0u);
    }
    // hval (the frequency sums) has been consumed by its key's owner: it now carries the key's record word
    // to pass 2 (list index, bit 31 = hot list)
    if constexpr (ORD) {
#pragma unroll
      for (int q = 0; q < PERU; ++q) {
        const unsigned u = tid * PERU + q;
        if (u < nu) hval[ulist[u]] = krank[q] | (kcnt[q] > (unsigned)LCOLD ? 0x80000000u : 0u);
      }
    }
    __syncthreads();
    KV_STAMPP(2);

    // ---- rows that need lanes: init of new rows, flag recompute, scatter / mark bodies ---------
    if (MODE != MODE_UNIQUE) {
      const int lane8 = tid & 7;
      const unsigned nn = lnnew;
      const unsigned npad = (nn + 7u) & ~7u;
      for (unsigned j = tid >> 3; j < npad; j += TBK / 8) {
        const bool live = j < nn;
        const unsigned sv = live ? lnew[j] : 0u;
        const unsigned s = sv & 0x7FFFu;
        const bool isnew = (sv & 0x8000u) != 0;
        const long long key = (s == HSK) ? EMPTY_KEY : hkey[s];
        const unsigned r = live ? (hrow[s] & ROW_MASK) : 0u;
        float* row = row_ptr(a.tv, r);
        bool big = false, touch = false;
        if (live && r != 0) {
          if (isnew) big = init_row_coop(a.tv, key, row, lane8, 8);
          if (MODE == MODE_LOOKUP) {
            if (!isnew)
              for (int e = lane8; e < D; e += 8) big |= fabsf(row[e]) >= CUTOFF;
          } else if (MODE == MODE_APPLYIDX) {
            // UpdateUnderThreshold of the fresh row (insert_func, kv_variable.h:398-399)
          } else if (MODE == MODE_MARK) {
            if (a.mark_what == 0) {
              for (int e = lane8; e < D; e += 8) row[e] = 0.f;
            } else if (lane8 == 0) {
              *freq_ptr(a.tv, r) = a.fvals[hval[s]];
            }
          } else {
            // ScatterUpdate kv_variable.h:616-734 leaves blacklisted rows alone (:690).
            // InsertOrUpdate :423-485 copies the values but the key stays blacklisted and keeps
            // reading zeros (table_manager.h:224-226), so the zeroed row is left as it is too.
            const unsigned fl = isnew ? 0u : *flags_ptr(a.tv, r);
            // DeltaImport (is_insert == 3) overwrites the value and lifts the blacklist (dynamic_restore.hpp:65-75)
            touch = !(fl & FLAG_BLACK) || a.is_insert == 3;
            const float* src = a.grad + (size_t)hval[s] * D;
            if (touch) {
              big = false;
              for (int e = lane8; e < D; e += 8) {
                const float l = row[e], v = src[e];
                float o;
                switch (a.scatter_op) {
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
          }
        }
        const unsigned long long m = __ballot(big);
        const bool any = ((m >> ((tid & 63) & ~7)) & 0xFFull) != 0;
        if (live && r != 0 && lane8 == 0) {
          unsigned char* fp = flags_ptr(a.tv, r);
          if (MODE == MODE_LOOKUP) {
            const unsigned black = isnew ? 0u : (*fp & FLAG_BLACK);
            *fp = (unsigned char)(black ? (FLAG_BLACK | FLAG_UNDER) : (any ? 0u : FLAG_UNDER));
          } else if (MODE == MODE_APPLYIDX) {
            *fp = (unsigned char)(any ? 0u : FLAG_UNDER);
          } else if (MODE == MODE_MARK) {
            // a key first seen by the blacklist is inserted blacklisted with under_threshold
            // still false (EmbeddingValue(nullptr, true, 1, ...), table_manager.h:343-346)
            if (a.mark_what == 0) *fp = (unsigned char)(isnew ? FLAG_BLACK : (FLAG_BLACK | FLAG_UNDER));
          } else if (a.is_insert == 2) {
            if (isnew) *fp = 0;  // ImportValues does not evaluate under_threshold (dynamic_restore.hpp:183-194)
          } else if (a.is_insert == 3) {
            *fp = (unsigned char)(any ? 0u : FLAG_UNDER);  // RemoveBlacklist + UpdateUnderThreshold
          } else if (touch || isnew) {
            const unsigned black = isnew ? 0u : (*fp & FLAG_BLACK);
            *fp = (unsigned char)(black ? (FLAG_BLACK | FLAG_UNDER) : (any ? 0u : FLAG_UNDER));
          }
        }
      }
    }
    KV_STAMPP(3);

    // ---- pass 2: every entry learns its key's row / dense index and where its positions go ------
    if constexpr (ORD) {
      auto place = [&](unsigned ge, unsigned h, unsigned occ) {
        const unsigned off = atomicAdd(&hrun[h], occ);
        const unsigned rv = hrow[h];
        w.ent_b[ge] = rv & ROW_MASK;
        w.ent_base[ge] = (hocc[h] + off) | (off == 0u ? HEAD_BIT : 0u);
        if (off == 0u) w.ent_rec[ge] = hval[h];
      };
      if (a.det) {
        // deterministic mode: a key's entries take their positions in tile order = ascending x.  Entries are
        // taken in x order, TBK per round; inside a wave an entry adds up the occurrences of the lanes before
        // it that hold the same key (shuffle loop), across waves and rounds hrun[h] carries the key's total so
        // far, read and advanced wave by wave.
        const int lane = tid & 63, wave = tid >> 6;
        for (unsigned x0 = 0; x0 < E; x0 += TBK) {   // block-uniform
          const unsigned x = x0 + tid;
          bool valid = x < E;
          size_t ge = 0;
          unsigned h = 0xFFFFFFFFu, occ = 0;
          if (valid) {
            ge = seg_entry(tpre, tstart, NT, x);
            const long long key = w.ent_key[ge];
            valid = in_round(key, R, round);
            if (valid) {
              bool first;
              h = lds_key_slot<HSK>(hkey, &lsent, key, false, &first);
              occ = w.ent_a[ge] & 0xFFFFu;
            }
          }
          unsigned within = 0;
          for (int j = 0; j < 63; ++j) {   // lanes before this one with the same key
            const unsigned hj = __shfl(h, j), oj = __shfl(occ, j);
            if (j < lane && hj == h) within += oj;
          }
          unsigned off = 0;
          for (int wv = 0; wv < TBK / 64; ++wv) {   // block-uniform
            if (wave == wv && valid) {
              off = hrun[h] + within;
              atomicAdd(&hrun[h], occ);   // after every lane of the wave has read (LDS ops of a wave are in order)
            }
            __syncthreads();
          }
          if (valid) {
            w.ent_b[ge] = hrow[h] & ROW_MASK;
            w.ent_base[ge] = (hocc[h] + off) | (off == 0u ? HEAD_BIT : 0u);
            if (off == 0u) w.ent_rec[ge] = hval[h];
          }
        }
      } else if (cached) {
#pragma unroll
        for (int k = 0; k < EB; ++k)
          if (cge[k] != 0xFFFFFFFFu) place(cge[k], cslot[k], cocc[k]);
      } else {
        for (unsigned x = tid; x < E; x += TBK) {
          const size_t ge = seg_entry(tpre, tstart, NT, x);
          const long long key = w.ent_key[ge];
          if (!in_round(key, R, round)) continue;
          bool first;
          place((unsigned)ge, lds_key_slot<HSK>(hkey, &lsent, key, false, &first), w.ent_a[ge] & 0xFFFFu);
        }
      }
    }
    __syncthreads();
    KV_STAMPP(4);
#ifdef KV_STAMPS
    if (tid == 0) { w.dbg[(size_t)(blockIdx.x + 4096) * 16 + 8] = lnu; w.dbg[(size_t)(blockIdx.x + 4096) * 16 + 9] = R; w.dbg[(size_t)(blockIdx.x + 4096) * 16 + 10] = lnnew; }
#endif
  }
  if constexpr (ORD) {
    // the partition's cold keys in batches of 64 / apply_lanes(dim) (one key per lane group of an apply wave), behind its
    // chunk items: {first cold list index, keys in the batch}
    const unsigned nc = lcold, nk = lchunk, gb = 64u / (unsigned)apply_lanes(a.tv.dim);
    const unsigned nb = (nc + gb - 1u) / gb;
    for (unsigned b = tid; b < nb; b += TBK) w.litem[pbase + nk + b] = make_uint4(pbase + b * gb, min(gb, nc - b * gb), 0u, 0u);
    if (tid == 0) w.pmeta[p] = make_uint4(nk + nb, nk, pbase, nc);

  }
}

// ------------------------------------------------------------------------------------------
// k_order: the sorted position list (+ the lookup's fix-up of rows inserted by this batch)
// ------------------------------------------------------------------------------------------
// one input position's place in the sorted list (sr = its slot_rank word): order[ent_base[entry] + rank] = i;
// the first position of a cold key also goes into the key's record (most cold keys have no other row, and the
// apply saves a hop)
__device__ __forceinline__ void order_pos(const WsDev& w, long long n, long long i, unsigned sr) {
  if (sr == 0xFFFFFFFFu) return;   // a skipped record (padding of the sharded exchange)
  const unsigned e = sr & SLOT_MASK, rank = sr >> RANK_SHIFT;
  const unsigned eb = w.ent_base[e];
  const unsigned j = (eb & BASE_MASK) + rank;
  const bool head = (eb & HEAD_BIT) && rank == 0u;
  if (j < (unsigned)n) w.order[j] = (unsigned)i | (head ? HEAD_BIT : 0u);
  if (head) {
    const unsigned rec = w.ent_rec[e];
    if (!(rec >> 31)) reinterpret_cast<unsigned*>(&w.coldlist[2 * (size_t)rec + 1])[2] = (unsigned)i;
  }
}
// The partitions left their work items in their own stretches (litem, pmeta).  The first ITEM_BLOCKS blocks of
// the kernel that follows the partition pass make ONE dense directory of them: each scans pmeta (P entries)
// for itself and copies the items of its share of the partitions, hot chunk items getting their batch-wide
// chunk number.  Totals go to ctr[2] (items) and ctr[3] (chunks).
constexpr int ITEM_BLOCKS = 16;
__device__ __forceinline__ void items_body(const WsDev& w) {
  __shared__ unsigned sit[MAX_P + 1], sck[MAX_P + 1];
  __shared__ unsigned wt[8];
  const unsigned P = w.P;
  const int tid = threadIdx.x, T = blockDim.x;   // 256 threads
  const unsigned per = (P + T - 1) / T;
  const unsigned p0 = min(P, tid * per), p1 = min(P, p0 + per);
  unsigned si = 0, sc = 0;
  for (unsigned q = p0; q < p1; ++q) { const uint4 m = w.pmeta[q]; sit[q] = m.x; sck[q] = m.y; si += m.x; sc += m.y; }
  unsigned ti, tc;
  unsigned ri = block_excl_scan<TB / 64>(si, wt, &ti);
  unsigned rc = block_excl_scan<TB / 64>(sc, wt, &tc);
  for (unsigned q = p0; q < p1; ++q) { const unsigned a_ = sit[q], b_ = sck[q]; sit[q] = ri; sck[q] = rc; ri += a_; rc += b_; }
  __syncthreads();
  if (blockIdx.x == 0 && tid == 0) { w.ctr[2] = ti; w.ctr[3] = tc; }
  // this block's share of the items; an item's partition = last q with sit[q] <= its number
  const unsigned nblk = min((unsigned)ITEM_BLOCKS, gridDim.x);
  const unsigned ipb = (ti + nblk - 1) / nblk;
  const unsigned i0 = min(ti, blockIdx.x * ipb), i1 = min(ti, i0 + ipb);
  for (unsigned i = i0 + tid; i < i1; i += T) {
    unsigned lo = 0, hi = P;
    while (hi - lo > 1) {
      const unsigned mid = (lo + hi) >> 1;
      if (sit[mid] <= i) lo = mid; else hi = mid;
    }
    const uint4 m = w.pmeta[lo];
    uint4 it = w.litem[m.z + (i - sit[lo])];
    if (it.x & HEAD_BIT) it.z += sck[lo];
    w.items[i] = it;
  }
}

__device__ __forceinline__ void order_body(const TableDev& t, const WsDev& w, long long n) {
  if (*reinterpret_cast<volatile unsigned*>(&t.counters[1])) return;   // a partition overflowed: the lists are not valid
  if (blockIdx.x == 0 && threadIdx.x == 0) w.order[n] = HEAD_BIT;
  if (blockIdx.x < ITEM_BLOCKS) { items_body(w); return; }   // ITEM_BLOCKS extra blocks in front: the directory only
  const long long stride = (long long)(gridDim.x - ITEM_BLOCKS) * blockDim.x;
  for (long long i = (long long)(blockIdx.x - ITEM_BLOCKS) * blockDim.x + threadIdx.x; i < n; i += stride)
    order_pos(w, n, i, __builtin_nontemporal_load(&w.slot_rank[i]));
}
__global__ void __launch_bounds__(TB) k_order(TableDev t, WsDev w, long long n) { order_body(t, w, n); }

// ------------------------------------------------------------------------------------------
// k_apply / k_apply_fin: segmented sum over the sorted position list + fused row update
// ------------------------------------------------------------------------------------------
// The partition pass left two lists of key records {key, row, slot-row hint | start, count, ...}: cold keys
// (<= LCOLD occurrences in the batch: ~98 % of the keys of a Zipf batch, ~25 % of its rows) and hot keys, whose
// rows are cut into chunks of HC rows.  The unit of work is a WAVE, not a block — no LDS, no barrier, no atomic:
//   hot chunk   the wave's 64 / LPR lane groups sum the chunk's rows, RB rows in flight per group, and
//               meet through shuffles; a key with one chunk is updated on the spot, else the chunk's sum
//               goes to hpart and k_apply_fin adds the key's chunks up in chunk order and updates it
//   cold batch  one lane group per key: its rows (order[start .. start + count)), its state rows and the
//               record of the hinted slot row are requested together, then the fused update
// Every wave gets its share of both kinds (hot chunks: streaming, bandwidth-bound; cold batches: dependent
// hops, latency-bound), so the two overlap on every CU.
constexpr int TBS = 256;

// combine two partial results of a key under the fold operation (sum for the optimizers)
__device__ __forceinline__ float fold2(int op, float acc, float v) {
  switch (op) {
    case KV_SCATTER_MUL: return acc * v;
    case KV_SCATTER_MIN: return fminf(acc, v);
    case KV_SCATTER_MAX: return fmaxf(acc, v);
    default: return acc + v;
  }
}
__device__ __forceinline__ float fold_identity(int op) {
  switch (op) {
    case KV_SCATTER_MUL: return 1.f;
    case KV_SCATTER_MIN: return INFINITY;
    case KV_SCATTER_MAX: return -INFINITY;
    default: return 0.f;
  }
}

// ------------------------------------------------------------------------------------------
// k_occ_sum: occurrence order (PartArgs::det == 2) — the hot keys' sums, one chain per key
// ------------------------------------------------------------------------------------------
// TF-core's unsorted_segment_sum adds a segment's rows one by one in input order, starting from +0; fp32 addition does not
// associate, so the only way to the same bits is the same chain.  A cold key (<= LCOLD rows) is summed that way by its lane
// group in k_apply anyway.  A hot key has ONE work item in this mode (chunk_rows) and this kernel in front of k_apply takes
// its chain: one block per item; wave 0 carries the running sum (lane = column, NC columns per lane) over the rows of a
// stage in LDS while the other waves fetch the next stage's rows (positions of the stage after that): the chain's wave
// never waits for HBM, only for its own additions.  The sum goes to hpart[the item's chunk number], where k_apply's hot path
// picks it up.  Fold ops of the scatter family ride the same chain.
constexpr int OCC_TB = 512;            // 1 chain wave + 7 fetch waves
constexpr int OCC_STAGE = 14336;       // floats per stage (two stages in LDS: 124 KB with the positions — one block per CU;
                                       // a block's fetch is one HBM round trip per stage, so the stage is what it can hold)
constexpr int OCC_ROWS = 1024;         // rows per stage at most (the stage's positions: three stages in LDS)
inline size_t occ_smem_bytes() { return (size_t)2 * OCC_STAGE * 4 + (size_t)3 * OCC_ROWS * 4; }
template <int NC, int FOP, int DC = 0>   // DC: the dim when it is one of the common ones (the rows' LDS offsets become immediates), else 0
__device__ __forceinline__ void occ_chain(const float* __restrict__ buf, unsigned rows, int D_, int lane, float (&acc)[NC]) {
  const int D = DC ? DC : D_;
  // rows of the stage in order; a lane's columns are independent chains.  The LDS reads of a batch of rows are issued
  // together, one batch ahead of the additions; lanes past the row's end read column 0 and their sums are dropped.
  // Measured on a key of 188 k rows of dim 32 (tools/occ_step.py): 1.3 ms for the chain with computed addresses (~15
  // cycles a row); with the rows as immediates (DC) ~10 cycles a row — 0.8 ms, level with the fetch's 0.9 (a fetch twice as
  // fast, two groups on alternate stages, changed nothing: a dependent fp32 addition of a wave64 is ~8 cycles by itself).
  constexpr int UJ = NC >= 8 ? 2 : 16 / NC;
  int col[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) col[c] = (lane + c * 64 < D) ? lane + c * 64 : 0;
  auto rd = [&](unsigned r0, float (&x)[UJ][NC]) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const float* pc = buf + (size_t)r0 * D + col[c];   // (one address a batch and column: the rows are immediates when DC != 0)
#pragma unroll
      for (int j = 0; j < UJ; ++j) x[j][c] = pc[j * D];
    }
  };
  auto add = [&](const float (&x)[UJ][NC]) {
#pragma unroll
    for (int j = 0; j < UJ; ++j)
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[c] = fold2(FOP, acc[c], x[j][c]);   // (FOP is a constant: a run-time switch per addition cost 80 cycles a row, 6.6 ms)
  };
  float xa[UJ][NC], xb[UJ][NC];
  const unsigned full = rows / (2 * UJ) * (2 * UJ);
  unsigned r = 0;
  if (full) {
    rd(0, xa);
    for (; r < full; r += 2 * UJ) {   // wave-uniform
      rd(r + UJ, xb);
      add(xa);
      if (r + 2 * UJ < full) rd(r + 2 * UJ, xa);
      add(xb);
    }
  }
  for (; r < rows; ++r) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const float x = buf[(size_t)r * D + col[c]];
      acc[c] = fold2(FOP, acc[c], x);
    }
  }
}
template <int NC>
__device__ __forceinline__ void occ_sum_body(const WsDev& w, const PartArgs& a, int fop) {
  if (*reinterpret_cast<volatile unsigned*>(&a.tv.counters[1])) return;   // the index pass gave up on this batch
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* buf = reinterpret_cast<float*>(smem_raw);                               // [2][OCC_STAGE]
  unsigned* lpos = reinterpret_cast<unsigned*>(smem_raw + (size_t)2 * OCC_STAGE * 4);   // [3][OCC_ROWS]
  const unsigned total = w.ctr[2];
  const int D = a.tv.dim;
  const unsigned S = min((unsigned)OCC_ROWS, (unsigned)OCC_STAGE / (unsigned)D);   // rows per stage (D <= 1024: at least 14)
  const int tid = threadIdx.x, lane = tid & 63;
  const bool chain = tid < 64;
  const int ft = tid - 64;                       // fetch thread number
  constexpr int FT = OCC_TB - 64;
  const float ident = fold_identity(fop);
  const bool vec = (D & 3) == 0;
  for (unsigned it = blockIdx.x; it < total; it += gridDim.x) {   // block-uniform
    const uint4 item = w.items[it];
    if (!(item.x & HEAD_BIT)) continue;
    const uint4 rb = w.hotlist[2 * (size_t)(item.x & ~HEAD_BIT) + 1];
    const unsigned lo = rb.x, cnt = rb.y;
    const unsigned nst = (cnt + S - 1) / S;
    auto rows_of = [&](unsigned st) { return min(S, cnt - st * S); };
    constexpr int PV = (OCC_ROWS + FT - 1) / FT;
    auto ask_pos = [&](unsigned st, unsigned (&pv)[PV]) {   // fetch waves: the stage's positions, asked for ...
      if (st >= nst) return;
      const unsigned n_ = rows_of(st);
#pragma unroll
      for (int k = 0; k < PV; ++k) pv[k] = w.order[lo + st * S + min((unsigned)(ft + k * FT), n_ - 1u)] & ~HEAD_BIT;
    };
    auto put_pos = [&](unsigned st, const unsigned (&pv)[PV]) {   // ... and filed (behind the rows' loads: one round trip for both)
      if (st >= nst) return;
      unsigned* dst = lpos + (size_t)(st % 3u) * OCC_ROWS;
      const unsigned n_ = rows_of(st);
#pragma unroll
      for (int k = 0; k < PV; ++k) {
        const unsigned r = ft + k * FT;
        if (r < n_) dst[r] = pv[k];
      }
    };
    auto fetch_rows = [&](unsigned st) {         // fetch waves: the stage's rows, every load of a thread in flight together
      if (st >= nst) return;
      const unsigned* ps = lpos + (size_t)(st % 3u) * OCC_ROWS;
      float* dst = buf + (size_t)(st & 1u) * OCC_STAGE;
      const unsigned n_ = rows_of(st);
      if (vec) {
        const unsigned q = (unsigned)D >> 2, nv = n_ * q;
        constexpr int UB = 8;
        for (unsigned e0 = 0; e0 < nv; e0 += UB * FT) {
          float4 v[UB];
#pragma unroll
          for (int u = 0; u < UB; ++u) {
            // (no branch around a load: a thread past the stage's end reads its last element again — a load under a
            //  condition made the compiler wait for the one before it, eight round trips per stage instead of one: 2.2 ms
            //  against 0.9 for the fetch of the same key)
            const unsigned e = min(e0 + u * FT + ft, nv - 1u);
            const unsigned r = e / q, c4 = e - r * q;
            const unsigned pos = ps[r];
            const float* base = (pos & EP_TAG) ? a.epart : a.grad;
            const float* src = base + (size_t)(pos & ~EP_TAG) * D + 4u * c4;
            float t4[4];
            ldv_stream<4>(src, t4);
            v[u] = make_float4(t4[0], t4[1], t4[2], t4[3]);
          }
#pragma unroll
          for (int u = 0; u < UB; ++u) {
            const unsigned e = e0 + u * FT + ft;
            if (e < nv) reinterpret_cast<float4*>(dst)[e] = v[u];
          }
        }
      } else {
        const unsigned ne = n_ * (unsigned)D;
        constexpr int UB = 8;
        for (unsigned e0 = 0; e0 < ne; e0 += UB * FT) {
          float v[UB];
#pragma unroll
          for (int u = 0; u < UB; ++u) {
            const unsigned e = min(e0 + u * FT + ft, ne - 1u);
            const unsigned r = e / (unsigned)D, c = e - r * (unsigned)D;
            const unsigned pos = ps[r];
            const float* base = (pos & EP_TAG) ? a.epart : a.grad;
            v[u] = __builtin_nontemporal_load(base + (size_t)(pos & ~EP_TAG) * D + c);
          }
#pragma unroll
          for (int u = 0; u < UB; ++u) {
            const unsigned e = e0 + u * FT + ft;
            if (e < ne) dst[e] = v[u];
          }
        }
      }
    };
    float acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = ident;
    __syncthreads();                             // the previous item's buffers have been read
    unsigned pv[PV];
    if (!chain) { ask_pos(0, pv); put_pos(0, pv); ask_pos(1, pv); put_pos(1, pv); }
    __syncthreads();
    if (!chain) fetch_rows(0);
    __syncthreads();
    for (unsigned st = 0; st < nst; ++st) {      // block-uniform
      if (chain) {
        const float* sb = buf + (size_t)(st & 1u) * OCC_STAGE;
        switch (fop) {
          case KV_SCATTER_MUL: occ_chain<NC, KV_SCATTER_MUL>(sb, rows_of(st), D, lane, acc); break;
          case KV_SCATTER_MIN: occ_chain<NC, KV_SCATTER_MIN>(sb, rows_of(st), D, lane, acc); break;
          case KV_SCATTER_MAX: occ_chain<NC, KV_SCATTER_MAX>(sb, rows_of(st), D, lane, acc); break;
          default:   // the optimizers' sum: the common dims get their own copy (one LDS read and one addition a row, nothing else)
            if (NC == 1 && D == 32) occ_chain<NC, KV_SCATTER_ADD, 32>(sb, rows_of(st), D, lane, acc);
            else if (NC == 1 && D == 64) occ_chain<NC, KV_SCATTER_ADD, 64>(sb, rows_of(st), D, lane, acc);
            else if (NC == 1 && D == 16) occ_chain<NC, KV_SCATTER_ADD, 16>(sb, rows_of(st), D, lane, acc);
            else if (NC == 1 && D == 8) occ_chain<NC, KV_SCATTER_ADD, 8>(sb, rows_of(st), D, lane, acc);
            else if (NC == 2 && D == 128) occ_chain<NC, KV_SCATTER_ADD, 128>(sb, rows_of(st), D, lane, acc);
            else occ_chain<NC, KV_SCATTER_ADD>(sb, rows_of(st), D, lane, acc);
            break;
        }
      }
      else { ask_pos(st + 2, pv); fetch_rows(st + 1); put_pos(st + 2, pv); }
      __syncthreads();
    }
    if (chain) {
      float* dst = w.hpart + (size_t)item.z * D;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int d = lane + c * 64;
        if (d < D) dst[d] = acc[c];
      }
    }
  }
}
template <int NC>
__global__ void __launch_bounds__(OCC_TB) k_occ_sum(WsDev w, PartArgs a, int fop) { occ_sum_body<NC>(w, a, fop); }


// The slot-table rows of one key, resolved by the group leader.  FindOrInsertUnsafe(var, filter_out !=
// nullptr) kv_variable.h:382-408 and FindOrInsertUnsafe(slot, nullptr) :409-414; FTRL probes linear
// before accum (training_ops.cc:701-704).  `m0` is the record of the hinted slot row (requested early).
struct RowsOf { unsigned tag, r0, r1, nb; };   // nb: bit 1 / 2 = slot row 0 / 1 inserted now, bit 3 = slot row 0 is the hinted one
template <int OPT>
__device__ __forceinline__ RowsOf resolve_rows(const PartArgs& a, long long key, unsigned rvw, unsigned hint,
                                               bool hint_loaded, const RowMeta& m0) {
  const unsigned rv = rvw & ROW_MASK;
  const bool vnew = (rvw >> 31) != 0u;   // inserted by this apply: not filtered (kv_variable.h:400-407, succ == false)
  RowsOf o{rv, 0u, 0u, 0u};
  if (rv == 0u) return o;
  // the var record is only needed for the frequency filter: a blacklisted row is all zeros already
  // (RemoveBlacklistUnsafe hands out a zero row, table_manager.h:359-372) and the group optimizers
  // rewrite the flags after the update, so with enter_threshold == 0 they never read it
  const bool need_vmeta = OPT == OPT_ADAGRAD || a.tv.enter_threshold != 0u;
  if (need_vmeta && !vnew) {
    const uint2 mv = load_freq_flags(a.tv, rv);
    if ((mv.x & 0xFFFFu) < a.tv.enter_threshold) { o.tag = rv | ROW_FILTERED; return o; }  // kv_variable.h:910
    if (mv.y & FLAG_BLACK) meta_ptr(a.tv, rv)->flags = FLAG_UNDER;   // RemoveBlacklistUnsafe: fresh zero row (ours already is)
  }
  // slot rows are only created for keys the update will touch (filtered keys returned above)
  bool hinted = false;
  auto slot_row = [&](const TableDev& t, bool use_hint, bool* isnew) -> unsigned {
    *isnew = false;
    unsigned r = 0, f = 0;
    if (use_hint && hint_loaded && m0.key == key && !(m0.flags & FLAG_FREE)) {
      r = hint; f = m0.freq; hinted = true;
    } else {
      r = table_find(t, key);
      if (__builtin_expect(r == 0u, 0)) {
        r = table_find_or_insert(t, key, isnew);
        if (r && *isnew) { RowMeta* m = meta_ptr(t, r); m->freq = 1u; m->flags = 0; }
      }
      if (r && !*isnew) f = meta_ptr(t, r)->freq;
      if (use_hint && r) {   // remember it in the var's index entry
        Entry* e = table_entry_of(a.tv, key);
        if (e) e->hint = r;
      }
    }
    // AddFrequency(1, today) on a slot row that already existed (kv_variable.h:409-414); a new one keeps word 1
    if (r && !*isnew) {
      unsigned lo = (f & 0xFFFFu) + 1u;
      if (lo > 65535u) lo = 65535u;
      *freq_ptr(t, r) = (a.day << 16) | lo;
    }
    return r;
  };
  bool new0 = false, new1 = false;
  if (OPT == OPT_FTRL) o.r1 = slot_row(a.ts1, false, &new1);
  o.r0 = slot_row(a.ts0, a.use_hints != 0, &new0);
  // MarkAsDeltaListElements on every table of the op, for the keys the update reaches (training_ops.cc:7196-7201)
  if (__builtin_expect(a.tv.track_delta | a.ts0.track_delta | (OPT == OPT_FTRL ? a.ts1.track_delta : 0u), 0)) {
    mark_delta(a.tv, rv);
    if (o.r0) mark_delta(a.ts0, o.r0);
    if (OPT == OPT_FTRL && o.r1) mark_delta(a.ts1, o.r1);
  }
  o.nb = (new0 ? 2u : 0u) | (new1 ? 4u : 0u) | (hinted ? 8u : 0u);
  return o;
}

// Everything the update of one key needs besides its gradient, requested in ONE round trip: the var row, the
// hinted slot row and that row's own record (the hint is validated against it in resolve_rows).  Without this the
// finish walks slot index -> slot record -> rows, three dependent hops.  Called by the LPR lanes of the key's group.
template <int OPT, int V, int LPR, int K>
__device__ __forceinline__ void prefetch_state(const PartArgs& a, const uint4& ra, bool live, int lane, int D, RowMeta& m0,
                                               bool& hint_loaded, PreRows<V, K>& pre, bool& have_x, bool& have_s) {
  hint_loaded = false; have_x = false; have_s = false;
  if (!live || (ra.z & ROW_MASK) == 0u) return;
  const bool hok = a.use_hints && ra.w != 0u && ra.w < a.ts0.max_rows;
  if (lane == 0 && hok) {
    const uint4 mm = *reinterpret_cast<const uint4*>(meta_ptr(a.ts0, ra.w));
    m0.key = (long long)(((unsigned long long)mm.y << 32) | mm.x);
    m0.freq = mm.z;
    m0.flags = (unsigned char)(mm.w & 0xFFu);
    hint_loaded = true;
  }
  const float* xr = row_ptr(a.tv, ra.z & ROW_MASK);
  const float* sr = hok ? row_ptr(a.ts0, ra.w) : nullptr;
  constexpr int NS0 = (OPT == OPT_ADAM_V4 || OPT == OPT_ADAM_V3) ? 3 : 1;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int e0 = (lane + k * LPR) * V;
    if (e0 < D) {
      ldv<V>(xr + e0, pre.x[k]);
      if (hok) {
#pragma unroll
        for (int b3 = 0; b3 < NS0; ++b3) ldv<V>(sr + e0 + b3 * D, pre.s[b3][k]);
      }
    }
  }
  have_x = true; have_s = hok;
}

// finish one key whose combined gradient is in gv: optimizer update (MODE_APPLY) or emit (MODE_DEDUP).
// All LPR lanes of every group of the wave call it (shuffles inside); `live` masks groups without a key.
// hd = {key lo, key hi, row word, slot-row hint}
template <int MODE, int OPT, int V, int LPR, int K>
__device__ __forceinline__ void finish_key(const PartArgs& a, const uint4 hd, bool live, bool hint_loaded,
                                           const RowMeta& m0, float (&gv)[K][V], int lane,
                                           const PreRows<V, K>* pre = nullptr, bool have_x = false, bool have_s = false) {
  const int D = a.tv.dim;
  const long long key = (long long)(((unsigned long long)hd.y << 32) | hd.x);
  if (MODE == MODE_APPLY) {
    RowsOf ro{0u, 0u, 0u, 0u};
    if (live && lane == 0) ro = resolve_rows<OPT>(a, key, hd.z, hd.w, hint_loaded, m0);
    if (LPR > 1) {
      ro.tag = __shfl(ro.tag, 0, LPR); ro.r0 = __shfl(ro.r0, 0, LPR);
      ro.r1 = __shfl(ro.r1, 0, LPR); ro.nb = __shfl(ro.nb, 0, LPR);
    }
    // the slot rows in `pre` are those of the hinted row: good only if the hint stood up
    opt_update_row<OPT, V, LPR, K>(a.tv, a.ts0, a.ts1, key, ro.tag, ro.r0, (ro.nb & 2u) != 0, ro.r1, (ro.nb & 4u) != 0,
                                   live, gv, a.opt, lane, pre, have_x, have_s && (ro.nb & 8u) != 0);
    // the key's slot record as this update left it goes into the var row's mirror (clean: the slot table's own record is
    // up to date), so that the key's NEXT apply takes the lean path without reading it
    if (OPT != OPT_FTRL && a.use_mirror && live && lane == 0 && ro.r0 != 0u && (ro.tag & ROW_MASK) != 0u && !(ro.tag & ROW_FILTERED)) {
      const uint2 sm = load_freq_flags(a.ts0, ro.r0);
      SlotMirror nm;
      nm.srow = ro.r0; nm.freq = sm.x; nm.flags = (unsigned char)(sm.y & 0xFFu); nm.state = (unsigned char)MIRROR_CLEAN;
      nm.epoch = (unsigned short)a.mirror_epoch; nm.pad = 0u;
      *mirror_ptr(a.tv, ro.tag & ROW_MASK) = nm;
    }
  } else if (live && hd.z != ROW_MASK) {
    const size_t orow = a.out_map ? (size_t)a.out_map[hd.z] : (size_t)hd.z;   // sharded apply: the unique id's exchange slot
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int e0 = (lane + k * LPR) * V;
      if (e0 < D) stv<V>(a.out_sum + orow * D + e0, gv[k]);
    }
  }
}

template <int MODE, int OPT, int V, int LPR, int K>
__device__ __forceinline__ void apply_body(const WsDev& w, const PartArgs& a) {
  static_assert(LPR <= 64, "a row is handled by the lanes of one wave");
  const unsigned errflag = *reinterpret_cast<volatile unsigned*>(&a.tv.counters[1]);
  const unsigned total = w.ctr[2];   // work items: hot chunks and cold batches, every one about the same weight
  if (errflag) return;   // the index pass gave up on this batch
  const int D = a.tv.dim;
  constexpr int G = 64 / LPR;                    // lane groups (keys / rows in flight side by side) per wave
  constexpr int RB = (8 / K) > 0 ? (8 / K) : 1;  // rows in flight per group, hot chunks
  constexpr int RC = (2 / K) > 0 ? (2 / K) : 1;  // further rows of a cold key in flight per group (most keys have none)
  const int wl = threadIdx.x & 63;
  const int lane = wl % LPR;
  const int g = wl / LPR;
  const int fop = (MODE == MODE_APPLY) ? KV_SCATTER_ADD : a.fold_op;
  const float ident = (MODE == MODE_APPLY) ? 0.f : fold_identity(fop);
  // (both bases in registers: a select between a.epart and a.grad themselves compiles to a per-lane load of the
  // pointer from the argument block in front of every row)
  const float* const gbase = a.grad;
  const float* const ebase = a.epart;
  auto load_row = [&](unsigned pos, float (&dst)[K][V]) {
    const float* base = (pos & EP_TAG) ? ebase : gbase;
    const float* src = base + (size_t)(pos & ~EP_TAG) * D;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int e0 = (lane + k * LPR) * V;
      if (e0 < D) ldv_stream<V>(src + e0, dst[k]);
    }
  };
  auto acc_row = [&](float (&gv)[K][V], const float (&v)[K][V]) {
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int cc = 0; cc < V; ++cc) gv[k][cc] = (MODE == MODE_APPLY) ? gv[k][cc] + v[k][cc] : fold2(fop, gv[k][cc], v[k][cc]);
  };
  // Round-robin hand-out: the items weigh about the same (HC rows of a hot key ~ one batch of cold keys), and
  // consecutive items come from one hash partition, so a wave's items are a random sample of the batch.
  // (Run-time tickets were tried: 18 k returning atomics per launch more than doubled every item's time.)
  const unsigned W = gridDim.x * (TBS / 64);
#ifdef KV_STAMPS
  unsigned long long st_t0 = wall_clock64(), st_hot = 0, st_cold = 0, st_nh = 0, st_nc = 0;
#endif
  for (unsigned it = blockIdx.x * (TBS / 64) + (threadIdx.x >> 6); it < total; it += W) {
    const uint4 item = w.items[it];
    const bool is_hot = (item.x & HEAD_BIT) != 0u;
#ifdef KV_STAMPS
    const unsigned long long st_a = wall_clock64();
#endif
    float gv[K][V];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int cc = 0; cc < V; ++cc) gv[k][cc] = ident;
    // the state of the key that gets finished here (prefetch_state): one set of registers for both kinds of item
    RowMeta m0{};
    bool hint_loaded = false, have_x = false, have_s = false;
    PreRows<V, K> pre;
    if (is_hot) {
      // ---- hot chunk: rows [lo, hi) of one key, G * RB of them per step ------------------------------------
      const unsigned hx = item.x & ~HEAD_BIT;
      const uint4 ra = w.hotlist[2 * (size_t)hx], rb = w.hotlist[2 * (size_t)hx + 1];
      const unsigned lo = rb.x + item.y * rb.w, hi = min(rb.x + rb.y, lo + rb.w);
      constexpr int SR = G * RB;              // rows per step: row r of a step goes to group r % G
      const unsigned nst = (hi - lo + SR - 1) / SR;
      auto ldpos = [&](unsigned st, unsigned (&pp)[RB]) {
#pragma unroll
        for (int j = 0; j < RB; ++j) {
          const unsigned idx = lo + st * SR + j * G + g;
          pp[j] = idx < hi ? (w.order[idx] & ~HEAD_BIT) : 0xFFFFFFFFu;
        }
      };
      auto ldrows = [&](const unsigned (&pp)[RB], float (&dst)[RB][K][V]) {
#pragma unroll
        for (int j = 0; j < RB; ++j) {
#pragma unroll
          for (int k = 0; k < K; ++k)
#pragma unroll
            for (int cc = 0; cc < V; ++cc) dst[j][k][cc] = ident;
          if (pp[j] != 0xFFFFFFFFu) load_row(pp[j], dst[j]);
        }
      };
      // one step at a time per wave (registers are what limits the waves per SIMD, and the waves are what
      // hides the hops: a second row buffer costs more than it gains); positions one step ahead
      unsigned pa_[RB], pb_[RB];
      float va[RB][K][V];
      const bool occ_order = a.det == 2;   // uniform over the launch
      if (occ_order) {
        // occurrence order: the key's rows were added one by one, in list order, by k_occ_sum (one chain per key: a block
        // with its own load pipeline); every group reads the sum
        const float* src = w.hpart + (size_t)item.z * D;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const int e0 = (lane + k * LPR) * V;
          if (e0 < D) ldv<V>(src + e0, gv[k]);
        }
      } else {
        ldpos(0, pa_);
        for (unsigned st = 0; st < nst; ++st) {
          ldrows(pa_, va);
          ldpos(st + 1, pb_);
#pragma unroll
          for (int j = 0; j < RB; ++j) { acc_row(gv, va[j]); pa_[j] = pb_[j]; }
        }
      }
      // a key with a single chunk is finished here: its state rows are requested now (the row buffers are free),
      // one round trip under the shuffle tree instead of three hops behind it
      const unsigned nch = (rb.y + rb.w - 1u) / rb.w;
      if (MODE == MODE_APPLY && nch == 1u) prefetch_state<OPT, V, LPR, K>(a, ra, g == 0, lane, D, m0, hint_loaded, pre, have_x, have_s);
      // the groups' sums meet: a fixed shuffle tree, every lane ends with the chunk's sum
      // (occurrence order: every group holds the whole sum already)
      if (!occ_order) {
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) {
#pragma unroll
          for (int k = 0; k < K; ++k)
#pragma unroll
            for (int cc = 0; cc < V; ++cc) {
              const float x = __shfl_xor(gv[k][cc], o);
              gv[k][cc] = (MODE == MODE_APPLY) ? gv[k][cc] + x : fold2(fop, gv[k][cc], x);
            }
        }
      }
      if (nch > 1u) {
        if (g == 0) {
          float* dst = w.hpart + (size_t)item.z * D;
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const int e0 = (lane + k * LPR) * V;
            if (e0 < D) stv<V>(dst + e0, gv[k]);
          }
        }
      } else {
        finish_key<MODE, OPT, V, LPR, K>(a, ra, g == 0, hint_loaded, m0, gv, lane, &pre, have_x, have_s);
      }
    } else {
      // ---- cold batch: one key per lane group ------------------------------------------------------------
      // One round trip fetches everything a key with a single row needs (three quarters of the cold keys):
      // its gradient row (the record carries the first position), the var row, the hinted slot row and that
      // row's own record.  Further rows of the key, if any, follow RC at a time.
      const unsigned u = item.x + g;
      const bool live = (unsigned)g < item.y;
      uint4 ra = make_uint4(0u, 0u, 0u, 0u), rb = ra;
      if (live) { ra = w.coldlist[2 * (size_t)u]; rb = w.coldlist[2 * (size_t)u + 1]; }
      const unsigned start = rb.x, cnt = live ? rb.y : 0u;
      if (cnt > 0u) {
        load_row(rb.z, gv);     // ident op with one operand: the first row IS the partial result
        if (a.det == 2) {       // ... up to the sign of a zero: TF-core's segment sum starts from +0 (0 + -0 = +0)
#pragma unroll
          for (int k = 0; k < K; ++k)
#pragma unroll
            for (int cc = 0; cc < V; ++cc) gv[k][cc] = (MODE == MODE_APPLY) ? 0.f + gv[k][cc] : fold2(fop, ident, gv[k][cc]);
        }
      }
      if (MODE == MODE_APPLY) prefetch_state<OPT, V, LPR, K>(a, ra, live, lane, D, m0, hint_loaded, pre, have_x, have_s);
      for (unsigned j0 = 1; j0 < cnt; j0 += RC) {
        float val[RC][K][V];
        unsigned pos[RC];
#pragma unroll
        for (int j = 0; j < RC; ++j) pos[j] = j0 + j < cnt ? (w.order[start + j0 + j] & ~HEAD_BIT) : 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < RC; ++j) {
#pragma unroll
          for (int k = 0; k < K; ++k)
#pragma unroll
            for (int cc = 0; cc < V; ++cc) val[j][k][cc] = ident;
          if (pos[j] != 0xFFFFFFFFu) load_row(pos[j], val[j]);
        }
#pragma unroll
        for (int j = 0; j < RC; ++j) acc_row(gv, val[j]);
      }
      finish_key<MODE, OPT, V, LPR, K>(a, ra, live, hint_loaded, m0, gv, lane, &pre, have_x, have_s);
    }
#ifdef KV_STAMPS
    {
      const unsigned long long now = wall_clock64();
      if (is_hot) { st_hot += now - st_a; ++st_nh; } else { st_cold += now - st_a; ++st_nc; }
    }
#endif
  }
#ifdef KV_STAMPS
  if (wl == 0) {
    unsigned long long* d = w.dbg + (size_t)(8192 + blockIdx.x * (TBS / 64) + (threadIdx.x >> 6)) * 16;
    d[0] = st_t0; d[1] = wall_clock64(); d[2] = st_hot; d[3] = st_cold; d[4] = st_nh; d[5] = st_nc; d[6] = total; d[7] = w.ctr[3]; d[8] = 0;
  }
#endif
}

// hot keys with more than one chunk: the chunks' sums (hpart) are added up — the block's waves take
// consecutive runs of the key's chunks, lane group g of a wave every G-th chunk of the run, the groups meet
// through the shuffle tree and the waves through LDS in wave order — and the key is finished.  A block reads
// its share of the item directory at once (a key's first chunk item names the key); keys of up to 64 chunks
// go one to a wave, bigger ones to the whole block.  The order of the additions depends on nothing but the key.
constexpr int TBF = 1024;
template <int MODE, int OPT, int V, int LPR, int K>
__device__ __forceinline__ void apply_fin_body(const WsDev& w, const PartArgs& a) {
  const unsigned errflag = *reinterpret_cast<volatile unsigned*>(&a.tv.counters[1]);
  const unsigned total = w.ctr[2];
  if (errflag) return;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* lsum = reinterpret_cast<float*>(smem_raw);   // [TBF / 64][dim]
  __shared__ unsigned lkeys[TBF][7];                  // multi-chunk keys among this block's items: {hot index, first chunk, chunks, record a}
  __shared__ unsigned lnk;
  const int D = a.tv.dim;
  constexpr int G = 64 / LPR;
  constexpr int RB = (8 / K) > 0 ? (8 / K) : 1;
  constexpr int NW = TBF / 64;
  const int wl = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int lane = wl % LPR;
  const int g = wl / LPR;
  const int fop = (MODE == MODE_APPLY) ? KV_SCATTER_ADD : a.fold_op;
  const float ident = (MODE == MODE_APPLY) ? 0.f : fold_identity(fop);
  // the block's contiguous share of the item directory, read at once
  const unsigned per_b = (total + gridDim.x - 1) / gridDim.x;
  const unsigned b0 = min(total, blockIdx.x * per_b), b1 = min(total, b0 + per_b);
  if (threadIdx.x == 0) lnk = 0;
  __syncthreads();
  for (unsigned i = b0 + threadIdx.x; i < b1; i += TBF) {
    const uint4 item = w.items[i];
    if ((item.x & HEAD_BIT) && item.y == 0u) {
      const uint4 ra = w.hotlist[2 * (size_t)(item.x & ~HEAD_BIT)];       // both halves of the record in one hop
      const uint4 rb = w.hotlist[2 * (size_t)(item.x & ~HEAD_BIT) + 1];
      const unsigned nch = (rb.y + rb.w - 1u) / rb.w;
      if (nch > 1u) {
        const unsigned q = atomicAdd(&lnk, 1u);
        if (q < (unsigned)TBF) {
          lkeys[q][0] = item.x & ~HEAD_BIT; lkeys[q][1] = item.z; lkeys[q][2] = nch;
          lkeys[q][3] = ra.x; lkeys[q][4] = ra.y; lkeys[q][5] = ra.z; lkeys[q][6] = ra.w;
        }
      }
    }
  }
  __syncthreads();
  const unsigned nk = min(lnk, (unsigned)TBF);
  // sum of chunks [c0, c1) of a key by one wave: lane group g takes every G-th chunk, the groups meet in the
  // shuffle tree; every lane returns the sum
  auto wave_sum = [&](unsigned first, unsigned c0, unsigned c1, float (&gv)[K][V]) {
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int cc = 0; cc < V; ++cc) gv[k][cc] = ident;
    for (unsigned cb = c0; cb < c1; cb += G * RB) {
      float val[RB][K][V];
#pragma unroll
      for (int j = 0; j < RB; ++j) {
        const unsigned ci = cb + j * G + g;
#pragma unroll
        for (int k = 0; k < K; ++k) {
#pragma unroll
          for (int cc = 0; cc < V; ++cc) val[j][k][cc] = ident;
          const int e0 = (lane + k * LPR) * V;
          if (ci < c1 && e0 < D) ldv<V>(w.hpart + (size_t)(first + ci) * D + e0, val[j][k]);
        }
      }
#pragma unroll
      for (int j = 0; j < RB; ++j)
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
          for (int cc = 0; cc < V; ++cc)
            gv[k][cc] = (MODE == MODE_APPLY) ? gv[k][cc] + val[j][k][cc] : fold2(fop, gv[k][cc], val[j][k][cc]);
    }
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) {
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int cc = 0; cc < V; ++cc) {
          const float x = __shfl_xor(gv[k][cc], o);
          gv[k][cc] = (MODE == MODE_APPLY) ? gv[k][cc] + x : fold2(fop, gv[k][cc], x);
        }
    }
  };
  // keys with up to 64 chunks: one wave each
  for (unsigned q = wv; q < nk; q += NW) {
    if (lkeys[q][2] > 64u) continue;
    float gv[K][V];
    const uint4 ra = make_uint4(lkeys[q][3], lkeys[q][4], lkeys[q][5], lkeys[q][6]);
    RowMeta m0{};
    bool hl = false, hx = false, hs = false;
    PreRows<V, K> pre;
    if (MODE == MODE_APPLY) prefetch_state<OPT, V, LPR, K>(a, ra, g == 0, lane, D, m0, hl, pre, hx, hs);   // with the partials
    wave_sum(lkeys[q][1], 0u, lkeys[q][2], gv);
    finish_key<MODE, OPT, V, LPR, K>(a, ra, g == 0, hl, m0, gv, lane, &pre, hx, hs);
  }
  // keys with more: the block's waves take consecutive runs of the chunks and meet in LDS in wave order
  for (unsigned q = 0; q < nk; ++q) {   // block-uniform
    const unsigned nch = lkeys[q][2];
    if (nch <= 64u) continue;
    const unsigned per = (nch + NW - 1) / NW;
    const unsigned c0 = min(nch, wv * per), c1 = min(nch, c0 + per);
    float gv[K][V];
    // wave 0 finishes the key: it asks for the state rows with its share of the partials
    uint4 ra = make_uint4(0u, 0u, 0u, 0u);
    RowMeta m0{};
    bool hl = false, hx = false, hs = false;
    PreRows<V, K> pre;
    if (wv == 0) {
      ra = make_uint4(lkeys[q][3], lkeys[q][4], lkeys[q][5], lkeys[q][6]);
      if (MODE == MODE_APPLY) prefetch_state<OPT, V, LPR, K>(a, ra, g == 0, lane, D, m0, hl, pre, hx, hs);
    }
    wave_sum(lkeys[q][1], c0, c1, gv);
    __syncthreads();   // lsum of the previous key has been read
    if (g == 0) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int e0 = (lane + k * LPR) * V;
        if (e0 < D) stv<V>(lsum + (size_t)wv * D + e0, gv[k]);
      }
    }
    __syncthreads();
    if (wv == 0) {
      const unsigned nwv = (nch + per - 1) / per;   // waves that had chunks
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int e0 = (lane + k * LPR) * V;
        if (e0 < D) {
          ldv<V>(lsum + e0, gv[k]);
          for (unsigned x = 1; x < nwv; ++x) {
            float v[V];
            ldv<V>(lsum + (size_t)x * D + e0, v);
#pragma unroll
            for (int cc = 0; cc < V; ++cc) gv[k][cc] = (MODE == MODE_APPLY) ? gv[k][cc] + v[cc] : fold2(fop, gv[k][cc], v[cc]);
          }
        }
      }
      finish_key<MODE, OPT, V, LPR, K>(a, ra, g == 0, hl, m0, gv, lane, &pre, hx, hs);
    }
  }
}

// ------------------------------------------------------------------------------------------
// k_gather: out[i, :] = rows[ent_b[slot_rank[i] & SLOT_MASK]]
// ------------------------------------------------------------------------------------------
// VQ = float4 vectors per row (dim / 4) when > 0 (power of two); VQ = 0 -> generic dim
// ORDER: the same pass files every position in the sorted list (order_pos): the training lookup's last kernel
// the row a position reads: its entry's row, through w.row_map when the rows live in an exchange buffer (sharded
// lookup: entry -> dense unique index -> the record the id was sent in); skipped records read the zero row
__device__ __forceinline__ unsigned gather_row(const WsDev& w, unsigned sr) {
  if (sr == 0xFFFFFFFFu) return 0u;
  const unsigned r = w.ent_b[sr & SLOT_MASK];
  return w.row_map ? (unsigned)w.row_map[r] : r;
}

template <int VQ, bool ORDER = false>
__device__ __forceinline__ void gather_body(const TableDev& t, const WsDev& w, float* __restrict__ out,
                                            long long n) {
  // ORDER: the launch carries ITEM_BLOCKS extra blocks in front that only build the item directory — a serial
  // little job that would otherwise sit in front of the first blocks' share of the rows (small batches: the whole
  // kernel is one generation of blocks, and it ended when those blocks did)
  unsigned bid = blockIdx.x, nbk = gridDim.x;
  if (ORDER) {
    if (*reinterpret_cast<volatile unsigned*>(&t.counters[1])) return;   // a partition overflowed: nothing is valid
    if (blockIdx.x == 0 && threadIdx.x == 0) w.order[n] = HEAD_BIT;
    if (blockIdx.x < ITEM_BLOCKS) { items_body(w); return; }
    bid -= ITEM_BLOCKS; nbk -= ITEM_BLOCKS;
  }
  if constexpr (VQ > 0 && VQ <= 64) {
    // One wave takes 64 consecutive output rows per step.  Lane l resolves row l's table row id
    // (slot_rank -> ent_b: two dependent loads, 64 rows in flight per wave and no redundancy);
    // then the wave copies the rows VQ lanes per row, 64 / VQ rows per instruction, CW
    // instructions in flight, the row ids passed between lanes with ds_bpermute.
    constexpr int RW = 64 / VQ;            // rows per copy instruction
    constexpr int CW = VQ < 16 ? VQ : 16;  // copy instructions in flight
    const int lane = threadIdx.x & 63;
    const int v = lane % VQ, sub = lane / VQ;
    const long long wave = (long long)bid * (TB / 64) + (threadIdx.x >> 6);
    const long long nwaves = (long long)nbk * (TB / 64);
    // software pipeline over the wave's steps: while step i copies rows, step i+1's row ids and
    // step i+2's slots are already in flight (the three dependent hops overlap across steps)
    const long long stride = nwaves * 64;
    long long r0 = wave * 64;
    unsigned sl1 = 0, sl2 = 0, rr = 0;
    if (r0 + lane < n) {
      const unsigned sr = __builtin_nontemporal_load(&w.slot_rank[r0 + lane]);
      rr = gather_row(w, sr);
      if (ORDER) order_pos(w, n, r0 + lane, sr);
    }
    if (r0 + stride + lane < n) sl1 = __builtin_nontemporal_load(&w.slot_rank[r0 + stride + lane]);
    for (; r0 < n; r0 += stride) {
      unsigned rr1 = 0;
      if (r0 + stride + lane < n) {
        rr1 = gather_row(w, sl1);
        if (ORDER) order_pos(w, n, r0 + stride + lane, sl1);
      }
      if (r0 + 2 * stride + lane < n) sl2 = __builtin_nontemporal_load(&w.slot_rank[r0 + 2 * stride + lane]);
#pragma unroll
      for (int j0 = 0; j0 < VQ; j0 += CW) {
        float4 val[CW];
        unsigned rj[CW];
#pragma unroll
        for (int j = 0; j < CW; ++j) rj[j] = __shfl(rr, (j0 + j) * RW + sub);
#pragma unroll
        for (int j = 0; j < CW; ++j) val[j] = reinterpret_cast<const float4*>(row_ptr(t, rj[j]))[v];
#pragma unroll
        for (int j = 0; j < CW; ++j) {
          const long long ii = r0 + (j0 + j) * RW + sub;
          if (ii < n) {  // the output is not read again by this launch: streaming store, keep L2 for the rows
            float4* dst = reinterpret_cast<float4*>(out + (size_t)ii * (VQ * 4)) + v;
            __builtin_nontemporal_store(val[j].x, &dst->x); __builtin_nontemporal_store(val[j].y, &dst->y);
            __builtin_nontemporal_store(val[j].z, &dst->z); __builtin_nontemporal_store(val[j].w, &dst->w);
          }
        }
      }
      rr = rr1;
      sl1 = sl2;
    }
  } else if constexpr (VQ > 64) {
    constexpr int RPB = TB / VQ;  // rows per block per step (VQ = 128, 256)
    const int v = threadIdx.x % VQ;
    for (long long i = (long long)bid * RPB + threadIdx.x / VQ; i < n; i += (long long)nbk * RPB) {
      const unsigned sr = w.slot_rank[i];
      const unsigned r = gather_row(w, sr);
      if (ORDER && v == 0) order_pos(w, n, i, sr);
      reinterpret_cast<float4*>(out + (size_t)i * (VQ * 4))[v] = reinterpret_cast<const float4*>(row_ptr(t, r))[v];
    }
  } else {
    const int D = t.dim;
    const long long total = n * D;
    for (long long x = (long long)bid * TB + threadIdx.x; x < total; x += (long long)nbk * TB) {
      const long long i = x / D;
      const int e = (int)(x - i * D);
      const unsigned sr = w.slot_rank[i];
      if (ORDER && e == 0) order_pos(w, n, i, sr);
      out[x] = row_ptr(t, gather_row(w, sr))[e];
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

// The same op for dims 4·VQ (VQ a power of two <= 64), wave-shaped like k_gather: one wave takes 64
// consecutive ids per step, lane l probes id l (64 independent probes in flight per wave, none of them
// repeated by neighbouring lanes), then the wave copies the rows VQ lanes per row, CH copy instructions
// in flight, row ids handed over by shuffle, streaming stores (the output is not read again here).
// ids_kind: 0 int64, 1 int32, 2 (id, count) int64 pairs.  `wave` of `nwaves` waves share the rows.
// Software pipeline over the wave's steps: while step i copies its rows, the home index entries of step
// i + 1 and the ids of step i + 2 are already in flight, so a step costs one round trip, not three — the
// gather keeps the store bandwidth busy from a few waves per CU (it runs beside the partition pass).
template <int VQ, int CWMAX = 4>
__device__ __forceinline__ void goz_wave(const TableDev& t, const void* __restrict__ ids, int ids_kind,
                                         float* __restrict__ out, long long n, long long wave, long long nwaves) {
  constexpr int RW = 64 / VQ;            // rows per copy instruction
  constexpr int CW = VQ < CWMAX ? VQ : CWMAX;    // copy instructions in flight (4 with many waves per CU: the probe hop
                                                 // wants the occupancy; 8 for the few gather waves beside the partition pass)
  const int lane = threadIdx.x & 63;
  const int v = lane % VQ, sub = lane / VQ;
  const long long stride = nwaves * 64;
  auto load_key = [&](long long i) -> long long {
    if (i >= n) return EMPTY_KEY;
    return ids_kind == 1 ? (long long)reinterpret_cast<const int*>(ids)[i]
                         : reinterpret_cast<const long long*>(ids)[i << (ids_kind == 2 ? 1 : 0)];
  };
  long long r0 = wave * 64;
  if (r0 >= n) return;
  long long k1 = load_key(r0 + lane);                  // step i + 1's key (first: step 0's)
  long long k2 = load_key(r0 + stride + lane);         // step i + 2's
  unsigned long long p1 = home_of(t, k1, mix64((unsigned long long)k1));
  Entry e1 = load_entry(&t.entries[p1]);
  for (; r0 < n; r0 += stride) {
    // this step's rows: finish the probe started one step ago (row 0 reads zeros: misses, lanes past the end)
    const unsigned rr = (r0 + lane < n) ? table_find_from(t, k1, p1, e1) : 0u;
    // next step: its home entries leave now, the ids of the step after it too
    k1 = k2;
    p1 = home_of(t, k1, mix64((unsigned long long)k1));
    if (r0 + stride < n) e1 = load_entry(&t.entries[p1]);
    k2 = load_key(r0 + 2 * stride + lane);
#pragma unroll
    for (int j0 = 0; j0 < VQ; j0 += CW) {
      float4 val[CW];
      unsigned rj[CW];
#pragma unroll
      for (int j = 0; j < CW; ++j) rj[j] = __shfl(rr, (j0 + j) * RW + sub);
#pragma unroll
      for (int j = 0; j < CW; ++j) val[j] = reinterpret_cast<const float4*>(row_ptr(t, rj[j]))[v];
#pragma unroll
      for (int j = 0; j < CW; ++j) {
        const long long ii = r0 + (j0 + j) * RW + sub;
        if (ii < n) {
          float4* dst = reinterpret_cast<float4*>(out + (size_t)ii * (VQ * 4)) + v;
          __builtin_nontemporal_store(val[j].x, &dst->x); __builtin_nontemporal_store(val[j].y, &dst->y);
          __builtin_nontemporal_store(val[j].z, &dst->z); __builtin_nontemporal_store(val[j].w, &dst->w);
        }
      }
    }
  }
}
template <typename IdT, int VQ>
__global__ void __launch_bounds__(TB) k_gather_or_zeros_w(TableDev t, const IdT* __restrict__ ids,
                                                          float* __restrict__ out, long long n) {
  goz_wave<VQ>(t, ids, sizeof(IdT) == 4 ? 1 : 0, out, n, (long long)blockIdx.x * (TB / 64) + (threadIdx.x >> 6),
               (long long)gridDim.x * (TB / 64));
}

// the gather for any dim behind one entry: wave-shaped for dims 4, 8, ..., 256, else 8 lanes per row
template <int CWMAX = 4>
__device__ __forceinline__ void goz_any(const TableDev& t, const void* __restrict__ ids, int ids_kind,
                                        float* __restrict__ out, long long n, long long blk, long long nblk) {
  const int D = t.dim;
  const long long wave = blk * (blockDim.x / 64) + (threadIdx.x >> 6);
  const long long nwaves = nblk * (blockDim.x / 64);
  if ((D & 3) == 0) {  // block-uniform
    switch (D >> 2) {
      case 1: goz_wave<1, CWMAX>(t, ids, ids_kind, out, n, wave, nwaves); return;
      case 2: goz_wave<2, CWMAX>(t, ids, ids_kind, out, n, wave, nwaves); return;
      case 4: goz_wave<4, CWMAX>(t, ids, ids_kind, out, n, wave, nwaves); return;
      case 8: goz_wave<8, CWMAX>(t, ids, ids_kind, out, n, wave, nwaves); return;
      case 16: goz_wave<16, CWMAX>(t, ids, ids_kind, out, n, wave, nwaves); return;
      case 32: goz_wave<32, CWMAX>(t, ids, ids_kind, out, n, wave, nwaves); return;
      case 64: goz_wave<64, CWMAX>(t, ids, ids_kind, out, n, wave, nwaves); return;
      default: break;
    }
  }
  const int lane8 = threadIdx.x & 7;
  const long long gpb = blockDim.x / 8;
  for (long long i = blk * gpb + (threadIdx.x >> 3); i < n; i += nblk * gpb) {
    const long long key = ids_kind == 1 ? (long long)reinterpret_cast<const int*>(ids)[i]
                                        : reinterpret_cast<const long long*>(ids)[i << (ids_kind == 2 ? 1 : 0)];
    const unsigned r = table_find(t, key);
    const float* row = row_ptr(t, r);
    float* o = out + (size_t)i * D;
    if ((D & 3) == 0) {
      for (int q = lane8; q < (D >> 2); q += 8)
        reinterpret_cast<float4*>(o)[q] = reinterpret_cast<const float4*>(row)[q];
    } else {
      for (int e = lane8; e < D; e += 8) o[e] = row[e];
    }
  }
}

// ---- kernel entry points: single table (by value) and many tables (descriptor array) ------------
struct MultiDesc {
  WsDev w;
  PartArgs a;            // a.tv is the table of this entry (lookup) / the var table (apply)
  const void* ids;
  const int* counts;
  float* out;            // lookup output rows
  long long n;
};

template <int MODE>
__global__ void __launch_bounds__(TBK, 4) k_part_keys(WsDev w, PartArgs a) { part_keys_body<MODE>(w, a); }
template <int MODE, int OPT, int V, int LPR, int K>
__global__ void __launch_bounds__(TBS, (K == 1 ? 5 : 1)) k_apply(WsDev w, PartArgs a) { apply_body<MODE, OPT, V, LPR, K>(w, a); }
template <int MODE, int OPT, int V, int LPR, int K>
__global__ void __launch_bounds__(TBF) k_apply_fin(WsDev w, PartArgs a) { apply_fin_body<MODE, OPT, V, LPR, K>(w, a); }
template <int VQ, bool ORDER>
__global__ void __launch_bounds__(TB) k_gather(TableDev t, WsDev w, float* __restrict__ out, long long n) {
  gather_body<VQ, ORDER>(t, w, out, n);
}

template <bool FIRST, typename IdT>
__global__ void __launch_bounds__(TBT) k_tile_multi(const MultiDesc* __restrict__ descs) {
  const MultiDesc& m = descs[blockIdx.y];
  if (blockIdx.x >= m.w.ntiles) return;
  tile_body<FIRST, IdT>(m.w, reinterpret_cast<const IdT*>(m.ids), m.counts, m.n, m.a.det);
}
template <int MODE>
__global__ void __launch_bounds__(TBK, 4) k_part_keys_multi(const MultiDesc* __restrict__ descs) {
  const MultiDesc& m = descs[blockIdx.y];
  if (blockIdx.x >= m.w.P || m.n == 0) return;
  part_keys_body<MODE>(m.w, m.a);
}
__global__ void __launch_bounds__(TB) k_order_multi(const MultiDesc* __restrict__ descs) {
  const MultiDesc& m = descs[blockIdx.y];
  if (m.n == 0) return;
  order_body(m.a.tv, m.w, m.n);
}
template <int MODE, int OPT, int V, int LPR, int K>
__global__ void __launch_bounds__(TBS, (K == 1 ? 4 : 1)) k_apply_multi(const MultiDesc* __restrict__ descs) {
  const MultiDesc& m = descs[blockIdx.y];
  if (m.n == 0) return;
  apply_body<MODE, OPT, V, LPR, K>(m.w, m.a);
}
template <int MODE, int OPT, int V, int LPR, int K>
__global__ void __launch_bounds__(TBF) k_apply_fin_multi(const MultiDesc* __restrict__ descs) {
  const MultiDesc& m = descs[blockIdx.y];
  if (m.n == 0) return;
  apply_fin_body<MODE, OPT, V, LPR, K>(m.w, m.a);
}
template <int VQ>
__global__ void __launch_bounds__(TB) k_gather_multi(const MultiDesc* __restrict__ descs) {
  const MultiDesc& m = descs[blockIdx.y];
  gather_body<VQ>(m.a.tv, m.w, m.out, m.n);
}

// BatchKvVariableGatherOrZerosV2 (kernels/kv_variable_ops.cc:431-470): N tables, N id lists, N
// outputs — the reference loops over the tables; here ONE launch covers them all (blockIdx.y =
// table, tables may differ in dim), which is what a 26-feature serving step needs.
struct BatchGatherDesc {
  TableDev t;
  const void* ids;
  float* out;
  long long n;
  int ids_int32;
};
__global__ void __launch_bounds__(TB) k_batch_gather_or_zeros(const BatchGatherDesc* __restrict__ descs) {
  const BatchGatherDesc& d = descs[blockIdx.y];
  goz_any(d.t, d.ids, d.ids_int32 ? 1 : 0, d.out, d.n, blockIdx.x, gridDim.x);
}

__global__ void k_store_count(const unsigned* ctr, long long* out) { *out = (long long)*ctr; }

// kv_dedup_segment_sum: inverse[i] = dense unique index of input position i
__global__ void k_dedup_inverse(WsDev w, long long n, int* inverse) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    inverse[i] = (int)w.ent_b[w.slot_rank[i] & SLOT_MASK];
}

// ------------------------------------------------------------------------------------------
// multi-GPU routing: stable-by-tile counting sort of ids by owner rank = floor_mod(id, world)
// (kernels/utility.h:90-107).  world <= 64.  hist is [world][ntiles] (owner-major for the scan).
// ------------------------------------------------------------------------------------------
constexpr int RT = 1024;  // ids per routing tile (256 threads x 4)
constexpr int MAXW = 64;

// rule 0 (default): (mix64(id) >> 32) % world — balanced whatever the ids look like; rule 1: floor_mod(id, world), the
// reference's `ids % num_shards` (python/ops/embedding_ops.py:121-127), for checkpoint compatibility
__device__ __forceinline__ unsigned owner_rank(long long id, int world, int rule) {
  // (the HIGH half of the hash: the index's home slot is mix64(key) & mask — with the low bits every key of a rank
  // would share its low home-slot bits and the probe chains of a rank's table would cluster)
  if (rule == 0) return (unsigned)((mix64((unsigned long long)id) >> 32) % (unsigned long long)world);
  long long m = id % world;
  return (unsigned)(m < 0 ? m + world : m);
}

template <typename IdT>
__global__ void __launch_bounds__(TB) k_owner_hist(const IdT* __restrict__ ids, long long n, int world, int rule,
                                                   unsigned ntiles, unsigned* __restrict__ hist,
                                                   const long long* __restrict__ n_dev) {
  if (n_dev) n = min(n, *n_dev);   // the list's length is still on the device (kv_unique without a sync)
  __shared__ unsigned h[MAXW];
  if (threadIdx.x < MAXW) h[threadIdx.x] = 0;
  __syncthreads();
  const long long base = (long long)blockIdx.x * RT;
#pragma unroll
  for (int k = 0; k < RT / TB; ++k) {
    const long long i = base + k * TB + threadIdx.x;
    if (i < n) atomicAdd(&h[owner_rank(load_id(ids, (size_t)i), world, rule)], 1u);
  }
  __syncthreads();
  if ((int)threadIdx.x < world) hist[(size_t)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

// one block: exclusive scan of hist in (owner, tile) order -> base offsets; counts[w] = ids owned by w
__global__ void __launch_bounds__(1024) k_owner_scan(unsigned* __restrict__ hist, unsigned total,
                                                     unsigned ntiles, int world, long long* __restrict__ counts) {
  __shared__ unsigned wtot[17];
  const unsigned per = (total + 1023) / 1024;
  const unsigned b0 = min(total, threadIdx.x * per), b1 = min(total, b0 + per);
  unsigned sum = 0;
  for (unsigned i = b0; i < b1; ++i) sum += hist[i];
  unsigned tot;
  unsigned run = block_excl_scan<16>(sum, wtot, &tot);
  for (unsigned i = b0; i < b1; ++i) { const unsigned c = hist[i]; hist[i] = run; run += c; }
  __syncthreads();
  if ((int)threadIdx.x < world) {
    const unsigned lo = hist[(size_t)threadIdx.x * ntiles];
    const unsigned hi = ((int)threadIdx.x + 1 < world) ? hist[(size_t)(threadIdx.x + 1) * ntiles] : tot;
    counts[threadIdx.x] = (long long)hi - (long long)lo;
  }
}

template <typename IdT>
__global__ void __launch_bounds__(TB) k_owner_scatter(const IdT* __restrict__ ids, long long n, int world, int rule,
                                                      unsigned ntiles, const unsigned* __restrict__ base_off,
                                                      long long* __restrict__ out_ids, int* __restrict__ perm,
                                                      const long long* __restrict__ n_dev,
                                                      const int* __restrict__ counts_in,
                                                      long long* __restrict__ pairs_out, int* __restrict__ pos_out) {
  if (n_dev) n = min(n, *n_dev);
  __shared__ unsigned h[MAXW];
  if ((int)threadIdx.x < world) h[threadIdx.x] = base_off[(size_t)threadIdx.x * ntiles + blockIdx.x];
  __syncthreads();
  const long long base = (long long)blockIdx.x * RT;
#pragma unroll
  for (int k = 0; k < RT / TB; ++k) {
    const long long i = base + k * TB + threadIdx.x;
    if (i < n) {
      const long long id = load_id(ids, (size_t)i);
      const unsigned pos = atomicAdd(&h[owner_rank(id, world, rule)], 1u);
      out_ids[pos] = id;
      perm[pos] = (int)i;
      // optional extras of the sharded lookup: the exchange payload (id, occurrence count) in
      // bucket order, and where input position i went (the inverse of perm)
      if (pairs_out) { pairs_out[2 * (size_t)pos] = id; pairs_out[2 * (size_t)pos + 1] = counts_in ? (long long)counts_in[i] : 1ll; }
      if (pos_out) pos_out[i] = (int)pos;
    }
  }
}
// The sharded lookup's exchange payload in FIXED-CAPACITY segments (no size collective, no host sync): owner d's
// segment is seg[d][0 .. C]: record 0 = header {pairs in the segment, 0}, records 1 .. = (id, occurrence count).
// A record whose count is 0 is skipped by the owner's tile pass, so padding costs nothing but its bytes.  slot_of[u]
// = where unique id u went (its row comes back at the same place).  More than C ids for one owner: the extra
// ones are dropped and *overflow is raised (the host doubles C; hashed ownership keeps this from happening).
__global__ void __launch_bounds__(TB) k_owner_scatter_fixed(const long long* __restrict__ ids, const int* __restrict__ cnts,
                                                            long long n, int world, int rule,
                                                            unsigned ntiles, const unsigned* __restrict__ base_off,
                                                            unsigned C, long long* __restrict__ seg, int* __restrict__ slot_of,
                                                            unsigned* __restrict__ overflow) {
  __shared__ unsigned h[MAXW];
  if ((int)threadIdx.x < world)   // rank inside the owner's bucket = global offset - the bucket's start
    h[threadIdx.x] = base_off[(size_t)threadIdx.x * ntiles + blockIdx.x] - base_off[(size_t)threadIdx.x * ntiles];
  __syncthreads();
  const long long base = (long long)blockIdx.x * RT;
#pragma unroll
  for (int k = 0; k < RT / TB; ++k) {
    const long long i = base + k * TB + threadIdx.x;
    if (i < n && cnts[i] > 0) {   // the list has gaps (sparse unique numbers): a count of 0 names no key
      const long long id = ids[i];
      const unsigned d = owner_rank(id, world, rule);
      const unsigned r = atomicAdd(&h[d], 1u);
      if (r < C) {
        const size_t slot = (size_t)d * (C + 1) + 1 + r;
        seg[2 * slot] = id;
        seg[2 * slot + 1] = (long long)cnts[i];
        slot_of[i] = (int)slot;
      } else {
        slot_of[i] = 0;        // record 0 is a header: its "row" is never a real one
        atomicExch(overflow, 1u);
      }
    }
  }
}
// The same in two launches instead of four (hist + scan + scatter in one): a block counts its tile's ids per owner in LDS, takes
// its place in every owner's segment with one atomic per owner, writes its records.  The order of the records
// inside a segment then depends on block timing — it decides nothing but the owner's row numbering — so the
// deterministic mode keeps the four-kernel version above.  gcount: [world] segment fill, zero between batches.
__global__ void __launch_bounds__(TB) k_owner_route_fixed(const long long* __restrict__ ids, const int* __restrict__ cnts,
                                                          long long n, int world, int rule, unsigned C,
                                                          long long* __restrict__ seg, int* __restrict__ slot_of,
                                                          unsigned* __restrict__ overflow, unsigned* __restrict__ gcount) {
  __shared__ unsigned h[MAXW], base[MAXW];
  if (threadIdx.x < MAXW) h[threadIdx.x] = 0;
  __syncthreads();
  const long long b0 = (long long)blockIdx.x * RT;
  long long id[RT / TB];
  unsigned d[RT / TB], r[RT / TB];
  if (b0 < n) {
#pragma unroll
    for (int k = 0; k < RT / TB; ++k) {
      const long long i = b0 + k * TB + threadIdx.x;
      d[k] = 0xFFFFFFFFu;
      if (i < n && cnts[i] > 0) {   // gaps of the sparse unique numbering carry a count of 0
        id[k] = ids[i];
        d[k] = owner_rank(id[k], world, rule);
        r[k] = atomicAdd(&h[d[k]], 1u);
      }
    }
    __syncthreads();
    if ((int)threadIdx.x < world) base[threadIdx.x] = h[threadIdx.x] ? atomicAdd(&gcount[threadIdx.x], h[threadIdx.x]) : 0u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RT / TB; ++k) {
      const long long i = b0 + k * TB + threadIdx.x;
      if (d[k] != 0xFFFFFFFFu) {
        const unsigned rr = base[d[k]] + r[k];
        if (rr < C) {
          const size_t slot = (size_t)d[k] * (C + 1) + 1 + rr;
          seg[2 * slot] = id[k];
          seg[2 * slot + 1] = (long long)cnts[i];
          slot_of[i] = (int)slot;
        } else {
          slot_of[i] = 0;
          atomicExch(overflow, 1u);
        }
      }
    }
  }
}
// ... and its headers, from the fill counters, which it zeroes for the next batch.  (A last-block-done epilogue in
// the kernel above would need a device-scope release fence per block; on gfx950 that writes the XCD's L2 back and
// cost 60 us behind the partition pass.)
// need[0] (may be null) = the largest segment this batch WANTED, capped or not: what peer_capacity would have had to be
// (one block: thread 0 clears it, every owner's thread raises it)
// uhint (pinned host word, may be null) = the batch's distinct ids: the next route picks its partition count by it
__global__ void k_seg_headers_take(unsigned* __restrict__ gcount, int world, unsigned C, long long* __restrict__ seg,
                                   unsigned* __restrict__ need, unsigned* __restrict__ uhint) {
  __shared__ unsigned tot;
  const int d = threadIdx.x;
  if (d == 0) { if (need) *need = 0u; tot = 0u; }
  __syncthreads();
  if (d < world) {
    const unsigned c = gcount[d];
    gcount[d] = 0;
    seg[2 * (size_t)d * (C + 1)] = c < C ? c : C;
    seg[2 * (size_t)d * (C + 1) + 1] = 0;
    if (need) atomicMax(need, c);
    atomicAdd(&tot, c);
  }
  __syncthreads();
  if (d == 0 && uhint) __hip_atomic_store(uhint, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// the segments' headers {records in the segment (at most C), 0}
__global__ void k_seg_headers(const long long* __restrict__ counts, int world, unsigned C, long long* __restrict__ seg,
                              unsigned* __restrict__ need) {
  const int d = threadIdx.x;
  if (need && d == 0) *need = 0u;
  __syncthreads();
  if (d < world) {
    seg[2 * (size_t)d * (C + 1)] = counts[d] < (long long)C ? counts[d] : (long long)C;
    seg[2 * (size_t)d * (C + 1) + 1] = 0;
    if (need) atomicMax(need, (unsigned)(counts[d] < 0x7FFFFFFFll ? counts[d] : 0x7FFFFFFFll));
  }
}
// k_owner_hist over the sparse unique list of the sharded route (entries with a count of 0 name no key)
__global__ void __launch_bounds__(TB) k_owner_hist_u32(const long long* __restrict__ ids, const int* __restrict__ cnts, long long n,
                                                       int world, int rule, unsigned ntiles, unsigned* __restrict__ hist) {
  __shared__ unsigned h[MAXW];
  if (threadIdx.x < MAXW) h[threadIdx.x] = 0;
  __syncthreads();
  const long long base = (long long)blockIdx.x * RT;
#pragma unroll
  for (int k = 0; k < RT / TB; ++k) {
    const long long i = base + k * TB + threadIdx.x;
    if (i < n && cnts[i] > 0) atomicAdd(&h[owner_rank(ids[i], world, rule)], 1u);
  }
  __syncthreads();
  if ((int)threadIdx.x < world) hist[(size_t)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}
// ---------------------------------------------------------------------------------------------
// embedding_lookup_sparse (python/ops/embedding_ops.py:279-441) fused behind the lookup index:
//   k_seg_offsets   CSR offsets of the sorted segment ids: off[s] = first position of segment s
//   k_seg_combine   out[s] = combine_j( w_j * rows[row(id_j)] ) over the segment's positions, in
//                   position order (tf.segment_sum order); mean: / sum w, sqrtn: / sqrt(sum w^2)
// Segment ids outside [prev, num_segments) are clamped (memory safety only; TF rejects them).
template <typename SegT>
__global__ void __launch_bounds__(TB) k_seg_offsets(const SegT* __restrict__ seg, long long n, long long nseg,
                                                    unsigned* __restrict__ off) {
  for (long long i = (long long)blockIdx.x * TB + threadIdx.x; i <= n; i += (long long)gridDim.x * TB) {
    long long prev = i > 0 ? (long long)seg[i - 1] : -1;
    long long cur = i < n ? (long long)seg[i] : nseg;
    prev = prev < -1 ? -1 : (prev > nseg ? nseg : prev);
    cur = cur < 0 ? 0 : (cur > nseg ? nseg : cur);
    for (long long sgi = prev + 1; sgi <= cur; ++sgi) off[sgi] = (unsigned)i;
  }
}

// VQ = float4 lanes per row (dim / 4, power of two <= 64) or 0 = one thread per element.
// has_w: sp_weights given (the reference multiplies, sums and divides by the weight sums);
// otherwise tf.sparse_segment_{sum,mean,sqrt_n} (empty segment -> zeros).
template <int VQ>
__global__ void __launch_bounds__(TB) k_seg_combine(TableDev t, WsDev w, const unsigned* __restrict__ off,
                                                    const float* __restrict__ wts, long long nseg,
                                                    int combiner, float* __restrict__ out) {
  const int D = t.dim;
  constexpr int LPS = VQ > 0 ? VQ : 1;          // lanes per segment
  const int v = threadIdx.x % LPS;
  const long long g0 = ((long long)blockIdx.x * TB + threadIdx.x) / LPS;
  const long long gstride = (long long)gridDim.x * TB / LPS;
  for (long long sgi = g0; sgi < nseg; sgi += gstride) {
    const unsigned lo = off[sgi], hi = off[sgi + 1];
    float wsum = 0.f, w2 = 0.f;
    if constexpr (VQ > 0) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      // SU ids of the segment at a time: their three dependent hops (slot -> row id -> row) overlap;
      // the sums are still taken in id order
      constexpr int SU = 4;
      for (unsigned j = lo; j < hi; j += SU) {
        unsigned sl[SU], r[SU];
        float wj[SU];
        float4 x[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
          const bool ok = j + u < hi;
          sl[u] = ok ? (w.slot_rank[j + u] & SLOT_MASK) : 0xFFFFFFFFu;
          wj[u] = ok ? (wts ? wts[j + u] : 1.f) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) r[u] = sl[u] != 0xFFFFFFFFu ? w.ent_b[sl[u]] : 0u;
#pragma unroll
        for (int u = 0; u < SU; ++u) x[u] = reinterpret_cast<const float4*>(row_ptr(t, r[u]))[v];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
          if (sl[u] == 0xFFFFFFFFu) continue;
          acc.x += x[u].x * wj[u]; acc.y += x[u].y * wj[u]; acc.z += x[u].z * wj[u]; acc.w += x[u].w * wj[u];
          wsum += wj[u]; w2 += wj[u] * wj[u];
        }
      }
      float den = 1.f;
      if (combiner == 1) den = wsum; else if (combiner == 2) den = sqrtf(w2);
      if (combiner != 0 && (wts || hi > lo)) { acc.x /= den; acc.y /= den; acc.z /= den; acc.w /= den; }
      reinterpret_cast<float4*>(out + (size_t)sgi * D)[v] = acc;
    } else {
      for (int e = 0; e < D; ++e) {
        float acc = 0.f;
        wsum = 0.f; w2 = 0.f;
        for (unsigned j = lo; j < hi; ++j) {
          const unsigned r = w.ent_b[w.slot_rank[j] & SLOT_MASK];
          const float wj = wts ? wts[j] : 1.f;
          acc += row_ptr(t, r)[e] * wj;
          wsum += wj; w2 += wj * wj;
        }
        float den = 1.f;
        if (combiner == 1) den = wsum; else if (combiner == 2) den = sqrtf(w2);
        if (combiner != 0 && (wts || hi > lo)) acc /= den;
        out[(size_t)sgi * D + e] = acc;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_take_rows: out[i] = src[idx[i]] (SCATTER = 0) or out[idx[i]] = src[i] (SCATTER = 1) over rows of
// `nu` units of type U (float4 when the row is a multiple of 16 bytes).  The exchange's permute /
// un-permute / expand steps of the sharded path.
template <typename U, int SCATTER>
__global__ void __launch_bounds__(TB) k_take_rows(const U* __restrict__ src, const int* __restrict__ idx,
                                                  long long n, unsigned nu, int sh, U* __restrict__ out,
                                                  const int* __restrict__ idx_outer = nullptr) {
  const long long total = n * nu;
  const long long stride = (long long)gridDim.x * TB;
  for (long long x = (long long)blockIdx.x * TB + threadIdx.x; x < total; x += stride) {
    long long i;
    unsigned e;
    if (sh >= 0) { i = x >> sh; e = (unsigned)(x & (nu - 1)); }
    else { i = x / nu; e = (unsigned)(x - i * nu); }
    const long long j = idx_outer ? idx[idx_outer[i]] : idx[i];   // two-level gather: src[idx[idx_outer[i]]]
    if (SCATTER) out[j * nu + e] = src[x];
    else out[x] = src[j * nu + e];
  }
}

