// kv_fused.h — the entry-list pipeline of the training step (included after kv_kernels.h).
//
// The sorted-position pipeline of kv_kernels.h puts three launches in front of the lookup's output rows (tile pass,
// partition pass with two dependent random hops per key, gather that files 1 M positions with scattered stores) and
// walks 1 M random gradient rows plus a second kernel for the keys that span chunks.  Measured on the box
// (profiles/r03_calibration.txt): random 128-B rows read at 5 TB/s whatever the order, the per-key state update of
// GroupAdam costs 27-35 us for 109 k keys by itself, so what is left to remove is everything that is not a row.
// This pipeline keeps the tile and the partition idea but changes who does what:
//
//   k_ltile   one block per TILE ids: LDS dedup; ONE index probe per distinct key of the tile, requested as soon as
//             the keys are known and completed behind the counting sort (a key absent from the table is inserted
//             here: a 64-bit CAS on the index entry decides between tiles, the loser needs no row — the row of a new
//             key is a function of (key, seed)); the tile's entries {key, occurrences | count, row word, slot-row
//             hint, source}; the rows of the entries that occur more than once, in entry order (mrow: position,
//             epart row, head flag — built in LDS, coalesced); and the OUTPUT ROWS of its 2048 positions.  The
//             lookup's result is complete after this one launch.
//   k_part2   one block per hash partition over the tiles' entries (5 MB, not 1 M positions): exact occurrence
//             counts, frequency word + day + flags of every key with ONE hop (the row came with the entry), rows of
//             new keys initialised, key records sorted into classes (1 source, 2 sources, 3 .. LCOLD, hot), the ENTRY
//             LIST — a key's entries contiguous, each naming its source — and the work items.  Nothing in the
//             lookup's output depends on it, so a lookup that hands out a batch token leaves it pending: the
//             optimizer apply of the same batch runs it first, or whatever op the table sees next does.
//   k_tsum    per tile: the gradient rows of every entry with more than one occurrence are summed into epart by a
//             segmented reduction over mrow (all reads inside the tile's 256 KB of gradient rows, every wave the
//             same number of rows whatever the skew).  An entry with one occurrence IS its gradient row.  Its
//             first ITEM_BLOCKS blocks turn the partitions' work items into the dense directory.
//   k_apply2  over the entry list: a key has at most one entry per tile, so the hottest key of 1 M ids is 489
//             sources instead of 180 k positions — one chunk, no k_apply_fin.  1024-thread blocks take items from
//             an LDS ticket, the next item's records are requested while the current one is worked on, a cold
//             batch holds keys of one class so that its loads are unconditional.
//
// Summation order: inside an entry by rank (LDS-atomic arrival; input order in deterministic mode), then the key's
// entries in list order (arrival in k_part2; (tile, key) order in deterministic mode): a fixed tree given those two.
#pragma once

// ------------------------------------------------------------------------------------------
// find-or-insert from a TILE: several tiles may meet the same absent key at once
// ------------------------------------------------------------------------------------------
// Index entry states of a key: {EMPTY,0,0} -> {key,0,0} (claimed, row not published yet) -> {key,row,HINT_NEW};
// a deleted key's {key,ROW_TOMB,-} -> {key,0,-} -> {key,row,HINT_NEW}.  Whoever wins the CAS allocates the row and
// publishes it; every other tile that meets the key in this launch sees one of the "new" states (or a stale cached
// line, which the CAS resolves) and reports NEW_BIT with or without the row — k_part2 takes the row from the
// entries that know it.  The row's contents and its RowMeta are written by k_part2 (the key's single owner there).
__device__ __forceinline__ unsigned tile_insert(const TableDev& t, long long key) {
  Entry* slot;
  unsigned long long stored;
  unsigned long long p = 0;
  const bool sentinel = key == EMPTY_KEY;
  if (sentinel) { slot = &t.entries[t.mask + 1]; stored = 0ull; }
  else { stored = (unsigned long long)key; p = mix64((unsigned long long)key) & t.mask; slot = &t.entries[p]; }
  for (;;) {
    const Entry e = load_entry(slot);
    if ((unsigned long long)e.key == stored && !(sentinel && e.key == EMPTY_KEY)) {
      if (e.row == ROW_TOMB) {   // deleted earlier: the entry is still this key's — one tile gives it a row again
        if (atomicCAS(&slot->row, ROW_TOMB, 0u) == ROW_TOMB) break;
        return NEW_BIT;
      }
      if (e.row == 0u || e.hint == HINT_NEW) return NEW_BIT | e.row;
      return e.row;              // (a hit that raced with nothing: the caller's probe would have found it)
    }
    if (e.key == EMPTY_KEY) {
      const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&slot->key),
                                               (unsigned long long)EMPTY_KEY, stored);
      if (old == (unsigned long long)EMPTY_KEY) break;   // claimed
      if (old == stored) return NEW_BIT;                 // another tile is inserting the same key right now
      if (sentinel) continue;                            // (cannot happen: only the sentinel key lives there)
    } else if (sentinel) {
      return NEW_BIT;   // unreachable: entries[cap] holds EMPTY or 0
    }
    if (!sentinel) { p = (p + 1) & t.mask; slot = &t.entries[p]; }
  }
  // claimed: allocate the row (rows released by Delete first) and publish {row, HINT_NEW} with one 8-byte store
  unsigned r = 0;
  bool have = false;
  if (t.free_rows) {
    const int f = atomicSub(reinterpret_cast<int*>(&t.counters[2]), 1);
    if (f > 0) { r = t.free_rows[f - 1]; have = true; }
    else atomicAdd(reinterpret_cast<int*>(&t.counters[2]), 1);
  }
  if (!have) {
    r = atomicAdd(&t.counters[0], 1u);
    if (r >= t.max_rows) { raise_error(t, 1u); return NEW_BIT; }
  }
  *reinterpret_cast<uint2*>(&slot->row) = make_uint2(r, HINT_NEW);
  return NEW_BIT | r;
}

// ------------------------------------------------------------------------------------------
// k_ltile
// ------------------------------------------------------------------------------------------
// (Barriers: lds_barrier, kv_device.h — the probes and the id loads stay in flight across them; the deterministic
// mode's re-read of ent_key through global memory keeps the full barrier.)
// The position whose LDS insert created a key's slot is the key's WINNER: it probes the table for the key (the
// request leaves right behind the hash insert and is collected behind the counting sort), takes the key's place in
// the partition sort and writes its entry.  No list of occupied slots is built.
struct LtSmem {
  long long* lkeys;        // [LS + 1] (slot LS: the key that equals EMPTY_KEY); dead once the hash insert is done:
  unsigned* lrow;          //   [LS + 1] row word of the slot's key           \  live in lkeys' storage
  unsigned* escan;         //   [TILE + 1] per entry: packed prefix (below)   /
  unsigned* lcnt;          // [LS + 1] occurrences of the slot's key
  unsigned short* lpos;    // [LS + 1] entry number of the slot's key
  unsigned* mr;            // [TILE] the tile's mrow image
  unsigned short* lrun;    // [TILE + 1] deterministic mode: running count per entry
  unsigned* hist;          // [MAX_P + 1] per partition: entries (low 16) | positions (high 16); then the entries' frequency sums
  unsigned* wtot;          // [8]
  // bucket mode (no partition sort, see ltile_body): per hash slot instead of per entry
  unsigned* lpre;          //   [LS + 1] packed prefix of the slot's key (in lkeys' storage, behind lrow — where escan lives otherwise)
  unsigned* lfq;           //   [LS + 1] frequency sum of the slot's key (lookups with counts; in lpos / lrun / hist's storage)
};
static_assert((size_t)(LS + 1) * 8 >= (size_t)(LS + 1) * 4 + 16 + (size_t)(TILE + 1) * 4, "aliases fit");
static_assert((((size_t)(LS + 1) * 2 + 15) & ~(size_t)15) + (((size_t)(TILE + 1) * 2 + 15) & ~(size_t)15) + (size_t)(MAX_P + 1) * 4 >=
              (size_t)(LS + 1) * 4, "lfq fits lpos + lrun + hist");

__host__ __device__ inline size_t ltile_smem_bytes() {
  size_t b = (size_t)(LS + 1) * 8 + 16;   // lkeys
  b += (size_t)(LS + 1) * 4 + 16;         // lcnt
  b += (size_t)(LS + 1) * 2 + 16;         // lpos
  b += (size_t)TILE * 4 + 16;             // mr
  b += (size_t)(TILE + 1) * 2 + 16;       // lrun
  b += (size_t)(MAX_P + 1) * 4 + 16;      // hist
  b += 64;                                // wtot
  return b;
}
__device__ __forceinline__ LtSmem carve_ltile(char* base) {
  LtSmem s;
  auto take = [&](size_t bytes) { char* p = base; base += (bytes + 15) & ~(size_t)15; return p; };
  char* k0 = take((size_t)(LS + 1) * 8 + 16);
  s.lkeys = reinterpret_cast<long long*>(k0);
  s.lrow = reinterpret_cast<unsigned*>(k0);
  s.escan = reinterpret_cast<unsigned*>(k0 + (((size_t)(LS + 1) * 4 + 15) & ~(size_t)15));
  s.lpre = s.escan;
  s.lcnt = reinterpret_cast<unsigned*>(take((size_t)(LS + 1) * 4));
  s.mr = reinterpret_cast<unsigned*>(take((size_t)TILE * 4));
  s.lpos = reinterpret_cast<unsigned short*>(take((size_t)(LS + 1) * 2));
  s.lfq = reinterpret_cast<unsigned*>(s.lpos);
  s.lrun = reinterpret_cast<unsigned short*>(take((size_t)(TILE + 1) * 2));
  s.hist = reinterpret_cast<unsigned*>(take((size_t)(MAX_P + 1) * 4));
  s.wtot = reinterpret_cast<unsigned*>(take(64));
  return s;
}

// packed per-entry scan word, over the entries with more than one occurrence: their rows (bits 0..11, sum <= 2048) |
// their number (bits 12..22)
constexpr unsigned ES_POS = 0xFFFu, ES_NSH = 12;

// One entry of a partition's BUCKET (bucket mode): what k_papply needs of a tile's distinct key, in one 32-byte record
struct __attribute__((aligned(16))) BktRec {
  long long key;
  unsigned ea;     // occurrences in the tile (low 16) | saturating frequency count (high 16)
  unsigned rw;     // row word (NEW_BIT: inserted by this batch)
  unsigned hint;   // slot-row hint
  unsigned src;    // the entry's source: its input position, or EP_TAG | its row of epart
  unsigned pad[2];
};
static_assert(sizeof(BktRec) == 32, "BktRec layout");

// VQ = float4 per row (power of two <= 64); GATHER: copy the rows of the tile's positions to `out`.
// BUCKET: no counting sort by partition.  The tile appends each distinct key's record to its partition's bucket in
// global memory (one returning atomic per entry on the bucket's cursor, in flight together with the index probe), so
// the partition pass reads its entries as ONE contiguous stream — no toff directory, no per-tile segments, no binary
// search — and the tile pass loses its partition histogram, the scan over it and the toff row.  The tile-local
// numbering that mrow / epart need (the entries with more than one occurrence) comes from one block scan over the
// winners in thread order.  Not for the deterministic mode (the order of a bucket is arrival order), nor the
// (id, count) pair input of the sharded serve.
// NOTABLE: no table behind the batch (the sharded route's index of the local ids): no probes, no inserts — the entries carry
// keys, counts and sources only.
template <typename IdT, int VQ, bool GATHER, bool BUCKET = false, bool NOTABLE = false>
__device__ __forceinline__ void ltile_body(const TableDev& t, const WsDev& w, const IdT* __restrict__ ids,
                                           const int* __restrict__ counts, long long n, int det,
                                           float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  LtSmem sm = carve_ltile(smem_raw);
  __shared__ unsigned lsent;

  const int tid = threadIdx.x;
  const unsigned tile = blockIdx.x;
  const long long base = (long long)tile * TILE;
  const unsigned P = w.P;
  constexpr bool PAIRS = std::is_same<IdT, IdCount>::value;
  const bool has_counts = PAIRS || counts != nullptr;
  KV_STAMP(0);
  KV_STAMP_HW(10);

  // the tile's ids: every load unconditional (a position past the end re-reads the last id) so that they are all
  // in flight together
  long long kreg[IPT];
  unsigned creg[IPT];
  unsigned there = 0;   // bit k: position k * TBT + tid holds an id
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    const long long i = base + (long long)k * TBT + tid;
    if (i < n) there |= 1u << k;
    const long long ic = i < n ? i : n - 1;
    if constexpr (PAIRS) kreg[k] = load_id(((unsigned)ic - w.self_lo < w.self_len) ? static_cast<const IdT*>(w.ids_self) : ids, (size_t)ic);
    else kreg[k] = load_id(ids, (size_t)ic);
    creg[k] = 1;
  }
  if constexpr (PAIRS) {
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
      const long long i = base + (long long)k * TBT + tid;
      const long long ic = i < n ? i : n - 1;
      const long long ci = (((unsigned)ic - w.self_lo < w.self_len) ? static_cast<const IdT*>(w.ids_self) : ids)[ic].count;
      creg[k] = (unsigned)(unsigned short)(ci < 65535 ? ci : 65535);
    }
  } else if (counts != nullptr) {
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
      const long long i = base + (long long)k * TBT + tid;
      const int ci = counts[i < n ? i : n - 1];   // SaturateMaxFrequency(int32) -> uint16 (utility.h:57-59)
      creg[k] = (unsigned)(unsigned short)(ci < 65535 ? ci : 65535);
    }
  }
  for (int s = tid; s <= LS; s += TBT) { sm.lkeys[s] = EMPTY_KEY; sm.lcnt[s] = 0; }
  if constexpr (BUCKET) {
    if (has_counts) for (int s = tid; s <= LS; s += TBT) sm.lfq[s] = 0;
    // the bucket cursors of the NEXT batch (the other parity: nobody touches them during this launch)
    { const unsigned pz = tile * TBT + tid; if (pz < (unsigned)(MAX_P * NXCD)) w.bcnt_other[(size_t)pz * BCNT_STRIDE] = 0; }
    if (w.ntiles * TBT < (unsigned)(MAX_P * NXCD) && tile == 0)
      for (unsigned pz = w.ntiles * TBT + tid; pz < (unsigned)(MAX_P * NXCD); pz += TBT) w.bcnt_other[(size_t)pz * BCNT_STRIDE] = 0;
  } else {
    for (unsigned p = tid; p <= P; p += TBT) sm.hist[p] = 0;
  }
  if (tid == 0) lsent = 0;
  if (tile == 0 && tid == 0 && t.err_host)   // the distinct keys the previous index pass counted: a hint for the host (partitions)
    __hip_atomic_store(t.err_host + 1, w.ctr[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  lds_barrier();
  if (tile == 0 && tid < 8) w.ctr[tid] = 0;
  if constexpr (PAIRS) {
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
      const long long i = base + (long long)k * TBT + tid;
      if (!((there >> k) & 1u)) continue;
      bool ok = creg[k] != 0u;
      if (ok && w.seg_cap) {   // fixed-capacity exchange segments: record 0 is the header, records past its count are stale
        const long long r = i % w.seg_cap;
        const long long ih = i - r;   // the segment's header
        ok = r >= 1 && r <= (((unsigned)ih - w.self_lo < w.self_len) ? static_cast<const IdT*>(w.ids_self) : ids)[ih].id;
      }
      if (!ok) there &= ~(1u << k);
    }
  }
  lds_barrier();
  KV_STAMP(7);

  // ---- phase 1: LDS hash insert of the tile's ids; `win` bit k: this position created its key's slot ------------
  unsigned tslot[IPT], myrank[IPT];
  unsigned win = 0;
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    tslot[k] = 0xFFFFFFFFu;
    myrank[k] = 0;
    if ((there >> k) & 1u) {
      const long long key = kreg[k];
      unsigned h;
      if (key == EMPTY_KEY) {
        h = LS;
        if (atomicCAS(&lsent, 0u, 1u) == 0u) win |= 1u << k;
      } else {
        h = (unsigned)(mix64((unsigned long long)key) >> 40) & (LS - 1);
        for (;;) {
          const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&sm.lkeys[h]),
                                                   (unsigned long long)EMPTY_KEY, (unsigned long long)key);
          if (old == (unsigned long long)EMPTY_KEY) { win |= 1u << k; break; }
          if (old == (unsigned long long)key) break;
          h = (h + 1) & (LS - 1);
        }
      }
      myrank[k] = atomicAdd(&sm.lcnt[h], 1u);
      if constexpr (BUCKET) { if (has_counts) atomicAdd(&sm.lfq[h], creg[k]); }
      tslot[k] = h;
    }
  }
  KV_STAMP(8);
  // ---- the winners' probes leave (one per distinct key; a position that is no winner asks for entry 0) -----------
  unsigned long long pp[IPT];
  Entry en[IPT];
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    pp[k] = 0ull;
    en[k] = Entry{0, 0u, 0u};
    if constexpr (!NOTABLE) {
      pp[k] = ((win >> k) & 1u) ? home_of(t, kreg[k], mix64((unsigned long long)kreg[k])) : 0ull;
      en[k] = load_entry(&t.entries[pp[k]]);
    }
  }
  if constexpr (NOTABLE) {   // sparse unique numbers (sharded route): a count of 0 = "names no key"; the partition pass sets the real ones
    if (w.zero_counts) {
#pragma unroll
      for (int k = 0; k < IPT; ++k) {
        const long long i = base + (long long)k * TBT + tid;
        if (i < n) w.zero_counts[i] = 0;
      }
    }
  }
  KV_STAMP(9);
  lds_barrier();   // lcnt is final; lkeys is dead: its storage is lrow / escan from here on
  KV_STAMP(1);

  if constexpr (BUCKET) {
    // ---- phase 2 (bucket mode): every winner reserves its place in its partition's bucket (the atomic travels beside
    //      the probe); one block scan numbers the entries that have more than one occurrence ---------------------------
    unsigned wcnt[IPT], wp[IPT], gpos[IPT], pre[IPT];
    unsigned packed = 0;
    const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg(63508) & (unsigned)(NXCD - 1);   // XCC_ID: the XCD this block runs on
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
      wcnt[k] = 0; wp[k] = 0; gpos[k] = 0; pre[k] = 0;
      if ((win >> k) & 1u) {
        wcnt[k] = sm.lcnt[tslot[k]];
        wp[k] = part_of(kreg[k], w.pshift);
        wp[k] = wp[k] * (unsigned)NXCD + xcc;   // the sub-bucket
        gpos[k] = __hip_atomic_fetch_add(&w.bcnt[(size_t)wp[k] * BCNT_STRIDE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (wcnt[k] > 1u) packed += wcnt[k] | (1u << ES_NSH);
      }
    }
    {
      unsigned tot;
      unsigned run = block_excl_scan<TBT / 64, true>(packed, sm.wtot, &tot);
#pragma unroll
      for (int k = 0; k < IPT; ++k) {
        pre[k] = run;
        if (((win >> k) & 1u) && wcnt[k] > 1u) run += wcnt[k] | (1u << ES_NSH);
      }
      if (tid == 0) w.mcount[tile] = (tot & ES_POS) | ((tot >> ES_NSH) << 16);
    }
    KV_STAMP(2);
    // ---- the probes come back: row word + slot-row hint of every distinct key; absent keys are inserted; the entry's
    //      record goes to its bucket in one piece ---------------------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
      if (!((win >> k) & 1u)) continue;
      unsigned hint = 0;
      unsigned r = table_find_from(t, kreg[k], pp[k], en[k], &hint);
      if (__builtin_expect(r == 0u, 0)) { r = tile_insert(t, kreg[k]); hint = 0; }
      else if (__builtin_expect(hint == HINT_NEW, 0)) { r |= NEW_BIT; hint = 0; }
      sm.lrow[tslot[k]] = r;
      sm.lpre[tslot[k]] = pre[k];
      unsigned f = wcnt[k];
      if (has_counts) { f = sm.lfq[tslot[k]]; f = f > 65535u ? 65535u : f; }
      const unsigned src = wcnt[k] > 1u ? (EP_TAG | (tile * (unsigned)(TILE / 2) + (pre[k] >> ES_NSH)))
                                        : (unsigned)base + (unsigned)(k * TBT + tid);
      if (__builtin_expect(gpos[k] < w.bcap, 1)) {
        uint4* rec = w.bkt + 2 * ((size_t)wp[k] * w.bcap + gpos[k]);   // BktRec
        rec[0] = make_uint4((unsigned)kreg[k], (unsigned)((unsigned long long)kreg[k] >> 32), wcnt[k] | (f << 16), r);
        rec[1] = make_uint4(hint, src, 0u, 0u);
      } else {
        raise_error(t, 2u);   // a bucket is full (see bucket_capacity in kvhip.hip): the batch is void, the next call reports it
      }
    }
    lds_barrier();
    KV_STAMP(3);
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
      if (tslot[k] == 0xFFFFFFFFu || sm.lcnt[tslot[k]] <= 1u) continue;
      const unsigned pr = sm.lpre[tslot[k]];
      sm.mr[(pr & ES_POS) + myrank[k]] = (unsigned)(k * TBT + tid) | ((pr >> ES_NSH) << 11) | (myrank[k] == 0u ? 0x80000000u : 0u);
    }
    lds_barrier();
    {
      unsigned* dst = w.mrow + (size_t)tile * TILE;
      for (int j = tid; j < TILE; j += TBT) dst[j] = sm.mr[j];
    }
    KV_STAMP(4);
  } else {
  // ---- phase 2: the distinct keys counting-sorted by owning partition ----------------------------------------------
  unsigned wcnt[IPT], wp[IPT], wr[IPT];
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    wcnt[k] = 0; wp[k] = 0; wr[k] = 0;
    if ((win >> k) & 1u) {
      wcnt[k] = sm.lcnt[tslot[k]];
      wp[k] = part_of(kreg[k], w.pshift);
      wr[k] = atomicAdd(&sm.hist[wp[k]], 1u | (wcnt[k] << 16)) & 0xFFFFu;
    }
  }
  lds_barrier();
  {
    const unsigned per = (P + TBT - 1) / TBT;
    const unsigned p0 = tid * per, p1 = min(p0 + per, P);
    unsigned sum = 0;
    for (unsigned p = p0; p < p1; ++p) sum += sm.hist[p];
    unsigned tot;
    unsigned run = block_excl_scan<TBT / 64, true>(sum, sm.wtot, &tot);
    for (unsigned p = p0; p < p1; ++p) { const unsigned c = sm.hist[p]; sm.hist[p] = run; run += c; }
    if (tid == 0) sm.hist[P] = tot;
  }
  lds_barrier();
  const unsigned nent = sm.hist[P] & 0xFFFFu;   // entries of the tile
  // tile-major: toff[tile][0..P] — one contiguous run per tile.  (Partition-major rows, which the partition blocks read as two
  // contiguous runs, cost the tile pass P + 1 scattered 4-byte stores per tile — a million at low skew; the column the
  // partition block reads instead stays in its XCD's L2, the blocks of an XCD being neighbours: k_ltile -1.5 us at
  // configs[1], the lookup -7 .. -9 us at Zipf 0.3 / 0.8.)
  for (unsigned p = tid; p <= P; p += TBT) w.toff[(size_t)tile * (P + 1) + p] = sm.hist[p];
  unsigned wpos[IPT];
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    wpos[k] = 0xFFFFFFFFu;
    if ((win >> k) & 1u) {
      wpos[k] = (sm.hist[wp[k]] & 0xFFFFu) + wr[k];
      w.ent_key[(size_t)tile * TILE + wpos[k]] = kreg[k];
    }
  }
  if (det) {
    // deterministic mode: the order of a partition's entries inside the tile was the arrival order of LDS atomics;
    // it becomes the order of their keys (k_tsum's additions follow the entries' places in the tile's row list)
    __syncthreads();   // the tile's ent_key is written
    unsigned npos[IPT];
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
      npos[k] = 0xFFFFFFFFu;
      if (wpos[k] == 0xFFFFFFFFu) continue;
      const unsigned p0 = sm.hist[wp[k]] & 0xFFFFu, p1 = sm.hist[wp[k] + 1u] & 0xFFFFu;
      unsigned less = 0;
      for (unsigned j = p0; j < p1; ++j) less += w.ent_key[(size_t)tile * TILE + j] < kreg[k] ? 1u : 0u;
      npos[k] = p0 + less;
    }
    __syncthreads();   // ... and read by everyone
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
      if (npos[k] == 0xFFFFFFFFu) continue;
      wpos[k] = npos[k];
      w.ent_key[(size_t)tile * TILE + wpos[k]] = kreg[k];
    }
  }
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    if (wpos[k] == 0xFFFFFFFFu) continue;
    sm.lpos[tslot[k]] = (unsigned short)wpos[k];
    if (!has_counts) w.ent_a[(size_t)tile * TILE + wpos[k]] = wcnt[k] | (wcnt[k] << 16);   // <= TILE: the frequency count equals the occurrences
    // per entry for the scan below: its rows | 1, if it has more than one
    sm.escan[wpos[k]] = wcnt[k] > 1u ? (wcnt[k] | (1u << ES_NSH)) : 0u;
  }
  lds_barrier();
  KV_STAMP(2);

  // ---- phase 3: the multi-occurrence entries: where their rows start in the tile's mrow image, their numbers --------
  {
    constexpr unsigned PER = (TILE + 1 + TBT - 1) / TBT;
    const unsigned e0 = tid * PER, e1 = min(e0 + PER, nent);
    unsigned sum = 0;
    for (unsigned e = e0; e < e1; ++e) sum += sm.escan[e];
    unsigned tot;
    unsigned run = block_excl_scan<TBT / 64, true>(sum, sm.wtot, &tot);
    for (unsigned e = e0; e < e1; ++e) { const unsigned c = sm.escan[e]; sm.escan[e] = run; run += c; }
    if (tid == 0) w.mcount[tile] = (tot & ES_POS) | ((tot >> ES_NSH) << 16);
  }
  // ---- the probes come back: row word + slot-row hint of every distinct key; absent keys are inserted ---------------
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    if (wpos[k] == 0xFFFFFFFFu) continue;
    unsigned hint = 0;
    unsigned r = 0;
    if constexpr (!NOTABLE) {
      r = table_find_from(t, kreg[k], pp[k], en[k], &hint);
      if (__builtin_expect(r == 0u, 0)) { r = tile_insert(t, kreg[k]); hint = 0; }
      else if (__builtin_expect(hint == HINT_NEW, 0)) { r |= NEW_BIT; hint = 0; }
    }
    const size_t e = (size_t)tile * TILE + wpos[k];
    sm.lrow[tslot[k]] = r;
    w.ent_b[e] = r;
    w.ent_base[e] = hint;
  }
  lds_barrier();
  KV_STAMP(3);
  if (w.pos_ent) {   // every position's entry in its tile (the sharded finish reads position -> entry -> record)
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
      const long long i = base + (long long)k * TBT + tid;
      if (i < n) w.pos_ent[i] = tslot[k] != 0xFFFFFFFFu ? sm.lpos[tslot[k]] : (unsigned short)0xFFFFu;
    }
  }
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    if (wpos[k] == 0xFFFFFFFFu) continue;
    const unsigned pre = sm.escan[wpos[k]];
    // the entry's source: its sum — row (pre >> ES_NSH) of the tile's epart rows — or, alone, its one position
    w.ent_rec[(size_t)tile * TILE + wpos[k]] = wcnt[k] > 1u ? (EP_TAG | (tile * (unsigned)(TILE / 2) + (pre >> ES_NSH)))
                                                            : (unsigned)base + (unsigned)(k * TBT + tid);
  }
  if (det) {
    // rank = occurrences of the key at smaller input positions (see tile_body of kv_kernels.h)
    unsigned short* run = sm.lrun;
    for (int e = tid; e <= TILE; e += TBT) run[e] = 0;
    lds_barrier();
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
      const bool valid = tslot[k] != 0xFFFFFFFFu;
      const unsigned e = valid ? (unsigned)sm.lpos[tslot[k]] : 0u;
      unsigned long long mask = __ballot(valid);
#pragma unroll
      for (int b = 0; b < 11; ++b) {
        const bool bit = (e >> b) & 1u;
        const unsigned long long bal = __ballot(bit);
        mask &= bit ? bal : ~bal;
      }
      const unsigned rw = (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
      const unsigned cnt = (unsigned)__popcll(mask);
      unsigned before = 0;
      for (int wv = 0; wv < TBT / 64; ++wv) {
        if (wave == wv && valid) {
          before = run[e];
          if (rw == 0u) run[e] = (unsigned short)(before + cnt);
        }
        lds_barrier();
      }
      myrank[k] = valid ? before + rw : 0u;
    }
  }
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    if (tslot[k] == 0xFFFFFFFFu || sm.lcnt[tslot[k]] <= 1u) continue;
    const unsigned pre = sm.escan[sm.lpos[tslot[k]]];
    sm.mr[(pre & ES_POS) + myrank[k]] = (unsigned)(k * TBT + tid) | ((pre >> ES_NSH) << 11) | (myrank[k] == 0u ? 0x80000000u : 0u);
  }
  // ---- per-occurrence counts: frequency sum per entry (hist is dead: reused) ----------------------------------
  if (has_counts) {
    lds_barrier();
    for (unsigned e = tid; e <= (unsigned)TILE; e += TBT) sm.hist[e] = 0;
    lds_barrier();
#pragma unroll
    for (int k = 0; k < IPT; ++k)
      if (tslot[k] != 0xFFFFFFFFu) atomicAdd(&sm.hist[sm.lpos[tslot[k]]], creg[k]);
    lds_barrier();
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
      if (wpos[k] == 0xFFFFFFFFu) continue;
      const unsigned f = sm.hist[wpos[k]];
      w.ent_a[(size_t)tile * TILE + wpos[k]] = wcnt[k] | ((f > 65535u ? 65535u : f) << 16);
    }
  }
  lds_barrier();
  {
    unsigned* dst = w.mrow + (size_t)tile * TILE;
    for (int j = tid; j < TILE; j += TBT) dst[j] = sm.mr[j];
  }
  KV_STAMP(4);

  }
  // ---- phase 4: the output rows.  Wave wv, round k holds positions k * TBT + wv * 64 + lane in its own registers ---
  if constexpr (GATHER) {
    constexpr int RW = 64 / VQ;            // rows per copy instruction
    constexpr int CW = VQ < 8 ? VQ : 8;    // copy instructions in flight
    const int lane = tid & 63;
    const int v = lane % VQ, sub = lane / VQ;
    // a row is D4 = dim / 4 float4; VQ is the next power of two (dims 12, 20, 100 ...: the lanes v >= D4 of a row's
    // group load float4 0 of the row instead — no branch around the load — and do not store)
    const int D4 = t.dim >> 2;
    const bool vlive = v < D4;
    const int vv = vlive ? v : 0;
    // SINGLE: the slab is one chunk — rows are addressed without the chunk-table branch, which would put a wait in
    // front of every load; with two blocks per CU the CW loads of a step must really be in flight together
    auto copy_rows = [&](auto single_tag) {
      constexpr bool SINGLE = decltype(single_tag)::value;
      const float4* rows0 = reinterpret_cast<const float4*>(t.c0.rows);
      constexpr int SPK = VQ / CW;          // pieces (CW copy instructions) per round k
      constexpr int NP = IPT * SPK;
      unsigned rr[IPT];
#pragma unroll
      for (int k = 0; k < IPT; ++k) rr[k] = tslot[k] != 0xFFFFFFFFu ? sm.lrow[tslot[k]] : 0u;   // skipped records read the zero row
      auto issue = [&](int pc, float4 (&val)[CW]) {
        const int k = pc / SPK, j0 = (pc % SPK) * CW;
#pragma unroll
        for (int j = 0; j < CW; ++j) {
          const unsigned rj = __shfl(rr[k], (j0 + j) * RW + sub) & ROW_MASK;
          if constexpr (SINGLE) val[j] = rows0[(size_t)rj * D4 + vv];
          else val[j] = reinterpret_cast<const float4*>(row_ptr(t, rj))[vv];
        }
      };
      auto flush = [&](int pc, float4 (&val)[CW]) {
        const int k = pc / SPK, j0 = (pc % SPK) * CW;
        const long long r0 = base + (long long)k * TBT + (tid & ~63);
        if (__builtin_expect(__ballot((rr[k] & NEW_BIT) != 0u) != 0ull, 0)) {
          // a key inserted by this batch: its row is the init rule's value (kv_variable.h:889-898), written to the
          // table by k_part2; here it is computed, not read
#pragma unroll
          for (int j = 0; j < CW; ++j) {
            const unsigned rj = __shfl(rr[k], (j0 + j) * RW + sub);
            const long long kj = __shfl(kreg[k], (j0 + j) * RW + sub);
            if (rj & NEW_BIT) {
              const unsigned long long h = pick64((unsigned long long)kj ^ (t.seed * 0x9E3779B97F4A7C15ULL));
              const float4 a = reinterpret_cast<const float4*>(t.init_table + (size_t)((unsigned)h % t.init_rows) * t.dim)[vv];
              const float4 b = reinterpret_cast<const float4*>(t.init_table + (size_t)((unsigned)(h >> 32) % t.init_rows) * t.dim)[vv];
              val[j] = make_float4((a.x + b.x) * 0.5f, (a.y + b.y) * 0.5f, (a.z + b.z) * 0.5f, (a.w + b.w) * 0.5f);
            }
          }
        }
#pragma unroll
        for (int j = 0; j < CW; ++j) {
          const long long ii = r0 + (j0 + j) * RW + sub;
          if (ii < n && vlive) {
            float4* dst = reinterpret_cast<float4*>(out + (size_t)ii * t.dim) + v;
            if constexpr (PAIRS) {
              // the sharded owner lookup: its rows are read again at once — by the finish of this rank's own segment, which
              // stays in this buffer, and by the exchange — so they are kept in the caches (k_shard_finish 40.9 -> 25 us)
              *dst = val[j];
            } else {
              __builtin_nontemporal_store(val[j].x, &dst->x); __builtin_nontemporal_store(val[j].y, &dst->y);
              __builtin_nontemporal_store(val[j].z, &dst->z); __builtin_nontemporal_store(val[j].w, &dst->w);
            }
          }
        }
      };
      // two pieces in flight (NP is even: IPT = 4) where the registers allow two blocks per CU with them; the compiler
      // barriers pin that schedule in the unrolled loop
      float4 va[CW];
      if constexpr (VQ <= 16) {
        float4 vb[CW];
        issue(0, va);
#pragma unroll
        for (int pc = 0; pc < NP; pc += 2) {
          issue(pc + 1, vb);
          asm volatile("" ::: "memory");
          flush(pc, va);
          if (pc + 2 < NP) issue(pc + 2, va);
          asm volatile("" ::: "memory");
          flush(pc + 1, vb);
        }
      } else {
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) {
          issue(pc, va);
          flush(pc, va);
          asm volatile("" ::: "memory");
        }
      }
    };
    if (single_chunk(t)) copy_rows(std::true_type{}); else copy_rows(std::false_type{});
  }
  KV_STAMP(5);
}

template <typename IdT, int VQ, bool GATHER, bool BUCKET = false, bool NOTABLE = false>
__global__ void __launch_bounds__(TBT) k_ltile(TableDev t, WsDev w, const IdT* __restrict__ ids,
                                              const int* __restrict__ counts, long long n, int det, float* __restrict__ out) {
  ltile_body<IdT, VQ, GATHER, BUCKET, NOTABLE>(t, w, ids, counts, n, det, out);
}

// ------------------------------------------------------------------------------------------
// k_shard_finish: the sharded lookup's output rows from the records that came back
// ------------------------------------------------------------------------------------------
// out[i] = rows[slot_of[uniq_of_entry[tile * TILE + pos_ent[i]]]]: position -> its entry in the tile (k_ltile<NOTABLE>
// filed it) -> the entry's distinct-id number (k_papply PA_UNIQUE wrote it to ent_b) -> the record the id was sent in
// -> the row the owner returned.  A wave takes 64 positions, lane l resolves position l, the rows go VQ lanes per row
// with streaming stores (the copy of goz_wave).
template <int VQ, int CW = 4>
__device__ __forceinline__ void shard_finish_body(const unsigned short* __restrict__ pos_ent, const unsigned* __restrict__ ent_u,
                                                  const int* __restrict__ slot_of, const float* __restrict__ rows,
                                                  float* __restrict__ out, long long n, int dim,
                                                  const float* __restrict__ rows_self, unsigned self_lo, unsigned self_len) {
  constexpr int RW = 64 / VQ;
  const int lane = threadIdx.x & 63;
  const int v = lane % VQ, sub = lane / VQ;
  const int D4 = dim >> 2;   // float4 per row (<= VQ: lanes past it are masked)
  const bool vlive = v < D4;
  const int vv = vlive ? v : 0;
  const long long nwaves = (long long)gridDim.x * (TB / 64);
  for (long long r0 = ((long long)blockIdx.x * (TB / 64) + (threadIdx.x >> 6)) * 64; r0 < n; r0 += nwaves * 64) {
    const long long i = r0 + lane;
    unsigned rec = 0;   // record 0: a header's row (zeros) — positions past the end, ids that found no room in their segment
    if (i < n) {
      const unsigned e = pos_ent[i];
      if (e != 0xFFFFu) rec = (unsigned)slot_of[ent_u[(size_t)(i / TILE) * TILE + e]];
    }
#pragma unroll
    for (int j0 = 0; j0 < VQ; j0 += CW) {
      float4 val[CW];
      unsigned rj[CW];
#pragma unroll
      for (int j = 0; j < CW && j0 + j < VQ; ++j) rj[j] = __shfl(rec, (j0 + j) * RW + sub);
#pragma unroll
      for (int j = 0; j < CW && j0 + j < VQ; ++j)   // (records [self_lo, self_lo + self_len): this rank's own segment, read where the serve wrote it)
        val[j] = reinterpret_cast<const float4*>((((unsigned)rj[j] - self_lo < self_len) ? rows_self : rows) + (size_t)rj[j] * dim)[vv];
#pragma unroll
      for (int j = 0; j < CW && j0 + j < VQ; ++j) {
        const long long ii = r0 + (j0 + j) * RW + sub;
        if (ii < n && vlive) {
          float4* dst = reinterpret_cast<float4*>(out + (size_t)ii * dim) + v;
          __builtin_nontemporal_store(val[j].x, &dst->x); __builtin_nontemporal_store(val[j].y, &dst->y);
          __builtin_nontemporal_store(val[j].z, &dst->z); __builtin_nontemporal_store(val[j].w, &dst->w);
        }
      }
    }
  }
}
template <int VQ, int CW = 4>
__global__ void __launch_bounds__(TB) k_shard_finish(const unsigned short* __restrict__ pos_ent, const unsigned* __restrict__ ent_u,
                                                     const int* __restrict__ slot_of, const float* __restrict__ rows,
                                                     float* __restrict__ out, long long n, int dim,
                                                     const float* __restrict__ rows_self, unsigned self_lo, unsigned self_len) {
  shard_finish_body<VQ, CW>(pos_ent, ent_u, slot_of, rows, out, n, dim, rows_self, self_lo, self_len);
}
// several tables of one row geometry in one launch (blockIdx.y = table)
struct FinishDesc {
  const unsigned short* pos_ent;
  const unsigned* ent_u;
  const int* slot_of;
  const float* rows;
  float* out;
  long long n;
  const float* rows_self;
  unsigned self_lo, self_len;
  int dim, pad;
};
template <int VQ, int CW = 4>
__global__ void __launch_bounds__(TB) k_shard_finish_multi(const FinishDesc* __restrict__ descs) {
  const FinishDesc d = descs[blockIdx.y];
  shard_finish_body<VQ, CW>(d.pos_ent, d.ent_u, d.slot_of, d.rows, d.out, d.n, d.dim, d.rows_self, d.self_lo, d.self_len);
}

// ------------------------------------------------------------------------------------------
// k_part2: the partition pass over entries that already carry their rows
// ------------------------------------------------------------------------------------------
// seg_directory of kv_kernels.h over the tile-major toff (toff[tile][0..P]): thread k takes tiles k, k + T, ... — column p and
// p + 1 of each — the exclusive prefix over tiles is taken round by round.  *pbase = ENTRIES of the partitions before p.
template <int T, int NW>
__device__ __forceinline__ unsigned seg_directory_t(const WsDev& w, unsigned p, unsigned short* tpre,
                                                    unsigned short* tstart, unsigned* wtot, unsigned* pbase) {
  const unsigned NT = w.ntiles;
  unsigned run0 = 0, pb = 0;
  for (unsigned tb = 0; tb < NT; tb += T) {   // block-uniform
    const unsigned t = tb + threadIdx.x;
    unsigned len = 0, s0 = 0;
    if (t < NT) {
      const unsigned* c0 = w.toff + (size_t)t * (w.P + 1) + p;
      const unsigned a = c0[0], b = c0[1];
      s0 = a & 0xFFFFu;
      len = (b & 0xFFFFu) - s0;
      tstart[t] = (unsigned short)s0;
    }
    // one packed scan: entries of the partition (low 16, < 65536 or the caller gives up) | entries before it (high)
    unsigned tot;
    const unsigned ex = block_excl_scan<NW>(len, wtot, &tot);
    unsigned tot2;
    block_excl_scan<NW>(s0, wtot, &tot2);
    if (t < NT) tpre[t] = (unsigned short)min(run0 + ex, 65535u);
    run0 += tot;
    pb += tot2;
  }
  __syncthreads();
  *pbase = pb;
  return run0;
}
// MODE_LOOKUP: FindOrInsert bookkeeping (frequency += occurrences, day, under-threshold flag, delta marks);
// MODE_APPLYIDX: FindOrInsertUnsafe (a key the optimizer meets first: frequency word 1, not filtered).
// REC = false: the bookkeeping alone — no key records, entry list or work items.  For a lookup that no optimizer apply
// takes over through k_apply2 (a lookup without a token; a token lookup whose pass is settled by another op): what the
// records would serve is k_apply2, and an apply that still comes with the token runs k_papply over the tiles' entries
// (PA_NONE).  Saves the three scans, the record and item stores and the whole second pass over the entries.
template <int MODE, bool REC = true>
__device__ __forceinline__ void part2_body(const WsDev& w, const PartArgs& a) {
  constexpr int HSK = 1024;
  constexpr int UCAPK = HSK - TBK;
  constexpr int EB = 8;
  __shared__ long long hkey[HSK + 1];
  __shared__ unsigned hval[HSK + 1];   // summed frequency count; after the owner work: the key's record word
  __shared__ unsigned hrow[HSK + 1];   // max over the key's entries of the row word (an entry that knows the row wins)
  __shared__ unsigned hhint[HSK + 1];  // slot-row hint
  __shared__ unsigned hocc[HSK + 1];   // entries of the key, then its start in the entry list
  __shared__ unsigned hrun[HSK + 1];   // entries placed so far (pass 2)
  __shared__ unsigned short lnew[UCAPK + 8];
  __shared__ unsigned short ulist[UCAPK + 8];
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
  const unsigned E = seg_directory_t<TBK, TBK / 64>(w, p, tpre, tstart, wtot, &pbase);
  __shared__ unsigned lcold, lhot, lchunk, litm, lnbig;   // litm: cold batches so far (they fill the stretch from its end)
  __shared__ unsigned lbig[16][3];
  if (E == 0) { if (REC && tid == 0) w.pmeta[p] = make_uint4(0u, 0u, pbase, 0u); return; }
  __shared__ unsigned stkR[24], stkr[24];
  __shared__ int sp;
  if (tid == 0) { stkR[0] = 1; stkr[0] = 0; sp = 1; lpcur = pbase; lcold = 0; lhot = 0; lchunk = 0; litm = 0; }
  __syncthreads();
  if (E > 65535u) {
    if (tid == 0) { raise_error(a.tv, 2u); if (REC) w.pmeta[p] = make_uint4(0u, 0u, pbase, 0u); }
    return;
  }
  const unsigned hcr = w.hc;
  while (sp > 0) {
    const unsigned R = stkR[sp - 1], round = stkr[sp - 1];
    __syncthreads();
    if (tid == 0) --sp;
    for (int s = tid; s <= HSK; s += TBK) { hkey[s] = EMPTY_KEY; hval[s] = 0; hrow[s] = 0; hhint[s] = 0; hocc[s] = 0; hrun[s] = 0; }
    if (tid == 0) { lnu = 0; lsent = 0; lnnew = 0; lnbig = 0; }
    __syncthreads();
    // ---- pass 1: distinct keys, their counts, rows and hints --------------------------------------------------
    unsigned cge[EB];
    unsigned short cslot[EB];
    const bool cached = (R == 1 && E <= (unsigned)(EB * TBK));
    for (unsigned x0 = 0; x0 < E; x0 += EB * TBK) {
      unsigned ge[EB];
      long long key[EB];
      unsigned ea[EB], rw[EB], hi[EB];
#pragma unroll
      for (int k = 0; k < EB; ++k) {
        const unsigned x = x0 + k * TBK + tid;
        ge[k] = x < E ? (unsigned)seg_entry(tpre, tstart, NT, x) : 0xFFFFFFFFu;
      }
#pragma unroll
      for (int k = 0; k < EB; ++k) {
        key[k] = 0; ea[k] = 0; rw[k] = 0; hi[k] = 0;
        if (ge[k] != 0xFFFFFFFFu) { key[k] = w.ent_key[ge[k]]; ea[k] = w.ent_a[ge[k]]; rw[k] = w.ent_b[ge[k]]; hi[k] = w.ent_base[ge[k]]; }
      }
#pragma unroll
      for (int k = 0; k < EB; ++k) {
        if (x0 == 0) { cge[k] = ge[k]; cslot[k] = 0; }
        if (ge[k] == 0xFFFFFFFFu || !in_round(key[k], R, round)) continue;
        if (lnu >= (unsigned)UCAPK) continue;
        bool first;
        const unsigned h = lds_key_slot<HSK>(hkey, &lsent, key[k], true, &first);
        if (first) {
          const unsigned u = atomicAdd(&lnu, 1u);
          if (u < (unsigned)UCAPK) ulist[u] = (unsigned short)h;
        }
        if (MODE == MODE_LOOKUP) atomicAdd(&hval[h], ea[k] >> 16);
        atomicAdd(&hocc[h], 1u);
        atomicMax(&hrow[h], rw[k]);
        if (hi[k]) atomicMax(&hhint[h], hi[k]);
        if (x0 == 0) cslot[k] = (unsigned short)h;
      }
    }
    __syncthreads();
    if (lnu >= (unsigned)UCAPK) {
      __syncthreads();
      if (tid == 0) {
        if (sp + 2 <= 24) {
          stkR[sp] = 2 * R; stkr[sp] = round; ++sp;
          stkR[sp] = 2 * R; stkr[sp] = round + R; ++sp;
        } else {
          raise_error(a.tv, 2u);
        }
      }
      __syncthreads();
      continue;
    }
    KV_STAMPP(1);
    const unsigned nu = lnu;
    KV_STAMPPV(6, E); KV_STAMPPV(7, nu); KV_STAMPPV(8, ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492));

    // ---- the keys' places in the entry list; their numbers: cold keys by class (1, 2, 3..LCOLD entries — a batch of
    //      the apply then holds keys of ONE class, and the first two sources ride in the record), hot keys and their
    //      chunks; the round's work items: its chunk items, then its cold batches
    constexpr int PERU = (UCAPK + TBK - 1) / TBK;
    unsigned kst[PERU], kcnt[PERU], krank[PERU], kchunk[PERU];
    const unsigned gb = 64u / (unsigned)apply_lanes(a.tv.dim);   // keys per cold batch
    if constexpr (!REC) {
      if (tid == 0) lcold += nu;   // (only the batch's distinct keys are counted: ctr[5])
    } else {
      unsigned sum = 0, ch = 0, hh = 0;
#pragma unroll
      for (int q = 0; q < PERU; ++q) {
        const unsigned u = tid * PERU + q;
        kcnt[q] = u < nu ? hocc[ulist[u]] : 0u;
        sum += kcnt[q];
        if (u < nu) {
          if (kcnt[q] == 1u) ch += 1u;
          else if (kcnt[q] == 2u) ch += 1u << 10;
          else if (kcnt[q] <= (unsigned)LCOLD) ch += 1u << 20;
          else hh += 1u | (((kcnt[q] + hcr - 1u) / hcr) << 10);
        }
      }
      const unsigned cur = lpcur;
      unsigned tot, chtot, htot;
      unsigned run = cur + block_excl_scan<TBK / 64>(sum, wtot, &tot);
      unsigned chrun = block_excl_scan<TBK / 64>(ch, wtot, &chtot);
      unsigned hrn = block_excl_scan<TBK / 64>(hh, wtot, &htot);
      const unsigned t1 = chtot & 1023u, t2 = (chtot >> 10) & 1023u, t3 = chtot >> 20, th = htot & 1023u, tk = htot >> 10;
      const unsigned c0 = lcold, h0 = lhot, k0 = lchunk, i0 = litm;

#pragma unroll
      for (int q = 0; q < PERU; ++q) {
        const unsigned u = tid * PERU + q;
        kst[q] = run; krank[q] = 0; kchunk[q] = 0;
        if (u < nu) {
          hocc[ulist[u]] = run; run += kcnt[q];
          if (kcnt[q] == 1u) { krank[q] = pbase + c0 + (chrun & 1023u); chrun += 1u; }
          else if (kcnt[q] == 2u) { krank[q] = pbase + c0 + t1 + ((chrun >> 10) & 1023u); chrun += 1u << 10; }
          else if (kcnt[q] <= (unsigned)LCOLD) { krank[q] = pbase + c0 + t1 + t2 + (chrun >> 20); chrun += 1u << 20; }
          else {
            krank[q] = pbase + h0 + (hrn & 1023u);
            kchunk[q] = k0 + (hrn >> 10);
            hrn += 1u | (((kcnt[q] + hcr - 1u) / hcr) << 10);
          }
        }
      }
      // the round's cold batches, class by class: {first cold list index, keys, class}.  Chunk items fill the
      // partition's stretch of litem from its front (by chunk number), cold batches from its back: the directory
      // (items2_body) deals every hot chunk of the batch before any cold batch
      {
        const unsigned nb1 = (t1 + gb - 1u) / gb, nb2 = (t2 + gb - 1u) / gb, nb3 = (t3 + gb - 1u) / gb;
        const unsigned ib = pbase + E - 1u - i0;
        for (unsigned bq = tid; bq < nb1 + nb2 + nb3; bq += TBK) {
          unsigned cls, bb, first, cntc;
          if (bq < nb1) { cls = 1u; bb = bq; first = pbase + c0; cntc = t1; }
          else if (bq < nb1 + nb2) { cls = 2u; bb = bq - nb1; first = pbase + c0 + t1; cntc = t2; }
          else { cls = 3u; bb = bq - nb1 - nb2; first = pbase + c0 + t1 + t2; cntc = t3; }
          w.litem[ib - bq] = make_uint4(first + bb * gb, min(gb, cntc - bb * gb), cls, 0u);
        }
        __syncthreads();
        if (tid == 0) {
          lpcur = cur + tot; lcold = c0 + t1 + t2 + t3; lhot = h0 + th; lchunk = k0 + tk;
          litm = i0 + nb1 + nb2 + nb3;
        }
      }
    }
    auto put_rec = [&](int q, long long key, unsigned roww, unsigned hint) {
      const uint4 ra = make_uint4((unsigned)key, (unsigned)((unsigned long long)key >> 32), roww, hint);
      if (kcnt[q] <= (unsigned)LCOLD) {
        w.coldlist[2 * (size_t)krank[q]] = ra;
        // {start in the entry list, entries, first source, second source} — the sources are filed by pass 2
        reinterpret_cast<uint2*>(&w.coldlist[2 * (size_t)krank[q] + 1])[0] = make_uint2(kst[q], kcnt[q]);
      } else {
        w.hotlist[2 * (size_t)krank[q]] = ra;
        w.hotlist[2 * (size_t)krank[q] + 1] = make_uint4(kst[q], kcnt[q], kchunk[q], hcr);
        const unsigned nch = (kcnt[q] + hcr - 1u) / hcr;
        const unsigned ib = pbase + kchunk[q];
        if (nch <= 16u) {
          for (unsigned i = 0; i < nch; ++i) w.litem[ib + i] = make_uint4(krank[q] | HEAD_BIT, i, kchunk[q] + i, 0u);
        } else {
          const unsigned b = atomicAdd(&lnbig, 1u);
          if (b < 16u) { lbig[b][0] = krank[q]; lbig[b][1] = kchunk[q]; lbig[b][2] = nch; }
          else for (unsigned i = 0; i < nch; ++i) w.litem[ib + i] = make_uint4(krank[q] | HEAD_BIT, i, kchunk[q] + i, 0u);
        }
      }
    };

    // ---- owner work: one thread per distinct key; ONE hop (the row's record), none in the optimizer's index pass ----
    {
      unsigned sl[PERU], r[PERU];
      bool isnew[PERU];
      uint2 m[PERU];
#pragma unroll
      for (int k = 0; k < PERU; ++k) {
        const unsigned u = tid * PERU + k;
        sl[k] = 0xFFFFFFFFu; r[k] = 0; isnew[k] = false; m[k] = make_uint2(0u, (unsigned)FLAG_DIRTY);
        if (u < nu) {
          sl[k] = ulist[u];
          const unsigned rw = hrow[sl[k]];
          r[k] = rw & ROW_MASK; isnew[k] = (rw & NEW_BIT) != 0u;
          if (MODE == MODE_LOOKUP && r[k] != 0u && !isnew[k]) m[k] = load_freq_flags(a.tv, r[k]);
        }
      }
#pragma unroll
      for (int k = 0; k < PERU; ++k) {
        if (sl[k] == 0xFFFFFFFFu) continue;
        const unsigned s = sl[k];
        const long long key = (s == HSK) ? EMPTY_KEY : hkey[s];
        if (__builtin_expect(isnew[k], 0)) {
          // the tile that won the key published {row, HINT_NEW}; the hint goes back to "none"
          Entry* e = table_entry_of(a.tv, key);
          if (e) {
            if (r[k] == 0u) { const unsigned er = load_entry(e).row; r[k] = er != ROW_TOMB ? er : 0u; }
            e->hint = 0u;
          }
          hrow[s] = r[k] | NEW_BIT;
        }
        if constexpr (REC) put_rec(k, key, r[k] | ((MODE == MODE_APPLYIDX && isnew[k]) ? NEW_BIT : 0u), isnew[k] ? 0u : hhint[s]);
        if (r[k] == 0u) continue;   // row slab overflow: the error flag is up
        RowMeta* mp = meta_ptr(a.tv, r[k]);
        if (isnew[k]) { mp->key = key; mp->delta_train = 0; mp->delta_pred = 0; }
        if (MODE == MODE_LOOKUP) {
          mark_delta(a.tv, r[k]);
          // find_func / insert_func (kv_variable.h:320-363)
          const unsigned cnt = a.count_once ? 1u : hval[s];
          unsigned lo = (m[k].x & 0xFFFFu) + (cnt > 65535u ? 65535u : cnt);
          if (lo > 65535u) lo = 65535u;
          mp->freq = (a.day << 16) | lo;
          if (isnew[k]) mp->flags = (unsigned char)FLAG_DIRTY;
          if (m[k].y & FLAG_DIRTY) lnew[atomicAdd(&lnnew, 1u)] = (unsigned short)(s | (isnew[k] ? 0x8000u : 0u));
        } else if (isnew[k]) {
          // FindOrInsertUnsafe (kv_variable.h:382-416): init rule, frequency word 1; existing rows are not touched
          mp->freq = 1u; mp->flags = 0;
          lnew[atomicAdd(&lnnew, 1u)] = (unsigned short)(s | 0x8000u);
        }
      }
    }
    __syncthreads();
    if constexpr (REC) {
      const unsigned nbig = min(lnbig, 16u);
      for (unsigned b = 0; b < nbig; ++b)
        for (unsigned i = tid; i < lbig[b][2]; i += TBK)
          w.litem[pbase + lbig[b][1] + i] = make_uint4(lbig[b][0] | HEAD_BIT, i, lbig[b][1] + i, 0u);
#pragma unroll
      for (int q = 0; q < PERU; ++q) {
        const unsigned u = tid * PERU + q;
        if (u < nu) hval[ulist[u]] = krank[q] | (kcnt[q] > (unsigned)LCOLD ? 0x80000000u : 0u);
      }
      __syncthreads();
    }
    KV_STAMPP(2);

    // ---- rows that need lanes: contents of new rows, under-threshold flag of rows that changed ------------------
    {
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
        bool big = false;
        if (live && r != 0) {
          if (isnew) big = init_row_coop(a.tv, key, row, lane8, 8);
          else if (MODE == MODE_LOOKUP)
            for (int e = lane8; e < D; e += 8) big |= fabsf(row[e]) >= CUTOFF;
        }
        const unsigned long long mb = __ballot(big);
        const bool any = ((mb >> ((tid & 63) & ~7)) & 0xFFull) != 0;
        if (live && r != 0 && lane8 == 0) {
          unsigned char* fp = flags_ptr(a.tv, r);
          if (MODE == MODE_LOOKUP) {
            const unsigned black = isnew ? 0u : (*fp & FLAG_BLACK);
            *fp = (unsigned char)(black ? (FLAG_BLACK | FLAG_UNDER) : (any ? 0u : FLAG_UNDER));
          } else {
            *fp = (unsigned char)(any ? 0u : FLAG_UNDER);
          }
        }
      }
    }
    KV_STAMPP(3);

    // ---- pass 2: the entry list — entry x of key h goes to order[start(h) + its number among h's entries] -------
    if constexpr (REC) {
      auto place = [&](unsigned ge, unsigned h) {
        const unsigned idx = atomicAdd(&hrun[h], 1u);
        const unsigned src = w.ent_rec[ge];
        w.order[hocc[h] + idx] = src;
        if (idx < 2u) {   // a cold key's first two sources ride in its record
          const unsigned rec = hval[h];
          if (!(rec >> 31)) reinterpret_cast<unsigned*>(&w.coldlist[2 * (size_t)rec + 1])[2 + idx] = src;
        }
      };
      if (a.det) {
        // deterministic mode: a key's entries in tile order = ascending x (TBK entries per round, wave by wave)
        const int lane = tid & 63, wave = tid >> 6;
        for (unsigned x0 = 0; x0 < E; x0 += TBK) {
          const unsigned x = x0 + tid;
          bool valid = x < E;
          size_t ge = 0;
          unsigned h = 0xFFFFFFFFu;
          if (valid) {
            ge = seg_entry(tpre, tstart, NT, x);
            const long long key = w.ent_key[ge];
            valid = in_round(key, R, round);
            if (valid) { bool first; h = lds_key_slot<HSK>(hkey, &lsent, key, false, &first); }
          }
          unsigned within = 0;
          for (int j = 0; j < 63; ++j) {
            const unsigned hj = __shfl(h, j);
            if (j < lane && hj == h) within += 1u;
          }
          unsigned idx = 0;
          for (int wv = 0; wv < TBK / 64; ++wv) {
            if (wave == wv && valid) { idx = hrun[h] + within; atomicAdd(&hrun[h], 1u); }
            __syncthreads();
          }
          if (valid) {
            const unsigned src = w.ent_rec[ge];
            w.order[hocc[h] + idx] = src;
            if (idx < 2u) {
              const unsigned rec = hval[h];
              if (!(rec >> 31)) reinterpret_cast<unsigned*>(&w.coldlist[2 * (size_t)rec + 1])[2 + idx] = src;
            }
          }
        }
      } else if (cached) {
#pragma unroll
        for (int k = 0; k < EB; ++k)
          if (cge[k] != 0xFFFFFFFFu) place(cge[k], cslot[k]);
      } else {
        for (unsigned x = tid; x < E; x += TBK) {
          const size_t ge = seg_entry(tpre, tstart, NT, x);
          const long long key = w.ent_key[ge];
          if (!in_round(key, R, round)) continue;
          bool first;
          place((unsigned)ge, lds_key_slot<HSK>(hkey, &lsent, key, false, &first));
        }
      }
    }
    __syncthreads();
    KV_STAMPP(4);
  }
  if (tid == 0) {
    if (REC) w.pmeta[p] = make_uint4(lchunk + litm, lchunk, pbase, E);
    atomicAdd(&w.ctr[5], lcold + lhot);   // distinct keys of the batch (no value returned: nothing waits for it)
  }
}
template <int MODE, bool REC = true>
__global__ void __launch_bounds__(TBK, 4) k_part2(WsDev w, PartArgs a) { part2_body<MODE, REC>(w, a); }

// ------------------------------------------------------------------------------------------
// items2_body: the dense work-item directory, every hot chunk of the batch in front of every cold batch
// ------------------------------------------------------------------------------------------
// Waves take items round-robin, so with the hot chunks first every wave gets at most one more of them than any
// other (a random mix left some waves with four, and the kernel ended when they did).  pmeta[q] = {items, hot
// chunks, first entry, entries}: the chunk items of partition q are litem[first + j], its cold batches
// litem[first + entries - 1 - j].
template <int NW>
__device__ __forceinline__ void items2_body(const WsDev& w, unsigned nib) {
  __shared__ unsigned sit[MAX_P + 1], sck[MAX_P + 1];
  __shared__ unsigned wt[8];
  const unsigned P = w.P;
  const int tid = threadIdx.x, T = NW * 64;
  const unsigned per = (P + T - 1) / T;
  const unsigned p0 = min(P, tid * per), p1 = min(P, p0 + per);
  unsigned si = 0, sc = 0;
  for (unsigned q = p0; q < p1; ++q) { const uint4 m = w.pmeta[q]; sit[q] = m.x - m.y; sck[q] = m.y; si += m.x - m.y; sc += m.y; }
  unsigned ti, tc;
  unsigned ri = block_excl_scan<NW>(si, wt, &ti);   // ti: cold batches of the batch
  unsigned rc = block_excl_scan<NW>(sc, wt, &tc);   // tc: hot chunks
  for (unsigned q = p0; q < p1; ++q) { const unsigned a_ = sit[q], b_ = sck[q]; sit[q] = ri; sck[q] = rc; ri += a_; rc += b_; }
  __syncthreads();
  if (blockIdx.x == 0 && tid == 0) { w.ctr[2] = ti + tc; w.ctr[3] = tc; }
  const unsigned total = ti + tc;
  const unsigned nblk = min(nib, gridDim.x);
  const unsigned ipb = (total + nblk - 1) / nblk;
  const unsigned i0 = min(total, blockIdx.x * ipb), i1 = min(total, i0 + ipb);
  for (unsigned i = i0 + tid; i < i1; i += T) {
    const bool hot = i < tc;
    const unsigned x = hot ? i : i - tc;
    const unsigned* pre = hot ? sck : sit;
    unsigned lo = 0, hi = P;
    while (hi - lo > 1) {
      const unsigned mid = (lo + hi) >> 1;
      if (pre[mid] <= x) lo = mid; else hi = mid;
    }
    const uint4 m = w.pmeta[lo];
    const unsigned j = x - pre[lo];
    uint4 it = w.litem[hot ? m.z + j : m.z + m.w - 1u - j];
    if (hot) it.z += sck[lo];   // the chunk's number in the batch (its row of hpart)
    w.items[i] = it;
  }
}

// ------------------------------------------------------------------------------------------
// k_tsum: per tile, the gradient sums of the entries that have more than one occurrence
// ------------------------------------------------------------------------------------------
// One block of TBC threads per tile.  The tile kernel left the rows of those entries in entry order (mrow: position,
// epart row of the entry, head flag), so the sums are a SEGMENTED REDUCTION over one list: wave w takes a contiguous
// share of the rows, RS = G * RB rows per step (lane group g the RB consecutive rows g * RB ..), every wave the same
// number whatever the skew — the hottest key's 370 rows of a tile are six waves' work, not one wave's.
//   inside a group  rows are added in order; a run between two heads inside the group is complete: stored
//   across groups   a segmented scan over the groups' open runs (shuffles), the wave's open run carried from step to
//                   step in registers
//   across waves    a wave's first run (it began in an earlier wave) and its last (it may go on) meet in LDS, wave
//                   by wave in order
// The order of the additions depends on nothing but the list.  The positions of the next step are requested with
// the rows of this one.  ITEM_BLOCKS blocks in front build the dense work-item directory (items2_body).
constexpr int TBC = 512;
// FROM_LDS: the tile's mrow image is read from LDS at mrow_l — the tile pass's own image (k_ltsum: LtSmem::mr), or the
// copy k_tsum stages while the count is still on its way (one round trip in front of the rows instead of two); mc_l = mcount[tile]
template <int V, int LPR, int K, bool FROM_LDS = false>
__device__ __forceinline__ void tsum_body(const WsDev& w, const float* __restrict__ grad, int D, unsigned tile,
                                          const unsigned* mrow_l = nullptr, unsigned mc_l = 0u) {
  constexpr int G = 64 / LPR;
  constexpr int RB = (8 / K) > 0 ? (8 / K) : 1;
  constexpr int RS = G * RB;            // rows per wave and step
  constexpr int NWC = TBC / 64;
  constexpr int RF = K * V * LPR;       // floats of a row image in LDS
  __shared__ float lfs[2][NWC][RF];     // [0]: a wave's rows before its first head; [1]: its last run
  __shared__ unsigned lmeta[NWC][2];    // {1 = the wave saw a head, epart row of its last run}
  const unsigned mc = FROM_LDS ? mc_l : w.mcount[tile];
  const unsigned nm = mc & 0xFFFFu;
  if (nm == 0u) return;   // block-uniform
  const unsigned* const mrow_g = w.mrow + (size_t)tile * TILE;   // (two pointers: a select between LDS and global memory
                                                                 //  would make every read a flat load)
  const float* g0 = grad + (size_t)tile * TILE * D;
  const float* g0s = w.grad_self + (size_t)tile * TILE * D;   // (only dereferenced for positions inside the self range)
  const unsigned tbase = tile * (unsigned)TILE;
  float* ep = w.epart + (size_t)tile * (TILE / 2) * D;
  const int wl = threadIdx.x & 63, lane = wl % LPR, g = wl / LPR, wv = threadIdx.x >> 6;
  int eoff[K];   // a lane past the row's end reads element 0 instead of branching around the load
  bool evalid[K];
#pragma unroll
  for (int k = 0; k < K; ++k) { const int e0 = (lane + k * LPR) * V; evalid[k] = e0 < D; eoff[k] = evalid[k] ? e0 : 0; }
  auto store_row = [&](unsigned slot, const float (&v)[K][V]) {
    float* dst = ep + (size_t)slot * D;
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (evalid[k]) stv<V>(dst + eoff[k], v[k]);
  };
  // the wave's share: whole steps, the same number for every wave
  const unsigned steps = (nm + (unsigned)(NWC * RS) - 1u) / (unsigned)(NWC * RS);
  const unsigned r0 = (unsigned)wv * steps * RS, r1 = min(nm, r0 + steps * RS);
  float carry[K][V];          // the wave's open run so far
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int cc = 0; cc < V; ++cc) carry[k][cc] = 0.f;
  bool seen = false;          // a head was met in this wave: `carry` began here
  unsigned last_slot = 0;     // epart row of the last row taken so far
  unsigned mw[RB], mn[RB];
  auto ldm = [&](unsigned rbase, unsigned (&m)[RB]) {
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const unsigned r = rbase + g * RB + i, rc = r < r1 ? r : (r0 < nm ? r0 : 0u);
      if constexpr (FROM_LDS) m[i] = mrow_l[rc]; else m[i] = mrow_g[rc];
    }
  };
  if (r0 < r1) ldm(r0, mw);
  for (unsigned rb = r0; rb < r1; rb += RS) {   // wave-uniform
    float val[RB][K][V];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const float* src = ((tbase + (mw[i] & 0x7FFu) - w.self_lo < w.self_len) ? g0s : g0) + (size_t)(mw[i] & 0x7FFu) * D;
#pragma unroll
      for (int k = 0; k < K; ++k) ldv_stream<V>(src + eoff[k], val[i][k]);
    }
    ldm(rb + RS, mn);
    // ---- inside the group: runs in row order ----
    float acc[K][V], pfx[K][V];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int cc = 0; cc < V; ++cc) { acc[k][cc] = 0.f; pfx[k][cc] = 0.f; }
    bool has_head = false;
    unsigned cur_slot = 0, first_slot = 0;   // first_slot: epart row of the group's first row
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const bool ok = rb + g * RB + i < r1;
      const bool head = ok && (mw[i] >> 31) != 0u;
      const unsigned slot = (mw[i] >> 11) & 0x3FFu;
      if (i == 0) first_slot = slot;
      if (head) {
        if (!has_head) {
#pragma unroll
          for (int k = 0; k < K; ++k)
#pragma unroll
            for (int cc = 0; cc < V; ++cc) pfx[k][cc] = acc[k][cc];
          has_head = true;
        } else {
          store_row(cur_slot, acc);   // a run that begins and ends inside the group
        }
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
          for (int cc = 0; cc < V; ++cc) acc[k][cc] = 0.f;
      }
      if (ok) cur_slot = slot;
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int cc = 0; cc < V; ++cc) acc[k][cc] += ok ? val[i][k][cc] : 0.f;
    }
    // the group's contribution to the run that is open at its end: its suffix, or all of it when it has no head
    // (then pfx is not set: the whole group is acc)
    // ---- across the groups: inclusive segmented scan of (acc, has_head) ----
    float sc[K][V];
    bool sf = has_head;
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int cc = 0; cc < V; ++cc) sc[k][cc] = acc[k][cc];
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) {
      const bool of = __shfl_up((int)sf, o) != 0;
      const bool take = wl >= o && !sf;
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int cc = 0; cc < V; ++cc) {
          const float x = __shfl_up(sc[k][cc], o);
          if (take) sc[k][cc] += x;
        }
      if (wl >= o) sf = sf || of;
    }
    // what arrives at the group's first row: the scan of the groups before it, plus the wave's carry if none of
    // them had a head
    bool pf = false;            // a head in an earlier group of this step
    float arr[K][V];
    {
      const bool pf_ = __shfl_up((int)sf, LPR) != 0;
      pf = g > 0 && pf_;
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int cc = 0; cc < V; ++cc) {
          const float x = __shfl_up(sc[k][cc], LPR);
          arr[k][cc] = (g > 0 ? x : 0.f) + (pf ? 0.f : carry[k][cc]);
        }
    }
    const unsigned prev_slot_ = __shfl_up(cur_slot, LPR);
    const unsigned arr_slot = g > 0 ? prev_slot_ : last_slot;   // epart row of the run that arrives
    const bool arr_here = pf || seen;                           // it began in this wave
    if (has_head) {
      // the arriving run ends at this group's first head: arriving sum + the group's rows before that head
      float tot[K][V];
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int cc = 0; cc < V; ++cc) tot[k][cc] = arr[k][cc] + pfx[k][cc];
      const bool has_pfx = (mw[0] >> 31) == 0u;   // the group's first row continues the arriving run
      const unsigned slot = has_pfx ? first_slot : arr_slot;
      if (arr_here) {
        if (has_pfx || g > 0 || rb > r0 || true) store_row(slot, tot);
      } else {
        // the wave's first run: it began in an earlier wave (or this is the tile's very first row: nothing arrives)
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
          for (int cc = 0; cc < V; ++cc) lfs[0][wv][(k * LPR + lane) * V + cc] = tot[k][cc];
      }
    }
    // ---- the wave's carry for the next step: the scan at the last group ----
    {
      const int lastl = (G - 1) * LPR + lane;
      const bool any = __shfl((int)sf, lastl) != 0;
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int cc = 0; cc < V; ++cc) {
          const float x = __shfl(sc[k][cc], lastl);
          carry[k][cc] = any ? x : carry[k][cc] + x;
        }
      seen = seen || any;
      // epart row of the last row taken: the last group that had a row in range
      const unsigned ls = __shfl(cur_slot, lastl);
      const unsigned nrows = min((unsigned)RS, r1 - rb);
      const int lg = (int)((nrows - 1u) / RB);
      last_slot = __shfl(cur_slot, lg * LPR + lane);
      (void)ls;
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) mw[i] = mn[i];
  }
  // ---- the wave's ends meet in LDS ----
  if (g == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int cc = 0; cc < V; ++cc) {
        if (!seen) lfs[0][wv][(k * LPR + lane) * V + cc] = carry[k][cc];   // no head at all: everything continues the arriving run
        lfs[1][wv][(k * LPR + lane) * V + cc] = carry[k][cc];
      }
    if (lane == 0) { lmeta[wv][0] = (r0 < r1 ? 2u : 0u) | (seen ? 1u : 0u); lmeta[wv][1] = last_slot; }
  }
  __syncthreads();
  if (wv == 0 && g == 0) {
    float run[K][V];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int cc = 0; cc < V; ++cc) run[k][cc] = 0.f;
    unsigned rslot = 0;
    bool active = false;
    for (int x = 0; x < NWC; ++x) {
      const unsigned mt = lmeta[x][0];
      if (!(mt & 2u)) break;            // waves past the list's end
      if (mt & 1u) {
        // wave x met a head: the run that arrived ends there
        if (active) {
#pragma unroll
          for (int k = 0; k < K; ++k)
#pragma unroll
            for (int cc = 0; cc < V; ++cc) run[k][cc] += lfs[0][x][(k * LPR + lane) * V + cc];
          store_row(rslot, run);
        }
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
          for (int cc = 0; cc < V; ++cc) run[k][cc] = lfs[1][x][(k * LPR + lane) * V + cc];
        rslot = lmeta[x][1];
        active = true;
      } else {
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
          for (int cc = 0; cc < V; ++cc) run[k][cc] += lfs[0][x][(k * LPR + lane) * V + cc];
      }
    }
    if (active) store_row(rslot, run);
  }
}
// what: 0 = the directory blocks in front of the tiles' blocks; 1 = the tiles only; 2 = the directory only (overlap
// mode: the sums run beside the partition pass, the directory behind it)
template <int V, int LPR, int K>
__global__ void __launch_bounds__(TBC) k_tsum(TableDev t, WsDev w, const float* __restrict__ grad, int what) {
  if (*reinterpret_cast<volatile unsigned*>(&t.counters[1])) return;   // the index pass gave up on this batch
  KV_STAMPT(0);
  const unsigned nib = what == 1 ? 0u : w.nib;
  if (blockIdx.x < nib) { items2_body<TBC / 64>(w, nib); KV_STAMPT(1); return; }
  // the tile's mrow image (TILE words = 16 bytes per thread) is requested together with its count
  static_assert(TILE * 4 == TBC * 16, "one uint4 of mrow per thread");
  extern __shared__ __attribute__((aligned(16))) char ts_smem_raw[];
  const unsigned tile = blockIdx.x - nib;
  const uint4 img = reinterpret_cast<const uint4*>(w.mrow + (size_t)tile * TILE)[threadIdx.x];
  const unsigned mc = w.mcount[tile];
  reinterpret_cast<uint4*>(ts_smem_raw)[threadIdx.x] = img;
  __syncthreads();
  tsum_body<V, LPR, K, true>(w, grad, t.dim, tile, reinterpret_cast<const unsigned*>(ts_smem_raw), mc);
  KV_STAMPT(1);
}

// ------------------------------------------------------------------------------------------
// k_apply2: the fused optimizer update over the entry list
// ------------------------------------------------------------------------------------------
// Same work items as k_apply (kv_kernels.h) — a hot chunk per wave, a cold batch of 64 / LPR keys per wave, dealt
// round-robin — but built for a launch in which every dependent hop costs 3-5 us:
//   * the NEXT item's descriptor and records are requested before the current item is worked on (two-deep software
//     pipeline), so an item pays one exposed hop: its rows;
//   * a cold batch holds keys of one class: one source (75 % of the keys), two, or 3 .. LCOLD; the first two sources
//     come inside the record, so only the last class walks the entry list;
//   * every load of a batch is unconditional (a lane group without a key reads row 0) — a branch around a load makes
//     the compiler wait for the loads before it;
//   * a hot key's state rows are requested with its first gradient rows, not behind its last.
constexpr int TBA = 1024;   // 16 waves share a block's items
template <int OPT, int V, int LPR, int K>
__device__ __forceinline__ void apply2_body(const WsDev& w, const PartArgs& a) {
  const unsigned errflag = *reinterpret_cast<volatile unsigned*>(&a.tv.counters[1]);
  const unsigned total = w.ctr[2];
  if (errflag) return;
  const int D = a.tv.dim;
  constexpr int G = 64 / LPR;
  constexpr int RB = (8 / K) > 0 ? (8 / K) : 1;
  constexpr int RC = (2 / K) > 0 ? (2 / K) : 1;
  const int wl = threadIdx.x & 63;
  const int lane = wl % LPR;
  const int g = wl / LPR;
  // the two bases in registers (a select between a.epart and a.grad themselves becomes a per-lane LOAD of the
  // pointer out of the argument block, with a wait behind it, in front of every row); a lane past the row's end
  // reads element 0 instead of branching around the load
  const float* const gbase = a.grad;
  const float* const ebase = a.epart;
  int eoff[K];
#pragma unroll
  for (int k = 0; k < K; ++k) { const int e0 = (lane + k * LPR) * V; eoff[k] = e0 < D ? e0 : 0; }
  auto load_row = [&](unsigned pos, float (&dst)[K][V]) {
    const float* base = (pos & EP_TAG) ? ebase : gbase;
    const float* src = base + (size_t)(pos & ~EP_TAG) * D;
#pragma unroll
    for (int k = 0; k < K; ++k) ldv_stream<V>(src + eoff[k], dst[k]);
  };
  // ---- the lean update.  When the var and its hinted slot table are single-chunk tables without delta tracking
  //      (every pre-sized training table), a key whose slot-row hint stands up is updated with nothing but the
  //      base pointers below, all in registers; any other key of the batch takes the general finish_key ----------
  const bool fast = (OPT != OPT_FTRL) && a.tv.single != 0u && a.ts0.single != 0u && a.use_hints != 0 &&
                    (a.tv.track_delta | a.ts0.track_delta) == 0u;
  float* const vrows = a.tv.c0.rows;
  RowMeta* const vmeta = a.tv.c0.meta;
  float* const srows = a.ts0.c0.rows;
  RowMeta* const smeta = a.ts0.c0.meta;
  const int SD = a.ts0.dim;
  const unsigned smax = a.ts0.max_rows, thr = a.tv.enter_threshold;
  const bool need_vmeta = OPT == OPT_ADAGRAD || thr != 0u;
  constexpr int NS0 = (OPT == OPT_ADAM_V4 || OPT == OPT_ADAM_V3) ? 3 : 1;
  // the state of one key, every load unconditional (no key: row 0 of both tables)
  auto prefetch_fast = [&](const uint4& rq, bool live, RowMeta& m0, uint2& vm, PreRows<V, K>& pre) {
    const unsigned row = live ? (rq.z & ROW_MASK) : 0u;
    const unsigned hint = (live && rq.w < smax) ? rq.w : 0u;
    const uint4 mm = *reinterpret_cast<const uint4*>(smeta + hint);
    m0.key = (long long)(((unsigned long long)mm.y << 32) | mm.x);
    m0.freq = mm.z;
    m0.flags = (unsigned char)(mm.w & 0xFFu);
    vm = make_uint2(0u, 0u);
    if (need_vmeta) vm = *reinterpret_cast<const uint2*>(&vmeta[row].freq);   // uniform
    const float* xr = vrows + (size_t)row * D;
    const float* sr = srows + (size_t)hint * SD;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      ldv<V>(xr + eoff[k], pre.x[k]);
#pragma unroll
      for (int b3 = 0; b3 < NS0; ++b3) ldv<V>(sr + b3 * D + eoff[k], pre.s[b3][k]);
    }
  };
  // Block b owns items b, b + NB, b + 2 NB, ... (the directory lists every hot chunk before any cold batch, so
  // each block's list starts with its hot chunks); its waves take them through a ticket in LDS: a wave that drew a
  // long item simply takes fewer.  One item is held ahead (descriptor and records), no more: what a wave holds
  // nobody else can take.
  __shared__ unsigned lnext;
  if (threadIdx.x == 0) lnext = 0;
  __syncthreads();
  const unsigned NB = gridDim.x;
  const unsigned nmine = total > blockIdx.x ? (total - blockIdx.x + NB - 1u) / NB : 0u;
  auto pull = [&]() -> unsigned {
    unsigned j = 0;
    if (wl == 0) j = atomicAdd(&lnext, 1u);
    j = (unsigned)__shfl((int)j, 0);
    return j < nmine ? blockIdx.x + j * NB : 0xFFFFFFFFu;
  };
  // an item's records: hot chunk -> the key's two record words (every lane the same); cold batch -> the lane group's
  // key (a group past the batch's last key reads the batch's first record and is masked later)
  auto load_item = [&](unsigned it) -> uint4 { return w.items[it != 0xFFFFFFFFu ? it : 0u]; };
  auto load_rec = [&](const uint4& item, uint4& ra, uint4& rb) {
    const bool hot = (item.x & HEAD_BIT) != 0u;
    const uint4* list = hot ? w.hotlist : w.coldlist;
    const unsigned u = hot ? (item.x & ~HEAD_BIT) : item.x + ((unsigned)g < item.y ? (unsigned)g : 0u);
    ra = list[2 * (size_t)u];
    rb = list[2 * (size_t)u + 1];
  };
  unsigned cur = pull();
  if (cur == 0xFFFFFFFFu) return;
  uint4 item = load_item(cur), ra, rb;
  load_rec(item, ra, rb);
#ifdef KV_STAMPS
  unsigned long long st_t0 = wall_clock64(), st_hot = 0, st_cold = 0, st_nh = 0, st_nc = 0, st_wait = 0;
#endif
  while (cur != 0xFFFFFFFFu) {
    // the next item: its descriptor leaves now, its records once this item's rows are on their way
    const unsigned nxt = pull();
    const uint4 item_n = load_item(nxt);
    uint4 ra_n, rb_n;
#ifdef KV_STAMPS
    const unsigned long long st_a = wall_clock64();
#endif
    const bool is_hot = (item.x & HEAD_BIT) != 0u;
    float gv[K][V];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int cc = 0; cc < V; ++cc) gv[k][cc] = 0.f;
    RowMeta m0{};
    uint2 vm = make_uint2(0u, 0u);
    bool hint_loaded = false, have_x = false, have_s = false;
    PreRows<V, K> pre;
    bool fin_live = false;
    if (is_hot) {
      // ---- hot chunk: entries [lo, hi) of one key, G * RB of them per step ------------------------------------
      const unsigned lo = rb.x + item.y * rb.w, hi = min(rb.x + rb.y, lo + rb.w);
      constexpr int SR = G * RB;
      const unsigned nst = (hi - lo + SR - 1) / SR;
      const unsigned nch = (rb.y + rb.w - 1u) / rb.w;
      auto ldpos = [&](unsigned st, unsigned (&pp)[RB]) {
#pragma unroll
        for (int j = 0; j < RB; ++j) {
          const unsigned idx = lo + st * SR + j * G + g;
          pp[j] = w.order[idx < hi ? idx : lo] & ~HEAD_BIT;   // unconditional: a slot past the end re-reads the first source, masked below
        }
      };
      unsigned pa_[RB], pb_[RB];
      float va[RB][K][V];
      ldpos(0, pa_);
      // a key with a single chunk is finished here: its state rows leave with the first gradient rows
      if (nch == 1u) {
        if (fast) prefetch_fast(ra, g == 0, m0, vm, pre);
        else prefetch_state<OPT, V, LPR, K>(a, ra, g == 0, lane, D, m0, hint_loaded, pre, have_x, have_s);
      }
      for (unsigned st = 0; st < nst; ++st) {
#pragma unroll
        for (int j = 0; j < RB; ++j) load_row(pa_[j], va[j]);
        ldpos(st + 1, pb_);
#pragma unroll
        for (int j = 0; j < RB; ++j) {
          const bool ok = lo + st * SR + j * G + g < hi;
#pragma unroll
          for (int k = 0; k < K; ++k)
#pragma unroll
            for (int cc = 0; cc < V; ++cc) gv[k][cc] += ok ? va[j][k][cc] : 0.f;
          pa_[j] = pb_[j];
        }
      }
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) {
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
          for (int cc = 0; cc < V; ++cc) gv[k][cc] += __shfl_xor(gv[k][cc], o);
      }
      fin_live = g == 0;
      if (nch > 1u) {   // the key's chunks meet in k_apply_fin
        if (g == 0) {
          float* dst = w.hpart + (size_t)item.z * D;
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const int e0 = (lane + k * LPR) * V;
            if (e0 < D) stv<V>(dst + e0, gv[k]);
          }
        }
        fin_live = false;
      }
    } else {
      // ---- cold batch: one key per lane group, all of one class (item.z: 1, 2 or 3 = up to LCOLD sources) -------
      const bool live = (unsigned)g < item.y;
      const unsigned start = rb.x, cnt = live ? rb.y : 0u;
      const unsigned cls = item.z;
      float g2[K][V];
      load_row(rb.z, gv);                       // (a group without a key reads the first key's row: masked by `live`)
      if (cls >= 2u) load_row(rb.w, g2);        // uniform over the wave
      if (fast) {
        prefetch_fast(ra, live, m0, vm, pre);
      } else {
        // the state: row 0 of each table for a group without a key (no branch around the loads)
        uint4 rq = ra;
        if (!live) { rq.z = 0u; rq.w = 0u; }
        prefetch_state<OPT, V, LPR, K>(a, rq, true, lane, D, m0, hint_loaded, pre, have_x, have_s);
      }
      if (cls >= 2u) {
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
          for (int cc = 0; cc < V; ++cc) gv[k][cc] += g2[k][cc];
      }
      if (cls >= 3u) {
        for (unsigned j0 = 2; j0 < cnt; j0 += RC) {
          float val[RC][K][V];
          unsigned pos[RC];
#pragma unroll
          for (int j = 0; j < RC; ++j) pos[j] = w.order[start + (j0 + j < cnt ? j0 + j : 0u)] & ~HEAD_BIT;
#pragma unroll
          for (int j = 0; j < RC; ++j) load_row(pos[j], val[j]);
#pragma unroll
          for (int j = 0; j < RC; ++j) {
            const bool ok = j0 + j < cnt;
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
              for (int cc = 0; cc < V; ++cc) gv[k][cc] += ok ? val[j][k][cc] : 0.f;
          }
        }
      }
      fin_live = live;
    }
    load_rec(item_n, ra_n, rb_n);
#ifdef KV_STAMPS
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    const unsigned long long st_b = wall_clock64();
    if (!is_hot) st_wait += st_b - st_a;
#endif
    // ONE copy of the update for both kinds of item (two would double the kernel's registers)
    bool general = fin_live;
    if (fast) {
      const long long key = (long long)(((unsigned long long)ra.y << 32) | ra.x);
      const unsigned row = ra.z & ROW_MASK;
      const unsigned hint = ra.w < smax ? ra.w : 0u;
      // the hint stands up: the slot row carries this key and is not released (what resolve_rows checks)
      const bool ok = fin_live && row != 0u && hint != 0u && m0.key == key && !(m0.flags & FLAG_FREE);
      bool act = ok;
      if (need_vmeta && ok && !(ra.z >> 31)) {   // frequency filter / un-blacklisting (resolve_rows; kv_variable.h:910)
        if ((vm.x & 0xFFFFu) < thr) act = false;
        else if ((vm.y & FLAG_BLACK) && lane == 0) vmeta[row].flags = FLAG_UNDER;
      }
      if (act && lane == 0) {   // AddFrequency(1, today) on the slot row (kv_variable.h:409-414)
        unsigned lo = (m0.freq & 0xFFFFu) + 1u;
        if (lo > 65535u) lo = 65535u;
        smeta[hint].freq = (a.day << 16) | lo;
      }
      const unsigned rr = act ? row : 0u, hh = act ? hint : 0u;
      opt_core<OPT, V, LPR, K>(vrows + (size_t)rr * D, srows + (size_t)hh * SD, nullptr, &vmeta[rr].flags, &smeta[hh].flags,
                               nullptr, act, false, D, gv, a.opt, lane, pre.x, pre.s);
      general = fin_live && !ok;
      hint_loaded = hint != 0u; have_x = true; have_s = hint != 0u;
    }
    if (!fast || __ballot(general) != 0ull)
      finish_key<MODE_APPLY, OPT, V, LPR, K>(a, ra, general, hint_loaded && general, m0, gv, lane, &pre, have_x && general, have_s && general);
#ifdef KV_STAMPS
    {
      const unsigned long long now = wall_clock64();
      if (is_hot) { st_hot += now - st_a; ++st_nh; } else { st_cold += now - st_a; ++st_nc; }
    }
#endif
    item = item_n; ra = ra_n; rb = rb_n; cur = nxt;
  }
#ifdef KV_STAMPS
  if (wl == 0) {
    unsigned long long* d = w.dbg + (size_t)(8192 + blockIdx.x * (TBA / 64) + (threadIdx.x >> 6)) * 16;
    d[0] = st_t0; d[1] = wall_clock64(); d[2] = st_hot; d[3] = st_cold; d[4] = st_nh; d[5] = st_nc; d[6] = total; d[7] = w.ctr[3]; d[8] = st_wait;
  }
#endif
}
template <int OPT, int V, int LPR, int K>
__global__ void __launch_bounds__(TBA, (K == 1 ? 4 : 1)) k_apply2(WsDev w, PartArgs a) { apply2_body<OPT, V, LPR, K>(w, a); }

// ------------------------------------------------------------------------------------------
// k_copy: the training lookup's output rows by themselves (overlap mode)
// ------------------------------------------------------------------------------------------
// goz_wave (kv_kernels.h) with the training lookup's answer for a key the table does not hold yet: the init rule's
// value (kv_variable.h:889-898) instead of zeros.  It runs BESIDE k_ltile<GATHER = false> of the same batch, which
// inserts those keys: whatever state of a new key's index entry a probe meets — empty, claimed, published with
// HINT_NEW — the answer is the init value, and the row of a key that was there before the batch is not written by
// anything the lookup runs.  One wave per 64 ids and step, lane l probes id l, the rows go VQ lanes per row.
template <typename IdT, int VQ>
__global__ void __launch_bounds__(TB) k_copy(TableDev t, const IdT* __restrict__ ids, float* __restrict__ out, long long n) {
  constexpr int RW = 64 / VQ;
  constexpr int CW = VQ < 8 ? VQ : 8;
  const int lane = threadIdx.x & 63;
  const int v = lane % VQ, sub = lane / VQ;
  const long long wave = (long long)blockIdx.x * (TB / 64) + (threadIdx.x >> 6);
  const long long stride = (long long)gridDim.x * (TB / 64) * 64;
  const bool single = single_chunk(t);
  const float4* rows0 = reinterpret_cast<const float4*>(t.c0.rows);
  for (long long r0 = wave * 64; r0 < n; r0 += stride) {
    const long long i = r0 + lane;
    const long long key = (long long)ids[i < n ? i : n - 1];
    const unsigned long long p = home_of(t, key, mix64((unsigned long long)key));
    const Entry e = load_entry(&t.entries[p]);
    unsigned hint = 0;
    unsigned rr = table_find_from(t, key, p, e, &hint);
    if (rr == 0u || hint == HINT_NEW) rr = NEW_BIT;
    const bool anynew = __ballot((rr & NEW_BIT) != 0u) != 0ull;
#pragma unroll
    for (int j0 = 0; j0 < VQ; j0 += CW) {
      float4 val[CW];
      unsigned rj[CW];
#pragma unroll
      for (int j = 0; j < CW; ++j) rj[j] = __shfl(rr, (j0 + j) * RW + sub);
      if (single) {
#pragma unroll
        for (int j = 0; j < CW; ++j) val[j] = rows0[(size_t)(rj[j] & ROW_MASK) * VQ + v];
      } else {
#pragma unroll
        for (int j = 0; j < CW; ++j) val[j] = reinterpret_cast<const float4*>(row_ptr(t, rj[j] & ROW_MASK))[v];
      }
      if (__builtin_expect(anynew, 0)) {
#pragma unroll
        for (int j = 0; j < CW; ++j) {
          const long long kj = __shfl(key, (j0 + j) * RW + sub);
          if (rj[j] & NEW_BIT) {
            const unsigned long long h = pick64((unsigned long long)kj ^ (t.seed * 0x9E3779B97F4A7C15ULL));
            const float4 a = reinterpret_cast<const float4*>(t.init_table + (size_t)((unsigned)h % t.init_rows) * t.dim)[v];
            const float4 b = reinterpret_cast<const float4*>(t.init_table + (size_t)((unsigned)(h >> 32) % t.init_rows) * t.dim)[v];
            val[j] = make_float4((a.x + b.x) * 0.5f, (a.y + b.y) * 0.5f, (a.z + b.z) * 0.5f, (a.w + b.w) * 0.5f);
          }
        }
      }
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

// ------------------------------------------------------------------------------------------
// k_lrows: the training lookup's output rows by per-position probe; the tile pass is deferred
// ------------------------------------------------------------------------------------------
// What a training lookup RETURNS needs no de-duplication: out[i] = the row of ids[i], and for a key the table does not
// hold yet the init rule's value, a function of (key, seed) (kv_variable.h:889-898).  goz_wave (kv_kernels.h) with that
// answer for absent keys: a wave takes 64 ids per step, lane l probes id l, the rows go VQ lanes per row with streaming
// stores; rows of absent keys (or of keys whose index entry is being published: HINT_NEW) are filled in a second
// pass over the step, so the common path is the inference gather's.  The ids are copied to `ids_copy` on the way: the
// tile pass that inserts the new keys, counts frequencies and builds the batch index (k_ltile<GATHER = false>, or
// k_ltsum in front of the optimizer apply) runs later, when the caller's ids may be gone.
template <typename IdT, int VQ, int CWMAX = 4>
__device__ __forceinline__ void lrows_wave(const TableDev& t, const IdT* __restrict__ ids, IdT* __restrict__ ids_copy,
                                           float* __restrict__ out, long long n, long long wave, long long nwaves) {
  constexpr int RW = 64 / VQ;
  constexpr int CW = VQ < CWMAX ? VQ : CWMAX;
  const int lane = threadIdx.x & 63;
  const int v = lane % VQ, sub = lane / VQ;
  const long long stride = nwaves * 64;
  long long r0 = wave * 64;
  if (r0 >= n) return;
  auto load_raw = [&](long long i) -> IdT { return i < n ? ids[i] : (IdT)0; };
  IdT raw1 = load_raw(r0 + lane), raw2 = load_raw(r0 + stride + lane);
  long long k1 = (long long)raw1;
  unsigned long long p1 = home_of(t, k1, mix64((unsigned long long)k1));
  Entry e1 = load_entry(&t.entries[p1]);
  for (; r0 < n; r0 += stride) {
    const bool valid = r0 + lane < n;
    unsigned hint = 0;
    unsigned rr = valid ? table_find_from(t, k1, p1, e1, &hint) : 0u;
    const bool isnew = valid && (rr == 0u || hint == HINT_NEW);
    if (isnew) rr = NEW_BIT;
    if (ids_copy != nullptr && valid) ids_copy[r0 + lane] = raw1;
    const long long kcur = k1;
    // next step: its home entries leave now, the ids of the step after it too
    raw1 = raw2;
    k1 = (long long)raw1;
    p1 = home_of(t, k1, mix64((unsigned long long)k1));
    if (r0 + stride < n) e1 = load_entry(&t.entries[p1]);
    raw2 = load_raw(r0 + 2 * stride + lane);
#pragma unroll
    for (int j0 = 0; j0 < VQ; j0 += CW) {
      float4 val[CW];
      unsigned rj[CW];
#pragma unroll
      for (int j = 0; j < CW; ++j) rj[j] = __shfl(rr, (j0 + j) * RW + sub);
#pragma unroll
      for (int j = 0; j < CW; ++j) val[j] = reinterpret_cast<const float4*>(row_ptr(t, rj[j] & ROW_MASK))[v];
#pragma unroll
      for (int j = 0; j < CW; ++j) {
        const long long ii = r0 + (j0 + j) * RW + sub;
        if (ii < n && !(rj[j] >> 31)) {
          float4* dst = reinterpret_cast<float4*>(out + (size_t)ii * (VQ * 4)) + v;
          __builtin_nontemporal_store(val[j].x, &dst->x); __builtin_nontemporal_store(val[j].y, &dst->y);
          __builtin_nontemporal_store(val[j].z, &dst->z); __builtin_nontemporal_store(val[j].w, &dst->w);
        }
      }
    }
    if (__builtin_expect(__ballot(isnew) != 0ull, 0)) {
      for (int j = 0; j < VQ; ++j) {
        const unsigned rjn = __shfl(rr, j * RW + sub);
        const long long kj = __shfl(kcur, j * RW + sub);
        const long long ii = r0 + j * RW + sub;
        if ((rjn >> 31) && ii < n) {
          const unsigned long long h = pick64((unsigned long long)kj ^ (t.seed * 0x9E3779B97F4A7C15ULL));
          const float4 a = reinterpret_cast<const float4*>(t.init_table + (size_t)((unsigned)h % t.init_rows) * t.dim)[v];
          const float4 b = reinterpret_cast<const float4*>(t.init_table + (size_t)((unsigned)(h >> 32) % t.init_rows) * t.dim)[v];
          float4* dst = reinterpret_cast<float4*>(out + (size_t)ii * (VQ * 4)) + v;
          __builtin_nontemporal_store((a.x + b.x) * 0.5f, &dst->x); __builtin_nontemporal_store((a.y + b.y) * 0.5f, &dst->y);
          __builtin_nontemporal_store((a.z + b.z) * 0.5f, &dst->z); __builtin_nontemporal_store((a.w + b.w) * 0.5f, &dst->w);
        }
      }
    }
  }
}
template <typename IdT, int VQ>
__global__ void __launch_bounds__(TB) k_lrows(TableDev t, const IdT* __restrict__ ids, IdT* __restrict__ ids_copy,
                                              float* __restrict__ out, long long n) {
  lrows_wave<IdT, VQ>(t, ids, ids_copy, out, n, (long long)blockIdx.x * (TB / 64) + (threadIdx.x >> 6),
                      (long long)gridDim.x * (TB / 64));
}

// ------------------------------------------------------------------------------------------
// k_ltsum: the tile pass and the tile sums of an optimizer apply in one launch
// ------------------------------------------------------------------------------------------
// The apply has the gradient rows at hand when it runs the batch's tile pass (the lookup deferred it, or the optimizer
// meets the ids first): the block that de-duplicated a tile goes on to sum the rows of its repeated ids (tsum_body, fed
// from the mrow image still in LDS).  The dedup chain of one block — LDS phases, one probe round trip — overlaps with
// the row reads of the others; k_ltile<GATHER = false> + k_tsum one after the other were 23.5 + 24 us at configs[1].
template <typename IdT, int V, int LPR, int K, bool BUCKET = false>
__global__ void __launch_bounds__(TBT) k_ltsum(TableDev t, WsDev w, const IdT* __restrict__ ids, const int* __restrict__ counts,
                                              long long n, int det, const float* __restrict__ grad) {
  ltile_body<IdT, 1, false, BUCKET>(t, w, ids, counts, n, det, nullptr);
  __syncthreads();   // the tile's mrow image and mcount are written (and every phase of the tile pass is behind us)
  if (*reinterpret_cast<volatile unsigned*>(&t.counters[1])) return;   // the tile pass gave up on this batch
  extern __shared__ __attribute__((aligned(16))) char lt_smem_raw[];
  tsum_body<V, LPR, K, true>(w, grad, t.dim, blockIdx.x, carve_ltile(lt_smem_raw).mr, w.mcount[blockIdx.x]);
}

// ------------------------------------------------------------------------------------------
// the same kernels over many tables in one launch (blockIdx.y = table; arguments from the MultiDesc array)
// ------------------------------------------------------------------------------------------
template <typename IdT, int VQ, bool GATHER>
__global__ void __launch_bounds__(TBT) k_ltile_multi(const MultiDesc* __restrict__ descs) {
  const MultiDesc& m = descs[blockIdx.y];
  if (blockIdx.x >= m.w.ntiles) return;
  ltile_body<IdT, VQ, GATHER>(m.a.tv, m.w, reinterpret_cast<const IdT*>(m.ids), m.counts, m.n, m.a.det, m.out);
}
// the table-less tile pass of the sharded route (int64 ids), several tables in one launch
__global__ void __launch_bounds__(TBT) k_ltile_multi_notable(const MultiDesc* __restrict__ descs) {
  const MultiDesc& m = descs[blockIdx.y];
  if (blockIdx.x >= m.w.ntiles) return;
  ltile_body<long long, 1, false, false, true>(m.a.tv, m.w, reinterpret_cast<const long long*>(m.ids), nullptr, m.n, m.a.det, nullptr);
}
template <int MODE>
__global__ void __launch_bounds__(TBK, 4) k_part2_multi(const MultiDesc* __restrict__ descs) {
  const MultiDesc& m = descs[blockIdx.y];
  if (blockIdx.x >= m.w.P || m.n == 0) return;
  part2_body<MODE>(m.w, m.a);
}
// tiles_only: no directory blocks in front (the batch goes on to k_papply, which needs no work items)
template <int V, int LPR, int K>
__global__ void __launch_bounds__(TBC) k_tsum_multi(const MultiDesc* __restrict__ descs, int tiles_only) {
  const MultiDesc& m = descs[blockIdx.y];
  if (m.n == 0) return;
  if (*reinterpret_cast<volatile unsigned*>(&m.a.tv.counters[1])) return;
  const unsigned nib = tiles_only ? 0u : (unsigned)ITEM_BLOCKS;
  if (blockIdx.x < nib) { items2_body<TBC / 64>(m.w, (unsigned)ITEM_BLOCKS); return; }
  if (blockIdx.x - nib >= m.w.ntiles) return;
  tsum_body<V, LPR, K>(m.w, m.a.grad, m.a.tv.dim, blockIdx.x - nib);
}
template <int OPT, int V, int LPR, int K>
__global__ void __launch_bounds__(TBA, (K == 1 ? 4 : 1)) k_apply2_multi(const MultiDesc* __restrict__ descs) {
  const MultiDesc& m = descs[blockIdx.y];
  if (m.n == 0) return;
  apply2_body<OPT, V, LPR, K>(m.w, m.a);
}
