// kv_fused.h — the entry-list pipeline of the training step (included after kv_kernels.h).
//
// Measured on the box (profiles/r03_calibration.txt): random 128-B rows read at 5 TB/s whatever the order, the per-key
// state update of GroupAdam costs 27-35 us for 109 k keys by itself, so what can be removed is everything that is not a
// row.  An ENTRY is one distinct key of one tile (TILE consecutive ids); everything behind the tile pass works on
// entries, only the tile pass and the tile sums touch positions, and both stay inside their tile.
//
//   k_ltile   one block per tile: a table probe for every position, requested right behind the id loads; the output rows
//             (before or behind the dedup chain, by the block's residency generation); LDS dedup; a key the table does
//             not hold is inserted by its winner (a 64-bit CAS on the index entry decides between tiles, the loser needs
//             no row — the row of a new key is a function of (key, seed)); the tile's entries {key, occurrences | count,
//             row word, slot-row hint, source} counting-sorted by hash partition; the rows of the entries that occur more
//             than once, in entry order (mrow: position, epart row, head flag — built in LDS, coalesced).
//   k_tsum    per tile: the gradient rows of every entry with more than one occurrence are summed into epart by a
//             segmented reduction over mrow (all reads inside the tile's 256 KB of gradient rows, every wave the
//             same number of rows whatever the skew).  An entry with one occurrence IS its gradient row.
//   k_papply  (kv_papply.h) one block per hash partition over the tiles' entries: the lookup's bookkeeping and the fused
//             optimizer update of every key by its single owner.
//   k_part2   the bookkeeping alone, for a lookup that no apply takes over.
//   k_ltsum   k_ltile without rows + k_tsum in one launch (an apply that meets the ids first).
//
// Summation order: inside an entry by rank (LDS-atomic arrival; input order in deterministic mode), then the key's
// entries in source-list order (arrival in k_papply; (tile, key) order in deterministic mode): a fixed tree given those two.
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
// The position whose LDS insert created a key's slot is the key's WINNER: it takes the key's place in the partition sort,
// inserts a key the table does not hold yet and writes the key's entry.  No list of occupied slots is built.
//
// The winner's probe leaves right behind the hash insert and is collected behind the counting sort.
//
// Measured in round 5 and dropped (profiles/r05_ltile_orders.txt): a probe for EVERY position right behind the id loads
// (so that the output rows no longer depend on the dedup chain), with the rows copied in front of the chain by every
// block, by no block, or by the second resident block of every CU.  All three orders end within 2 us of each other
// (51-53 us under the stamps): the two blocks of a CU share one memory pipeline, the chain's few loads and stores queue
// behind the other block's row traffic (a chain-first block waited 12 us for probes it had issued at 3 us; a rows-first
// block's counting-sort phase, which only STORES toff and ent_key, took 14 us), and the row phase itself runs at the
// chip's write rate (128 MB in ~30 us).  The work is conserved whatever the order.
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
};
// lkeys' storage (+ 64 bytes) holds, once the hash insert is done: lrow [LS + 1], escan [TILE + 1] and the mrow image mr [TILE]
constexpr size_t LT_KEYS_BYTES = (size_t)(LS + 1) * 8 + 64;
static_assert(LT_KEYS_BYTES >= (((size_t)(LS + 1) * 4 + 15) & ~(size_t)15) + (((size_t)(TILE + 1) * 4 + 15) & ~(size_t)15) + (size_t)TILE * 4, "aliases fit");

__host__ __device__ inline size_t ltile_smem_bytes() {
  size_t b = LT_KEYS_BYTES + 16;          // lkeys (then lrow, escan, mr)
  b += (size_t)(LS + 1) * 4 + 16;         // lcnt
  b += (size_t)(LS + 1) * 2 + 16;         // lpos
  b += (size_t)(TILE + 1) * 2 + 16;       // lrun
  b += (size_t)(MAX_P + 1) * 4 + 16;      // hist
  b += 64;                                // wtot
  return b;
}
__device__ __forceinline__ LtSmem carve_ltile(char* base) {
  LtSmem s;
  auto take = [&](size_t bytes) { char* p = base; base += (bytes + 15) & ~(size_t)15; return p; };
  char* k0 = take(LT_KEYS_BYTES);
  s.lkeys = reinterpret_cast<long long*>(k0);
  s.lrow = reinterpret_cast<unsigned*>(k0);
  s.escan = reinterpret_cast<unsigned*>(k0 + (((size_t)(LS + 1) * 4 + 15) & ~(size_t)15));
  s.mr = reinterpret_cast<unsigned*>(k0 + (((size_t)(LS + 1) * 4 + 15) & ~(size_t)15) + (((size_t)(TILE + 1) * 4 + 15) & ~(size_t)15));
  s.lcnt = reinterpret_cast<unsigned*>(take((size_t)(LS + 1) * 4));
  s.lpos = reinterpret_cast<unsigned short*>(take((size_t)(LS + 1) * 2));
  s.lrun = reinterpret_cast<unsigned short*>(take((size_t)(TILE + 1) * 2));
  s.hist = reinterpret_cast<unsigned*>(take((size_t)(MAX_P + 1) * 4));
  s.wtot = reinterpret_cast<unsigned*>(take(64));
  return s;
}

// packed per-entry scan word, over the entries with more than one occurrence: their rows (bits 0..11, sum <= 2048) |
// their number (bits 12..22)
constexpr unsigned ES_POS = 0xFFFu, ES_NSH = 12;
// mrow words: tile-local position (bits 0..10) | the entry's epart row in the tile (bits 11..20) | bit 31: first row of its entry
static_assert(TILE <= (1 << 11), "mrow: the tile-local position is an 11-bit field");
static_assert(TILE / 2 <= (1 << 10), "mrow: the entry's epart row is a 10-bit field (an entry in mrow has >= 2 rows)");
static_assert(TILE <= (int)ES_POS && TILE / 2 < (1 << (23 - ES_NSH)), "escan: rows and entry numbers of a tile fit their fields");
static_assert(TILE <= 65535, "toff / hist / mcount pack per-tile counts into 16-bit halves");
constexpr int LT_TOUCH_MAXENT = TILE / 2;   // early row touches: only in tiles with at most this many distinct keys

// VQ = float4 per row (power of two <= 64); GATHER: copy the rows of the tile's positions to `out`.
// NOTABLE: no table behind the batch (the sharded route's index of the local ids): no probes, no inserts — the entries carry
// keys, counts and sources only.
template <typename IdT, int VQ, bool GATHER, bool NOTABLE = false>
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
  // ---- the probes (one 16-byte index entry each: row + slot-row hint).  `mask` bit k: position k asks; the others read
  //      entry 0 (no branch around the load) ---------------------------------------------------------------------------
  unsigned long long pp[IPT];
  Entry en[IPT];
  auto issue_probes = [&](unsigned mask) {
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
      pp[k] = 0ull;
      en[k] = Entry{0, 0u, 0u};
      if constexpr (!NOTABLE) {
        pp[k] = ((mask >> k) & 1u) ? home_of(t, kreg[k], mix64((unsigned long long)kreg[k])) : 0ull;
        en[k] = load_entry(&t.entries[pp[k]]);
      }
    }
  };
  // row word (NEW_BIT: the table does not hold the key yet, or a tile of this launch is inserting it — row part 0 when the
  // row is not known) and slot-row hint of the positions in `mask`
  unsigned rr[IPT], hn[IPT];
  float pf = 0.f;   // (sum of the early row touches: keeps the loads alive, see below)
#pragma unroll
  for (int k = 0; k < IPT; ++k) { rr[k] = 0u; hn[k] = 0u; }
  auto resolve_probes = [&](unsigned mask) {
    if constexpr (!NOTABLE) {
#pragma unroll
      for (int k = 0; k < IPT; ++k) {
        if (!((mask >> k) & 1u)) continue;
        unsigned hint = 0;
        unsigned r = table_find_from(t, kreg[k], pp[k], en[k], &hint);
        if (__builtin_expect(r == 0u, 0)) { r = NEW_BIT; hint = 0; }
        else if (__builtin_expect(hint == HINT_NEW, 0)) { r |= NEW_BIT; hint = 0; }
        rr[k] = r; hn[k] = hint;
      }
    }
  };
  for (int s = tid; s <= LS; s += TBT) { sm.lkeys[s] = EMPTY_KEY; sm.lcnt[s] = 0; }
  for (unsigned p = tid; p <= P; p += TBT) sm.hist[p] = 0;
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

  // ---- the output rows.  Wave wv, round k holds positions k * TBT + wv * 64 + lane in its own registers (rr) ------------
  auto output_rows = [&]() {
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
        auto issue = [&](int pc, float4 (&val)[CW]) {
          const int k = pc / SPK, j0 = (pc % SPK) * CW;
#pragma unroll
          for (int j = 0; j < CW; ++j) {
            const unsigned rj = __shfl(rr[k], (j0 + j) * RW + sub) & ROW_MASK;   // (skipped records, positions past the end: the zero row)
            if constexpr (SINGLE) val[j] = rows0[(size_t)rj * D4 + vv];
            else val[j] = reinterpret_cast<const float4*>(row_ptr(t, rj))[vv];
          }
        };
        auto flush = [&](int pc, float4 (&val)[CW]) {
          const int k = pc / SPK, j0 = (pc % SPK) * CW;
          const long long r0 = base + (long long)k * TBT + (tid & ~63);
          if (__builtin_expect(__ballot((rr[k] & NEW_BIT) != 0u) != 0ull, 0)) {
            // a key inserted by this batch: its row is the init rule's value (kv_variable.h:889-898), written to the
            // table by the partition pass; here it is computed, not read
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
  };
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
      tslot[k] = h;
    }
  }
  KV_STAMP(8);
  // ---- the winners' probes leave (one per distinct key; a position that is no winner asks for entry 0) ------------------
  issue_probes(win);
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
  // ---- the probes are back: row word + slot-row hint; a key the table does not hold is inserted by its winner ----------
  resolve_probes(win);
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    if (wpos[k] == 0xFFFFFFFFu) continue;
    unsigned r = rr[k];
    if constexpr (!NOTABLE) {
      if (__builtin_expect((r & ROW_MASK) == 0u, 0)) r = tile_insert(t, kreg[k]);   // absent (or claimed by another tile: NEW_BIT comes back)
    }
    const size_t e = (size_t)tile * TILE + wpos[k];
    sm.lrow[tslot[k]] = r;
    w.ent_b[e] = r;
    w.ent_base[e] = hn[k];
    // The winner touches its key's row NOW.  A key the batch has not met for a few steps is two dependent HBM round trips
    // away (index entry, then row), and the row copy below keeps only two pieces per wave in flight: with the rows' first
    // lines already on their way into L2 while the block files its entries and its mrow image, the copy finds them there
    // (k_ltile 54.3 -> 48.0 us at configs[1]).  Only for a tile with at most LT_TOUCH_MAXENT distinct keys: the touched
    // lines of a CU's blocks must still be in the XCD's 4 MB L2 when the copy comes (at Zipf 0.8 / 0.3, 1700 / 2030
    // distinct keys per tile, the touches were fetched twice: +16 / +21 us; profiles/r05_row_touch.txt).
    if (GATHER && nent <= (unsigned)LT_TOUCH_MAXENT && (r & ROW_MASK) != 0u && !(r & NEW_BIT))
      pf += *reinterpret_cast<const volatile float*>(row_ptr(t, r & ROW_MASK));
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
  // ---- phase 4: the output rows.  Every position takes its key's row word from the key's winner (through LDS) -----------
  if constexpr (GATHER) {
#pragma unroll
    for (int k = 0; k < IPT; ++k) rr[k] = tslot[k] != 0xFFFFFFFFu ? sm.lrow[tslot[k]] : 0u;   // skipped records read the zero row
    if (__builtin_expect(pf == 1.2345678e-38f, 0)) sm.wtot[0] = 1u;   // (the touches' only use; never true, and harmless if it were)
    output_rows();
  }
  KV_STAMP(5);
}

template <typename IdT, int VQ, bool GATHER, bool NOTABLE = false>
__global__ void __launch_bounds__(TBT, 4) k_ltile(TableDev t, WsDev w, const IdT* __restrict__ ids,
                                              const int* __restrict__ counts, long long n, int det, float* __restrict__ out) {
  ltile_body<IdT, VQ, GATHER, NOTABLE>(t, w, ids, counts, n, det, out);
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
// k_seg_combine_e: embedding_lookup_sparse's combiner over the tiles' entries
// ------------------------------------------------------------------------------------------
// out[s] = combine_j( w_j * rows[row(id_j)] ) over segment s's positions in position order (tf.segment_sum's order;
// embedding_ops.py:395-441); position j -> its entry in its tile (pos_ent, filed by k_ltile) -> the entry's row word.  A
// key this batch inserted in ANOTHER tile may not know its row yet (NEW_BIT, row part 0): it is probed — the partition
// pass in front of this kernel has published every new row.  VQ lanes (power of two >= dim / 4) per segment, SU
// positions of a segment in flight; the sums are taken in position order whatever SU is.
template <int VQ>
__global__ void __launch_bounds__(TB) k_seg_combine_e(TableDev t, const unsigned short* __restrict__ pos_ent,
                                                      const unsigned* __restrict__ ent_b, const long long* __restrict__ ent_key,
                                                      const unsigned* __restrict__ off, const float* __restrict__ wts,
                                                      long long nseg, int combiner, float* __restrict__ out) {
  const int D4 = t.dim >> 2;
  const int v = threadIdx.x % VQ;
  const bool vlive = v < D4;
  const int vv = vlive ? v : 0;
  const long long g0 = ((long long)blockIdx.x * TB + threadIdx.x) / VQ;
  const long long gstride = (long long)gridDim.x * TB / VQ;
  constexpr int SU = 4;
  for (long long sgi = g0; sgi < nseg; sgi += gstride) {
    const unsigned lo = off[sgi], hi = off[sgi + 1];
    float wsum = 0.f, w2 = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (unsigned j = lo; j < hi; j += SU) {
      unsigned e[SU], r[SU];
      float wj[SU];
      float4 x[SU];
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const bool ok = j + u < hi;
        const unsigned pe = ok ? (unsigned)pos_ent[j + u] : 0xFFFFu;
        e[u] = pe != 0xFFFFu ? ((j + u) / (unsigned)TILE) * (unsigned)TILE + pe : 0xFFFFFFFFu;
        wj[u] = ok ? (wts ? wts[j + u] : 1.f) : 0.f;
      }
#pragma unroll
      for (int u = 0; u < SU; ++u) r[u] = e[u] != 0xFFFFFFFFu ? ent_b[e[u]] : 0u;
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        if (__builtin_expect((r[u] & NEW_BIT) != 0u && (r[u] & ROW_MASK) == 0u, 0)) r[u] = table_find(t, ent_key[e[u]]);
        r[u] &= ROW_MASK;
      }
#pragma unroll
      for (int u = 0; u < SU; ++u) x[u] = reinterpret_cast<const float4*>(row_ptr(t, r[u]))[vv];
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        if (!(j + u < hi)) continue;
        acc.x += x[u].x * wj[u]; acc.y += x[u].y * wj[u]; acc.z += x[u].z * wj[u]; acc.w += x[u].w * wj[u];
        wsum += wj[u]; w2 += wj[u] * wj[u];
      }
    }
    float den = 1.f;
    if (combiner == 1) den = wsum; else if (combiner == 2) den = sqrtf(w2);
    if (combiner != 0 && (wts || hi > lo)) { acc.x /= den; acc.y /= den; acc.z /= den; acc.w /= den; }
    if (vlive) reinterpret_cast<float4*>(out + (size_t)sgi * t.dim)[v] = acc;
  }
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
// k_part2: the lookup's bookkeeping alone — FindOrInsert's frequency += occurrences, day stamp, under-threshold flag, delta
// marks, the record and the row of a key the batch inserted (kv_variable.h:320-363) — for a training lookup that no
// optimizer apply takes over: a lookup without a batch token, or a token lookup whose pending pass another op settles.
// (An apply that comes with the token completes the pass itself, inside k_papply; one that comes later runs k_papply
// PA_NONE over the tiles' entries.)  One block (256 threads) per hash partition: directory of the partition's segment in
// every tile -> its entries -> LDS hash of the distinct keys -> ONE thread per key: one hop, the row's record.
__device__ __forceinline__ void part2_body(const WsDev& w, const PartArgs& a) {
  constexpr int HSK = 1024;
  constexpr int UCAPK = HSK - TBK;
  constexpr int EB = 8;
  __shared__ long long hkey[HSK + 1];
  __shared__ unsigned hval[HSK + 1];   // summed frequency count
  __shared__ unsigned hrow[HSK + 1];   // max over the key's entries of the row word (an entry that knows the row wins)
  __shared__ unsigned short lnew[UCAPK + 8];
  __shared__ unsigned short ulist[UCAPK + 8];
  __shared__ unsigned lnu, lsent, lnnew, lkeys;
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
  if (E == 0) return;
  __shared__ unsigned stkR[24], stkr[24];
  __shared__ int sp;
  if (tid == 0) { stkR[0] = 1; stkr[0] = 0; sp = 1; lkeys = 0; }
  __syncthreads();
  if (E > 65535u) {
    if (tid == 0) raise_error(a.tv, 2u);
    return;
  }
  while (sp > 0) {
    const unsigned R = stkR[sp - 1], round = stkr[sp - 1];
    __syncthreads();
    if (tid == 0) --sp;
    for (int s = tid; s <= HSK; s += TBK) { hkey[s] = EMPTY_KEY; hval[s] = 0; hrow[s] = 0; }
    if (tid == 0) { lnu = 0; lsent = 0; lnnew = 0; }
    __syncthreads();
    // ---- pass 1: distinct keys, their counts and rows ---------------------------------------------------------
    for (unsigned x0 = 0; x0 < E; x0 += EB * TBK) {
      unsigned ge[EB];
      long long key[EB];
      unsigned ea[EB], rw[EB];
#pragma unroll
      for (int k = 0; k < EB; ++k) {
        const unsigned x = x0 + k * TBK + tid;
        ge[k] = x < E ? (unsigned)seg_entry(tpre, tstart, NT, x) : 0xFFFFFFFFu;
      }
#pragma unroll
      for (int k = 0; k < EB; ++k) {
        key[k] = 0; ea[k] = 0; rw[k] = 0;
        if (ge[k] != 0xFFFFFFFFu) { key[k] = w.ent_key[ge[k]]; ea[k] = w.ent_a[ge[k]]; rw[k] = w.ent_b[ge[k]]; }
      }
#pragma unroll
      for (int k = 0; k < EB; ++k) {
        if (ge[k] == 0xFFFFFFFFu || !in_round(key[k], R, round)) continue;
        if (lnu >= (unsigned)UCAPK) continue;
        bool first;
        const unsigned h = lds_key_slot<HSK>(hkey, &lsent, key[k], true, &first);
        if (first) {
          const unsigned u = atomicAdd(&lnu, 1u);
          if (u < (unsigned)UCAPK) ulist[u] = (unsigned short)h;
        }
        atomicAdd(&hval[h], ea[k] >> 16);
        atomicMax(&hrow[h], rw[k]);
      }
    }
    __syncthreads();
    if (lnu >= (unsigned)UCAPK) {   // more distinct keys than the hash holds: two sub-hash classes, each on its own
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
    if (tid == 0) lkeys += nu;   // (only the batch's distinct keys are counted: ctr[5])

    // ---- owner work: one thread per distinct key; ONE hop (the row's record) -------------------------------------
    constexpr int PERU = (UCAPK + TBK - 1) / TBK;
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
          if (r[k] != 0u && !isnew[k]) m[k] = load_freq_flags(a.tv, r[k]);
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
        if (r[k] == 0u) continue;   // row slab overflow: the error flag is up
        RowMeta* mp = meta_ptr(a.tv, r[k]);
        if (isnew[k]) { mp->key = key; mp->delta = 0; mp->stamp = 0; }
        mark_delta(a.tv, r[k]);
        // find_func / insert_func (kv_variable.h:320-363)
        const unsigned cnt = a.count_once ? 1u : hval[s];
        unsigned lo = (m[k].x & 0xFFFFu) + (cnt > 65535u ? 65535u : cnt);
        if (lo > 65535u) lo = 65535u;
        mp->freq = (a.day << 16) | lo;
        if (isnew[k]) mp->flags = (unsigned char)FLAG_DIRTY;
        if (m[k].y & FLAG_DIRTY) lnew[atomicAdd(&lnnew, 1u)] = (unsigned short)(s | (isnew[k] ? 0x8000u : 0u));
      }
    }
    __syncthreads();
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
          else
            for (int e = lane8; e < D; e += 8) big |= fabsf(row[e]) >= CUTOFF;
        }
        const unsigned long long mb = __ballot(big);
        const bool any = ((mb >> ((tid & 63) & ~7)) & 0xFFull) != 0;
        if (live && r != 0 && lane8 == 0) {
          unsigned char* fp = flags_ptr(a.tv, r);
          const unsigned black = isnew ? 0u : (*fp & FLAG_BLACK);
          *fp = (unsigned char)(black ? (FLAG_BLACK | FLAG_UNDER) : (any ? 0u : FLAG_UNDER));
        }
      }
    }
    __syncthreads();
    KV_STAMPP(3);
  }
  if (tid == 0) atomicAdd(&w.ctr[5], lkeys);   // distinct keys of the batch (no value returned: nothing waits for it)
}
__global__ void __launch_bounds__(TBK, 4) k_part2(WsDev w, PartArgs a) { part2_body(w, a); }

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
// the rows of this one.
constexpr int TBC = TBT;
// k_ltsum runs tsum_body in the tile pass's own block: its waves' shares and the wave-by-wave meeting in LDS assume TBC
// threads, so TBC IS TBT.  (Round 3's TILE = 1024 experiment rebuilt the tile kernel with 256 threads and left TBC at 512:
// the waves tsum_body waited on never ran, their lmeta words were never written and the sums went to epart rows nobody
// owns — past the buffer's end for the last tile.  With TBC = TBT the 1024-id tile runs: -DKV_TBT=256)
static_assert(TBC == TBT, "k_ltsum: the tile pass and the tile sums share one block");
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
template <int V, int LPR, int K>
__global__ void __launch_bounds__(TBC) k_tsum(TableDev t, WsDev w, const float* __restrict__ grad) {
  if (*reinterpret_cast<volatile unsigned*>(&t.counters[1])) return;   // the index pass gave up on this batch
  KV_STAMPT(0);
  // the tile's mrow image (TILE words = 16 bytes per thread) is requested together with its count
  static_assert(TILE * 4 == TBC * 16, "one uint4 of mrow per thread");
  extern __shared__ __attribute__((aligned(16))) char ts_smem_raw[];
  const unsigned tile = blockIdx.x;
  const uint4 img = reinterpret_cast<const uint4*>(w.mrow + (size_t)tile * TILE)[threadIdx.x];
  const unsigned mc = w.mcount[tile];
  reinterpret_cast<uint4*>(ts_smem_raw)[threadIdx.x] = img;
  __syncthreads();
  tsum_body<V, LPR, K, true>(w, grad, t.dim, tile, reinterpret_cast<const unsigned*>(ts_smem_raw), mc);
  KV_STAMPT(1);
}

// ------------------------------------------------------------------------------------------
// k_ltsum: the tile pass and the tile sums of an optimizer apply in one launch
// ------------------------------------------------------------------------------------------
// The apply has the gradient rows at hand when it runs the batch's tile pass (the lookup deferred it, or the optimizer
// meets the ids first): the block that de-duplicated a tile goes on to sum the rows of its repeated ids (tsum_body, fed
// from the mrow image still in LDS).  The dedup chain of one block — LDS phases, one probe round trip — overlaps with
// the row reads of the others; k_ltile<GATHER = false> + k_tsum one after the other were 23.5 + 24 us at configs[1].
template <typename IdT, int V, int LPR, int K>
__global__ void __launch_bounds__(TBT) k_ltsum(TableDev t, WsDev w, const IdT* __restrict__ ids, const int* __restrict__ counts,
                                              long long n, int det, const float* __restrict__ grad) {
  ltile_body<IdT, 1, false>(t, w, ids, counts, n, det, nullptr);
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
  ltile_body<long long, 1, false, true>(m.a.tv, m.w, reinterpret_cast<const long long*>(m.ids), nullptr, m.n, m.a.det, nullptr);
}
__global__ void __launch_bounds__(TBK, 4) k_part2_multi(const MultiDesc* __restrict__ descs) {
  const MultiDesc& m = descs[blockIdx.y];
  if (blockIdx.x >= m.w.P || m.n == 0) return;
  part2_body(m.w, m.a);
}
template <int V, int LPR, int K>
__global__ void __launch_bounds__(TBC) k_tsum_multi(const MultiDesc* __restrict__ descs) {
  const MultiDesc& m = descs[blockIdx.y];
  if (m.n == 0) return;
  if (*reinterpret_cast<volatile unsigned*>(&m.a.tv.counters[1])) return;
  if (blockIdx.x >= m.w.ntiles) return;
  tsum_body<V, LPR, K>(m.w, m.a.grad, m.a.tv.dim, blockIdx.x);
}
