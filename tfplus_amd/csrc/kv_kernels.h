// kv_kernels.h — the batch pipeline (included by kvhip.hip only).
//
// Why this shape.  The first version de-duplicated a batch through a global scratch hash with
// atomics.  On MI355X a returning or non-returning device-scope atomic on ONE address costs
// ~40 ns and same-address atomics serialise; a Zipf(1.2) batch of 1M ids has ~100 keys that
// occur in (nearly) every tile, so each of those addresses took ~1000 serial atomics (in-kernel
// stamps: 73 % of the dedup kernel).  fp32 atomic accumulation of gradient rows hit the same
// wall.  This pipeline has NO global atomics on the data path:
//
//   k_tile       one block (TBT threads) per TILE input positions: LDS hash dedup of the tile; the
//                tile's unique keys ("entries") are counting-sorted by the key's hash partition and
//                written with plain coalesced stores (ent_*), with the partition boundaries in
//                toff[tile][0..P].  Optimizer ops also fold the gradient rows of keys that repeat
//                inside the tile into one partial-sum row (sorted-run register sums).
//   k_part_keys  (lookup / scatter / import marks / unique) and
//   k_part_sum   (optimizers / dedup): one block per partition p: takes partition p's entries from
//                EVERY tile, so it sees all occurrences of its keys: exact counts, exclusive
//                ownership of the table rows (find / insert / frequency / flags), and for optimizer
//                ops the sum of the per-tile contributions in registers followed by the fused row
//                update.  Results that input positions need (row ids) are written back per entry.
//   k_gather     out[i] = rows[ent_b[slot_of_id[i]]], one wave per 64 output rows.
//
// Every kernel body is a __device__ function with two entry points: one table (arguments by
// value) and many tables in one launch (grid.y = table, arguments from a MultiDesc array).
//
// Summation order of repeated ids depends on LDS-atomic ranks (not run-to-run deterministic);
// there are no float atomics in global memory.
#pragma once

// ------------------------------------------------------------------------------------------
// k_tile
// ------------------------------------------------------------------------------------------
// MODE_LOOKUP : ent_a = min(sum of per-occurrence counts, 65535)     (kv_variable.h:320-350)
// others      : ent_a = one input position of the key in the tile;
//   MODE_APPLY / MODE_DEDUP additionally ent_b = gradient locator, VPL = float4 per lane per row
//   (8 lanes per row; VPL = 0 -> scalar lanes for dims that are not multiples of 4)
struct TileSmem {
  long long* lkeys;        // [LS + 1]   (slot LS: the key that equals EMPTY_KEY)
  unsigned* lcnt;          // [LS + 1]
  unsigned short* lfirst;  // [LS + 1]   (not MODE_LOOKUP) tile-local position of one occurrence; later:
                           //            sorted-row offset of the key
  unsigned short* lpos;    // [LS + 1]   entry position of the slot's key
  unsigned short* lwork;   // [TILE + 1] occupied slots
  unsigned* hist;          // [MAX_P + 1]
  unsigned* wtot;          // [8]
  unsigned short* lpart;   // [LS + 1]   partial row of the slot's key (apply / dedup); ALIASES lwork + hist,
                           //            written once those are dead
  // aliases of lkeys, valid after the entries are written:
  unsigned short* perm;    // [TILE]     tile rows grouped by key
  unsigned short* pslot;   // [TILE]     slot (key) of each sorted row
  float* red;              // [TBT / fold_lanes][dim] sums of keys whose rows span two groups' chunks
};

// gradient fold geometry: lanes per row (8 up to dim 64, 16 up to 128, 32 up to 256) so that the
// LDS rows of chunk-spanning keys, red[TBT / lanes][dim], stay <= 16 KB and always fit in the dead
// lkeys region: every dim keeps two 512-thread tile blocks per CU (dim 128 with 8 lanes: 81 KB -> one)
__host__ __device__ inline int fold_lanes(int D) { return D <= 64 ? 8 : (D <= 128 ? 16 : 32); }
__host__ __device__ inline size_t fold_red_bytes(int D) {
  return ((D & 3) == 0 && D <= 256) ? (size_t)(TBT / fold_lanes(D)) * D * 4 : 0;   // scalar path: no LDS rows
}

// lpart may reuse lwork + hist once those are dead, if they are big enough
constexpr bool LPART_ALIAS = ((TILE + 1) * 2 + 15) / 16 * 16 + (MAX_P + 1) * 4 >= (LS + 1) * 2;

__host__ __device__ inline size_t tile_smem_bytes(int mode, int D) {
  size_t b = (size_t)(LS + 1) * 8 + 16;        // lkeys
  b += (size_t)(LS + 1) * 4 + 16;              // lcnt
  b += (size_t)(LS + 1) * 2 + 16;              // lpos
  b += (size_t)(TILE + 1) * 2 + 16;            // lwork
  b += (size_t)(MAX_P + 1) * 4 + 16;           // hist
  b += 64;                                     // wtot
  if (mode != MODE_LOOKUP) b += (size_t)(LS + 1) * 2 + 16;  // lfirst
  if (mode == MODE_APPLY || mode == MODE_DEDUP) {
    if (!LPART_ALIAS) b += (size_t)(LS + 1) * 2 + 16;  // lpart
    const size_t alias = (size_t)TILE * 2 + (size_t)TILE * 2 + fold_red_bytes(D) + 64;
    const size_t lk = (size_t)(LS + 1) * 8 + 16;
    if (alias > lk) b += alias - lk;           // big dims: the fold scratch outgrows lkeys
  }
  return b;
}

template <int MODE>
__device__ __forceinline__ TileSmem carve_tile(char* base, int D) {
  TileSmem s;
  auto take = [&](size_t bytes) { char* p = base; base += (bytes + 15) & ~(size_t)15; return p; };
  char* lk = take((size_t)(LS + 1) * 8);
  s.lkeys = reinterpret_cast<long long*>(lk);
  if (MODE == MODE_APPLY || MODE == MODE_DEDUP) {
    const size_t alias = (size_t)TILE * 2 + (size_t)TILE * 2 + fold_red_bytes(D) + 64;
    const size_t lkb = ((size_t)(LS + 1) * 8 + 15) & ~(size_t)15;
    if (alias > lkb) take(alias - lkb);
    s.perm = reinterpret_cast<unsigned short*>(lk);
    s.pslot = reinterpret_cast<unsigned short*>(lk + (size_t)TILE * 2);
    s.red = reinterpret_cast<float*>(lk + (size_t)TILE * 4 + 32);
  } else {
    s.perm = nullptr; s.pslot = nullptr; s.red = nullptr;
  }
  s.lcnt = reinterpret_cast<unsigned*>(take((size_t)(LS + 1) * 4));
  s.lpos = reinterpret_cast<unsigned short*>(take((size_t)(LS + 1) * 2));
  char* lw = take((size_t)(TILE + 1) * 2);
  s.lwork = reinterpret_cast<unsigned short*>(lw);
  s.hist = reinterpret_cast<unsigned*>(take((size_t)(MAX_P + 1) * 4));
  s.wtot = reinterpret_cast<unsigned*>(take(64));
  s.lfirst = (MODE != MODE_LOOKUP) ? reinterpret_cast<unsigned short*>(take((size_t)(LS + 1) * 2)) : nullptr;
  s.lpart = nullptr;
  if (MODE == MODE_APPLY || MODE == MODE_DEDUP)
    s.lpart = LPART_ALIAS ? reinterpret_cast<unsigned short*>(lw) : reinterpret_cast<unsigned short*>(take((size_t)(LS + 1) * 2));
  return s;
}

template <int MODE, typename IdT, int VPL>
__device__ __forceinline__ void tile_body(const WsDev& w, const IdT* __restrict__ ids,
                                          const int* __restrict__ counts,
                                          const float* __restrict__ grad, long long n, int D) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  TileSmem sm = carve_tile<MODE>(smem_raw, D);
  __shared__ unsigned lnwork, lsent, lnpart, lM;

  const int tid = threadIdx.x;
  const unsigned tile = blockIdx.x;
  const long long base = (long long)tile * TILE;
  const unsigned P = w.P;
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
      if constexpr (std::is_same<IdT, IdCount>::value) {
        const long long ci = ids[i].count;             // counts travel with the ids
        creg[k] = (unsigned)(unsigned short)(ci < 65535 ? ci : 65535);
      } else if (MODE == MODE_LOOKUP && counts != nullptr) {
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
  if (tid == 0) { lnwork = 0; lsent = 0; lnpart = 0; }
  __syncthreads();

  // ---- phase 1: LDS hash insert of this tile's ids --------------------------------------
  unsigned tslot[IPT], myrank[IPT];
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    const long long i = base + (long long)k * TBT + tid;
    tslot[k] = 0xFFFFFFFFu;
    myrank[k] = 0;
    if (i < n) {
      const long long key = kreg[k];
      const unsigned c = creg[k];
      unsigned h;
      if (key == EMPTY_KEY) {
        h = LS;
        if (atomicCAS(&lsent, 0u, 1u) == 0u && MODE != MODE_LOOKUP) sm.lfirst[LS] = (unsigned short)(k * TBT + tid);
      } else {
        h = (unsigned)(mix64((unsigned long long)key) >> 40) & (LS - 1);
        for (;;) {
          const unsigned long long old =
              atomicCAS(reinterpret_cast<unsigned long long*>(&sm.lkeys[h]),
                        (unsigned long long)EMPTY_KEY, (unsigned long long)key);
          if (old == (unsigned long long)EMPTY_KEY) {
            if (MODE != MODE_LOOKUP) sm.lfirst[h] = (unsigned short)(k * TBT + tid);
            break;
          }
          if (old == (unsigned long long)key) break;
          h = (h + 1) & (LS - 1);
        }
      }
      myrank[k] = atomicAdd(&sm.lcnt[h], c);
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
      wr[q] = atomicAdd(&sm.hist[wp[q]], 1u);
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
    if (tid == 0) sm.hist[P] = nwork;
  }
  __syncthreads();
  unsigned short* toff = w.toff + (size_t)tile * (P + 1);
  for (unsigned p = tid; p <= P; p += TBT) toff[p] = (unsigned short)sm.hist[p];
  unsigned short wk[WPT];  // partial row of this thread's work items (lpart aliases lwork / hist)
#pragma unroll
  for (int q = 0; q < WPT; ++q) {
    wk[q] = 0xFFFFu;
    const unsigned wi = tid + q * TBT;
    if (wi < nwork) {
      const unsigned s = sm.lwork[wi];
      const long long key = (s == LS) ? EMPTY_KEY : sm.lkeys[s];
      const unsigned pos = sm.hist[wp[q]] + wr[q];
      const size_t e = (size_t)tile * TILE + pos;
      sm.lpos[s] = (unsigned short)pos;
      w.ent_key[e] = key;
      if (MODE == MODE_LOOKUP) {
        const unsigned c = sm.lcnt[s];
        w.ent_a[e] = c > 65535u ? 65535u : c;  // saturating add is order independent: clamp early
      } else {
        const unsigned first = (unsigned)base + sm.lfirst[s];
        w.ent_a[e] = first;
        if (MODE == MODE_APPLY || MODE == MODE_DEDUP) {
          if (sm.lcnt[s] >= 2u) {
            const unsigned k = atomicAdd(&lnpart, 1u);
            wk[q] = (unsigned short)k;
            w.ent_b[e] = PART_BIT | (tile * PARTCAP + k);
          } else {
            w.ent_b[e] = first;
          }
        }
      }
    }
  }
  __syncthreads();
  if constexpr (MODE == MODE_APPLY || MODE == MODE_DEDUP) {
    unsigned short ws[WPT];
#pragma unroll
    for (int q = 0; q < WPT; ++q) ws[q] = (tid + q * TBT) < nwork ? sm.lwork[tid + q * TBT] : (unsigned short)0xFFFF;
    __syncthreads();  // lwork / hist are dead now: lpart takes their place
#pragma unroll
    for (int q = 0; q < WPT; ++q)
      if (ws[q] != 0xFFFFu) sm.lpart[ws[q]] = wk[q];
    __syncthreads();
  }
  KV_STAMP(2);

  // ---- phase 4: every input position learns its key's entry -------------------------------
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    const long long i = base + (long long)k * TBT + tid;
    if (i < n) w.slot_of_id[i] = tile * TILE + sm.lpos[tslot[k]];
  }
  KV_STAMP(3);

  // ---- phase 5 (optimizer ops): fold the rows of keys that repeat inside the tile ---------
  if constexpr (MODE == MODE_APPLY || MODE == MODE_DEDUP) {
    // offsets of the multi-row keys in the sorted row list (lfirst is free now)
    {
      constexpr int PER = (LS + 1 + TBT - 1) / TBT;  // 9
      unsigned c[PER];
      unsigned sum = 0;
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const int s = tid * PER + q;
        c[q] = (s <= LS && sm.lcnt[s] >= 2u) ? sm.lcnt[s] : 0u;
        sum += c[q];
      }
      unsigned tot;
      unsigned run = block_excl_scan<TBT / 64>(sum, sm.wtot, &tot);
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const int s = tid * PER + q;
        if (s <= LS) { sm.lfirst[s] = (unsigned short)run; run += c[q]; }
      }
      if (tid == 0) lM = tot;
    }
    __syncthreads();  // lkeys is dead from here on: perm / pslot / red alias it
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
      if (tslot[k] != 0xFFFFFFFFu && sm.lcnt[tslot[k]] >= 2u) {
        const unsigned pos = sm.lfirst[tslot[k]] + myrank[k];
        sm.perm[pos] = (unsigned short)(k * TBT + tid);
        sm.pslot[pos] = (unsigned short)tslot[k];
      }
    }
    // VPL (host's row width class: 1, 2, 4, 8 float4 per 8 lanes) -> LPF lanes per row x VPF float4 per lane
    constexpr int LPF = VPL <= 2 ? 8 : (VPL == 4 ? 16 : 32);
    constexpr int VPF = VPL <= 2 ? (VPL > 0 ? VPL : 1) : 2;
    constexpr unsigned G = TBT / LPF;  // row groups
    if constexpr (VPL > 0)
      for (unsigned x = tid; x < G * (unsigned)D; x += TBT) sm.red[x] = 0.f;
    __syncthreads();
    KV_STAMP(4);
    const unsigned M = lM;
    const int lane8 = tid % LPF;
    const unsigned grp = tid / LPF;
    float* prow0 = w.part + (size_t)tile * PARTCAP * D;
    const unsigned C = (M + G - 1) / G;  // sorted rows per group
    if constexpr (VPL > 0) {
      // Each group folds one contiguous chunk of the sorted row list in registers, rows loaded
      // RB at a time (independent 16-byte loads).  A key whose rows lie inside the chunk is
      // stored once; a key that spans chunks (tile-hot keys) meets in the LDS row of the chunk
      // it starts in — at most one such key per chunk, so red[G][D] always suffices.
      const int NV = D >> 2;
      constexpr int RB = 16 / VPF;
      const unsigned c0 = min(M, grp * C), c1 = min(M, (grp + 1) * C);
      unsigned cur = 0xFFFFFFFFu;
      float4 acc[VPF];
      auto flush = [&]() {
        if (cur == 0xFFFFFFFFu) return;
        const unsigned st = sm.lfirst[cur], en = st + sm.lcnt[cur];
        if (st >= c0 && en <= c1) {
          float4* dst = reinterpret_cast<float4*>(prow0 + (size_t)sm.lpart[cur] * D);
#pragma unroll
          for (int v = 0; v < VPF; ++v) {
            const int q = lane8 + LPF * v;
            if (q < NV) dst[q] = acc[v];
          }
        } else {
          float* rd = sm.red + (size_t)(st / C) * D;
#pragma unroll
          for (int v = 0; v < VPF; ++v) {
            const int q = lane8 + LPF * v;
            if (q < NV) {
              atomicAdd(&rd[4 * q + 0], acc[v].x); atomicAdd(&rd[4 * q + 1], acc[v].y);
              atomicAdd(&rd[4 * q + 2], acc[v].z); atomicAdd(&rd[4 * q + 3], acc[v].w);
            }
          }
        }
      };
      for (unsigned eb = c0; eb < c1; eb += RB) {
        float4 val[RB][VPF];
        unsigned ks[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          const unsigned e = eb + r;
          ks[r] = 0xFFFFFFFFu;
#pragma unroll
          for (int v = 0; v < VPF; ++v) val[r][v] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (e < c1) {
            ks[r] = sm.pslot[e];
            const float4* g4 = reinterpret_cast<const float4*>(grad + (size_t)(base + sm.perm[e]) * D);
#pragma unroll
            for (int v = 0; v < VPF; ++v) {
              const int q = lane8 + LPF * v;
              if (q < NV) {  // read once: streaming load
                const float* gp = reinterpret_cast<const float*>(g4 + q);
                val[r][v] = make_float4(__builtin_nontemporal_load(gp), __builtin_nontemporal_load(gp + 1),
                                        __builtin_nontemporal_load(gp + 2), __builtin_nontemporal_load(gp + 3));
              }
            }
          }
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          if (ks[r] == 0xFFFFFFFFu) continue;
          if (ks[r] != cur) {
            flush();
            cur = ks[r];
#pragma unroll
            for (int v = 0; v < VPF; ++v) acc[v] = val[r][v];
          } else {
#pragma unroll
            for (int v = 0; v < VPF; ++v) {
              acc[v].x += val[r][v].x; acc[v].y += val[r][v].y;
              acc[v].z += val[r][v].z; acc[v].w += val[r][v].w;
            }
          }
        }
      }
      flush();
      __syncthreads();
      KV_STAMP(5);
      // keys that span chunks: the chunk they start in owns their LDS row
      for (unsigned x = tid; x < G * (unsigned)D; x += TBT) {
        const unsigned b = x / D, e = x % D;
        const unsigned bend = min(M, (b + 1) * C);
        if (b * C >= M || bend == 0) continue;
        const unsigned sl = sm.pslot[bend - 1];
        const unsigned st = sm.lfirst[sl], en = st + sm.lcnt[sl];
        if (en > bend && st / C == b) prow0[(size_t)sm.lpart[sl] * D + e] = sm.red[x];
      }
    } else {
      // any dim: one thread per element of one key at a time, rows in sorted order
      for (unsigned e0 = 0; e0 < M;) {
        const unsigned sl = sm.pslot[e0];
        const unsigned cnt = sm.lcnt[sl];
        for (int e = tid; e < D; e += TBT) {
          float sum = 0.f;
          for (unsigned r = 0; r < cnt; ++r) sum += grad[(size_t)(base + sm.perm[e0 + r]) * D + e];
          prow0[(size_t)sm.lpart[sl] * D + e] = sum;
        }
        e0 += cnt;
      }
    }
  }
  KV_STAMP(6);
}

// One op on one table (arguments by value) or the same op on many tables in one launch
// (blockIdx.y = table; arguments from a descriptor array in device memory, see MultiDesc below).
struct MultiDesc;
template <int MODE, typename IdT, int VPL>
__global__ void __launch_bounds__(TBT) k_tile(WsDev w, const IdT* __restrict__ ids,
                                             const int* __restrict__ counts,
                                             const float* __restrict__ grad, long long n, int D) {
  tile_body<MODE, IdT, VPL>(w, ids, counts, grad, n, D);
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
  long long* out_keys;        // MODE_DEDUP / MODE_UNIQUE
  float* out_sum;
  int* out_counts;            // MODE_UNIQUE: occurrences (saturating) of each unique key
  int count_once;             // MODE_LOOKUP: frequency += 1 per unique key instead of per occurrence
  long long direct_rows;      // MODE_DEDUP, > 0: keys ARE output row indices in [0, direct_rows)
                              // (tf.unsorted_segment_sum): out_sum[key] = sum, no key list, no counter
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

// ---- k_part_keys: MODE_LOOKUP / MODE_SCATTER / MODE_MARK ---------------------------------------
// Streams the partition's entries twice and keeps only the unique keys in LDS, so a key that
// occurs in every tile costs nothing extra.  256 threads, ~21 KB LDS, so every block of a
// 1024-partition launch is resident at once (4 per CU needs < 40 KB).
constexpr int TBK = 256;
constexpr int HSK = 1024;          // LDS hash slots
constexpr int UCAPK = HSK * 3 / 4; // unique keys per round

// per-partition segment directory in LDS: tpre[t] = entries of tiles < t, tstart[t] = first entry of
// tile t's segment.  Entry x of the partition (0 <= x < E) lives at tile t = last tpre[t] <= x.
// Filled by seg_directory(); lets every thread take entries x = tid, tid + T, ... whatever the number
// of tiles (one tile with 1000 entries is as parallel as 1000 tiles with one entry).
template <int T, int NW>
__device__ __forceinline__ unsigned seg_directory(const WsDev& w, unsigned p, unsigned short* tpre,
                                                  unsigned short* tstart, unsigned* wtot) {
  const unsigned NT = w.ntiles, P = w.P;
  const unsigned per = (NT + T - 1) / T;
  const unsigned t0 = min(NT, threadIdx.x * per), t1 = min(NT, t0 + per);
  unsigned sum = 0;
  for (unsigned t = t0; t < t1; ++t) {
    const unsigned short* to = w.toff + (size_t)t * (P + 1) + p;
    const unsigned s0 = to[0], s1 = to[1];
    tstart[t] = (unsigned short)s0;
    tpre[t] = (unsigned short)(s1 - s0);  // length for now
    sum += s1 - s0;
  }
  unsigned E;
  unsigned run = block_excl_scan<NW>(sum, wtot, &E);
  for (unsigned t = t0; t < t1; ++t) { const unsigned len = tpre[t]; tpre[t] = (unsigned short)min(run, 65535u); run += len; }
  __syncthreads();
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

template <int MODE>
__device__ __forceinline__ void part_keys_body(const WsDev& w, const PartArgs& a) {
  __shared__ long long hkey[HSK + 1];
  __shared__ unsigned hval[HSK + 1];   // lookup: summed count; scatter / mark: an input position
  __shared__ unsigned hrow[HSK + 1];   // row id of the key
  __shared__ unsigned short lnew[UCAPK + 8];  // slots whose row was inserted now / needs a row scan
  __shared__ unsigned short ulist[UCAPK + 8]; // slots of the unique keys, in order of first sight
  __shared__ unsigned lnu, lsent, lnnew;
  __shared__ unsigned wtot[8];
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  unsigned short* tpre = reinterpret_cast<unsigned short*>(smem_raw);
  unsigned short* tstart = tpre + w.ntiles;

  const int tid = threadIdx.x;
  const unsigned p = xcd_partition(blockIdx.x, w.P);
  const unsigned NT = w.ntiles;
  const int D = a.tv.dim;
  KV_STAMPP(0);
  const unsigned E = seg_directory<TBK, TBK / 64>(w, p, tpre, tstart, wtot);
  if (E == 0) return;
  const bool wide = E > 65535u;  // needs a key set crafted against the partition hash; guarded, not handled

  // work list of (R, round) sub-hash classes; an overflowing class is split in two and each
  // class is processed exactly once (block-uniform control flow)
  __shared__ unsigned stkR[24], stkr[24];
  __shared__ int sp;
  if (tid == 0) { stkR[0] = 1; stkr[0] = 0; sp = 1; }
  __syncthreads();
  if (wide) {  // never silent: the next synchronous call on the table reports it
    if (tid == 0) atomicExch(&a.tv.counters[1], 2u);
    return;
  }
  while (sp > 0) {
    const unsigned R = stkR[sp - 1], round = stkr[sp - 1];
    __syncthreads();
    if (tid == 0) --sp;
    for (int s = tid; s <= HSK; s += TBK) { hkey[s] = EMPTY_KEY; hval[s] = 0; }
    if (tid == 0) { lnu = 0; lsent = 0; lnnew = 0; }
    __syncthreads();
    // ---- pass 1: unique keys of the partition + their summed counts ---------------------------
    // entries are taken EB per thread at a time with all their global loads in flight together (a
    // partition that holds a key present in every tile has ~5x the median number of entries); the
    // slot and position of the first EB entries stay in registers for pass 2
    constexpr int EB = 8;
    unsigned cge[EB];
    unsigned short cslot[EB];
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
        if (x0 == 0) { cge[k] = ge[k]; cslot[k] = 0; }
        if (ge[k] == 0xFFFFFFFFu || !in_round(key[k], R, round)) continue;
        if (lnu >= (unsigned)UCAPK) continue;  // overflow: this class is split below
        bool first;
        const unsigned h = lds_key_slot<HSK>(hkey, &lsent, key[k], true, &first);
        if (first) {
          const unsigned u = atomicAdd(&lnu, 1u);
          if (u < (unsigned)UCAPK) ulist[u] = (unsigned short)h;
        }
        if (MODE == MODE_LOOKUP || MODE == MODE_UNIQUE) atomicAdd(&hval[h], ea[k]);
        else if (first) hval[h] = ea[k];
        if (x0 == 0) cslot[k] = (unsigned short)h;
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
          atomicExch(&a.tv.counters[1], 2u);   // keys that no sub-hash separates: reported, not applied
        }
      }
      __syncthreads();
      continue;
    }
    KV_STAMPP(1);

    // ---- owner work: one thread per unique key ------------------------------------------------
    const unsigned nu = lnu;
    if (MODE == MODE_UNIQUE) {
      // tf.unique_with_counts: dense index = block base (ONE global atomic per block) + local rank
      __shared__ unsigned lbase;
      if (tid == 0) lbase = atomicAdd(&w.ctr[0], nu);
      __syncthreads();
      for (unsigned u = tid; u < nu; u += TBK) {
        const unsigned s = ulist[u];
        const unsigned dense = lbase + u;
        hrow[s] = dense;
        a.out_keys[dense] = (s == HSK) ? EMPTY_KEY : hkey[s];
        if (a.out_counts) a.out_counts[dense] = (int)(hval[s] > 65535u ? 65535u : hval[s]);
      }
      __syncthreads();
    }
    // one thread per unique key, all keys of the partition at once: probe, then frequency word and
    // flags with a single load (RowMeta).  A thread that owns several keys (more than 256 distinct
    // keys in the partition: low-skew batches) takes them OB at a time with their probes in flight
    // together, so it pays the two dependent hops once per batch, not once per key.
    constexpr int OB = 4;
    for (unsigned u0 = tid; u0 < nu && MODE != MODE_UNIQUE; u0 += OB * TBK) {
      unsigned sl[OB], r[OB];
      long long key[OB];
      unsigned long long pp[OB];
      Entry en[OB];
      bool isnew[OB];
#pragma unroll
      for (int k = 0; k < OB; ++k) {
        const unsigned u = u0 + k * TBK;
        sl[k] = 0xFFFFFFFFu; r[k] = 0; isnew[k] = false; key[k] = 0; pp[k] = 0;
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
        r[k] = table_find_from(a.tv, key[k], pp[k], en[k]);
        if (__builtin_expect(r[k] == 0u, 0) && !(MODE == MODE_MARK && a.mark_what == 1))
          r[k] = table_find_or_insert(a.tv, key[k], &isnew[k]);
        hrow[sl[k]] = r[k];
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
        } else {
          if (isnew[k]) { mp->freq = 1u; mp->flags = 0; }  // EmbeddingValue ctor: freq_val 1, day 0 (table_manager.h:94)
          lnew[atomicAdd(&lnnew, 1u)] = (unsigned short)(s | (isnew[k] ? 0x8000u : 0u));
        }
      }
    }
    __syncthreads();
    KV_STAMPP(2);

    // ---- rows that need lanes: init of new rows, flag recompute, scatter / mark bodies ---------
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
        const unsigned r = live ? hrow[s] : 0u;
        float* row = row_ptr(a.tv, r);
        bool big = false, touch = false;
        if (live && r != 0) {
          if (isnew) big = init_row_coop(a.tv, key, row, lane8, 8);
          if (MODE == MODE_LOOKUP) {
            if (!isnew)
              for (int e = lane8; e < D; e += 8) big |= fabsf(row[e]) >= CUTOFF;
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

    // ---- pass 2 (lookup / unique): every entry learns its key's row / dense index --------------
    if (MODE == MODE_LOOKUP || MODE == MODE_UNIQUE) {
      if (cached) {
#pragma unroll
        for (int k = 0; k < EB; ++k)
          if (cge[k] != 0xFFFFFFFFu) w.ent_b[cge[k]] = hrow[cslot[k]];
      } else {
        for (unsigned x = tid; x < E; x += TBK) {
          const size_t ge = seg_entry(tpre, tstart, NT, x);
          const long long key = w.ent_key[ge];
          if (!in_round(key, R, round)) continue;
          bool first;
          w.ent_b[ge] = hrow[lds_key_slot<HSK>(hkey, &lsent, key, false, &first)];
        }
      }
    }
    __syncthreads();
    KV_STAMPP(4);
#ifdef KV_STAMPS
    if (tid == 0) { w.dbg[(size_t)(blockIdx.x + 4096) * 16 + 8] = lnu; w.dbg[(size_t)(blockIdx.x + 4096) * 16 + 9] = R; w.dbg[(size_t)(blockIdx.x + 4096) * 16 + 10] = lnnew; }
#endif
  }
}

// ---- k_part_sum: MODE_APPLY / MODE_DEDUP ---------------------------------------------------------
// Per partition: (1) copy the entries (8 B each) into LDS and hash their keys, (2) group the
// entries by key (counting sort), (3) ONE thread per unique key probes the var and slot tables
// (so every row address is known before any row is touched), (4) each LPR-lane group walks a
// contiguous chunk of the grouped contribution list: contributions are loaded RB at a time and
// summed in registers; when a key ends inside the chunk its rows are updated on the spot; a key
// that spans chunks (present in many tiles) meets in the LDS row of the chunk it starts in and
// is finished after a barrier.  Work is balanced by contributions, not by keys.
constexpr int TBS = 256;
constexpr int HSS = 1024;
constexpr int UCAPS = HSS / 2;   // 512 keys per round: keeps FTRL (one more row-id list) and 2M-id batches (4 KB directory) at 4 blocks per CU
constexpr int ECAPS = 1600;
// heavy keys per round folded by the whole block: their LDS sum rows are capped at 2 KB so that
// four blocks still fit a CU's 160 KB at every dim (D = 64 with 16 rows: 41.5 KB per block -> 3 per CU,
// a quarter of the 1024 blocks then starts late)
__host__ __device__ inline int heavy_rows(int D) {
  int r = 2048 / (4 * D);
  return r < 1 ? 1 : (r > 16 ? 16 : r);
}
constexpr int HMAXS = 16;                  // upper bound of heavy_rows() (the rest
                                           // are summed by single groups)
constexpr unsigned LOC_LDS = 0xFFFFFFE0u;  // gradient locator: row of the block's LDS hsum

__host__ __device__ inline size_t part_sum_smem_bytes(int mode, int opt, int D, int lpr, unsigned ntiles) {
  size_t b = (size_t)ntiles * 4 + 32;  // tpre, tstart
  b += (size_t)(HSS + 1) * 8 + 16 + (size_t)(HSS + 1) * 4 + 16 + (size_t)(HSS + 1) * 2 + 16;  // hkey, hval, hu
  b += (size_t)ECAPS * 4 + 16 + (size_t)ECAPS * 2 * 2 + 32;            // eb, eslot, perm
  b += (size_t)UCAPS * 2 + 16 + 64;                                    // ulist, wtot
  b += (size_t)UCAPS * 4 * 2 + 32 + (size_t)UCAPS + 16;                // utag, ur0, unew
  if (opt == OPT_FTRL) b += (size_t)UCAPS * 4 + 16;                    // ur1
  b += (size_t)heavy_rows(D) * D * 4 + 16;                             // hsum
  return b;
}

// one thread: var + slot table probes of one key, read-only probes first so the loads overlap.
// FindOrInsertUnsafe(var, filter_out != nullptr) kv_variable.h:382-408 and
// FindOrInsertUnsafe(slot, nullptr) :409-414; FTRL probes linear before accum (training_ops.cc:701-704)
template <int OPT>
__device__ __forceinline__ void probe_issue(const PartArgs& a, long long key, Entry* ev, Entry* e0, Entry* e1) {
  // hop 1: the home entries of every table, together
  const unsigned long long hh = mix64((unsigned long long)key);
  *ev = load_entry(&a.tv.entries[home_of(a.tv, key, hh)]);
  *e0 = load_entry(&a.ts0.entries[home_of(a.ts0, key, hh)]);
  *e1 = *e0;
  if (OPT == OPT_FTRL) *e1 = load_entry(&a.ts1.entries[home_of(a.ts1, key, hh)]);
}
struct ProbeMid {  // between the two halves of a probe: rows found / created, hop-2 words in flight
  unsigned rv, s0, s1, nb;
  uint2 mv;
  unsigned f0, f1;
};
template <int OPT>
__device__ __forceinline__ void probe_rows(const PartArgs& a, long long key, const Entry& ev, const Entry& e0,
                                           const Entry& e1, ProbeMid* pm) {
  const unsigned long long hh = mix64((unsigned long long)key);
  const unsigned long long pv = home_of(a.tv, key, hh), p0 = home_of(a.ts0, key, hh);
  const unsigned long long p1 = (OPT == OPT_FTRL) ? home_of(a.ts1, key, hh) : 0ull;
  unsigned rv = table_find_from(a.tv, key, pv, ev);
  unsigned s0 = table_find_from(a.ts0, key, p0, e0);
  unsigned s1 = (OPT == OPT_FTRL) ? table_find_from(a.ts1, key, p1, e1) : 0u;
  // cold block (a key that is new to one of the tables; rare once the table is warm): ALL inserts
  // happen here, so the steady-state path below is free of atomics and insert code
  bool vnew = false, new0 = false, new1 = false;
  if (__builtin_expect(rv == 0u || s0 == 0u || (OPT == OPT_FTRL && s1 == 0u), 0)) {
    if (rv == 0u) {
      rv = table_find_or_insert(a.tv, key, &vnew);
      if (rv && vnew) { RowMeta* m = meta_ptr(a.tv, rv); m->freq = 1u; m->flags = 0; }  // table_manager.h:94
    }
    // slot rows are only created for keys the update will touch (kv_variable.h:910: filtered keys return first)
    const bool filtered = rv == 0u || (!vnew && (meta_ptr(a.tv, rv)->freq & 0xFFFFu) < a.tv.enter_threshold);
    if (!filtered) {
      if (OPT == OPT_FTRL && s1 == 0u) {
        s1 = table_find_or_insert(a.ts1, key, &new1);
        if (s1 && new1) { RowMeta* m = meta_ptr(a.ts1, s1); m->freq = 1u; m->flags = 0; }
      }
      if (s0 == 0u) {
        s0 = table_find_or_insert(a.ts0, key, &new0);
        if (s0 && new0) { RowMeta* m = meta_ptr(a.ts0, s0); m->freq = 1u; m->flags = 0; }
      }
    }
  }
  // hop 2: the rows' frequency words / flags, together (row 0 always exists, so absent keys load too).
  // The var record is only needed for the frequency filter: a blacklisted row is all zeros already
  // (RemoveBlacklistUnsafe hands out a zero row, table_manager.h:359-372) and the group optimizers
  // rewrite the flags after the update, so with enter_threshold == 0 they never read it
  const bool need_vmeta = OPT == OPT_ADAGRAD || a.tv.enter_threshold != 0u;
  pm->mv = need_vmeta ? load_freq_flags(a.tv, rv) : make_uint2(0xFFFFu, 0u);
  pm->f0 = meta_ptr(a.ts0, s0)->freq;
  pm->f1 = (OPT == OPT_FTRL) ? meta_ptr(a.ts1, s1)->freq : 0u;
  pm->rv = rv; pm->s0 = s0; pm->s1 = s1;
  pm->nb = (vnew ? 1u : 0u) | (new0 ? 2u : 0u) | (new1 ? 4u : 0u);
}
template <int OPT>
__device__ __forceinline__ void probe_commit(const PartArgs& a, const ProbeMid& pm, unsigned* tag, unsigned* r0,
                                             unsigned* r1, unsigned* newbits) {
  const unsigned rv = pm.rv, s0 = pm.s0, s1 = pm.s1, f0 = pm.f0, f1 = pm.f1;
  const uint2 mv = pm.mv;
  const bool vnew = pm.nb & 1u, new0 = pm.nb & 2u, new1 = pm.nb & 4u;
  *tag = rv; *r0 = 0; *r1 = 0; *newbits = vnew ? 1u : 0u;
  if (rv == 0u) return;
  if (!vnew) {
    if ((mv.x & 0xFFFFu) < a.tv.enter_threshold) { *tag = rv | ROW_FILTERED; return; }  // kv_variable.h:910
    // RemoveBlacklistUnsafe: fresh zero row (ours already is)
    if (mv.y & FLAG_BLACK) meta_ptr(a.tv, rv)->flags = FLAG_UNDER;
  }
  // AddFrequency(1, today) on the slot rows that already existed (kv_variable.h:409-414); new ones keep freq word 1
  auto touch = [&](const TableDev& t, unsigned r, bool isnew, unsigned fold) {
    if (r == 0u || isnew) return;
    unsigned lo = (fold & 0xFFFFu) + 1u;
    if (lo > 65535u) lo = 65535u;
    *freq_ptr(t, r) = (a.day << 16) | lo;
  };
  if (OPT == OPT_FTRL) { touch(a.ts1, s1, new1, f1); *r1 = s1; }  // FTRL probes linear before accum (training_ops.cc:701-704)
  touch(a.ts0, s0, new0, f0);
  // MarkAsDeltaListElements on every table of the op, for the keys the update reaches (training_ops.cc:7196-7201)
  if (__builtin_expect(a.tv.track_delta | a.ts0.track_delta | (OPT == OPT_FTRL ? a.ts1.track_delta : 0u), 0)) {
    mark_delta(a.tv, rv);
    if (s0) mark_delta(a.ts0, s0);
    if (OPT == OPT_FTRL && s1) mark_delta(a.ts1, s1);
  }
  *r0 = s0;
  *newbits = (vnew ? 1u : 0u) | (new0 ? 2u : 0u) | (new1 ? 4u : 0u);
}
template <int OPT>
__device__ __forceinline__ void probe_for_apply(const PartArgs& a, long long key, unsigned* tag,
                                                unsigned* r0, unsigned* r1, unsigned* newbits) {
  Entry ev, e0, e1;
  ProbeMid pm;
  probe_issue<OPT>(a, key, &ev, &e0, &e1);
  probe_rows<OPT>(a, key, ev, e0, e1, &pm);
  probe_commit<OPT>(a, pm, tag, r0, r1, newbits);
}

template <int MODE, int OPT, int V, int LPR, int K>
__device__ __forceinline__ void part_sum_body(const WsDev& w, const PartArgs& a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int D = a.tv.dim;
  char* smp = smem_raw;
  auto take = [&](size_t bytes) { char* q = smp; smp += (bytes + 15) & ~(size_t)15; return q; };
  long long* hkey = reinterpret_cast<long long*>(take((size_t)(HSS + 1) * 8));
  unsigned* hval = reinterpret_cast<unsigned*>(take((size_t)(HSS + 1) * 4));  // entries, then cnt<<16 | offset
  unsigned short* hu = reinterpret_cast<unsigned short*>(take((size_t)(HSS + 1) * 2));  // slot -> unique idx
  unsigned* eb = reinterpret_cast<unsigned*>(take((size_t)ECAPS * 4));        // gradient locator
  unsigned short* eslot = reinterpret_cast<unsigned short*>(take((size_t)ECAPS * 2));
  unsigned short* perm = reinterpret_cast<unsigned short*>(take((size_t)ECAPS * 2));  // rank, then grouped entries
  unsigned short* ulist = reinterpret_cast<unsigned short*>(take((size_t)UCAPS * 2));
  unsigned* wtot = reinterpret_cast<unsigned*>(take(64));
  unsigned* utag = reinterpret_cast<unsigned*>(take((size_t)UCAPS * 4));  // dedup: dense output index
  unsigned* ur0 = reinterpret_cast<unsigned*>(take((size_t)UCAPS * 4));
  unsigned* ur1 = (OPT == OPT_FTRL) ? reinterpret_cast<unsigned*>(take((size_t)UCAPS * 4)) : ur0;
  unsigned char* unew = reinterpret_cast<unsigned char*>(take((size_t)UCAPS));
  const unsigned hmax = (unsigned)heavy_rows(D);
  float* hsum = reinterpret_cast<float*>(take((size_t)hmax * D * 4));  // sums of the heavy keys
  __shared__ unsigned lnu, lsent, lbase, lovf;

  const int tid = threadIdx.x;
  const unsigned p = xcd_partition(blockIdx.x, w.P);
  const unsigned NT = w.ntiles;
  constexpr unsigned GPB = TBS / LPR;
  const int lane = tid % LPR;
  const unsigned grp = tid / LPR;
  KV_STAMPP(0);

  unsigned short* tpre = reinterpret_cast<unsigned short*>(take((size_t)NT * 2));
  unsigned short* tstart = reinterpret_cast<unsigned short*>(take((size_t)NT * 2));
  const unsigned E = seg_directory<TBS, TBS / 64>(w, p, tpre, tstart, wtot);
  if (E == 0) return;
  if (E > 65535u) {  // a key set crafted against the partition hash; reported by the next synchronous call
    if (tid == 0) atomicExch(&a.tv.counters[1], 2u);
    return;
  }

  __shared__ unsigned stkR[24], stkr[24];
  __shared__ int sp;
  __shared__ unsigned lcls;
  if (tid == 0) { stkR[0] = 1; stkr[0] = 0; sp = 1; }
  __syncthreads();
  auto split = [&](unsigned R, unsigned round) {  // replace class (R, round) by its two halves
    __syncthreads();
    if (tid == 0) {
      if (sp + 2 <= 24) {
        stkR[sp] = 2 * R; stkr[sp] = round; ++sp;
        stkR[sp] = 2 * R; stkr[sp] = round + R; ++sp;
      } else {
        atomicExch(&a.tv.counters[1], 2u);   // keys that no sub-hash separates: reported, not applied
      }
    }
    __syncthreads();
  };
  while (sp > 0) {
    const unsigned R = stkR[sp - 1], round = stkr[sp - 1];
    __syncthreads();
    if (tid == 0) { --sp; lcls = 0; }
    __syncthreads();
    // ---- entries of this class ------------------------------------------------------------------
    unsigned Er = E;
    if (R > 1) {
      unsigned c = 0;
      for (unsigned x = tid; x < E; x += TBS) c += in_round(w.ent_key[seg_entry(tpre, tstart, NT, x)], R, round);
      if (c) atomicAdd(&lcls, c);
      __syncthreads();
      Er = lcls;
      __syncthreads();
      if (tid == 0) lcls = 0;
    }
    if (Er == 0) { __syncthreads(); continue; }
    if (Er > (unsigned)ECAPS) {  // block-uniform; nothing of this class has been applied yet
      split(R, round);
      continue;
    }
    for (int s = tid; s <= HSS; s += TBS) { hkey[s] = EMPTY_KEY; hval[s] = 0; }
    for (unsigned x = tid; x < hmax * (unsigned)D; x += TBS) hsum[x] = 0.f;
    if (tid == 0) { lnu = 0; lsent = 0; lovf = 0; }
    __syncthreads();
    // every thread takes entries x = tid, tid + TBS, ...: balanced whatever the number of tiles;
    // EB at a time so their global loads are in flight together
    constexpr int EB = (ECAPS + TBS - 1) / TBS;
    for (unsigned x0 = 0; x0 < E; x0 += EB * TBS) {
      unsigned ge[EB], lb[EB];
      long long key[EB];
#pragma unroll
      for (int k = 0; k < EB; ++k) {
        const unsigned x = x0 + k * TBS + tid;
        ge[k] = x < E ? (unsigned)seg_entry(tpre, tstart, NT, x) : 0xFFFFFFFFu;
      }
#pragma unroll
      for (int k = 0; k < EB; ++k) {
        key[k] = 0; lb[k] = 0;
        if (ge[k] != 0xFFFFFFFFu) { key[k] = w.ent_key[ge[k]]; lb[k] = w.ent_b[ge[k]]; }
      }
#pragma unroll
      for (int k = 0; k < EB; ++k) {
        if (ge[k] == 0xFFFFFFFFu || !in_round(key[k], R, round)) continue;
        if (lnu >= (unsigned)UCAPS) { lovf = 1; continue; }
        const unsigned pos = (R == 1) ? (x0 + k * TBS + tid) : atomicAdd(&lcls, 1u);  // R == 1: deterministic order
        bool first;
        const unsigned h = lds_key_slot<HSS>(hkey, &lsent, key[k], true, &first);
        if (first) {
          const unsigned u = atomicAdd(&lnu, 1u);
          if (u < (unsigned)UCAPS) { ulist[u] = (unsigned short)h; hu[h] = (unsigned short)u; }
        }
        eb[pos] = lb[k];
        eslot[pos] = (unsigned short)h;
        perm[pos] = (unsigned short)atomicAdd(&hval[h], 1u);  // rank among the key's entries
      }
    }
    __syncthreads();
    if (lovf || lnu > (unsigned)UCAPS) {  // too many distinct keys for the LDS hash: split the class
      split(R, round);
      continue;
    }
    KV_STAMPP(1);
    const unsigned nu = lnu;
    // apply: the first key of every thread has its home entries (hop 1 of the probes) requested
    // here and used after the grouping and the heavy-key fold below, which hide that round trip
    Entry pev, pe0, pe1;
    long long pkey = 0;
    constexpr bool EARLY = (MODE == MODE_APPLY) && K == 1;  // K = 2 rows have no registers to spare (occupancy)
    if (EARLY && (unsigned)tid < nu) {
      const unsigned h = ulist[tid];
      pkey = (h == HSS) ? EMPTY_KEY : hkey[h];
      probe_issue<OPT>(a, pkey, &pev, &pe0, &pe1);
    }
    // ---- group the entries by key: offsets by a scan over the unique list ----------------------
    {
      constexpr int PER = (UCAPS + TBS - 1) / TBS;
      unsigned c[PER];
      unsigned sum = 0;
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const unsigned u = tid * PER + q;
        c[q] = u < nu ? hval[ulist[u]] : 0u;
        sum += c[q];
      }
      unsigned tot;
      unsigned run = block_excl_scan<TBS / 64>(sum, wtot, &tot);
      unsigned rank[(ECAPS + TBS - 1) / TBS];
#pragma unroll
      for (int q = 0; q < (ECAPS + TBS - 1) / TBS; ++q) {
        const unsigned e = tid + q * TBS;
        rank[q] = e < Er ? perm[e] : 0u;
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const unsigned u = tid * PER + q;
        if (u < nu) { hval[ulist[u]] = (c[q] << 16) | run; run += c[q]; }
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < (ECAPS + TBS - 1) / TBS; ++q) {
        const unsigned e = tid + q * TBS;
        if (e < Er) perm[(hval[eslot[e]] & 0xFFFFu) + rank[q]] = (unsigned short)e;
      }
    }
    // ---- one thread per unique key: table probes (apply) / output slots (dedup) ----------------
    if (MODE == MODE_DEDUP) {
      if (a.direct_rows > 0) {
        for (unsigned u = tid; u < nu; u += TBS) {
          const unsigned h = ulist[u];
          const long long key = (h == HSS) ? EMPTY_KEY : hkey[h];
          utag[u] = (key >= 0 && key < a.direct_rows) ? (unsigned)key : 0xFFFFFFFFu;  // out of range: dropped
        }
      } else {
        if (tid == 0) lbase = atomicAdd(&w.ctr[0], nu);  // one atomic per partition block and round
        __syncthreads();
        for (unsigned u = tid; u < nu; u += TBS) {
          utag[u] = lbase + u;
          const unsigned h = ulist[u];
          a.out_keys[lbase + u] = (h == HSS) ? EMPTY_KEY : hkey[h];
        }
      }
    }
    auto finish_probes = [&]() {
      if (MODE != MODE_APPLY) return;
      for (unsigned u = tid; u < nu; u += TBS) {
        unsigned tag, r0, r1, nb;
        if (EARLY && u == (unsigned)tid) {
          ProbeMid pm;  // (starting hop 2 before the fold as well measured no further gain)
          probe_rows<OPT>(a, pkey, pev, pe0, pe1, &pm);
          probe_commit<OPT>(a, pm, &tag, &r0, &r1, &nb);
        } else {
          const unsigned h = ulist[u];
          probe_for_apply<OPT>(a, (h == HSS) ? EMPTY_KEY : hkey[h], &tag, &r0, &r1, &nb);
        }
        utag[u] = tag; ur0[u] = r0; unew[u] = (unsigned char)nb;
        if (OPT == OPT_FTRL) ur1[u] = r1;
      }
    };
    __syncthreads();
    KV_STAMPP(2);

    // finish one key whose summed gradient is in gv: optimizer update, or emit (dedup)
    auto finish = [&](unsigned h, bool live, float (&gv)[K][V]) {
      const unsigned u = live ? hu[h] : 0u;
      if (MODE == MODE_APPLY) {
        const long long key = (h == HSS) ? EMPTY_KEY : hkey[h];
        const unsigned tag = live ? utag[u] : 0u;
        const unsigned nb = live ? unew[u] : 0u;
        bool big = false;
        const bool doinit = live && (nb & 1u) && (tag & ROW_MASK);
        if (doinit) big = init_row_coop(a.tv, key, row_ptr(a.tv, tag & ROW_MASK), lane, LPR);
        const bool any = group_any<LPR>(big);
        if (doinit && lane == 0) *flags_ptr(a.tv, tag & ROW_MASK) = any ? 0 : FLAG_UNDER;
        opt_update_row<OPT, V, LPR, K>(a.tv, a.ts0, a.ts1, key, tag, live ? ur0[u] : 0u, (nb & 2u) != 0,
                                       live ? ur1[u] : 0u, (nb & 4u) != 0, live, gv, a.opt, lane);
      } else if (live && utag[u] != 0xFFFFFFFFu) {
        const unsigned dense = utag[u];
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const int e0 = (lane + k * LPR) * V;
          if (e0 < D) stv<V>(a.out_sum + (size_t)dense * D + e0, gv[k]);
        }
      }
    };

    // ---- (a) keys present in many tiles (> HEAVY entries): their contributions, concatenated,
    //      are folded by ALL groups (chunks with run detection); partial sums meet in one LDS row
    //      per heavy key (ds_add_f32), one barrier for all of them.  Afterwards such a key looks
    //      like a key with a single contribution (LOC_LDS) and phase (b) finishes it.
    {
      __shared__ unsigned short hk[HMAXS];      // slot of heavy key j
      __shared__ unsigned hpre[HMAXS + 1];      // prefix of their entry counts
      __shared__ unsigned lnh;
      if (tid == 0) lnh = 0;
      __syncthreads();
      for (unsigned u = tid; u < nu; u += TBS) {
        const unsigned h = ulist[u];
        if ((hval[h] >> 16) > (unsigned)HEAVY) {
          const unsigned j = atomicAdd(&lnh, 1u);
          if (j < hmax) hk[j] = (unsigned short)h;
        }
      }
      __syncthreads();
      if (tid == 0) {
        const unsigned n = min(lnh, hmax);
        unsigned run = 0;
        for (unsigned j = 0; j < n; ++j) { hpre[j] = run; run += hval[hk[j]] >> 16; }
        hpre[n] = run;
        lnh = n;
      }
      __syncthreads();
      const unsigned nh = lnh;
      if (nh > 0) {  // block-uniform
        const unsigned Hn = hpre[nh];
        const unsigned C = (Hn + GPB - 1) / GPB;
        const unsigned c0 = min(Hn, grp * C), c1 = min(Hn, (grp + 1) * C);
        constexpr int RB = (8 / K) > 0 ? (8 / K) : 1;
        unsigned j = 0;
        while (j + 1 < nh && hpre[j + 1] <= c0) ++j;
        float gv[K][V];
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
          for (int c = 0; c < V; ++c) gv[k][c] = 0.f;
        auto flush = [&](unsigned jj) {
          float* rd = hsum + (size_t)jj * D;
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const int e0 = (lane + k * LPR) * V;
            if (e0 < D) {
#pragma unroll
              for (int c = 0; c < V; ++c) { atomicAdd(&rd[e0 + c], gv[k][c]); gv[k][c] = 0.f; }
            }
          }
        };
        for (unsigned vb = c0; vb < c1; vb += RB) {
          float val[RB][K][V];
          unsigned kj[RB];
#pragma unroll
          for (int r = 0; r < RB; ++r) {
            kj[r] = 0xFFFFFFFFu;
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
              for (int c = 0; c < V; ++c) val[r][k][c] = 0.f;
            const unsigned vi = vb + r;
            if (vi < c1) {
              unsigned jj = j;
              while (hpre[jj + 1] <= vi) ++jj;
              kj[r] = jj;
              const unsigned loc = eb[perm[(hval[hk[jj]] & 0xFFFFu) + (vi - hpre[jj])]];
              const float* src = (loc & PART_BIT) ? w.part + (size_t)(loc & ~PART_BIT) * D
                                                  : a.grad + (size_t)loc * D;
#pragma unroll
              for (int k = 0; k < K; ++k) {
                const int e0 = (lane + k * LPR) * V;
                if (e0 < D) ldv_stream<V>(src + e0, val[r][k]);
              }
            }
          }
#pragma unroll
          for (int r = 0; r < RB; ++r) {
            if (kj[r] == 0xFFFFFFFFu) continue;
            if (kj[r] != j) { flush(j); j = kj[r]; }
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
              for (int c = 0; c < V; ++c) gv[k][c] += val[r][k][c];
          }
        }
        if (c0 < c1) flush(j);
        __syncthreads();
        for (unsigned jj = tid; jj < nh; jj += TBS) {
          const unsigned h = hk[jj];
          const unsigned o = hval[h] & 0xFFFFu;
          eb[perm[o]] = LOC_LDS + jj;
          hval[h] = (1u << 16) | o;
        }
        __syncthreads();
      }
    }
    finish_probes();
    __syncthreads();
    KV_STAMPP(3);
    // ---- (b) one group per key, keys in converged rounds: contributions and state rows are
    //      loaded together (all addresses known), then the fused update -------------------------
    {
      const unsigned upad = (nu + GPB - 1) / GPB * GPB;
      for (unsigned u = grp; u < upad; u += GPB) {
        const bool live = u < nu;
        const unsigned h = live ? ulist[u] : 0u;
        const unsigned cn = live ? (hval[h] >> 16) : 0u;
        const unsigned o = hval[h] & 0xFFFFu;
        float gv[K][V];
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
          for (int c = 0; c < V; ++c) gv[k][c] = 0.f;
        // touch this lane's part of the state rows now, so the update's real loads (issued after
        // the contribution sum) hit cache instead of paying a second HBM round trip
        float touch = 0.f;
        if (MODE == MODE_APPLY && live) {
          const unsigned tg = utag[hu[h]];
          if (!(tg & ROW_FILTERED) && (tg & ROW_MASK)) {
            const int e0 = lane * V;
            if (e0 < D) {
              const float* xr = row_ptr(a.tv, tg & ROW_MASK);
              const float* sr = row_ptr(a.ts0, ur0[hu[h]]);
              touch = xr[e0] + sr[e0];
              if (OPT == OPT_ADAM_V4 || OPT == OPT_ADAM_V3) touch += sr[e0 + D] + sr[e0 + 2 * D];
              if (OPT == OPT_FTRL) touch += row_ptr(a.ts1, ur1[hu[h]])[e0];
            }
          }
        }
        constexpr int RB = (8 / K) > 0 ? (8 / K) : 1;
        for (unsigned jb = 0; jb < cn; jb += RB) {
          float val[RB][K][V];
#pragma unroll
          for (int r = 0; r < RB; ++r) {
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
              for (int c = 0; c < V; ++c) val[r][k][c] = 0.f;
            if (jb + r < cn) {
              const unsigned loc = eb[perm[o + jb + r]];
              if (loc >= LOC_LDS) {
                const float* src = hsum + (size_t)(loc - LOC_LDS) * D;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                  const int e0 = (lane + k * LPR) * V;
                  if (e0 < D) ldv<V>(src + e0, val[r][k]);
                }
              } else {
                const float* src = (loc & PART_BIT) ? w.part + (size_t)(loc & ~PART_BIT) * D
                                                    : a.grad + (size_t)loc * D;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                  const int e0 = (lane + k * LPR) * V;
                  if (e0 < D) ldv_stream<V>(src + e0, val[r][k]);
                }
              }
            }
          }
#pragma unroll
          for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
              for (int c = 0; c < V; ++c) gv[k][c] += val[r][k][c];
        }
        asm volatile("" ::"v"(touch));  // keep the touch loads
#ifdef KV_STAMPS
        if (tid == 0 && u / GPB < 4) w.dbg[(size_t)(blockIdx.x + 4096) * 16 + 11 + (u / GPB)] = wall_clock64();
#endif
        finish(h, live, gv);
      }
    }
    if (MODE == MODE_DEDUP) {
      __syncthreads();
      // every entry learns its key's dense index: a second pass over the entries with a read-only
      // LDS hash lookup (an entry-position list in LDS would cost 6.4 KB and the fourth block per CU)
      for (unsigned x = tid; x < E; x += TBS) {
        const size_t ge = seg_entry(tpre, tstart, NT, x);
        const long long key = w.ent_key[ge];
        if (!in_round(key, R, round)) continue;
        bool first;
        w.ent_b[ge] = utag[hu[lds_key_slot<HSS>(hkey, &lsent, key, false, &first)]];
      }
    }
    __syncthreads();
    KV_STAMPP(4);
#ifdef KV_STAMPS
    if (tid == 0) { w.dbg[(size_t)(blockIdx.x + 4096) * 16 + 8] = Er; w.dbg[(size_t)(blockIdx.x + 4096) * 16 + 9] = R; w.dbg[(size_t)(blockIdx.x + 4096) * 16 + 10] = nu; }
#endif
  }
}

// ------------------------------------------------------------------------------------------
// k_gather: out[i, :] = rows[ent_b[slot_of_id[i]]]
// ------------------------------------------------------------------------------------------
// VQ = float4 vectors per row (dim / 4) when > 0 (power of two); VQ = 0 -> generic dim
template <int VQ>
__device__ __forceinline__ void gather_body(const TableDev& t, const WsDev& w, float* __restrict__ out,
                                            long long n) {
  if constexpr (VQ > 0 && VQ <= 64) {
    // One wave takes 64 consecutive output rows per step.  Lane l resolves row l's table row id
    // (slot_of_id -> ent_b: two dependent loads, 64 rows in flight per wave and no redundancy);
    // then the wave copies the rows VQ lanes per row, 64 / VQ rows per instruction, CH
    // instructions in flight, the row ids passed between lanes with ds_bpermute.
    constexpr int RW = 64 / VQ;            // rows per copy instruction
    constexpr int CH = VQ < 16 ? VQ : 16;  // copy instructions in flight
    const int lane = threadIdx.x & 63;
    const int v = lane % VQ, sub = lane / VQ;
    const long long wave = (long long)blockIdx.x * (TB / 64) + (threadIdx.x >> 6);
    const long long nwaves = (long long)gridDim.x * (TB / 64);
    // software pipeline over the wave's steps: while step i copies rows, step i+1's row ids and
    // step i+2's slots are already in flight (the three dependent hops overlap across steps)
    const long long stride = nwaves * 64;
    long long r0 = wave * 64;
    unsigned sl1 = 0, sl2 = 0, rr = 0;
    if (r0 + lane < n) rr = w.ent_b[__builtin_nontemporal_load(&w.slot_of_id[r0 + lane])];
    if (r0 + stride + lane < n) sl1 = __builtin_nontemporal_load(&w.slot_of_id[r0 + stride + lane]);
    for (; r0 < n; r0 += stride) {
      unsigned rr1 = 0;
      if (r0 + stride + lane < n) rr1 = w.ent_b[sl1];
      if (r0 + 2 * stride + lane < n) sl2 = __builtin_nontemporal_load(&w.slot_of_id[r0 + 2 * stride + lane]);
#pragma unroll
      for (int j0 = 0; j0 < VQ; j0 += CH) {
        float4 val[CH];
        unsigned rj[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) rj[j] = __shfl(rr, (j0 + j) * RW + sub);
#pragma unroll
        for (int j = 0; j < CH; ++j) val[j] = reinterpret_cast<const float4*>(row_ptr(t, rj[j]))[v];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
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
    for (long long i = (long long)blockIdx.x * RPB + threadIdx.x / VQ; i < n; i += (long long)gridDim.x * RPB) {
      const unsigned r = w.ent_b[w.slot_of_id[i]];
      reinterpret_cast<float4*>(out + (size_t)i * (VQ * 4))[v] = reinterpret_cast<const float4*>(row_ptr(t, r))[v];
    }
  } else {
    const int D = t.dim;
    const long long total = n * D;
    for (long long x = (long long)blockIdx.x * TB + threadIdx.x; x < total; x += (long long)gridDim.x * TB) {
      const long long i = x / D;
      const int e = (int)(x - i * D);
      out[x] = row_ptr(t, w.ent_b[w.slot_of_id[i]])[e];
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
__global__ void __launch_bounds__(TBK) k_part_keys(WsDev w, PartArgs a) { part_keys_body<MODE>(w, a); }
template <int MODE, int OPT, int V, int LPR, int K>
__global__ void __launch_bounds__(TBS, (K == 1 ? 4 : 1)) k_part_sum(WsDev w, PartArgs a) { part_sum_body<MODE, OPT, V, LPR, K>(w, a); }
template <int VQ>
__global__ void __launch_bounds__(TB) k_gather(TableDev t, WsDev w, float* __restrict__ out, long long n) {
  gather_body<VQ>(t, w, out, n);
}

template <int MODE, typename IdT, int VPL>
__global__ void __launch_bounds__(TBT) k_tile_multi(const MultiDesc* __restrict__ descs) {
  const MultiDesc& m = descs[blockIdx.y];
  if (blockIdx.x >= m.w.ntiles) return;
  tile_body<MODE, IdT, VPL>(m.w, reinterpret_cast<const IdT*>(m.ids), m.counts, m.a.grad, m.n, m.a.tv.dim);
}
template <int MODE>
__global__ void __launch_bounds__(TBK) k_part_keys_multi(const MultiDesc* __restrict__ descs) {
  const MultiDesc& m = descs[blockIdx.y];
  if (blockIdx.x >= m.w.P || m.n == 0) return;
  part_keys_body<MODE>(m.w, m.a);
}
template <int MODE, int OPT, int V, int LPR, int K>
__global__ void __launch_bounds__(TBS, (K == 1 ? 4 : 1)) k_part_sum_multi(const MultiDesc* __restrict__ descs) {
  const MultiDesc& m = descs[blockIdx.y];
  if (blockIdx.x >= m.w.P || m.n == 0) return;
  part_sum_body<MODE, OPT, V, LPR, K>(m.w, m.a);
}
template <int VQ>
__global__ void __launch_bounds__(TB) k_gather_multi(const MultiDesc* __restrict__ descs) {
  const MultiDesc& m = descs[blockIdx.y];
  gather_body<VQ>(m.a.tv, m.w, m.out, m.n);
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
template <int VQ>
__device__ __forceinline__ void goz_wave(const TableDev& t, const void* __restrict__ ids, bool ids_int32,
                                         float* __restrict__ out, long long n) {
  constexpr int RW = 64 / VQ;            // rows per copy instruction
  constexpr int CH = VQ < 4 ? VQ : 4;    // copy instructions in flight
  const int lane = threadIdx.x & 63;
  const int v = lane % VQ, sub = lane / VQ;
  const long long wave = (long long)blockIdx.x * (TB / 64) + (threadIdx.x >> 6);
  const long long stride = (long long)gridDim.x * (TB / 64) * 64;
  for (long long r0 = wave * 64; r0 < n; r0 += stride) {
    unsigned rr = 0;  // row 0 reads zeros: misses, and lanes past the end
    if (r0 + lane < n)
      rr = table_find(t, ids_int32 ? (long long)reinterpret_cast<const int*>(ids)[r0 + lane]
                                   : reinterpret_cast<const long long*>(ids)[r0 + lane]);
#pragma unroll
    for (int j0 = 0; j0 < VQ; j0 += CH) {
      float4 val[CH];
      unsigned rj[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) rj[j] = __shfl(rr, (j0 + j) * RW + sub);
#pragma unroll
      for (int j = 0; j < CH; ++j) val[j] = reinterpret_cast<const float4*>(row_ptr(t, rj[j]))[v];
#pragma unroll
      for (int j = 0; j < CH; ++j) {
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
  goz_wave<VQ>(t, ids, sizeof(IdT) == 4, out, n);
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
  const TableDev t = d.t;
  const int D = t.dim;
  if ((D & 3) == 0) {  // block-uniform: the wave-shaped body for dims 4, 8, ..., 256
    switch (D >> 2) {
      case 1: goz_wave<1>(t, d.ids, d.ids_int32, d.out, d.n); return;
      case 2: goz_wave<2>(t, d.ids, d.ids_int32, d.out, d.n); return;
      case 4: goz_wave<4>(t, d.ids, d.ids_int32, d.out, d.n); return;
      case 8: goz_wave<8>(t, d.ids, d.ids_int32, d.out, d.n); return;
      case 16: goz_wave<16>(t, d.ids, d.ids_int32, d.out, d.n); return;
      case 32: goz_wave<32>(t, d.ids, d.ids_int32, d.out, d.n); return;
      case 64: goz_wave<64>(t, d.ids, d.ids_int32, d.out, d.n); return;
      default: break;
    }
  }
  const int lane8 = threadIdx.x & 7;
  for (long long i = (long long)blockIdx.x * (TB / 8) + (threadIdx.x >> 3); i < d.n;
       i += (long long)gridDim.x * (TB / 8)) {
    const long long key = d.ids_int32 ? (long long)reinterpret_cast<const int*>(d.ids)[i]
                                      : reinterpret_cast<const long long*>(d.ids)[i];
    const unsigned r = table_find(t, key);
    const float* row = row_ptr(t, r);
    float* o = d.out + (size_t)i * D;
    if ((D & 3) == 0) {
      for (int q = lane8; q < (D >> 2); q += 8)
        reinterpret_cast<float4*>(o)[q] = reinterpret_cast<const float4*>(row)[q];
    } else {
      for (int e = lane8; e < D; e += 8) o[e] = row[e];
    }
  }
}

__global__ void k_store_count(const unsigned* ctr, long long* out) { *out = (long long)*ctr; }

// kv_dedup_segment_sum: inverse[i] = dense unique index of input position i
__global__ void k_dedup_inverse(WsDev w, long long n, int* inverse) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    inverse[i] = (int)w.ent_b[w.slot_of_id[i]];
}

// ------------------------------------------------------------------------------------------
// multi-GPU routing: stable-by-tile counting sort of ids by owner rank = floor_mod(id, world)
// (kernels/utility.h:90-107).  world <= 64.  hist is [world][ntiles] (owner-major for the scan).
// ------------------------------------------------------------------------------------------
constexpr int RT = 1024;  // ids per routing tile (256 threads x 4)
constexpr int MAXW = 64;

__device__ __forceinline__ unsigned owner_rank(long long id, int world) {
  long long m = id % world;
  return (unsigned)(m < 0 ? m + world : m);
}

template <typename IdT>
__global__ void __launch_bounds__(TB) k_owner_hist(const IdT* __restrict__ ids, long long n, int world,
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
    if (i < n) atomicAdd(&h[owner_rank(load_id(ids, (size_t)i), world)], 1u);
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
__global__ void __launch_bounds__(TB) k_owner_scatter(const IdT* __restrict__ ids, long long n, int world,
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
      const unsigned pos = atomicAdd(&h[owner_rank(id, world)], 1u);
      out_ids[pos] = id;
      perm[pos] = (int)i;
      // optional extras of the sharded lookup: the exchange payload (id, occurrence count) in
      // bucket order, and where input position i went (the inverse of perm)
      if (pairs_out) { pairs_out[2 * (size_t)pos] = id; pairs_out[2 * (size_t)pos + 1] = counts_in ? (long long)counts_in[i] : 1ll; }
      if (pos_out) pos_out[i] = (int)pos;
    }
  }
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
          sl[u] = ok ? w.slot_of_id[j + u] : 0xFFFFFFFFu;
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
          const unsigned r = w.ent_b[w.slot_of_id[j]];
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

