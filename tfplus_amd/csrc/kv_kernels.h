// kv_kernels.h — the batch pipeline (included by kvhip.hip only).
//
// Why this shape.  The first version de-duplicated a batch through a global scratch hash with
// atomics.  On MI355X a returning or non-returning device-scope atomic on ONE address costs
// ~40 ns and same-address atomics serialise; a Zipf(1.2) batch of 1M ids has ~100 keys that
// occur in (nearly) every 1024-id tile, so each of those addresses took ~1000 serial atomics
// (in-kernel stamps: 73 % of the dedup kernel).  fp32 atomic accumulation of gradient rows hit
// the same wall.  This pipeline has NO global atomics on the data path:
//
//   k_tile  one block per 1024 input positions: LDS hash dedup of the tile; the tile's unique
//           keys ("entries") are counting-sorted by the key's hash partition and written with
//           plain coalesced stores (ent_*), with the partition boundaries in toff[tile][0..P].
//           Optimizer ops also fold the gradient rows of keys that repeat inside the tile into
//           one partial-sum row (registers; whole block for tile-hot keys).
//   k_part  one block per partition p: gathers partition p's entries from EVERY tile, so it sees
//           all occurrences of its keys: exact counts, exclusive ownership of the table rows
//           (find / insert / frequency / flags), and for optimizer ops the sum of the per-tile
//           contributions in registers followed by the fused row update.  Results that input
//           positions need (row ids) are written back per entry.
//   k_gather  out[i] = rows[ent_b[slot_of_id[i]]], 16 bytes per lane.
//
// Summation order of repeated ids is deterministic (tile order inside a partition, sorted-run
// order inside a tile up to the LDS-atomic rank; no float atomics in global memory).
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
  unsigned* lfirst;        // [LS + 1]   (not MODE_LOOKUP) later: sorted-row offset of the key
  unsigned short* lpos;    // [LS + 1]   entry position of the slot's key
  unsigned short* lwork;   // [TILE + 1] occupied slots
  unsigned* hist;          // [MAX_P + 1]
  unsigned* wtot;          // [8]
  unsigned short* lpart;   // [LS + 1]   partial row of the slot's key (apply / dedup)
  // aliases of lkeys, valid after the entries are written:
  unsigned short* perm;    // [TILE]     tile rows grouped by key
  unsigned short* mlist;   // [PARTCAP]  partial row -> slot
  float* red;              // [TB / 8][dim] block fold scratch
};

__host__ __device__ inline size_t tile_smem_bytes(int mode, int D) {
  size_t b = (size_t)(LS + 1) * 8 + 16;        // lkeys
  b += (size_t)(LS + 1) * 4 + 16;              // lcnt
  b += (size_t)(LS + 1) * 2 + 16;              // lpos
  b += (size_t)(TILE + 1) * 2 + 16;            // lwork
  b += (size_t)(MAX_P + 1) * 4 + 16;           // hist
  b += 64;                                     // wtot
  if (mode != MODE_LOOKUP) b += (size_t)(LS + 1) * 4 + 16;  // lfirst
  if (mode == MODE_APPLY || mode == MODE_DEDUP) {
    b += (size_t)(LS + 1) * 2 + 16;            // lpart
    const size_t alias = (size_t)TILE * 2 + (size_t)PARTCAP * 2 + (size_t)(TB / 8) * D * 4 + 64;
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
    const size_t alias = (size_t)TILE * 2 + (size_t)PARTCAP * 2 + (size_t)(TB / 8) * D * 4 + 64;
    const size_t lkb = ((size_t)(LS + 1) * 8 + 15) & ~(size_t)15;
    if (alias > lkb) take(alias - lkb);
    s.perm = reinterpret_cast<unsigned short*>(lk);
    s.mlist = reinterpret_cast<unsigned short*>(lk + (size_t)TILE * 2);
    s.red = reinterpret_cast<float*>(lk + (size_t)TILE * 2 + (size_t)PARTCAP * 2 + 32);
  } else {
    s.perm = nullptr; s.mlist = nullptr; s.red = nullptr;
  }
  s.lcnt = reinterpret_cast<unsigned*>(take((size_t)(LS + 1) * 4));
  s.lpos = reinterpret_cast<unsigned short*>(take((size_t)(LS + 1) * 2));
  s.lwork = reinterpret_cast<unsigned short*>(take((size_t)(TILE + 1) * 2));
  s.hist = reinterpret_cast<unsigned*>(take((size_t)(MAX_P + 1) * 4));
  s.wtot = reinterpret_cast<unsigned*>(take(64));
  s.lfirst = (MODE != MODE_LOOKUP) ? reinterpret_cast<unsigned*>(take((size_t)(LS + 1) * 4)) : nullptr;
  s.lpart = (MODE == MODE_APPLY || MODE == MODE_DEDUP)
                ? reinterpret_cast<unsigned short*>(take((size_t)(LS + 1) * 2)) : nullptr;
  return s;
}

// fold rows [e0, e1) of the tile's sorted row list into acc (8 lanes per row, lane8 = 0..7)
template <int VPL>
__device__ __forceinline__ void fold_rows(const float* __restrict__ grad, long long base, int D,
                                          const unsigned short* perm, unsigned e0, unsigned e1,
                                          int lane8, float4 (&acc)[VPL > 0 ? VPL : 1]) {
  constexpr int RB = VPL > 0 ? (8 / VPL > 0 ? 8 / VPL : 1) : 1;  // rows loaded together
  const int NV = D >> 2;
  for (unsigned eb = e0; eb < e1; eb += RB) {
    float4 val[RB][VPL > 0 ? VPL : 1];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const unsigned e = eb + r;
#pragma unroll
      for (int v = 0; v < VPL; ++v) val[r][v] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < e1) {
        const float4* g4 = reinterpret_cast<const float4*>(grad + (size_t)(base + perm[e]) * D);
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
          const int q = lane8 + 8 * v;
          if (q < NV) val[r][v] = g4[q];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        acc[v].x += val[r][v].x; acc[v].y += val[r][v].y;
        acc[v].z += val[r][v].z; acc[v].w += val[r][v].w;
      }
    }
  }
}

template <int MODE, typename IdT, int VPL>
__global__ void __launch_bounds__(TB) k_tile(WsDev w, const IdT* __restrict__ ids,
                                             const int* __restrict__ counts,
                                             const float* __restrict__ grad, long long n, int D) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  TileSmem sm = carve_tile<MODE>(smem_raw, D);
  __shared__ unsigned lnwork, lsent, lnpart;

  const int tid = threadIdx.x;
  const unsigned tile = blockIdx.x;
  const long long base = (long long)tile * TILE;
  const unsigned P = w.P;
  KV_STAMP(0);

  for (int s = tid; s <= LS; s += TB) {
    sm.lkeys[s] = EMPTY_KEY;
    sm.lcnt[s] = 0;
  }
  for (unsigned p = tid; p <= P; p += TB) sm.hist[p] = 0;
  if (tid == 0) { lnwork = 0; lsent = 0; lnpart = 0; }
  __syncthreads();

  // ---- phase 1: LDS hash insert of this tile's ids --------------------------------------
  unsigned tslot[IPT], myrank[IPT];
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    const long long i = base + (long long)k * TB + tid;
    tslot[k] = 0xFFFFFFFFu;
    myrank[k] = 0;
    if (i < n) {
      const long long key = load_id(ids, (size_t)i);
      unsigned c = 1;
      if (MODE == MODE_LOOKUP && counts != nullptr) {
        // SaturateMaxFrequency(int32) -> uint16 (utility.h:57-59)
        const int ci = counts[i];
        c = (unsigned)(unsigned short)(ci < 65535 ? ci : 65535);
      }
      unsigned h;
      if (key == EMPTY_KEY) {
        h = LS;
        if (atomicCAS(&lsent, 0u, 1u) == 0u && MODE != MODE_LOOKUP) sm.lfirst[LS] = (unsigned)i;
      } else {
        h = (unsigned)(mix64((unsigned long long)key) >> 40) & (LS - 1);
        for (;;) {
          const unsigned long long old =
              atomicCAS(reinterpret_cast<unsigned long long*>(&sm.lkeys[h]),
                        (unsigned long long)EMPTY_KEY, (unsigned long long)key);
          if (old == (unsigned long long)EMPTY_KEY) {
            if (MODE != MODE_LOOKUP) sm.lfirst[h] = (unsigned)i;
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
  for (int s = tid; s < LS; s += TB)
    if (sm.lkeys[s] != EMPTY_KEY) sm.lwork[atomicAdd(&lnwork, 1u)] = (unsigned short)s;
  if (tid == 0 && lsent) sm.lwork[atomicAdd(&lnwork, 1u)] = (unsigned short)LS;
  __syncthreads();
  const unsigned nwork = lnwork;

  // ---- phase 3: counting sort of the tile's unique keys by owning partition ---------------
  constexpr int WPT = (TILE + 1 + TB - 1) / TB;  // work items per thread (5)
  unsigned wp[WPT], wr[WPT];
#pragma unroll
  for (int q = 0; q < WPT; ++q) {
    const unsigned wi = tid + q * TB;
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
    const unsigned per = (P + TB - 1) / TB;
    const unsigned p0 = tid * per, p1 = min(p0 + per, P);
    unsigned sum = 0;
    for (unsigned p = p0; p < p1; ++p) sum += sm.hist[p];
    unsigned tot;
    unsigned run = block_excl_scan<TB / 64>(sum, sm.wtot, &tot);
    for (unsigned p = p0; p < p1; ++p) { const unsigned c = sm.hist[p]; sm.hist[p] = run; run += c; }
    if (tid == 0) sm.hist[P] = nwork;
  }
  __syncthreads();
  unsigned short* toff = w.toff + (size_t)tile * (P + 1);
  for (unsigned p = tid; p <= P; p += TB) toff[p] = (unsigned short)sm.hist[p];
#pragma unroll
  for (int q = 0; q < WPT; ++q) {
    const unsigned wi = tid + q * TB;
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
        w.ent_a[e] = sm.lfirst[s];
        if (MODE == MODE_APPLY || MODE == MODE_DEDUP) {
          if (sm.lcnt[s] >= 2u) {
            const unsigned k = atomicAdd(&lnpart, 1u);
            sm.lpart[s] = (unsigned short)k;
            w.ent_b[e] = PART_BIT | (tile * PARTCAP + k);
          } else {
            sm.lpart[s] = 0xFFFFu;
            w.ent_b[e] = sm.lfirst[s];
          }
        }
      }
    }
  }
  __syncthreads();
  KV_STAMP(2);

  // ---- phase 4: every input position learns its key's entry -------------------------------
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    const long long i = base + (long long)k * TB + tid;
    if (i < n) w.slot_of_id[i] = tile * TILE + sm.lpos[tslot[k]];
  }
  KV_STAMP(3);

  // ---- phase 5 (optimizer ops): fold the rows of keys that repeat inside the tile ---------
  if constexpr (MODE == MODE_APPLY || MODE == MODE_DEDUP) {
    // offsets of the multi-row keys in the sorted row list (lfirst is free now)
    {
      constexpr int PER = (LS + 1 + TB - 1) / TB;  // 9
      unsigned c[PER];
      unsigned sum = 0;
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const int s = tid * PER + q;
        c[q] = (s <= LS && sm.lcnt[s] >= 2u) ? sm.lcnt[s] : 0u;
        sum += c[q];
      }
      unsigned tot;
      unsigned run = block_excl_scan<TB / 64>(sum, sm.wtot, &tot);
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const int s = tid * PER + q;
        if (s <= LS) { sm.lfirst[s] = run; run += c[q]; }
      }
      (void)tot;
    }
    __syncthreads();  // lkeys is dead from here on: perm / mlist / red alias it
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
      if (tslot[k] != 0xFFFFFFFFu && sm.lcnt[tslot[k]] >= 2u)
        sm.perm[sm.lfirst[tslot[k]] + myrank[k]] = (unsigned short)(k * TB + tid);
    }
    for (int s = tid; s <= LS; s += TB)
      if (sm.lcnt[s] >= 2u) sm.mlist[sm.lpart[s]] = (unsigned short)s;
    __syncthreads();
    KV_STAMP(4);
    const unsigned npart = lnpart;
    const int lane8 = tid & 7;
    const int grp = tid >> 3;
    float* prow0 = w.part + (size_t)tile * PARTCAP * D;
    if constexpr (VPL > 0) {
      const int NV = D >> 2;
      // keys with few rows: one 8-lane group folds all of them, one plain store
      for (unsigned k = grp; k < npart; k += TB / 8) {
        const unsigned s = sm.mlist[k];
        const unsigned cnt = sm.lcnt[s];
        if (cnt > (unsigned)HOT_MIN) continue;
        float4 acc[VPL];
#pragma unroll
        for (int v = 0; v < VPL; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
        fold_rows<VPL>(grad, base, D, sm.perm, sm.lfirst[s], sm.lfirst[s] + cnt, lane8, acc);
        float4* dst = reinterpret_cast<float4*>(prow0 + (size_t)k * D);
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
          const int q = lane8 + 8 * v;
          if (q < NV) dst[q] = acc[v];
        }
      }
      KV_STAMP(5);
      // tile-hot keys: every group folds a slice of every hot key; the slices meet in LDS
      // accumulators (one ds_add_f32 row per group and key), then ONE barrier for all of them
      __shared__ unsigned short hotk[TILE / HOT_MIN];
      __shared__ unsigned lnhot;
      if (tid == 0) lnhot = 0;
      __syncthreads();
      for (unsigned k = tid; k < npart; k += TB)
        if (sm.lcnt[sm.mlist[k]] > (unsigned)HOT_MIN) hotk[atomicAdd(&lnhot, 1u)] = (unsigned short)k;
      __syncthreads();
      const unsigned nhot = lnhot;
      for (unsigned x = tid; x < nhot * (unsigned)D; x += TB) sm.red[x] = 0.f;
      __syncthreads();
      for (unsigned j = 0; j < nhot; ++j) {
        const unsigned s = sm.mlist[hotk[j]];
        const unsigned cnt = sm.lcnt[s];
        const unsigned per = (cnt + TB / 8 - 1) / (TB / 8);
        const unsigned e0 = sm.lfirst[s] + min(cnt, grp * per), e1 = sm.lfirst[s] + min(cnt, (grp + 1) * per);
        if (e0 >= e1) continue;
        float4 acc[VPL];
#pragma unroll
        for (int v = 0; v < VPL; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
        fold_rows<VPL>(grad, base, D, sm.perm, e0, e1, lane8, acc);
        float* rd = sm.red + (size_t)j * D;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
          const int q = lane8 + 8 * v;
          if (q < NV) {
            atomicAdd(&rd[4 * q + 0], acc[v].x); atomicAdd(&rd[4 * q + 1], acc[v].y);
            atomicAdd(&rd[4 * q + 2], acc[v].z); atomicAdd(&rd[4 * q + 3], acc[v].w);
          }
        }
      }
      __syncthreads();
      for (unsigned x = tid; x < nhot * (unsigned)D; x += TB)
        prow0[(size_t)hotk[x / D] * D + (x % D)] = sm.red[x];
    } else {
      // any dim: one thread per element, rows in sorted order
      for (unsigned k = 0; k < npart; ++k) {
        const unsigned s = sm.mlist[k];
        const unsigned cnt = sm.lcnt[s], o = sm.lfirst[s];
        for (int e = tid; e < D; e += TB) {
          float sum = 0.f;
          for (unsigned r = 0; r < cnt; ++r) sum += grad[(size_t)(base + sm.perm[o + r]) * D + e];
          prow0[(size_t)k * D + e] = sum;
        }
      }
    }
  }
  KV_STAMP(6);
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
  long long* out_keys;        // MODE_DEDUP
  float* out_sum;
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

// ---- k_part_keys: MODE_LOOKUP / MODE_SCATTER / MODE_MARK ---------------------------------------
// Streams the partition's entries twice and keeps only the unique keys in LDS, so a key that
// occurs in every tile costs nothing extra.  256 threads, ~33 KB LDS, 4 blocks per CU.
constexpr int TBK = 256;
constexpr int HSK = 2048;          // LDS hash slots
constexpr int UCAPK = HSK * 3 / 4; // unique keys per round

template <int MODE>
__global__ void __launch_bounds__(TBK) k_part_keys(WsDev w, PartArgs a) {
  __shared__ long long hkey[HSK + 1];
  __shared__ unsigned hval[HSK + 1];   // lookup: summed count; scatter / mark: an input position
  __shared__ unsigned hrow[HSK + 1];   // row id of the key
  __shared__ unsigned short lnew[UCAPK + 8];  // slots whose row was inserted now / needs a row scan
  __shared__ unsigned lnu, lsent, lnnew;

  const int tid = threadIdx.x;
  const unsigned p = blockIdx.x;
  const unsigned P = w.P, NT = w.ntiles;
  const int D = a.tv.dim;
  KV_STAMPP(0);

  unsigned R = 1;
  for (unsigned round = 0; round < R; ++round) {
    for (int s = tid; s <= HSK; s += TBK) { hkey[s] = EMPTY_KEY; hval[s] = 0; }
    if (tid == 0) { lnu = 0; lsent = 0; lnnew = 0; }
    __syncthreads();
    // ---- pass 1: unique keys of the partition + their summed counts ---------------------------
    // (segment bounds of up to 8 of this thread's tiles are loaded together: independent loads)
    for (unsigned tb = tid; tb < NT; tb += TBK * 8) {
      unsigned so[8][2];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const unsigned t = tb + q * TBK;
        so[q][0] = so[q][1] = 0;
        if (t < NT) {
          const unsigned short* to = w.toff + (size_t)t * (P + 1) + p;
          so[q][0] = to[0]; so[q][1] = to[1];
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
      const unsigned t = tb + q * TBK;
      const unsigned s0 = so[q][0], s1 = so[q][1];
      for (unsigned e = s0; e < s1; ++e) {
        const size_t ge = (size_t)t * TILE + e;
        const long long key = w.ent_key[ge];
        if (!in_round(key, R, round)) continue;
        if (lnu >= (unsigned)UCAPK) break;  // overflow: this round is abandoned below
        bool first;
        const unsigned h = lds_key_slot<HSK>(hkey, &lsent, key, true, &first);
        if (first) atomicAdd(&lnu, 1u);
        if (MODE == MODE_LOOKUP) atomicAdd(&hval[h], w.ent_a[ge]);
        else if (first) hval[h] = w.ent_a[ge];
      }
      }
    }
    __syncthreads();
    if (lnu >= (unsigned)UCAPK) {  // block-uniform: too many keys -> split by sub-hash and redo
      R = R * 2;
      round = (unsigned)-1;
      __syncthreads();
      continue;
    }
    KV_STAMPP(1);

    // ---- owner work: one thread per unique key ------------------------------------------------
    for (int s = tid; s <= HSK; s += TBK) {
      const bool occ = (s == HSK) ? (lsent != 0) : (hkey[s] != EMPTY_KEY);
      if (!occ) continue;
      const long long key = (s == HSK) ? EMPTY_KEY : hkey[s];
      bool isnew;
      const unsigned r = table_find_or_insert(a.tv, key, &isnew);
      hrow[s] = r;
      if (r == 0) continue;
      unsigned* fp = freq_ptr(a.tv, r);
      unsigned char* fl = flags_ptr(a.tv, r);
      if (MODE == MODE_LOOKUP) {
        // find_func / insert_func (kv_variable.h:320-363): lo16 = sat_add(lo16, batch count),
        // hi16 = today; UpdateUnderThreshold only has work to do when the row changed since the
        // flag was computed (FLAG_DIRTY) or the row is new — every other writer keeps it current
        const unsigned cnt = hval[s];
        unsigned lo = (isnew ? 0u : (*fp & 0xFFFFu)) + (cnt > 65535u ? 65535u : cnt);
        if (lo > 65535u) lo = 65535u;
        *fp = (a.day << 16) | lo;
        const unsigned f = isnew ? FLAG_DIRTY : *fl;
        if (isnew) *fl = (unsigned char)FLAG_DIRTY;
        if (f & FLAG_DIRTY) lnew[atomicAdd(&lnnew, 1u)] = (unsigned short)(s | (isnew ? 0x8000u : 0u));
      } else {
        if (isnew) { *fp = 1u; *fl = 0; }  // EmbeddingValue ctor: freq_val 1, day 0 (table_manager.h:94)
        lnew[atomicAdd(&lnnew, 1u)] = (unsigned short)(s | (isnew ? 0x8000u : 0u));
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
            // ScatterUpdate kv_variable.h:616-734 leaves blacklisted rows alone (:690);
            // InsertOrUpdate :423-485 overwrites
            const unsigned fl = isnew ? 0u : *flags_ptr(a.tv, r);
            touch = a.is_insert || !(fl & FLAG_BLACK);
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
            if (a.mark_what == 0) *fp = (unsigned char)(FLAG_BLACK | FLAG_UNDER);
            else if (isnew) *fp = (unsigned char)(any ? 0u : FLAG_UNDER);
          } else if (touch || isnew) {
            const unsigned black = isnew ? 0u : (*fp & FLAG_BLACK);
            *fp = (unsigned char)(black ? (FLAG_BLACK | FLAG_UNDER) : (any ? 0u : FLAG_UNDER));
          }
        }
      }
    }
    KV_STAMPP(3);

    // ---- pass 2 (lookup): every entry learns its key's row ------------------------------------
    if (MODE == MODE_LOOKUP) {
      for (unsigned tb = tid; tb < NT; tb += TBK * 8) {
        unsigned so[8][2];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const unsigned t = tb + q * TBK;
          so[q][0] = so[q][1] = 0;
          if (t < NT) {
            const unsigned short* to = w.toff + (size_t)t * (P + 1) + p;
            so[q][0] = to[0]; so[q][1] = to[1];
          }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const unsigned t = tb + q * TBK;
          for (unsigned e = so[q][0]; e < so[q][1]; ++e) {
            const size_t ge = (size_t)t * TILE + e;
            const long long key = w.ent_key[ge];
            if (!in_round(key, R, round)) continue;
            bool first;
            w.ent_b[ge] = hrow[lds_key_slot<HSK>(hkey, &lsent, key, false, &first)];
          }
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
// Keeps 6-10 bytes per entry in LDS to group the per-tile contributions of each key, sums them in
// registers (whole block for keys spread over many tiles) and runs the fused row update.
constexpr int TBS = 256;
constexpr int HSS = 2048;
constexpr int UCAPS = HSS * 3 / 4;
constexpr int ECAPS = 2560;
constexpr int HMAXS = 16;                  // pre-summed heavy keys per round kept in LDS
constexpr unsigned LOC_LDS = 0xFFFFFFF0u;  // gradient locator: row of the block's LDS hsum

__host__ __device__ inline size_t part_sum_smem_bytes(int mode, int D, int lpr) {
  size_t b = (size_t)(HSS + 1) * 8 + 16 + (size_t)(HSS + 1) * 4 + 16;  // hkey, hval
  b += (size_t)ECAPS * 4 + 16 + (size_t)ECAPS * 2 * 2 + 32;            // eb, eslot, perm
  b += (size_t)UCAPS * 2 + 16 + 64;                                    // ulist, wtot
  b += (size_t)(TBS / lpr) * D * 4 + 16;                               // red
  b += (size_t)HMAXS * D * 4 + 16;                                     // hsum
  if (mode == MODE_DEDUP) b += (size_t)ECAPS * 4 + 16 + (size_t)(HSS + 1) * 4 + 16;  // eloc, hrow
  return b;
}

// leader lane: var + slot table probes of one key, issued together.
// FindOrInsertUnsafe(var, filter_out != nullptr) kv_variable.h:382-408 and
// FindOrInsertUnsafe(slot, nullptr) :409-414; FTRL probes linear before accum (training_ops.cc:701-704)
template <int OPT>
__device__ __forceinline__ void probe_for_apply(const PartArgs& a, long long key, unsigned* tag,
                                                bool* vnew, unsigned* r0, bool* new0, unsigned* r1,
                                                bool* new1) {
  // read-only probes first: independent loads overlap; inserts (rare) afterwards
  unsigned rv = table_find(a.tv, key);
  unsigned s0 = table_find(a.ts0, key);
  unsigned s1 = (OPT == OPT_FTRL) ? table_find(a.ts1, key) : 0u;
  *vnew = false; *new0 = false; *new1 = false;
  if (rv == 0) {
    rv = table_find_or_insert(a.tv, key, vnew);
    if (rv && *vnew) { *freq_ptr(a.tv, rv) = 1u; *flags_ptr(a.tv, rv) = 0; }
  }
  *tag = rv;
  *r0 = 0; *r1 = 0;
  if (rv == 0) return;
  if (!*vnew) {
    const unsigned f = *freq_ptr(a.tv, rv);
    if ((f & 0xFFFFu) < a.tv.enter_threshold) { *tag = rv | ROW_FILTERED; return; }  // kv_variable.h:910
    unsigned char* fl = flags_ptr(a.tv, rv);
    if (*fl & FLAG_BLACK) *fl = FLAG_UNDER;  // RemoveBlacklistUnsafe: fresh zero row (ours already is)
  }
  if (OPT == OPT_FTRL) {
    if (s1 == 0) s1 = table_find_or_insert(a.ts1, key, new1);
    if (s1) {
      unsigned* fp = freq_ptr(a.ts1, s1);
      if (*new1) *fp = 1u;
      else { unsigned lo = (*fp & 0xFFFFu) + 1u; if (lo > 65535u) lo = 65535u; *fp = (a.day << 16) | lo; }
    }
    *r1 = s1;
  }
  if (s0 == 0) s0 = table_find_or_insert(a.ts0, key, new0);
  if (s0) {
    unsigned* fp = freq_ptr(a.ts0, s0);
    if (*new0) *fp = 1u;
    else { unsigned lo = (*fp & 0xFFFFu) + 1u; if (lo > 65535u) lo = 65535u; *fp = (a.day << 16) | lo; }
  }
  *r0 = s0;
}

template <int MODE, int OPT, int V, int LPR, int K>
__global__ void __launch_bounds__(TBS) k_part_sum(WsDev w, PartArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int D = a.tv.dim;
  char* sp = smem_raw;
  auto take = [&](size_t bytes) { char* q = sp; sp += (bytes + 15) & ~(size_t)15; return q; };
  long long* hkey = reinterpret_cast<long long*>(take((size_t)(HSS + 1) * 8));
  unsigned* hval = reinterpret_cast<unsigned*>(take((size_t)(HSS + 1) * 4));  // entries, then perm offset
  unsigned* eb = reinterpret_cast<unsigned*>(take((size_t)ECAPS * 4));        // gradient locator
  unsigned short* eslot = reinterpret_cast<unsigned short*>(take((size_t)ECAPS * 2));
  unsigned short* perm = reinterpret_cast<unsigned short*>(take((size_t)ECAPS * 2));  // rank, then grouped entries
  unsigned short* ulist = reinterpret_cast<unsigned short*>(take((size_t)UCAPS * 2));
  unsigned* wtot = reinterpret_cast<unsigned*>(take(64));
  float* red = reinterpret_cast<float*>(take((size_t)(TBS / LPR) * D * 4));
  float* hsum = reinterpret_cast<float*>(take((size_t)HMAXS * D * 4));
  unsigned* eloc = nullptr;
  unsigned* hrow = nullptr;
  if (MODE == MODE_DEDUP) {
    eloc = reinterpret_cast<unsigned*>(take((size_t)ECAPS * 4));
    hrow = reinterpret_cast<unsigned*>(take((size_t)(HSS + 1) * 4));
  }
  __shared__ unsigned lnu, lsent, lbase, lovf;

  const int tid = threadIdx.x;
  const unsigned p = blockIdx.x;
  const unsigned P = w.P, NT = w.ntiles;
  constexpr unsigned GPB = TBS / LPR;
  const int lane = tid % LPR;
  const unsigned grp = tid / LPR;
  KV_STAMPP(0);

  unsigned R = 1;
  for (unsigned round = 0; round < R; ++round) {
    // ---- entries of this round: count, then copy in a deterministic order ---------------------
    unsigned cnt = 0;
    for (unsigned t = tid; t < NT; t += TBS) {
      const unsigned short* to = w.toff + (size_t)t * (P + 1) + p;
      const unsigned s0 = to[0], s1 = to[1];
      if (R == 1) cnt += s1 - s0;
      else
        for (unsigned e = s0; e < s1; ++e) cnt += in_round(w.ent_key[(size_t)t * TILE + e], R, round);
    }
    unsigned Er;
    unsigned pos = block_excl_scan<TBS / 64>(cnt, wtot, &Er);
    if (Er == 0) { if (R == 1) return; __syncthreads(); continue; }
    if (Er > (unsigned)ECAPS) {  // block-uniform
      R = R * 2;
      round = (unsigned)-1;
      __syncthreads();
      continue;
    }
    for (int s = tid; s <= HSS; s += TBS) { hkey[s] = EMPTY_KEY; hval[s] = 0; }
    if (tid == 0) { lnu = 0; lsent = 0; lovf = 0; }
    __syncthreads();
    for (unsigned t = tid; t < NT; t += TBS) {
      const unsigned short* to = w.toff + (size_t)t * (P + 1) + p;
      const unsigned s0 = to[0], s1 = to[1];
      for (unsigned e = s0; e < s1; ++e) {
        const size_t ge = (size_t)t * TILE + e;
        const long long key = w.ent_key[ge];
        if (!in_round(key, R, round)) continue;
        if (lnu >= (unsigned)UCAPS) { lovf = 1; break; }
        bool first;
        const unsigned h = lds_key_slot<HSS>(hkey, &lsent, key, true, &first);
        if (first) ulist[atomicAdd(&lnu, 1u)] = (unsigned short)h;
        eb[pos] = w.ent_b[ge];
        if (MODE == MODE_DEDUP) eloc[pos] = (unsigned)ge;
        eslot[pos] = (unsigned short)h;
        perm[pos] = (unsigned short)atomicAdd(&hval[h], 1u);  // rank among the key's entries
        ++pos;
      }
    }
    __syncthreads();
    if (lovf) {  // too many distinct keys for the LDS hash: split further
      R = R * 2;
      round = (unsigned)-1;
      __syncthreads();
      continue;
    }
    KV_STAMPP(1);
    const unsigned nu = lnu;
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
      // hval: count -> (count << 16 | offset) would not fit; keep counts in the high half of a
      // second pass instead: offsets go to hval, counts are recovered as off[next] - off
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
      __syncthreads();
    }
    KV_STAMPP(2);
    if (MODE == MODE_DEDUP) {
      if (tid == 0) lbase = atomicAdd(&w.ctr[0], nu);  // one atomic per partition block and round
      __syncthreads();
    }

    auto sum_slice = [&](unsigned o, unsigned j0, unsigned j1, float (&gv)[K][V]) {
      constexpr int RB = (8 / K) > 0 ? (8 / K) : 1;
      for (unsigned jb = j0; jb < j1; jb += RB) {
        float val[RB][K][V];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
#pragma unroll
          for (int k = 0; k < K; ++k)
#pragma unroll
            for (int c = 0; c < V; ++c) val[r][k][c] = 0.f;
          if (jb + r < j1) {
            const unsigned loc = eb[perm[o + jb + r]];
            if (loc >= LOC_LDS) {  // a heavy key the block pre-summed
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
                if (e0 < D) ldv<V>(src + e0, val[r][k]);
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
    };
    // finish one key: optimizer update, or emit (dedup)
    auto finish = [&](unsigned u, unsigned h, bool live, float (&gv)[K][V]) {
      const long long key = (h == HSS) ? EMPTY_KEY : hkey[h];
      if (MODE == MODE_APPLY) {
        unsigned tag = 0, r0 = 0, r1 = 0;
        bool vnew = false, new0 = false, new1 = false;
        if (live && lane == 0) probe_for_apply<OPT>(a, key, &tag, &vnew, &r0, &new0, &r1, &new1);
        tag = __shfl(tag, 0, LPR);
        r0 = __shfl(r0, 0, LPR);
        r1 = __shfl(r1, 0, LPR);
        const unsigned nb = __shfl((unsigned)vnew | ((unsigned)new0 << 1) | ((unsigned)new1 << 2), 0, LPR);
        vnew = nb & 1u; new0 = (nb >> 1) & 1u; new1 = (nb >> 2) & 1u;
        bool big = false;
        const bool doinit = live && vnew && (tag & ROW_MASK);
        if (doinit) big = init_row_coop(a.tv, key, row_ptr(a.tv, tag & ROW_MASK), lane, LPR);
        const bool any = group_any<LPR>(big);
        if (doinit && lane == 0) *flags_ptr(a.tv, tag & ROW_MASK) = any ? 0 : FLAG_UNDER;
        opt_update_row<OPT, V, LPR, K>(a.tv, a.ts0, a.ts1, key, tag, r0, new0, r1, new1, live, gv, a.opt, lane);
      } else if (live) {
        const unsigned dense = lbase + u;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const int e0 = (lane + k * LPR) * V;
          if (e0 < D) stv<V>(a.out_sum + (size_t)dense * D + e0, gv[k]);
        }
        if (lane == 0) { a.out_keys[dense] = key; hrow[h] = dense; }
      }
    };

    // (a) keys spread over many tiles: the whole block folds slices into one LDS row and the key
    //     then looks like a key with a single contribution; phase (b) finishes it in parallel
    //     with the others (beyond HMAXS such keys per round the first wave finishes in place)
    unsigned nheavy = 0;
    for (unsigned u = 0; u < nu; ++u) {
      const unsigned h = ulist[u];
      const unsigned cn = hval[h] >> 16;
      if (cn <= (unsigned)HEAVY) continue;  // block-uniform
      const unsigned o = hval[h] & 0xFFFFu;
      const unsigned per = (cn + GPB - 1) / GPB;
      float gv[K][V];
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int c = 0; c < V; ++c) gv[k][c] = 0.f;
      sum_slice(o, min(cn, grp * per), min(cn, (grp + 1) * per), gv);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int e0 = (lane + k * LPR) * V;
        if (e0 < D) stv<V>(red + (size_t)grp * D + e0, gv[k]);
      }
      __syncthreads();
      if (nheavy < (unsigned)HMAXS) {
        for (int e = tid; e < D; e += TBS) {
          float sum = 0.f;
          for (unsigned g2 = 0; g2 < GPB; ++g2) sum += red[(size_t)g2 * D + e];  // fixed order
          hsum[(size_t)nheavy * D + e] = sum;
        }
        if (tid == 0) {
          eb[perm[o]] = LOC_LDS + nheavy;
          hval[h] = (1u << 16) | o;
        }
        ++nheavy;
      } else if (tid < 64) {
        const bool live = grp == 0;
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
          for (int c = 0; c < V; ++c) gv[k][c] = 0.f;
        if (live) {
          for (unsigned g2 = 0; g2 < GPB; ++g2) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
              const int e0 = (lane + k * LPR) * V;
              if (e0 < D) {
                float t4[V];
                ldv<V>(red + (size_t)g2 * D + e0, t4);
#pragma unroll
                for (int c = 0; c < V; ++c) gv[k][c] += t4[c];
              }
            }
          }
        }
        finish(u, h, live, gv);
        if (tid == 0) hval[h] = 0xFFFFu << 16;  // done: phase (b) must skip it
      }
      __syncthreads();
    }
    KV_STAMPP(3);
    // (b) everything else: one group per key
    const unsigned upad = (nu + GPB - 1) / GPB * GPB;
    for (unsigned u = grp; u < upad; u += GPB) {
      const unsigned h = u < nu ? ulist[u] : 0u;
      const unsigned cn = u < nu ? (hval[h] >> 16) : 0u;
      const bool live = u < nu && cn != 0xFFFFu;
      float gv[K][V];
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int c = 0; c < V; ++c) gv[k][c] = 0.f;
      if (live) sum_slice(hval[h] & 0xFFFFu, 0, cn, gv);
      finish(u, h, live, gv);
    }
    if (MODE == MODE_DEDUP) {
      __syncthreads();
      for (unsigned e = tid; e < Er; e += TBS) w.ent_b[eloc[e]] = hrow[eslot[e]];
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
__global__ void __launch_bounds__(TB) k_gather(TableDev t, WsDev w, float* __restrict__ out,
                                               long long n) {
  if constexpr (VQ > 0) {
    constexpr int RPB = TB / VQ;  // rows per block per step
    const int v = threadIdx.x % VQ;
    const long long r0 = (long long)blockIdx.x * RPB + threadIdx.x / VQ;
    const long long stride = (long long)gridDim.x * RPB;
    constexpr int UNR = 4;
    for (long long i = r0; i < n; i += stride * UNR) {
      unsigned sl[UNR], rr[UNR];
      float4 val[UNR];
#pragma unroll
      for (int k = 0; k < UNR; ++k) {
        const long long ii = i + k * stride;
        sl[k] = ii < n ? w.slot_of_id[ii] : 0u;
      }
#pragma unroll
      for (int k = 0; k < UNR; ++k) {
        const long long ii = i + k * stride;
        rr[k] = ii < n ? w.ent_b[sl[k]] : 0u;
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
    for (long long x = (long long)blockIdx.x * TB + threadIdx.x; x < total; x += (long long)gridDim.x * TB) {
      const long long i = x / D;
      const int e = (int)(x - i * D);
      out[x] = row_ptr(t, w.ent_b[w.slot_of_id[i]])[e];
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

// kv_dedup_segment_sum: inverse[i] = dense unique index of input position i
__global__ void k_dedup_inverse(WsDev w, long long n, int* inverse) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    inverse[i] = (int)w.ent_b[w.slot_of_id[i]];
}
