// kv_uapply.h — the optimizer apply at the reference's REAL op boundary: unique ids + pre-summed gradient rows, one launch.
//
// In an unchanged TF1 graph the reference's processor patch (python/ops/variable_scope.py:1096-1106) routes a KvVariable's
// IndexedSlices gradient through TF-core's _deduplicate_indexed_slices, so KvVariableGroupSparseApplyAdamV4 and its
// siblings receive `indices` that are already unique and `grad` rows that are already summed (training_ops.cc:7011-7021).
// For that input the batch pipeline (tile dedup, tile sums, partition pass) does work whose answer is known: every id is
// its own key, every gradient row its own sum.  With the caller's PROMISE that the ids are unique (kv_apply_*_unique) the
// op is this one kernel:
//
//   one lane group (LPR lanes) per id, 64 / LPR ids per wave and step: the id -> its index entry (row + slot-row hint; a key
//   the table does not hold yet is inserted on the spot: FindOrInsertUnsafe, kv_variable.h:382-416) -> ONE round trip for
//   the gradient row, the var row and its record, the hinted slot row and its record -> frequency filter / un-blacklisting,
//   AddFrequency on the slot row, the fused row update (opt_core: training_ops.cc row math) -> stores.
//
// A broken promise is caught, not raced.  Every row record carries a 16-bit stamp (RowMeta::stamp): the serial of the last
// unique-apply launch that updated the row.  The key's group reads the stamp with the record and flips it to this
// launch's serial with ONE returning atomic XOR on the record's flag word (performed at the memory side, so it orders
// groups on different XCDs):
//   * the stamp read already equals the serial   -> a group of this launch updated the row before: duplicate, skipped;
//   * the XOR returns a stamp other than the one read -> another group flipped it in between: duplicate.
// Either way the table's error word is raised (code 4; the next call on the table returns KV_INVALID_ARGUMENT) and the
// caller learns that the batch was not applied as the reference would have (sequentially, under the table lock).  The
// host clears all stamps when the 16-bit serial wraps (every 65535 launches per table: one pass over the records).
#pragma once

// FindOrInsertUnsafe (kv_variable.h:382-416) for the unique apply: the caller's probe found no row for `key`.
// *state: 1 = inserted by this call (the row is returned), 2 = the key has an index entry after all — another group of this
// launch holds the same id (returns its row, or 0 while that group has not published it), 3 = the row slab is full.
// The new row's stamp is set by an atomic exchange that has been PERFORMED (at the memory side) before the row is
// published: a group that finds the row through the index reads this launch's serial there, or — from a stale cached
// line — a stamp that its own atomic XOR then shows to be stale.  Either way the duplicate is caught.
__device__ __forceinline__ unsigned uniq_insert(const TableDev& t, long long key, unsigned serial, unsigned* state) {
  Entry* slot;
  unsigned long long stored;
  unsigned long long p = 0;
  const bool sentinel = key == EMPTY_KEY;
  if (sentinel) { slot = &t.entries[t.mask + 1]; stored = 0ull; }
  else { stored = (unsigned long long)key; p = mix64((unsigned long long)key) & t.mask; slot = &t.entries[p]; }
  for (;;) {
    const Entry e = load_entry(slot);
    if ((unsigned long long)e.key == stored && !(sentinel && e.key == EMPTY_KEY)) {
      if (e.row == ROW_TOMB) {   // deleted earlier: the entry is still this key's — give it a row again
        if (atomicCAS(&slot->row, ROW_TOMB, 0u) == ROW_TOMB) break;
        *state = 2u; return 0u;
      }
      *state = 2u; return e.row;
    }
    if (e.key == EMPTY_KEY) {
      const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&slot->key),
                                               (unsigned long long)EMPTY_KEY, stored);
      if (old == (unsigned long long)EMPTY_KEY) break;   // claimed
      if (old == stored) { *state = 2u; return 0u; }
      if (sentinel) continue;
    } else if (sentinel) {
      *state = 2u; return 0u;   // unreachable: entries[cap] holds EMPTY or 0
    }
    if (!sentinel) { p = (p + 1) & t.mask; slot = &t.entries[p]; }
  }
  unsigned r = 0;
  bool have = false;
  if (t.free_rows) {   // rows released by Delete first
    const int f = atomicSub(reinterpret_cast<int*>(&t.counters[2]), 1);
    if (f > 0) { r = t.free_rows[f - 1]; have = true; }
    else atomicAdd(reinterpret_cast<int*>(&t.counters[2]), 1);
  }
  if (!have) {
    r = atomicAdd(&t.counters[0], 1u);
    if (r >= t.max_rows) { raise_error(t, 1u); *state = 3u; return 0u; }
  }
  (void)__hip_atomic_exchange(reinterpret_cast<unsigned*>(&meta_ptr(t, r)->flags), (serial & 0xFFFFu) << 16, __ATOMIC_RELAXED,
                              __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stamp is in place before anyone can find the row
  __hip_atomic_store(&slot->row, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  *state = 1u;
  return r;
}

template <int OPT, int V, int LPR, int K>
__device__ __forceinline__ void uapply_body(const PartArgs& a, const void* __restrict__ ids, int ids32, long long n) {
  constexpr int G = 64 / LPR;
  constexpr int NS0 = (OPT == OPT_ADAM_V4 || OPT == OPT_ADAM_V3) ? 3 : 1;
  if (*reinterpret_cast<volatile unsigned*>(&a.tv.counters[1])) return;   // an earlier batch left the error flag up
  const int D = a.tv.dim;
  const int wl = threadIdx.x & 63, lane = wl % LPR, g = wl / LPR;
  int eoff[K];
  bool evalid[K];
#pragma unroll
  for (int k = 0; k < K; ++k) { const int e0 = (lane + k * LPR) * V; evalid[k] = e0 < D; eoff[k] = evalid[k] ? e0 : 0; }
  // the lean update: single-chunk tables, the var's index entries remember the slot rows, no delta lists (every pre-sized
  // training table); any other key or table goes through finish_key (kv_kernels.h)
  const bool fast = (OPT != OPT_FTRL) && a.tv.single != 0u && a.ts0.single != 0u && a.use_hints != 0 &&
                    (a.tv.track_delta | a.ts0.track_delta) == 0u && a.use_mirror != 0;   // ... and slot mirrors (kv_device.h)
  float* const vrows = a.tv.c0.rows;
  RowMeta* const vmeta = a.tv.c0.meta;
  float* const srows = a.ts0.c0.rows;
  const int SD = a.ts0.dim;
  const unsigned smax = a.ts0.max_rows, thr = a.tv.enter_threshold;
  const bool need_vmeta = OPT == OPT_ADAGRAD || thr != 0u;
  const unsigned mepoch = a.mirror_epoch & 0xFFFFu;
  const unsigned serial = a.uniq_serial & 0xFFFFu;
  const float* const gbase = a.grad;
  const long long nbatch = (n + G - 1) / G;
  const long long nwaves = (long long)gridDim.x * (blockDim.x >> 6);
  // Two batches ahead: the id of batch b + 2 strides and the index entry of batch b + 1 stride are requested while batch b
  // is worked on, so a batch pays one exposed round trip (its rows), not three (id -> entry -> rows).
  auto id_at = [&](long long bb) -> long long {
    long long j = bb * G + g;
    if (j >= n) j = n - 1;   // (past the end: a valid address, the value is not used)
    return ids32 ? (long long)static_cast<const int*>(ids)[j] : static_cast<const long long*>(ids)[j];
  };
  const long long b0 = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  long long key_n = id_at(b0);
  long long key_nn = id_at(b0 + nwaves);
  unsigned long long p_n = home_of(a.tv, key_n, mix64((unsigned long long)key_n));
  Entry e_n = load_entry(&a.tv.entries[p_n]);
  for (long long b = b0; b < nbatch; b += nwaves) {   // wave-uniform
    const long long i = b * G + g;
    const bool live0 = i < n;
    const long long ic = live0 ? i : n - 1;
    const long long key = key_n;
    const unsigned long long p = p_n;
    const Entry e0 = e_n;
    key_n = key_nn;
    p_n = home_of(a.tv, key_n, mix64((unsigned long long)key_n));
    e_n = load_entry(&a.tv.entries[p_n]);
    key_nn = id_at(b + 2 * nwaves);
    unsigned hint = 0;
    unsigned row = table_find_from(a.tv, key, p, e0, &hint);
    bool isnew = false, dup = false;
    if (__builtin_expect(__ballot(live0 && (row == 0u || hint == HINT_NEW)) != 0ull, 0)) {
      // a key the table does not hold yet is inserted by the group's leader (uniq_insert).  An entry marked HINT_NEW belongs
      // to an insert in flight: within one stream-ordered op that is another group of this launch — a duplicate.
      unsigned r2 = row, fl = 0;
      if (live0 && lane == 0) {
        if (row == 0u) r2 = uniq_insert(a.tv, key, serial, &fl);
        else fl = 2u;
      }
      if (LPR > 1) { r2 = __shfl(r2, 0, LPR); fl = __shfl(fl, 0, LPR); }
      if (live0 && (row == 0u || hint == HINT_NEW)) { row = fl == 1u ? r2 : 0u; isnew = fl == 1u; dup = fl == 2u; hint = 0; }
    }
    const bool live = live0 && !dup;
    const bool st_live = live && row != 0u;

    // ---- everything the key needs, in one round trip ---------------------------------------------------------------------
    float gv[K][V];
    {
      const float* src = gbase + (size_t)ic * D;
#pragma unroll
      for (int k = 0; k < K; ++k) ldv_stream<V>(src + eoff[k], gv[k]);
    }
    RowMeta m0{};
    uint2 vm = make_uint2(0u, 0u);
    uint4 mir = make_uint4(0u, 0u, 0u, 0u);   // {srow, freq, flags | state << 8 | epoch << 16, -}
    bool hint_loaded = false, have_x = false, have_s = false;
    PreRows<V, K> pre;
    if (fast) {
      const unsigned rr = st_live ? row : 0u;
      const unsigned hh = (st_live && hint < smax) ? hint : 0u;
      const RowMeta* const vrec = vmeta + (size_t)rr * META_STRIDE;
      mir = *reinterpret_cast<const uint4*>(vrec + 1);   // the slot record's copy, in the var record's own line
      vm = *reinterpret_cast<const uint2*>(&vrec->freq);
      const float* xr = vrows + (size_t)rr * D;
      const float* sr = srows + (size_t)hh * SD;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        ldv<V>(xr + eoff[k], pre.x[k]);
#pragma unroll
        for (int b3 = 0; b3 < NS0; ++b3) ldv<V>(sr + b3 * D + eoff[k], pre.s[b3][k]);
      }
      hint_loaded = false; have_x = true; have_s = hh != 0u;   // (the general path reads the slot record itself)
    } else {
      if (st_live && !isnew) vm = load_freq_flags(a.tv, row);
      const uint4 rq = make_uint4((unsigned)key, (unsigned)((unsigned long long)key >> 32), st_live ? row : 0u, st_live ? hint : 0u);
      prefetch_state<OPT, V, LPR, K>(a, rq, st_live, lane, D, m0, hint_loaded, pre, have_x, have_s);
      if (!have_x) {
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
          for (int cc = 0; cc < V; ++cc) pre.x[k][cc] = 0.f;
      }
    }

    // ---- the promise: nobody else of this launch holds the row (see the head of this file) ----------------------------------
    const unsigned st0 = vm.y >> 16;
    bool mine = st_live && !isnew;
    if (mine && st0 == serial) { dup = true; mine = false; }   // a group of this launch has been here already
    unsigned old = 0;
    const bool asked = mine && lane == 0;
    if (asked)
      old = __hip_atomic_fetch_xor(reinterpret_cast<unsigned*>(&meta_ptr(a.tv, row)->flags), (st0 ^ serial) << 16, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
    const bool go = live && !dup;   // (a duplicate caught by the stamp it read is skipped; one caught by the XOR has raced already)

    // ---- a key inserted now: the init rule's row, record with frequency word 1, never filtered (kv_variable.h:400-407) -------
    bool vnew = false;
    {
      const bool nk = go && st_live && isnew;
      if (__builtin_expect(__ballot(nk) != 0ull, 0)) {
        bool big = false;
        if (nk) {
          const unsigned long long h = pick64((unsigned long long)key ^ (a.tv.seed * 0x9E3779B97F4A7C15ULL));
          const float* ia = a.tv.init_table + (size_t)((unsigned)h % a.tv.init_rows) * D;
          const float* ib = a.tv.init_table + (size_t)((unsigned)(h >> 32) % a.tv.init_rows) * D;
          float* xrow = row_ptr(a.tv, row);
#pragma unroll
          for (int k = 0; k < K; ++k) {
            float va_[V], vb_[V];
            ldv<V>(ia + eoff[k], va_);
            ldv<V>(ib + eoff[k], vb_);
#pragma unroll
            for (int cc = 0; cc < V; ++cc) {
              pre.x[k][cc] = (va_[cc] + vb_[cc]) * 0.5f;
              big |= evalid[k] && fabsf(pre.x[k][cc]) >= CUTOFF;
            }
            if (evalid[k]) stv<V>(xrow + eoff[k], pre.x[k]);
          }
          have_x = true;
        }
        const bool any = group_any<LPR>(big);
        if (nk) {
          const unsigned nfl = any ? 0u : (unsigned)FLAG_UNDER;
          if (lane == 0) {
            RowMeta nm; nm.key = key; nm.freq = 1u; nm.flags = (unsigned char)nfl; nm.delta = 0; nm.stamp = (unsigned short)serial;
            *meta_ptr(a.tv, row) = nm;
          }
          vm.x = 1u; vm.y = nfl | (serial << 16);
          vnew = true;
        }
      }
    }

    // ---- the update ----------------------------------------------------------------------------------------------------------
    const uint4 ra = make_uint4((unsigned)key, (unsigned)((unsigned long long)key >> 32), row | (vnew ? NEW_BIT : 0u), hint);
    bool general = go;
    if (fast) {
      const unsigned hh = hint < smax ? hint : 0u;
      // the hint stands up: the slot row carries this key and is not released (what resolve_rows checks)
      // ... or, with mirrors: the var row's mirror stands for exactly that slot row in this epoch (kv_device.h SlotMirror)
      const bool ok = go && row != 0u && hh != 0u && ((mir.z >> 8) & 0xFFu) != MIRROR_INVALID && (mir.z >> 16) == mepoch && mir.x == hh;
      const unsigned sfreq = mir.y;   // the slot row's frequency word
      bool act = ok;
      if (need_vmeta && ok && !vnew) {   // frequency filter / un-blacklisting (resolve_rows; kv_variable.h:910)
        if ((vm.x & 0xFFFFu) < thr) act = false;
        else if ((vm.y & FLAG_BLACK) && lane == 0) vmeta[(size_t)row * META_STRIDE].flags = FLAG_UNDER;
      }
      const unsigned rr = act ? row : 0u, h2 = act ? hh : 0u;
      SlotMirror* const mp = reinterpret_cast<SlotMirror*>(vmeta + (size_t)rr * META_STRIDE + 1);
      if (act && lane == 0) {   // AddFrequency(1, today) on the slot row (kv_variable.h:409-414)
        unsigned lo16 = (sfreq & 0xFFFFu) + 1u;
        if (lo16 > 65535u) lo16 = 65535u;
        mp->freq = (a.day << 16) | lo16;
        mp->state = (unsigned char)MIRROR_DIRTY;
      }
      opt_core<OPT, V, LPR, K>(vrows + (size_t)rr * D, srows + (size_t)h2 * SD, nullptr, &vmeta[(size_t)rr * META_STRIDE].flags,
                               &mp->flags, nullptr, act, false, D, gv, a.opt, lane, pre.x, pre.s);
      general = go && !ok;
    }
    if (!fast || __ballot(general) != 0ull)
      finish_key<MODE_APPLY, OPT, V, LPR, K>(a, ra, general, hint_loaded && general, m0, gv, lane, &pre, have_x && general,
                                             have_s && general);
    // ---- what the XOR found -----------------------------------------------------------------------------------------------------
    if (asked && (old >> 16) != st0) dup = true;
    if (__builtin_expect(__ballot(dup && lane == 0) != 0ull, 0)) {
      if (dup && lane == 0) raise_error(a.tv, 4u);
    }
  }
}

template <int OPT, int V, int LPR, int K>
__global__ void __launch_bounds__(256, (K == 1 ? 4 : 2)) k_uapply(PartArgs a, const void* __restrict__ ids, int ids32, long long n) {
  uapply_body<OPT, V, LPR, K>(a, ids, ids32, n);
}

// many tables in one launch (blockIdx.y = table; arguments from the MultiDesc array)
template <int OPT, int V, int LPR, int K>
__global__ void __launch_bounds__(256, (K == 1 ? 4 : 2)) k_uapply_multi(const MultiDesc* __restrict__ descs, int ids32) {
  const MultiDesc& m = descs[blockIdx.y];
  if (m.n == 0) return;
  uapply_body<OPT, V, LPR, K>(m.a, m.ids, ids32, m.n);
}

// dispatch on the row geometry (as k_papply: float4 rows, a power-of-two lane count).  KV_UNIMPLEMENTED for other dims:
// the caller falls back to the batch pipeline, which serves them.  md != nullptr: `ntab` tables in one launch (n = the
// largest table's ids, pa = any table's arguments: only the dim is read)
template <int OPT>
int launch_uapply_t(const PartArgs& pa, const void* ids, int ids32, long long n, hipStream_t s, const MultiDesc* md = nullptr,
                    int ntab = 0) {
  const int D = pa.tv.dim;
  if ((D & 3) != 0) return KV_UNIMPLEMENTED;
#define KV_UA(V, LPR, K)                                                                                        \
  do {                                                                                                          \
    constexpr int G = 64 / LPR;                                                                                 \
    struct UaTag {};                                                                                            \
    const int resident = resident_blocks<UaTag>(k_uapply<OPT, V, LPR, K>, 256, 6, 2);   /* per device */          \
    const long long nbatch = (n + G - 1) / G;                                                                   \
    const long long want = (nbatch + 3) / 4;                                                                    \
    int grid = (int)(want < 1 ? 1 : want > resident ? resident : want);                                         \
    if (md) {   /* all tables' blocks are one resident generation */                                            \
      if (ntab > 0 && (long long)grid * ntab > resident) grid = resident / ntab > 1 ? resident / ntab : 1;       \
      k_uapply_multi<OPT, V, LPR, K><<<dim3((unsigned)grid, (unsigned)ntab), 256, 0, s>>>(md, ids32);            \
    } else {                                                                                                    \
      k_uapply<OPT, V, LPR, K><<<grid, 256, 0, s>>>(pa, ids, ids32, n);                                          \
    }                                                                                                           \
    return KV_OK;                                                                                               \
  } while (0)
  const int q = D / 4;
  if (q <= 1) KV_UA(4, 1, 1);
  if (q <= 2) KV_UA(4, 2, 1);
  if (q <= 4) KV_UA(4, 4, 1);
  if (q <= 8) KV_UA(4, 8, 1);
  if (q <= 16) KV_UA(4, 8, 2);
  if (q <= 32) KV_UA(4, 16, 2);
  if (q <= 64) KV_UA(4, 64, 1);
#undef KV_UA
  return KV_UNIMPLEMENTED;
}
