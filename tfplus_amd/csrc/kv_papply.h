// kv_papply.h — the partition pass and the optimizer apply of a batch in ONE launch (included after kv_fused.h).
//
// The round-3 kernels (a partition pass, then an apply over key records) handed the batch from one to the other through
// global memory — key records, the entry list, work items, a dense item directory — and through two dependent chains
// during which HBM idled (28 + 46 us at configs[1] for 112 MB of state traffic).  Here the block that owns a hash
// partition keeps what it learned in LDS and goes on to update its keys itself:
//
//   directory of the partition's segment in every tile -> its entries {key, counts, row word, hint, source}, all loads
//   in flight -> LDS hash of the distinct keys (summed frequency count, row, hint, entries) -> block scans: every
//   key's stretch of the partition's SOURCE LIST (in LDS), keys ordered hot / 1 / 2 / 3.. sources -> the entries'
//   sources filed -> the waves take items from an LDS ticket: a hot key (more than LCOLD sources) per wave, or a batch
//   of 64 / LPR cold keys, one per lane group: sources (gradient rows, or tile sums of k_tsum), the var row and its
//   record, the hinted slot row and its record in ONE round trip -> the lookup's bookkeeping for the key (frequency
//   word, day, under-threshold flag, a new key's record and row: what k_part2 does alone) and the fused row update
//   (opt_core), both by the key's single owner.
//
// No key record, entry list, work item or directory is written; the var record is read once for both purposes; a key
// never spans blocks, so there are no chunks and no k_apply_fin whatever the batch size.  k_tsum (tiles only) runs in
// front: its sums are sources here.
//
// mode: PA_LOOKUP   the batch's training lookup left its partition pass pending: FindOrInsert bookkeeping, then apply
//       PA_APPLYIDX the optimizer meets the ids first (FindOrInsertUnsafe: a new key gets frequency word 1, unfiltered)
//       PA_NONE     the entries of a batch whose bookkeeping is done (a second optimizer on the same token)
//       PA_UNIQUE   no table (the sharded route's index of a rank's local ids; kv_unique / kv_dedup_segment_sum): the distinct
//                   ids get numbers — sparse_unique: partition base + local number, with gaps; else dense, from one counter
//                   (ctr[0], an atomic per block) — and go to out_keys / out_counts (occurrences; dense: the summed counts,
//                   saturating); every entry learns its id's number (ent_b).  No apply phase.
//       PA_DEDUP    the gradient rows of the batch PA_UNIQUE numbered are summed per distinct id and written to
//                   out_sum[out_map[number]] — the records the ids were sent in — instead of updating rows
#pragma once

enum PaMode { PA_LOOKUP = 0, PA_APPLYIDX = 1, PA_NONE = 2, PA_UNIQUE = 3, PA_DEDUP = 4,
              PA_DEDUP_NUM = 5 };   // (a template constant only: PA_DEDUP that numbers the ids itself — kv_dedup_segment_sum)
#ifndef KV_PA_WAVES
#define KV_PA_WAVES 4      // waves per SIMD the register budget is set for (A/B knob: tools/mkvariant.sh)
#endif
#ifndef KV_PA_HOTRB
#define KV_PA_HOTRB 8      // a hot key's sources in flight per lane group and step
#endif
// Two block shapes (A/B at configs[1], 109 k keys: 256 threads x 1024 partitions 64.7 us, 512 x 512 60 us, 1024 x 256
// 75 us; at 773 k keys 512-thread blocks lose: four generations of blocks instead of two):
//   TBP = 512: 2048 hash slots, 2048 sources in LDS — chosen when the batch has at most 512 partitions (one resident
//              generation of two blocks per CU)
//   TBP = 256: 1024 hash slots, 1536 sources — more partitions than that, the deterministic mode, the batched ops
template <int TBP> struct PaShape { static constexpr int HSK = TBP >= 512 ? 2048 : 1024, LSRC = TBP >= 512 ? 2048 : 1536; };

// FM: the mode when the kernel is compiled for one — PA_UNIQUE (k_papply_uniq: numbering only, no row geometry, none of
// the apply's registers) and PA_DEDUP (k_papply_dedup: sources summed and stored, no state, no optimizer) have their own
// kernels and the kernels of the table modes carry none of their code; -1: PA_LOOKUP / PA_APPLYIDX / PA_NONE, an argument
template <int OPT, int V, int LPR, int K, int TBP, int FM = -1>
__device__ __forceinline__ void papply_body(const WsDev& w, const PartArgs& a, const int mode_) {
  constexpr bool UQ = FM == PA_UNIQUE, DN = FM == PA_DEDUP_NUM, DD = FM == PA_DEDUP || DN;
  const int mode = FM >= 0 ? (DN ? (int)PA_DEDUP : FM) : (mode_ & 0xFF);
  constexpr int HSK = PaShape<TBP>::HSK;
  constexpr int PA_LSRC = PaShape<TBP>::LSRC;
  constexpr int UCAPK = (HSK - TBP) < 1023 ? (HSK - TBP) : 1023;   // (the class counters of the key scan are 10-bit fields)
  constexpr int EB = 8;
  constexpr int G = 64 / LPR;
  constexpr int RB = (KV_PA_HOTRB / K) > 0 ? (KV_PA_HOTRB / K) : 1;
  constexpr int NW = TBP / 64;
  __shared__ long long hkey[HSK + 1];
  __shared__ unsigned hval[DD ? 1 : HSK + 1];    // summed frequency count of the key's entries (the per-id sums need none)
  __shared__ unsigned hrow[HSK + 1];    // max over the key's entries of the row word (an entry that knows the row wins)
  __shared__ unsigned hhint[(UQ || DD) ? 1 : HSK + 1];   // slot-row hint (table modes only)
  __shared__ unsigned hocc[HSK + 1];    // entries of the key; then the cursor into the source list (ends at the stretch's end)
  __shared__ unsigned short hcn[HSK + 1];   // entries of the key (final)
  __shared__ unsigned short ulist[UCAPK + 8];
  __shared__ unsigned short kord[UCAPK + 8];   // key slots: hot keys, then 1 / 2 / 3.. sources
  __shared__ unsigned lsrc[PA_LSRC];
  __shared__ unsigned lnu, lsent, lnext, lkeys, lbase;
  __shared__ unsigned rh[MAXW], rbase[MAXW];   // PA_UNIQUE with a route: the round's ids per owner, their first records
  __shared__ unsigned wtot[8];
  __shared__ unsigned stkR[24], stkr[24];
  __shared__ int sp;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  unsigned short* tpre = reinterpret_cast<unsigned short*>(smem_raw);
  unsigned short* tstart = tpre + w.ntiles;

  const int tid = threadIdx.x;
  const unsigned p = xcd_partition(blockIdx.x, w.P);
  const unsigned NT = w.ntiles;
  const int D = a.tv.dim;
  const unsigned errflag = *reinterpret_cast<volatile unsigned*>(&a.tv.counters[1]);   // the tile pass gave up: bookkeeping only
  KV_STAMPP(0);
  unsigned pbase = 0;
  const unsigned E = seg_directory_t<TBP, NW>(w, p, tpre, tstart, wtot, &pbase);
  KV_STAMPP(5);
  if (E == 0) return;
  if (tid == 0) { sp = 0; lkeys = 0; }
  __syncthreads();
  if (E > 65535u) {
    if (tid == 0) raise_error(a.tv, 2u);
    return;
  }
  // the round's sources live in LDS; a larger partition files them in its stretch of w.order
  const bool in_lds = E <= (unsigned)PA_LSRC;
  unsigned* const gsrc = w.order + pbase;   // (the partitions' stretches of w.order are disjoint)
  auto src_at = [&](unsigned i) -> unsigned {
    // a stretch filed in global memory was written by other waves of this block: read past the CU's vector cache
    return in_lds ? lsrc[i] : __hip_atomic_load(gsrc + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };

  // One round = the keys of one sub-hash class (R == 1: all of them).  Returns true when the class holds more distinct
  // keys than the LDS hash: the caller splits it.  A lambda, expanded twice: the common single round is straight-line
  // code (what the partition phases hold in registers is dead when the apply begins), the split rounds loop below.
  auto do_round = [&](const unsigned R, const unsigned round) -> bool {
    __syncthreads();
    for (int s = tid; s <= HSK; s += TBP) {
      hkey[s] = EMPTY_KEY; hrow[s] = 0; hocc[s] = 0;
      if constexpr (!DD) hval[s] = 0;
      if constexpr (!UQ && !DD) hhint[s] = 0;
    }
    if (tid == 0) { lnu = 0; lsent = 0; lnext = 0; }
    __syncthreads();
    // ---- pass 1: distinct keys, their counts, rows and hints (the entries' sources ride along) -------------------
    unsigned csrc[EB];
    unsigned short cslot[EB];
    bool cin[EB];
    const bool cached = (R == 1 && E <= (unsigned)(EB * TBP));
    for (unsigned x0 = 0; x0 < E; x0 += EB * TBP) {
      unsigned ge[EB];
      long long key[EB];
      unsigned ea[EB], rw[EB], hi[EB], sr[EB];
#pragma unroll
      for (int k = 0; k < EB; ++k) {
        const unsigned x = x0 + k * TBP + tid;
        ge[k] = x < E ? (unsigned)seg_entry(tpre, tstart, NT, x) : 0xFFFFFFFFu;
      }
#pragma unroll
      for (int k = 0; k < EB; ++k) {
        key[k] = 0; ea[k] = 0; rw[k] = 0; hi[k] = 0; sr[k] = 0;
        if (ge[k] != 0xFFFFFFFFu) {
          key[k] = w.ent_key[ge[k]]; ea[k] = w.ent_a[ge[k]]; rw[k] = w.ent_b[ge[k]]; hi[k] = w.ent_base[ge[k]];
          if (cached) sr[k] = w.ent_rec[ge[k]];
        }
      }
#pragma unroll
      for (int k = 0; k < EB; ++k) {
        if (x0 == 0) { csrc[k] = sr[k]; cslot[k] = 0; cin[k] = false; }
        if (ge[k] == 0xFFFFFFFFu || !in_round(key[k], R, round)) continue;
        if (lnu >= (unsigned)UCAPK) continue;
        bool first;
        const unsigned h = lds_key_slot<HSK>(hkey, &lsent, key[k], true, &first);
        if (first) {
          const unsigned u = atomicAdd(&lnu, 1u);
          if (u < (unsigned)UCAPK) ulist[u] = (unsigned short)h;
        }
        if constexpr (UQ) atomicAdd(&hval[h], a.sparse_unique ? (ea[k] & 0xFFFFu) : (ea[k] >> 16));   // occurrences of the id in the batch (dense: its summed counts)
        else if constexpr (!DD) { if (mode == PA_LOOKUP) atomicAdd(&hval[h], ea[k] >> 16); }
        atomicAdd(&hocc[h], 1u);
        atomicMax(&hrow[h], rw[k]);
        if constexpr (!UQ && !DD) { if (hi[k]) atomicMax(&hhint[h], hi[k]); }
        if (x0 == 0) { cslot[k] = (unsigned short)h; cin[k] = true; }
      }
    }
    __syncthreads();
    if (lnu >= (unsigned)UCAPK) return true;   // more distinct keys than the hash holds: the caller splits the round
    KV_STAMPP(1);
    const unsigned nu = lnu;
    const unsigned lkeys_before = lkeys;   // distinct keys of the partition's earlier rounds (read before the barrier below lets thread 0 add)
    __syncthreads();
    if (tid == 0) {
      lkeys += nu;
      // dense numbers (kv_unique / kv_dedup_segment_sum): the round's keys take the next nu of one counter
      if ((UQ && !a.sparse_unique) || DN) lbase = atomicAdd(&w.ctr[0], nu);   // (kv_dedup_segment_sum: the sums' pass numbers the ids itself)
    }

    // ---- the keys' stretches of the source list; their order: hot keys, then 1 / 2 / 3.. sources --------------------
    constexpr int PERU = (UCAPK + TBP - 1) / TBP;
    unsigned nhot, ncold;
    {
      unsigned kcnt[PERU];
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
          else hh += 1u;
        }
      }
      unsigned tot, chtot, htot;
      unsigned run = block_excl_scan<NW>(sum, wtot, &tot);
      unsigned chrun = block_excl_scan<NW>(ch, wtot, &chtot);
      unsigned hrn = block_excl_scan<NW>(hh, wtot, &htot);
      const unsigned t1 = chtot & 1023u, t2 = (chtot >> 10) & 1023u;
      nhot = htot; ncold = nu - htot;
#pragma unroll
      for (int q = 0; q < PERU; ++q) {
        const unsigned u = tid * PERU + q;
        if (u < nu) {
          const unsigned s = ulist[u];
          unsigned rank;
          if (kcnt[q] == 1u) { rank = nhot + (chrun & 1023u); chrun += 1u; }
          else if (kcnt[q] == 2u) { rank = nhot + t1 + ((chrun >> 10) & 1023u); chrun += 1u << 10; }
          else if (kcnt[q] <= (unsigned)LCOLD) { rank = nhot + t1 + t2 + (chrun >> 20); chrun += 1u << 20; }
          else { rank = hrn; hrn += 1u; }
          kord[rank] = (unsigned short)s;
          hocc[s] = run; run += kcnt[q];
          hcn[s] = (unsigned short)kcnt[q];
          if (UQ) {   // the id's number: the partition's entries before it bound the numbers before it
            const unsigned num = a.sparse_unique ? pbase + lkeys_before + u : lbase + u;   // (lbase: written before the scans' barriers)
            hrow[s] = num;
            a.out_keys[num] = (s == (unsigned)HSK) ? EMPTY_KEY : hkey[s];
            if (a.sparse_unique) a.out_counts[num] = (int)hval[s];
            else if (a.out_counts) a.out_counts[num] = (int)(hval[s] > 65535u ? 65535u : hval[s]);
          }
          if (DN) {   // the id's dense number = the row of its sum (and what its entries remember for the inverse map)
            const unsigned num = lbase + u;
            hrow[s] = num;
            a.out_keys[num] = (s == (unsigned)HSK) ? EMPTY_KEY : hkey[s];
          }
        }
      }
    }
    __syncthreads();
    if (UQ && a.route_world > 0) {
      // ---- the sharded route: every distinct id to its owner's segment of the send buffer.  The block counts its ids per
      //      owner in LDS and reserves their records with ONE atomic per owner (what k_owner_route_fixed does per 1024 ids)
      if (tid < MAXW) rh[tid] = 0;
      __syncthreads();
      unsigned rd[PERU], rr[PERU];
#pragma unroll
      for (int q = 0; q < PERU; ++q) {
        const unsigned u = tid * PERU + q;
        rd[q] = 0xFFFFFFFFu; rr[q] = 0;
        if (u < nu) {
          const unsigned s = ulist[u];
          const long long key = (s == (unsigned)HSK) ? EMPTY_KEY : hkey[s];
          rd[q] = owner_rank(key, a.route_world, a.route_rule);
          rr[q] = atomicAdd(&rh[rd[q]], 1u);
        }
      }
      __syncthreads();
      if (tid < a.route_world) rbase[tid] = rh[tid] ? atomicAdd(&a.route_gcount[tid], rh[tid]) : 0u;
      __syncthreads();
#pragma unroll
      for (int q = 0; q < PERU; ++q) {
        const unsigned u = tid * PERU + q;
        if (u < nu) {
          const unsigned s = ulist[u];
          const unsigned num = hrow[s];
          const unsigned at = rbase[rd[q]] + rr[q];
          if (at < a.route_C) {
            const size_t slot = (size_t)rd[q] * (a.route_C + 1) + 1 + at;
            a.route_seg[2 * slot] = (s == (unsigned)HSK) ? EMPTY_KEY : hkey[s];
            a.route_seg[2 * slot + 1] = (long long)(int)hval[s];
            a.route_slot_of[num] = (int)slot;
          } else {
            a.route_slot_of[num] = 0;
            atomicExch(a.route_overflow, 1u);
          }
        }
      }
    }

    // ---- pass 2: the source list — entry x of key h goes to its key's stretch ---------------------------------------
    {
      constexpr bool numbers = UQ || DN;   // the entries learn their id's number
      auto file = [&](unsigned pos, unsigned src) { if (in_lds) lsrc[pos] = src; else gsrc[pos] = src; };
      if (a.det) {
        // deterministic mode: a key's entries in tile order = ascending x (TBP entries per round, wave by wave)
        const int wave = tid >> 6, wl = tid & 63;
        for (unsigned x0 = 0; x0 < E; x0 += TBP) {
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
            if (j < wl && hj == h) within += 1u;
          }
          unsigned pos = 0;
          for (int wv = 0; wv < NW; ++wv) {
            if (wave == wv && valid) { pos = hocc[h] + within; atomicAdd(&hocc[h], 1u); }
            __syncthreads();
          }
          if (valid) { file(pos, w.ent_rec[ge]); if (numbers) w.ent_b[ge] = hrow[h]; }
        }
      } else if (cached) {
#pragma unroll
        for (int k = 0; k < EB; ++k)
          if (cin[k]) {
            file(atomicAdd(&hocc[cslot[k]], 1u), csrc[k]);
            if (numbers) w.ent_b[seg_entry(tpre, tstart, NT, (unsigned)(k * TBP + tid))] = hrow[cslot[k]];   // the entry learns its id's number
          }
      } else {
        for (unsigned x = tid; x < E; x += TBP) {
          const size_t ge = seg_entry(tpre, tstart, NT, x);
          const long long key = w.ent_key[ge];
          const unsigned src = w.ent_rec[ge];
          if (!in_round(key, R, round)) continue;
          bool first;
          const unsigned h = lds_key_slot<HSK>(hkey, &lsent, key, false, &first);
          file(atomicAdd(&hocc[h], 1u), src);
          if (numbers) w.ent_b[ge] = hrow[h];
        }
      }
    }
    if (!in_lds) __threadfence_block();
    __syncthreads();
    KV_STAMPP(2);

    // ---- what the apply phase needs per wave --------------------------------------------------------------------
    // (the lane's own numbers through an opaque move: what derives from them is computed HERE, per round, instead of
    //  being hoisted in front of the whole kernel and kept — spilled — across the partition phases)
    unsigned tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    const int wl = (int)(tid2 & 63u), lane = wl % LPR, g = wl / LPR;
    const float* const gbase = a.grad;
    const float* const ebase = a.epart;
    const float* const gself = w.grad_self;   // gradient rows of the positions in the self range (sharded owner apply)
    int eoff[K];
    bool evalid[K];
  #pragma unroll
    for (int k = 0; k < K; ++k) { const int e0 = (lane + k * LPR) * V; evalid[k] = e0 < D; eoff[k] = evalid[k] ? e0 : 0; }
    // the lean update: single-chunk tables, hints, no delta lists — and the var rows carry mirrors of the slot records
    // (kv_device.h SlotMirror; the host's mirror_decide): any other launch or key goes through finish_key
    const bool fast = (OPT != OPT_FTRL) && a.tv.single != 0u && a.ts0.single != 0u && a.use_hints != 0 &&
                      (a.tv.track_delta | a.ts0.track_delta) == 0u && a.use_mirror != 0;
    float* const vrows = a.tv.c0.rows;
    RowMeta* const vmeta = a.tv.c0.meta;
    float* const srows = a.ts0.c0.rows;
    const int SD = a.ts0.dim;
    const unsigned smax = a.ts0.max_rows, thr = a.tv.enter_threshold;
    const bool need_vmeta = OPT == OPT_ADAGRAD || thr != 0u;
    const unsigned mepoch = a.mirror_epoch & 0xFFFFu;
    constexpr int NS0 = (OPT == OPT_ADAM_V4 || OPT == OPT_ADAM_V3) ? 3 : 1;

    // ---- the apply: items = the round's hot keys (one per wave), then batches of G cold keys (one per lane group) -------
    const unsigned nbatch = (ncold + (unsigned)G - 1u) / (unsigned)G;
    const unsigned nitems = nhot + nbatch;
    KV_STAMPPV(6, E); KV_STAMPPV(7, nu); KV_STAMPPV(8, nhot);
#ifdef KV_PA_X_NOAPPLY
    return false;
#endif
    if (UQ) return false;   // (block-uniform) numbering only
#ifdef KV_STAMPS
    unsigned long long st_t0 = wall_clock64(), st_hot = 0, st_cold = 0, st_nh = 0, st_nc = 0;
#endif
    for (;;) {
#ifdef KV_STAMPS
      const unsigned long long st_a = wall_clock64();
#endif
      unsigned it = 0;
      if (wl == 0) it = atomicAdd(&lnext, 1u);
      it = (unsigned)__builtin_amdgcn_readfirstlane((int)it);   // wave-uniform (a scalar): the branches on it are scalar branches
      if (it >= nitems) break;
      // The lane's element offsets through an opaque move, per item.  An address "base + row * D + offset" whose per-lane part
      // is loop-invariant is hoisted in front of the loop as one 64-bit pointer per base — and spilled: the reload from
      // scratch sat in front of the state loads behind an s_waitcnt vmcnt(0) that also waited for the records, a second
      // round trip per item (the ISA of round 5's kernel; profiles/r05_papply_spill.txt).
      int eo[K];
#pragma unroll
      for (int k = 0; k < K; ++k) { eo[k] = eoff[k]; asm volatile("" : "+v"(eo[k])); }
      int lane_i = lane;   // (opt_core / finish_key / prefetch_state derive their own offsets from the lane: the same move)
      asm volatile("" : "+v"(lane_i));
      auto load_row = [&](unsigned pos, float (&dst)[K][V]) {
        const float* base = (pos & EP_TAG) ? ebase : (pos - w.self_lo < w.self_len) ? gself : gbase;
        const float* src = base + (size_t)(pos & ~EP_TAG) * D;
#pragma unroll
        for (int k = 0; k < K; ++k) ldv_stream<V>(src + eo[k], dst[k]);
      };
#ifdef KV_PA_X_NOHOT
      const bool is_hot = false;
#else
      const bool is_hot = it < nhot;
#endif
      // the key of this lane group (hot: every group the same key, group 0 finishes it)
      const unsigned kr = is_hot ? it : nhot + (it - nhot) * (unsigned)G + (unsigned)g;
      const bool have = is_hot || kr < nu;
      const unsigned s = kord[have ? kr : nhot + (it - nhot) * (unsigned)G];
      const long long key = (s == (unsigned)HSK) ? EMPTY_KEY : hkey[s];
      const unsigned cnt = have ? (unsigned)hcn[s] : 0u;
      const unsigned st = hocc[s] - (unsigned)hcn[s];
      const unsigned rww = hrow[s];
      unsigned row = rww & ROW_MASK;
      const bool isnew = have && mode != PA_NONE && (rww & NEW_BIT) != 0u;
      unsigned hint = 0u, fsum = 0u;
      if constexpr (!UQ && !DD) { hint = isnew ? 0u : hhint[s]; fsum = hval[s]; }
      const bool live = is_hot ? g == 0 : have;
      if (__builtin_expect(__ballot(isnew) != 0ull, 0)) {
        // the tile that won the key published {row, HINT_NEW}; the hint goes back to "none"
        unsigned r2 = row;
        if (isnew && lane == 0 && live) {
          Entry* e = table_entry_of(a.tv, key);
          if (e) {
            if (r2 == 0u) { const unsigned er = load_entry(e).row; r2 = er != ROW_TOMB ? er : 0u; }
            e->hint = 0u;
          }
        }
        if (LPR > 1) r2 = __shfl(r2, 0, LPR);
        if (isnew) row = r2;
      }

      // ---- everything the key needs, in one round trip ---------------------------------------------------------------
      float gv[K][V];
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int cc = 0; cc < V; ++cc) gv[k][cc] = 0.f;
      RowMeta m0{};
      uint2 vm = make_uint2(0u, 0u);
      uint4 mir = make_uint4(0u, 0u, 0u, 0u);   // {srow, freq, flags | state << 8 | epoch << 16, -}
      bool hint_loaded = false, have_x = false, have_s = false;
      PreRows<V, K> pre;
      const bool st_live = live && row != 0u;
      auto prefetch = [&]() {
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
            ldv<V>(xr + eo[k], pre.x[k]);
#pragma unroll
            for (int b3 = 0; b3 < NS0; ++b3) ldv<V>(sr + b3 * D + eo[k], pre.s[b3][k]);
          }
          hint_loaded = false; have_x = true; have_s = hh != 0u;   // (the general path reads the slot record itself)
        } else {
          if (st_live && !isnew) vm = load_freq_flags(a.tv, row);
          uint4 rq = make_uint4((unsigned)key, (unsigned)((unsigned long long)key >> 32), st_live ? row : 0u, st_live ? hint : 0u);
          prefetch_state<OPT, V, LPR, K>(a, rq, st_live, lane_i, D, m0, hint_loaded, pre, have_x, have_s);
          if (!have_x) {
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
              for (int cc = 0; cc < V; ++cc) pre.x[k][cc] = 0.f;
          }
        }
      };
      constexpr bool dedup = DD;
      if (is_hot) {
        // ---- hot key: its sources, G * RB per step, summed by the whole wave ------------------------------------------
        const unsigned lo = st, hi = st + cnt;
        constexpr int SR = G * RB;
        const unsigned nst = (cnt + SR - 1) / SR;
        auto ldpos = [&](unsigned stp, unsigned (&pp)[RB]) {
#pragma unroll
          for (int j = 0; j < RB; ++j) {
            const unsigned idx = lo + stp * SR + j * G + g;
            pp[j] = src_at(idx < hi ? idx : lo);   // a slot past the end re-reads the first source, masked below
          }
        };
        unsigned pa_[RB], pb_[RB];
        float va[RB][K][V];
        ldpos(0, pa_);
        if (!dedup) prefetch();
        for (unsigned stp = 0; stp < nst; ++stp) {
#pragma unroll
          for (int j = 0; j < RB; ++j) load_row(pa_[j], va[j]);
          ldpos(stp + 1, pb_);
#pragma unroll
          for (int j = 0; j < RB; ++j) {
            const bool ok = lo + stp * SR + j * G + g < hi;
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
      } else {
        // ---- cold batch: one key per lane group; sources in list order ---------------------------------------------
        float g2[K][V];
        const unsigned p0 = src_at(st < E ? st : 0u);
        const bool two = __ballot(live && cnt >= 2u) != 0ull;
        const unsigned p1 = src_at((cnt >= 2u ? st + 1u : st) < E ? (cnt >= 2u ? st + 1u : st) : 0u);
        load_row(p0, gv);
        if (two) load_row(p1, g2);   // uniform over the wave
        if (!dedup) prefetch();
        if (two) {
#pragma unroll
          for (int k = 0; k < K; ++k)
#pragma unroll
            for (int cc = 0; cc < V; ++cc) gv[k][cc] += (cnt >= 2u) ? g2[k][cc] : 0.f;
        }
#ifndef KV_PA_RC
#define KV_PA_RC 4
#endif
        constexpr int RC = (KV_PA_RC / K) > 0 ? (KV_PA_RC / K) : 1;
        for (unsigned j0 = 2; __ballot(live && j0 < cnt) != 0ull; j0 += RC) {
          float val[RC][K][V];
          unsigned pj[RC];
#pragma unroll
          for (int j = 0; j < RC; ++j) pj[j] = src_at(j0 + j < cnt ? st + j0 + j : (st < E ? st : 0u));
#pragma unroll
          for (int j = 0; j < RC; ++j) load_row(pj[j], val[j]);
#pragma unroll
          for (int j = 0; j < RC; ++j) {
            const bool okj = j0 + j < cnt;
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
              for (int cc = 0; cc < V; ++cc) gv[k][cc] += okj ? val[j][k][cc] : 0.f;
          }
        }
      }

      // ---- the lookup's bookkeeping for the key (k_part2's owner work), on what the round trip brought ----------------
      // kv_variable.h:320-363 (find_func / insert_func) for PA_LOOKUP, :382-416 (FindOrInsertUnsafe) for PA_APPLYIDX
      if (dedup) {   // the id's sum goes to the record the id was sent in (sharded apply: out_map = the id's exchange slot)
        // (direct_rows > 0 — tf.unsorted_segment_sum: the keys ARE the output rows; a key outside [0, direct_rows) is dropped)
        const bool direct = a.direct_rows > 0;
        if (live && (!direct || (unsigned long long)key < (unsigned long long)a.direct_rows)) {
          const size_t orow = direct ? (size_t)key : (size_t)(a.out_map ? a.out_map[row] : (int)row);
          float* dst = a.out_sum + orow * D;
#pragma unroll
          for (int k = 0; k < K; ++k)
            if (evalid[k]) stv<V>(dst + eo[k], gv[k]);
        }
        continue;
      }
      bool vnew = false;   // the apply below treats the key as inserted by itself: never filtered (kv_variable.h:400-407)
      if (mode != PA_NONE) {
        const bool nk = st_live && isnew;
        if (__builtin_expect(__ballot(nk) != 0ull, 0)) {
          if (nk) {   // the init rule's value (kv_variable.h:889-898): the row the update starts from, and what the table holds if it does not act
            const unsigned long long h = pick64((unsigned long long)key ^ (a.tv.seed * 0x9E3779B97F4A7C15ULL));
            const float* ia = a.tv.init_table + (size_t)((unsigned)h % a.tv.init_rows) * D;
            const float* ib = a.tv.init_table + (size_t)((unsigned)(h >> 32) % a.tv.init_rows) * D;
            float* xrow = row_ptr(a.tv, row);
#pragma unroll
            for (int k = 0; k < K; ++k) {
              float va_[V], vb_[V];
              ldv<V>(ia + eo[k], va_);
              ldv<V>(ib + eo[k], vb_);
#pragma unroll
              for (int cc = 0; cc < V; ++cc) pre.x[k][cc] = (va_[cc] + vb_[cc]) * 0.5f;
              if (evalid[k]) stv<V>(xrow + eo[k], pre.x[k]);
            }
            have_x = true;
          }
        }
        const unsigned oflags = nk ? (unsigned)FLAG_DIRTY : (vm.y & 0xFFu);
        const bool recompute = st_live && (mode == PA_LOOKUP ? (oflags & FLAG_DIRTY) != 0u : nk);
        bool big = false;
        if (__ballot(recompute) != 0ull) {
#pragma unroll
          for (int k = 0; k < K; ++k)
#pragma unroll
            for (int cc = 0; cc < V; ++cc) big |= recompute && evalid[k] && fabsf(pre.x[k][cc]) >= CUTOFF;
        }
        const bool any = group_any<LPR>(big);
        if (mode == PA_LOOKUP) {
          if (st_live) {
            const unsigned c = a.count_once ? 1u : fsum;
            unsigned lo16 = ((nk ? 0u : vm.x) & 0xFFFFu) + (c > 65535u ? 65535u : c);
            if (lo16 > 65535u) lo16 = 65535u;
            const unsigned nf = (a.day_lk << 16) | lo16;
            unsigned nfl = oflags;
            if (oflags & FLAG_DIRTY) {
              const unsigned black = nk ? 0u : (oflags & FLAG_BLACK);
              nfl = black ? (FLAG_BLACK | FLAG_UNDER) : (any ? 0u : (unsigned)FLAG_UNDER);
            }
            if (lane == 0) {
              RowMeta* mp = meta_ptr(a.tv, row);
              if (nk) {
                RowMeta nm; nm.key = key; nm.freq = nf; nm.flags = (unsigned char)nfl;
                nm.delta = a.tv.track_delta ? (unsigned char)DELTA_TRAIN : 0; nm.stamp = 0;
                *mp = nm;
              } else {
                mp->freq = nf;
                if (nfl != oflags) mp->flags = (unsigned char)nfl;
                if (a.tv.track_delta) mp->delta |= (unsigned char)DELTA_TRAIN;
              }
            }
            vm.x = nf; vm.y = (vm.y & ~0xFFu) | nfl;
          }
        } else if (nk) {
          const unsigned nfl = any ? 0u : (unsigned)FLAG_UNDER;
          if (lane == 0) {
            RowMeta nm; nm.key = key; nm.freq = 1u; nm.flags = (unsigned char)nfl;
            nm.delta = 0; nm.stamp = 0;
            *meta_ptr(a.tv, row) = nm;
          }
          vm.x = 1u; vm.y = nfl;
          vnew = true;
        }
      }

      // ---- the update (one copy for both kinds of item) ------------------------------------------------------------------
      const bool fin_live = live && errflag == 0u;
      const uint4 ra = make_uint4((unsigned)key, (unsigned)((unsigned long long)key >> 32), row | (vnew ? NEW_BIT : 0u), hint);
      bool general = fin_live;
      if (fast) {
        const unsigned hh = hint < smax ? hint : 0u;
        // the hint stands up: the slot row carries this key and is not released (what resolve_rows checks)
        // ... or, with mirrors: the var row's mirror stands for exactly that slot row in this epoch (established by the
        // general path below or by kv_attach_slot; kv_device.h SlotMirror)
        const bool ok = fin_live && row != 0u && hh != 0u && ((mir.z >> 8) & 0xFFu) != MIRROR_INVALID && (mir.z >> 16) == mepoch &&
                        mir.x == hh;
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
                                 &mp->flags, nullptr, act, false, D, gv, a.opt, lane_i, pre.x, pre.s);
        general = fin_live && !ok;
      }
#ifndef KV_PA_X_NOGENERAL
      if (!fast || __ballot(general) != 0ull)
        finish_key<MODE_APPLY, OPT, V, LPR, K>(a, ra, general, hint_loaded && general, m0, gv, lane_i, &pre, have_x && general,
                                               have_s && general);
#endif
#ifdef KV_STAMPS
      {
        const unsigned long long now = wall_clock64();
        if (is_hot) { st_hot += now - st_a; ++st_nh; } else { st_cold += now - st_a; ++st_nc; }
      }
#endif
    }
#ifdef KV_STAMPS
    if (wl == 0) {
      unsigned long long* d = w.dbg + (size_t)(8192 + blockIdx.x * NW + (tid >> 6)) * 16;
      d[0] = st_t0; d[1] = wall_clock64(); d[2] = st_hot; d[3] = st_cold; d[4] = st_nh; d[5] = st_nc;
    }
#endif
    __syncthreads();
    KV_STAMPP(3);
    return false;
  };
  if (do_round(1u, 0u)) {
    // two sub-hash classes, each on its own; a class that still overflows splits again
    __syncthreads();
    if (tid == 0) { stkR[0] = 2; stkr[0] = 0; stkR[1] = 2; stkr[1] = 1; sp = 2; }
    __syncthreads();
    while (sp > 0) {
      const unsigned R = stkR[sp - 1], round = stkr[sp - 1];
      __syncthreads();
      if (tid == 0) --sp;
      const bool ovf = do_round(R, round);   // (block-uniform)
      __syncthreads();
      if (ovf && tid == 0) {
        if (sp + 2 <= 24) {
          stkR[sp] = 2 * R; stkr[sp] = round; ++sp;
          stkR[sp] = 2 * R; stkr[sp] = round + R; ++sp;
        } else {
          raise_error(a.tv, 2u);
        }
      }
      __syncthreads();
    }
  }
  if (tid == 0 && mode != PA_NONE && !DD) atomicAdd(&w.ctr[5], lkeys);   // distinct keys of the batch: the host's hint for the next batch's partitions
}

// The sharded route (PA_UNIQUE with route_need set): the block that finishes LAST writes every segment's header {records,
// 0}, the largest segment any owner was asked for and the batch's distinct ids, and clears the counters for the next
// launch — k_seg_headers_take without its launch.  Every block comes through here, whatever its partition held.
__device__ __forceinline__ void papply_route_tail(const PartArgs& a, const int mode_, const unsigned nblocks) {
  if ((mode_ & 0xFF) != PA_UNIQUE || a.route_world <= 0 || a.route_need == nullptr) return;   // (uniform over the launch)
  // No fence: the block's adds to the owners' counters are RETURNING device-scope atomics (their values place the
  // block's records), so they have been performed — at the memory side, where every XCD sees them — when the barrier
  // below is passed, before the block takes its ticket; the last block reads the counters with device-scope atomic
  // loads, which bypass its L2 as well.  This rests on how gfx950 performs device-scope atomics, not on the HIP memory
  // model; the model's form — the ticket ACQ_REL, the last block's loads ACQUIRE, one thread per block — was
  // built and measured in round 5: a release is a write-back of the XCD's L2, and 512 of them behind a kernel that left
  // megabytes dirty take the route from 42 to 56 us (sharded world-1 step 0.219 -> 0.233 ms).  (A __threadfence() per
  // thread cost it 100 us.)  Relaxed stays; tests/test_gpu_sharded_two_ranks.py checks the headers on every run.
  __shared__ unsigned rt_last, rt_tot, rt_max;
  __syncthreads();
  if (threadIdx.x == 0) {
    rt_last = __hip_atomic_fetch_add(&a.route_gcount[MAXW], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1u ? 1u : 0u;
    rt_tot = 0u; rt_max = 0u;
  }
  __syncthreads();
  if (!rt_last) return;
  const int d = (int)threadIdx.x;
  if (d < a.route_world) {
    const unsigned c = __hip_atomic_load(&a.route_gcount[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&a.route_gcount[d], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a.route_seg[2 * (size_t)d * (a.route_C + 1)] = c < a.route_C ? c : a.route_C;
    a.route_seg[2 * (size_t)d * (a.route_C + 1) + 1] = 0;
    atomicMax(&rt_max, c);
    atomicAdd(&rt_tot, c);
  }
  __syncthreads();
  if (d == 0) {
    __hip_atomic_store(&a.route_gcount[MAXW], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *a.route_need = rt_max;
    if (a.route_uhint) __hip_atomic_store(a.route_uhint, rt_tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

template <int OPT, int V, int LPR, int K, int TBP>
__global__ void __launch_bounds__(TBP, KV_PA_WAVES) k_papply(WsDev w, PartArgs a, int mode) {
  papply_body<OPT, V, LPR, K, TBP>(w, a, mode);
}

// many tables in one launch (blockIdx.y = table; arguments from the MultiDesc array; grid.x = the largest table's partitions)
// PA_UNIQUE alone (the sharded route's index, kv_unique, kv_dedup_segment_sum's numbering)
template <int TBP>
__global__ void __launch_bounds__(TBP, KV_PA_WAVES) k_papply_uniq(WsDev w, PartArgs a, int mode) {
  papply_body<OPT_ADAM_V4, 4, 1, 1, TBP, PA_UNIQUE>(w, a, mode);
  papply_route_tail(a, mode, gridDim.x);
}
// PA_DEDUP alone (the sharded apply's gradient pre-sum, kv_dedup_segment_sum's sums)
#ifndef KV_PD_WAVES
#define KV_PD_WAVES 6   // (no optimizer state in registers, 41 KB of LDS per 512-thread block: three blocks per CU)
#endif
template <int V, int LPR, int K, int TBP, bool NUM = false>   // NUM: the pass numbers the ids it sums (PartArgs::dd_number)
__global__ void __launch_bounds__(TBP, TBP >= 512 ? KV_PD_WAVES : KV_PA_WAVES) k_papply_dedup(WsDev w, PartArgs a) {
  papply_body<OPT_ADAGRAD, V, LPR, K, TBP, NUM ? PA_DEDUP_NUM : PA_DEDUP>(w, a, PA_DEDUP);
}
template <int V, int LPR, int K>
__global__ void __launch_bounds__(256, KV_PA_WAVES) k_papply_dedup_multi(const MultiDesc* __restrict__ descs) {
  const MultiDesc& m = descs[blockIdx.y];
  if (blockIdx.x >= m.w.P || m.n == 0) return;
  papply_body<OPT_ADAGRAD, V, LPR, K, 256, PA_DEDUP>(m.w, m.a, PA_DEDUP);
}
__global__ void __launch_bounds__(256, KV_PA_WAVES) k_papply_uniq_multi(const MultiDesc* __restrict__ descs, int mode) {
  const MultiDesc& m = descs[blockIdx.y];
  if (blockIdx.x >= m.w.P || m.n == 0) return;
  papply_body<OPT_ADAM_V4, 4, 1, 1, 256, PA_UNIQUE>(m.w, m.a, mode);
  papply_route_tail(m.a, mode, m.w.P);   // (the table's own blocks: the sharded route of several tables in one launch)
}
template <int OPT, int V, int LPR, int K>
__global__ void __launch_bounds__(256, KV_PA_WAVES) k_papply_multi(const MultiDesc* __restrict__ descs, int mode) {
  const MultiDesc& m = descs[blockIdx.y];
  if (blockIdx.x >= m.w.P || m.n == 0) return;
  papply_body<OPT, V, LPR, K, 256>(m.w, m.a, mode);
}

// dispatch on the row geometry (the dims fused_ok() admits: float4 rows, a power-of-two lane count); one block per
// partition.  Returns KV_OK, or KV_UNIMPLEMENTED for a dim the entry-list pipeline does not serve.
// md != nullptr: `ntab` tables in one launch (wd = the largest ntiles / P of the batch of tables)
template <int OPT>
int launch_papply_t(const WsDev& wd, const PartArgs& pa, int mode, hipStream_t s, const MultiDesc* md = nullptr, int ntab = 0) {
  const int D = pa.tv.dim;
  const size_t sh = (size_t)wd.ntiles * 4 + 32;
  if ((D & 3) != 0 || (mode & 0xFF) > PA_NONE) return KV_UNIMPLEMENTED;   // (PA_UNIQUE / PA_DEDUP: launch_papply_ud_t)
#define KV_PA(V, LPR, K)                                                     \
  do {                                                                       \
    if (md) k_papply_multi<OPT, V, LPR, K><<<dim3(wd.P, (unsigned)ntab), 256, sh, s>>>(md, mode);   \
    else if (wd.P <= 512u && !pa.det) k_papply<OPT, V, LPR, K, 512><<<(int)wd.P, 512, sh, s>>>(wd, pa, mode);   \
    else k_papply<OPT, V, LPR, K, 256><<<(int)wd.P, 256, sh, s>>>(wd, pa, mode);       \
    return KV_OK;                                                            \
  } while (0)
  const int q = D / 4;
  if (q <= 1) KV_PA(4, 1, 1);
  if (q <= 2) KV_PA(4, 2, 1);
  if (q <= 4) KV_PA(4, 4, 1);
  if (q <= 8) KV_PA(4, 8, 1);   // (dim 32 with 4 lanes x 2 vectors or 2 x 4 per key — 16 / 32 keys per wave and step: 102 / 177 us against 62.7)
  if (q <= 16) KV_PA(4, 8, 2);
  if (q <= 32) KV_PA(4, 16, 2);
  if (q <= 64) KV_PA(4, 64, 1);
#undef KV_PA
  return KV_UNIMPLEMENTED;
}

// PA_UNIQUE (numbering only: one kernel whatever the dim) and PA_DEDUP (the per-id sums, by row geometry): their own
// kernels, instantiated in their own translation unit (kv_papply_c.hip).  A template so that only that unit holds them.
template <int UNIT>
int launch_papply_ud_t(const WsDev& wd, const PartArgs& pa, int mode, hipStream_t s, const MultiDesc* md = nullptr, int ntab = 0) {
  const size_t sh = (size_t)wd.ntiles * 4 + 32;
  if (mode == PA_UNIQUE) {
    if (md) k_papply_uniq_multi<<<dim3(wd.P, (unsigned)ntab), 256, sh, s>>>(md, mode);
    else if (wd.P <= 512u && !pa.det) k_papply_uniq<512><<<(int)wd.P, 512, sh, s>>>(wd, pa, mode);
    else k_papply_uniq<256><<<(int)wd.P, 256, sh, s>>>(wd, pa, mode);
    return KV_OK;
  }
  const int D = pa.tv.dim;
  if (mode != PA_DEDUP || (D & 3) != 0) return KV_UNIMPLEMENTED;
#define KV_PD(V, LPR, K)                                                     \
  do {                                                                       \
    if (md) k_papply_dedup_multi<V, LPR, K><<<dim3(wd.P, (unsigned)ntab), 256, sh, s>>>(md);   \
    else if (pa.dd_number) {                                                 \
      if (wd.P <= 512u && !pa.det) k_papply_dedup<V, LPR, K, 512, true><<<(int)wd.P, 512, sh, s>>>(wd, pa);   \
      else k_papply_dedup<V, LPR, K, 256, true><<<(int)wd.P, 256, sh, s>>>(wd, pa);   \
    }                                                                        \
    else if (wd.P <= 512u && !pa.det) k_papply_dedup<V, LPR, K, 512><<<(int)wd.P, 512, sh, s>>>(wd, pa);   \
    else k_papply_dedup<V, LPR, K, 256><<<(int)wd.P, 256, sh, s>>>(wd, pa);   \
    return KV_OK;                                                            \
  } while (0)
  const int q = D / 4;
  if (q <= 1) KV_PD(4, 1, 1);
  if (q <= 2) KV_PD(4, 2, 1);
  if (q <= 4) KV_PD(4, 4, 1);
  if (q <= 8) KV_PD(4, 8, 1);
  if (q <= 16) KV_PD(4, 8, 2);
  if (q <= 32) KV_PD(4, 16, 2);
  if (q <= 64) KV_PD(4, 64, 1);
#undef KV_PD
  return KV_UNIMPLEMENTED;
}
