// kv_apply_launch.h — the k_apply_sorted / k_apply_span dispatch on the row geometry, shared by the
// translation units that instantiate it (kv_apply_a.hip: GroupAdam V4 / V3; kv_apply_b.hip: Adagrad, FTRL,
// the plain segment fold).  Splitting the instantiations over two files lets `make -j` build them in
// parallel.  Included inside the anonymous namespace of those files, after kv_device.h and kv_kernels.h.
//
// D % 4 == 0 -> float4 lanes, else scalar lanes.  md != nullptr: one launch over `ntab` tables (grid.y),
// nchunks = blocks per table; only MODE_APPLY on float4 rows is instantiated for it.
// span == 0: k_apply, span == 1: k_apply_fin.  Returns KV_OK, or KV_UNIMPLEMENTED for an unsupported dim.
#pragma once

template <int MODE, int OPT>
int launch_apply_t(const WsDev& wd, const PartArgs& pa, hipStream_t s, const MultiDesc* md, int ntab,
                   unsigned nchunks, int span) {
  const int D = pa.tv.dim;
  int grid_ = (int)nchunks;
#define KV_APPLY(V, LPR, K)                                                                        \
  do {                                                                                             \
    const size_t sh = span == 1 ? (size_t)(TBF / 64) * D * 4 + 16 : 0;                              \
    if (span != 1) {  /* one resident generation of blocks: a second one would start when the first ends */  \
      struct ApTag {};                                                                             \
      const int resident = resident_blocks<ApTag>(k_apply<MODE, OPT, V, LPR, K>, TBS, 6, 4);   /* per device */ \
      if ((int)nchunks > resident) grid_ = resident; else grid_ = (int)nchunks;                     \
    }                                                                                              \
    if constexpr (MODE == MODE_APPLY && V == 4) {                                                  \
      if (md) {                                                                                    \
        if (span) k_apply_fin_multi<MODE, OPT, V, LPR, K><<<dim3((unsigned)grid_, (unsigned)ntab), TBF, sh, s>>>(md);   \
        else k_apply_multi<MODE, OPT, V, LPR, K><<<dim3((unsigned)grid_, (unsigned)ntab), TBS, sh, s>>>(md);      \
        return KV_OK;                                                                              \
      }                                                                                            \
    }                                                                                              \
    if (md) return KV_UNIMPLEMENTED;                                                               \
    if (span) k_apply_fin<MODE, OPT, V, LPR, K><<<grid_, TBF, sh, s>>>(wd, pa);                     \
    else k_apply<MODE, OPT, V, LPR, K><<<grid_, TBS, sh, s>>>(wd, pa);                        \
    return KV_OK;                                                                                  \
  } while (0)
  if ((D & 3) == 0) {
    const int q = D / 4;
    if (q <= 1) KV_APPLY(4, 1, 1);
    if (q <= 2) KV_APPLY(4, 2, 1);
    if (q <= 4) KV_APPLY(4, 4, 1);
    if (q <= 8) KV_APPLY(4, 8, 1);
    if (q <= 16) KV_APPLY(4, 8, 2);    // dims 36..64: 8 lanes x 2 float4
    if (q <= 32) KV_APPLY(4, 16, 2);   // dims 68..128: 16 lanes x 2 float4
    if (q <= 64) KV_APPLY(4, 64, 1);
    if (q <= 128) KV_APPLY(4, 64, 2);
    if (q <= 256) KV_APPLY(4, 64, 4);
  } else {
    if (D <= 1) KV_APPLY(1, 1, 1);
    if (D <= 2) KV_APPLY(1, 2, 1);
    if (D <= 4) KV_APPLY(1, 4, 1);
    if (D <= 8) KV_APPLY(1, 8, 1);
    if (D <= 16) KV_APPLY(1, 16, 1);
    if (D <= 32) KV_APPLY(1, 32, 1);
    if (D <= 64) KV_APPLY(1, 64, 1);
    if (D <= 128) KV_APPLY(1, 64, 2);
    if (D <= 256) KV_APPLY(1, 64, 4);
  }
#undef KV_APPLY
  return KV_UNIMPLEMENTED;
}

// k_tsum (kv_fused.h): the tile sums in front of k_papply; same row geometry as k_apply.  grid = ntiles blocks of TBC
// threads.  Instantiated once (kv_apply_b.hip).  md != nullptr: `ntab` tables in one launch (wd = the largest ntiles).
inline int launch_tsum_t(const TableDev& td, const WsDev& wd, const float* grad, hipStream_t s,
                         const MultiDesc* md = nullptr, int ntab = 0) {
  const int D = td.dim;
#define KV_TSUM(V, LPR, K)                                                                              \
  do {                                                                                                  \
    if (md) k_tsum_multi<V, LPR, K><<<dim3(wd.ntiles, (unsigned)ntab), TBC, 0, s>>>(md);                 \
    else k_tsum<V, LPR, K><<<wd.ntiles, TBC, (size_t)TILE * 4, s>>>(td, wd, grad);                       \
    return KV_OK;                                                                                       \
  } while (0)
  if ((D & 3) != 0) return KV_UNIMPLEMENTED;
  const int q = D / 4;
  if (q <= 1) KV_TSUM(4, 1, 1);
  if (q <= 2) KV_TSUM(4, 2, 1);
  if (q <= 4) KV_TSUM(4, 4, 1);
  if (q <= 8) KV_TSUM(4, 8, 1);
  if (q <= 16) KV_TSUM(4, 8, 2);
  if (q <= 32) KV_TSUM(4, 16, 2);
  if (q <= 64) KV_TSUM(4, 64, 1);
#undef KV_TSUM
  return KV_UNIMPLEMENTED;
}

// k_ltsum (kv_fused.h): the tile pass of a batch and its tile sums in one launch; one block per tile.
template <typename IdT>
inline int launch_ltsum_t(const TableDev& td, const WsDev& wd, const IdT* ids, long long n, int det, const float* grad,
                          hipStream_t s) {
  const int D = td.dim;
  const size_t sh = ltile_smem_bytes();
#define KV_LTSUM(V, LPR, K)                                                                              \
  do {                                                                                                   \
    k_ltsum<IdT, V, LPR, K><<<(int)wd.ntiles, TBT, sh, s>>>(td, wd, ids, nullptr, n, det, grad);           \
    return KV_OK;                                                                                        \
  } while (0)
  if ((D & 3) != 0) return KV_UNIMPLEMENTED;
  const int q = D / 4;
  if (q <= 1) KV_LTSUM(4, 1, 1);
  if (q <= 2) KV_LTSUM(4, 2, 1);
  if (q <= 4) KV_LTSUM(4, 4, 1);
  if (q <= 8) KV_LTSUM(4, 8, 1);
  if (q <= 16) KV_LTSUM(4, 8, 2);
  if (q <= 32) KV_LTSUM(4, 16, 2);
  if (q <= 64) KV_LTSUM(4, 64, 1);
#undef KV_LTSUM
  return KV_UNIMPLEMENTED;
}
