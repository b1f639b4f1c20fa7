// kv_part_launch.h — the k_part_sum dispatch on the row geometry, shared by the translation units
// that instantiate it (kv_part_sum_a.hip: GroupAdam V4 / V3; kv_part_sum_b.hip: Adagrad, FTRL,
// dedup).  Splitting the ~120 instantiations over two files lets `make -j` build them in parallel.
// Included inside the anonymous namespace of those files, after kv_device.h and kv_kernels.h.
//
// D % 4 == 0 -> float4 lanes, else scalar lanes.  md != nullptr: one launch over `ntab` tables
// (grid.y), wd carrying the LARGEST ntiles / P of the batch (LDS sizing, grid.x); only MODE_APPLY on
// float4 rows is instantiated for it.  Returns KV_OK, or KV_UNIMPLEMENTED for an unsupported dim.
#pragma once

template <int MODE, int OPT>
int launch_part_sum_t(const WsDev& wd, const PartArgs& pa, hipStream_t s, const MultiDesc* md, int ntab) {
  const int D = pa.tv.dim;
  const int grid = (int)wd.P;
#define KV_PART(V, LPR, K)                                                                       \
  do {                                                                                           \
    if constexpr (MODE == MODE_APPLY && V == 4) {                                                \
      if (md) {                                                                                  \
        k_part_sum_multi<MODE, OPT, V, LPR, K><<<dim3((unsigned)grid, (unsigned)ntab), TBS,      \
            part_sum_smem_bytes(MODE, OPT, D, LPR, wd.ntiles), s>>>(md);                         \
        return KV_OK;                                                                            \
      }                                                                                          \
    }                                                                                            \
    if (md) return KV_UNIMPLEMENTED;                                                             \
    k_part_sum<MODE, OPT, V, LPR, K><<<grid, TBS, part_sum_smem_bytes(MODE, OPT, D, LPR, wd.ntiles), s>>>(wd, pa); \
    return KV_OK;                                                                                \
  } while (0)
  if ((D & 3) == 0) {
    const int q = D / 4;
    if (q <= 1) KV_PART(4, 1, 1);
    if (q <= 2) KV_PART(4, 2, 1);
    if (q <= 4) KV_PART(4, 4, 1);
    if (q <= 8) KV_PART(4, 8, 1);
    if (q <= 16) KV_PART(4, 8, 2);    // dims 36..64: 8 lanes x 2 float4 (measured: 139 -> 120 us at D = 64)
    if (q <= 32) KV_PART(4, 16, 2);   // dims 68..128: 16 lanes x 2 float4 (225 -> 183 us at D = 128)
    if (q <= 64) KV_PART(4, 64, 1);
    if (q <= 128) KV_PART(4, 64, 2);
    if (q <= 256) KV_PART(4, 64, 4);
  } else {
    if (D <= 1) KV_PART(1, 1, 1);
    if (D <= 2) KV_PART(1, 2, 1);
    if (D <= 4) KV_PART(1, 4, 1);
    if (D <= 8) KV_PART(1, 8, 1);
    if (D <= 16) KV_PART(1, 16, 1);
    if (D <= 32) KV_PART(1, 32, 1);
    if (D <= 64) KV_PART(1, 64, 1);
    if (D <= 128) KV_PART(1, 64, 2);
    if (D <= 256) KV_PART(1, 64, 4);
  }
#undef KV_PART
  return KV_UNIMPLEMENTED;
}
