// kv_apply_b.hip — instantiates k_apply_sorted / k_apply_span for Adagrad, SparseGroupFtrl and the plain segment fold (dedup)
// (see kv_apply_launch.h); the entry point below is called by launch_apply() in kvhip.hip.  Arguments
// travel as void pointers because WsDev / PartArgs / MultiDesc live in each file's anonymous
// namespace (same headers, same layout).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <type_traits>

#include "../../include/kvhip.h"

namespace {
#include "kv_device.h"
#include "kv_kernels.h"
#include "kv_fused.h"
#include "kv_apply_launch.h"
}  // namespace

extern "C" __attribute__((visibility("hidden"))) int kvp_launch_apply_b(int mode, int opt, const void* wd_, const void* pa_,
                                                                    void* stream, const void* md_, int ntab,
                                                                    unsigned nchunks, int span) {
  const WsDev& wd = *static_cast<const WsDev*>(wd_);
  const PartArgs& pa = *static_cast<const PartArgs*>(pa_);
  const MultiDesc* md = static_cast<const MultiDesc*>(md_);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (mode == MODE_APPLY && opt == OPT_ADAGRAD) return launch_apply_t<MODE_APPLY, OPT_ADAGRAD>(wd, pa, s, md, ntab, nchunks, span);
  if (mode == MODE_APPLY && opt == OPT_FTRL) return launch_apply_t<MODE_APPLY, OPT_FTRL>(wd, pa, s, md, ntab, nchunks, span);
  if (mode == MODE_DEDUP) return launch_apply_t<MODE_DEDUP, OPT_ADAGRAD>(wd, pa, s, md, ntab, nchunks, span);
  return KV_INTERNAL;
}

// k_tsum of the entry-list pipeline (kv_fused.h); td_ = TableDev of the var table
extern "C" __attribute__((visibility("hidden"))) int kvp_launch_tsum(const void* td_, const void* wd_, const float* grad,
                                                                 void* stream, const void* md, int ntab) {
  return launch_tsum_t(*static_cast<const TableDev*>(td_), *static_cast<const WsDev*>(wd_), grad,
                       static_cast<hipStream_t>(stream), static_cast<const MultiDesc*>(md), ntab);
}

// k_ltsum of the entry-list pipeline (kv_fused.h): tile pass + tile sums; ids_kind 0 int64, 1 int32
extern "C" __attribute__((visibility("hidden"))) int kvp_launch_ltsum(const void* td_, const void* wd_, const void* ids, int ids_kind,
                                                                  long long n, int det, const float* grad, void* stream) {
  const TableDev& td = *static_cast<const TableDev*>(td_);
  const WsDev& wd = *static_cast<const WsDev*>(wd_);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (ids_kind == 1) return launch_ltsum_t<int>(td, wd, static_cast<const int*>(ids), n, det, grad, s);
  return launch_ltsum_t<long long>(td, wd, static_cast<const long long*>(ids), n, det, grad, s);
}
