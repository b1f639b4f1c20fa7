// kv_papply_c.hip — instantiates the table-less forms of the partition pass (kv_papply.h): k_papply_uniq (PA_UNIQUE: the
// distinct ids of a batch numbered — the sharded route, kv_unique, kv_dedup_segment_sum) and k_papply_dedup (PA_DEDUP: the
// gradient rows summed per distinct id).  A translation unit of its own (see kv_papply_a.hip).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <type_traits>

#include "../../include/kvhip.h"

namespace {
#include "kv_device.h"
#include "kv_kernels.h"
#include "kv_fused.h"
#include "kv_papply.h"
}  // namespace

extern "C" __attribute__((visibility("hidden"))) int kvp_launch_papply_ud(const void* wd_, const void* pa_, int mode, void* stream,
                                                                       const void* md_, int ntab) {
  const WsDev& wd = *static_cast<const WsDev*>(wd_);
  const PartArgs& pa = *static_cast<const PartArgs*>(pa_);
  return launch_papply_ud_t<0>(wd, pa, mode, static_cast<hipStream_t>(stream), static_cast<const MultiDesc*>(md_), ntab);
}
