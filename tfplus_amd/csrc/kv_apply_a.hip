// kv_apply_a.hip — instantiates k_apply_sorted / k_apply_span for GroupAdam V4 / V3
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

extern "C" __attribute__((visibility("hidden"))) int kvp_launch_apply_a(int mode, int opt, const void* wd_, const void* pa_,
                                                                    void* stream, const void* md_, int ntab,
                                                                    unsigned nchunks, int span) {
  const WsDev& wd = *static_cast<const WsDev*>(wd_);
  const PartArgs& pa = *static_cast<const PartArgs*>(pa_);
  const MultiDesc* md = static_cast<const MultiDesc*>(md_);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (mode == MODE_APPLY && opt == OPT_ADAM_V4) return launch_apply_t<MODE_APPLY, OPT_ADAM_V4>(wd, pa, s, md, ntab, nchunks, span);
  if (mode == MODE_APPLY && opt == OPT_ADAM_V3) return launch_apply_t<MODE_APPLY, OPT_ADAM_V3>(wd, pa, s, md, ntab, nchunks, span);
  return KV_INTERNAL;
}
