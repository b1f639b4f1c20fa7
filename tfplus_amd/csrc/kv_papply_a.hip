// kv_papply_a.hip — instantiates k_papply (kv_papply.h: partition pass + optimizer apply in one launch) for GroupAdam V4 / V3.
// A translation unit of its own so that `make -j` builds the instantiations side by side; arguments travel as void
// pointers because WsDev / PartArgs live in each file's anonymous namespace (same headers, same layout).
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
#include "kv_uapply.h"
}  // namespace

extern "C" __attribute__((visibility("hidden"))) int kvp_launch_papply_a(int opt, const void* wd_, const void* pa_, int mode,
                                                                      void* stream, const void* md_, int ntab) {
  const WsDev& wd = *static_cast<const WsDev*>(wd_);
  const PartArgs& pa = *static_cast<const PartArgs*>(pa_);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const MultiDesc* md = static_cast<const MultiDesc*>(md_);
  if (opt == OPT_ADAM_V4) return launch_papply_t<OPT_ADAM_V4>(wd, pa, mode, s, md, ntab);
  if (opt == OPT_ADAM_V3) return launch_papply_t<OPT_ADAM_V3>(wd, pa, mode, s, md, ntab);
  return KV_INTERNAL;
}

// k_uapply (kv_uapply.h): the apply on unique ids + pre-summed rows, one launch
extern "C" __attribute__((visibility("hidden"))) int kvp_launch_uapply_a(int opt, const void* pa_, const void* ids, int ids32,
                                                                      long long n, void* stream, const void* md_, int ntab) {
  const PartArgs& pa = *static_cast<const PartArgs*>(pa_);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const MultiDesc* md = static_cast<const MultiDesc*>(md_);
  if (opt == OPT_ADAM_V4) return launch_uapply_t<OPT_ADAM_V4>(pa, ids, ids32, n, s, md, ntab);
  if (opt == OPT_ADAM_V3) return launch_uapply_t<OPT_ADAM_V3>(pa, ids, ids32, n, s, md, ntab);
  return KV_INTERNAL;
}
