// MOCK (see tensorflow/core/framework/op_kernel.h in this directory): the one declaration the shim's DEVICE_GPU
// kernels take from TensorFlow's GPU helpers.  In TF 2.13 built with TENSORFLOW_USE_ROCM, gpuStream_t is hipStream_t
// and GetGpuStream(ctx) returns the stream of the op's device context.
#pragma once
#include <hip/hip_runtime.h>

#include "tensorflow/core/framework/op_kernel.h"

namespace tensorflow {
using gpuStream_t = hipStream_t;
const gpuStream_t& GetGpuStream(OpKernelContext* context);
}  // namespace tensorflow
