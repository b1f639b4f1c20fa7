// kv_variable_ops_hip.cc — TensorFlow custom-op plug-in over libkvhip.so (include/kvhip.h).
//
// Built where TensorFlow 2.13 headers exist (INTEGRATION.md has the build line; this repository's image
// has none, so tests/test_tf_shim_schema.py checks the op schemas below against the reference's
// REGISTER_OP text instead of compiling them).  Every op of the hot path is registered with
// the reference's name, input order, attr names and defaults:
//   tfplus/kv_variable/ops/kv_variable_ops.cc:37-201 (KvVariable, V2, V3, V4), :203-268 (shape / init / size /
//   frequency / read / destroy), :285-344 (gathers, insert), :520-574 (scatter family);
//   tfplus/kv_variable/ops/training_ops.cc:135-150, 214-226, 1086-1105, 1266-1285 (FtrlV2, Adagrad, GroupAdam V3/V4)
// and forwards Compute() to one C-ABI call.  All semantics live behind the C ABI; this file only moves
// tensors.  tensorflow-cpu keeps tensors in host memory, so they cross PCIe through a per-resource ring of
// pinned staging buffers (no allocation per call once warm; DESIGN.md §4 gives the PCIe bound): the DEVICE_CPU
// kernels.  The DEVICE_GPU kernels at the end of the file (a TensorFlow-ROCm build) hand tensor.data() straight to the
// C ABI on TF's stream: no staging, the measured path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "kvhip.h"
#include "unique_input.h"
#include "tensorflow/core/framework/common_shape_fns.h"
#include "tensorflow/core/framework/node_def.pb.h"
#include "tensorflow/core/framework/op.h"
#include "tensorflow/core/framework/op_kernel.h"
#include "tensorflow/core/framework/resource_mgr.h"
#include "tensorflow/core/framework/shape_inference.h"

namespace tfplus_hip {
using namespace tensorflow;  // NOLINT
using shape_inference::InferenceContext;
using shape_inference::ShapeAndType;
using shape_inference::ShapeHandle;

static Status FromKv(int rc) {
  if (rc == KV_OK) return OkStatus();
  return Status(static_cast<tsl::error::Code>(rc), kv_last_error());
}
static Status FromHip(hipError_t e, const char* what) {
  if (e == hipSuccess) return OkStatus();
  return errors::Internal(what, ": ", hipGetErrorString(e));
}
#define HIP_OK(ctx, expr) OP_REQUIRES_OK(ctx, FromHip((expr), #expr))
#define HIP_RET(expr)                                 \
  do {                                                \
    Status _s = FromHip((expr), #expr);               \
    if (!_s.ok()) return _s;                          \
  } while (0)

// Ring of pinned host + device staging buffers.  A slot is reused only after the stream has passed the event
// recorded behind its last use; buffers grow geometrically and are never freed before the resource dies.
class StagingRing {
 public:
  struct Slot {
    char* host = nullptr;
    char* dev = nullptr;
    size_t cap = 0;
    hipEvent_t done = nullptr;
  };
  ~StagingRing() {
    for (Slot& s : slots_) {
      if (s.done) { hipEventSynchronize(s.done); hipEventDestroy(s.done); }
      if (s.host) hipHostFree(s.host);
      if (s.dev) hipFree(s.dev);
    }
  }
  // a slot with room for `bytes` on both sides (waits for the slot's previous user)
  Status Acquire(size_t bytes, Slot** out) {
    Slot& s = slots_[cursor_++ % kSlots];
    if (s.done) HIP_RET(hipEventSynchronize(s.done));
    if (s.cap < bytes) {
      size_t cap = s.cap ? s.cap : (1u << 20);
      while (cap < bytes) cap *= 2;
      char *h = nullptr, *d = nullptr;
      HIP_RET(hipHostMalloc(reinterpret_cast<void**>(&h), cap));
      hipError_t e = hipMalloc(reinterpret_cast<void**>(&d), cap);
      if (e != hipSuccess) { hipHostFree(h); return FromHip(e, "hipMalloc(staging)"); }
      if (s.host) hipHostFree(s.host);
      if (s.dev) hipFree(s.dev);
      s.host = h; s.dev = d; s.cap = cap;
    }
    if (!s.done) HIP_RET(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    *out = &s;
    return OkStatus();
  }
  // host tensor -> device (asynchronous; the slot stays busy until Release)
  static Status Upload(Slot* s, const void* src, size_t bytes, hipStream_t st) {
    if (bytes == 0) return OkStatus();
    std::memcpy(s->host, src, bytes);
    HIP_RET(hipMemcpyAsync(s->dev, s->host, bytes, hipMemcpyHostToDevice, st));
    return OkStatus();
  }
  // device -> host tensor (synchronous: the op's output must be complete when Compute returns)
  static Status Download(Slot* s, void* dst, size_t bytes, hipStream_t st) {
    if (bytes == 0) return OkStatus();
    HIP_RET(hipMemcpyAsync(s->host, s->dev, bytes, hipMemcpyDeviceToHost, st));
    HIP_RET(hipStreamSynchronize(st));
    std::memcpy(dst, s->host, bytes);
    return OkStatus();
  }
  static Status Release(Slot* s, hipStream_t st) { return FromHip(hipEventRecord(s->done, st), "hipEventRecord"); }

 private:
  static constexpr int kSlots = 6;
  Slot slots_[kSlots];
  unsigned cursor_ = 0;
};

// The resource the handle points at: owns one kv_handle_t, its stream and its staging ring.
class KvHipResource : public ResourceBase {
 public:
  KvHipResource(kv_handle_t h, int dim, DataType key_dtype) : h_(h), dim_(dim), key_dtype_(key_dtype) {
    hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking);
  }
  ~KvHipResource() override {
    if (stream_) hipStreamSynchronize(stream_);
    kv_destroy(h_);
    if (stream_) hipStreamDestroy(stream_);
  }
  string DebugString() const override { return "KvHipResource"; }
  kv_handle_t h() const { return h_; }
  int dim() const { return dim_; }
  DataType key_dtype() const { return key_dtype_; }
  hipStream_t stream() const { return stream_; }
  StagingRing* ring() { return &ring_; }
  std::mutex* mu() { return &mu_; }
  // the batch a training lookup left behind (kv_gather_or_insert_tok): token + what identified the ids
  kv_batch_token_t token = 0;
  const void* token_ids = nullptr;
  int64_t token_n = 0;
  uint64_t token_sum = 0;
  Tensor token_keep;   // DEVICE_GPU: a reference to the lookup's ids tensor, held until the token is used or dropped — while
                       // it is held the allocator cannot hand the buffer to another tensor, so "same address and length"
                       // means "same tensor" (BFC readily reuses a freed buffer of the same size)
  // the init table as the graph gave it (KvVariableExport returns it: dynamic_save.hpp:110-118)
  std::vector<float> init_host;
  int64_t init_rows = 0;

 private:
  kv_handle_t h_;
  int dim_;
  DataType key_dtype_;
  hipStream_t stream_ = nullptr;
  StagingRing ring_;
  std::mutex mu_;   // one Compute at a time moves data through this resource's ring
};

// content fingerprint of an ids tensor: EVERY byte takes part (the token is only passed on when the optimizer op
// receives the very ids the lookup saw — kvhip.h promises a right result only then; a sampled hash would let two
// batches that differ in an unsampled id share an index).  The ids are on the host here and about to be copied
// anyway; one multiply-xor per 8 bytes is nothing next to the PCIe transfer.  TF-core's de-duplicated indices
// never match and take the general path.
static uint64_t Fingerprint(const void* p, size_t bytes) {
  const unsigned char* c = static_cast<const unsigned char*>(p);
  uint64_t h = 1469598103934665603ull ^ bytes;
  size_t i = 0;
  for (; i + 8 <= bytes; i += 8) {
    uint64_t w;
    std::memcpy(&w, c + i, 8);
    h = (h ^ w) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
  }
  for (; i < bytes; ++i) h = (h ^ c[i]) * 1099511628211ull;
  return h;
}

static Status KeyTypeMatches(const KvHipResource* r, const Tensor& ids) {
  const DataType t = ids.dtype();
  const bool ok = (r->key_dtype() == DT_INT32 && t == DT_INT32) ||
                  (r->key_dtype() != DT_INT32 && (t == DT_INT64 || t == DT_UINT64));
  if (!ok) return errors::InvalidArgument("indices dtype ", DataTypeString(t), " does not match the table's key dtype ",
                                          DataTypeString(r->key_dtype()));
  return OkStatus();
}

// handle shape [?, value_shape...] and dtype on the resource output (ops/kv_variable_ops.cc:49-73)
static Status KvVariableShapeFn(InferenceContext* c) {
  c->set_output(0, c->Scalar());
  DataType dtype;
  TF_RETURN_IF_ERROR(c->GetAttr("value_dtype", &dtype));
  PartialTensorShape shape;
  TF_RETURN_IF_ERROR(c->GetAttr("value_shape", &shape));
  if (shape.dims() == 0) return shape_inference::UnknownShape(c);
  shape.InsertDim(0, InferenceContext::kUnknownDim);
  ShapeHandle output_shape;
  TF_RETURN_IF_ERROR(c->MakeShapeFromPartialTensorShape(shape, &output_shape));
  c->set_output_handle_shapes_and_types(0, std::vector<ShapeAndType>{{output_shape, dtype}});
  return OkStatus();
}
static Status ScalarOutput(InferenceContext* c) {
  c->set_output(0, c->Scalar());
  return OkStatus();
}
static Status UnknownOutput(InferenceContext* c) {
  c->set_output(0, c->UnknownShape());
  return OkStatus();
}

// ---- KvVariable / V2 / V3 / V4 : ops/kv_variable_ops.cc:37-201, kernels/kv_variable_ops.cc:31-125 ---------
REGISTER_OP("KvVariable")
    .Output("table_handle: resource")
    .Attr("container: string = ''")
    .Attr("shared_name: string = ''")
    .Attr("use_node_name_sharing: bool = false")
    .Attr("key_dtype: type")
    .Attr("value_dtype: type")
    .Attr("key_shape: shape = {}")
    .Attr("value_shape: shape")
    .Attr("enter_threshold: int = 0")
    .SetIsStateful()
    .SetShapeFn(KvVariableShapeFn);

REGISTER_OP("KvVariableV2")
    .Output("table_handle: resource")
    .Attr("container: string = ''")
    .Attr("shared_name: string = ''")
    .Attr("use_node_name_sharing: bool = false")
    .Attr("key_dtype: type")
    .Attr("value_dtype: type")
    .Attr("key_shape: shape = {}")
    .Attr("value_shape: shape")
    .Attr("initial_num_buckets: int = 131072")
    .Attr("max_load_factor: float = 0.8")
    .Attr("enter_threshold: int = 5")
    .Attr("total_iteration: int = 100000")
    .Attr("worker_num: int = 16")
    .SetIsStateful()
    .SetShapeFn(KvVariableShapeFn);

REGISTER_OP("KvVariableV3")
    .Output("table_handle: resource")
    .Attr("container: string = ''")
    .Attr("shared_name: string = ''")
    .Attr("use_node_name_sharing: bool = false")
    .Attr("key_dtype: type")
    .Attr("value_dtype: type")
    .Attr("key_shape: shape = {}")
    .Attr("value_shape: shape")
    .Attr("enter_threshold: int = 0")
    .Attr("phstore_path: string = ''")
    .SetIsStateful()
    .SetShapeFn(KvVariableShapeFn);

REGISTER_OP("KvVariableV4")
    .Output("table_handle: resource")
    .Attr("container: string = ''")
    .Attr("shared_name: string = ''")
    .Attr("use_node_name_sharing: bool = false")
    .Attr("key_dtype: type")
    .Attr("value_dtype: type")
    .Attr("key_shape: shape = {}")
    .Attr("value_shape: shape")
    .Attr("storage_option: string")
    .Attr("enter_threshold: int = 0")
    .SetIsStateful()
    .SetShapeFn(KvVariableShapeFn);

class CreateKvVariableHipOp : public OpKernel {
 public:
  explicit CreateKvVariableHipOp(OpKernelConstruction* c) : OpKernel(c) {
    OP_REQUIRES_OK(c, c->GetAttr("use_node_name_sharing", &use_node_name_sharing_));
    OP_REQUIRES_OK(c, c->GetAttr("key_dtype", &key_dtype_));
    OP_REQUIRES_OK(c, c->GetAttr("value_dtype", &value_dtype_));
    OP_REQUIRES_OK(c, c->GetAttr("enter_threshold", &enter_threshold_));
    OP_REQUIRES_OK(c, c->GetAttr("value_shape", &value_shape_));
    OP_REQUIRES(c, key_dtype_ == DT_INT32 || key_dtype_ == DT_INT64 || key_dtype_ == DT_UINT64,
                errors::InvalidArgument("key_dtype must be int32, int64 or uint64"));   // kernels/kv_variable_ops.cc:149-156
    OP_REQUIRES(c, value_dtype_ == DT_FLOAT,
                errors::Unimplemented("value_dtype: only float has optimizer kernels (kernels/training_ops.cc:7232)"));
  }
  void Compute(OpKernelContext* ctx) override {
    mutex_lock l(mu_);
    if (!set_) {
      OP_REQUIRES_OK(ctx, cinfo_.Init(ctx->resource_manager(), def(), use_node_name_sharing_));
      KvHipResource* res = nullptr;
      const int dim = static_cast<int>(value_shape_.num_elements());
      const DataType kd = key_dtype_;
      const int thr = enter_threshold_;
      OP_REQUIRES_OK(ctx, cinfo_.resource_manager()->LookupOrCreate<KvHipResource>(
                              cinfo_.container(), cinfo_.name(), &res, [dim, kd, thr](KvHipResource** out) {
                                int dev = 0;
                                TF_RETURN_IF_ERROR(FromHip(hipGetDevice(&dev), "hipGetDevice"));
                                kv_handle_t h = nullptr;
                                TF_RETURN_IF_ERROR(FromKv(kv_create(static_cast<int>(kd), KV_DT_FLOAT, dim, thr, 0, dev, &h)));
                                // the constructor's environment switches (kernels/kv_variable.h:100-111)
                                const char* d = std::getenv("SUPPORT_DELTA_EXPORT");
                                const char* p = std::getenv("SUPPORT_PREDICTION_DELTA_EXPORT");
                                const bool dd = d && std::strcmp(d, "1") == 0, pp = p && std::strcmp(p, "1") == 0;
                                if (dd || pp) TF_RETURN_IF_ERROR(FromKv(kv_set_delta_tracking(h, dd, pp)));
                                // this library's own switch, for graphs that hand the optimizer ops repeated ids (the
                                // batch-token path): TFPLUS_KV_REDUCTION_ORDER = 0 arrival, 1 fixed, 2 TF-core's
                                // occurrence order — the reference's CPU bits (include/kvhip.h kv_set_deterministic)
                                if (const char* ro = std::getenv("TFPLUS_KV_REDUCTION_ORDER"))
                                  if (ro[0] >= '0' && ro[0] <= '2' && ro[1] == 0) TF_RETURN_IF_ERROR(FromKv(kv_set_deterministic(h, ro[0] - '0')));
                                *out = new KvHipResource(h, dim, kd);
                                return OkStatus();
                              }));
      core::ScopedUnref unref(res);
      handle_ = MakeResourceHandle<KvHipResource>(ctx, cinfo_.container(), cinfo_.name());
      set_ = true;
    }
    Tensor* out = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, TensorShape({}), &out));
    out->scalar<ResourceHandle>()() = handle_;
  }

 private:
  mutex mu_;
  bool set_ = false, use_node_name_sharing_ = false;
  DataType key_dtype_, value_dtype_;
  int enter_threshold_ = 0;
  TensorShape value_shape_;
  ContainerInfo cinfo_;
  ResourceHandle handle_;
};
REGISTER_KERNEL_BUILDER(Name("KvVariable").Device(DEVICE_CPU), CreateKvVariableHipOp);
REGISTER_KERNEL_BUILDER(Name("KvVariableV2").Device(DEVICE_CPU), CreateKvVariableHipOp);
REGISTER_KERNEL_BUILDER(Name("KvVariableV3").Device(DEVICE_CPU), CreateKvVariableHipOp);
REGISTER_KERNEL_BUILDER(Name("KvVariableV4").Device(DEVICE_CPU), CreateKvVariableHipOp);

// every kernel below starts the same way
#define KV_RESOURCE(ctx, index, var)                                          \
  KvHipResource* var = nullptr;                                               \
  OP_REQUIRES_OK(ctx, LookupResource(ctx, HandleFromInput(ctx, index), &var)); \
  core::ScopedUnref unref_##var(var)

// ---- KvVariableShapeV2 : ops :203-210, kernels/kv_variable_ops.cc:159-177 -------------------------------
REGISTER_OP("KvVariableShapeV2")
    .Input("table_handle: resource")
    .Output("output: out_type")
    .Attr("out_type: {int32, int64} = DT_INT32")
    .SetShapeFn([](InferenceContext* c) {
      c->set_output(0, c->Vector(InferenceContext::kUnknownDim));
      return OkStatus();
    });

template <typename T>
class KvShapeHipOp : public OpKernel {
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    KV_RESOURCE(ctx, 0, r);
    int64_t rows = 0;
    OP_REQUIRES_OK(ctx, FromKv(kv_map_size(r->h(), &rows, r->stream())));
    Tensor* out = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, TensorShape({2}), &out));
    out->vec<T>()(0) = static_cast<T>(rows);
    out->vec<T>()(1) = static_cast<T>(r->dim());
  }
};
REGISTER_KERNEL_BUILDER(Name("KvVariableShapeV2").Device(DEVICE_CPU).TypeConstraint<int32>("out_type"), KvShapeHipOp<int32>);
REGISTER_KERNEL_BUILDER(Name("KvVariableShapeV2").Device(DEVICE_CPU).TypeConstraint<int64_t>("out_type"), KvShapeHipOp<int64_t>);

// ---- InitKvVariableV2 : ops :212-222, kernels/kv_variable_ops.cc:188-200 ---------------------------------
REGISTER_OP("InitKvVariableV2")
    .Input("table_handle: resource")
    .Input("random_initializer: T")
    .Attr("T: type")
    .SetShapeFn([](InferenceContext* c) {
      ShapeHandle handle;
      TF_RETURN_IF_ERROR(c->WithRank(c->input(1), 2, &handle));   // the second input must be 2-D
      return OkStatus();
    });

class InitKvVariableHipOp : public OpKernel {
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    KV_RESOURCE(ctx, 0, r);
    const Tensor& t = ctx->input(1);
    OP_REQUIRES(ctx, t.dtype() == DT_FLOAT, errors::InvalidArgument("random_initializer must be float"));
    OP_REQUIRES(ctx, t.dims() == 2 && t.dim_size(1) == r->dim(),
                errors::InvalidArgument("random_initializer must be [rows, ", r->dim(), "]"));
    std::lock_guard<std::mutex> l(*r->mu());
    StagingRing::Slot* s = nullptr;
    OP_REQUIRES_OK(ctx, r->ring()->Acquire(t.TotalBytes(), &s));
    OP_REQUIRES_OK(ctx, StagingRing::Upload(s, t.data(), t.TotalBytes(), r->stream()));
    OP_REQUIRES_OK(ctx, FromKv(kv_init_table(r->h(), reinterpret_cast<const float*>(s->dev), t.dim_size(0), r->stream())));
    OP_REQUIRES_OK(ctx, StagingRing::Release(s, r->stream()));
    if (r->init_rows == 0) {   // first call wins, like the table's own copy (kv_variable.h:188-193)
      const float* src = static_cast<const float*>(t.data());
      r->init_host.assign(src, src + t.NumElements());
      r->init_rows = t.dim_size(0);
    }
  }
};
REGISTER_KERNEL_BUILDER(Name("InitKvVariableV2").Device(DEVICE_CPU), InitKvVariableHipOp);

// ---- IsInitialized / Size / Frequency : ops :224-247, kernels/kv_variable_ops.cc:202-293 ---------------
REGISTER_OP("KvVariableIsInitializedV2")
    .Input("table_handle: resource")
    .Output("is_initialized: bool")
    .SetShapeFn(ScalarOutput);

class KvIsInitializedHipOp : public OpKernel {
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    Tensor* out = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, TensorShape({}), &out));
    KvHipResource* r = nullptr;
    if (!LookupResource(ctx, HandleFromInput(ctx, 0), &r).ok()) {   // no resource yet: not initialized (:207-213)
      out->scalar<bool>()() = false;
      return;
    }
    core::ScopedUnref unref(r);
    int v = 0;
    OP_REQUIRES_OK(ctx, FromKv(kv_is_initialized(r->h(), &v)));
    out->scalar<bool>()() = v != 0;
  }
};
REGISTER_KERNEL_BUILDER(Name("KvVariableIsInitializedV2").Device(DEVICE_CPU), KvIsInitializedHipOp);

REGISTER_OP("KvVariableSizeV2")
    .Input("table_handle: resource")
    .Output("size: T")
    .Attr("T: {int32, int64} = DT_INT64")
    .SetShapeFn(ScalarOutput);

REGISTER_OP("KvVariableFrequency")
    .Input("table_handle: resource")
    .Output("size: T")
    .Attr("T: {int32, int64} = DT_INT64")
    .SetShapeFn(ScalarOutput);

template <typename T, bool FREQ>
class KvSizeHipOp : public OpKernel {
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    KV_RESOURCE(ctx, 0, r);
    int64_t v = 0;
    OP_REQUIRES_OK(ctx, FromKv(FREQ ? kv_sum_freq(r->h(), &v, r->stream()) : kv_size(r->h(), &v, r->stream())));
    Tensor* out = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, TensorShape({}), &out));
    out->scalar<T>()() = static_cast<T>(v);
  }
};
REGISTER_KERNEL_BUILDER(Name("KvVariableSizeV2").Device(DEVICE_CPU).TypeConstraint<int32>("T"), KvSizeHipOp<int32, false>);
REGISTER_KERNEL_BUILDER(Name("KvVariableSizeV2").Device(DEVICE_CPU).TypeConstraint<int64_t>("T"), KvSizeHipOp<int64_t, false>);
REGISTER_KERNEL_BUILDER(Name("KvVariableFrequency").Device(DEVICE_CPU).TypeConstraint<int32>("T"), KvSizeHipOp<int32, true>);
REGISTER_KERNEL_BUILDER(Name("KvVariableFrequency").Device(DEVICE_CPU).TypeConstraint<int64_t>("T"), KvSizeHipOp<int64_t, true>);

// ---- ReadKvVariableOpV2 : ops :249-266, kernels/kv_variable_ops.cc:325-346 -> ExportValues(first_n = 2) ---
REGISTER_OP("ReadKvVariableOpV2")
    .Input("table_handle: resource")
    .Output("keys: Tkeys")
    .Output("values: Tvalues")
    .Attr("Tkeys: type")
    .Attr("Tvalues: type")
    .SetShapeFn([](InferenceContext* c) {
      ShapeHandle handle;
      TF_RETURN_IF_ERROR(c->WithRank(c->input(0), 0, &handle));
      ShapeHandle values = c->UnknownShape();
      TF_RETURN_IF_ERROR(c->WithRankAtLeast(values, 1, &values));
      c->set_output(0, c->Vector(c->Dim(values, 0)));
      c->set_output(1, values);
      return OkStatus();
    });

class ReadKvVariableHipOp : public OpKernel {
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    KV_RESOURCE(ctx, 0, r);
    OP_REQUIRES(ctx, r->key_dtype() != DT_INT32, errors::Unimplemented("ReadKvVariableOpV2 with int32 keys"));
    std::lock_guard<std::mutex> l(*r->mu());
    int64_t counts[3] = {0, 0, 0};
    OP_REQUIRES_OK(ctx, FromKv(kv_export_count(r->h(), 2, counts, r->stream())));
    const int64_t m = counts[0];
    Tensor *keys = nullptr, *values = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, TensorShape({m}), &keys));
    OP_REQUIRES_OK(ctx, ctx->allocate_output(1, TensorShape({m, r->dim()}), &values));
    if (m == 0) return;
    StagingRing::Slot *sk = nullptr, *sv = nullptr;
    OP_REQUIRES_OK(ctx, r->ring()->Acquire(keys->TotalBytes(), &sk));
    OP_REQUIRES_OK(ctx, r->ring()->Acquire(values->TotalBytes(), &sv));
    // the table cannot change between the two calls: this resource's mutex is held, and every op that writes
    // the table goes through a kernel of this file
    OP_REQUIRES_OK(ctx, FromKv(kv_export_fill(r->h(), 2, reinterpret_cast<int64_t*>(sk->dev), reinterpret_cast<float*>(sv->dev),
                                              nullptr, nullptr, nullptr, r->stream())));
    OP_REQUIRES_OK(ctx, StagingRing::Download(sk, keys->data(), keys->TotalBytes(), r->stream()));
    OP_REQUIRES_OK(ctx, StagingRing::Download(sv, values->data(), values->TotalBytes(), r->stream()));
    OP_REQUIRES_OK(ctx, StagingRing::Release(sk, r->stream()));
    OP_REQUIRES_OK(ctx, StagingRing::Release(sv, r->stream()));
  }
};
REGISTER_KERNEL_BUILDER(Name("ReadKvVariableOpV2").Device(DEVICE_CPU), ReadKvVariableHipOp);

// ---- DestroyKvVariableOpV2 : ops :268-272, kernels/kv_variable_ops.cc:295-323 ----------------------------
REGISTER_OP("DestroyKvVariableOpV2")
    .Input("table_handle: resource")
    .Attr("ignore_lookup_error: bool = true")
    .SetIsStateful()
    .SetShapeFn(shape_inference::NoOutputs);

class DestroyKvVariableHipOp : public OpKernel {
 public:
  explicit DestroyKvVariableHipOp(OpKernelConstruction* c) : OpKernel(c) {
    OP_REQUIRES_OK(c, c->GetAttr("ignore_lookup_error", &ignore_lookup_error_));
  }
  void Compute(OpKernelContext* ctx) override {
    const Status s = DeleteResource(ctx, HandleFromInput(ctx, 0));
    if (ignore_lookup_error_ && errors::IsNotFound(s)) return;
    OP_REQUIRES_OK(ctx, s);
  }

 private:
  bool ignore_lookup_error_ = true;
};
REGISTER_KERNEL_BUILDER(Name("DestroyKvVariableOpV2").Device(DEVICE_CPU), DestroyKvVariableHipOp);

// ---- lookups : ops :285-332, kernels/kv_variable_ops.cc:348-405, 498-538, 564-606 ------------------------
REGISTER_OP("KvVariableGatherOrZerosV2")
    .Input("table_handle: resource")
    .Input("indices: Tindices")
    .Output("output: dtype")
    .Attr("dtype: type")
    .Attr("Tindices: {int32, int64, uint64, string}")
    .SetShapeFn(UnknownOutput);

REGISTER_OP("KvVariableGatherOrInsertV2")
    .Input("table_handle: resource")
    .Input("indices: Tindices")
    .Output("output: dtype")
    .Attr("dtype: type")
    .Attr("Tindices: {int32, int64, uint64, string}")
    .SetShapeFn(UnknownOutput);

REGISTER_OP("KvVariableGatherOrInsertWithCounts")
    .Input("table_handle: resource")
    .Input("indices: Tindices")
    .Input("counts: int32")
    .Output("output: dtype")
    .Attr("dtype: type")
    .Attr("Tindices: {int32, int64, uint64, string}")
    .SetShapeFn(UnknownOutput);

// MODE 0 = GatherOrZeros, 1 = GatherOrInsert, 2 = GatherOrInsertWithCounts
template <int MODE>
class KvGatherHipOp : public OpKernel {
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    KV_RESOURCE(ctx, 0, r);
    const Tensor& ids = ctx->input(1);
    OP_REQUIRES_OK(ctx, KeyTypeMatches(r, ids));
    TensorShape shape = ids.shape();
    shape.AddDim(r->dim());                               // output shape = indices.shape + [dim] (kernels :516-522)
    Tensor* out = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, shape, &out));
    const int64_t n = ids.NumElements();
    if (n == 0) return;                                   // :530-532
    const Tensor* counts = nullptr;
    if (MODE == 2) {
      counts = &ctx->input(2);
      // kernels/kv_variable.h:268-280
      OP_REQUIRES(ctx, counts->dtype() == DT_INT32, errors::InvalidArgument("increment count, counts dtype must be int32"));
      OP_REQUIRES(ctx, counts->shape() == ids.shape(),
                  errors::InvalidArgument("increment count, indices shape ", ids.shape().DebugString(),
                                          " does not match with counts shape ", counts->shape().DebugString()));
    }
    std::lock_guard<std::mutex> l(*r->mu());
    StagingRing::Slot *si = nullptr, *so = nullptr, *sc = nullptr;
    OP_REQUIRES_OK(ctx, r->ring()->Acquire(ids.TotalBytes(), &si));
    OP_REQUIRES_OK(ctx, r->ring()->Acquire(out->TotalBytes(), &so));
    OP_REQUIRES_OK(ctx, StagingRing::Upload(si, ids.data(), ids.TotalBytes(), r->stream()));
    if (counts) {
      OP_REQUIRES_OK(ctx, r->ring()->Acquire(counts->TotalBytes(), &sc));
      OP_REQUIRES_OK(ctx, StagingRing::Upload(sc, counts->data(), counts->TotalBytes(), r->stream()));
    }
    int rc;
    if (MODE == 0) {
      rc = kv_gather_or_zeros(r->h(), si->dev, n, reinterpret_cast<float*>(so->dev), r->stream());
    } else {
      kv_batch_token_t tok = 0;
      rc = kv_gather_or_insert_tok(r->h(), si->dev, sc ? reinterpret_cast<const int32_t*>(sc->dev) : nullptr, n,
                                   reinterpret_cast<float*>(so->dev), &tok, r->stream());
      r->token = tok; r->token_ids = si->dev; r->token_n = n;
      r->token_sum = Fingerprint(ids.data(), ids.TotalBytes());
    }
    OP_REQUIRES_OK(ctx, FromKv(rc));
    OP_REQUIRES_OK(ctx, StagingRing::Download(so, out->data(), out->TotalBytes(), r->stream()));
    OP_REQUIRES_OK(ctx, StagingRing::Release(si, r->stream()));
    OP_REQUIRES_OK(ctx, StagingRing::Release(so, r->stream()));
    if (sc) OP_REQUIRES_OK(ctx, StagingRing::Release(sc, r->stream()));
  }
};
// table_handle is HostMemory like in the reference (kernels/kv_variable_ops.cc:540-546); string keys have no kernel
#define KV_REGISTER_GATHER(NAME, MODE)                                                                                     \
  REGISTER_KERNEL_BUILDER(Name(NAME).Device(DEVICE_CPU).HostMemory("table_handle").TypeConstraint<int32>("Tindices")      \
                              .TypeConstraint<float>("dtype"), KvGatherHipOp<MODE>);                                         \
  REGISTER_KERNEL_BUILDER(Name(NAME).Device(DEVICE_CPU).HostMemory("table_handle").TypeConstraint<int64_t>("Tindices")    \
                              .TypeConstraint<float>("dtype"), KvGatherHipOp<MODE>);                                         \
  REGISTER_KERNEL_BUILDER(Name(NAME).Device(DEVICE_CPU).HostMemory("table_handle").TypeConstraint<uint64>("Tindices")     \
                              .TypeConstraint<float>("dtype"), KvGatherHipOp<MODE>)
KV_REGISTER_GATHER("KvVariableGatherOrZerosV2", 0);
KV_REGISTER_GATHER("KvVariableGatherOrInsertV2", 1);
KV_REGISTER_GATHER("KvVariableGatherOrInsertWithCounts", 2);

// ---- KvVariableInsertV2 and the scatter family : ops :334-344, 520-574, kernels/kv_variable_ops.cc:703-747, 1097-1161
REGISTER_OP("KvVariableInsertV2")
    .Input("table_handle: resource")
    .Input("indices: Tindices")
    .Input("values: dtype")
    .Attr("dtype: type")
    .Attr("Tindices: {int32, int64, uint64, string}")
    .SetShapeFn([](InferenceContext*) { return OkStatus(); });

// OP < 0: InsertOrUpdate; else the kv_scatter_update operation
template <int OP>
class KvScatterHipOp : public OpKernel {
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    KV_RESOURCE(ctx, 0, r);
    const Tensor& ids = ctx->input(1);
    const Tensor& vals = ctx->input(2);
    OP_REQUIRES_OK(ctx, KeyTypeMatches(r, ids));
    const int64_t n = ids.NumElements();
    OP_REQUIRES(ctx, vals.dtype() == DT_FLOAT && vals.NumElements() == n * r->dim(),
                errors::InvalidArgument("updates must be [indices..., ", r->dim(), "] float"));
    if (n == 0) return;
    std::lock_guard<std::mutex> l(*r->mu());
    StagingRing::Slot *si = nullptr, *sv = nullptr;
    OP_REQUIRES_OK(ctx, r->ring()->Acquire(ids.TotalBytes(), &si));
    OP_REQUIRES_OK(ctx, r->ring()->Acquire(vals.TotalBytes(), &sv));
    OP_REQUIRES_OK(ctx, StagingRing::Upload(si, ids.data(), ids.TotalBytes(), r->stream()));
    OP_REQUIRES_OK(ctx, StagingRing::Upload(sv, vals.data(), vals.TotalBytes(), r->stream()));
    const int rc = OP < 0 ? kv_insert(r->h(), si->dev, reinterpret_cast<const float*>(sv->dev), n, r->stream())
                          : kv_scatter_update(r->h(), si->dev, reinterpret_cast<const float*>(sv->dev), n, OP, r->stream());
    OP_REQUIRES_OK(ctx, FromKv(rc));
    OP_REQUIRES_OK(ctx, StagingRing::Release(si, r->stream()));
    OP_REQUIRES_OK(ctx, StagingRing::Release(sv, r->stream()));
    r->token = 0;
  }
};
#define KV_REGISTER_SCATTER(NAME, OP)                                                                               \
  REGISTER_KERNEL_BUILDER(Name(NAME).Device(DEVICE_CPU).HostMemory("table_handle").TypeConstraint<int32>("Tindices") \
                              .TypeConstraint<float>("dtype"), KvScatterHipOp<OP>);                                   \
  REGISTER_KERNEL_BUILDER(Name(NAME).Device(DEVICE_CPU).HostMemory("table_handle").TypeConstraint<int64_t>("Tindices") \
                              .TypeConstraint<float>("dtype"), KvScatterHipOp<OP>);                                   \
  REGISTER_KERNEL_BUILDER(Name(NAME).Device(DEVICE_CPU).HostMemory("table_handle").TypeConstraint<uint64>("Tindices") \
                              .TypeConstraint<float>("dtype"), KvScatterHipOp<OP>)
KV_REGISTER_SCATTER("KvVariableInsertV2", -1);

REGISTER_OP("KvVariableScatterAddV2")
    .Input("table_handle: resource")
    .Input("indices: Tindices")
    .Input("updates: dtype")
    .Attr("dtype: numbertype")
    .Attr("Tindices: {int32, int64, uint64, string}")
    .SetShapeFn([](InferenceContext*) { return OkStatus(); });
KV_REGISTER_SCATTER("KvVariableScatterAddV2", KV_SCATTER_ADD);

REGISTER_OP("KvVariableScatterSubV2")
    .Input("table_handle: resource")
    .Input("indices: Tindices")
    .Input("updates: dtype")
    .Attr("dtype: numbertype")
    .Attr("Tindices: {int32, int64, uint64, string}")
    .SetShapeFn([](InferenceContext*) { return OkStatus(); });
KV_REGISTER_SCATTER("KvVariableScatterSubV2", KV_SCATTER_SUB);

REGISTER_OP("KvVariableScatterMulV2")
    .Input("table_handle: resource")
    .Input("indices: Tindices")
    .Input("updates: dtype")
    .Attr("dtype: numbertype")
    .Attr("Tindices: {int32, int64, uint64, string}")
    .SetShapeFn([](InferenceContext*) { return OkStatus(); });
KV_REGISTER_SCATTER("KvVariableScatterMulV2", KV_SCATTER_MUL);

REGISTER_OP("KvVariableScatterDivV2")
    .Input("table_handle: resource")
    .Input("indices: Tindices")
    .Input("updates: dtype")
    .Attr("dtype: numbertype")
    .Attr("Tindices: {int32, int64, uint64, string}")
    .SetShapeFn([](InferenceContext*) { return OkStatus(); });
KV_REGISTER_SCATTER("KvVariableScatterDivV2", KV_SCATTER_DIV);

REGISTER_OP("KvVariableScatterMinV2")
    .Input("table_handle: resource")
    .Input("indices: Tindices")
    .Input("updates: dtype")
    .Attr("dtype: numbertype")
    .Attr("Tindices: {int32, int64, uint64, string}")
    .SetShapeFn([](InferenceContext*) { return OkStatus(); });
KV_REGISTER_SCATTER("KvVariableScatterMinV2", KV_SCATTER_MIN);

REGISTER_OP("KvVariableScatterMaxV2")
    .Input("table_handle: resource")
    .Input("indices: Tindices")
    .Input("updates: dtype")
    .Attr("dtype: numbertype")
    .Attr("Tindices: {int32, int64, uint64, string}")
    .SetShapeFn([](InferenceContext*) { return OkStatus(); });
KV_REGISTER_SCATTER("KvVariableScatterMaxV2", KV_SCATTER_MAX);

REGISTER_OP("KvVariableScatterUpdateV2")
    .Input("table_handle: resource")
    .Input("indices: Tindices")
    .Input("updates: dtype")
    .Attr("dtype: type")
    .Attr("Tindices: {int32, int64, uint64, string}")
    .SetShapeFn([](InferenceContext*) { return OkStatus(); });
KV_REGISTER_SCATTER("KvVariableScatterUpdateV2", KV_SCATTER_ASSIGN);

// ---- KvVariableGroupSparseApplyAdamV3 / V4 : ops/training_ops.cc:1086-1105, 1266-1285 ---------------------
REGISTER_OP("KvVariableGroupSparseApplyAdamV3")
    .Input("var: resource")
    .Input("m_v_linear: resource")
    .Input("grad: T")
    .Input("indices: Tindices")
    .Input("lr: T")
    .Input("beta1_power: T")
    .Input("beta2_power: T")
    .Input("beat1: T")
    .Input("beta2: T")
    .Input("epsilon: T")
    .Input("l1: T")
    .Input("l2: T")
    .Input("l21: T")
    .Attr("T: numbertype")
    .Attr("Tindices: {int32, int64, uint64, string}")
    .Attr("use_locking: bool = false")
    .SetShapeFn(shape_inference::NoOutputs);

REGISTER_OP("KvVariableGroupSparseApplyAdamV4")
    .Input("var: resource")
    .Input("m_v_linear: resource")
    .Input("grad: T")
    .Input("indices: Tindices")
    .Input("lr: T")
    .Input("beta1_power: T")
    .Input("beta2_power: T")
    .Input("beat1: T")
    .Input("beta2: T")
    .Input("epsilon: T")
    .Input("l1: T")
    .Input("l2: T")
    .Input("l21: T")
    .Attr("T: numbertype")
    .Attr("Tindices: {int32, int64, uint64, string}")
    .Attr("use_locking: bool = false")
    .SetShapeFn(shape_inference::NoOutputs);

// Are the `indices` of this optimizer node unique?  In an UNCHANGED reference graph they are: the processor patch
// (python/ops/variable_scope.py:1096-1106) sends the gradient through TF-core's _deduplicate_indexed_slices, whose
// array_ops.unique produces the node ".../Unique" whose output 0 feeds input `input` here.  Then the op is ONE launch
// (kv_apply_*_unique, include/kvhip.h).  The test is the producer's name and output slot (unique_input.h: the exact leaf
// Unique / UniqueV2 with TensorFlow's _<n> suffix, slot 0 only — ":1" is the inverse index vector) — a heuristic; the
// promise it makes is also guarded on the device (an id listed twice raises the table's error word: the next op on the
// table fails with InvalidArgument).  TFPLUS_KV_UNIQUE_INDICES=0 never takes that path, =1 always does.
static bool IndicesComeFromUnique(const NodeDef& def, int input) {
  const char* e = std::getenv("TFPLUS_KV_UNIQUE_INDICES");
  if (e && e[0] == '0') return false;
  if (e && e[0] == '1') return true;
  if (input >= def.input_size()) return false;
  return kv_shim::InputIsUniqueValues(def.input(input));
}

// gradient + indices of an optimizer op on the var's ring; the var's stream carries the whole op
struct GradIds {
  StagingRing::Slot *sg = nullptr, *si = nullptr;
  int64_t n = 0;
  kv_batch_token_t token = 0;
};
static Status StageGradIds(OpKernelContext* ctx, KvHipResource* var, const Tensor& grad, const Tensor& ids, GradIds* g) {
  if (!TensorShapeUtils::IsVector(ids.shape())) return errors::InvalidArgument("indices must be one-dimensional");
  TF_RETURN_IF_ERROR(KeyTypeMatches(var, ids));
  if (grad.dims() < 1 || grad.dim_size(0) != ids.dim_size(0))
    return errors::InvalidArgument("grad must be the same size as indices in the first dimension.");
  if (grad.NumElements() != ids.dim_size(0) * var->dim())
    return errors::InvalidArgument("var and grad must match in dimension 1");
  g->n = ids.dim_size(0);
  if (g->n == 0) return OkStatus();
  TF_RETURN_IF_ERROR(var->ring()->Acquire(grad.TotalBytes(), &g->sg));
  TF_RETURN_IF_ERROR(var->ring()->Acquire(ids.TotalBytes(), &g->si));
  TF_RETURN_IF_ERROR(StagingRing::Upload(g->sg, grad.data(), grad.TotalBytes(), var->stream()));
  TF_RETURN_IF_ERROR(StagingRing::Upload(g->si, ids.data(), ids.TotalBytes(), var->stream()));
  // the batch token of the forward lookup, when these are the very ids it saw (same length and content); TF-core's
  // _deduplicate_indexed_slices hands over unique ids, which never match: those take the general path
  if (var->token != 0 && var->token_n == g->n && var->token_sum == Fingerprint(ids.data(), ids.TotalBytes()))
    g->token = var->token;
  return OkStatus();
}
static Status ReleaseGradIds(KvHipResource* var, GradIds* g) {
  if (g->sg) TF_RETURN_IF_ERROR(StagingRing::Release(g->sg, var->stream()));
  if (g->si) TF_RETURN_IF_ERROR(StagingRing::Release(g->si, var->stream()));
  return OkStatus();
}
// the slot resources' own streams must see the work the var's stream was given, and vice versa: the C ABI
// orders ops of one table across streams itself (kvhip.h), so nothing to do here beyond using var's stream

template <int VERSION>
class KvGroupAdamHipOp : public OpKernel {
 public:
  explicit KvGroupAdamHipOp(OpKernelConstruction* c) : OpKernel(c), unique_(IndicesComeFromUnique(c->def(), 3)) {}
  void Compute(OpKernelContext* ctx) override {
    KV_RESOURCE(ctx, 0, var);
    KV_RESOURCE(ctx, 1, slot);
    for (int i = 4; i <= 12; ++i)   // kernels/training_ops.cc:7034-7068
      OP_REQUIRES(ctx, TensorShapeUtils::IsScalar(ctx->input(i).shape()),
                  errors::InvalidArgument("input ", i, " is not a scalar: ", ctx->input(i).shape().DebugString()));
    auto f = [&](int i) { return ctx->input(i).scalar<float>()(); };
    std::lock_guard<std::mutex> l(*var->mu());
    GradIds g;
    OP_REQUIRES_OK(ctx, StageGradIds(ctx, var, ctx->input(2), ctx->input(3), &g));
    if (g.n == 0) return;
    if (g.token == 0 && unique_)
      OP_REQUIRES_OK(ctx, FromKv(kv_apply_group_adam_unique(var->h(), slot->h(), reinterpret_cast<const float*>(g.sg->dev), g.si->dev,
                                                            g.n, f(4), f(5), f(6), f(7), f(8), f(9), f(10), f(11), f(12), VERSION,
                                                            var->stream())));
    else
      OP_REQUIRES_OK(ctx, FromKv(kv_apply_group_adam_tok(var->h(), slot->h(), reinterpret_cast<const float*>(g.sg->dev), g.si->dev,
                                                         g.n, f(4), f(5), f(6), f(7), f(8), f(9), f(10), f(11), f(12), VERSION,
                                                         g.token, var->stream())));
    OP_REQUIRES_OK(ctx, ReleaseGradIds(var, &g));
  }

 private:
  const bool unique_;
};
#define KV_REGISTER_APPLY(NAME, CLASS)                                                                                    \
  REGISTER_KERNEL_BUILDER(Name(NAME).Device(DEVICE_CPU).TypeConstraint<float>("T").TypeConstraint<int32>("Tindices"), CLASS);   \
  REGISTER_KERNEL_BUILDER(Name(NAME).Device(DEVICE_CPU).TypeConstraint<float>("T").TypeConstraint<int64_t>("Tindices"), CLASS); \
  REGISTER_KERNEL_BUILDER(Name(NAME).Device(DEVICE_CPU).TypeConstraint<float>("T").TypeConstraint<uint64>("Tindices"), CLASS)
KV_REGISTER_APPLY("KvVariableGroupSparseApplyAdamV3", KvGroupAdamHipOp<3>);
KV_REGISTER_APPLY("KvVariableGroupSparseApplyAdamV4", KvGroupAdamHipOp<4>);

// ---- KvVariableSparseApplyAdagrad : ops/training_ops.cc:214-226, kernels/training_ops.cc:1372-1498 ------
REGISTER_OP("KvVariableSparseApplyAdagrad")
    .Input("var: resource")
    .Input("accum: resource")
    .Input("lr: T")
    .Input("grad: T")
    .Input("indices: Tindices")
    .Attr("T: numbertype")
    .Attr("Tindices: {int32, int64, uint64, string}")
    .Attr("use_locking: bool = false")
    .Attr("update_slots: bool = true")
    .SetShapeFn(shape_inference::NoOutputs);

class KvAdagradHipOp : public OpKernel {
 public:
  explicit KvAdagradHipOp(OpKernelConstruction* c) : OpKernel(c), unique_(IndicesComeFromUnique(c->def(), 4)) {
    OP_REQUIRES_OK(c, c->GetAttr("update_slots", &update_slots_));
  }
  void Compute(OpKernelContext* ctx) override {
    KV_RESOURCE(ctx, 0, var);
    KV_RESOURCE(ctx, 1, acc);
    OP_REQUIRES(ctx, TensorShapeUtils::IsScalar(ctx->input(2).shape()),
                errors::InvalidArgument("lr is not a scalar: ", ctx->input(2).shape().DebugString()));
    std::lock_guard<std::mutex> l(*var->mu());
    GradIds g;
    OP_REQUIRES_OK(ctx, StageGradIds(ctx, var, ctx->input(3), ctx->input(4), &g));
    if (g.n == 0) return;
    if (g.token == 0 && unique_)
      OP_REQUIRES_OK(ctx, FromKv(kv_apply_adagrad_unique(var->h(), acc->h(), ctx->input(2).scalar<float>()(),
                                                         reinterpret_cast<const float*>(g.sg->dev), g.si->dev, g.n,
                                                         update_slots_ ? 1 : 0, var->stream())));
    else
      OP_REQUIRES_OK(ctx, FromKv(kv_apply_adagrad_tok(var->h(), acc->h(), ctx->input(2).scalar<float>()(),
                                                      reinterpret_cast<const float*>(g.sg->dev), g.si->dev, g.n,
                                                      update_slots_ ? 1 : 0, g.token, var->stream())));
    OP_REQUIRES_OK(ctx, ReleaseGradIds(var, &g));
  }

 private:
  const bool unique_;
  bool update_slots_ = true;
};
KV_REGISTER_APPLY("KvVariableSparseApplyAdagrad", KvAdagradHipOp);

// ---- KvVariableSparseGroupSparseApplyFtrlV2 : ops/training_ops.cc:135-150, kernels/training_ops.cc:532-801
REGISTER_OP("KvVariableSparseGroupSparseApplyFtrlV2")
    .Input("var: resource")
    .Input("accum: resource")
    .Input("linear: resource")
    .Input("grad: T")
    .Input("indices: Tindices")
    .Input("lr: T")
    .Input("l1: T")
    .Input("l2: T")
    .Input("l21: T")
    .Input("l2_shrinkage: T")
    .Input("lr_power: T")
    .Attr("T: numbertype")
    .Attr("Tindices: {int32, int64, uint64, string}")
    .Attr("use_locking: bool = false")
    .SetShapeFn(shape_inference::NoOutputs);

class KvGroupFtrlHipOp : public OpKernel {
 public:
  explicit KvGroupFtrlHipOp(OpKernelConstruction* c) : OpKernel(c), unique_(IndicesComeFromUnique(c->def(), 4)) {}
  void Compute(OpKernelContext* ctx) override {
    KV_RESOURCE(ctx, 0, var);
    KV_RESOURCE(ctx, 1, acc);
    KV_RESOURCE(ctx, 2, lin);
    for (int i = 5; i <= 10; ++i)
      OP_REQUIRES(ctx, TensorShapeUtils::IsScalar(ctx->input(i).shape()),
                  errors::InvalidArgument("input ", i, " is not a scalar: ", ctx->input(i).shape().DebugString()));
    auto f = [&](int i) { return ctx->input(i).scalar<float>()(); };
    std::lock_guard<std::mutex> l(*var->mu());
    GradIds g;
    OP_REQUIRES_OK(ctx, StageGradIds(ctx, var, ctx->input(3), ctx->input(4), &g));
    if (g.n == 0) return;
    if (g.token == 0 && unique_) {
      OP_REQUIRES_OK(ctx, FromKv(kv_apply_sparse_group_ftrl_unique(var->h(), acc->h(), lin->h(), reinterpret_cast<const float*>(g.sg->dev),
                                                                g.si->dev, g.n, f(5), f(6), f(7), f(8), f(9), f(10),
                                                                var->stream())));
    } else {
      OP_REQUIRES_OK(ctx, FromKv(kv_apply_sparse_group_ftrl_tok(var->h(), acc->h(), lin->h(), reinterpret_cast<const float*>(g.sg->dev),
                                                                g.si->dev, g.n, f(5), f(6), f(7), f(8), f(9), f(10), g.token,
                                                                var->stream())));
    }
    OP_REQUIRES_OK(ctx, ReleaseGradIds(var, &g));
  }

 private:
  const bool unique_;
};
KV_REGISTER_APPLY("KvVariableSparseGroupSparseApplyFtrlV2", KvGroupFtrlHipOp);

// =====================================================================================================================
// The ops the reference's Python layer needs to build a graph with a Saver (KvVariableSaveable,
// python/ops/kv_variable_ops.py:1225-1227,1487,1629,1737; example/dcn/train.py:548) and its table hygiene calls.
// Every kernel forwards to the C ABI call that already exists for it.
// =====================================================================================================================

// ---- KvVariableSizeV3 : ops :235-238, kernels/kv_variable_ops.cc:260-272 -> CountStorageSize
//      (table_manager.h:473-478): one entry per storage tier; there is one tier: every key of the map -------------
REGISTER_OP("KvVariableSizeV3")
    .Input("table_handle: resource")
    .Output("sizes: T")
    .Attr("T: {int32, int64} = DT_INT64");

class KvSizeV3HipOp : public OpKernel {
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    KV_RESOURCE(ctx, 0, r);
    int64_t v = 0;
    OP_REQUIRES_OK(ctx, FromKv(kv_map_size(r->h(), &v, r->stream())));
    Tensor* out = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, TensorShape({1}), &out));
    out->flat<int64_t>()(0) = v;   // the reference fills an int64 tensor whatever T says (kv_variable.h:576-583)
  }
};
REGISTER_KERNEL_BUILDER(Name("KvVariableSizeV3").Device(DEVICE_CPU), KvSizeV3HipOp);

// ---- KvVariableGetCountV2 / KvVariableGetTimeStamp : ops :349-358, 687-697 -> kv_variable.h:503-561 ----------------
REGISTER_OP("KvVariableGetCountV2")
    .Input("table_handle: resource")
    .Input("indices: Tindices")
    .Output("output: dtype")
    .Attr("dtype: {int32} = DT_INT32")
    .Attr("Tindices: {int32, int64, uint64, string}")
    .SetShapeFn(UnknownOutput);

REGISTER_OP("KvVariableGetTimeStamp")
    .Input("table_handle: resource")
    .Input("indices: Tindices")
    .Output("output: dtype")
    .Attr("dtype: {uint32} = DT_UINT32")
    .Attr("Tindices: {int32, int64, uint64, string}")
    .SetShapeFn(UnknownOutput);

template <bool STAMP>
class KvCountHipOp : public OpKernel {
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    KV_RESOURCE(ctx, 0, r);
    const Tensor& ids = ctx->input(1);
    OP_REQUIRES_OK(ctx, KeyTypeMatches(r, ids));
    Tensor* out = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, ids.shape(), &out));
    const int64_t n = ids.NumElements();
    if (n == 0) return;
    std::lock_guard<std::mutex> l(*r->mu());
    StagingRing::Slot *si = nullptr, *so = nullptr;
    OP_REQUIRES_OK(ctx, r->ring()->Acquire(ids.TotalBytes(), &si));
    OP_REQUIRES_OK(ctx, r->ring()->Acquire(out->TotalBytes(), &so));
    OP_REQUIRES_OK(ctx, StagingRing::Upload(si, ids.data(), ids.TotalBytes(), r->stream()));
    OP_REQUIRES_OK(ctx, FromKv(STAMP ? kv_get_timestamp(r->h(), si->dev, n, reinterpret_cast<uint32_t*>(so->dev), r->stream())
                                     : kv_get_count(r->h(), si->dev, n, reinterpret_cast<int32_t*>(so->dev), r->stream())));
    OP_REQUIRES_OK(ctx, StagingRing::Download(so, out->data(), out->TotalBytes(), r->stream()));
    OP_REQUIRES_OK(ctx, StagingRing::Release(si, r->stream()));
    OP_REQUIRES_OK(ctx, StagingRing::Release(so, r->stream()));
  }
};
#define KV_REGISTER_IDS_OP(NAME, ...)                                                                                      \
  REGISTER_KERNEL_BUILDER(Name(NAME).Device(DEVICE_CPU).HostMemory("table_handle").TypeConstraint<int32>("Tindices"), __VA_ARGS__);   \
  REGISTER_KERNEL_BUILDER(Name(NAME).Device(DEVICE_CPU).HostMemory("table_handle").TypeConstraint<int64_t>("Tindices"), __VA_ARGS__); \
  REGISTER_KERNEL_BUILDER(Name(NAME).Device(DEVICE_CPU).HostMemory("table_handle").TypeConstraint<uint64>("Tindices"), __VA_ARGS__)
KV_REGISTER_IDS_OP("KvVariableGetCountV2", KvCountHipOp<false>);
KV_REGISTER_IDS_OP("KvVariableGetTimeStamp", KvCountHipOp<true>);

// ---- KvVariableIncreaseCountV2 : ops :342-347; the reference's kernel does nothing (kernels/kv_variable_ops.cc:749-764)
REGISTER_OP("KvVariableIncreaseCountV2")
    .Input("table_handle: resource")
    .Input("indices: Tindices")
    .Input("counts: int32")
    .Attr("Tindices: {int32, int64, uint64, string}");

class KvIncreaseCountHipOp : public OpKernel {
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    KV_RESOURCE(ctx, 0, r);   // the handle must name a table; nothing else happens, as in the reference
    (void)r;
  }
};
KV_REGISTER_IDS_OP("KvVariableIncreaseCountV2", KvIncreaseCountHipOp);

// ---- KvVariableDelete / KvVariableDeleteWithTimestamp : ops :681-685, 698-706 -> kv_variable.h:737-789 -------------
REGISTER_OP("KvVariableDelete")
    .Input("table_handle: resource")
    .Input("indices: Tindices")
    .Attr("Tindices: {int32, int64, uint64, string}");

class KvDeleteHipOp : public OpKernel {
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    KV_RESOURCE(ctx, 0, r);
    const Tensor& ids = ctx->input(1);
    OP_REQUIRES_OK(ctx, KeyTypeMatches(r, ids));
    const int64_t n = ids.NumElements();
    if (n == 0) return;
    std::lock_guard<std::mutex> l(*r->mu());
    r->token = 0; r->token_keep = Tensor();   // rows go away: no index of an earlier batch may be taken over
    StagingRing::Slot* si = nullptr;
    OP_REQUIRES_OK(ctx, r->ring()->Acquire(ids.TotalBytes(), &si));
    OP_REQUIRES_OK(ctx, StagingRing::Upload(si, ids.data(), ids.TotalBytes(), r->stream()));
    OP_REQUIRES_OK(ctx, FromKv(kv_delete(r->h(), si->dev, n, nullptr, r->stream())));
    OP_REQUIRES_OK(ctx, StagingRing::Release(si, r->stream()));
  }
};
KV_REGISTER_IDS_OP("KvVariableDelete", KvDeleteHipOp);

REGISTER_OP("KvVariableDeleteWithTimestamp")
    .Input("table_handle: resource")
    .Output("delete_keys: Tkeys")
    .Attr("Tkeys: type")
    .Attr("threshold: int=7");

class KvDeleteWithTimestampHipOp : public OpKernel {
 public:
  explicit KvDeleteWithTimestampHipOp(OpKernelConstruction* c) : OpKernel(c) { OP_REQUIRES_OK(c, c->GetAttr("threshold", &threshold_)); }
  void Compute(OpKernelContext* ctx) override {
    KV_RESOURCE(ctx, 0, r);
    OP_REQUIRES(ctx, r->key_dtype() != DT_INT32, errors::Unimplemented("KvVariableDeleteWithTimestamp with int32 keys"));
    std::lock_guard<std::mutex> l(*r->mu());
    r->token = 0;
    int64_t count = 0;   // dry run: how many keys are old enough — sizes the output (kv_variable.h:757-789)
    OP_REQUIRES_OK(ctx, FromKv(kv_delete_with_timestamp(r->h(), threshold_, 1, nullptr, &count, r->stream())));
    Tensor* out = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, TensorShape({count}), &out));
    if (count == 0) return;
    StagingRing::Slot* sk = nullptr;
    OP_REQUIRES_OK(ctx, r->ring()->Acquire(out->TotalBytes(), &sk));
    OP_REQUIRES_OK(ctx, FromKv(kv_delete_with_timestamp(r->h(), threshold_, 0, reinterpret_cast<int64_t*>(sk->dev), &count, r->stream())));
    OP_REQUIRES_OK(ctx, StagingRing::Download(sk, out->data(), out->TotalBytes(), r->stream()));
    OP_REQUIRES_OK(ctx, StagingRing::Release(sk, r->stream()));
  }

 private:
  int threshold_ = 7;
};
REGISTER_KERNEL_BUILDER(Name("KvVariableDeleteWithTimestamp").Device(DEVICE_CPU), KvDeleteWithTimestampHipOp);

// ---- BatchKvVariableGatherOrZerosV2 : ops :297-308, kernels/kv_variable_ops.cc:431-470 (the reference loops over the
//      tables; kv_batch_gather_or_zeros covers them with one launch).  Inputs: N handles, then N index tensors. --------
REGISTER_OP("BatchKvVariableGatherOrZerosV2")
    .Input("table_handles: N * resource")
    .Input("indices: N * Tindices")
    .Output("output: N * dtype")
    .Attr("N: int >= 1")
    .Attr("dtype: type")
    .Attr("Tindices: {int32, int64, uint64, string}")
    .SetShapeFn(UnknownOutput);

class KvBatchGatherHipOp : public OpKernel {
 public:
  explicit KvBatchGatherHipOp(OpKernelConstruction* c) : OpKernel(c) { OP_REQUIRES_OK(c, c->GetAttr("N", &n_)); }
  void Compute(OpKernelContext* ctx) override {
    std::vector<KvHipResource*> rs((size_t)n_, nullptr);
    std::vector<core::ScopedUnref> unref;
    unref.reserve((size_t)n_);
    for (int i = 0; i < n_; ++i) {
      OP_REQUIRES_OK(ctx, LookupResource(ctx, HandleFromInput(ctx, i), &rs[(size_t)i]));
      unref.emplace_back(rs[(size_t)i]);
    }
    // every table's ids and rows travel through the FIRST table's stream (one launch serves them all); each
    // resource's ring stages its own tensors
    // (the library orders the read behind every table's own last op whatever its stream: kv_batch_gather_or_zeros
    //  hands each table over to `st`).  Every resource's mutex is held, in address order, until the rows are back:
    // a concurrent op on one of the resources cannot wrap its staging ring onto a slot this op still uses.
    hipStream_t st = rs[0]->stream();
    std::vector<std::mutex*> mus;
    for (KvHipResource* r : rs) mus.push_back(r->mu());
    std::sort(mus.begin(), mus.end());
    mus.erase(std::unique(mus.begin(), mus.end()), mus.end());
    for (std::mutex* m : mus) m->lock();
    struct UnlockAll { std::vector<std::mutex*>* v; ~UnlockAll() { for (auto it = v->rbegin(); it != v->rend(); ++it) (*it)->unlock(); } } unlock_all{&mus};
    std::vector<kv_handle_t> hs((size_t)n_);
    std::vector<const void*> idp((size_t)n_);
    std::vector<int64_t> ns((size_t)n_);
    std::vector<float*> outp((size_t)n_);
    std::vector<StagingRing::Slot*> si((size_t)n_, nullptr), so((size_t)n_, nullptr);
    std::vector<Tensor*> outs((size_t)n_, nullptr);
    for (int i = 0; i < n_; ++i) {
      KvHipResource* r = rs[(size_t)i];
      const Tensor& ids = ctx->input(n_ + i);
      OP_REQUIRES_OK(ctx, KeyTypeMatches(r, ids));
      TensorShape shape = ids.shape();
      shape.AddDim(r->dim());
      OP_REQUIRES_OK(ctx, ctx->allocate_output(i, shape, &outs[(size_t)i]));
      hs[(size_t)i] = r->h();
      ns[(size_t)i] = ids.NumElements();
      if (ns[(size_t)i] == 0) continue;
      OP_REQUIRES_OK(ctx, r->ring()->Acquire(ids.TotalBytes(), &si[(size_t)i]));
      OP_REQUIRES_OK(ctx, r->ring()->Acquire(outs[(size_t)i]->TotalBytes(), &so[(size_t)i]));
      OP_REQUIRES_OK(ctx, StagingRing::Upload(si[(size_t)i], ids.data(), ids.TotalBytes(), st));
      idp[(size_t)i] = si[(size_t)i]->dev;
      outp[(size_t)i] = reinterpret_cast<float*>(so[(size_t)i]->dev);
    }
    OP_REQUIRES_OK(ctx, FromKv(kv_batch_gather_or_zeros(n_, hs.data(), idp.data(), ns.data(), outp.data(), st)));
    for (int i = 0; i < n_; ++i) {
      if (ns[(size_t)i] == 0) continue;
      OP_REQUIRES_OK(ctx, StagingRing::Download(so[(size_t)i], outs[(size_t)i]->data(), outs[(size_t)i]->TotalBytes(), st));
      OP_REQUIRES_OK(ctx, StagingRing::Release(si[(size_t)i], st));
      OP_REQUIRES_OK(ctx, StagingRing::Release(so[(size_t)i], st));
    }
  }

 private:
  int n_ = 1;
};
// (this op's resource input is the list "table_handles": no HostMemory("table_handle") — the reference registers the
//  kernel without one, kernels/kv_variable_ops.cc:475-480; on DEVICE_CPU every input is host memory anyway)
REGISTER_KERNEL_BUILDER(Name("BatchKvVariableGatherOrZerosV2").Device(DEVICE_CPU).TypeConstraint<int32>("Tindices"), KvBatchGatherHipOp);
REGISTER_KERNEL_BUILDER(Name("BatchKvVariableGatherOrZerosV2").Device(DEVICE_CPU).TypeConstraint<int64_t>("Tindices"), KvBatchGatherHipOp);
REGISTER_KERNEL_BUILDER(Name("BatchKvVariableGatherOrZerosV2").Device(DEVICE_CPU).TypeConstraint<uint64>("Tindices"), KvBatchGatherHipOp);

// ---- KvVariableExport / KvVariableFullOrDeltaExport : ops :421-447, 633-665, kernels/kv_variable_ops.cc:990-1017,
//      1064-1095 -> ExportValues / DeltaExport (dynamic_save.hpp:47-195, 198-451).  Outputs: keys, values, init_table,
//      blacklist, freq_keys, freq_values (uint16 counts for Export, uint32 words for FullOrDelta), and for the latter
//      need_full_import, delete_keys.  enable_cutoff re-thresholds the rows before the export; not offered here. ------
REGISTER_OP("KvVariableExport")
    .Input("table_handle: resource")
    .Output("keys: Tkeys")
    .Output("values: Tvalues")
    .Output("init_table: Tvalues")
    .Output("blacklist: Tkeys")
    .Output("freq_keys: Tkeys")
    .Output("freq_values: uint16")
    .Attr("Tkeys: type")
    .Attr("Tvalues: type")
    .Attr("enable_cutoff: bool = false")
    .Attr("cutoff_value: float = 0.0")
    .Attr("first_n: int=3");

REGISTER_OP("KvVariableFullOrDeltaExport")
    .Input("table_handle: resource")
    .Input("do_full_export: bool")
    .Output("keys: Tkeys")
    .Output("values: Tvalues")
    .Output("init_table: Tvalues")
    .Output("blacklist: Tkeys")
    .Output("freq_keys: Tkeys")
    .Output("freq_values: uint32")
    .Output("need_full_import: bool")
    .Output("delete_keys: Tkeys")
    .Attr("Tkeys: type")
    .Attr("Tvalues: type")
    .Attr("enable_cutoff: bool = false")
    .Attr("cutoff_value: float = 0.0")
    .Attr("first_n: int=3");

template <bool FULL_OR_DELTA>
class KvExportHipOp : public OpKernel {
 public:
  explicit KvExportHipOp(OpKernelConstruction* c) : OpKernel(c) {
    OP_REQUIRES_OK(c, c->GetAttr("first_n", &first_n_));
    OP_REQUIRES_OK(c, c->GetAttr("enable_cutoff", &enable_cutoff_));
  }
  void Compute(OpKernelContext* ctx) override {
    KV_RESOURCE(ctx, 0, r);
    OP_REQUIRES(ctx, r->key_dtype() != DT_INT32, errors::Unimplemented("export with int32 keys"));
    OP_REQUIRES(ctx, !enable_cutoff_, errors::Unimplemented("export with enable_cutoff = true"));
    bool full = true;
    if (FULL_OR_DELTA) full = ctx->input(1).flat<bool>()(0);
    std::lock_guard<std::mutex> l(*r->mu());
    int64_t counts[4] = {0, 0, 0, 0};
    if (full) OP_REQUIRES_OK(ctx, FromKv(kv_export_count(r->h(), first_n_, counts, r->stream())));
    else OP_REQUIRES_OK(ctx, FromKv(kv_export_delta_count(r->h(), first_n_, counts, r->stream())));
    Tensor *keys = nullptr, *values = nullptr, *init = nullptr, *black = nullptr, *fk = nullptr, *fv = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, TensorShape({counts[0]}), &keys));
    OP_REQUIRES_OK(ctx, ctx->allocate_output(1, TensorShape({counts[0], r->dim()}), &values));
    // the init table travels only when first_n > 3 (FIRST_N_EXPORT_BLACK_LIST), else an empty [0, dim] one (dynamic_save.hpp:104-115)
    const int64_t init_out_rows = first_n_ > 3 ? r->init_rows : 0;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(2, TensorShape({init_out_rows, r->dim()}), &init));
    OP_REQUIRES_OK(ctx, ctx->allocate_output(3, TensorShape({counts[1]}), &black));
    OP_REQUIRES_OK(ctx, ctx->allocate_output(4, TensorShape({counts[2]}), &fk));
    OP_REQUIRES_OK(ctx, ctx->allocate_output(5, TensorShape({counts[2]}), &fv));
    Tensor *need = nullptr, *del = nullptr;
    if (FULL_OR_DELTA) {
      OP_REQUIRES_OK(ctx, ctx->allocate_output(6, TensorShape({1}), &need));   // shape {1}, dynamic_save.hpp:35-38
      OP_REQUIRES_OK(ctx, ctx->allocate_output(7, TensorShape({counts[3]}), &del));
      need->flat<bool>()(0) = full;
    }
    if (init_out_rows) std::memcpy(init->data(), r->init_host.data(), r->init_host.size() * sizeof(float));
    StagingRing::Slot *sk = nullptr, *sv = nullptr, *sb = nullptr, *sfk = nullptr, *sfv = nullptr, *sd = nullptr;
    auto want = [&](int64_t m, size_t elem, StagingRing::Slot** sl) -> Status {
      return m > 0 ? r->ring()->Acquire((size_t)m * elem, sl) : OkStatus();
    };
    OP_REQUIRES_OK(ctx, want(counts[0], 8, &sk));
    OP_REQUIRES_OK(ctx, want(counts[0] * r->dim(), 4, &sv));
    OP_REQUIRES_OK(ctx, want(counts[1], 8, &sb));
    OP_REQUIRES_OK(ctx, want(counts[2], 8, &sfk));
    OP_REQUIRES_OK(ctx, want(counts[2], 4, &sfv));     // the library hands out the 32-bit frequency words
    OP_REQUIRES_OK(ctx, want(counts[3], 8, &sd));
    auto dp = [](StagingRing::Slot* sl) -> void* { return sl ? sl->dev : nullptr; };
    // the table cannot change between count and fill: this resource's mutex is held and every op that writes the
    // table goes through a kernel of this file
    if (full)
      OP_REQUIRES_OK(ctx, FromKv(kv_export_fill(r->h(), first_n_, static_cast<int64_t*>(dp(sk)), static_cast<float*>(dp(sv)),
                                                static_cast<int64_t*>(dp(sb)), static_cast<int64_t*>(dp(sfk)),
                                                static_cast<uint32_t*>(dp(sfv)), r->stream())));
    else
      OP_REQUIRES_OK(ctx, FromKv(kv_export_delta_fill(r->h(), first_n_, static_cast<int64_t*>(dp(sk)), static_cast<float*>(dp(sv)),
                                                      static_cast<int64_t*>(dp(sb)), static_cast<int64_t*>(dp(sfk)),
                                                      static_cast<uint32_t*>(dp(sfv)), static_cast<int64_t*>(dp(sd)), r->stream())));
    if (sk) OP_REQUIRES_OK(ctx, StagingRing::Download(sk, keys->data(), keys->TotalBytes(), r->stream()));
    if (sv) OP_REQUIRES_OK(ctx, StagingRing::Download(sv, values->data(), values->TotalBytes(), r->stream()));
    if (sb) OP_REQUIRES_OK(ctx, StagingRing::Download(sb, black->data(), black->TotalBytes(), r->stream()));
    if (sfk) OP_REQUIRES_OK(ctx, StagingRing::Download(sfk, fk->data(), fk->TotalBytes(), r->stream()));
    if (sfv) {
      if (FULL_OR_DELTA) {
        OP_REQUIRES_OK(ctx, StagingRing::Download(sfv, fv->data(), fv->TotalBytes(), r->stream()));
      } else {
        // KvVariableExport's freq_values are uint16 counts: the low half of the word (dynamic_save.hpp:160-166)
        std::vector<uint32_t> w32((size_t)counts[2]);
        OP_REQUIRES_OK(ctx, StagingRing::Download(sfv, w32.data(), w32.size() * 4, r->stream()));
        uint16* dst = fv->flat<uint16>().data();
        for (size_t i = 0; i < w32.size(); ++i) dst[i] = static_cast<uint16>(w32[i] & 0xFFFFu);
      }
    }
    if (sd) OP_REQUIRES_OK(ctx, StagingRing::Download(sd, del->data(), del->TotalBytes(), r->stream()));
    for (StagingRing::Slot* sl : {sk, sv, sb, sfk, sfv, sd})
      if (sl) OP_REQUIRES_OK(ctx, StagingRing::Release(sl, r->stream()));
  }

 private:
  int first_n_ = 3;
  bool enable_cutoff_ = false;
};
REGISTER_KERNEL_BUILDER(Name("KvVariableExport").Device(DEVICE_CPU), KvExportHipOp<false>);
REGISTER_KERNEL_BUILDER(Name("KvVariableFullOrDeltaExport").Device(DEVICE_CPU), KvExportHipOp<true>);

// ---- KvVariableImport / KvVariableFullOrDeltaImport[V2] : ops :361-378, 576-631, kernels/kv_variable_ops.cc:779-988
//      -> ImportValues / DeltaImport (dynamic_restore.hpp:29-195).  Inputs: keys, values, init_table, blacklist,
//      freq_keys, freq_values (uint16 | uint32) [, need_full_import, delete_keys [, is_loading_finished]] -----------
REGISTER_OP("KvVariableImport")
    .Input("table_handle: resource")
    .Input("keys: Tin")
    .Input("values: Tout")
    .Input("init_table: Tout")
    .Input("blacklist: Tin")
    .Input("freq_keys: Tin")
    .Input("freq_values: uint16")
    .Attr("Tin: type")
    .Attr("Tout: type")
    .Attr("first_n: int=6");

REGISTER_OP("KvVariableFullOrDeltaImport")
    .Input("table_handle: resource")
    .Input("keys: Tin")
    .Input("values: Tout")
    .Input("init_table: Tout")
    .Input("blacklist: Tin")
    .Input("freq_keys: Tin")
    .Input("freq_values: uint32")
    .Input("need_full_import: bool")
    .Input("delete_keys: Tin")
    .Attr("Tin: type")
    .Attr("Tout: type")
    .Attr("first_n: int=6");

REGISTER_OP("KvVariableFullOrDeltaImportV2")
    .Input("table_handle: resource")
    .Input("keys: Tin")
    .Input("values: Tout")
    .Input("init_table: Tout")
    .Input("blacklist: Tin")
    .Input("freq_keys: Tin")
    .Input("freq_values: uint32")
    .Input("need_full_import: bool")
    .Input("delete_keys: Tin")
    .Input("is_loading_finished: bool")
    .Attr("Tin: type")
    .Attr("Tout: type")
    .Attr("first_n: int=6");

// KIND 0 = Import (uint16 counts, always full), 1 = FullOrDeltaImport, 2 = ...V2 (is_loading_finished is the
// reference's hint to its own loader threads; nothing to do with it here)
template <int KIND>
class KvImportHipOp : public OpKernel {
 public:
  explicit KvImportHipOp(OpKernelConstruction* c) : OpKernel(c) { OP_REQUIRES_OK(c, c->GetAttr("first_n", &first_n_)); }
  void Compute(OpKernelContext* ctx) override {
    KV_RESOURCE(ctx, 0, r);
    OP_REQUIRES(ctx, r->key_dtype() != DT_INT32, errors::Unimplemented("import with int32 keys"));
    const Tensor& keys = ctx->input(1);
    const Tensor& values = ctx->input(2);
    const Tensor& init = ctx->input(3);
    OP_REQUIRES(ctx, values.dims() == 2 && values.dim_size(0) == keys.NumElements() && values.dim_size(1) == r->dim(),
                errors::InvalidArgument("values must be [", keys.NumElements(), ", ", r->dim(), "]: ", values.shape().DebugString()));
    const bool full = KIND == 0 ? true : ctx->input(7).flat<bool>()(0);
    const Tensor* black = first_n_ > 3 ? &ctx->input(4) : nullptr;          // kernels/kv_variable_ops.cc:808-814
    const Tensor* fk = first_n_ > 4 ? &ctx->input(5) : nullptr;             // :817-824
    const Tensor* fv = first_n_ > 4 ? &ctx->input(6) : nullptr;
    const Tensor* del = KIND >= 1 ? &ctx->input(8) : nullptr;
    if (fk) OP_REQUIRES(ctx, fk->NumElements() == fv->NumElements(), errors::InvalidArgument("freq_keys and freq_values differ in length"));
    std::lock_guard<std::mutex> l(*r->mu());
    r->token = 0;
    // the init table travels with the checkpoint: a table that has none yet takes this one (dynamic_restore.hpp:160-175)
    if (init.NumElements() > 0 && init.dims() == 2 && init.dim_size(1) == r->dim()) {
      StagingRing::Slot* st = nullptr;
      OP_REQUIRES_OK(ctx, r->ring()->Acquire(init.TotalBytes(), &st));
      OP_REQUIRES_OK(ctx, StagingRing::Upload(st, init.data(), init.TotalBytes(), r->stream()));
      OP_REQUIRES_OK(ctx, FromKv(kv_init_table(r->h(), reinterpret_cast<const float*>(st->dev), init.dim_size(0), r->stream())));
      OP_REQUIRES_OK(ctx, StagingRing::Release(st, r->stream()));
      if (r->init_rows == 0) {
        const float* src = static_cast<const float*>(init.data());
        r->init_host.assign(src, src + init.NumElements());
        r->init_rows = init.dim_size(0);
      }
    }
    StagingRing::Slot *sk = nullptr, *sv = nullptr, *sb = nullptr, *sfk = nullptr, *sfv = nullptr, *sd = nullptr;
    auto up = [&](const Tensor* t, StagingRing::Slot** sl) -> Status {
      if (!t || t->NumElements() == 0) return OkStatus();
      TF_RETURN_IF_ERROR(r->ring()->Acquire(t->TotalBytes(), sl));
      return StagingRing::Upload(*sl, t->data(), t->TotalBytes(), r->stream());
    };
    OP_REQUIRES_OK(ctx, up(&keys, &sk));
    OP_REQUIRES_OK(ctx, up(&values, &sv));
    OP_REQUIRES_OK(ctx, up(black, &sb));
    OP_REQUIRES_OK(ctx, up(fk, &sfk));
    if (fv && fv->NumElements() > 0) {
      if (KIND == 0) {   // uint16 counts -> the 32-bit words the library stores (day 0)
        std::vector<uint32_t> w32((size_t)fv->NumElements());
        const uint16* src = fv->flat<uint16>().data();
        for (size_t i = 0; i < w32.size(); ++i) w32[i] = src[i];
        OP_REQUIRES_OK(ctx, r->ring()->Acquire(w32.size() * 4, &sfv));
        OP_REQUIRES_OK(ctx, StagingRing::Upload(sfv, w32.data(), w32.size() * 4, r->stream()));
      } else {
        OP_REQUIRES_OK(ctx, up(fv, &sfv));
      }
    }
    if (!full) OP_REQUIRES_OK(ctx, up(del, &sd));
    auto dp = [](StagingRing::Slot* sl) -> const void* { return sl ? sl->dev : nullptr; };
    auto cnt = [](const Tensor* t) -> int64_t { return t ? t->NumElements() : 0; };
    int rc;
    if (full)
      rc = kv_import(r->h(), static_cast<const int64_t*>(dp(sk)), static_cast<const float*>(dp(sv)), keys.NumElements(),
                     static_cast<const int64_t*>(dp(sb)), cnt(black), static_cast<const int64_t*>(dp(sfk)),
                     static_cast<const uint32_t*>(dp(sfv)), cnt(fk), r->stream());
    else
      rc = kv_import_delta(r->h(), static_cast<const int64_t*>(dp(sk)), static_cast<const float*>(dp(sv)), keys.NumElements(),
                           static_cast<const int64_t*>(dp(sb)), cnt(black), static_cast<const int64_t*>(dp(sfk)),
                           static_cast<const uint32_t*>(dp(sfv)), cnt(fk), static_cast<const int64_t*>(dp(sd)), cnt(del),
                           first_n_, r->stream());
    OP_REQUIRES_OK(ctx, FromKv(rc));
    for (StagingRing::Slot* sl : {sk, sv, sb, sfk, sfv, sd})
      if (sl) OP_REQUIRES_OK(ctx, StagingRing::Release(sl, r->stream()));
  }

 private:
  int first_n_ = 6;
};
REGISTER_KERNEL_BUILDER(Name("KvVariableImport").Device(DEVICE_CPU), KvImportHipOp<0>);
REGISTER_KERNEL_BUILDER(Name("KvVariableFullOrDeltaImport").Device(DEVICE_CPU), KvImportHipOp<1>);
REGISTER_KERNEL_BUILDER(Name("KvVariableFullOrDeltaImportV2").Device(DEVICE_CPU), KvImportHipOp<2>);

// =====================================================================================================================
// DEVICE_GPU kernels — TensorFlow-ROCm's device (DEVICE_GPU is the name of the ROCm device in a TF build with
// TENSORFLOW_USE_ROCM; nothing here is a CUDA path).  A graph whose KvVariable is placed on the GPU runs the hot path
// with NO staging: `indices`, `grad` and `output` are device tensors, their pointers go straight to the C ABI on TF's
// own compute stream (GetGpuStream), and what bench.py measures is what the graph gets.  As in the reference, the
// resource handle is host memory (kernels/kv_variable_ops.cc:540-546); so are the scalar hyper-parameters (`lr`,
// `beta1_power`, ...), which the C ABI takes by value.
//
// The batch token.  A training lookup leaves the batch's index behind (kv_gather_or_insert_tok); the optimizer op
// takes it over when it is handed THE TENSOR the lookup saw: same device buffer, same length.  In a TF1 graph that is
// the case exactly when the optimizer receives the raw IndexedSlices of the lookup's gradient (_GatherGrad returns
// `indices = reshape(ids)`: the same buffer) — INTEGRATION.md §2a has the one-line processor patch that makes it so;
// TF-core's _deduplicate_indexed_slices produces a new tensor and takes the general path.  The ids buffer is an input
// of the gradient op, so it is alive (its address is not reused) until the optimizer op has run; the token is
// dropped by every other op on the table.
//
// Every other op of this file is registered for DEVICE_GPU with all its tensor arguments in host memory, so that a
// GPU-placed variable finds a kernel for each of them (a resource is only visible to kernels of its own device); those
// run the kernel bodies above unchanged — they are savers, counters and maintenance, not the hot path.
// =====================================================================================================================
}  // namespace tfplus_hip
#include "tensorflow/core/util/gpu_kernel_helper.h"   // GetGpuStream: gpuStream_t is hipStream_t under TENSORFLOW_USE_ROCM
namespace tfplus_hip {

static hipStream_t TfStream(OpKernelContext* ctx) { return GetGpuStream(ctx); }

template <int MODE>
class KvGatherGpuOp : public OpKernel {
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    KV_RESOURCE(ctx, 0, r);
    const Tensor& ids = ctx->input(1);
    OP_REQUIRES_OK(ctx, KeyTypeMatches(r, ids));
    TensorShape shape = ids.shape();
    shape.AddDim(r->dim());
    Tensor* out = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, shape, &out));   // device memory
    const int64_t n = ids.NumElements();
    if (n == 0) return;
    const Tensor* counts = nullptr;
    if (MODE == 2) {
      counts = &ctx->input(2);
      OP_REQUIRES(ctx, counts->dtype() == DT_INT32, errors::InvalidArgument("increment count, counts dtype must be int32"));
      OP_REQUIRES(ctx, counts->shape() == ids.shape(),
                  errors::InvalidArgument("increment count, indices shape ", ids.shape().DebugString(),
                                          " does not match with counts shape ", counts->shape().DebugString()));
    }
    std::lock_guard<std::mutex> l(*r->mu());
    hipStream_t st = TfStream(ctx);
    if (MODE == 0) {
      OP_REQUIRES_OK(ctx, FromKv(kv_gather_or_zeros(r->h(), ids.data(), n, static_cast<float*>(out->data()), st)));
      return;
    }
    kv_batch_token_t tok = 0;
    OP_REQUIRES_OK(ctx, FromKv(kv_gather_or_insert_tok(r->h(), ids.data(), counts ? static_cast<const int32_t*>(counts->data()) : nullptr,
                                                       n, static_cast<float*>(out->data()), &tok, st)));
    r->token = tok; r->token_ids = ids.data(); r->token_n = n; r->token_sum = 0;
    r->token_keep = ids;   // (shares the buffer: it cannot be freed and handed to another tensor while the token lives)
  }
};
#define KV_REGISTER_GATHER_GPU(NAME, MODE)                                                                                \
  REGISTER_KERNEL_BUILDER(Name(NAME).Device(DEVICE_GPU).HostMemory("table_handle").TypeConstraint<int32>("Tindices")      \
                              .TypeConstraint<float>("dtype"), KvGatherGpuOp<MODE>);                                         \
  REGISTER_KERNEL_BUILDER(Name(NAME).Device(DEVICE_GPU).HostMemory("table_handle").TypeConstraint<int64_t>("Tindices")    \
                              .TypeConstraint<float>("dtype"), KvGatherGpuOp<MODE>);                                         \
  REGISTER_KERNEL_BUILDER(Name(NAME).Device(DEVICE_GPU).HostMemory("table_handle").TypeConstraint<uint64>("Tindices")     \
                              .TypeConstraint<float>("dtype"), KvGatherGpuOp<MODE>)
KV_REGISTER_GATHER_GPU("KvVariableGatherOrZerosV2", 0);
KV_REGISTER_GATHER_GPU("KvVariableGatherOrInsertV2", 1);
KV_REGISTER_GATHER_GPU("KvVariableGatherOrInsertWithCounts", 2);

// gradient + indices of an optimizer op, device-resident: shape checks, and the lookup's token when these are its ids
static Status DeviceGradIds(KvHipResource* var, const Tensor& grad, const Tensor& ids, int64_t* n, kv_batch_token_t* token) {
  if (!TensorShapeUtils::IsVector(ids.shape())) return errors::InvalidArgument("indices must be one-dimensional");
  TF_RETURN_IF_ERROR(KeyTypeMatches(var, ids));
  if (grad.dims() < 1 || grad.dim_size(0) != ids.dim_size(0))
    return errors::InvalidArgument("grad must be the same size as indices in the first dimension.");
  if (grad.NumElements() != ids.dim_size(0) * var->dim())
    return errors::InvalidArgument("var and grad must match in dimension 1");
  *n = ids.dim_size(0);
  *token = (var->token != 0 && var->token_n == *n && var->token_ids == ids.data()) ? var->token : 0;
  var->token = 0;   // one apply per lookup: a second optimizer op on the same ids rebuilds the index
  var->token_keep = Tensor();
  return OkStatus();
}

template <int VERSION>
class KvGroupAdamGpuOp : public OpKernel {
 public:
  explicit KvGroupAdamGpuOp(OpKernelConstruction* c) : OpKernel(c), unique_(IndicesComeFromUnique(c->def(), 3)) {}
  void Compute(OpKernelContext* ctx) override {
    KV_RESOURCE(ctx, 0, var);
    KV_RESOURCE(ctx, 1, slot);
    for (int i = 4; i <= 12; ++i)   // kernels/training_ops.cc:7034-7068
      OP_REQUIRES(ctx, TensorShapeUtils::IsScalar(ctx->input(i).shape()),
                  errors::InvalidArgument("input ", i, " is not a scalar: ", ctx->input(i).shape().DebugString()));
    auto f = [&](int i) { return ctx->input(i).scalar<float>()(); };   // host memory (registration below)
    std::lock_guard<std::mutex> l(*var->mu());
    int64_t n = 0;
    kv_batch_token_t token = 0;
    OP_REQUIRES_OK(ctx, DeviceGradIds(var, ctx->input(2), ctx->input(3), &n, &token));
    if (n == 0) return;
    if (token == 0 && unique_)
      OP_REQUIRES_OK(ctx, FromKv(kv_apply_group_adam_unique(var->h(), slot->h(), static_cast<const float*>(ctx->input(2).data()),
                                                            ctx->input(3).data(), n, f(4), f(5), f(6), f(7), f(8), f(9), f(10), f(11),
                                                            f(12), VERSION, TfStream(ctx))));
    else
      OP_REQUIRES_OK(ctx, FromKv(kv_apply_group_adam_tok(var->h(), slot->h(), static_cast<const float*>(ctx->input(2).data()),
                                                         ctx->input(3).data(), n, f(4), f(5), f(6), f(7), f(8), f(9), f(10), f(11),
                                                         f(12), VERSION, token, TfStream(ctx))));
  }

 private:
  const bool unique_;
};
#define KV_GPU_ADAM_HOST .HostMemory("var").HostMemory("m_v_linear").HostMemory("lr").HostMemory("beta1_power")           \
      .HostMemory("beta2_power").HostMemory("beat1").HostMemory("beta2").HostMemory("epsilon").HostMemory("l1")          \
      .HostMemory("l2").HostMemory("l21")
#define KV_REGISTER_APPLY_GPU(NAME, HOSTMEM, CLASS)                                                                         \
  REGISTER_KERNEL_BUILDER(Name(NAME).Device(DEVICE_GPU) HOSTMEM.TypeConstraint<float>("T").TypeConstraint<int32>("Tindices"), CLASS);   \
  REGISTER_KERNEL_BUILDER(Name(NAME).Device(DEVICE_GPU) HOSTMEM.TypeConstraint<float>("T").TypeConstraint<int64_t>("Tindices"), CLASS); \
  REGISTER_KERNEL_BUILDER(Name(NAME).Device(DEVICE_GPU) HOSTMEM.TypeConstraint<float>("T").TypeConstraint<uint64>("Tindices"), CLASS)
KV_REGISTER_APPLY_GPU("KvVariableGroupSparseApplyAdamV3", KV_GPU_ADAM_HOST, KvGroupAdamGpuOp<3>);
KV_REGISTER_APPLY_GPU("KvVariableGroupSparseApplyAdamV4", KV_GPU_ADAM_HOST, KvGroupAdamGpuOp<4>);

class KvAdagradGpuOp : public OpKernel {
 public:
  explicit KvAdagradGpuOp(OpKernelConstruction* c) : OpKernel(c), unique_(IndicesComeFromUnique(c->def(), 4)) {
    OP_REQUIRES_OK(c, c->GetAttr("update_slots", &update_slots_));
  }
  void Compute(OpKernelContext* ctx) override {
    KV_RESOURCE(ctx, 0, var);
    KV_RESOURCE(ctx, 1, acc);
    OP_REQUIRES(ctx, TensorShapeUtils::IsScalar(ctx->input(2).shape()),
                errors::InvalidArgument("lr is not a scalar: ", ctx->input(2).shape().DebugString()));
    std::lock_guard<std::mutex> l(*var->mu());
    int64_t n = 0;
    kv_batch_token_t token = 0;
    OP_REQUIRES_OK(ctx, DeviceGradIds(var, ctx->input(3), ctx->input(4), &n, &token));
    if (n == 0) return;
    if (token == 0 && unique_)
      OP_REQUIRES_OK(ctx, FromKv(kv_apply_adagrad_unique(var->h(), acc->h(), ctx->input(2).scalar<float>()(),
                                                         static_cast<const float*>(ctx->input(3).data()), ctx->input(4).data(), n,
                                                         update_slots_ ? 1 : 0, TfStream(ctx))));
    else
      OP_REQUIRES_OK(ctx, FromKv(kv_apply_adagrad_tok(var->h(), acc->h(), ctx->input(2).scalar<float>()(),
                                                      static_cast<const float*>(ctx->input(3).data()), ctx->input(4).data(), n,
                                                      update_slots_ ? 1 : 0, token, TfStream(ctx))));
  }

 private:
  const bool unique_;
  bool update_slots_ = true;
};
#define KV_GPU_ADAGRAD_HOST .HostMemory("var").HostMemory("accum").HostMemory("lr")
KV_REGISTER_APPLY_GPU("KvVariableSparseApplyAdagrad", KV_GPU_ADAGRAD_HOST, KvAdagradGpuOp);

class KvGroupFtrlGpuOp : public OpKernel {
 public:
  explicit KvGroupFtrlGpuOp(OpKernelConstruction* c) : OpKernel(c), unique_(IndicesComeFromUnique(c->def(), 4)) {}
  void Compute(OpKernelContext* ctx) override {
    KV_RESOURCE(ctx, 0, var);
    KV_RESOURCE(ctx, 1, acc);
    KV_RESOURCE(ctx, 2, lin);
    for (int i = 5; i <= 10; ++i)
      OP_REQUIRES(ctx, TensorShapeUtils::IsScalar(ctx->input(i).shape()),
                  errors::InvalidArgument("input ", i, " is not a scalar: ", ctx->input(i).shape().DebugString()));
    auto f = [&](int i) { return ctx->input(i).scalar<float>()(); };
    std::lock_guard<std::mutex> l(*var->mu());
    int64_t n = 0;
    kv_batch_token_t token = 0;
    OP_REQUIRES_OK(ctx, DeviceGradIds(var, ctx->input(3), ctx->input(4), &n, &token));
    if (n == 0) return;
    if (token == 0 && unique_) {
      OP_REQUIRES_OK(ctx, FromKv(kv_apply_sparse_group_ftrl_unique(var->h(), acc->h(), lin->h(), static_cast<const float*>(ctx->input(3).data()),
                                                                ctx->input(4).data(), n, f(5), f(6), f(7), f(8), f(9), f(10),
                                                                TfStream(ctx))));
    } else {
      OP_REQUIRES_OK(ctx, FromKv(kv_apply_sparse_group_ftrl_tok(var->h(), acc->h(), lin->h(), static_cast<const float*>(ctx->input(3).data()),
                                                                ctx->input(4).data(), n, f(5), f(6), f(7), f(8), f(9), f(10), token,
                                                                TfStream(ctx))));
    }
  }

 private:
  const bool unique_;
};
#define KV_GPU_FTRL_HOST .HostMemory("var").HostMemory("accum").HostMemory("linear").HostMemory("lr").HostMemory("l1")    \
      .HostMemory("l2").HostMemory("l21").HostMemory("l2_shrinkage").HostMemory("lr_power")
KV_REGISTER_APPLY_GPU("KvVariableSparseGroupSparseApplyFtrlV2", KV_GPU_FTRL_HOST, KvGroupFtrlGpuOp);

// ---- create / init on the GPU device: the resource lives in the GPU device's resource manager ------------------------
REGISTER_KERNEL_BUILDER(Name("KvVariable").Device(DEVICE_GPU).HostMemory("table_handle"), CreateKvVariableHipOp);
REGISTER_KERNEL_BUILDER(Name("KvVariableV2").Device(DEVICE_GPU).HostMemory("table_handle"), CreateKvVariableHipOp);
REGISTER_KERNEL_BUILDER(Name("KvVariableV3").Device(DEVICE_GPU).HostMemory("table_handle"), CreateKvVariableHipOp);
REGISTER_KERNEL_BUILDER(Name("KvVariableV4").Device(DEVICE_GPU).HostMemory("table_handle"), CreateKvVariableHipOp);

class InitKvVariableGpuOp : public OpKernel {
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    KV_RESOURCE(ctx, 0, r);
    const Tensor& t = ctx->input(1);   // device memory: the initializer ran on the GPU
    OP_REQUIRES(ctx, t.dtype() == DT_FLOAT, errors::InvalidArgument("random_initializer must be float"));
    OP_REQUIRES(ctx, t.dims() == 2 && t.dim_size(1) == r->dim(),
                errors::InvalidArgument("random_initializer must be [rows, ", r->dim(), "]"));
    std::lock_guard<std::mutex> l(*r->mu());
    hipStream_t st = TfStream(ctx);
    OP_REQUIRES_OK(ctx, FromKv(kv_init_table(r->h(), static_cast<const float*>(t.data()), t.dim_size(0), st)));
    if (r->init_rows == 0) {   // the host copy KvVariableExport hands out (once per variable: the one synchronisation here)
      r->init_host.resize(static_cast<size_t>(t.NumElements()));
      HIP_OK(ctx, hipMemcpyAsync(r->init_host.data(), t.data(), t.TotalBytes(), hipMemcpyDeviceToHost, st));
      HIP_OK(ctx, hipStreamSynchronize(st));
      r->init_rows = t.dim_size(0);
    }
  }
};
REGISTER_KERNEL_BUILDER(Name("InitKvVariableV2").Device(DEVICE_GPU).HostMemory("table_handle"), InitKvVariableGpuOp);

// ---- everything else: the kernels above, every tensor argument in host memory ----------------------------------------
#define KV_GPU_IDX3(NAME, HOSTMEM, ...)                                                                                \
  REGISTER_KERNEL_BUILDER(Name(NAME).Device(DEVICE_GPU) HOSTMEM.TypeConstraint<int32>("Tindices"), __VA_ARGS__);         \
  REGISTER_KERNEL_BUILDER(Name(NAME).Device(DEVICE_GPU) HOSTMEM.TypeConstraint<int64_t>("Tindices"), __VA_ARGS__);       \
  REGISTER_KERNEL_BUILDER(Name(NAME).Device(DEVICE_GPU) HOSTMEM.TypeConstraint<uint64>("Tindices"), __VA_ARGS__)
#define KV_H1(a) .HostMemory(a)
#define KV_H2(a, b) .HostMemory(a).HostMemory(b)
#define KV_H3(a, b, c) .HostMemory(a).HostMemory(b).HostMemory(c)
REGISTER_KERNEL_BUILDER(Name("KvVariableShapeV2").Device(DEVICE_GPU) KV_H2("table_handle", "output").TypeConstraint<int32>("out_type"), KvShapeHipOp<int32>);
REGISTER_KERNEL_BUILDER(Name("KvVariableShapeV2").Device(DEVICE_GPU) KV_H2("table_handle", "output").TypeConstraint<int64_t>("out_type"), KvShapeHipOp<int64_t>);
REGISTER_KERNEL_BUILDER(Name("KvVariableIsInitializedV2").Device(DEVICE_GPU) KV_H2("table_handle", "is_initialized"), KvIsInitializedHipOp);
REGISTER_KERNEL_BUILDER(Name("KvVariableSizeV2").Device(DEVICE_GPU) KV_H2("table_handle", "size").TypeConstraint<int32>("T"), KvSizeHipOp<int32, false>);
REGISTER_KERNEL_BUILDER(Name("KvVariableSizeV2").Device(DEVICE_GPU) KV_H2("table_handle", "size").TypeConstraint<int64_t>("T"), KvSizeHipOp<int64_t, false>);
REGISTER_KERNEL_BUILDER(Name("KvVariableFrequency").Device(DEVICE_GPU) KV_H2("table_handle", "size").TypeConstraint<int32>("T"), KvSizeHipOp<int32, true>);
REGISTER_KERNEL_BUILDER(Name("KvVariableFrequency").Device(DEVICE_GPU) KV_H2("table_handle", "size").TypeConstraint<int64_t>("T"), KvSizeHipOp<int64_t, true>);
REGISTER_KERNEL_BUILDER(Name("KvVariableSizeV3").Device(DEVICE_GPU) KV_H2("table_handle", "sizes"), KvSizeV3HipOp);
REGISTER_KERNEL_BUILDER(Name("ReadKvVariableOpV2").Device(DEVICE_GPU) KV_H3("table_handle", "keys", "values"), ReadKvVariableHipOp);
REGISTER_KERNEL_BUILDER(Name("DestroyKvVariableOpV2").Device(DEVICE_GPU) KV_H1("table_handle"), DestroyKvVariableHipOp);
#define KV_REGISTER_SCATTER_GPU(NAME, VALS, OP)                                                                           \
  REGISTER_KERNEL_BUILDER(Name(NAME).Device(DEVICE_GPU) KV_H3("table_handle", "indices", VALS).TypeConstraint<int32>("Tindices")    \
                              .TypeConstraint<float>("dtype"), KvScatterHipOp<OP>);                                       \
  REGISTER_KERNEL_BUILDER(Name(NAME).Device(DEVICE_GPU) KV_H3("table_handle", "indices", VALS).TypeConstraint<int64_t>("Tindices")  \
                              .TypeConstraint<float>("dtype"), KvScatterHipOp<OP>);                                       \
  REGISTER_KERNEL_BUILDER(Name(NAME).Device(DEVICE_GPU) KV_H3("table_handle", "indices", VALS).TypeConstraint<uint64>("Tindices")   \
                              .TypeConstraint<float>("dtype"), KvScatterHipOp<OP>)
KV_REGISTER_SCATTER_GPU("KvVariableInsertV2", "values", -1);
KV_REGISTER_SCATTER_GPU("KvVariableScatterAddV2", "updates", KV_SCATTER_ADD);
KV_REGISTER_SCATTER_GPU("KvVariableScatterSubV2", "updates", KV_SCATTER_SUB);
KV_REGISTER_SCATTER_GPU("KvVariableScatterMulV2", "updates", KV_SCATTER_MUL);
KV_REGISTER_SCATTER_GPU("KvVariableScatterDivV2", "updates", KV_SCATTER_DIV);
KV_REGISTER_SCATTER_GPU("KvVariableScatterMinV2", "updates", KV_SCATTER_MIN);
KV_REGISTER_SCATTER_GPU("KvVariableScatterMaxV2", "updates", KV_SCATTER_MAX);
KV_REGISTER_SCATTER_GPU("KvVariableScatterUpdateV2", "updates", KV_SCATTER_ASSIGN);
KV_GPU_IDX3("KvVariableGetCountV2", KV_H3("table_handle", "indices", "output"), KvCountHipOp<false>);
KV_GPU_IDX3("KvVariableGetTimeStamp", KV_H3("table_handle", "indices", "output"), KvCountHipOp<true>);
KV_GPU_IDX3("KvVariableIncreaseCountV2", KV_H3("table_handle", "indices", "counts"), KvIncreaseCountHipOp);
KV_GPU_IDX3("KvVariableDelete", KV_H2("table_handle", "indices"), KvDeleteHipOp);
REGISTER_KERNEL_BUILDER(Name("KvVariableDeleteWithTimestamp").Device(DEVICE_GPU) KV_H2("table_handle", "delete_keys"), KvDeleteWithTimestampHipOp);
KV_GPU_IDX3("BatchKvVariableGatherOrZerosV2", KV_H3("table_handles", "indices", "output"), KvBatchGatherHipOp);
#define KV_H_EXPORT .HostMemory("table_handle").HostMemory("keys").HostMemory("values").HostMemory("init_table")         \
      .HostMemory("blacklist").HostMemory("freq_keys").HostMemory("freq_values")
REGISTER_KERNEL_BUILDER(Name("KvVariableExport").Device(DEVICE_GPU) KV_H_EXPORT, KvExportHipOp<false>);
REGISTER_KERNEL_BUILDER(Name("KvVariableFullOrDeltaExport").Device(DEVICE_GPU) KV_H_EXPORT.HostMemory("do_full_export")
                            .HostMemory("need_full_import").HostMemory("delete_keys"), KvExportHipOp<true>);
REGISTER_KERNEL_BUILDER(Name("KvVariableImport").Device(DEVICE_GPU) KV_H_EXPORT, KvImportHipOp<0>);
REGISTER_KERNEL_BUILDER(Name("KvVariableFullOrDeltaImport").Device(DEVICE_GPU) KV_H_EXPORT.HostMemory("need_full_import")
                            .HostMemory("delete_keys"), KvImportHipOp<1>);
REGISTER_KERNEL_BUILDER(Name("KvVariableFullOrDeltaImportV2").Device(DEVICE_GPU) KV_H_EXPORT.HostMemory("need_full_import")
                            .HostMemory("delete_keys").HostMemory("is_loading_finished"), KvImportHipOp<2>);

}  // namespace tfplus_hip
