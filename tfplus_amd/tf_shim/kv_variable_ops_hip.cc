// kv_variable_ops_hip.cc — TensorFlow custom-op shim over libkvhip.so (include/kvhip.h).
//
// NOT built in this repository's image (no TensorFlow headers there; see INTEGRATION.md for the
// build line).  It re-registers the reference's op names with identical input order, attrs and
// shape functions (tfplus/kv_variable/ops/kv_variable_ops.cc:37-74,212-222,285-332;
// ops/training_ops.cc:135-150,214-226,1086-1105,1266-1285) and forwards each Compute() to one C
// ABI call, so `tfplus.kv_variable.python.*` works unchanged on top of it.  All semantics live
// behind the C ABI; this file only moves tensors: tensorflow-cpu keeps tensors in host memory, so
// they are staged through HBM (pinned staging would be the next step; DESIGN.md §4 gives the PCIe
// bound).  A TF build with a ROCm device would pass tensor.data() straight through instead.
#include <hip/hip_runtime.h>

#include "kvhip.h"
#include "tensorflow/core/framework/op.h"
#include "tensorflow/core/framework/op_kernel.h"
#include "tensorflow/core/framework/resource_mgr.h"
#include "tensorflow/core/framework/shape_inference.h"

namespace tfplus_hip {
using namespace tensorflow;  // NOLINT

// The resource the handle points at: owns one kv_handle_t.
class KvHipResource : public ResourceBase {
 public:
  explicit KvHipResource(kv_handle_t h, int dim) : h_(h), dim_(dim) {}
  ~KvHipResource() override { kv_destroy(h_); }
  string DebugString() const override { return "KvHipResource"; }
  kv_handle_t h() const { return h_; }
  int dim() const { return dim_; }

 private:
  kv_handle_t h_;
  int dim_;
};

static Status FromKv(int rc) {
  if (rc == KV_OK) return OkStatus();
  return Status(static_cast<tsl::error::Code>(rc), kv_last_error());
}

// device staging buffer that frees itself
struct DevBuf {
  void* p = nullptr;
  explicit DevBuf(size_t bytes) { if (bytes) hipMalloc(&p, bytes); }
  ~DevBuf() { if (p) hipFree(p); }
};

// ---- KvVariable (handle creation) : kernels/kv_variable_ops.cc:31-125 -------------------------
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
    .SetShapeFn(shape_inference::ScalarShape);

class CreateKvVariableHipOp : public OpKernel {
 public:
  explicit CreateKvVariableHipOp(OpKernelConstruction* c) : OpKernel(c) {
    OP_REQUIRES_OK(c, c->GetAttr("use_node_name_sharing", &use_node_name_sharing_));
    OP_REQUIRES_OK(c, c->GetAttr("key_dtype", &key_dtype_));
    OP_REQUIRES_OK(c, c->GetAttr("enter_threshold", &enter_threshold_));
    OP_REQUIRES_OK(c, c->GetAttr("value_shape", &value_shape_));
  }
  void Compute(OpKernelContext* ctx) override {
    mutex_lock l(mu_);
    if (!set_) {
      OP_REQUIRES_OK(ctx, cinfo_.Init(ctx->resource_manager(), def(), use_node_name_sharing_));
      KvHipResource* res = nullptr;
      const int dim = value_shape_.num_elements();
      OP_REQUIRES_OK(ctx, cinfo_.resource_manager()->LookupOrCreate<KvHipResource>(
                              cinfo_.container(), cinfo_.name(), &res, [&](KvHipResource** out) {
                                kv_handle_t h;
                                TF_RETURN_IF_ERROR(FromKv(kv_create(key_dtype_, KV_DT_FLOAT, dim,
                                                                    enter_threshold_, 0, 0, &h)));
                                *out = new KvHipResource(h, dim);
                                return OkStatus();
                              }));
      core::ScopedUnref unref(res);
      handle_ = MakeResourceHandle<KvHipResource>(ctx, cinfo_.container(), cinfo_.name());
      set_ = true;
    }
    Tensor* out;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, TensorShape({}), &out));
    out->scalar<ResourceHandle>()() = handle_;
  }

 private:
  mutex mu_;
  bool set_ = false, use_node_name_sharing_ = false;
  DataType key_dtype_;
  int enter_threshold_ = 0;
  TensorShape value_shape_;
  ContainerInfo cinfo_;
  ResourceHandle handle_;
};
REGISTER_KERNEL_BUILDER(Name("KvVariable").Device(DEVICE_CPU), CreateKvVariableHipOp);

// ---- InitKvVariableV2 : kernels/kv_variable_ops.cc:188-200 -------------------------------------
REGISTER_OP("InitKvVariableV2")
    .Input("table_handle: resource")
    .Input("random_initializer: T")
    .Attr("T: type")
    .SetShapeFn(shape_inference::NoOutputs);

class InitKvVariableHipOp : public OpKernel {
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    KvHipResource* r;
    OP_REQUIRES_OK(ctx, LookupResource(ctx, HandleFromInput(ctx, 0), &r));
    core::ScopedUnref unref(r);
    const Tensor& t = ctx->input(1);
    DevBuf d(t.TotalBytes());
    hipMemcpy(d.p, t.data(), t.TotalBytes(), hipMemcpyHostToDevice);
    OP_REQUIRES_OK(ctx, FromKv(kv_init_table(r->h(), static_cast<const float*>(d.p), t.dim_size(0), nullptr)));
    hipDeviceSynchronize();
  }
};
REGISTER_KERNEL_BUILDER(Name("InitKvVariableV2").Device(DEVICE_CPU), InitKvVariableHipOp);

// ---- KvVariableGatherOrInsertV2 / GatherOrZerosV2 : kernels/kv_variable_ops.cc:348-538 ---------
#define KV_GATHER_OP(NAME)                                                        \
  REGISTER_OP(NAME)                                                               \
      .Input("table_handle: resource")                                            \
      .Input("indices: Tindices")                                                 \
      .Output("output: dtype")                                                    \
      .Attr("dtype: type")                                                        \
      .Attr("Tindices: {int32, int64, uint64, string}")                           \
      .SetShapeFn([](shape_inference::InferenceContext* c) {                      \
        c->set_output(0, c->UnknownShape());                                      \
        return OkStatus();                                                        \
      })
KV_GATHER_OP("KvVariableGatherOrInsertV2");
KV_GATHER_OP("KvVariableGatherOrZerosV2");

template <bool INSERT>
class KvGatherHipOp : public OpKernel {
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    KvHipResource* r;
    OP_REQUIRES_OK(ctx, LookupResource(ctx, HandleFromInput(ctx, 0), &r));
    core::ScopedUnref unref(r);
    const Tensor& ids = ctx->input(1);
    TensorShape shape = ids.shape();
    shape.AddDim(r->dim());
    Tensor* out;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, shape, &out));
    const int64_t n = ids.NumElements();
    if (n == 0) return;
    DevBuf d_ids(ids.TotalBytes()), d_out(out->TotalBytes());
    hipMemcpy(d_ids.p, ids.data(), ids.TotalBytes(), hipMemcpyHostToDevice);
    const int rc = INSERT ? kv_gather_or_insert(r->h(), d_ids.p, nullptr, n, static_cast<float*>(d_out.p), nullptr)
                          : kv_gather_or_zeros(r->h(), d_ids.p, n, static_cast<float*>(d_out.p), nullptr);
    OP_REQUIRES_OK(ctx, FromKv(rc));
    hipMemcpy(out->data(), d_out.p, out->TotalBytes(), hipMemcpyDeviceToHost);
  }
};
REGISTER_KERNEL_BUILDER(Name("KvVariableGatherOrInsertV2").Device(DEVICE_CPU).HostMemory("table_handle"),
                        KvGatherHipOp<true>);
REGISTER_KERNEL_BUILDER(Name("KvVariableGatherOrZerosV2").Device(DEVICE_CPU).HostMemory("table_handle"),
                        KvGatherHipOp<false>);

// ---- KvVariableGroupSparseApplyAdamV4 / V3 : kernels/training_ops.cc:5709-5965,6980-7213 --------
#define KV_GROUP_ADAM_OP(NAME)                   \
  REGISTER_OP(NAME)                              \
      .Input("var: resource")                    \
      .Input("m_v_linear: resource")             \
      .Input("grad: T")                          \
      .Input("indices: Tindices")                \
      .Input("lr: T")                            \
      .Input("beta1_power: T")                   \
      .Input("beta2_power: T")                   \
      .Input("beat1: T")                         \
      .Input("beta2: T")                         \
      .Input("epsilon: T")                       \
      .Input("l1: T")                            \
      .Input("l2: T")                            \
      .Input("l21: T")                           \
      .Attr("T: numbertype")                     \
      .Attr("Tindices: {int32, int64, uint64, string}") \
      .Attr("use_locking: bool = false")         \
      .SetShapeFn(shape_inference::NoOutputs)
KV_GROUP_ADAM_OP("KvVariableGroupSparseApplyAdamV4");
KV_GROUP_ADAM_OP("KvVariableGroupSparseApplyAdamV3");

template <int VERSION>
class KvGroupAdamHipOp : public OpKernel {
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    KvHipResource *var, *slot;
    OP_REQUIRES_OK(ctx, LookupResource(ctx, HandleFromInput(ctx, 0), &var));
    core::ScopedUnref u0(var);
    OP_REQUIRES_OK(ctx, LookupResource(ctx, HandleFromInput(ctx, 1), &slot));
    core::ScopedUnref u1(slot);
    const Tensor& grad = ctx->input(2);
    const Tensor& ids = ctx->input(3);
    OP_REQUIRES(ctx, TensorShapeUtils::IsVector(ids.shape()),
                errors::InvalidArgument("indices must be one-dimensional"));
    for (int i = 4; i <= 12; ++i)
      OP_REQUIRES(ctx, TensorShapeUtils::IsScalar(ctx->input(i).shape()),
                  errors::InvalidArgument("input ", i, " is not a scalar"));
    OP_REQUIRES(ctx, grad.dim_size(0) == ids.dim_size(0),
                errors::InvalidArgument("grad must be the same size as indices in the first dimension."));
    auto f = [&](int i) { return ctx->input(i).scalar<float>()(); };
    const int64_t n = ids.dim_size(0);
    DevBuf d_ids(ids.TotalBytes()), d_grad(grad.TotalBytes());
    hipMemcpy(d_ids.p, ids.data(), ids.TotalBytes(), hipMemcpyHostToDevice);
    hipMemcpy(d_grad.p, grad.data(), grad.TotalBytes(), hipMemcpyHostToDevice);
    OP_REQUIRES_OK(ctx, FromKv(kv_apply_group_adam(var->h(), slot->h(), static_cast<const float*>(d_grad.p),
                                                   d_ids.p, n, f(4), f(5), f(6), f(7), f(8), f(9), f(10),
                                                   f(11), f(12), VERSION, nullptr)));
    hipDeviceSynchronize();
  }
};
REGISTER_KERNEL_BUILDER(Name("KvVariableGroupSparseApplyAdamV4").Device(DEVICE_CPU).TypeConstraint<float>("T"),
                        KvGroupAdamHipOp<4>);
REGISTER_KERNEL_BUILDER(Name("KvVariableGroupSparseApplyAdamV3").Device(DEVICE_CPU).TypeConstraint<float>("T"),
                        KvGroupAdamHipOp<3>);

// KvVariableSparseApplyAdagrad, KvVariableSparseGroupSparseApplyFtrlV2, KvVariableSizeV2,
// KvVariableFrequency, ReadKvVariableOpV2, KvVariableScatter*V2, KvVariableInsertV2 and
// KvVariableImport/Export and KvVariableFullOrDeltaImport/Export follow the same pattern over
// kv_apply_adagrad, kv_apply_sparse_group_ftrl, kv_size, kv_sum_freq, kv_export_*, kv_export_delta_*,
// kv_scatter_update, kv_insert, kv_import, kv_import_delta (INTEGRATION.md lists the one-line mapping
// for each); the KvVariable op's kernel calls kv_set_delta_tracking once from SUPPORT_DELTA_EXPORT /
// SUPPORT_PREDICTION_DELTA_EXPORT, as the reference constructor reads them (kv_variable.h:100-111).
}  // namespace tfplus_hip
