// A MOCK of the slice of TensorFlow 2.13's C++ op API that tfplus_amd/tf_shim/kv_variable_ops_hip.cc uses —
// declarations only, enough for `g++ -fsyntax-only` to type-check the shim in an image without TensorFlow
// (tests/test_tf_shim_schema.py).  It is test infrastructure: nothing here is linked or shipped, and a real
// build includes the real headers.
#pragma once
#include <cstdint>
#include <functional>
#include <initializer_list>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

namespace tsl {
namespace error {
enum Code { OK = 0, CANCELLED = 1, UNKNOWN = 2, INVALID_ARGUMENT = 3, NOT_FOUND = 5, RESOURCE_EXHAUSTED = 8,
            FAILED_PRECONDITION = 9, UNIMPLEMENTED = 12, INTERNAL = 13 };
}  // namespace error
}  // namespace tsl

namespace tensorflow {
using string = std::string;
using int64 = long long;
using int32 = int;
using uint32 = unsigned;
typedef unsigned long long uint64;
typedef unsigned short uint16;
class mutex {
 public:
  void lock();
  void unlock();
};
class mutex_lock {
 public:
  explicit mutex_lock(mutex& m) : m_(&m) { m_->lock(); }
  ~mutex_lock() { m_->unlock(); }

 private:
  mutex* m_;
};
// (tensorflow/core/framework/node_def.pb.h) what the shim reads of a node: its input edge names
class NodeDef {
 public:
  int input_size() const;
  const std::string& input(int i) const;
};

class Status {
 public:
  Status() = default;
  Status(tsl::error::Code code, const std::string& msg) : code_(code), msg_(msg) {}
  bool ok() const { return code_ == tsl::error::OK; }
  tsl::error::Code code() const { return code_; }
  const std::string& message() const { return msg_; }
  std::string ToString() const { return msg_; }

 private:
  tsl::error::Code code_ = tsl::error::OK;
  std::string msg_;
};
inline Status OkStatus() { return Status(); }

namespace errors {
template <typename... A>
std::string Cat_(const A&... a) { std::ostringstream o; (void)std::initializer_list<int>{((o << a), 0)...}; return o.str(); }
template <typename... A> Status InvalidArgument(const A&... a) { return Status(tsl::error::INVALID_ARGUMENT, Cat_(a...)); }
template <typename... A> Status Unimplemented(const A&... a) { return Status(tsl::error::UNIMPLEMENTED, Cat_(a...)); }
template <typename... A> Status Internal(const A&... a) { return Status(tsl::error::INTERNAL, Cat_(a...)); }
template <typename... A> Status FailedPrecondition(const A&... a) { return Status(tsl::error::FAILED_PRECONDITION, Cat_(a...)); }
template <typename... A> Status NotFound(const A&... a) { return Status(tsl::error::NOT_FOUND, Cat_(a...)); }
inline bool IsNotFound(const Status& s) { return s.code() == tsl::error::NOT_FOUND; }
}  // namespace errors

enum DataType { DT_INVALID = 0, DT_FLOAT = 1, DT_INT32 = 3, DT_INT64 = 9, DT_RESOURCE = 20, DT_UINT64 = 23 };
std::string DataTypeString(DataType t);

class TensorShape {
 public:
  TensorShape() = default;
  TensorShape(std::initializer_list<int64_t> d) : d_(d) {}
  int dims() const { return (int)d_.size(); }
  int64_t dim_size(int i) const { return d_[(size_t)i]; }
  int64_t num_elements() const { int64_t n = 1; for (auto v : d_) n *= v; return n; }
  void AddDim(int64_t v) { d_.push_back(v); }
  void InsertDim(int i, int64_t v) { d_.insert(d_.begin() + i, v); }
  std::string DebugString() const;
  bool operator==(const TensorShape& o) const { return d_ == o.d_; }
  bool operator!=(const TensorShape& o) const { return d_ != o.d_; }

 private:
  std::vector<int64_t> d_;
};
class PartialTensorShape {
 public:
  PartialTensorShape() = default;
  PartialTensorShape(std::initializer_list<int64_t>) {}
  PartialTensorShape(const TensorShape&) {}
  int dims() const;
  int64_t dim_size(int i) const;
  void InsertDim(int i, int64_t v);
  void AddDim(int64_t v);
  std::string DebugString() const;
};
struct TensorShapeUtils {
  static bool IsScalar(const TensorShape& s) { return s.dims() == 0; }
  static bool IsVector(const TensorShape& s) { return s.dims() == 1; }
  static bool IsVectorOrHigher(const TensorShape& s) { return s.dims() >= 1; }
  static bool IsMatrix(const TensorShape& s) { return s.dims() == 2; }
};

template <typename T>
struct ScalarView_ { T* p; T& operator()() const { return *p; } };
template <typename T>
struct FlatView_ { T* p; int64_t n; T& operator()(int64_t i) const { return p[i]; } T* data() const { return p; } int64_t size() const { return n; } };

class ResourceHandle {
 public:
  const std::string& container() const;
  const std::string& name() const;
};

class Tensor {
 public:
  Tensor() = default;
  const TensorShape& shape() const;
  DataType dtype() const;
  int dims() const;
  int64_t dim_size(int i) const;
  int64_t NumElements() const;
  size_t TotalBytes() const;
  void* data() const;
  template <typename T> ScalarView_<T> scalar();
  template <typename T> ScalarView_<const T> scalar() const;
  template <typename T> FlatView_<T> flat();
  template <typename T> FlatView_<const T> flat() const;
  template <typename T> FlatView_<T> vec();
  template <typename T> FlatView_<const T> vec() const;
};

class OpKernelConstruction {
 public:
  const NodeDef& def() const;
  template <typename T> Status GetAttr(const char* name, T* v) const;
  void CtxFailure(const char* file, int line, const Status& s);
  void CtxFailureWithWarning(const char* file, int line, const Status& s);
};

class ResourceMgr;
class OpKernelContext {
 public:
  const Tensor& input(int i);
  int num_inputs() const;
  Status allocate_output(int i, const TensorShape& shape, Tensor** out);
  Status allocate_temp(DataType t, const TensorShape& shape, Tensor* out);
  ResourceMgr* resource_manager() const;
  void SetStatus(const Status& s);
  void CtxFailure(const char* file, int line, const Status& s);
  void CtxFailureWithWarning(const char* file, int line, const Status& s);
  const Status& status() const;
};

class OpKernel {
 public:
  explicit OpKernel(OpKernelConstruction*) {}
  virtual ~OpKernel() {}
  virtual void Compute(OpKernelContext* ctx) = 0;
  const std::string& name() const;
  const NodeDef& def() const;
};

#define OP_REQUIRES(CTX, EXP, STATUS)                          \
  do {                                                         \
    if (!(EXP)) { (CTX)->CtxFailure(__FILE__, __LINE__, (STATUS)); return; } \
  } while (0)
#define OP_REQUIRES_OK(CTX, ...)                               \
  do {                                                         \
    ::tensorflow::Status _s(__VA_ARGS__);                      \
    if (!_s.ok()) { (CTX)->CtxFailureWithWarning(__FILE__, __LINE__, _s); return; } \
  } while (0)
#define TF_RETURN_IF_ERROR(...)                                \
  do {                                                         \
    ::tensorflow::Status _status = (__VA_ARGS__);              \
    if (!_status.ok()) return _status;                         \
  } while (0)

extern const char* const DEVICE_CPU;
extern const char* const DEVICE_GPU;
struct KernelDefBuilder {
  explicit KernelDefBuilder(const char*) {}
  KernelDefBuilder& Device(const char*) { return *this; }
  KernelDefBuilder& HostMemory(const char*) { return *this; }
  template <typename T> KernelDefBuilder& TypeConstraint(const char*) { return *this; }
};
inline KernelDefBuilder Name(const char* n) { return KernelDefBuilder(n); }
#define TF_MOCK_CAT2_(a, b) a##b
#define TF_MOCK_CAT_(a, b) TF_MOCK_CAT2_(a, b)
#define REGISTER_KERNEL_BUILDER(BUILDER, ...)                                                      \
  static const ::tensorflow::KernelDefBuilder TF_MOCK_CAT_(tf_mock_kernel_, __COUNTER__) = (BUILDER); \
  static_assert(std::is_base_of<::tensorflow::OpKernel, __VA_ARGS__>::value, "not an OpKernel");
}  // namespace tensorflow
