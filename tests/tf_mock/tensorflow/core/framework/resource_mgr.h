// MOCK (see op_kernel.h in this directory).
#pragma once
#include <functional>
#include <string>

#include "tensorflow/core/framework/op_kernel.h"

namespace tensorflow {
class ResourceBase {
 public:
  virtual ~ResourceBase() {}
  virtual std::string DebugString() const = 0;
  void Ref() const;
  bool Unref() const;
};
namespace core {
class ScopedUnref {
 public:
  explicit ScopedUnref(const ResourceBase* r) : r_(r) {}
  ~ScopedUnref() { if (r_) r_->Unref(); }

 private:
  const ResourceBase* r_;
};
template <typename T>
class RefCountPtr {
 public:
  RefCountPtr() = default;
  T* get() const { return p_; }
  T* operator->() const { return p_; }
  T** operator&() { return &p_; }
  explicit operator bool() const { return p_ != nullptr; }

 private:
  T* p_ = nullptr;
};
}  // namespace core
class ResourceMgr {
 public:
  template <typename T> Status Create(const std::string& container, const std::string& name, T* resource);
  template <typename T> Status Lookup(const std::string& container, const std::string& name, T** resource) const;
  template <typename T> Status LookupOrCreate(const std::string& container, const std::string& name, T** resource,
                                              std::function<Status(T**)> creator);
  template <typename T> Status Delete(const std::string& container, const std::string& name);
  const std::string& default_container() const;
};
class ContainerInfo {
 public:
  Status Init(ResourceMgr* rmgr, const NodeDef& ndef, bool use_node_name_as_default);
  Status Init(ResourceMgr* rmgr, const NodeDef& ndef);
  ResourceMgr* resource_manager() const;
  const std::string& container() const;
  const std::string& name() const;
};
template <typename T> ResourceHandle MakeResourceHandle(OpKernelContext* ctx, const std::string& container, const std::string& name);
template <typename T> Status MakeResourceHandleToOutput(OpKernelContext* ctx, int output, const std::string& container, const std::string& name);
const ResourceHandle& HandleFromInput(OpKernelContext* ctx, int input);
template <typename T> Status LookupResource(OpKernelContext* ctx, const ResourceHandle& h, T** value);
template <typename T> Status DeleteResource(OpKernelContext* ctx, const ResourceHandle& h);
Status DeleteResource(OpKernelContext* ctx, const ResourceHandle& h);
}  // namespace tensorflow
