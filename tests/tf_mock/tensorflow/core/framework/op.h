// MOCK (see op_kernel.h in this directory).
#pragma once
#include <functional>

#include "tensorflow/core/framework/shape_inference.h"

namespace tensorflow {
struct OpDefBuilderMock {
  explicit OpDefBuilderMock(const char*) {}
  OpDefBuilderMock& Input(const char*) { return *this; }
  OpDefBuilderMock& Output(const char*) { return *this; }
  OpDefBuilderMock& Attr(const char*) { return *this; }
  OpDefBuilderMock& SetIsStateful() { return *this; }
  OpDefBuilderMock& SetIsCommutative() { return *this; }
  OpDefBuilderMock& Doc(const char*) { return *this; }
  OpDefBuilderMock& SetShapeFn(std::function<Status(shape_inference::InferenceContext*)>) { return *this; }
};
#define REGISTER_OP(NAME) \
  static const ::tensorflow::OpDefBuilderMock TF_MOCK_CAT_(tf_mock_op_, __COUNTER__) = ::tensorflow::OpDefBuilderMock(NAME)
}  // namespace tensorflow
