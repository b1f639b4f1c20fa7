// MOCK (see op_kernel.h in this directory).
#pragma once
#include <memory>
#include <vector>

#include "tensorflow/core/framework/op_kernel.h"

namespace tensorflow {
namespace shape_inference {
class Shape;
class Dimension;
struct ShapeHandle { const Shape* p = nullptr; };
struct DimensionHandle { const Dimension* p = nullptr; };
struct ShapeAndType {
  ShapeAndType() = default;
  ShapeAndType(ShapeHandle s, DataType t) : shape(s), dtype(t) {}
  ShapeHandle shape;
  DataType dtype = DT_INVALID;
};
class InferenceContext {
 public:
  static constexpr int64_t kUnknownDim = -1;
  ShapeHandle input(int i) const;
  int num_inputs() const;
  void set_output(int i, ShapeHandle s);
  ShapeHandle Scalar();
  ShapeHandle Vector(DimensionHandle d);
  ShapeHandle Vector(int64_t d);
  ShapeHandle Matrix(DimensionHandle a, DimensionHandle b);
  ShapeHandle UnknownShape();
  ShapeHandle UnknownShapeOfRank(int r);
  DimensionHandle UnknownDim();
  DimensionHandle MakeDim(int64_t v);
  DimensionHandle Dim(ShapeHandle s, int i);
  Status WithRank(ShapeHandle s, int r, ShapeHandle* out);
  Status WithRankAtLeast(ShapeHandle s, int r, ShapeHandle* out);
  Status WithRankAtMost(ShapeHandle s, int r, ShapeHandle* out);
  Status Merge(ShapeHandle a, ShapeHandle b, ShapeHandle* out);
  Status Concatenate(ShapeHandle a, ShapeHandle b, ShapeHandle* out);
  Status Subshape(ShapeHandle s, int start, ShapeHandle* out);
  Status MakeShapeFromPartialTensorShape(const PartialTensorShape& p, ShapeHandle* out);
  template <typename T> Status GetAttr(const char* name, T* v) const;
  void set_output_handle_shapes_and_types(int i, const std::vector<ShapeAndType>& v);
  const std::vector<ShapeAndType>* input_handle_shapes_and_types(int i) const;
  bool RankKnown(ShapeHandle s) const;
  int Rank(ShapeHandle s) const;
};
Status NoOutputs(InferenceContext* c);
Status UnknownShape(InferenceContext* c);
Status ScalarShape(InferenceContext* c);
Status UnchangedShape(InferenceContext* c);
}  // namespace shape_inference
}  // namespace tensorflow
