// unique_input.h — does a NodeDef input string name output 0 of a tf.unique node?  Plain strings, no TensorFlow types,
// so that tests/test_tf_shim_schema.py can compile and run the rule on its own.
//
// In an UNCHANGED reference graph the optimizer op's `indices` input is the first output of the Unique node that TF-core's
// _deduplicate_indexed_slices creates (python/ops/variable_scope.py:1096-1106 routes the gradient there).  TensorFlow
// names a node after its op type unless the caller names it: "Unique", "Unique_1", ... (array_ops.unique -> op "Unique";
// tf.unique with an axis -> "UniqueV2"), under any name scope.  Only that exact leaf counts, and only output slot 0:
//   ".../Unique:1" is the INVERSE index vector (full of repeats), "UniqueWithCounts" has counts at :2 and is not what
//   _deduplicate_indexed_slices builds, "UniqueIds" is somebody's placeholder, "^Unique" is a control edge.
#pragma once
#include <string>

namespace kv_shim {

inline bool InputIsUniqueValues(const std::string& input) {
  if (input.empty() || input[0] == '^') return false;   // a control input carries no tensor
  std::string in = input;
  const size_t colon = in.rfind(':');
  if (colon != std::string::npos) {
    if (in.compare(colon, std::string::npos, ":0") != 0) return false;   // y is output 0; idx (:1) repeats
    in.resize(colon);
  }
  const size_t slash = in.rfind('/');
  const std::string leaf = slash == std::string::npos ? in : in.substr(slash + 1);
  size_t at = 0;
  if (leaf.compare(0, 6, "Unique") != 0) return false;
  at = 6;
  if (leaf.compare(at, 2, "V2") == 0) at += 2;
  if (at == leaf.size()) return true;
  if (leaf[at] != '_' || at + 1 == leaf.size()) return false;   // the uniquifying suffix TensorFlow appends: _<digits>
  for (size_t i = at + 1; i < leaf.size(); ++i)
    if (leaf[i] < '0' || leaf[i] > '9') return false;
  return true;
}

}  // namespace kv_shim
