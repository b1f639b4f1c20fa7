// mock of tensorflow/core/framework/node_def.pb.h: NodeDef lives in op_kernel.h of this mock tree
#pragma once
#include "tensorflow/core/framework/op_kernel.h"
