// MOCK (see op_kernel.h in this directory).
#pragma once
#include "tensorflow/core/framework/shape_inference.h"
