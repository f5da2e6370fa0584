// See ../cuda_runtime.h: src/cuda/utils.h includes this header for its THRUST_CALL macro, which
// only the reference's own .cu files expand.
#pragma once
