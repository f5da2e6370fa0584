// See cuda_runtime.h in this directory: the two cuBLAS names src/cuda/utils.h declares functions
// with (get_cublas_handle, cublasGetStatusName). The MI355X backend has no BLAS handle: GEMMs are
// the hand-written kernels behind ct2x_gemm_*.
#pragma once
typedef int cublasStatus_t;
static const cublasStatus_t CUBLAS_STATUS_SUCCESS = 0;
typedef void* cublasHandle_t;
