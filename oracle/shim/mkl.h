/* TEST INFRASTRUCTURE ONLY (oracle/_ref build).
 * Prototype-only stand-in for Intel MKL's mkl.h. The image ships the MKL 2021.4
 * RUNTIME (/opt/conda/lib/libmkl_*.so) but no headers. Only the entry points the
 * reference's CPU backend calls are declared (call sites: src/cpu/primitives.cc:119-401,
 * 549-1160; src/cpu/allocator.cc:57-68), with the signatures of Intel's public
 * oneMKL C API documentation, ILP64 interface (MKL_INT = 64-bit).
 */
#pragma once
#include <stddef.h>
#include <stdint.h>

#define __INTEL_MKL__ 2021
#define __INTEL_MKL_MINOR__ 0
#define __INTEL_MKL_UPDATE__ 4

#ifdef __cplusplus
extern "C" {
#endif

typedef long long MKL_INT;
typedef int8_t MKL_INT8;
typedef uint8_t MKL_UINT8;
typedef int16_t MKL_INT16;
typedef int32_t MKL_INT32;
typedef size_t CBLAS_INDEX;

typedef enum { CblasRowMajor = 101, CblasColMajor = 102 } CBLAS_LAYOUT;
typedef enum { CblasNoTrans = 111, CblasTrans = 112, CblasConjTrans = 113 } CBLAS_TRANSPOSE;
typedef enum { CblasPacked = 151 } CBLAS_STORAGE;
typedef enum { CblasAMatrix = 161, CblasBMatrix = 162 } CBLAS_IDENTIFIER;
typedef enum { CblasRowOffset = 171, CblasColOffset = 172, CblasFixOffset = 173 } CBLAS_OFFSET;

/* The lower-case names in libmkl_intel_ilp64 are Fortran (by-reference) entry
 * points; the C service functions are the MKL_* spellings. */
void* MKL_malloc(size_t size, int align);
void MKL_free(void* ptr);
void MKL_Free_Buffers(void);
#define mkl_malloc MKL_malloc
#define mkl_free MKL_free
#define mkl_free_buffers MKL_Free_Buffers

CBLAS_INDEX cblas_isamax(const MKL_INT n, const float* x, const MKL_INT incx);
void cblas_saxpby(const MKL_INT n, const float alpha, const float* x, const MKL_INT incx,
                  const float beta, float* y, const MKL_INT incy);

void cblas_sgemm(const CBLAS_LAYOUT layout, const CBLAS_TRANSPOSE transa,
                 const CBLAS_TRANSPOSE transb, const MKL_INT m, const MKL_INT n,
                 const MKL_INT k, const float alpha, const float* a, const MKL_INT lda,
                 const float* b, const MKL_INT ldb, const float beta, float* c,
                 const MKL_INT ldc);
size_t cblas_sgemm_pack_get_size(const CBLAS_IDENTIFIER identifier, const MKL_INT m,
                                 const MKL_INT n, const MKL_INT k);
void cblas_sgemm_pack(const CBLAS_LAYOUT layout, const CBLAS_IDENTIFIER identifier,
                      const CBLAS_TRANSPOSE trans, const MKL_INT m, const MKL_INT n,
                      const MKL_INT k, const float alpha, const float* src,
                      const MKL_INT ld, float* dest);
void cblas_sgemm_compute(const CBLAS_LAYOUT layout, const MKL_INT transa,
                         const MKL_INT transb, const MKL_INT m, const MKL_INT n,
                         const MKL_INT k, const float* a, const MKL_INT lda,
                         const float* b, const MKL_INT ldb, const float beta, float* c,
                         const MKL_INT ldc);
void cblas_sgemm_batch(const CBLAS_LAYOUT layout, const CBLAS_TRANSPOSE* transa_array,
                       const CBLAS_TRANSPOSE* transb_array, const MKL_INT* m_array,
                       const MKL_INT* n_array, const MKL_INT* k_array,
                       const float* alpha_array, const float** a_array,
                       const MKL_INT* lda_array, const float** b_array,
                       const MKL_INT* ldb_array, const float* beta_array, float** c_array,
                       const MKL_INT* ldc_array, const MKL_INT group_count,
                       const MKL_INT* group_size);
void cblas_sgemm_batch_strided(const CBLAS_LAYOUT layout, const CBLAS_TRANSPOSE transa,
                               const CBLAS_TRANSPOSE transb, const MKL_INT m,
                               const MKL_INT n, const MKL_INT k, const float alpha,
                               const float* a, const MKL_INT lda, const MKL_INT stridea,
                               const float* b, const MKL_INT ldb, const MKL_INT strideb,
                               const float beta, float* c, const MKL_INT ldc,
                               const MKL_INT stridec, const MKL_INT batch_size);

void cblas_gemm_s8u8s32(const CBLAS_LAYOUT layout, const CBLAS_TRANSPOSE transa,
                        const CBLAS_TRANSPOSE transb, const CBLAS_OFFSET offsetc,
                        const MKL_INT m, const MKL_INT n, const MKL_INT k,
                        const float alpha, const void* a, const MKL_INT lda,
                        const MKL_INT8 ao, const void* b, const MKL_INT ldb,
                        const MKL_INT8 bo, const float beta, MKL_INT32* c,
                        const MKL_INT ldc, const MKL_INT32* co);
size_t cblas_gemm_s8u8s32_pack_get_size(const CBLAS_IDENTIFIER identifier, const MKL_INT m,
                                        const MKL_INT n, const MKL_INT k);
void cblas_gemm_s8u8s32_pack(const CBLAS_LAYOUT layout, const CBLAS_IDENTIFIER identifier,
                             const CBLAS_TRANSPOSE trans, const MKL_INT m, const MKL_INT n,
                             const MKL_INT k, const void* src, const MKL_INT ld, void* dest);
void cblas_gemm_s8u8s32_compute(const CBLAS_LAYOUT layout, const MKL_INT transa,
                                const MKL_INT transb, const CBLAS_OFFSET offsetc,
                                const MKL_INT m, const MKL_INT n, const MKL_INT k,
                                const float alpha, const void* a, const MKL_INT lda,
                                const MKL_INT8 ao, const void* b, const MKL_INT ldb,
                                const MKL_INT8 bo, const float beta, MKL_INT32* c,
                                const MKL_INT ldc, const MKL_INT32* co);

void cblas_gemm_s16s16s32(const CBLAS_LAYOUT layout, const CBLAS_TRANSPOSE transa,
                          const CBLAS_TRANSPOSE transb, const CBLAS_OFFSET offsetc,
                          const MKL_INT m, const MKL_INT n, const MKL_INT k,
                          const float alpha, const MKL_INT16* a, const MKL_INT lda,
                          const MKL_INT16 ao, const MKL_INT16* b, const MKL_INT ldb,
                          const MKL_INT16 bo, const float beta, MKL_INT32* c,
                          const MKL_INT ldc, const MKL_INT32* co);
size_t cblas_gemm_s16s16s32_pack_get_size(const CBLAS_IDENTIFIER identifier,
                                          const MKL_INT m, const MKL_INT n, const MKL_INT k);
void cblas_gemm_s16s16s32_pack(const CBLAS_LAYOUT layout, const CBLAS_IDENTIFIER identifier,
                               const CBLAS_TRANSPOSE trans, const MKL_INT m,
                               const MKL_INT n, const MKL_INT k, const MKL_INT16* src,
                               const MKL_INT ld, MKL_INT16* dest);
void cblas_gemm_s16s16s32_compute(const CBLAS_LAYOUT layout, const MKL_INT transa,
                                  const MKL_INT transb, const CBLAS_OFFSET offsetc,
                                  const MKL_INT m, const MKL_INT n, const MKL_INT k,
                                  const float alpha, const MKL_INT16* a, const MKL_INT lda,
                                  const MKL_INT16 ao, const MKL_INT16* b, const MKL_INT ldb,
                                  const MKL_INT16 bo, const float beta, MKL_INT32* c,
                                  const MKL_INT ldc, const MKL_INT32* co);

void vsAdd(const MKL_INT n, const float a[], const float b[], float r[]);
void vsSub(const MKL_INT n, const float a[], const float b[], float r[]);
void vsMul(const MKL_INT n, const float a[], const float b[], float r[]);
void vsFmax(const MKL_INT n, const float a[], const float b[], float r[]);
void vsFmin(const MKL_INT n, const float a[], const float b[], float r[]);
void vsExp(const MKL_INT n, const float a[], float r[]);
void vsLn(const MKL_INT n, const float a[], float r[]);
void vsSin(const MKL_INT n, const float a[], float r[]);
void vsCos(const MKL_INT n, const float a[], float r[]);
void vsTanh(const MKL_INT n, const float a[], float r[]);

#ifdef __cplusplus
}
#endif
