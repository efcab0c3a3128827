// Internal: throughput bf16 GEMMs (gemm_fast.hip).  Not part of the C ABI.
#pragma once
#include "common.h"

bool gemm_fast_nt_ok(const void* A, const void* B, const void* C, int M, int N, int K, long lda, long ldb);
bool gemm_fast_tn_ok(const void* A, const void* B, const void* C, int M, int N, int K, long lda, long ldb);
// C[M,N] = A[M,K] . B[N,K]^T (+ bias[n]); c_dtype 0 = f32, 1 = bf16
int gemm_nt_bf16(const bf16_t* A, const bf16_t* B, void* C, int c_dtype, const float* bias, int M, int N, int K, long lda,
                 long ldb, long ldc, hipStream_t st);
// C[M,N] (f32) (+)= A[K,M]^T . B[K,N]; with accumulate != 0 (or internal split-K) the result is ADDED atomically to C
int gemm_tn_bf16(const bf16_t* A, const bf16_t* B, float* C, int M, int N, int K, long lda, long ldb, long ldc, int accumulate,
                 hipStream_t st);
