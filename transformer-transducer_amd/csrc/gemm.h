// Internal descriptor of the generic MFMA GEMM (gemm.hip).  Not part of the C ABI.
#pragma once
#include "common.h"

enum GemmDType { DT_F32 = 0, DT_BF16 = 1 };
enum GemmFlags {
    GEMM_BIAS = 1,       // C[m,n] += bias[n]            (bias has its own batch strides)
    GEMM_RELU = 2,       // C = max(C, 0)
    GEMM_ATOMIC = 4,     // C += result by float atomics (C must be f32; used for split-K / batch reduction)
    GEMM_MASK_AUX = 8,   // C = aux[m,n] > 0 ? C : 0     (ReLU backward; aux has C's layout, f32)
    GEMM_A_KMAJOR = 16,  // A[m*lda + k]  (else A[k*lda + m])
    GEMM_B_KMAJOR = 32,  // B[n*ldb + k]  (else B[k*ldb + n])
    GEMM_BF16_MFMA = 64, // compute with v_mfma_f32_32x32x16_bf16 (else exact-f32 v_mfma_f32_32x32x2_f32)
    GEMM_BF16X3 = 128,   // f32 operands, f32 C: three bf16 MFMA terms hi.hi + lo.hi + hi.lo (TTMI_PRECISION=bf16x3, round 5); ignored with GEMM_BF16_MFMA
};

struct GemmDesc {
    const void* A = nullptr;
    const void* B = nullptr;
    void* C = nullptr;
    const float* bias = nullptr;
    const float* aux = nullptr;
    int a_dtype = DT_F32, b_dtype = DT_F32, c_dtype = DT_F32;
    int M = 0, N = 0, K = 0;
    long lda = 0, ldb = 0, ldc = 0;
    // batch index z = z1 * nz2 + z2
    int nz1 = 1, nz2 = 1;
    long sA1 = 0, sA2 = 0, sB1 = 0, sB2 = 0, sC1 = 0, sC2 = 0, sBias1 = 0, sBias2 = 0;
    float alpha = 1.f, beta = 0.f;   // C = alpha*A.B + beta*C_old (+bias, relu...)
    int flags = GEMM_A_KMAJOR | GEMM_B_KMAJOR;
    int splitk = 1;                  // >1 requires GEMM_ATOMIC
    DropSpec drop;                   // dropout on the epilogue result (after ReLU), element index m*ldc + n
};

// returns 0 / <0 invalid / >0 hipError_t
int ttmi_launch_gemm(const GemmDesc& d, hipStream_t st);
// f32 products (A K-major, f32 C, no atomics) with at most `rows` rows use the 32x32-tile split-reduction kernel (default 128; 0 = never)
void ttmi_gemm_set_skinny_rows(int rows);
