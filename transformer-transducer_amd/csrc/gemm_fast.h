// Internal: throughput bf16 GEMMs (gemm_fast.hip).  Not part of the C ABI.
#pragma once
#include "common.h"

bool gemm_fast_nt_ok(const void* A, const void* B, const void* C, int M, int N, int K, long lda, long ldb);
bool gemm_fast_tn_ok(const void* A, const void* B, const void* C, int M, int N, int K, long lda, long ldb);
// epilogue of the NT kernel: v = acc (+ bias[n]) (+ addend[m*ldc+n]); relu; v = mask[m*ldc+n] > 0 ? v : 0
// two-level batch: z = z1 * nz2 + z2; element offsets added to A / B / C (and bias, colsum) per batch index
struct FastBatch {
    int nz1 = 1, nz2 = 1;
    long sA1 = 0, sA2 = 0, sB1 = 0, sB2 = 0, sC1 = 0, sC2 = 0, sV1 = 0, sV2 = 0;   // sV*: bias (NT) / colsum_a (TN) strides
};

struct NtEpilogue {
    const float* bias = nullptr;
    const float* addend = nullptr;   // f32, same layout as C (residual add)
    const bf16_t* mask = nullptr;    // bf16, same layout as C: mask_mode 0 = ReLU backward (v = mask > 0 ? v * scale : 0), 1 = tanh backward (v *= 1 - mask^2)
    int mask_mode = 0;
    int relu = 0;
    float scale = 1.f;               // applied to the masked result (1/(1-p) of a dropped ReLU in backward)
    DropSpec drop;                   // dropout on the result (after ReLU), element index m*ldc + n
    // optional second operand pair accumulated into the same tile (128x128 kernel only): C = epi(A.B^T + A2.B2^T); A2 is batched with
    // A's strides, B2 with (sB1b, sB2b); colsum_mid (nullable, f32, batch strides sV1/sV2): += column sums of A.B^T alone
    const bf16_t* A2 = nullptr;
    const bf16_t* B2 = nullptr;
    int K2 = 0;
    long lda2 = 0, ldb2 = 0, sB1b = 0, sB2b = 0;
    float* colsum_mid = nullptr;
    // persistent 256x256 kernel only: "exp store" - C = bf16(exp(acc + bias - exp_shift)) with zeros in columns [N, ldc) and per-row partial
    // sums rowsum[m * nparts + part] (part < 4 * ceil(N / 256)); rowscale: per-row factor applied together with the tanh' mask; the mask operand is rewritten in place as rowscale[m] * mask
    float* rowsum = nullptr;
    int nparts = 0;
    const float* exp_shift = nullptr;   // device scalar, nullable = 0
    const float* rowscale = nullptr;
    // second bf16 term of the weight (B ~ B + B_lo, same shape and pitch): C = epi(A.(B + B_lo)^T) in ONE launch of the persistent kernels,
    // which walk A's K-tiles a second time against B_lo, or of the 128x128 kernel (as its second operand pair)
    const bf16_t* B_lo = nullptr;
    // ... with K_lo != 0 the second walk covers A's first K_lo columns only: C = epi(A.B^T + A[:, :K_lo].B_lo[:, :K_lo]^T) - the bf16x3 products,
    // A = [hi | lo], B = [hi | hi], B_lo = lo: hi.hi + lo.hi + hi.lo without a third copy of A's hi block (K_lo % 64 == 0, <= K)
    int K_lo = 0;
};
// C[M,N] = epilogue(A[M,K] . B[N,K]^T); c_dtype 0 = f32, 1 = bf16
int gemm_nt_bf16(const bf16_t* A, const bf16_t* B, void* C, int c_dtype, const NtEpilogue& epi, int M, int N, int K, long lda,
                 long ldb, long ldc, hipStream_t st, const FastBatch& batch = FastBatch());
inline int gemm_nt_bf16(const bf16_t* A, const bf16_t* B, void* C, int c_dtype, const float* bias, int M, int N, int K,
                        long lda, long ldb, long ldc, hipStream_t st) {
    NtEpilogue e;
    e.bias = bias;
    return gemm_nt_bf16(A, B, C, c_dtype, e, M, N, K, lda, ldb, ldc, st);
}
// C[M,N] (f32) (+)= A[K,M]^T . B[K,N]; with accumulate != 0 (or internal split-K) the result is ADDED atomically to C
// colsum_a (nullable, f32 [M], accumulated atomically): column sums of A over K - the bias gradient that belongs to this wgrad -
// taken from the LDS tiles the kernel stages anyway (no extra pass over A)
// colsum_w (nullable, bf16 [K], 16-byte aligned; persistent 256x256 kernel only): the column sums become sum_k colsum_w[k] A[k][m]
int gemm_tn_bf16(const bf16_t* A, const bf16_t* B, float* C, int M, int N, int K, long lda, long ldb, long ldc, int accumulate,
                 hipStream_t st, float* colsum_a = nullptr, const FastBatch& batch = FastBatch(), const bf16_t* colsum_w = nullptr);
bool gemm_fast_joint_exp_ok(int M, int V, int J, long ldv, bool fwd_only = false);   // sizes at which the exp-store / row-scale / weighted-colsum forms exist
// one weight-gradient problem of a grouped launch: C[M,N] += A[K,M]^T B[K,N] (bf16 operands, f32 C), colsum (nullable, f32 [M]) += column sums of A
struct TnProblem {
    const bf16_t* A; const bf16_t* B; float* C; float* colsum;
    int M, N, K;
    long lda, ldb, ldc;
};
bool gemm_tn_group_fits(const TnProblem& q);
int gemm_tn_bf16_group(const TnProblem* probs, int n, hipStream_t st);   // deterministic (no atomics) for problems that fit the 256x128 tiling
void gemm_fast_set_version(int v);   // kernel generation for A/B runs, see gemm_fast.hip (default 4)
void gemm_fast_set_tn_target(int n);
void gemm_fast_set_f32(int on);      // 0: gemm_nt_f32_ok() / gemm_nt_f32_mid_ok() answer no (A/B against the f32 kernels of csrc/gemm.hip); 2 / 3: one of the two only
int gemm_fast_f32_mode();
// exact-f32 NT product on the persistent 256x128 kernel (f32 operands through LDS-DMA, v_mfma_f32_16x16x4_f32): C[M,N] = epi(A[M,K] . B[N,K]^T),
// epilogue = bias / addend / ReLU / dropout of NtEpilogue; M >= 1024, N >= 128, K % 32 == 0, 16-byte aligned operands
// mid-sized exact-f32 NT product (64 x 64 tiles, any M / N, K % 32 == 0): C = [C +] A . B^T [+ bias] [ReLU]
bool gemm_nt_f32_mid_ok(const void* A, const void* B, const void* C, int M, int N, int K, long lda, long ldb, long ldc);
int gemm_nt_f32_mid(const float* A, const float* B, float* C, const float* bias, int accumulate, int relu, int M, int N, int K, long lda, long ldb,
                    long ldc, hipStream_t st);
bool gemm_nt_f32_ok(const void* A, const void* B, const void* C, int M, int N, int K, long lda, long ldb, long ldc);
int gemm_nt_f32(const float* A, const float* B, float* C, const NtEpilogue& epi, int M, int N, int K, long lda, long ldb, long ldc, hipStream_t st);
void gemm_fast_set_reserved_cus(int n);   // process-wide default (measurement switch)
void gemm_fast_set_tn_group_pieces(int on);   // grouped weight gradients under a CU reservation: surplus tiles as K-pieces (default on)
void gemm_fast_stream_reserve_cus(hipStream_t st, int n);   // per-stream state: launches on `st` (and on fork streams aliased to it) leave n CUs free
void gemm_fast_alias_stream(hipStream_t child, hipStream_t parent);
//   // CUs left free by the mid-sized persistent GEMMs (for concurrent RCCL kernels)
