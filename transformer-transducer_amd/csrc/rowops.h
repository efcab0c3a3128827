// Internal launchers of the HBM-bound row kernels (rowops.hip).  Not part of the C ABI.
#pragma once
#include "common.h"

enum MaskKind { MASK_NONE = 0, MASK_CAUSAL = 1, MASK_BAND = 2, MASK_TENSOR = 3, MASK_INTERVAL = 4 };

struct MaskDesc {
    int kind = MASK_NONE;
    int left = 0, right = 0;            // MASK_BAND: masked iff j > i + right or j < i - left
    const uint8_t* ptr = nullptr;       // MASK_TENSOR: nonzero = masked, element (b,i,j) at ptr[b*sb + i*si + j]
                                        // MASK_INTERVAL: int32 pairs (lo, hi) at ((const int*)ptr)[b*sb + 2*i]: masked iff j < lo || j > hi
    long sb = 0, si = 0;
};

// y = LN(x + res) * g + b ; optionally stores s = x + res, mean, rstd (for backward)
// y (f32, optional) and/or y16 (bf16, optional) receive the normalised output
// res_drop: dropout applied to `res` before the add; out_drop: dropout applied to the normalised output
// pre (optional): a second norm of the SAME row in the same pass - h16 = bf16(LN(y; pre->g, pre->b)), its statistics to pre->mean / rstd
struct LnPreNorm {
    const float* g = nullptr;
    const float* b = nullptr;
    bf16_t* h16 = nullptr;
    float* mean = nullptr;
    float* rstd = nullptr;
};
int ln_fwd(const float* x, const float* res, const float* g, const float* b, long rows, int d, float eps, float* s_out,
           float* y, float* mean, float* rstd, hipStream_t st, bf16_t* y16 = nullptr, DropSpec res_drop = DropSpec(),
           DropSpec out_drop = DropSpec(), const LnPreNorm* pre = nullptr);
extern int g_ln_bwd_grid;       // workgroups of the grid-stride LayerNorm backward kernels (ttmi_set_option(12, n))
// dx = dadd + LN'(dy) ; dgamma/dbeta accumulated atomically (caller zeroes them once per step)
// dy_drop: the forward applied dropout to the LN output, so dy is masked/scaled identically on the way in
// dx16 (optional, bf16 pipeline): bf16(dx * dropout(dx16_drop)) - the masked gradient the following GEMMs consume - and, with
// dx16_colsum, its column sums accumulated atomically (the bias gradient of the Linear in front of that dropout)
int ln_bwd(const float* dy, const float* s, const float* mean, const float* rstd, const float* g, const float* dadd, long rows,
           int d, float* dx, float* dgamma, float* dbeta, hipStream_t st, DropSpec dy_drop = DropSpec(), bf16_t* dx16 = nullptr,
           DropSpec dx16_drop = DropSpec(), float* dx16_colsum = nullptr);
// the FFN's pre-norm backward and the attention sub-layer's post-norm backward of one layer in one pass (d % 4 == 0, d <= 512):
//   dy = LN1'(dh; y, mean1, rstd1, g1) + dres1 ; dx = LN2'(dy; s, mean2, rstd2, g2) ; dx16 = bf16(dx * dropout(dx16_drop)); dy is not stored
bool ln_bwd_pair_supported(int d);
int ln_bwd_pair(const float* dh, const float* y, const float* mean1, const float* rstd1, const float* g1, const float* dres1, const float* s,
                const float* mean2, const float* rstd2, const float* g2, long rows, int d, float* dx, float* dgamma1, float* dbeta1,
                float* dgamma2, float* dbeta2, bf16_t* dx16, DropSpec dx16_drop, hipStream_t st);
// out[i] = in[i] * dropout_multiplier(i)  (out f32 and/or bf16); with p = 0 this is the plain f32 -> bf16 conversion
int dropout_apply(const float* in, long n, DropSpec ds, float* out32, bf16_t* out16, hipStream_t st);
// in-place P = softmax_j(scale * S) over the batched score view (nb slabs, L rows of ld floats each)
int softmax_fwd(float* S, int nb, int nh, int L, long ld, long slab, float scale, const MaskDesc& m, hipStream_t st);
// in-place dS = P * (dP - sum_j dP*P) * scale
int softmax_bwd(float* dP, const float* P, int nb, int L, long ld, long slab, float scale, hipStream_t st);
// out[r, c] = in[r*ldi + c] + bias[c]
int add_row_bias(const float* in, long ldi, const float* bias, long rows, int cols, float* out, long ldo, hipStream_t st);
int add_row_bias_bf16(const bf16_t* in, long ldi, const float* bias, long rows, int cols, bf16_t* out, long ldo, hipStream_t st);
// out[z][c] += sum_r in[z][r*ld + c]   (atomic; out zeroed by the caller).  z = z1*nz2+z2 with strides; the
// output offset for batch z is z2*so2 (z1 always accumulates into the same row) unless so1 != 0.
int colsum(const float* in, long ld, long rows, int cols, int nz1, int nz2, long si1, long si2, long so1, long so2, float* out,
           hipStream_t st);
// E[p,h,:] = r_emb[e(p),h,:], cT[h][p] = r_bias[e(p),h], e(p) = max(0, p + K - L)
int relpos_gather(const float* r_emb, const float* r_bias, int K, int L, int H, int Dh, float* E, float* cT, hipStream_t st,
                  bf16_t* E16 = nullptr);   // E16 (optional): bf16 copy of E from the same pass
// g_r_emb[e(p),h,:] += dE[p,h,:], g_r_bias[e(p),h] += dcT[h][p]
int relpos_scatter(const float* dE, const float* dcT, int K, int L, int H, int Dh, float* g_r_emb, float* g_r_bias,
                   hipStream_t st);
int embed_fwd(const long* tokens, const float* W, long n, int d, int V, float* out, hipStream_t st);
int embed_bwd(const long* tokens, const float* dout, long n, int d, int V, int padding_idx, float* gW, hipStream_t st);
// H[b,t,u,:] = tanh(PE[b,t,:] + PD[b,u,:] + bias)    (H f32 or bf16)
int joint_tanh_fwd(const float* PE, const float* PD, const float* bias, int B, int T, int U1, int J, void* H, int h_dtype,
                   hipStream_t st);
// the same with bf16 H, plus emis[row] = 4 f32 logits: (blank, the row's next label labels[b][u]) formed from the bf16 H and the bf16
// projection weight Wp16 [V, J] exactly as the GEMM multiplies them, and the same two from the unrounded hidden row and the f32 master
// weight Wp32 (exp-domain loss form; row (b,t,U1-1) has no label: its label entries = the blank's)
int joint_tanh_fwd_emis(const float* PE, const float* PD, const float* bias, int B, int T, int U1, int J, bf16_t* H, const bf16_t* Wp16,
                        const float* Wp32, const float* bp, const int* labels, int V, int blank, float* emis, hipStream_t st);
// dpre = dH * (1 - H^2);  dPE[b,t,:] = sum_u dpre, dPD[b,u,:] += sum_t dpre (atomic; caller zeroes dPD)
int joint_tanh_bwd(const void* dH, const void* H, int h_dtype, int B, int T, int U1, int J, float* dPE, float* dPD,
                   hipStream_t st);
// bf16x3: H as two-block rows [hi | lo] (pitch 2 Jp bf16) instead of f32, and the backward from them (rowops.hip)
int joint_tanh_fwd_x3(const float* PE, const float* PD, const float* bias, int B, int T, int U1, int J, int Jp, bf16_t* H3, hipStream_t st);
int joint_tanh_bwd_x3(const float* dH, const bf16_t* H3, int B, int T, int U1, int J, int Jp, float* dPE, float* dPD, hipStream_t st);
int x3_fold_blocks(const float* T, int Mp, int Np, float* C, int M, int N, long ldc, hipStream_t st);
size_t joint_sum_bwd_part_floats(int B, int T, int U1, int J);
int joint_sum_bwd_two_pass(const bf16_t* dP, int B, int T, int U1, int J, float* dPE, float* dPD, float* part, hipStream_t st);
int fill_zero(void* p, size_t bytes, hipStream_t st);             // a KERNEL (graph-safe), never a memset node: see rowops.hip
int fill_zero2d(void* p, size_t pitch, size_t width, size_t height, hipStream_t st);
// dst (bf16) = src (f32), n elements
int convert_bf16(const float* src, bf16_t* dst, long n, hipStream_t st);
// lo (bf16) = src - float(hi): the second term of the two-term bf16 split src ~ hi + lo
int bf16_residual(const float* src, const bf16_t* hi, bf16_t* lo, long n, hipStream_t st);
// r (f32) = src - float(bf16(src)): the same term for kernels that round f32 operands while staging them
int bf16_residual_f32(const float* src, float* r, long n, hipStream_t st);
// dst[c, r] (bf16, pitch ldd >= R, columns [R, ldd) zero) = src[r, c] (f32, [R, C] dense)
// plain (optional): also the untransposed bf16 copy [R, C] (forward passes get both from one read of the weight)
int transpose_convert_bf16(const float* src, int R, int C, bf16_t* dst, long ldd, hipStream_t st, bf16_t* plain = nullptr);
// bf16x3 mode: src f32 [R, C] (pitch ld) -> dst bf16 [R, 3 Cp] as blocks [hi | lo | hi] (mode 0: the A-like operand) or [hi | hi | lo]
// (mode 1: the B-like operand), or [R, 2 Cp] as [hi | lo] (mode 2: the planes of a TN product); columns [C, Cp) of every block zero;
// and the three-block forms of the transpose (dst [C, 3 Rp])
int split3_bf16(const float* src, long ld, long R, int C, int Cp, int mode, bf16_t* dst, hipStream_t st);
int relu_mask_scale(float* g, const float* a, long n, float scale, hipStream_t st);
int split3_transpose_bf16(const float* src, long ld, int R, int C, int Rp, bool hhl, bf16_t* dst, hipStream_t st);
// one launch rebuilding the plain + transposed bf16 copies of every weight in `table` (device, n rows of 8 longs, see rowops.hip)
int shadow_refresh(const long* table, int n, long total_tiles, hipStream_t st, long lo_delta = 0);
// out[c] += sum_r in[r*ld + c] for a bf16 matrix (atomic; caller zeroes / accumulates)
int colsum_bf16(const bf16_t* in, long ld, long rows, int cols, float* out, hipStream_t st, int nz1 = 1, int nz2 = 1, long si1 = 0,
                long si2 = 0, long so2 = 0);   // batch z = z1*nz2+z2 reads in + z1*si1 + z2*si2, adds into out + z2*so2
// out (device u64) = (first row whose argmax != blank) << 32 | that argmax, or n << 32 when every row is blank
int greedy_scan(const void* logits, int dtype, long ld, int n, int V, int blank, unsigned long long* out, hipStream_t st);
// batched lockstep greedy decoding (see rowops.hip): per-utterance scan of a block of frames and the state update that consumes it
int greedy_scan_batch(const void* logits, int dtype, long ld, int B, int n, int V, int blank, const int* t, const int* T_len, const int* need,
                      unsigned long long* key, hipStream_t st);
int greedy_advance(unsigned long long* key, int B, int n, int n_hist, long* hist, long ld_hist, int* t, const int* T_len, int* need, int* done,
                   int* count, int* flags, hipStream_t st);
// batched transpose to bf16: for z = z1*nz2+z2, dst[z][c][r] = src[z1*s1 + z2*s2 + r*ld + c] (r < R, c < C), dst pitch ldd >= R with
// zero fill in [R, ldd), dst slab = C*ldd.  src_dtype 0 = f32, 1 = bf16.  Produces the K-major operands of the position products.
int transpose_bf16_batched(const void* src, int src_dtype, long ld, int nz1, int nz2, long s1, long s2, int R, int C, bf16_t* dst,
                           long ldd, hipStream_t st);
