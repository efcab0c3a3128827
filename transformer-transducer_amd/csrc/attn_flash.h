// Internal: fused relative-position attention kernels (attn_flash.hip).  Not part of the C ABI.
#pragma once
#include "common.h"

struct FlashParams {
    // bf16 activations; row (b, i) of a [B*L, ld] buffer, head h at column h*Dh
    const bf16_t* qu = nullptr;   // q + r_w_bias
    const bf16_t* k = nullptr;
    const bf16_t* v = nullptr;
    long ld_qu = 0, ld_kv = 0, ld_o = 0;
    const bf16_t* bd = nullptr;   // shifted position scores (bf16): element (z, i, j) at bd[z*slab + i*L + j]  (z = b*H + h)
    long slab = 0;
    // position term formed inside the kernels (no slab): plain q rows (same row indexing as qu), the effective table E16 [L, ld_e] (row p =
    // r_emb[max(0, p + K - L)], head h at column h*Dh) and its bias cT [H][L]; used when e16 != nullptr
    const bf16_t* qp = nullptr;
    long ld_qp = 0;
    const bf16_t* e16 = nullptr;
    long ld_e = 0;
    const float* cT = nullptr;
    const float* u = nullptr;     // r_w_bias [H*Dh] (f32): the in-kernel variants form q + u themselves (no qu tensor)
    bf16_t* o = nullptr;          // attention output (fwd: written, bwd: read)
    float* lse = nullptr;         // [B*H, L] log-sum-exp of the scaled, masked scores
    // backward only
    const bf16_t* dO = nullptr;
    float* delta = nullptr;       // [B*H, L] scratch
    float* dS = nullptr;          // (unused by the kernels since the bf16 outputs below exist; kept for A/B)
    // dS is emitted twice in bf16 with 16-byte aligned rows (pitch ldp, multiple of 8):
    //   dS16[z][i][j]            -> content-term dgrad  dq = dS k
    //   dG16[z][r][c-1], (r, c) = divmod((i+1) L + j, L+1), c >= 1  -> position-term grads (dq += dG E, dE, dc);
    //   this is _rel_shift's adjoint written straight into an aligned [L][ldp] matrix (row 0 is pre-zeroed by the caller)
    bf16_t* dS16 = nullptr;
    bf16_t* dG16 = nullptr;
    long ldp = 0, slab16 = 0;
    float* dK = nullptr;          // f32 [B*L, ld_dkv] (+ h*Dh)
    float* dV = nullptr;
    long ld_dkv = 0;
    bf16_t* dK16 = nullptr;       // if set, dK / dV are written here in bf16 (same pitch ld_dkv, + h*Dh) INSTEAD of the f32 buffers
    bf16_t* dV16 = nullptr;
    int B = 0, L = 0, H = 0, Dh = 0;
    float scale = 1.f;
    int mask_kind = 0, mask_left = 0, mask_right = 0;   // kind 2: the band; kind 4: bounds on the intervals' reach from the diagonal (-1 = unknown)
    const unsigned char* mask = nullptr;
    long mask_sb = 0, mask_si = 0;
    float* zero_f32 = nullptr;    // flash_attn_bwd only: zero_n floats to clear (the dE / dc accumulators of the position gradients) and, with
    long zero_n = 0;              // zero_dg_row0, row 0 of every dG16 slab (the shifted store never writes it) - by the delta kernel's launch
    int zero_dg_row0 = 0;
    int bwd_skip = 0;             // set by flash_attn_bwd: dS16 / dG16 are pre-zeroed, the kernel walks only the query tiles its key block can meet
    int debug = 0;                // measurement only (ttmi_set_option(2, bits)): 1 = skip the bias read, 2 = skip the dS write
};

bool flash_supported(int Dh, long ld_qu, long ld_kv, long ld_o);
// dq (bf16, content + position), dE, dc and d r_w_bias from the two bf16 dS slabs of flash_attn_bwd in one pass (one workgroup per (b, h),
// Dh = 64, ldp <= 4096): see attn_dqde_kernel.  dE / dcT / gu are accumulated into (atomics); dq16 rows are overwritten.  Sequences longer than
// 512 run one workgroup per (b, h, group of 512 columns); the groups' partial dq rows go through `part` (attn_dqde_groups(ldp) * B * L * H * 64
// floats, 16-byte aligned) and are summed by a second small launch.  With g_emb / g_bias (the gradients of r_emb [K, H, 64] and r_bias [K, H]) the
// table-gradient rows go straight into them (effective row p -> table row max(0, p + K - L)): dE / dcT stay untouched, no relpos_scatter follows.
bool attn_dqde_supported(int Dh, int L, long ldp);
int attn_dqde_groups(long ldp);
int attn_dqde(const bf16_t* dS16, const bf16_t* dG16, long slab16, long ldp, const bf16_t* k, long ld_kv, const bf16_t* e16, long ld_e,
              const bf16_t* qp, long ld_qp, bf16_t* dq16, long ld_dq, float* dE, long ld_de, float* dcT, float* gu, int B, int L, int H,
              hipStream_t st, float* part = nullptr, float* g_emb = nullptr, float* g_bias = nullptr, int K = 0, float* part_e = nullptr,
              float* part_c = nullptr);
// With part_e [B, L, H 64] / part_c [B, H, L] attn_dqde leaves every (b, h)'s table-gradient rows by plain stores (no atomics); this sums them over
// b into the gradients of r_emb [K, H, 64] / r_bias [K, H] (+=; effective row p -> table row max(0, p + K - L)).  Meant for a side stream.
int attn_table_grads(const float* part_e, const float* part_c, int B, int L, int H, int K, float* g_emb, float* g_bias, hipStream_t st);
int flash_attn_fwd(const FlashParams& p, hipStream_t st);
void flash_set_resident(int v);     // 1 (default): one workgroup per head with the table resident in LDS where it applies; 0: round-3 kernels
void flash_set_bwd_gen(int v);      // 2 (default): flash_bwd_rel2_kernel (LDS-DMA staging, register skew, one barrier per step); 1: the round-3 kernel
int flash_attn_bwd(const FlashParams& p, hipStream_t st);
// G slab of the position term (q E^T + c, column 0 zero, row pitch L+1, bf16) for all (b, h): q rows (b, i) at q[(b*L+i)*ld_q + h*Dh],
// E rows p at E[p*ld_e + h*Dh], c[h][p]
int relpos_slab(const bf16_t* q, long ld_q, const bf16_t* E, long ld_e, const float* c, int B, int L, int H, int Dh, bf16_t* G,
                hipStream_t st);
void relpos_slab_set_debug(int bits);   // timing experiments only: 1 = skip the MFMA part, 2 = skip the stores
