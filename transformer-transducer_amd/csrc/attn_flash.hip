// Fused relative-position attention for the bf16 pipeline (gfx950): scores are never materialised as
// probabilities in HBM.
//
//   flash_fwd_kernel   S^T = K.(q+u)^T + BD^T  (bias = the shifted position term read through the pitch-L view of
//                      the G slab, tt/transformer.py:140-149), mask as kernel parameters (:154-159), online softmax
//                      (:164), O^T += V^T.P^T (:167).  Keys live on the MFMA rows, the query on the lane, so every
//                      softmax statistic is lane-local (one __shfl_xor 32 per tile) and the S^T accumulator is the B
//                      operand of the PV product with no lane movement; V^T fragments come from ds_read_b64_tr_b16.
//   flash_bwd_kernel   recomputes P from (q+u), k, BD and the saved log-sum-exp; dP = dO.V^T; dS = P(dP - delta)scale is
//                      written once (pitch-L view of the dS slab == dG in the pitch-(L+1) layout, consumed by the position
//                      GEMMs); dV^T += dO^T.P and dK^T += (q+u)^T.dS accumulate in registers (key on the lane: P and dS
//                      accumulators are the B operands as they stand; dO^T / (q+u)^T fragments by transposed LDS reads).
//                      No atomics: each wave owns 32 keys.
//
// 4 waves per workgroup; a wave owns 32 queries (fwd) / 32 keys (bwd); K/V (fwd) or (q+u)/dO (bwd) tiles are staged in
// LDS with the row XOR swizzle chunk ^= (row >> 1) & (chunks-1).  Head dims 32 and 64.
#include "attn_flash.h"
#include "rowops.h"
#include <stdlib.h>

void ttmi_probe_begin(int slot, hipStream_t st);
void ttmi_probe_end(int slot, hipStream_t st);

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr float NEGBIG = -1e30f;

__device__ __forceinline__ bf16x4 tr16(const char* lds_addr) {
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)lds_addr);
    return __builtin_bit_cast(bf16x4, v);
}

// MK = mask kind, a template parameter of the kernels: with a run-time switch the compiler kept ~30 scalar instructions of control
// flow per score element in the inner loops (the forward's instruction stream was mostly s_mov/s_and/s_cbranch)
template <int MK>
__device__ __forceinline__ bool is_masked(const FlashParams& p, int b, int i, int j) {
    if constexpr (MK == 1) return j > i;
    else if constexpr (MK == 2) return (j > i + p.mask_right) || (j < i - p.mask_left);
    else if constexpr (MK == 3) return p.mask[(long)b * p.mask_sb + (long)i * p.mask_si + j] != 0;
    else if constexpr (MK == 4) {                  // per-row key interval (the kernels keep lo / hi in registers or LDS instead)
        const int* r = reinterpret_cast<const int*>(p.mask) + (long)b * p.mask_sb + 2 * i;
        return j < r[0] || j > r[1];
    } else return false;
}

template <int DH>
struct Tile {                                    // [rows][DH] bf16 LDS image, 16-byte chunks XOR-swizzled per row
    static constexpr int NCH = DH / 8;
    static constexpr int ROWB = DH * 2;
    __device__ static __forceinline__ int off(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 1) & (NCH - 1))) << 4); }
};

// cooperative copy of `nrows` rows (global row r0 + row, clamped to rmax) of DH bf16 into a swizzled LDS tile
template <int DH>
__device__ __forceinline__ void stage_rows(char* lds, const bf16_t* g, long ld, int r0, int rmax, int nrows, int tid) {
    using T = Tile<DH>;
    for (int c = tid; c < nrows * T::NCH; c += 256) {
        const int row = c / T::NCH, ch = c % T::NCH;
        const int gr = min(r0 + row, rmax);
        const uint4 v = *reinterpret_cast<const uint4*>(g + (long)gr * ld + ch * 8);
        *reinterpret_cast<uint4*>(lds + T::off(row, ch)) = v;
    }
}

// register-staged variant: issue the global loads of the NEXT tile early, write them to LDS after the barrier that
// retires the current tile (T14 split).  NPT = 16-byte chunks per thread.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
template <int DH, int NROWS>
struct RowStage {
    using T = Tile<DH>;
    static constexpr int NPT = (NROWS * T::NCH + 255) / 256;
    u32x4_t v[NPT];   // native vector type: with HIP's uint4 struct the array stayed in scratch (load -> wait -> scratch)
    __device__ __forceinline__ void load(const bf16_t* g, long ld, int r0, int rmax, int tid) {
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
            const int c = tid + 256 * k;
            const int row = min(c / T::NCH, NROWS - 1), ch = c % T::NCH;
            v[k] = *reinterpret_cast<const u32x4_t*>(g + (long)min(r0 + row, rmax) * ld + ch * 8);
        }
    }
    __device__ __forceinline__ void store(char* lds, int tid) const {
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
            const int c = tid + 256 * k;
            if (c < NROWS * T::NCH) *reinterpret_cast<u32x4_t*>(lds + T::off(c / T::NCH, c % T::NCH)) = v[k];
        }
    }
};

__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int base) {
    bf16x8 r;
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = (__bf16)v[base + k];
    return r;
}

// transposed A-operand fragment of a [rows = reduction][cols = DH] tile: A[m = col c0 + (lane & 31)][k], k order
// = rows rbase + 8 (j >> 2) + 4 h + (j & 3), j = 0..7 (the order of an accumulator tile used as the B operand)
template <int DH>
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int rbase, int c0, int lane) {
    using T = Tile<DH>;
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3, h = g >> 1;
    const int col = c0 + 16 * (g & 1) + 4 * pp;
    const int r_lo = rbase + 4 * h + q, r_hi = r_lo + 8;
    const bf16x4 lo = tr16(tile + T::off(r_lo, col >> 3) + (col & 7) * 2);
    const bf16x4 hi = tr16(tile + T::off(r_hi, col >> 3) + (col & 7) * 2);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// ------------------------------------------------------------------ forward
#ifndef FWD_MINB
#define FWD_MINB 2
#endif
template <int DH, int MK>
__global__ __launch_bounds__(256, FWD_MINB) void flash_fwd_kernel(const FlashParams p) {
    using T = Tile<DH>;
    constexpr int KS = DH / 16, DT = DH / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ktile = smem;
    char* vtile = smem + 64 * T::ROWB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
    const int z = blockIdx.y, b = z / p.H, h = z % p.H;
    const int L = p.L;
    const int i = blockIdx.x * 128 + wave * 32 + (lane & 31);
    const int ic = min(i, L - 1);
    const bf16_t* qrow = p.qu + ((long)b * L + ic) * p.ld_qu + h * DH;
    bf16x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qrow + 16 * ks + 8 * hh);
    f32x16 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m = NEGBIG, l = 0.f;
    int mlo = 0, mhi = 0x7fffffff;                 // MK == 4: this lane's query row allows keys mlo..mhi
    if constexpr (MK == 4) {
        const int* r = reinterpret_cast<const int*>(p.mask) + (long)b * p.mask_sb + 2 * ic;
        mlo = r[0];
        mhi = r[1];
    }
    const bf16_t* bd_row = p.bd + (long)z * p.slab + (long)ic * L;
    const bf16_t* kbase = p.k + (long)b * L * p.ld_kv + h * DH;
    const bf16_t* vbase = p.v + (long)b * L * p.ld_kv + h * DH;

    // bias of a 64-key tile x the workgroup's 128 queries = 128 rows of 128 contiguous bytes: fetched one tile ahead with 8-byte loads
    // (8 per thread), parked in LDS behind the K / V tiles, read back per lane (own query row; keys jb + 4 hh + 8 g + (0..3)) with four
    // ds_read_b64.  (Per-lane row-strided global loads touched 64 cache lines per instruction.)
    constexpr int BP = 68;                                                       // row pitch (bf16): 136 B, so 32 query rows hit 32 distinct bank pairs
    bf16_t* btile = reinterpret_cast<bf16_t*>(smem + 2 * 64 * T::ROWB);          // [128][BP] bf16
    const bf16_t* bdz = p.bd + (long)z * p.slab;
    const bool bstage = (L % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.bd) & 7) == 0) && (p.slab % 4 == 0) && L >= 4 && !(p.debug & 1);
    uint2 bpre[8];
    auto fetch_bias = [&](int j0) {
        if (!bstage) return;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = tid + 256 * k;                                         // row c >> 4, columns 4 (c & 15)
            const int row = min((int)blockIdx.x * 128 + (c >> 4), L - 1), col = min(j0 + 4 * (c & 15), L - 4);
            bpre[k] = *reinterpret_cast<const uint2*>(bdz + (unsigned)(row * L + col));
        }
    };
    auto park_bias = [&]() {
        if (!bstage) return;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = tid + 256 * k;
            *reinterpret_cast<uint2*>(btile + (c >> 4) * BP + 4 * (c & 15)) = bpre[k];
        }
    };
    auto read_bias = [&](int jb, int sub, float (&bv)[16]) {
        if (p.debug & 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) bv[r] = 0.f;
        } else if (bstage) {
            const bf16_t* rowp = btile + (wave * 32 + (lane & 31)) * BP + sub * 32 + 4 * hh;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const uint2 t = *reinterpret_cast<const uint2*>(rowp + 8 * g4);
                bv[4 * g4] = __uint_as_float(t.x << 16); bv[4 * g4 + 1] = __uint_as_float(t.x & 0xffff0000u);
                bv[4 * g4 + 2] = __uint_as_float(t.y << 16); bv[4 * g4 + 3] = __uint_as_float(t.y & 0xffff0000u);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) bv[r] = bf16_to_f32(bd_row[min(jb + (r & 3) + 8 * (r >> 2) + 4 * hh, L - 1)]);
        }
    };
    auto sub_step = [&](int jb, int sub) {
        float bcur[16];
        read_bias(jb, sub, bcur);
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ktile + T::off(32 * sub + (lane & 31), 2 * ks + hh));
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
        }
        float pmax = NEGBIG;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = jb + (r & 3) + 8 * (r >> 2) + 4 * hh;
            float v = NEGBIG;
            const bool msk = MK == 4 ? (j < mlo || j > mhi) : is_masked<MK>(p, b, ic, j);
            if (j < L && !msk) v = (s[r] + bcur[r]) * p.scale;
            s[r] = v;
            pmax = fmaxf(pmax, v);
        }
        pmax = fmaxf(pmax, __shfl_xor(pmax, 32, 64));
        const float mn = fmaxf(m, pmax);
        const float alpha = __expf(m - mn);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pr = s[r] > 0.5f * NEGBIG ? __expf(s[r] - mn) : 0.f;
            s[r] = pr;
            psum += pr;
        }
        l = l * alpha + psum;
        m = mn;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 pb = pack8(s, 8 * s2);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const bf16x8 vf = tr_frag<DH>(vtile, 32 * sub + 16 * s2, 32 * dt, lane);
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb, o[dt], 0, 0, 0);
            }
        }
    };
    // structured masks: key tiles that are masked for all 128 queries of the workgroup contribute exact zeros (p = 0, the running maximum
    // and sum unchanged) and are skipped - a band of left + right + 1 keys touches (left + right + 128) / 64 + 1 tiles instead of L / 64
    int jbeg = 0, jend = L;
    {
        const int i0w = blockIdx.x * 128, i1w = min(i0w + 127, L - 1);
        if constexpr (MK == 1) jend = i1w + 1;
        if constexpr (MK == 2) {
            jbeg = max(0, i0w - p.mask_left);
            jend = (int)min((long)L, (long)i1w + p.mask_right + 1);
        }
        if constexpr (MK == 4) {
            __shared__ int rng[2];
            if (tid == 0) { rng[0] = 0x7fffffff; rng[1] = -1; }
            __syncthreads();
            atomicMin(&rng[0], mlo);
            atomicMax(&rng[1], mhi);
            __syncthreads();
            jbeg = max(0, rng[0]);
            jend = (int)min((long)L, (long)rng[1] + 1);
        }
        if (jend <= jbeg) { jbeg = 0; jend = L; }                   // nothing visible at all: keep the unskipped behaviour
        jbeg &= ~63;
    }
    RowStage<DH, 64> stK, stV;
    stK.load(kbase, p.ld_kv, jbeg, L - 1, tid);
    stV.load(vbase, p.ld_kv, jbeg, L - 1, tid);
    fetch_bias(jbeg);
    for (int j0 = jbeg; j0 < jend; j0 += 64) {
        __syncthreads();                                            // everyone is done reading the previous tile
        stK.store(ktile, tid);
        stV.store(vtile, tid);
        park_bias();
        __syncthreads();
        if (j0 + 64 < jend) {                                       // next tile's loads fly under this tile's MFMAs
            stK.load(kbase, p.ld_kv, j0 + 64, L - 1, tid);
            stV.load(vbase, p.ld_kv, j0 + 64, L - 1, tid);
            fetch_bias(j0 + 64);
        }
        sub_step(j0, 0);
        if (j0 + 32 < jend) sub_step(j0 + 32, 1);
    }
    l += __shfl_xor(l, 32, 64);
    if (i < L) {
        const float inv = 1.f / l;
        bf16_t* orow = p.o + ((long)b * L + i) * p.ld_o + h * DH;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = 32 * dt + 8 * g4 + 4 * hh;
                uint2 w;
                w.x = pack_bf16x2(o[dt][4 * g4] * inv, o[dt][4 * g4 + 1] * inv);
                w.y = pack_bf16x2(o[dt][4 * g4 + 2] * inv, o[dt][4 * g4 + 3] * inv);
                *reinterpret_cast<uint2*>(orow + d) = w;
            }
        if (hh == 0) p.lse[(long)z * L + i] = m + __logf(l);
    }
}

// ------------------------------------------------------------------ forward with the position term formed in the kernel
// BD[i][j] (tt/transformer.py:143-149 incl. _rel_shift) depends on (i, j) only through p' = L-1-i+j and on WHICH query row multiplies the
// table: with the extended table Eext[p'] = E[p'] (p' <= L-1), 0 (p' = L), E[p'-L-1] (p' >= L+1) and cext likewise,
//     BD[i][j] = (p' <= L-1 ? q_i : q_{i+1}) . Eext[p'] + cext[p'],        p' = L-1-i+j
// (lower triangle incl. diagonal / forced zero at j = i+1 / the shift's wrap-around above it).  For a wave's 32
// queries and a 64-key tile the p' values form ONE window of 95 table rows: G^T = Eext_window . q^T is three 32x32 MFMA blocks (rows = p',
// columns = queries; blocks entirely below L use q_i, entirely above L use q_{i+1}, the one block that straddles L takes both products with
// the other side's table rows zeroed), and the skew p' -> j is a read of the block through a private LDS image at [query][31 - ii + jj].
// No [B, H, L, L] slab exists: the table window (192 rows for the workgroup's 128 queries) is staged in LDS next to the K / V tiles.
template <int DH, int MK>
__global__ __launch_bounds__(256, 2) void flash_fwd_rel_kernel(const FlashParams p) {
    using T = Tile<DH>;
    constexpr int KS = DH / 16, DT = DH / 32;
    constexpr int EROWS = 256, GP = 100;                       // table ring of 256 rows (slot = p' & 255): a tile's window is 192 rows, each key tile adds 64
                                                               // Gs row pitch in bf16 (96 window columns + 4): 200 bytes
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ktile = smem;
    char* vtile = smem + 64 * T::ROWB;
    char* etile = smem + 128 * T::ROWB;                        // [192][DH] swizzled like the K tile
    float* ctile = reinterpret_cast<float*>(smem + (128 + EROWS) * T::ROWB);        // cext of the window
    bf16_t* gs_all = reinterpret_cast<bf16_t*>(ctile + EROWS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, ii = lane & 31;
    bf16_t* gs = gs_all + wave * 32 * GP + ii * GP;            // this lane's query row of the wave's private image
    const int z = blockIdx.y, b = z / p.H, h = z % p.H;
    const int L = p.L;
    const float c2 = p.scale * 1.4426950408889634f;            // scale * log2(e)
    const int i0w = blockIdx.x * 128;
    const int i = i0w + wave * 32 + ii;
    const int ic = min(i, L - 1);
    const bf16_t* prow = p.qp + ((long)b * L + ic) * p.ld_qp + h * DH;
    const bf16_t* prow1 = p.qp + ((long)b * L + min(ic + 1, L - 1)) * p.ld_qp + h * DH;
    bf16x8 qf[KS], qp[KS], qp1[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        qp[ks] = *reinterpret_cast<const bf16x8*>(prow + 16 * ks + 8 * hh);
        qp1[ks] = *reinterpret_cast<const bf16x8*>(prow1 + 16 * ks + 8 * hh);
        // q + r_w_bias (the content term's query, tt/transformer.py:140) formed here: f32 add of the bf16 q, rounded to bf16 once - the values
        // a separate add_row_bias pass used to store as a [B*L, H*Dh] tensor
        const float* uh = p.u + h * DH + 16 * ks + 8 * hh;
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[ks][e] = (__bf16)((float)qp[ks][e] + uh[e]);
    }
    f32x16 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m = NEGBIG, l = 0.f;
    int mlo = 0, mhi = 0x7fffffff;
    if constexpr (MK == 4) {
        const int* r = reinterpret_cast<const int*>(p.mask) + (long)b * p.mask_sb + 2 * ic;
        mlo = r[0];
        mhi = r[1];
    }
    const bf16_t* kbase = p.k + (long)b * L * p.ld_kv + h * DH;
    const bf16_t* vbase = p.v + (long)b * L * p.ld_kv + h * DH;
    const bf16_t* ebase = p.e16 + h * DH;
    const float* cbase = p.cT + (long)h * L;

    // table window of a key tile: rows p' = wbase .. wbase + 191, wbase = L - 128 - i0w + j0 (wave w reads rows 96 - 32 w .. + 95 of it),
    // kept in a ring: row p' lives in slot (p' - L) & 255; the first tile stages all 192 rows, every later tile only its 64 new ones
    u32x4_t epre[2];                                           // the 64 new rows of the next tile (DH = 64: 512 chunks, 2 per thread)
    float cpre = 0.f;
    auto ext_row = [&](int pe, int& src) -> bool {             // extended-table row p' -> source row of E / c, false = zero row
        src = pe < L ? pe : pe - L - 1;
        const bool ok = pe >= 0 && pe != L && src < L;
        src = min(max(src, 0), L - 1);
        return ok;
    };
    auto load_chunk = [&](int pe, int ch) -> u32x4_t {
        int src;
        const bool ok = ext_row(pe, src);
        u32x4_t v = *reinterpret_cast<const u32x4_t*>(ebase + (long)src * p.ld_e + ch * 8);
        if (!ok) v = u32x4_t{0u, 0u, 0u, 0u};
        return v;
    };
    auto load_c = [&](int pe) -> float {
        int src;
        const bool ok = ext_row(pe, src);
        const float cv = cbase[src];
        return ok ? cv : 0.f;
    };
    auto stage_window = [&](int j0) {                          // prologue: the whole first window, straight to LDS
        const int wbase = L - 128 - i0w + j0;
        for (int c = tid; c < 192 * T::NCH; c += 256) {
            const int pe = wbase + c / T::NCH, ch = c % T::NCH;
            *reinterpret_cast<u32x4_t*>(etile + T::off((pe - L) & 255, ch)) = load_chunk(pe, ch);
        }
        if (tid < 192) ctile[(wbase + tid - L) & 255] = load_c(wbase + tid);
    };
    auto fetch_e = [&](int j0) {                               // rows wbase(j0) + 128 .. + 191: the part of tile j0's window the previous tile did not have
        const int wnew = L - 128 - i0w + j0 + 128;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int c = tid + 256 * k;
            if (c < 64 * T::NCH) epre[k] = load_chunk(wnew + c / T::NCH, c % T::NCH);
        }
        if (tid < 64) cpre = load_c(wnew + tid);
    };
    auto park_e = [&](int j0) {
        const int wnew = L - 128 - i0w + j0 + 128;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int c = tid + 256 * k;
            if (c < 64 * T::NCH) *reinterpret_cast<u32x4_t*>(etile + T::off((wnew + c / T::NCH - L) & 255, c % T::NCH)) = epre[k];
        }
        if (tid < 64) ctile[(wnew + tid - L) & 255] = cpre;
    };
    const int eoff = 96 - 32 * wave;                           // this wave's first row inside the window
    // G^T blocks of the tile -> the wave's private image gs[query][window column]
    auto position = [&](int j0) {
        if (p.debug & 1) return;
        asm volatile("" ::: "memory");                         // (the image is written as uint2 and read as bf16: keep the compiler from reordering across)
        const int pe_w = L - 128 - i0w + j0 + eoff;            // p' of the wave's window row 0
        // cext of the block's window columns is the accumulators' INITIAL value (the chain adds E . q onto it: no add per element afterwards)
        auto cinit = [&](int blk, f32x16& g) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int col = 32 * blk + 8 * g4 + 4 * hh;
                const float4 cv = *reinterpret_cast<const float4*>(ctile + ((pe_w + col - L) & 255));     // slots count from p' - L: a multiple of 4 here
                g[4 * g4] = cv.x; g[4 * g4 + 1] = cv.y; g[4 * g4 + 2] = cv.z; g[4 * g4 + 3] = cv.w;
            }
        };
        auto emit = [&](int blk, const f32x16& g) {            // to bf16, into this lane's image row: window columns 32 blk + 8 g4 + 4 hh + (0..3)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int col = 32 * blk + 8 * g4 + 4 * hh;
                uint2 w;
                w.x = pack_bf16x2(g[4 * g4], g[4 * g4 + 1]);
                w.y = pack_bf16x2(g[4 * g4 + 2], g[4 * g4 + 3]);
                *reinterpret_cast<uint2*>(gs + col) = w;
            }
        };
        const bool all_low = pe_w + 95 <= L - 1, all_up = pe_w >= L + 1;
        if (all_low || all_up) {
            // the common case (the tile is wholly below or wholly above the j = i + 1 diagonal): three independent accumulator chains, their
            // MFMAs interleaved - a block's four dependent MFMAs alone leave the pipe idle for the accumulator latency
            f32x16 g0, g1, g2;
            cinit(0, g0);
            cinit(1, g1);
            cinit(2, g2);
            const int e0 = (pe_w + ii - L) & 255, e1 = (pe_w + 32 + ii - L) & 255, e2 = (pe_w + 64 + ii - L) & 255;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 qs = all_low ? qp[ks] : qp1[ks];
                const bf16x8 f0 = *reinterpret_cast<const bf16x8*>(etile + T::off(e0, 2 * ks + hh));
                const bf16x8 f1 = *reinterpret_cast<const bf16x8*>(etile + T::off(e1, 2 * ks + hh));
                const bf16x8 f2 = *reinterpret_cast<const bf16x8*>(etile + T::off(e2, 2 * ks + hh));
                g0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, qs, g0, 0, 0, 0);
                g1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, qs, g1, 0, 0, 0);
                g2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f2, qs, g2, 0, 0, 0);
            }
            emit(0, g0);
            emit(1, g1);
            emit(2, g2);
        } else {
#pragma unroll 1
        for (int blk = 0; blk < 3; ++blk) {
            const int pe0 = pe_w + 32 * blk;
            f32x16 g;
            cinit(blk, g);
            const int erow = (pe0 + ii - L) & 255;             // ring slot of this lane's table row
            if (pe0 + 31 <= L - 1) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 ef = *reinterpret_cast<const bf16x8*>(etile + T::off(erow, 2 * ks + hh));
                    g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ef, qp[ks], g, 0, 0, 0);
                }
            } else if (pe0 >= L + 1) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 ef = *reinterpret_cast<const bf16x8*>(etile + T::off(erow, 2 * ks + hh));
                    g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ef, qp1[ks], g, 0, 0, 0);
                }
            } else {                                           // the block that holds p' = L: rows below it take q_i, rows above it q_{i+1}
                const bool lower = pe0 + ii <= L - 1;
                bf16x8 zero;
#pragma unroll
                for (int e = 0; e < 8; ++e) zero[e] = (__bf16)0.0f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 ef = *reinterpret_cast<const bf16x8*>(etile + T::off(erow, 2 * ks + hh));
                    g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lower ? ef : zero, qp[ks], g, 0, 0, 0);
                    g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lower ? zero : ef, qp1[ks], g, 0, 0, 0);
                }
            }
            emit(blk, g);
        }
        }
        asm volatile("" ::: "memory");                         // DS operations of one wave execute in order: the reads below see these writes
    };
    auto sub_step = [&](int jb, int sub) {
        f32x16 s;                                                         // the position term initialises the score accumulator
        if (p.debug & (1 | 128)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
        } else {
            const bf16_t* gr = gs + 31 - ii + 32 * sub + 4 * hh;          // key jj of the tile sits at window column 31 - ii + jj
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = bf16_to_f32(gr[(r & 3) + 8 * (r >> 2)]);
        }
        if (!(p.debug & 32)) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ktile + T::off(32 * sub + (lane & 31), 2 * ks + hh));
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
        }
        }
        // The running maximum m is kept in log2 units (score * scale * log2 e) and is only RAISED when a sub-tile beats it by more than 8
        // (P then stays below 2^8: harmless in f32 sums and in the bf16 operand, whose rounding is relative): most sub-tiles rescale nothing.
        const bool plain = MK == 0 && jb + 32 <= L;                 // wave-uniform: no key of the sub-tile is masked or beyond the sequence
        float pmax = NEGBIG;
        if (plain) {
#pragma unroll
            for (int r = 0; r < 16; ++r) pmax = fmaxf(pmax, s[r]);
            pmax *= c2;
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = jb + (r & 3) + 8 * (r >> 2) + 4 * hh;
                float v = NEGBIG;
                const bool msk = MK == 4 ? (j < mlo || j > mhi) : is_masked<MK>(p, b, ic, j);
                if (j < L && !msk) v = s[r] * c2;
                s[r] = v;
                pmax = fmaxf(pmax, v);
            }
        }
        pmax = fmaxf(pmax, __shfl_xor(pmax, 32, 64));
        const float mn = pmax > m + 8.f ? pmax : m;
        if (__builtin_amdgcn_ballot_w64(mn != m)) {
            const float alpha = __builtin_amdgcn_exp2f(m - mn);
            l *= alpha;
            m = mn;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        }
        float psum = 0.f;
        if (p.debug & 8) {
#pragma unroll
            for (int r = 0; r < 16; ++r) psum += s[r];
        } else if (plain) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pr = __builtin_amdgcn_exp2f(fmaf(s[r], c2, -m));
                s[r] = pr;
                psum += pr;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pr = s[r] > 0.5f * NEGBIG ? __builtin_amdgcn_exp2f(s[r] - m) : 0.f;
                s[r] = pr;
                psum += pr;
            }
        }
        l += psum;
        if (p.debug & 16) {
#pragma unroll
            for (int r = 0; r < 16; ++r) o[0][r] += s[r];
            return;
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 pb = pack8(s, 8 * s2);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const bf16x8 vf = tr_frag<DH>(vtile, 32 * sub + 16 * s2, 32 * dt, lane);
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb, o[dt], 0, 0, 0);
            }
        }
    };
    int jbeg = 0, jend = L;
    {
        const int i1w = min(i0w + 127, L - 1);
        if constexpr (MK == 1) jend = i1w + 1;
        if constexpr (MK == 2) {
            jbeg = max(0, i0w - p.mask_left);
            jend = (int)min((long)L, (long)i1w + p.mask_right + 1);
        }
        if constexpr (MK == 4) {
            __shared__ int rng[2];
            if (tid == 0) { rng[0] = 0x7fffffff; rng[1] = -1; }
            __syncthreads();
            atomicMin(&rng[0], mlo);
            atomicMax(&rng[1], mhi);
            __syncthreads();
            jbeg = max(0, rng[0]);
            jend = (int)min((long)L, (long)rng[1] + 1);
        }
        if (jend <= jbeg) { jbeg = 0; jend = L; }
        jbeg &= ~63;
    }
    RowStage<DH, 64> stK, stV;
    stK.load(kbase, p.ld_kv, jbeg, L - 1, tid);
    stV.load(vbase, p.ld_kv, jbeg, L - 1, tid);
    stage_window(jbeg - 64);                                   // rows wbase(jbeg) - 64 .. + 127; the loop's first park adds + 128 .. + 191
    fetch_e(jbeg);
    for (int j0 = jbeg; j0 < jend; j0 += 64) {
        __syncthreads();
        if (!(p.debug & 64)) {
            stK.store(ktile, tid);
            stV.store(vtile, tid);
            park_e(j0);
        }
        __syncthreads();
        if (j0 + 64 < jend && !(p.debug & 256)) {
            stK.load(kbase, p.ld_kv, j0 + 64, L - 1, tid);
            stV.load(vbase, p.ld_kv, j0 + 64, L - 1, tid);
            fetch_e(j0 + 64);
        }
        position(j0);
        sub_step(j0, 0);
        if (j0 + 32 < jend) sub_step(j0 + 32, 1);
    }
    l += __shfl_xor(l, 32, 64);
    if (i < L) {
        const float inv = 1.f / l;
        bf16_t* orow = p.o + ((long)b * L + i) * p.ld_o + h * DH;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = 32 * dt + 8 * g4 + 4 * hh;
                uint2 w;
                w.x = pack_bf16x2(o[dt][4 * g4] * inv, o[dt][4 * g4 + 1] * inv);
                w.y = pack_bf16x2(o[dt][4 * g4 + 2] * inv, o[dt][4 * g4 + 3] * inv);
                *reinterpret_cast<uint2*>(orow + d) = w;
            }
        if (hh == 0) p.lse[(long)z * L + i] = m * 0.6931471805599453f + __logf(l);
    }
}

// ------------------------------------------------------------------ forward, ONE workgroup per (b, h) with the position table resident in LDS
// (round 4).  flash_fwd_rel_kernel above cuts a head into four 128-query workgroups that each stream all K / V tiles (PMC: 149 MB fetched for 49 MB of
// q, k, v) through a loop of 64-key steps with two barriers and a one-step register prefetch per step.  Here a head is one 512-thread workgroup
// (B*H = 256 heads = one per CU at C2 / C4, a single round): 8 waves x 64 queries (two 32-query blocks per wave, one after the other), the head's
// whole effective table E (L rows + a zero row, <= 66 KB, fetched by LDS-DMA) and its extended bias stay in LDS for the whole kernel, K / V arrive
// in PHASES of 128 keys (register prefetch one phase ahead, 2 barriers per phase), and inside a phase a wave runs 2 key tiles = 56 MFMAs with no
// synchronisation at all.  Same arithmetic per (query block, key tile) as flash_fwd_rel_kernel (same operands, same accumulation order over
// Dh; the online softmax visits the key tiles in the same order): bit-identical outputs.  Dh = 64, L <= 512.
// Measured on the way (C2 audio layer, tools/exp_attn.sh): the tiled kernel 99 us; the first resident version 87 us, of which 30 us were its
// skeleton (the table copied through registers in a dependent loop, q fragments reloaded at 8 block switches, row-strided output stores) and
// 17 us the image read - the compiler had merged the sixteen 2-byte reads of a sub-tile into four ds_read_b64 at 2-byte alignment, which the
// LDS executes several times slower than aligned ones.
constexpr int RES_PH = 128;                                     // keys per phase
constexpr int RES_GP = 100;                                     // image row pitch in bf16 (96 window columns + 4)
constexpr int RES_EROWS = 513;                                  // table rows in LDS: L <= 512 rows + the zero row
constexpr int RES_CX = 1028;                                    // extended bias: index p' - L + 512, p' - L in [-512, 515]
constexpr int RES_LDS = RES_EROWS * 128 + 2 * RES_PH * 128 + RES_CX * 4 + 64 * 4 + 8 * 32 * RES_GP * 2;
static_assert(RES_LDS <= 160 * 1024, "flash_fwd_res_kernel: LDS budget");
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk2(float lo, float hi) {               // ONE v_cvt_pk_bf16_f32 (pack_bf16x2 converts each half and ORs)
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}
// bf16 at an LDS address -> f32 in ONE aligned 2-byte read: ds_read_u16_d16_hi puts the halfword into bits [31:16] of a register that holds
// zero, which IS the float
__device__ __forceinline__ float lds_bf16_as_f32(const bf16_t* lds_addr) {
    float v = 0.f;
    asm volatile("ds_read_u16_d16_hi %0, %1" : "+v"(v) : "v"((unsigned)(size_t)lds_addr));
    return v;
}
template <int MK>
__global__ __launch_bounds__(512, 2) void flash_fwd_res_kernel(const FlashParams p) {
    constexpr int DH = 64;
    using T = Tile<DH>;
    constexpr int KS = DH / 16, DT = DH / 32, GP = RES_GP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* etile = smem;                                         // [L + 1][DH], row L = zeros
    char* ktile = smem + RES_EROWS * T::ROWB;                   // [128][DH]
    char* vtile = ktile + RES_PH * T::ROWB;
    float* cx = reinterpret_cast<float*>(vtile + RES_PH * T::ROWB);
    float* u_s = cx + RES_CX;                                   // r_w_bias of this head
    bf16_t* gs_all = reinterpret_cast<bf16_t*>(u_s + 64);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, ii = lane & 31;
    bf16_t* gs_w = gs_all + wave * 32 * GP;                     // the wave's private image [32 queries][GP]
    bf16_t* gs = gs_w + ii * GP;                                // this lane's query row of it
    const int z = blockIdx.x, b = z / p.H, h = z % p.H;
    const int L = p.L;
    const float c2 = p.scale * 1.4426950408889634f;             // scale * log2(e)
    const bf16_t* kbase = p.k + (long)b * L * p.ld_kv + h * DH;
    const bf16_t* vbase = p.v + (long)b * L * p.ld_kv + h * DH;
    const bf16_t* ebase = p.e16 + h * DH;
    const float* cbase = p.cT + (long)h * L;
    const bf16_t* qbase = p.qp + (long)b * L * p.ld_qp + h * DH;

    // ---- K / V of phase 0 into registers; the table by LDS-DMA (1 KiB = 8 rows per wave-instruction, swizzle applied to the SOURCE chunk);
    // its last L % 8 rows, the zero row and the bias through registers
    u32x4_t kpre[2], vpre[2];                                   // 128 rows x 8 chunks = 1024 chunks per operand: 2 per thread
    auto fetch_kv = [&](int j0) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int c = tid + 512 * k;
            const long off = (long)min(j0 + c / T::NCH, L - 1) * p.ld_kv + (c % T::NCH) * 8;
            kpre[k] = *reinterpret_cast<const u32x4_t*>(kbase + off);
            vpre[k] = *reinterpret_cast<const u32x4_t*>(vbase + off);
        }
    };
    auto park_kv = [&]() {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int c = tid + 512 * k;
            *reinterpret_cast<u32x4_t*>(ktile + T::off(c / T::NCH, c % T::NCH)) = kpre[k];
            *reinterpret_cast<u32x4_t*>(vtile + T::off(c / T::NCH, c % T::NCH)) = vpre[k];
        }
    };
    const int npiece = L / 8;
    for (int pc = wave; pc < npiece; pc += 8) {                 // (wave-uniform piece index: the LDS destination is M0 + lane * 16)
        const int row = pc * 8 + (lane >> 3), cdst = lane & 7;
        const int csrc = cdst ^ ((row >> 1) & (T::NCH - 1));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ebase + (long)row * p.ld_e + csrc * 8),
                                         (__attribute__((address_space(3))) void*)(etile + pc * 1024), 16, 0, 0);
    }
    fetch_kv(0);
    if (tid < (L + 1 - npiece * 8) * T::NCH) {
        const int row = npiece * 8 + tid / T::NCH, ch = tid % T::NCH;
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (row < L) v = *reinterpret_cast<const u32x4_t*>(ebase + (long)row * p.ld_e + ch * 8);
        *reinterpret_cast<u32x4_t*>(etile + T::off(row, ch)) = v;
    }
    {
        float cv[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {                            // cext[p'] at index p' - L + 512: c[p'] (p' < L), 0 (p' = L), c[p' - L - 1] (p' > L)
            const int c = tid + 512 * k, pe = c - 512 + L;
            const int src = pe < L ? pe : pe - L - 1;
            const bool ok = pe >= 0 && pe != L && src < L;
            cv[k] = cbase[ok ? src : 0];
            if (!ok) cv[k] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (tid + 512 * k < RES_CX) cx[tid + 512 * k] = cv[k];
    }
    if (tid < DH) u_s[tid] = p.u[h * DH + tid];                 // (q + u is formed when a block starts)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the table pieces have landed (the loop's first barrier publishes them)
    // extended-table row p' -> LDS row (L = the zero row)
    auto erow_of = [&](int pe) -> int {
        const int src = pe <= L ? pe : pe - L - 1;
        return (pe >= 0 && src < L) ? src : L;
    };

    const int nph = (L + RES_PH - 1) / RES_PH;
#pragma unroll 1
    for (int qb = 0; qb < 2; ++qb) {
        // ---- this pass's query block of the wave: 32 queries i0 .. i0 + 31
        const int i0 = wave * 64 + qb * 32;
        const bool live = i0 < L;                               // (wave-uniform)
        const int i = i0 + ii, ic = min(i, L - 1);
        f32x16 o[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
        float mq = NEGBIG, lq = 0.f;
        // key range this block's queries can see (wave-uniform), from the mask parameters
        int mlo = 0, mhi = 0x7fffffff;
        int jbeg = 0, jend = L;
        if constexpr (MK == 1) jend = min(L, i0 + 32);
        if constexpr (MK == 2) {
            jbeg = max(0, i0 - p.mask_left);
            jend = (int)min((long)L, (long)min(i0 + 31, L - 1) + p.mask_right + 1);
        }
        if constexpr (MK == 4) {
            const int* r = reinterpret_cast<const int*>(p.mask) + (long)b * p.mask_sb + 2 * ic;
            mlo = r[0];
            mhi = r[1];
            int lo = mlo, hi = mhi;
#pragma unroll
            for (int sft = 16; sft >= 1; sft >>= 1) {
                lo = min(lo, __shfl_xor(lo, sft, 64));
                hi = max(hi, __shfl_xor(hi, sft, 64));
            }
            jbeg = max(0, lo);
            jend = (int)min((long)L, (long)hi + 1);
        }
        if (jend <= jbeg) { jbeg = 0; jend = L; }
        jbeg &= ~63;
        const bf16_t* prow = qbase + (long)ic * p.ld_qp;
        const bf16_t* prow1 = qbase + (long)min(ic + 1, L - 1) * p.ld_qp;
        bf16x8 qf[KS], qp[KS], qp1[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qp[ks] = *reinterpret_cast<const bf16x8*>(prow + 16 * ks + 8 * hh);
            qp1[ks] = *reinterpret_cast<const bf16x8*>(prow1 + 16 * ks + 8 * hh);
        }

        // G^T blocks of a 64-key tile -> the wave's private image gs[query][window column]; window column 0 is p' = L - 32 - i0 + j0
        auto cinit = [&](int pe_w, int blk, f32x16& g) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int col = 32 * blk + 8 * g4 + 4 * hh;
                const float4 cv = *reinterpret_cast<const float4*>(cx + (pe_w + col - L + 512));     // (pe_w - L is a multiple of 32)
                g[4 * g4] = cv.x; g[4 * g4 + 1] = cv.y; g[4 * g4 + 2] = cv.z; g[4 * g4 + 3] = cv.w;
            }
        };
        auto emit = [&](int blk, const f32x16& g) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int col = 32 * blk + 8 * g4 + 4 * hh;
                uint2 w;
                w.x = cvt_pk2(g[4 * g4], g[4 * g4 + 1]);
                w.y = cvt_pk2(g[4 * g4 + 2], g[4 * g4 + 3]);
                *reinterpret_cast<uint2*>(gs + col) = w;
            }
        };
        auto one_side = [&](int pe_w, const bf16x8 (&qs)[KS]) {       // the whole window on one side of p' = L: three plain chains
            f32x16 ga, gb;
            auto chain = [&](int blk, f32x16& g) {
                cinit(pe_w, blk, g);
                const int er = erow_of(pe_w + 32 * blk + ii);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 f = *reinterpret_cast<const bf16x8*>(etile + T::off(er, 2 * ks + hh));
                    g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, qs[ks], g, 0, 0, 0);
                }
            };
            chain(0, ga);
            chain(1, gb);
            emit(0, ga);
            chain(2, ga);
            emit(1, gb);
            emit(2, ga);
        };
        auto position = [&](int j0) {
            if (p.debug & 1) return;
            asm volatile("" ::: "memory");
            const int pe_w = L - 32 - i0 + j0;
            if (pe_w + 95 <= L - 1) one_side(pe_w, qp);
            else if (pe_w >= L + 1) one_side(pe_w, qp1);
            else {
#pragma unroll 1
                for (int blk = 0; blk < 3; ++blk) {
                    const int pe0 = pe_w + 32 * blk;
                    f32x16 g;
                    cinit(pe_w, blk, g);
                    const int erow = erow_of(pe0 + ii);
                    if (pe0 + 31 <= L - 1) {
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) {
                            const bf16x8 ef = *reinterpret_cast<const bf16x8*>(etile + T::off(erow, 2 * ks + hh));
                            g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ef, qp[ks], g, 0, 0, 0);
                        }
                    } else if (pe0 >= L + 1) {
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) {
                            const bf16x8 ef = *reinterpret_cast<const bf16x8*>(etile + T::off(erow, 2 * ks + hh));
                            g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ef, qp1[ks], g, 0, 0, 0);
                        }
                    } else {                                   // the block that holds p' = L: rows below it take q_i, rows above it q_{i+1}
                        const bool lower = pe0 + ii <= L - 1;
                        bf16x8 zero;
#pragma unroll
                        for (int e = 0; e < 8; ++e) zero[e] = (__bf16)0.0f;
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) {
                            const bf16x8 ef = *reinterpret_cast<const bf16x8*>(etile + T::off(erow, 2 * ks + hh));
                            g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lower ? ef : zero, qp[ks], g, 0, 0, 0);
                            g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lower ? zero : ef, qp1[ks], g, 0, 0, 0);
                        }
                    }
                    emit(blk, g);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the image is read back by asm loads the compiler does not order against these writes
        };
        // 32 keys jb .. jb + 31 (rows kr0 .. of the phase's K / V tiles) against the block's 32 queries; sub = which half of the 64-key window
        auto sub_step = [&](int jb, int kr0, int sub) {
            f32x16 s;
            if (p.debug & (1 | 128)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = 0.f;
            } else {
                const bf16_t* gr = gs + 31 - ii + 32 * sub + 4 * hh;  // key jj of the tile sits at window column 31 - ii + jj
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = lds_bf16_as_f32(gr + (r & 3) + 8 * (r >> 2));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            if (!(p.debug & 32)) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ktile + T::off(kr0 + ii, 2 * ks + hh));
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
                }
            }
            const bool plain = MK == 0 && jb + 32 <= L;       // wave-uniform: no key of the sub-tile is masked or beyond the sequence
            float pmax = NEGBIG;
            if (plain) {
#pragma unroll
                for (int r = 0; r < 16; ++r) pmax = fmaxf(pmax, s[r]);
                pmax *= c2;
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int j = jb + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    float v = NEGBIG;
                    const bool msk = MK == 4 ? (j < mlo || j > mhi) : is_masked<MK>(p, b, ic, j);
                    if (j < L && !msk) v = s[r] * c2;
                    s[r] = v;
                    pmax = fmaxf(pmax, v);
                }
            }
            pmax = fmaxf(pmax, __shfl_xor(pmax, 32, 64));
            const float mn = pmax > mq + 8.f ? pmax : mq;
            if (__builtin_amdgcn_ballot_w64(mn != mq)) {
                const float alpha = __builtin_amdgcn_exp2f(mq - mn);
                lq *= alpha;
                mq = mn;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            }
            float psum = 0.f;
            if (p.debug & 8) {
#pragma unroll
                for (int r = 0; r < 16; ++r) psum += s[r];
            } else if (plain) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pr = __builtin_amdgcn_exp2f(fmaf(s[r], c2, -mq));
                    s[r] = pr;
                    psum += pr;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pr = s[r] > 0.5f * NEGBIG ? __builtin_amdgcn_exp2f(s[r] - mq) : 0.f;
                    s[r] = pr;
                    psum += pr;
                }
            }
            lq += psum;
            if (p.debug & 16) {
#pragma unroll
                for (int r = 0; r < 16; ++r) o[0][r] += s[r];
                return;
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pb = pack8(s, 8 * s2);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const bf16x8 vf = tr_frag<DH>(vtile, kr0 + 16 * s2, 32 * dt, lane);
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb, o[dt], 0, 0, 0);
                }
            }
        };

#pragma unroll 1
        for (int ph = 0; ph < nph; ++ph) {
            __syncthreads();                                    // every wave is done with the K / V rows in LDS (very first phase: nothing)
            park_kv();
            __syncthreads();                                    // (very first phase: also publishes the table, the bias and u)
            if (ph + 1 < nph) fetch_kv((ph + 1) * RES_PH);
            else if (qb == 0) fetch_kv(0);                      // the second pass starts over at key 0
            if (ph == 0) {                                      // q + u, formed once per block (u sits in LDS)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const float4 ua = *reinterpret_cast<const float4*>(u_s + 16 * ks + 8 * hh), ub = *reinterpret_cast<const float4*>(u_s + 16 * ks + 8 * hh + 4);
                    const float uv[8] = {ua.x, ua.y, ua.z, ua.w, ub.x, ub.y, ub.z, ub.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) qf[ks][e] = (__bf16)((float)qp[ks][e] + uv[e]);
                }
            }
            const int pj0 = ph * RES_PH;
            if (!live || pj0 >= jend || pj0 + RES_PH <= jbeg || (p.debug & 512)) continue;    // nothing of this phase is visible to the block
#pragma unroll 1
            for (int kt = 0; kt < RES_PH / 64; ++kt) {
                const int j0 = pj0 + 64 * kt;
                if (j0 >= jend || j0 + 64 <= jbeg) continue;
                position(j0);
                sub_step(j0, 64 * kt, 0);
                if (j0 + 32 < jend) sub_step(j0 + 32, 64 * kt + 32, 1);
            }
        }

        // ---- the block's output rows: through the wave's image (32 rows x 128 bytes, pitch 200) so that they leave as whole 128-byte rows
        lq += __shfl_xor(lq, 32, 64);
        if (live) {
            const float inv = 1.f / lq;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int d = 32 * dt + 8 * g4 + 4 * hh;
                    uint2 w;
                    w.x = cvt_pk2(o[dt][4 * g4] * inv, o[dt][4 * g4 + 1] * inv);
                    w.y = cvt_pk2(o[dt][4 * g4 + 2] * inv, o[dt][4 * g4 + 3] * inv);
                    *reinterpret_cast<uint2*>(gs + d) = w;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (one wave: its DS operations execute in order)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = 8 * k + (lane >> 3), ch = lane & 7;
                const uint2 a = *reinterpret_cast<const uint2*>(gs_w + row * GP + ch * 8), c = *reinterpret_cast<const uint2*>(gs_w + row * GP + ch * 8 + 4);
                if (i0 + row < L) *reinterpret_cast<uint4*>(p.o + ((long)b * L + i0 + row) * p.ld_o + h * DH + ch * 8) = make_uint4(a.x, a.y, c.x, c.y);
            }
            if (hh == 0 && i < L) p.lse[(long)z * L + i] = mq * 0.6931471805599453f + __logf(lq);
        }
    }
}

// ------------------------------------------------------------------ backward (dK, dV, dS), position term recomputed in the kernel
// Same structure as flash_bwd_kernel below (key on the lane, dS leaves twice in bf16 for the dq / dE products); the bias tile is not read
// from a slab but recomputed like in flash_fwd_rel_kernel: for a wave's 32 keys and a 32-query tile the p' = L-1-i+j values form one
// window of 63 table rows; G = Qsel . Eext_window^T is two 32x32 MFMA blocks (rows = queries, columns = p'; q_i below L, q_{i+1} above,
// both with the other side's table rows zeroed in the block that holds p' = L), skewed through a private LDS image [p'][query].
// Loads the compiler does not know about (flash_bwd_rel_kernel's tile prefetch): between their issue and their first use a step issues
// its 32 dS / dG store instructions (16 elements x 2 slabs per wave), and vmcnt is an in-order counter - the compiler's own wait for such a load is vmcnt(0), i.e. it drains all
// 64 stores once per step.  Issued by asm, the loads are invisible to its scoreboard; the kernel waits for them itself with
// s_waitcnt vmcnt(32) - everything but the 32 youngest operations, and EVERY path of a step issues at least its 32 slab stores after the
// prefetch (checked in the ISA: Sx32 per step) - and hands the registers over through an empty asm with "+v" operands.  (A first version
// waited for vmcnt(63), taking a step's stores for 64: that guaranteed nothing - the loads were merely always back by then, until B = 8 x
// L = 2000 under the full model's memory traffic let one land after the loop, in registers that by then held an address: a rare illegal access.)
// (outputs are early-clobber: the destination registers must not double as the address operand, and nothing may be allocated over them
// between the issue and the hand-over after the wait - the "+v" hand-over keeps them live, the compiler never sees a use before it)
__device__ __forceinline__ u32x4_t ld16_async(const void* p) {
    u32x4_t v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ float ld4f_async(const void* p) {
    float v;
    asm volatile("global_load_dword %0, %1, off" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ int ld4i_async(const void* p) {
    int v;
    asm volatile("global_load_dword %0, %1, off" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
template <int DH, int MK>
__global__ __launch_bounds__(256, 2) void flash_bwd_rel_kernel(const FlashParams p) {
    using T = Tile<DH>;
    constexpr int KS = DH / 16, DT = DH / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* qtile = smem;                          // (q+u) rows, ring of 64 like ptile: formed from the plain q rows when they are parked
    char* dotile = smem + 64 * T::ROWB;          // dO rows of the current query tile
    char* ptile = smem + 96 * T::ROWB;           // plain q rows, ring of 64 (slot = row & 63): rows i0 .. i0 + 32 are read (the upper part multiplies
                                                 // q_{i+1}), rows i0 + 32 .. i0 + 63 arrive one step ahead
    char* etile = smem + (96 + 64) * T::ROWB;    // ring of 256 table rows (the workgroup's 128 keys x 32 queries need 159)
    float* ctile = reinterpret_cast<float*>(smem + (96 + 64 + 256) * T::ROWB);
    float* lse_s = ctile + 256;
    float* del_s = lse_s + 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
    const int z = blockIdx.y, b = z / p.H, h = z % p.H;
    const int L = p.L;
    const int j = blockIdx.x * 128 + wave * 32 + (lane & 31);
    const int jc = min(j, L - 1);
    const bool kvalid = j < L;
    const bf16_t* krow = p.k + ((long)b * L + jc) * p.ld_kv + h * DH;
    const bf16_t* vrow = p.v + ((long)b * L + jc) * p.ld_kv + h * DH;
    bf16x8 kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        kf[ks] = *reinterpret_cast<const bf16x8*>(krow + 16 * ks + 8 * hh);
        vf[ks] = *reinterpret_cast<const bf16x8*>(vrow + 16 * ks + 8 * hh);
    }
    f32x16 dk[DT], dv[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
    // r_w_bias of this head sits in LDS (q + u is formed while plain q rows are parked; eight registers per thread were eight too many)
    float* u_s = reinterpret_cast<float*>(smem + (96 + 64 + 256) * T::ROWB + 256 * 4 + 128 * 4 + 4 * 64 * 36 * 2);     // behind ctile, lse / delta / lo / hi and the four images
    if (tid < DH) u_s[tid] = p.u[h * DH + tid];
    __syncthreads();
    auto add_u = [&](u32x4_t v) -> u32x4_t {
        const float4 ua = *reinterpret_cast<const float4*>(u_s + (tid % T::NCH) * 8), ub = *reinterpret_cast<const float4*>(u_s + (tid % T::NCH) * 8 + 4);
        u32x4_t o;
        o[0] = pack_bf16x2(__uint_as_float(v[0] << 16) + ua.x, __uint_as_float(v[0] & 0xffff0000u) + ua.y);
        o[1] = pack_bf16x2(__uint_as_float(v[1] << 16) + ua.z, __uint_as_float(v[1] & 0xffff0000u) + ua.w);
        o[2] = pack_bf16x2(__uint_as_float(v[2] << 16) + ub.x, __uint_as_float(v[2] & 0xffff0000u) + ub.y);
        o[3] = pack_bf16x2(__uint_as_float(v[3] << 16) + ub.z, __uint_as_float(v[3] & 0xffff0000u) + ub.w);
        return o;
    };
    const bf16_t* dobase = p.dO + (long)b * L * p.ld_o + h * DH;
    bf16_t* ds16 = p.dS16 + (long)z * p.slab16;
    bf16_t* dg16 = p.dG16 + (long)z * p.slab16;
    // the two bf16 slabs of this (b, h) as raw buffers: a store is [descriptor (SGPRs) + per-lane byte offset (one VGPR, fixed for the kernel)
    // + per-row byte offset (an SGPR)] - no vector address arithmetic per element (64-bit pointer adds were a third of the kernel's VALU
    // instructions), and a lane whose key does not exist carries an offset past the slab: the hardware drops its store, no branch
    const __amdgpu_buffer_rsrc_t rs_ds = __builtin_amdgcn_make_buffer_rsrc(ds16, 0, (int)(p.slab16 * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dg = __builtin_amdgcn_make_buffer_rsrc(dg16, 0, (int)(p.slab16 * 2), 0x00020000);
    constexpr unsigned OOB = 0xFFFFFF00u;

    const int ldp = (int)p.ldp;
    const float c2 = p.scale * 1.4426950408889634f;                         // scale * log2(e)
    int* lo_s = reinterpret_cast<int*>(del_s + 32);                          // MK == 4: the tile's 32 (lo, hi) pairs
    int* hi_s = lo_s + 32;
    constexpr int GPB = 36;                                                  // image row pitch in bf16 (32 queries + 4): 72 bytes
    bf16_t* gs = reinterpret_cast<bf16_t*>(hi_s + 32) + wave * 64 * GPB;     // the wave's private image [64 window columns][GPB]
    const int jw0 = blockIdx.x * 128;
    const bf16_t* pbase = p.qp + (long)b * L * p.ld_qp + h * DH;
    const bf16_t* ebase = p.e16 + h * DH;
    const float* cbase = p.cT + (long)h * L;
    u32x4_t epre, ppre;                                  // the 32 new table rows of the next query tile + the plain q rows i0 + 64 .. i0 + 95 (one chunk each)
    float cpre = 0.f;
    auto ext_row = [&](int pe, int& src) -> bool {
        src = pe < L ? pe : pe - L - 1;
        const bool ok = pe >= 0 && pe != L && src < L;
        src = min(max(src, 0), L - 1);
        return ok;
    };
    auto load_chunk = [&](int pe, int ch) -> u32x4_t {
        int src;
        const bool ok = ext_row(pe, src);
        u32x4_t v = *reinterpret_cast<const u32x4_t*>(ebase + (long)src * p.ld_e + ch * 8);
        if (!ok) v = u32x4_t{0u, 0u, 0u, 0u};
        return v;
    };
    auto load_c = [&](int pe) -> float {
        int src;
        const bool ok = ext_row(pe, src);
        const float cv = cbase[src];
        return ok ? cv : 0.f;
    };
    // table window of a query tile: rows p' = wbase .. wbase + 159, wbase = L - 32 - i0 + jw0, in a ring (slot = (p' - L) & 255); every step
    // moves the window down by 32 rows, so only those 32 are fetched per step; the prologue stages the 160 rows above the first window
    auto stage_window = [&](int i0) {                    // (i0 = first tile - 32) also: the first tile's own plain q rows
        for (int c = tid; c < 32 * T::NCH; c += 256) {           // (c % NCH == tid % NCH: 256 is a multiple of NCH, so uc is this chunk's bias)
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(pbase + (long)min(i0 + 32 + c / T::NCH, L - 1) * p.ld_qp + (c % T::NCH) * 8);
            *reinterpret_cast<u32x4_t*>(ptile + T::off((i0 + 32 + c / T::NCH) & 63, c % T::NCH)) = v;
            *reinterpret_cast<u32x4_t*>(qtile + T::off((i0 + 32 + c / T::NCH) & 63, c % T::NCH)) = add_u(v);
        }
        const int wbase = L - 32 - i0 + jw0;
        for (int c = tid; c < 160 * T::NCH; c += 256) {
            const int pe = wbase + c / T::NCH, ch = c % T::NCH;
            *reinterpret_cast<u32x4_t*>(etile + T::off((pe - L) & 255, ch)) = load_chunk(pe, ch);
        }
        if (tid < 160) ctile[(wbase + tid - L) & 255] = load_c(wbase + tid);
    };
    // (every load and every park below is executed by ALL threads - 32 rows x NCH chunks = one chunk per thread for Dh = 64; the few
    // scalars are fetched and parked redundantly by the 8 threads that share tid & 31.  Loads under `if (tid < ..)` put a control-flow
    // join behind them, and at a join the compiler waits for vmcnt(0): that drained the tile's dS / dG stores once per step.)
    constexpr bool ALLCH = 32 * T::NCH >= 256;
    auto fetch_bias = [&](int i0) {
        const int wbase = L - 32 - i0 + jw0;
        if (ALLCH || tid < 32 * T::NCH) {
            int src;
            ext_row(wbase + tid / T::NCH, src);
            epre = ld16_async(ebase + (long)src * p.ld_e + (tid % T::NCH) * 8);
            ppre = ld16_async(pbase + (long)min(i0 + 32 + tid / T::NCH, L - 1) * p.ld_qp + (tid % T::NCH) * 8);
        }
        int srcc;
        ext_row(wbase + (tid & 31), srcc);
        cpre = ld4f_async(cbase + srcc);
    };
    int parked_i0 = 0;
    auto park_bias = [&]() {
        const int wbase = L - 32 - parked_i0 + jw0;
        if (ALLCH || tid < 32 * T::NCH) {
            int src;                                              // (rows of the extended table that do not exist read as zero: decided here, after the wait)
            if (!ext_row(wbase + tid / T::NCH, src)) epre = u32x4_t{0u, 0u, 0u, 0u};
            *reinterpret_cast<u32x4_t*>(etile + T::off((wbase + tid / T::NCH - L) & 255, tid % T::NCH)) = epre;
            *reinterpret_cast<u32x4_t*>(ptile + T::off((parked_i0 + 32 + tid / T::NCH) & 63, tid % T::NCH)) = ppre;
            *reinterpret_cast<u32x4_t*>(qtile + T::off((parked_i0 + 32 + tid / T::NCH) & 63, tid % T::NCH)) = add_u(ppre);
        }
        int srcc;
        ctile[(wbase + (tid & 31) - L) & 255] = ext_row(wbase + (tid & 31), srcc) ? cpre : 0.f;
    };
    // result: the bias in the layout of the score accumulator, which it initialises (S = bias + (q+u).k comes out of the MFMA chain itself: no
    // separate registers for the bias, no add per element)
    auto read_bias = [&](int i0, f32x16& bv) {
        const int pe_w = L - 32 - i0 + jw0 + 32 * wave;           // p' of the wave's window column 0
        asm volatile("" ::: "memory");
        // lane = window column of the block; cext[column] is the accumulators' INITIAL value (the chain adds q . E onto it: no add per element)
        auto cinit = [&](int blk) -> float { return ctile[(pe_w + 32 * blk + (lane & 31) - L) & 255]; };
        auto emit = [&](int blk, const f32x16& g) {               // queries 8 g4 + 4 hh + (0..3) of this column, to bf16
            bf16_t* col = gs + (32 * blk + (lane & 31)) * GPB;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                uint2 w;
                w.x = pack_bf16x2(g[4 * g4], g[4 * g4 + 1]);
                w.y = pack_bf16x2(g[4 * g4 + 2], g[4 * g4 + 3]);
                *reinterpret_cast<uint2*>(col + 8 * g4 + 4 * hh) = w;
            }
        };
        const bool all_low = pe_w + 63 <= L - 1, all_up = pe_w >= L + 1;
        if (all_low || all_up) {                                  // both blocks on one side of p' = L: two independent MFMA chains, interleaved
            f32x16 g0, g1;
            const float c0v = cinit(0), c1v = cinit(1);
#pragma unroll
            for (int r = 0; r < 16; ++r) { g0[r] = c0v; g1[r] = c1v; }
            const int qrow = (i0 + (lane & 31) + (all_low ? 0 : 1)) & 63;
            const int e0 = (pe_w + (lane & 31) - L) & 255, e1 = (pe_w + 32 + (lane & 31) - L) & 255;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 qa = *reinterpret_cast<const bf16x8*>(ptile + T::off(qrow, 2 * ks + hh));
                const bf16x8 f0 = *reinterpret_cast<const bf16x8*>(etile + T::off(e0, 2 * ks + hh));
                const bf16x8 f1 = *reinterpret_cast<const bf16x8*>(etile + T::off(e1, 2 * ks + hh));
                g0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, f0, g0, 0, 0, 0);
                g1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, f1, g1, 0, 0, 0);
            }
            emit(0, g0);
            emit(1, g1);
        } else {
#pragma unroll 1
        for (int blk = 0; blk < 2; ++blk) {
            const int pe0 = pe_w + 32 * blk;
            f32x16 g;
            const float cv = cinit(blk);
#pragma unroll
            for (int r = 0; r < 16; ++r) g[r] = cv;
            const int erow = (pe0 + (lane & 31) - L) & 255;          // ring slot of this lane's table row
            if (pe0 + 31 <= L - 1) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 qa = *reinterpret_cast<const bf16x8*>(ptile + T::off((i0 + (lane & 31)) & 63, 2 * ks + hh));
                    const bf16x8 ef = *reinterpret_cast<const bf16x8*>(etile + T::off(erow, 2 * ks + hh));
                    g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, ef, g, 0, 0, 0);
                }
            } else if (pe0 >= L + 1) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 qa = *reinterpret_cast<const bf16x8*>(ptile + T::off((i0 + (lane & 31) + 1) & 63, 2 * ks + hh));
                    const bf16x8 ef = *reinterpret_cast<const bf16x8*>(etile + T::off(erow, 2 * ks + hh));
                    g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, ef, g, 0, 0, 0);
                }
            } else {
                const bool lower = pe0 + (lane & 31) <= L - 1;
                bf16x8 zero;
#pragma unroll
                for (int e = 0; e < 8; ++e) zero[e] = (__bf16)0.0f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 qa = *reinterpret_cast<const bf16x8*>(ptile + T::off((i0 + (lane & 31)) & 63, 2 * ks + hh));
                    const bf16x8 qb = *reinterpret_cast<const bf16x8*>(ptile + T::off((i0 + (lane & 31) + 1) & 63, 2 * ks + hh));
                    const bf16x8 ef = *reinterpret_cast<const bf16x8*>(etile + T::off(erow, 2 * ks + hh));
                    g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, lower ? ef : zero, g, 0, 0, 0);
                    g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qb, lower ? zero : ef, g, 0, 0, 0);
                }
            }
            emit(blk, g);
        }
        }
        asm volatile("" ::: "memory");                            // DS operations of one wave execute in order
        // score (query qi, own key jj = lane & 31) sits at window column 31 - qi + jj
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qi = (r & 3) + 8 * (r >> 2) + 4 * hh;
            const float v = bf16_to_f32(gs[(31 - qi + (lane & 31)) * GPB + qi]);
            bv[r] = kvalid ? v : NEGBIG;                          // a key that does not exist: exp2 underflows to 0, so P = dS = 0 with no further test
        }
    };
    // narrow structured masks (launcher's choice, p.bwd_skip): both bf16 slabs were zeroed up front and only the query tiles that can see
    // one of this workgroup's 128 keys are walked: query i meets key j iff i - left <= j <= i + right
    int ibeg = 0, iend = L;
    if (p.bwd_skip) {
        ibeg = max(0, jw0 - p.mask_right) & ~63;
        iend = (int)min((long)L, (long)jw0 + 127 + p.mask_left + 1);
        if (iend <= ibeg) { ibeg = 0; iend = 0; }
    }
    u32x4_t opre = {0u, 0u, 0u, 0u};                                    // dO rows of the next tile, one 16-byte chunk per thread
    auto fetch_do = [&](int i0) {
        if (ALLCH || tid < 32 * T::NCH) opre = ld16_async(dobase + (long)min(i0 + tid / T::NCH, L - 1) * p.ld_o + (tid % T::NCH) * 8);
    };
    fetch_do(ibeg);
    if (ibeg < iend) stage_window(ibeg - 32);        // rows wbase(ibeg) + 32 .. + 191: step(ibeg) parks the 32 below them
    fetch_bias(ibeg);
    float lse_pre = 0.f, del_pre = 0.f;
    int lo_pre = 0, hi_pre = 0;
    auto fetch_rows = [&](int i0) {                                     // log-sum-exp, delta (and the mask interval) of the 32 queries of tile i0
        const int ii = min(i0 + (tid & 31), L - 1);
        lse_pre = ld4f_async(p.lse + (long)z * L + ii);
        del_pre = ld4f_async(p.delta + (long)z * L + ii);
        if constexpr (MK == 4) {
            const int* r = reinterpret_cast<const int*>(p.mask) + (long)b * p.mask_sb + 2 * ii;
            lo_pre = ld4i_async(r);
            hi_pre = ld4i_async(r + 1);
        }
    };
    fetch_rows(ibeg);
    auto step = [&](int i0, bool first) {
        __syncthreads();
        // the prefetched registers: issued one step ago, in front of that step's 32 slab store instructions (the very first tile: in front of nothing)
        // (16 accumulator elements x one store into each of the two slabs: the count the wait below leaves in flight - keep the two in step)
        constexpr int SLAB_STORES_PER_STEP = 2 * 16;
        static_assert(SLAB_STORES_PER_STEP == 32 && SLAB_STORES_PER_STEP <= 63, "the prefetch wait below counts a step's slab store instructions");
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else TTMI_VM_WAIT("bwdrel", SLAB_STORES_PER_STEP);
        asm volatile("" : "+v"(opre), "+v"(epre), "+v"(ppre), "+v"(cpre), "+v"(lse_pre), "+v"(del_pre), "+v"(lo_pre), "+v"(hi_pre));
        if (ALLCH || tid < 32 * T::NCH) *reinterpret_cast<u32x4_t*>(dotile + T::off(tid / T::NCH, tid % T::NCH)) = opre;
        parked_i0 = i0;
        const char* qcur = qtile + (i0 & 32) * T::ROWB;     // this tile's 32 (q+u) rows inside the ring (same swizzle phase: 32 rows = 4 periods)
        park_bias();
        lse_s[tid & 31] = lse_pre * 1.4426950408889634f;            // the tile's row statistics arrived with its operands; parked pre-multiplied:
        del_s[tid & 31] = del_pre * p.scale;                        // p = exp2(s c2 - lse log2 e), dS = p (dP scale - delta scale)
        if constexpr (MK == 4) {
            lo_s[tid & 31] = lo_pre;
            hi_s[tid & 31] = hi_pre;
        }
        __syncthreads();
        if (i0 + 32 < iend) {                                       // next tile's operands and bias fly under this tile's MFMAs
            fetch_do(i0 + 32);
            fetch_bias(i0 + 32);
            fetch_rows(i0 + 32);
            TTMI_VM_GUARD("bwdrel");                                 // the prefetch is older than this point; the step's slab stores follow it
        }
        f32x16 s, dp;
        if (p.debug & 8) {                                          // (timing experiments: no position term)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
        } else {
            read_bias(i0, s);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) dp[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 qa = *reinterpret_cast<const bf16x8*>(qcur + T::off(lane & 31, 2 * ks + hh));
            const bf16x8 da = *reinterpret_cast<const bf16x8*>(dotile + T::off(lane & 31, 2 * ks + hh));
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[ks], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vf[ks], dp, 0, 0, 0);
        }
        // dS16[i][j] at i*ldp + j;  dG16[r][c-1], (r, c) = divmod((i+1) L + j, L+1): j <= i -> (i, L-i+j), j > i -> (i+1, j-i-1), i.e.
        // element i*(ldp-1) + (j <= i ? L-1+j : ldp+j-2), nothing for j == i+1 (c = 0)
        const int ds0 = (i0 + 4 * hh) * ldp + j, dg0 = (i0 + 4 * hh) * (ldp - 1);
        const unsigned v_ds = kvalid ? (unsigned)((4 * hh * ldp + j) * 2) : OOB;                       // lane part of a dS16 byte offset
        const unsigned v_lo = kvalid ? (unsigned)((4 * hh * (ldp - 1) + L - 1 + j) * 2) : OOB;         // dG16, j <= i
        const unsigned v_hi = kvalid ? (unsigned)((4 * hh * (ldp - 1) + ldp + j - 2) * 2) : OOB;       // dG16, j >= i + 2
        const int s_ds = i0 * ldp * 2, s_dg = i0 * (ldp - 1) * 2;                                       // row part (wave-uniform)
        // pad columns [L, ldp) of both bf16 slabs feed the K loop of the dq / dE products and must be zero: the lanes whose key index falls
        // there write the zeros (no separate strided memsets over B*H*L rows)
        if (!kvalid && j < ldp) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cq = (r & 3) + 8 * (r >> 2);
                if (i0 + cq + 4 * hh < L) {
                    ds16[(unsigned)(ds0 + cq * ldp)] = 0;
                    dg16[(unsigned)(ds0 + cq * ldp)] = 0;
                }
            }
        }
        // interior tiles (all 32 queries and all 128 keys of the workgroup in range, the j == i+1 diagonal not crossing the tile) take a
        // branch-free element loop; edge and diagonal tiles the general one
        const bool interior = (i0 + 32 <= L) && (jw0 > i0 + 32 || jw0 + 127 < i0 + 1);
        if (interior) {
            // all 32 queries in range, the whole tile on one side of the j == i+1 diagonal: branch-free score loop, then ONE predicated
            // region for the lanes whose key exists
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = (r & 3) + 8 * (r >> 2) + 4 * hh;
                float pr = 0.f, ds = 0.f;
                if (!(MK == 4 ? (j < lo_s[q] || j > hi_s[q]) : is_masked<MK>(p, b, i0 + q, j))) {
                    pr = __builtin_amdgcn_exp2f(fmaf(s[r], c2, -lse_s[q]));
                    ds = pr * fmaf(dp[r], p.scale, -del_s[q]);
                }
                s[r] = pr;
                dp[r] = ds;
            }
            {
                const unsigned v_g = (p.debug & 2) ? OOB : (jw0 < i0 ? v_lo : v_hi);
                const unsigned v_d = (p.debug & 2) ? OOB : v_ds;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cq = (r & 3) + 8 * (r >> 2);
                    const bf16_t d16 = f32_to_bf16(dp[r]);
                    __builtin_amdgcn_raw_buffer_store_b16(d16, rs_ds, v_d, s_ds + cq * ldp * 2, 0);
                    __builtin_amdgcn_raw_buffer_store_b16(d16, rs_dg, v_g, s_dg + cq * (ldp - 1) * 2, 0);
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cq = (r & 3) + 8 * (r >> 2);
                const int q = cq + 4 * hh;
                const int i = i0 + q;
                const bool inb = (i < L) && kvalid;
                float pr = 0.f, ds = 0.f;
                if (inb) {
                    if (!(MK == 4 ? (j < lo_s[q] || j > hi_s[q]) : is_masked<MK>(p, b, i, j))) {
                        pr = __builtin_amdgcn_exp2f(fmaf(s[r], c2, -lse_s[q]));
                        ds = pr * fmaf(dp[r], p.scale, -del_s[q]);
                    }
                }
                {
                    const bf16_t d16 = f32_to_bf16(ds);
                    __builtin_amdgcn_raw_buffer_store_b16(d16, rs_ds, inb ? v_ds : OOB, s_ds + cq * ldp * 2, 0);
                    __builtin_amdgcn_raw_buffer_store_b16(d16, rs_dg, (inb && j != i + 1) ? (j <= i ? v_lo : v_hi) : OOB, s_dg + cq * (ldp - 1) * 2, 0);
                }
                s[r] = pr;
                dp[r] = ds;
            }
        }
        if (p.debug & 16) {                                         // (timing experiments: no dV / dK products)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dv[0][r] += s[r]; dk[0][r] += dp[r]; }
            return;
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 pb = pack8(s, 8 * s2);
            const bf16x8 dsb = pack8(dp, 8 * s2);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const bf16x8 a_do = tr_frag<DH>(dotile, 16 * s2, 32 * dt, lane);
                const bf16x8 a_qu = tr_frag<DH>(qcur, 16 * s2, 32 * dt, lane);
                dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_do, pb, dv[dt], 0, 0, 0);
                dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_qu, dsb, dk[dt], 0, 0, 0);
            }
        }
    };
    for (int i0 = ibeg; i0 < iend; i0 += 64) {
        step(i0, i0 == ibeg);
        if (i0 + 32 < iend) step(i0 + 32, false);
    }
    if (kvalid) {
        float* dkrow = p.dK + ((long)b * L + j) * p.ld_dkv + h * DH;
        float* dvrow = p.dV + ((long)b * L + j) * p.ld_dkv + h * DH;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = 32 * dt + 8 * g4 + 4 * hh;
                if (p.dK16) {      // bf16, the form the qkv dgrad / wgrad GEMMs read (no f32 copy, no conversion pass over dqkv)
                    const long o16 = ((long)b * L + j) * p.ld_dkv + h * DH + d;
                    uint2 wk, wv;
                    wk.x = pack_bf16x2(dk[dt][4 * g4], dk[dt][4 * g4 + 1]); wk.y = pack_bf16x2(dk[dt][4 * g4 + 2], dk[dt][4 * g4 + 3]);
                    wv.x = pack_bf16x2(dv[dt][4 * g4], dv[dt][4 * g4 + 1]); wv.y = pack_bf16x2(dv[dt][4 * g4 + 2], dv[dt][4 * g4 + 3]);
                    *reinterpret_cast<uint2*>(p.dK16 + o16) = wk;
                    *reinterpret_cast<uint2*>(p.dV16 + o16) = wv;
                } else {
                    *reinterpret_cast<float4*>(dkrow + d) = make_float4(dk[dt][4 * g4], dk[dt][4 * g4 + 1], dk[dt][4 * g4 + 2], dk[dt][4 * g4 + 3]);
                    *reinterpret_cast<float4*>(dvrow + d) = make_float4(dv[dt][4 * g4], dv[dt][4 * g4 + 1], dv[dt][4 * g4 + 2], dv[dt][4 * g4 + 3]);
                }
            }
    }
}

// ------------------------------------------------------------------ backward (dK, dV, dS), position term in the kernel: second generation
// Same arithmetic, slabs and launch geometry as flash_bwd_rel_kernel (one 256-thread workgroup per 128 keys of a head, key on the lane,
// 32-query steps, Dh = 64); what changed is everything around the arithmetic.  Cycle stamps inside the round-3 kernel (tools/debug/
// bwd_stamps.py: one wave, C2 audio layer) read 169k cycles per workgroup: 23k of prologue (the 160-row table window staged through
// registers in a dependent loop), then per step ~6.9k on an interior tile - 2.0k for the position tile (two MFMA chains, a bf16 image
// written to LDS and read back by sixteen 2-byte reads, in front of the score MFMAs that wait for it as their accumulator), 1.75k for the
// element loop and its 32 slab stores, 0.85k waiting for and parking the prefetch, two barriers - and 10.9k on the 6 of 16 tiles that touch
// the diagonal or the sequence end (one branch per element).  Here:
//   * the table window, the dO tile and the rows of the next table block arrive by LDS-DMA (global_load_lds_dwordx4: no registers, no park;
//     rows of the extended table that do not exist are fetched from a zero page), the prologue's 6 KiB per wave are in flight at once;
//   * rings sized so that ONE barrier per step suffices (plain q / q + u rows: 128 slots, dO: two tiles, row statistics: two sets);
//   * the position tile is skewed in registers: both 32-column blocks of G are packed into one bf16 pair per element (the forward pass
//     rounds its position scores to bf16 too) and one ds_bpermute_b32 per element moves the pair from window column 31 - q + j to key j;
//     the score and dP chains start from per-lane constants and run beside the G chains, bias and content meet in one add per element;
//   * diagonal and end tiles run a branch-free element loop with per-element selects (stores dropped by out-of-range offsets).
__device__ __attribute__((aligned(128))) const unsigned g_attn_zero_page[32] = {};
// one LDS-DMA instruction: 16 bytes per lane from `g` to (wave-uniform LDS address) + lane * 16.  Issued by asm so that the compiler does
// not order its own LDS accesses behind it (it waits for vmcnt(0) in front of every LDS access it cannot prove disjoint from a DMA it
// knows about); completion is this kernel's own business (counted vmcnt wait + barrier).
__device__ __forceinline__ void lds_dma16(const void* g, unsigned lds_wave_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_wave_addr) : "memory", "m0");
}
// the same with the address as (wave-uniform base in SGPRs) + (32-bit unsigned byte offset per lane): no 64-bit vector arithmetic.
// HAZARD: a vector-memory instruction must not read an SGPR within 5 wait states of a VALU write to it (v_readlane restoring a spilled
// SGPR pair is such a write, and the compiler places it right in front of the asm); the hazard recognizer does not look inside inline
// asm, so every SGPR-based access here opens with `s_nop 4`.  (Found as a memory fault in the mask-kind-3 / -4 instances only - the ones
// whose SGPR pressure spills the bases to VGPR lanes.)
__device__ __forceinline__ void lds_dma16_s(const void* base, unsigned off, unsigned lds_wave_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(lds_wave_addr) : "memory", "m0");
}
__device__ __forceinline__ u32x4_t ld16_async_s(const void* base, unsigned off) {
    u32x4_t v;
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=&v"(v) : "v"(off), "s"(base) : "memory");
    return v;
}
__device__ __forceinline__ float ld4f_async_s(const void* base, unsigned off) {
    float v;
    asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=&v"(v) : "v"(off), "s"(base) : "memory");
    return v;
}
__device__ __forceinline__ int ld4i_async_s(const void* base, unsigned off) {
    int v;
    asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=&v"(v) : "v"(off), "s"(base) : "memory");
    return v;
}
#ifdef TTMI_STAMPS
__device__ unsigned long long g_bwd_stamps[1024];
#define STAMP(k) do { if (stamp_on) g_bwd_stamps[stamp_n * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define STAMP_NEXT() do { ++stamp_n; } while (0)
#else
#define STAMP(k) do {} while (0)
#define STAMP_NEXT() do {} while (0)
#endif
#ifndef BWD2_MINB
#define BWD2_MINB 2
#endif
constexpr int BWD2_LDS = (256 + 128 + 128 + 64) * 128 + 256 * 4 + 4 * 64 * 4 + 64 * 4;
template <int MK>
__global__ __launch_bounds__(256, BWD2_MINB) void flash_bwd_rel2_kernel(const FlashParams p) {
    constexpr int DH = 64, KS = 4, DT = 2;
    using T = Tile<DH>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* etile = smem;                                          // ring of 256 extended-table rows (slot = (p' - L) & 255)
    char* ptile = smem + 256 * 128;                              // plain q rows, ring of 128 (slot = row & 127)
    char* qtile = ptile + 128 * 128;                             // q + u rows, same ring
    char* dotile = qtile + 128 * 128;                            // two dO tiles of 32 rows (step parity)
    float* ctile = reinterpret_cast<float*>(dotile + 64 * 128);  // extended bias, ring of 256
    float* lse_s = ctile + 256;                                  // [2][32] row statistics of the tile (step parity)
    float* del_s = lse_s + 64;
    int* lo_s = reinterpret_cast<int*>(del_s + 64);              // MK == 4: the tile's key intervals
    int* hi_s = lo_s + 64;
    float* u_s = reinterpret_cast<float*>(hi_s + 64);            // r_w_bias of this head
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), hh = lane >> 5, jj = lane & 31;
    const int z = blockIdx.y, b = z / p.H, h = z % p.H;
    const int L = p.L;
#ifdef TTMI_STAMPS
    const bool stamp_on = blockIdx.x == 1 && blockIdx.y == 5 && tid == 0;
    int stamp_n = 0;
    STAMP(0);
    STAMP_NEXT();
#endif
    const int jw0 = blockIdx.x * 128;
    const int j = jw0 + wave * 32 + jj;
    const int jc = min(j, L - 1);
    const bool kvalid = j < L;
    const bf16_t* krow = p.k + ((long)b * L + jc) * p.ld_kv + h * DH;
    const bf16_t* vrow = p.v + ((long)b * L + jc) * p.ld_kv + h * DH;
    bf16x8 kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        kf[ks] = *reinterpret_cast<const bf16x8*>(krow + 16 * ks + 8 * hh);
        vf[ks] = *reinterpret_cast<const bf16x8*>(vrow + 16 * ks + 8 * hh);
    }
    f32x16 dk[DT], dv[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
    const bf16_t* dobase = p.dO + (long)b * L * p.ld_o + h * DH;
    const bf16_t* pbase = p.qp + (long)b * L * p.ld_qp + h * DH;
    const bf16_t* ebase = p.e16 + h * DH;
    const float* cbase = p.cT + (long)h * L;
    bf16_t* ds16 = p.dS16 + (long)z * p.slab16;
    bf16_t* dg16 = p.dG16 + (long)z * p.slab16;
    const __amdgpu_buffer_rsrc_t rs_ds = __builtin_amdgcn_make_buffer_rsrc(ds16, 0, (int)(p.slab16 * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dg = __builtin_amdgcn_make_buffer_rsrc(dg16, 0, (int)(p.slab16 * 2), 0x00020000);
    constexpr unsigned OOB = 0xFFFFFF00u;
    const int ldp = (int)p.ldp;
    const float c2 = p.scale * 1.4426950408889634f;                         // scale * log2(e)
    const unsigned lds0 = (unsigned)(size_t)smem;                            // LDS byte address of the arena (DMA destinations)
    // this lane inside a wave's 1 KiB DMA piece = 8 rows of a 32-row block: row 8 wave + lane / 8, and the source chunk that belongs at the
    // lane's position of the swizzled image (every 32-row block starts at a slot that is a multiple of 32: same swizzle phase)
    const int rib = 8 * wave + (lane >> 3);
    const int csrc8 = ((lane & 7) ^ ((rib >> 1) & 7)) * 8;
    // Extended table row p' -> table row: p' (p' < L), p' - L - 1 (p' > L); p' = L is the zero row.  Rows outside [0, 2 L] are never met by
    // a (query, key) pair that exists (p' = L - 1 - i + j), so they are fetched clamped and hold whatever row that is; the ONE row that must
    // read as zeros, p' = L, is fetched clamped as well and overwritten in LDS by the lanes that fetched it (patch_zero_row, after the wait
    // for the fetch and before the barrier that publishes the block).  All fetches are (uniform base) + (32-bit lane offset).
    auto tab_row = [&](int pe) -> int { return min(max(pe < L ? pe : pe - L - 1, 0), L - 1); };
    const unsigned ld_e2 = (unsigned)p.ld_e * 2u, ld_o2 = (unsigned)p.ld_o * 2u, ld_qp2 = (unsigned)p.ld_qp * 2u;
    const float* lse_z = p.lse + (long)z * L;
    const float* del_z = p.delta + (long)z * L;
    auto dma_e_block = [&](int pe0) {                                        // 32 extended-table rows pe0 .. pe0 + 31 (pe0 - L a multiple of 32) into the ring
        lds_dma16_s(ebase, (unsigned)tab_row(pe0 + rib) * ld_e2 + csrc8 * 2, lds0 + (((pe0 - L) & 255) + 8 * wave) * 128);
    };
    auto patch_zero_row = [&](int pe0) {                                     // this lane's 16 bytes of the block it fetched, if they belong to p' = L
        if (pe0 + rib == L) *reinterpret_cast<u32x4_t*>(etile + (((pe0 - L) & 255) + rib) * 128 + (lane & 7) * 16) = u32x4_t{0u, 0u, 0u, 0u};
    };
    auto dma_do = [&](int i0, int par) {
        lds_dma16_s(dobase, (unsigned)min(i0 + rib, L - 1) * ld_o2 + csrc8 * 2, lds0 + (256 + 128 + 128) * 128 + par * 4096 + wave * 1024);
    };
    auto add_u = [&](u32x4_t v) -> u32x4_t {
        const float4 ua = *reinterpret_cast<const float4*>(u_s + (tid & 7) * 8), ub = *reinterpret_cast<const float4*>(u_s + (tid & 7) * 8 + 4);
        u32x4_t o;
        o[0] = cvt_pk2(__uint_as_float(v[0] << 16) + ua.x, __uint_as_float(v[0] & 0xffff0000u) + ua.y);
        o[1] = cvt_pk2(__uint_as_float(v[1] << 16) + ua.z, __uint_as_float(v[1] & 0xffff0000u) + ua.w);
        o[2] = cvt_pk2(__uint_as_float(v[2] << 16) + ub.x, __uint_as_float(v[2] & 0xffff0000u) + ub.y);
        o[3] = cvt_pk2(__uint_as_float(v[3] << 16) + ub.z, __uint_as_float(v[3] & 0xffff0000u) + ub.w);
        return o;
    };
    // what still travels through registers (asm loads, handed over after the counted wait): the plain q rows i0 + 32 .. i0 + 63 (q + u is
    // formed when they are parked), the 32 new bias values, the tile's row statistics - each executed by ALL threads, redundantly where
    // fewer values exist (a load under `if (tid < ..)` puts a control-flow join behind it, where the compiler waits for vmcnt(0))
    u32x4_t ppre = {0u, 0u, 0u, 0u};
    float cpre = 0.f, lse_pre = 0.f, del_pre = 0.f;
    int lo_pre = 0, hi_pre = 0;
    auto prefetch = [&](int i0, int par) {                                   // everything tile i0 needs that is not in LDS yet
        const int wbase = L - 32 - i0 + jw0;
        dma_do(i0, par);
        dma_e_block(wbase);
        ppre = ld16_async_s(pbase, (unsigned)min(i0 + 32 + (tid >> 3), L - 1) * ld_qp2 + (tid & 7) * 16);
        cpre = ld4f_async_s(cbase, (unsigned)tab_row(wbase + (tid & 31)) * 4u);
        const unsigned ii4 = (unsigned)min(i0 + (tid & 31), L - 1) * 4u;
        lse_pre = ld4f_async_s(lse_z, ii4);
        del_pre = ld4f_async_s(del_z, ii4);
        if constexpr (MK == 4) {
            const int* r = reinterpret_cast<const int*>(p.mask) + (long)b * p.mask_sb;
            lo_pre = ld4i_async_s(r, 2u * ii4);
            hi_pre = ld4i_async_s(r, 2u * ii4 + 4u);
        }
    };
    int ibeg = 0, iend = L;
    if (p.bwd_skip) {
        ibeg = max(0, jw0 - p.mask_right) & ~63;
        iend = (int)min((long)L, (long)jw0 + 127 + p.mask_left + 1);
        if (iend <= ibeg) { ibeg = 0; iend = 0; }
    }
    // ---- prologue: the 160 window rows above the first tile's new block, the first tile's own plain q rows (through registers: + u), the bias
    // of those 160 rows, then the first tile's prefetch - all in flight together
    if (ibeg < iend) {
        const int wb = L - ibeg + jw0;                                       // = window base of tile ibeg, + 32
#pragma unroll
        for (int k = 0; k < 5; ++k) dma_e_block(wb + 32 * k);
        const u32x4_t q0 = *reinterpret_cast<const u32x4_t*>(pbase + (long)min(ibeg + (tid >> 3), L - 1) * p.ld_qp + (tid & 7) * 8);
        const bool cok = wb + tid != L;
        const float cv = cbase[tab_row(wb + tid)];
        if (tid < DH) u_s[tid] = p.u[h * DH + tid];
        prefetch(ibeg, 0);
        __syncthreads();                                                     // u_s
        const int slot = (ibeg + (tid >> 3)) & 127;
        *reinterpret_cast<u32x4_t*>(ptile + T::off(slot, tid & 7)) = q0;
        *reinterpret_cast<u32x4_t*>(qtile + T::off(slot, tid & 7)) = add_u(q0);
        if (tid < 160) ctile[(wb + tid - L) & 255] = cok ? cv : 0.f;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the window blocks have landed: the row p' = L, if one of them holds it, becomes zeros
#pragma unroll
        for (int k = 0; k < 5; ++k) patch_zero_row(wb + 32 * k);
    }
    // lane parts of the slab byte offsets (fixed for the kernel; a lane whose key does not exist carries an offset past the slab: the hardware drops its stores)
    const unsigned v_ds = kvalid ? (unsigned)((4 * hh * ldp + j) * 2) : OOB;                       // dS16[i][j]
    const unsigned v_lo = kvalid ? (unsigned)((4 * hh * (ldp - 1) + L - 1 + j) * 2) : OOB;         // dG16, j <= i
    const unsigned v_hi = kvalid ? (unsigned)((4 * hh * (ldp - 1) + ldp + j - 2) * 2) : OOB;       // dG16, j >= i + 2
    // skew of the position tile: element r of this lane (query qi = cq(r) + 4 hh, key jj) sits at window column 31 - qi + jj: block 1 iff jj > qi,
    // lane (31 - qi + jj) & 31 of the same half
    const int a4 = 4 * (31 - 4 * hh + jj), hb = 128 * hh, jq = jj - 4 * hh;
    // fragment reads of the swizzled tiles: chunk 2 ks + hh of row r sits at r * 128 + (((2 ks + hh) ^ ((r >> 1) & 7)) << 4) = r * 128 + (sw ^ (ks << 5)),
    // sw = (hh ^ ((r >> 1) & 7)) << 4: one v_xad_u32 per read.  Every row read below is jj or jj + 1 plus a multiple of 32: two lane constants.
    const unsigned sw0 = (unsigned)((hh ^ ((jj >> 1) & 7)) << 4), sw1 = (unsigned)((hh ^ (((jj + 1) >> 1) & 7)) << 4);
    auto frag = [&](const char* tile, unsigned rowbytes, unsigned sw, int ks) -> bf16x8 {
        return *reinterpret_cast<const bf16x8*>(tile + ((sw ^ (unsigned)(ks << 5)) + rowbytes));
    };
    constexpr int SLAB_STORES_PER_STEP = 2 * 16;
    static_assert(SLAB_STORES_PER_STEP == 32, "the prefetch wait counts a step's slab store instructions");

    auto step = [&](int i0, int par, bool first) {
        STAMP(0);
        // the prefetch of this tile: issued one step ago, in front of that step's 32 slab store instructions (the very first tile: in front of nothing)
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else TTMI_VM_WAIT("bwdrel2", SLAB_STORES_PER_STEP);
        asm volatile("" : "+v"(ppre), "+v"(cpre), "+v"(lse_pre), "+v"(del_pre), "+v"(lo_pre), "+v"(hi_pre));
        {
            const int slot = (i0 + 32 + (tid >> 3)) & 127;
            *reinterpret_cast<u32x4_t*>(ptile + T::off(slot, tid & 7)) = ppre;
            *reinterpret_cast<u32x4_t*>(qtile + T::off(slot, tid & 7)) = add_u(ppre);
            const int wbase = L - 32 - i0 + jw0;
            ctile[(wbase + (tid & 31) - L) & 255] = wbase + (tid & 31) != L ? cpre : 0.f;
            patch_zero_row(wbase);
            lse_s[32 * par + (tid & 31)] = lse_pre * 1.4426950408889634f;    // parked pre-multiplied: p = exp2(s c2 - lse log2 e), dS = p (dP scale - delta scale)
            del_s[32 * par + (tid & 31)] = del_pre * p.scale;
            if constexpr (MK == 4) {
                lo_s[32 * par + (tid & 31)] = lo_pre;
                hi_s[32 * par + (tid & 31)] = hi_pre;
            }
        }
        STAMP(1);
        __syncthreads();                                                     // the only barrier of a step (ring sizes: see the kernel's header)
        STAMP(2);
        if (i0 + 32 < iend) {
            prefetch(i0 + 32, par ^ 1);
            TTMI_VM_GUARD("bwdrel2");                                        // the prefetch is older than this point; the step's slab stores follow it
        }
        STAMP(3);
        const char* qcur = qtile + (i0 & 127) * 128;
        const char* docur = dotile + par * 4096;
        // the score accumulator starts as the position term (S = bias + (q + u) . k comes out of the MFMA chain itself).  A key that does not
        // exist gets no special value: its lane's scores only ever reach its own columns of dK / dV, which are not stored, and its slab stores
        // carry an out-of-range offset
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
        STAMP(4);
        if (!(p.debug & 8)) {
            // G = Qsel . Eext_window^T + cext: two 32-column blocks of the wave's 63-column window (rows = queries, lane = window column)
            const int pe_w = L - 32 - i0 + jw0 + 32 * wave;                 // p' of the wave's window column 0
            f32x16 g0, g1;
            const bool all_low = pe_w + 63 <= L - 1, all_up = pe_w >= L + 1;
            if (all_low || all_up) {                                         // both blocks on one side of p' = L: two independent chains
                const int e0 = (pe_w + jj - L) & 255, e1 = (pe_w + 32 + jj - L) & 255;
                const float c0v = ctile[e0], c1v = ctile[e1];
#pragma unroll
                for (int r = 0; r < 16; ++r) { g0[r] = c0v; g1[r] = c1v; }
                const unsigned qb = (unsigned)((i0 + jj + (all_low ? 0 : 1)) & 127) * 128u, qsw = all_low ? sw0 : sw1;
                const unsigned e0b = (unsigned)e0 * 128u, e1b = (unsigned)e1 * 128u;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 qa = frag(ptile, qb, qsw, ks);
                    const bf16x8 f0 = frag(etile, e0b, sw0, ks);
                    const bf16x8 f1 = frag(etile, e1b, sw0, ks);
                    g0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, f0, g0, 0, 0, 0);
                    g1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, f1, g1, 0, 0, 0);
                }
            } else {
                auto gblock = [&](int blk) -> f32x16 {
                    const int pe0 = pe_w + 32 * blk;
                    const int erow = (pe0 + jj - L) & 255;                   // ring slot of this lane's table row
                    f32x16 g;
                    const float cv = ctile[erow];
#pragma unroll
                    for (int r = 0; r < 16; ++r) g[r] = cv;
                    const int qlo = (i0 + jj) & 127, qup = (i0 + jj + 1) & 127;
                    if (pe0 + 31 <= L - 1 || pe0 >= L + 1) {
                        const int qrow = pe0 >= L + 1 ? qup : qlo;
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) {
                            const bf16x8 qa = *reinterpret_cast<const bf16x8*>(ptile + T::off(qrow, 2 * ks + hh));
                            const bf16x8 ef = *reinterpret_cast<const bf16x8*>(etile + T::off(erow, 2 * ks + hh));
                            g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, ef, g, 0, 0, 0);
                        }
                    } else {                                                 // the block that holds p' = L: columns below it take q_i, columns above it q_{i+1}
                        const bool lower = pe0 + jj <= L - 1;
                        bf16x8 zero;
#pragma unroll
                        for (int e = 0; e < 8; ++e) zero[e] = (__bf16)0.0f;
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) {
                            const bf16x8 qa = *reinterpret_cast<const bf16x8*>(ptile + T::off(qlo, 2 * ks + hh));
                            const bf16x8 qb = *reinterpret_cast<const bf16x8*>(ptile + T::off(qup, 2 * ks + hh));
                            const bf16x8 ef = *reinterpret_cast<const bf16x8*>(etile + T::off(erow, 2 * ks + hh));
                            g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, lower ? ef : zero, g, 0, 0, 0);
                            g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qb, lower ? zero : ef, g, 0, 0, 0);
                        }
                    }
                    return g;
                };
                g0 = gblock(0);
                g1 = gblock(1);
            }
            // skew in registers: one bf16 pair (block 0 | block 1) per element, one cross-lane move per element
            int a4s = a4, jqs = jq;
            asm volatile("" : "+v"(a4s), "+v"(jqs));                        // (opaque per step: sixteen hoisted addresses and sixteen hoisted lane masks cost more registers than the kernel has)
            // (all sixteen moves are issued before the first result is used: taken one at a time each costs an LDS round trip)
            unsigned got[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cq = (r & 3) + 8 * (r >> 2);
                got[r] = (unsigned)__builtin_amdgcn_ds_bpermute((int)((((unsigned)(a4s - 4 * cq)) & 124u) | (unsigned)hb), (int)cvt_pk2(g0[r], g1[r]));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cq = (r & 3) + 8 * (r >> 2);
                s[r] = __uint_as_float(jqs > cq ? (got[r] & 0xffff0000u) : (got[r] << 16));
            }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 qa = frag(qcur, (unsigned)jj * 128u, sw0, ks);
            const bf16x8 da = frag(docur, (unsigned)jj * 128u, sw0, ks);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[ks], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vf[ks], dp, 0, 0, 0);
        }
        STAMP(5);
        const int s_ds = i0 * ldp * 2, s_dg = i0 * (ldp - 1) * 2;            // row part of the slab offsets (wave-uniform)
        // pad columns [L, ldp) of both bf16 slabs feed the K loop of the dq / dE products and must be zero: the lanes whose key index falls
        // there write the zeros (no separate strided memsets over B*H*L rows)
        if (!kvalid && j < ldp) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cq = (r & 3) + 8 * (r >> 2);
                if (i0 + cq + 4 * hh < L) {
                    ds16[(unsigned)((i0 + 4 * hh + cq) * ldp + j)] = 0;
                    dg16[(unsigned)((i0 + 4 * hh + cq) * ldp + j)] = 0;
                }
            }
        }
        const float* lse_c = lse_s + 32 * par + 4 * hh;
        const float* del_c = del_s + 32 * par + 4 * hh;
        const int* lo_c = lo_s + 32 * par + 4 * hh;                          // (MK == 4; the interval test is two compares and an OR: `||` compiles to a branch per element)
        const int* hi_c = hi_s + 32 * par + 4 * hh;
        const int jw = jw0 + 32 * wave;
        // this wave's 32 x 32 tile entirely on one side of the j == i + 1 diagonal, all 32 queries in range: one slab offset pair for the tile
        const bool interior = (i0 + 32 <= L) && (jw + 31 <= i0 || jw >= i0 + 33);
        if (interior) {
            const unsigned v_g = (p.debug & 2) ? OOB : (jw < i0 ? v_lo : v_hi);
            const unsigned v_d = (p.debug & 2) ? OOB : v_ds;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cq = (r & 3) + 8 * (r >> 2), q = cq + 4 * hh;
                float pr = __builtin_amdgcn_exp2f(fmaf(s[r], c2, -lse_c[cq]));
                float ds = pr * fmaf(dp[r], p.scale, -del_c[cq]);
                if constexpr (MK != 0) {
                    const bool msk = MK == 4 ? (bool)((int)(j < lo_c[cq]) | (int)(j > hi_c[cq])) : is_masked<MK>(p, b, i0 + q, jc);
                    pr = msk ? 0.f : pr;
                    ds = msk ? 0.f : ds;
                }
                s[r] = pr;
                dp[r] = ds;
                const bf16_t d16 = f32_to_bf16(ds);
                __builtin_amdgcn_raw_buffer_store_b16(d16, rs_ds, v_d, s_ds + cq * ldp * 2, 0);
                __builtin_amdgcn_raw_buffer_store_b16(d16, rs_dg, v_g, s_dg + cq * (ldp - 1) * 2, 0);
            }
        } else {
            const int dj = j - i0 - 4 * hh;                                  // j - i = dj - cq
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cq = (r & 3) + 8 * (r >> 2), q = cq + 4 * hh;
                const bool rowok = i0 + q < L;
                float pr = __builtin_amdgcn_exp2f(fmaf(s[r], c2, -lse_c[cq]));
                float ds = pr * fmaf(dp[r], p.scale, -del_c[cq]);
                bool dead = !rowok;
                if constexpr (MK != 0) dead = (bool)((int)dead | (int)(MK == 4 ? (bool)((int)(j < lo_c[cq]) | (int)(j > hi_c[cq])) : is_masked<MK>(p, b, min(i0 + q, L - 1), jc)));
                pr = dead ? 0.f : pr;
                ds = dead ? 0.f : ds;
                s[r] = pr;
                dp[r] = ds;
                const bf16_t d16 = f32_to_bf16(ds);
                const unsigned v_d = rowok ? v_ds : OOB;
                const unsigned v_g = (rowok && dj != cq + 1) ? (dj <= cq ? v_lo : v_hi) : OOB;
                __builtin_amdgcn_raw_buffer_store_b16(d16, rs_ds, v_d, s_ds + cq * ldp * 2, 0);
                __builtin_amdgcn_raw_buffer_store_b16(d16, rs_dg, v_g, s_dg + cq * (ldp - 1) * 2, 0);
            }
        }
        STAMP(6);
        if (p.debug & 16) {                                                  // (timing experiments: no dV / dK products)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dv[0][r] += s[r]; dk[0][r] += dp[r]; }
            return;
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 pb = pack8(s, 8 * s2);
            const bf16x8 dsb = pack8(dp, 8 * s2);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const bf16x8 a_do = tr_frag<DH>(docur, 16 * s2, 32 * dt, lane);
                const bf16x8 a_qu = tr_frag<DH>(qcur, 16 * s2, 32 * dt, lane);
                dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_do, pb, dv[dt], 0, 0, 0);
                dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_qu, dsb, dk[dt], 0, 0, 0);
            }
        }
        STAMP(7);
        STAMP_NEXT();
    };
    for (int i0 = ibeg; i0 < iend; i0 += 64) {
        step(i0, 0, i0 == ibeg);
        if (i0 + 32 < iend) step(i0 + 32, 1, false);
    }
#ifdef TTMI_STAMPS
    stamp_n = 17;
    STAMP(0);
#endif
    if (kvalid) {
        float* dkrow = p.dK + ((long)b * L + j) * p.ld_dkv + h * DH;
        float* dvrow = p.dV + ((long)b * L + j) * p.ld_dkv + h * DH;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = 32 * dt + 8 * g4 + 4 * hh;
                if (p.dK16) {      // bf16, the form the qkv dgrad / wgrad GEMMs read (no f32 copy, no conversion pass over dqkv)
                    const long o16 = ((long)b * L + j) * p.ld_dkv + h * DH + d;
                    uint2 wk, wv;
                    wk.x = cvt_pk2(dk[dt][4 * g4], dk[dt][4 * g4 + 1]); wk.y = cvt_pk2(dk[dt][4 * g4 + 2], dk[dt][4 * g4 + 3]);
                    wv.x = cvt_pk2(dv[dt][4 * g4], dv[dt][4 * g4 + 1]); wv.y = cvt_pk2(dv[dt][4 * g4 + 2], dv[dt][4 * g4 + 3]);
                    *reinterpret_cast<uint2*>(p.dK16 + o16) = wk;
                    *reinterpret_cast<uint2*>(p.dV16 + o16) = wv;
                } else {
                    *reinterpret_cast<float4*>(dkrow + d) = make_float4(dk[dt][4 * g4], dk[dt][4 * g4 + 1], dk[dt][4 * g4 + 2], dk[dt][4 * g4 + 3]);
                    *reinterpret_cast<float4*>(dvrow + d) = make_float4(dv[dt][4 * g4], dv[dt][4 * g4 + 1], dv[dt][4 * g4 + 2], dv[dt][4 * g4 + 3]);
                }
            }
    }
#ifdef TTMI_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(1);
#endif
}
#ifdef TTMI_STAMPS
}  // namespace
extern "C" int ttmi_debug_bwd_stamps(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bwd_stamps), sizeof(unsigned long long) * n);
}
namespace {
#endif

// (A third-generation backward - 16 keys per wave on 16x16x32 tiles, eight waves per workgroup, four waves per SIMD at <= 128 registers - was built and measured in
// round 5: correct on all 47 cases of tests/test_flash_gpu.py and SLOWER, 250 - 257 us against 163 - 164 us per C2 audio layer, with 28 - 35 spilled registers and a
// per-step dependency chain as long as this kernel's.  Its source is in the history at commit d7f784a (flash_bwd_rel3_kernel); DESIGN.md section 4j has the findings.)

// ------------------------------------------------------------------ backward (dK, dV, dS)
template <int DH, int MK>
__global__ __launch_bounds__(256, 2) void flash_bwd_kernel(const FlashParams p) {
    using T = Tile<DH>;
    constexpr int KS = DH / 16, DT = DH / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* qtile = smem;                          // (q+u) rows of the current query tile
    char* dotile = smem + 32 * T::ROWB;          // dO rows
    float* lse_s = reinterpret_cast<float*>(smem + 64 * T::ROWB);
    float* del_s = lse_s + 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
    const int z = blockIdx.y, b = z / p.H, h = z % p.H;
    const int L = p.L;
    const int j = blockIdx.x * 128 + wave * 32 + (lane & 31);
    const int jc = min(j, L - 1);
    const bool kvalid = j < L;
    const bf16_t* krow = p.k + ((long)b * L + jc) * p.ld_kv + h * DH;
    const bf16_t* vrow = p.v + ((long)b * L + jc) * p.ld_kv + h * DH;
    bf16x8 kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        kf[ks] = *reinterpret_cast<const bf16x8*>(krow + 16 * ks + 8 * hh);
        vf[ks] = *reinterpret_cast<const bf16x8*>(vrow + 16 * ks + 8 * hh);
    }
    f32x16 dk[DT], dv[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
    const bf16_t* qbase = p.qu + (long)b * L * p.ld_qu + h * DH;
    const bf16_t* dobase = p.dO + (long)b * L * p.ld_o + h * DH;
    const bf16_t* bd = p.bd + (long)z * p.slab;
    bf16_t* ds16 = p.dS16 + (long)z * p.slab16;
    bf16_t* dg16 = p.dG16 + (long)z * p.slab16;

    // bias of one 32-query tile for this lane's key: rows i0 + q_r, column jc (coalesced across the lanes)
    // (32-bit element offsets from the per-(b,h) slab bases: the 64-bit products per element were a third of the loop's VALU work)
    const int ldp = (int)p.ldp;
    // bias of a 32-query tile x the workgroup's 128 keys = 32 rows of 256 contiguous bytes: fetched one tile ahead with 8-byte
    // loads (4 per thread), parked in LDS, read back per lane (row i0 + q, column = own key).  The per-element global 2-byte
    // loads this replaces were 42 % of the kernel's time.
    bf16_t* btile = reinterpret_cast<bf16_t*>(del_s + 32);                   // [32][128] bf16
    int* lo_s = reinterpret_cast<int*>(btile + 32 * 128);                    // MK == 4: the tile's 32 (lo, hi) pairs
    int* hi_s = lo_s + 32;
    const int jw0 = blockIdx.x * 128;
    const bool bstage = (L % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.bd) & 7) == 0) && (p.slab % 4 == 0) && L >= 4;
    uint2 bpre[4];
    const int bias_lim = (L - 1) * L + jc;
    auto fetch_bias = [&](int i0) {
        if (!bstage || (p.debug & 1)) return;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = tid + 256 * k;                                     // chunk of 4 columns: row c >> 5, columns 4 (c & 31)
            const int row = min(i0 + (c >> 5), L - 1), col = min(jw0 + 4 * (c & 31), L - 4);
            bpre[k] = *reinterpret_cast<const uint2*>(bd + (unsigned)(row * L + col));
        }
    };
    auto park_bias = [&]() {
        if (!bstage || (p.debug & 1)) return;
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<uint2*>(btile + 4 * (tid + 256 * k)) = bpre[k];
    };
    auto read_bias = [&](int i0, float (&bv)[16]) {
        if (p.debug & 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) bv[r] = 0.f;
        } else if (bstage) {
            // columns beyond L - 1 were fetched from a clamped chunk: those lanes' keys do not exist (kvalid false), any value does
            const bf16_t* col = btile + wave * 32 + (lane & 31) + 4 * hh * 128;
#pragma unroll
            for (int r = 0; r < 16; ++r) bv[r] = bf16_to_f32(col[((r & 3) + 8 * (r >> 2)) * 128]);
        } else {
            const int b0 = (i0 + 4 * hh) * L + jc;
#pragma unroll
            for (int r = 0; r < 16; ++r) bv[r] = bf16_to_f32(bd[(unsigned)min(b0 + ((r & 3) + 8 * (r >> 2)) * L, bias_lim)]);
        }
    };
    // narrow structured masks (launcher's choice, p.bwd_skip): both bf16 slabs were zeroed up front and only the query tiles that can see
    // one of this workgroup's 128 keys are walked: query i meets key j iff i - left <= j <= i + right
    int ibeg = 0, iend = L;
    if (p.bwd_skip) {
        ibeg = max(0, jw0 - p.mask_right) & ~63;
        iend = (int)min((long)L, (long)jw0 + 127 + p.mask_left + 1);
        if (iend <= ibeg) { ibeg = 0; iend = 0; }
    }
    RowStage<DH, 32> stQ, stO;
    stQ.load(qbase, p.ld_qu, ibeg, L - 1, tid);
    stO.load(dobase, p.ld_o, ibeg, L - 1, tid);
    fetch_bias(ibeg);
    auto step = [&](int i0) {
        __syncthreads();
        stQ.store(qtile, tid);
        stO.store(dotile, tid);
        park_bias();
        if (tid < 32) {
            const int ii = min(i0 + tid, L - 1);
            lse_s[tid] = p.lse[(long)z * L + ii];
            del_s[tid] = p.delta[(long)z * L + ii];
            if constexpr (MK == 4) {
                const int* r = reinterpret_cast<const int*>(p.mask) + (long)b * p.mask_sb + 2 * ii;
                lo_s[tid] = r[0];
                hi_s[tid] = r[1];
            }
        }
        __syncthreads();
        if (i0 + 32 < iend) {                                       // next tile's operands and bias fly under this tile's MFMAs
            stQ.load(qbase, p.ld_qu, i0 + 32, L - 1, tid);
            stO.load(dobase, p.ld_o, i0 + 32, L - 1, tid);
            fetch_bias(i0 + 32);
        }
        float bcur[16];
        read_bias(i0, bcur);
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 qa = *reinterpret_cast<const bf16x8*>(qtile + T::off(lane & 31, 2 * ks + hh));
            const bf16x8 da = *reinterpret_cast<const bf16x8*>(dotile + T::off(lane & 31, 2 * ks + hh));
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[ks], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vf[ks], dp, 0, 0, 0);
        }
        // dS16[i][j] at i*ldp + j;  dG16[r][c-1], (r, c) = divmod((i+1) L + j, L+1): j <= i -> (i, L-i+j), j > i -> (i+1, j-i-1), i.e.
        // element i*(ldp-1) + (j <= i ? L-1+j : ldp+j-2), nothing for j == i+1 (c = 0)
        const int ds0 = (i0 + 4 * hh) * ldp + j, dg0 = (i0 + 4 * hh) * (ldp - 1);
        // pad columns [L, ldp) of both bf16 slabs feed the K loop of the dq / dE products and must be zero: the lanes whose key index falls
        // there write the zeros (no separate strided memsets over B*H*L rows)
        if (!kvalid && j < ldp && !(p.debug & 2)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cq = (r & 3) + 8 * (r >> 2);
                if (i0 + cq + 4 * hh < L) {
                    ds16[(unsigned)(ds0 + cq * ldp)] = 0;
                    dg16[(unsigned)(ds0 + cq * ldp)] = 0;
                }
            }
        }
        const int gsel_lo = L - 1 + j, gsel_hi = ldp + j - 2;
        // interior tiles (all 32 queries and all 128 keys of the workgroup in range, the j == i+1 diagonal not crossing the tile) take a
        // branch-free element loop; edge and diagonal tiles the general one
        const bool interior = (i0 + 32 <= L) && (jw0 > i0 + 32 || jw0 + 127 < i0 + 1) && !(p.debug & 2);
        if (interior) {
            // all 32 queries in range, the whole tile on one side of the j == i+1 diagonal: branch-free score loop, then ONE predicated
            // region for the lanes whose key exists
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = (r & 3) + 8 * (r >> 2) + 4 * hh;
                float pr = 0.f, ds = 0.f;
                if (!(MK == 4 ? (j < lo_s[q] || j > hi_s[q]) : is_masked<MK>(p, b, i0 + q, j))) {
                    const float sc = (s[r] + bcur[r]) * p.scale;
                    pr = __expf(sc - lse_s[q]);
                    ds = pr * (dp[r] - del_s[q]) * p.scale;
                }
                s[r] = kvalid ? pr : 0.f;
                dp[r] = kvalid ? ds : 0.f;
            }
            if (kvalid) {
                const int gsel = dg0 + (jw0 < i0 ? gsel_lo : gsel_hi);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cq = (r & 3) + 8 * (r >> 2);
                    const bf16_t d16 = f32_to_bf16(dp[r]);
                    ds16[(unsigned)(ds0 + cq * ldp)] = d16;
                    dg16[(unsigned)(gsel + cq * (ldp - 1))] = d16;
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cq = (r & 3) + 8 * (r >> 2);
                const int q = cq + 4 * hh;
                const int i = i0 + q;
                const bool inb = (i < L) && kvalid;
                float pr = 0.f, ds = 0.f;
                if (inb) {
                    if (!(MK == 4 ? (j < lo_s[q] || j > hi_s[q]) : is_masked<MK>(p, b, i, j))) {
                        const float sc = (s[r] + bcur[r]) * p.scale;
                        pr = __expf(sc - lse_s[q]);
                        ds = pr * (dp[r] - del_s[q]) * p.scale;
                    }
                    if (!(p.debug & 2)) {
                        const bf16_t d16 = f32_to_bf16(ds);
                        ds16[(unsigned)(ds0 + cq * ldp)] = d16;
                        if (j != i + 1) dg16[(unsigned)(dg0 + cq * (ldp - 1) + (j <= i ? gsel_lo : gsel_hi))] = d16;
                    }
                }
                s[r] = pr;
                dp[r] = ds;
            }
        }
        if (p.debug & 16) {                                         // (timing experiments: no dV / dK products)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dv[0][r] += s[r]; dk[0][r] += dp[r]; }
            return;
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 pb = pack8(s, 8 * s2);
            const bf16x8 dsb = pack8(dp, 8 * s2);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const bf16x8 a_do = tr_frag<DH>(dotile, 16 * s2, 32 * dt, lane);
                const bf16x8 a_qu = tr_frag<DH>(qtile, 16 * s2, 32 * dt, lane);
                dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_do, pb, dv[dt], 0, 0, 0);
                dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_qu, dsb, dk[dt], 0, 0, 0);
            }
        }
    };
    for (int i0 = ibeg; i0 < iend; i0 += 64) {
        step(i0);
        if (i0 + 32 < iend) step(i0 + 32);
    }
    if (kvalid) {
        float* dkrow = p.dK + ((long)b * L + j) * p.ld_dkv + h * DH;
        float* dvrow = p.dV + ((long)b * L + j) * p.ld_dkv + h * DH;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = 32 * dt + 8 * g4 + 4 * hh;
                if (p.dK16) {      // bf16, the form the qkv dgrad / wgrad GEMMs read (no f32 copy, no conversion pass over dqkv)
                    const long o16 = ((long)b * L + j) * p.ld_dkv + h * DH + d;
                    uint2 wk, wv;
                    wk.x = pack_bf16x2(dk[dt][4 * g4], dk[dt][4 * g4 + 1]); wk.y = pack_bf16x2(dk[dt][4 * g4 + 2], dk[dt][4 * g4 + 3]);
                    wv.x = pack_bf16x2(dv[dt][4 * g4], dv[dt][4 * g4 + 1]); wv.y = pack_bf16x2(dv[dt][4 * g4 + 2], dv[dt][4 * g4 + 3]);
                    *reinterpret_cast<uint2*>(p.dK16 + o16) = wk;
                    *reinterpret_cast<uint2*>(p.dV16 + o16) = wv;
                } else {
                    *reinterpret_cast<float4*>(dkrow + d) = make_float4(dk[dt][4 * g4], dk[dt][4 * g4 + 1], dk[dt][4 * g4 + 2], dk[dt][4 * g4 + 3]);
                    *reinterpret_cast<float4*>(dvrow + d) = make_float4(dv[dt][4 * g4], dv[dt][4 * g4 + 1], dv[dt][4 * g4 + 2], dv[dt][4 * g4 + 3]);
                }
            }
    }
}

// delta[z, i] = sum_d dO[b,i,h,d] * O[b,i,h,d].  DH / 8 adjacent lanes share one (b, i, h): every load instruction of a wave reads 1 KiB of consecutive bytes
// (round 6; one lane per head row read its 128 bytes in eight instructions that each touched 64 different lines: 16.8 us per C2 audio layer for 32 MB)
template <int DH>
__global__ __launch_bounds__(256) void flash_delta_kernel(const bf16_t* __restrict__ dO, const bf16_t* __restrict__ O, long ld, int B,
                                                          int L, int H, float* __restrict__ delta, float* __restrict__ zero_f32, long zero_n,
                                                          bf16_t* __restrict__ dG16, long slab16, int ldp) {
    constexpr int LPI = DH / 8;                                       // lanes per (b, i, h): 16 bytes each
    const long tidx = (long)blockIdx.x * 256 + threadIdx.x;
    // small zero fills of the backward pass that would otherwise be launches of their own (FlashParams::zero_*)
    const long nthreads = (long)gridDim.x * 256;
    for (long t = tidx; t < zero_n; t += nthreads) zero_f32[t] = 0.f;
    if (dG16) {
        const int w = ldp >> 1;                                      // row 0 of a slab as 32-bit words (ldp % 8 == 0, slabs 16-byte aligned)
        for (long t = tidx; t < (long)B * H * w; t += nthreads)
            reinterpret_cast<uint32_t*>(dG16 + (t / w) * slab16)[t % w] = 0u;
    }
    const long idx = tidx / LPI;                                      // (b, i, h)
    const int part = (int)(tidx % LPI);
    float acc = 0.f;
    const bool live = idx < (long)B * L * H;                          // (whole groups of LPI lanes are live or not: 256 % LPI == 0)
    const int h = live ? (int)(idx % H) : 0;
    const long bi = live ? idx / H : 0;
    if (live) {
        const uint4 x = *reinterpret_cast<const uint4*>(dO + bi * ld + h * DH + part * 8);
        const uint4 y = *reinterpret_cast<const uint4*>(O + bi * ld + h * DH + part * 8);
        const uint32_t xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            acc += __uint_as_float(xs[k] << 16) * __uint_as_float(ys[k] << 16) +
                   __uint_as_float(xs[k] & 0xffff0000u) * __uint_as_float(ys[k] & 0xffff0000u);
    }
#pragma unroll
    for (int o = 1; o < LPI; o <<= 1) acc += __shfl_xor(acc, o, 64);
    if (live && part == 0) {
        const int i = (int)(bi % L), b = (int)(bi / L);
        delta[((long)b * H + h) * L + i] = acc;
    }
}

}  // namespace

bool flash_supported(int Dh, long ld_qu, long ld_kv, long ld_o) {
    return (Dh == 32 || Dh == 64) && ld_qu % 8 == 0 && ld_kv % 8 == 0 && ld_o % 8 == 0;
}

// ------------------------------------------------------------------ dq, dE, dc, d r_w_bias from the two dS slabs in ONE pass
// The attention backward kernel leaves dS twice ([i][j] and the shifted [r][c-1] = dG).  Round 2 consumed them with three launches of the
// generic 128x128 GEMMs - dq = dS k + dG E (both slabs read), dE = sum_b dG^T q (dG read again), plus two transposes that made k and E
// K-major - at 3 and 1.7 TB/s: N = Dh = 64 fills half a tile and K = L = 500 is eight K-steps.  This kernel is shaped for the operands:
// one workgroup per (b, h), 8 waves; wave w owns columns [64 w, 64 w + 64) of both slabs - keys of dS, table rows of dG - and keeps
//   * its 64 x 64 slices of k and of the effective table E as MFMA B fragments in registers for the whole kernel (no transposes),
//   * its 64 rows of dE (and of dc) as MFMA accumulators for the whole kernel (one atomic flush per (b, h) at the end).
// The slabs stream through ONCE, 32 rows at a time, straight into A fragments (16-byte loads; the loads of block n + 1 are issued before
// the cross-wave reduction and the dE products of block n).  Per block: dq partial = dS k (its column sums = d r_w_bias) + dG E, summed
// over the 8 waves in a fixed order through LDS and stored as bf16 rows; dE += dG^T q and dc += dG^T 1 with dG^T / q^T fragments read
// transposed (ds_read_b64_tr_b16) from a wave-private image of the block.  L <= 512 (8 waves x 64 columns); longer sequences keep the GEMMs.
struct PosGradParams {
    const bf16_t* dS16; const bf16_t* dG16; long slab16; int ldp;
    const bf16_t* k; long ld_kv;        // key rows: k[(b L + j) ld_kv + h 64 + d]
    const bf16_t* e16; long ld_e;       // effective table rows: e16[p ld_e + h 64 + d]
    const bf16_t* qp; long ld_qp;       // plain q rows
    bf16_t* dq16; long ld_dq;           // out: dq16[(b L + i) ld_dq + h 64 + d]
    float* dE; long ld_de;              // += : dE[p ld_de + h 64 + d]
    float* dcT;                         // += : dcT[h L + p]
    float* gu;                          // += : gu[h 64 + d]  (d r_w_bias)
    int B, L, H;
    float* part;                        // column groups (L > 512): f32 partial dq rows part[((group B + b) L + i) H 64 + h 64 + d] instead of dq16
    float* g_emb; float* g_bias; int K; // if set: the flush goes straight into the table gradients r_emb [K, H, 64] / r_bias [K, H] (row p of the
                                        // effective table = table row max(0, p + K - L)); dE / dcT are not touched and no relpos_scatter launch follows
    float* part_e; float* part_c;       // if set: no atomics at all - this (b, h)'s rows leave by plain stores, part_e[(b L + p) H 64 + h 64 + d],
                                        // part_c[(b H + h) L + p]; table_grad_reduce_kernel sums them over b (attn_table_grads)
    int dbg;                            // timing experiments (TTMI_PG_DEBUG): 1 no final atomics, 2 no table loads, 4 no main loop, 8 no slab loads
};
// Raw-buffer accesses issued from inline asm (descriptor in four SGPRs + 32-bit lane offset + scalar offset; a lane offset past num_records reads as
// zero / is not stored, like the builtins).  Why asm: while a load to LDS that the COMPILER knows about is in flight it waits for vmcnt(0) in front
// of every LDS access it cannot prove disjoint from it - in this kernel that put three full drains INSIDE the prefetch of the next block
// (`buffer_load ... lds` x1, vmcnt(1), x3, vmcnt(0), the dG loads, vmcnt(0), the q load, vmcnt(0): found in the ISA in round 4) and another in
// front of the partial-sum stores; the prefetch never overlapped anything.  Issued from asm, the kernel's own wait at the top of a block is the
// only one.  (s_nop 4: an SGPR written by VALU - v_readlane of a spilled descriptor - needs 5 wait states before a vector-memory read.)
typedef unsigned u32x4s_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4s_t raw_rsrc(const void* base, unsigned bytes) {
    // (readfirstlane: the descriptor must LIVE in SGPRs; without it one build of this file allocated it to a VGPR tuple under the "s" constraint,
    // which the assembler rejects - the value is wave-uniform anyway)
    const unsigned long long a = (unsigned long long)(size_t)base;
    return u32x4s_t{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu)),
                    (unsigned)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
}
__device__ __forceinline__ void buf_dma16(u32x4s_t rsrc, unsigned voff, int soff, unsigned lds_wave_addr) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rsrc), "s"(soff), "s"(lds_wave_addr) : "memory", "m0");
}
__device__ __forceinline__ u32x4_t buf_ld16_async(u32x4s_t rsrc, unsigned voff, int soff) {
    u32x4_t v;
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=&v"(v) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    return v;
}
constexpr int PG_TILE = 32 * 64 * 2;
constexpr int PG_SLOT = 64 * 36;                                               // floats: a wave's partial dq block as [d][row], row pitch 36 (conflict-free b128)
constexpr int PG_LDS = 8 * PG_SLOT * 4 + 2 * 8 * PG_TILE + 2 * PG_TILE;        // partial slots | per-wave dS and dG images | two q tiles
__global__ __launch_bounds__(512, 1) void attn_dqde_kernel(const PosGradParams p) {
    using T = Tile<64>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* slots = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), hh = lane >> 5, row = lane & 31;
    char* stile = smem + 8 * PG_SLOT * 4 + wave * PG_TILE;                   // this wave's 32 x 64 block of dS ...
    char* gtile = smem + 8 * PG_SLOT * 4 + 8 * PG_TILE + wave * PG_TILE;      // ... and of dG (swizzled Tile<64> images)
    char* qtiles = smem + 8 * PG_SLOT * 4 + 16 * PG_TILE;
    const int z = blockIdx.x, b = z / p.H, h = z % p.H, L = p.L, ldp = p.ldp;
#ifdef TTMI_STAMPS
    const bool stamp_on = (p.dbg & 64) && blockIdx.x == 5 && tid == 0;
    int stamp_n = 0;
    STAMP(0);
    STAMP_NEXT();
#endif
    // sequences longer than 512: blockIdx.y = column group of 512 (keys of dS, table rows of dG); a group's dq is a partial sum, written in
    // f32 and summed over the groups by dq_group_sum_kernel; dE / dc rows belong to exactly one group
    const int c0 = 512 * (int)blockIdx.y + 64 * wave;
    const bf16_t* kb = p.k + (long)b * L * p.ld_kv + h * 64;
    const bf16_t* eb = p.e16 + h * 64;
    const bf16_t* qb = p.qp + (long)b * L * p.ld_qp + h * 64;
    const int lrow = lane >> 3, lpos = lane & 7;
    // slab blocks through raw buffers (descriptor + per-lane offset + scalar row-block offset; a row past L is past the slab and reads as
    // zero, a chunk past the pitch gets an offset outside the buffer): 8 lanes cover one row's 128 bytes (this wave's 64 columns), a
    // wave-instruction 8 whole row pieces, and the chunk a lane fetches is the one that belongs at its position of the swizzled image.
    // dS goes straight into its image by LDS-DMA (issued as soon as the previous block's fragments are in registers), dG through
    // registers (its image is read until the end of a block).
    const u32x4s_t rs_s = raw_rsrc(p.dS16 + (long)z * p.slab16, (unsigned)(p.slab16 * 2));
    const u32x4s_t rs_g = raw_rsrc(p.dG16 + (long)z * p.slab16, (unsigned)(p.slab16 * 2));
    const unsigned stile_lds = (unsigned)(size_t)stile;
    int voff[4];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        const int tr = 8 * q4 + lrow, col = c0 + 8 * (lpos ^ ((tr >> 1) & 7));
        voff[q4] = (col + 8 <= ldp && !(p.dbg & 8)) ? (tr * ldp + col) * 2 : 0x7FFFFF00;
    }
    u32x4_t rg[4];
    // One quarter (8 rows) of the next block's two slab tiles.  The quarters are issued at four points of a block, not in one burst: a CU takes
    // in ~10 bytes per cycle, a block's 68 KB are ~7k cycles of that, and a wave that issues into a full queue stands still - with all nine loads
    // of every wave in one place (round 3, and the first asm version) the eight waves queued for 6 - 8k cycles and THEN computed for 5k with the
    // memory pipeline idle (tools/debug/dqde_stamps.py).
    auto fetch_piece = [&](int q4, int i0) {
        const int soff = i0 * ldp * 2;
        buf_dma16(rs_s, (unsigned)voff[q4], soff, stile_lds + q4 * 1024);
        rg[q4] = buf_ld16_async(rs_g, (unsigned)voff[q4], soff);
    };
    auto fetch = [&](int i0) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) fetch_piece(q4, i0);
    };
    // resident B fragments B[k = column c][n = d] (lane n = 32 nt + row; k = 16 ks + 8 hh + 0..7) of this wave's 64 rows of k and of the
    // table: the rows go through a 64 x 64 image (the wave's partial-sum slot, not yet in use) and are read back by columns - once
    bf16x8 bk[2][4], be[2][4];
    {
        char* img = reinterpret_cast<char*>(slots + wave * PG_SLOT);
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const bf16_t* src = pass == 0 ? kb : eb;
            const long ld = pass == 0 ? p.ld_kv : p.ld_e;
            u32x4_t tv[8];
#pragma unroll
            for (int it = 0; it < 8; ++it)               // all eight loads in flight, then the (branch-free) zeroing and the image stores
                tv[it] = *reinterpret_cast<const u32x4_t*>(src + (long)min(c0 + 8 * it + lrow, L - 1) * ld + lpos * 8);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int r = 8 * it + lrow;
                const unsigned keep = (c0 + r < L && !(p.dbg & 2)) ? 0xffffffffu : 0u;
                const u32x4_t v = {tv[it][0] & keep, tv[it][1] & keep, tv[it][2] & keep, tv[it][3] & keep};
                *reinterpret_cast<u32x4_t*>(img + T::off(r, lpos)) = v;
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    bf16x8 f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int d = 32 * nt + row;
                        f[j] = __builtin_bit_cast(__bf16, *reinterpret_cast<const unsigned short*>(img + T::off(16 * ks + 8 * hh + j, d >> 3) + (d & 7) * 2));
                    }
                    if (pass == 0) bk[nt][ks] = f;
                    else be[nt][ks] = f;
                }
        }
    }
    f32x16 acc_e[2][2];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_e[0][0][r] = acc_e[0][1][r] = acc_e[1][0][r] = acc_e[1][1][r] = 0.f;
    float dc4[4] = {0.f, 0.f, 0.f, 0.f};           // lane (cg = lane & 15, rows 8 (lane >> 4) ..+7 of every block): dG column sums of columns 4 cg ..+3
    float gu_run[2] = {0.f, 0.f};
    // q rows in and dq rows out through raw buffers as well (a row past L reads as zero / is not stored; no 64-bit addresses in registers)
    const u32x4s_t rs_q = raw_rsrc(qb, (unsigned)(((long)(L - 1) * p.ld_qp + 64) * 2));
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(p.dq16 + (long)b * L * p.ld_dq + h * 64, 0, (int)(((long)(L - 1) * p.ld_dq + 64) * 2), 0x00020000);
    const int ldq2 = (int)p.ld_qp * 2, ldo2 = (int)p.ld_dq * 2;
    const unsigned vq = tid < 32 * T::NCH ? (unsigned)(((tid >> 3) & 31) * ldq2 + (tid & 7) * 16) : 0xFFFFFF00u;   // thread (row tid / 8, chunk tid % 8) of a 32-row q tile (the other half: out of range, reads zeros)
    const int vo = (4 * (tid >> 6)) * ldo2 + (tid & 63) * 2;               // thread (rows 4 (tid / 64) ..+3, column tid % 64) of a 32-row dq block
    u32x4_t qpre = {0u, 0u, 0u, 0u};               // 32 plain q rows -> swizzled tile, fetched one block ahead with the slab rows, parked at the top of their block
    auto load_q = [&](int i0) {
        qpre = buf_ld16_async(rs_q, vq, i0 * ldq2);        // (every thread: a load under `if` puts a join behind it, where the compiler drains)
    };
    auto park_q = [&](int buf) {
        if (tid < 32 * T::NCH) *reinterpret_cast<u32x4_t*>(qtiles + buf * PG_TILE + T::off(tid >> 3, tid & 7)) = qpre;
    };
    fetch(0);
    load_q(0);
    __syncthreads();                                // every wave is done with its slot as a table image
    int cur = 0;
    for (int i0 = 0; i0 < ((p.dbg & 4) ? 0 : L); i0 += 32) {
        STAMP(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this block's dS image, dG registers and q rows (issued one block ago, from asm: handed over here)
        asm volatile("" : "+v"(rg[0]), "+v"(rg[1]), "+v"(rg[2]), "+v"(rg[3]), "+v"(qpre));
        STAMP(1);
        park_q(cur);                                 // this block's q rows (read by the dE products behind the block's first barrier; the other buffer may still be read by a slow wave's previous block)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) *reinterpret_cast<u32x4_t*>(gtile + (8 * q4 + lrow) * 128 + lpos * 16) = rg[q4];
        bf16x8 a[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) a[ks] = *reinterpret_cast<const bf16x8*>(stile + T::off(row, 2 * ks + hh));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the fragments are in registers: the image may be overwritten
        STAMP(2);
        const bool more = i0 + 32 < L;               // (workgroup-uniform)
        if (more) fetch_piece(0, i0 + 32);
        STAMP(3);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {             // one 32-column half of the block at a time: 16 accumulator registers, not 32
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)           // content: dS k
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks], bk[nt][ks], acc, 0, 0, 0);
            float cs = 0.f;                          // column sums of the content part: d r_w_bias
#pragma unroll
            for (int r = 0; r < 16; ++r) cs += acc[r];
            gu_run[nt] += cs;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {         // position: dG E
                const bf16x8 ag = *reinterpret_cast<const bf16x8*>(gtile + T::off(row, 2 * ks + hh));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ag, be[nt][ks], acc, 0, 0, 0);
            }
            // the wave's partial block as [d][row]: a lane's four consecutive rows are one 16-byte store
            float* sl = slots + wave * PG_SLOT + (32 * nt + row) * 36 + 4 * hh;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                *reinterpret_cast<float4*>(sl + 8 * g4) = make_float4(acc[4 * g4], acc[4 * g4 + 1], acc[4 * g4 + 2], acc[4 * g4 + 3]);
            if (more) fetch_piece(1 + nt, i0 + 32);
        }
        STAMP(4);
        __syncthreads();
        STAMP(5);
        {
            const int d = tid & 63, m0 = 4 * (tid >> 6);           // this thread: column d, rows m0 ..+3; a wave = 4 whole rows of 128 bytes
            float4 sum = *reinterpret_cast<const float4*>(slots + d * 36 + m0);
#pragma unroll
            for (int w = 1; w < 8; ++w) {
                const float4 v = *reinterpret_cast<const float4*>(slots + w * PG_SLOT + d * 36 + m0);
                sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
            }
            if (p.part) {                                  // (workgroup-uniform)
                float* pr = p.part + (((long)blockIdx.y * p.B + b) * L + i0 + m0) * ((long)p.H * 64) + h * 64 + d;
                const long rp = (long)p.H * 64;
                if (i0 + m0 < L) pr[0] = sum.x;
                if (i0 + m0 + 1 < L) pr[rp] = sum.y;
                if (i0 + m0 + 2 < L) pr[2 * rp] = sum.z;
                if (i0 + m0 + 3 < L) pr[3 * rp] = sum.w;
            } else {
            const int so = i0 * ldo2;
            __builtin_amdgcn_raw_buffer_store_b16(f32_to_bf16(sum.x), rs_o, vo, so, 0);
            __builtin_amdgcn_raw_buffer_store_b16(f32_to_bf16(sum.y), rs_o, vo, so + ldo2, 0);
            __builtin_amdgcn_raw_buffer_store_b16(f32_to_bf16(sum.z), rs_o, vo, so + 2 * ldo2, 0);
            __builtin_amdgcn_raw_buffer_store_b16(f32_to_bf16(sum.w), rs_o, vo, so + 3 * ldo2, 0);
            }
        }
        if (more) { fetch_piece(3, i0 + 32); load_q(i0 + 32); }
        STAMP(6);
        const char* qt = qtiles + cur * PG_TILE;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 q0 = tr_frag<64>(qt, 16 * s2, 0, lane), q1 = tr_frag<64>(qt, 16 * s2, 32, lane);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const bf16x8 a = tr_frag<64>(gtile, 16 * s2, 32 * mt, lane);
                acc_e[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, q0, acc_e[mt][0], 0, 0, 0);
                acc_e[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, q1, acc_e[mt][1], 0, 0, 0);
            }
        }
        STAMP(7);
        // dc: 8 rows x 4 columns of the dG image per lane and block (8-byte reads); the four row groups meet once, after the loop
        {
            const int cg = lane & 15, r8 = 8 * (lane >> 4);
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const uint2 w = *reinterpret_cast<const uint2*>(gtile + T::off(r8 + rr, cg >> 1) + (cg & 1) * 8);
                dc4[0] += __uint_as_float(w.x << 16); dc4[1] += __uint_as_float(w.x & 0xffff0000u);
                dc4[2] += __uint_as_float(w.y << 16); dc4[3] += __uint_as_float(w.y & 0xffff0000u);
            }
        }
        STAMP(8);
        cur ^= 1;
        __syncthreads();
        STAMP(9);
        STAMP_NEXT();
    }
#ifdef TTMI_STAMPS
    stamp_n = 20;
    STAMP(0);
#endif
    // one flush per (b, h): rows of the tables this wave owns
    if (p.dbg & 1) return;
    if (p.part_e) {
        // 32 workgroups (the batch) hold rows of the SAME table gradient; as f32 atomics their 33 MB per C2 layer take 27 us of the kernel's 97 (the
        // chip's ~1.3 TB/s of added bytes).  Plain stores here, summed over b by a small kernel that runs beside the GEMMs that follow.
        float* pe = p.part_e + (long)b * L * ((long)p.H * 64) + h * 64 + row;
        const long rp = (long)p.H * 64;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pr = c0 + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (pr < L) {
                    pe[pr * rp] = acc_e[mt][0][r];
                    pe[pr * rp + 32] = acc_e[mt][1][r];
                }
            }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = dc4[e];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int pc = c0 + 4 * (lane & 15) + e;
            if (lane < 16 && pc < L) p.part_c[((long)b * p.H + h) * L + pc] = v;
        }
    } else if (p.g_emb) {
        // straight into the table gradients: effective row pr is table row pr + K - L; the rows below L - K all ARE table row 0 (sequences
        // longer than the table) and are summed in the wave before they leave - one add per wave and column, not one per row
        const int HD = p.H * 64, shift = p.K - L;
        float f0 = 0.f, f1 = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pr = c0 + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (pr < L) {
                    const int e = pr + shift;
                    if (e <= 0) { f0 += acc_e[mt][0][r]; f1 += acc_e[mt][1][r]; }
                    else {
                        atomicAdd(p.g_emb + (long)e * HD + h * 64 + row, acc_e[mt][0][r]);
                        atomicAdd(p.g_emb + (long)e * HD + h * 64 + 32 + row, acc_e[mt][1][r]);
                    }
                }
            }
        if (c0 + shift <= 0 && c0 < L) {                     // (wave-uniform: this wave owns rows that fold onto table row 0)
            f0 += __shfl_xor(f0, 32, 64);
            f1 += __shfl_xor(f1, 32, 64);
            if (hh == 0) {
                atomicAdd(p.g_emb + h * 64 + row, f0);
                atomicAdd(p.g_emb + h * 64 + 32 + row, f1);
            }
        }
        float cf = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = dc4[e];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int pc = c0 + 4 * (lane & 15) + e;
            if (lane < 16 && pc < L) {
                if (pc + shift <= 0) cf += v;
                else atomicAdd(p.g_bias + (long)(pc + shift) * p.H + h, v);
            }
        }
        if (c0 + shift <= 0 && c0 < L) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) cf += __shfl_xor(cf, o, 64);
            if (lane == 0) atomicAdd(p.g_bias + h, cf);
        }
    } else {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pr = c0 + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (pr < L) {
                atomicAdd(p.dE + (long)pr * p.ld_de + h * 64 + row, acc_e[mt][0][r]);
                atomicAdd(p.dE + (long)pr * p.ld_de + h * 64 + 32 + row, acc_e[mt][1][r]);
            }
        }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float v = dc4[e];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        const int pc = c0 + 4 * (lane & 15) + e;
        if (lane < 16 && pc < L) atomicAdd(p.dcT + (long)h * L + pc, v);
    }
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const float v = gu_run[nt] + __shfl_xor(gu_run[nt], 32, 64);
        if (hh == 0) atomicAdd(p.gu + h * 64 + 32 * nt + row, v);
    }
}

// dq16 rows = bf16(sum over the column groups of their f32 partial rows): four consecutive d per thread
__global__ __launch_bounds__(256) void dq_group_sum_kernel(const float* __restrict__ part, int ngroup, long rows, int hd, bf16_t* __restrict__ dq16, long ld_dq) {
    const long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= rows * hd) return;
    const long r = e / hd;
    const int c = (int)(e - r * hd);
    float4 acc = *reinterpret_cast<const float4*>(part + e);
    for (int g = 1; g < ngroup; ++g) {
        const float4 v = *reinterpret_cast<const float4*>(part + (long)g * rows * hd + e);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    uint2 w;
    w.x = cvt_pk2(acc.x, acc.y);
    w.y = cvt_pk2(acc.z, acc.w);
    *reinterpret_cast<uint2*>(dq16 + r * ld_dq + c) = w;
}

// Table gradients from the per-(b, h) rows attn_dqde_kernel stored: g_emb[max(0, p + K - L)][h][d] += sum_b part_e[b][p][h][d], g_bias likewise.
// One thread per four columns of an effective row; rows that map one to one onto a table row (e > 0) have a single owner and are added in
// place, the rows that clamp onto table row 0 (and row e = 0 itself) go by atomics (a few dozen rows where this path is used: L - K < 256).
__global__ __launch_bounds__(256) void table_grad_reduce_kernel(const float* __restrict__ part_e, const float* __restrict__ part_c, int B, int L, int H,
                                                                int K, float* __restrict__ g_emb, float* __restrict__ g_bias) {
    const int HD = H * 64;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long n4 = (long)L * HD / 4;
    if (idx < n4) {
        const int pr = (int)(idx / (HD / 4)), c4 = (int)(idx % (HD / 4)) * 4;
        // eight rows in flight per thread (round 6: one dependent load per batch entry made this pass latency-bound - 39 us per C2 audio layer for 33 MB);
        // the order of the sum over b is unchanged
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* src = part_e + (long)pr * HD + c4;
        const long sb = (long)L * HD;
        int b = 0;
        for (; b + 8 <= B; b += 8) {
            float4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float4*>(src + (b + k) * sb);
#pragma unroll
            for (int k = 0; k < 8; ++k) { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }
        }
        for (; b < B; ++b) {
            const float4 v = *reinterpret_cast<const float4*>(src + b * sb);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        const int e = pr + K - L;
        if (e > 0) {
            float4* g = reinterpret_cast<float4*>(g_emb + (long)e * HD + c4);
            float4 o = *g;
            o.x += acc.x; o.y += acc.y; o.z += acc.z; o.w += acc.w;
            *g = o;
        } else {
            atomicAdd(g_emb + c4, acc.x); atomicAdd(g_emb + c4 + 1, acc.y); atomicAdd(g_emb + c4 + 2, acc.z); atomicAdd(g_emb + c4 + 3, acc.w);
        }
    } else if (idx < n4 + (long)H * L) {
        const long i2 = idx - n4;
        const int h = (int)(i2 / L), pr = (int)(i2 % L);
        float acc = 0.f;
        int b = 0;
        for (; b + 8 <= B; b += 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = part_c[((long)(b + k) * H + h) * L + pr];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += v[k];
        }
        for (; b < B; ++b) acc += part_c[((long)b * H + h) * L + pr];
        const int e = pr + K - L;
        if (e > 0) g_bias[(long)e * H + h] += acc;
        else atomicAdd(g_bias + h, acc);
    }
}

int attn_table_grads(const float* part_e, const float* part_c, int B, int L, int H, int K, float* g_emb, float* g_bias, hipStream_t st) {
    TTMI_REQUIRE(part_e && part_c && g_emb && g_bias && B > 0 && L > 0 && H > 0 && K > 0 && aligned16(part_e) && aligned16(g_emb), "attn_table_grads: bad arguments");
    const long n = (long)L * H * 64 / 4 + (long)H * L;
    hipLaunchKernelGGL(table_grad_reduce_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, part_e, part_c, B, L, H, K, g_emb, g_bias);
    TTMI_LAUNCH_CHECK("table_grad_reduce_kernel");
    return TTMI_OK;
}

bool attn_dqde_supported(int Dh, int L, long ldp) { return Dh == 64 && ldp <= 8 * 512 && ldp % 8 == 0 && L >= 1; }
int attn_dqde_groups(long ldp) { return (int)((ldp + 511) / 512); }

int attn_dqde(const bf16_t* dS16, const bf16_t* dG16, long slab16, long ldp, const bf16_t* k, long ld_kv, const bf16_t* e16, long ld_e,
              const bf16_t* qp, long ld_qp, bf16_t* dq16, long ld_dq, float* dE, long ld_de, float* dcT, float* gu, int B, int L, int H,
              hipStream_t st, float* part, float* g_emb, float* g_bias, int K, float* part_e, float* part_c) {
    TTMI_REQUIRE(dS16 && dG16 && k && e16 && qp && dq16 && dE && dcT && gu && B > 0 && H > 0 && attn_dqde_supported(64, L, ldp),
                 "attn_dqde: bad arguments");
    const int ngroup = attn_dqde_groups(ldp);
    TTMI_REQUIRE(ngroup == 1 || (part && aligned16(part)), "attn_dqde: sequences longer than 512 need the partial-sum workspace (groups x B x L x H x 64 floats)");
    TTMI_REQUIRE(aligned16(dS16) && aligned16(dG16) && aligned16(qp) && (slab16 * 2) % 16 == 0 && ld_qp % 8 == 0 && ld_dq % 4 == 0 &&
                 (reinterpret_cast<uintptr_t>(dq16) & 7) == 0, "attn_dqde: alignment");
    static bool enabled = false;
    if (!enabled) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(attn_dqde_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, PG_LDS) != hipSuccess) {
            ttmi_set_error("attn_dqde: LDS attribute");
            return TTMI_EINVAL;
        }
        enabled = true;
    }
    PosGradParams p;
    p.dS16 = dS16; p.dG16 = dG16; p.slab16 = slab16; p.ldp = (int)ldp; p.k = k; p.ld_kv = ld_kv; p.e16 = e16; p.ld_e = ld_e; p.qp = qp; p.ld_qp = ld_qp;
    p.dq16 = dq16; p.ld_dq = ld_dq; p.dE = dE; p.ld_de = ld_de; p.dcT = dcT; p.gu = gu; p.B = B; p.L = L; p.H = H;
    p.part = ngroup > 1 ? part : nullptr;
    TTMI_REQUIRE(!g_emb || (g_bias && K >= 1), "attn_dqde: direct table-gradient flush needs r_emb, r_bias gradients and the table length");
    p.g_emb = g_emb; p.g_bias = g_bias; p.K = K;
    TTMI_REQUIRE(!part_e || (part_c && aligned16(part_e)), "attn_dqde: per-(b, h) table-gradient rows need both buffers");
    p.part_e = part_e; p.part_c = part_c;
    static int dbg = -1;
    if (dbg < 0) { const char* e = getenv("TTMI_PG_DEBUG"); dbg = e ? atoi(e) : 0; }
    p.dbg = dbg;
    hipLaunchKernelGGL(attn_dqde_kernel, dim3(B * H, ngroup), dim3(512), PG_LDS, st, p);
    TTMI_LAUNCH_CHECK("attn_dqde_kernel");
    if (ngroup > 1) {
        const long rows = (long)B * L;
        hipLaunchKernelGGL(dq_group_sum_kernel, dim3(cdiv(rows * H * 64 / 4, 256)), dim3(256), 0, st, part, ngroup, rows, H * 64, dq16, ld_dq);
        TTMI_LAUNCH_CHECK("dq_group_sum_kernel");
    }
    return TTMI_OK;
}

// ------------------------------------------------------------------ position-term slab
// G[z][i][1 + p] = q_i . E[p] + c[h][p], G[z][i][0] = 0, written as the pitch-(L+1) f32 slab whose pitch-L view is _rel_shift
// (see layers.hip).  This is a write-bound op (L*(L+1)*4 bytes per (b, h), K = Dh): one workgroup builds 16 complete rows in LDS
// and streams them out as ONE contiguous run of 16-byte stores (rows of the slab are back to back; the LDS image is shifted by the
// run's misalignment so that LDS and global float4s coincide).  Replaces a batched GEMM whose 128x128 tiles wrote 512-byte row
// pieces (1.5 TB/s) plus a strided memset for column 0.
int g_slab_dbg = 0;
constexpr int SLAB_RB = 4;        // 16-row blocks per workgroup: E_h (L x Dh, read by every block) stays in registers across them
// SINGLE: L <= 512, one pass of 8 column tiles per wave covers a row and the E fragments are loaded once per workgroup (otherwise
// they are re-read per row block and pass).  Kept under 128 VGPRs (4 workgroups per CU): at 244 the kernel was latency-bound.
template <int DH, bool SINGLE>
__global__ __launch_bounds__(256, 4) void relpos_slab_kernel(const bf16_t* __restrict__ q, long ld_q, const bf16_t* __restrict__ E,
                                                             long ld_e, const float* __restrict__ c, int L, int H,
                                                             bf16_t* __restrict__ G, int dbg) {
    extern __shared__ __attribute__((aligned(16))) bf16_t img[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int z = blockIdx.y, b = z / H, h = z % H;
    const int fr = lane & 15, fq = lane >> 4;
    constexpr int KS = DH / 32;
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    const int ntile = (L + 15) / 16;
    const float* cr = c + (long)h * L;
    bf16x8_t ef[8][KS];
    // 8 column tiles per wave and pass, all E fragments in flight before the first MFMA
    auto load_tiles = [&](int c0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bf16_t* er = E + (long)min((c0 + 4 * u) * 16 + fr, L - 1) * ld_e + h * DH + fq * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) ef[u][ks] = *reinterpret_cast<const bf16x8_t*>(er + ks * 32);
        }
    };
    auto load_bias = [&](int ct, float (&cb)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) cb[j] = cr[min(ct * 16 + fq * 4 + j, L - 1)];
    };
    if (SINGLE) load_tiles(wave);
    bf16x8_t qn[KS];
    auto load_q = [&](int r0) {
        const bf16_t* qr = q + ((long)b * L + min(r0 + fr, L - 1)) * ld_q + h * DH + fq * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qn[ks] = *reinterpret_cast<const bf16x8_t*>(qr + ks * 32);
    };
    load_q(blockIdx.x * SLAB_RB * 16);
    for (int rbi = 0; rbi < SLAB_RB; ++rbi) {
        const int r0 = (blockIdx.x * SLAB_RB + rbi) * 16;
        if (r0 >= L) break;
        const int nrows = min(16, L - r0);
        const long g0 = (long)z * L * (L + 1) + (long)r0 * (L + 1);   // first element of this block's run
        const int sh = (int)(g0 & 7);
        bf16x8_t qf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = qn[ks];
        load_q(r0 + 16);                                              // next block's rows fly under this block's work
        if (tid < 16) img[sh + tid * (L + 1)] = 0;
        for (int c0 = wave; c0 < ntile && !(dbg & 1); c0 += 32) {
            if (!SINGLE) load_tiles(c0);
            float cbA[4], cbB[4];                          // bias values one tile ahead (a load per tile after its MFMA would serialise)
            load_bias(c0, cbA);
            auto tile = [&](int u, const float (&cb)[4]) {
                const int ct = c0 + 4 * u;
                if (ct >= ntile) return;
                f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ef[u][ks], qf[ks], acc, 0, 0, 0);
                // acc[j] = G[r0 + fr][p = ct*16 + fq*4 + j]
                const int p0 = ct * 16 + fq * 4;
                bf16_t* dst = img + sh + fr * (L + 1) + 1 + p0;
                if (ct * 16 + 16 <= L) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) dst[j] = f32_to_bf16(acc[j] + cb[j]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (p0 + j < L) dst[j] = f32_to_bf16(acc[j] + cb[j]);
                }
            };
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                load_bias(c0 + 4 * (u + 1), cbB);
                tile(u, cbA);
                if (u + 2 < 8) load_bias(c0 + 4 * (u + 2), cbA);
                tile(u + 1, cbB);
            }
        }
        __syncthreads();
        const int n = nrows * (L + 1);                       // elements in the run; LDS index sh + e <-> global g0 + e
        bf16_t* gal = G + (g0 - sh);                         // 16-byte aligned (G itself is)
        for (int i8 = tid * 8; i8 < sh + n && !(dbg & 2); i8 += 2048) {
            if (i8 >= sh && i8 + 7 < sh + n) {
                *reinterpret_cast<uint4*>(gal + i8) = *reinterpret_cast<const uint4*>(img + i8);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (i8 + e >= sh && i8 + e < sh + n) gal[i8 + e] = img[i8 + e];
            }
        }
        // the image is rebuilt for the next row block: its LDS reads must have retired, the global stores need not (a
        // __syncthreads() here would drain them - vmcnt(0) - once per row block)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

void relpos_slab_set_debug(int bits) { g_slab_dbg = bits; }

int relpos_slab(const bf16_t* q, long ld_q, const bf16_t* E, long ld_e, const float* c, int B, int L, int H, int Dh, bf16_t* G,
                hipStream_t st) {
    TTMI_REQUIRE(q && E && c && G && B > 0 && L > 0 && H > 0 && (Dh == 32 || Dh == 64), "relpos_slab: bad arguments");
    TTMI_REQUIRE(aligned16(q) && aligned16(E) && aligned16(G) && ld_q % 8 == 0 && ld_e % 8 == 0, "relpos_slab: alignment");
    const size_t lds = ((size_t)16 * (L + 1) + 16) * sizeof(bf16_t);
    TTMI_REQUIRE(lds <= 160 * 1024 && (long)B * H <= 65535, "relpos_slab: L = %d too long for one LDS image (or B*H too large)", L);
    dim3 grid(cdiv(L, 16 * SLAB_RB), B * H);
    const bool single = L <= 512;
#define SLAB_LAUNCH(DHV, SV) do { \
        if (lds > 64 * 1024) { \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(relpos_slab_kernel<DHV, SV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) { ttmi_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; } \
        } \
        hipLaunchKernelGGL((relpos_slab_kernel<DHV, SV>), grid, dim3(256), lds, st, q, ld_q, E, ld_e, c, L, H, G, g_slab_dbg); } while (0)
    if (Dh == 64) { if (single) SLAB_LAUNCH(64, true); else SLAB_LAUNCH(64, false); }
    else { if (single) SLAB_LAUNCH(32, true); else SLAB_LAUNCH(32, false); }
#undef SLAB_LAUNCH
    TTMI_LAUNCH_CHECK("relpos_slab_kernel");
    return TTMI_OK;
}

int g_fwd_resident = 1;         // ttmi_set_option(14, 0): the round-2/3 forward kernel (four workgroups per head) for A/B measurements
void flash_set_resident(int v) { g_fwd_resident = v; }
int flash_attn_fwd(const FlashParams& p, hipStream_t st) {
    TTMI_REQUIRE((p.qu || p.e16) && p.k && p.v && (p.bd || p.e16) && p.o && p.lse, "flash_attn_fwd: null pointer");
    if (p.e16) {                                    // position term (and q + u) formed in the kernel: no slab, no q+u tensor
        TTMI_REQUIRE(p.qp && p.cT && p.u && flash_supported(p.Dh, p.ld_qp, p.ld_kv, p.ld_o) && p.ld_qp % 8 == 0 && p.ld_e % 8 == 0 && aligned16(p.qp) &&
                     aligned16(p.e16) && aligned16(p.k) && aligned16(p.v) && (reinterpret_cast<uintptr_t>(p.o) & 7) == 0,
                     "flash_attn_fwd: in-kernel position term needs 16-byte aligned q / E with pitches %% 8 == 0");
        if (p.Dh == 64 && p.L >= 192 && p.L <= 512 && g_fwd_resident) {
            // one workgroup per head, table resident in LDS (flash_fwd_res_kernel)
#define FWDS_LAUNCH(MKV) do { \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(flash_fwd_res_kernel<MKV>), hipFuncAttributeMaxDynamicSharedMemorySize, RES_LDS) != hipSuccess) { ttmi_set_error("flash_attn_fwd: LDS attribute"); return TTMI_EINVAL; } \
            hipLaunchKernelGGL((flash_fwd_res_kernel<MKV>), dim3(p.B * p.H), dim3(512), RES_LDS, st, p); } while (0)
            switch (p.mask_kind) {
                case 1: FWDS_LAUNCH(1); break;
                case 2: FWDS_LAUNCH(2); break;
                case 3: FWDS_LAUNCH(3); break;
                case 4: FWDS_LAUNCH(4); break;
                default: FWDS_LAUNCH(0); break;
            }
#undef FWDS_LAUNCH
            TTMI_LAUNCH_CHECK("flash_fwd_res_kernel");
            return TTMI_OK;
        }
        dim3 grid(cdiv(p.L, 128), p.B * p.H);
        TTMI_REQUIRE(grid.y <= 65535, "flash_attn_fwd: B*H too large");
#define FWDR_LAUNCH(MKV) do { \
        if (p.Dh == 64) { const int lds = (128 + 256) * 128 + 256 * 4 + 4 * 32 * 100 * 2; \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(flash_fwd_rel_kernel<64, MKV>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) { ttmi_set_error("flash_attn_fwd: LDS attribute"); return TTMI_EINVAL; } \
            hipLaunchKernelGGL((flash_fwd_rel_kernel<64, MKV>), grid, dim3(256), lds, st, p); } \
        else { const int lds = (128 + 256) * 64 + 256 * 4 + 4 * 32 * 100 * 2; \
            hipLaunchKernelGGL((flash_fwd_rel_kernel<32, MKV>), grid, dim3(256), lds, st, p); } } while (0)
        switch (p.mask_kind) {
            case 1: FWDR_LAUNCH(1); break;
            case 2: FWDR_LAUNCH(2); break;
            case 3: FWDR_LAUNCH(3); break;
            case 4: FWDR_LAUNCH(4); break;
            default: FWDR_LAUNCH(0); break;
        }
#undef FWDR_LAUNCH
        TTMI_LAUNCH_CHECK("flash_fwd_rel_kernel");
        return TTMI_OK;
    }
    TTMI_REQUIRE(flash_supported(p.Dh, p.ld_qu, p.ld_kv, p.ld_o), "flash_attn_fwd: unsupported head dim %d / pitches", p.Dh);
    TTMI_REQUIRE(aligned16(p.qu) && aligned16(p.k) && aligned16(p.v) && (reinterpret_cast<uintptr_t>(p.o) & 7) == 0, "flash_attn_fwd: alignment");
    dim3 grid(cdiv(p.L, 128), p.B * p.H);
    TTMI_REQUIRE(grid.y <= 65535, "flash_attn_fwd: B*H too large");
#define FWD_LAUNCH(MKV) do { if (p.Dh == 64) hipLaunchKernelGGL((flash_fwd_kernel<64, MKV>), grid, dim3(256), 2 * 64 * 128 + 128 * 68 * 2, st, p); \
                             else hipLaunchKernelGGL((flash_fwd_kernel<32, MKV>), grid, dim3(256), 2 * 64 * 64 + 128 * 68 * 2, st, p); } while (0)
    switch (p.mask_kind) {
        case 1: FWD_LAUNCH(1); break;
        case 2: FWD_LAUNCH(2); break;
        case 3: FWD_LAUNCH(3); break;
        case 4: FWD_LAUNCH(4); break;
        default: FWD_LAUNCH(0); break;
    }
#undef FWD_LAUNCH
    TTMI_LAUNCH_CHECK("flash_fwd_kernel");
    return TTMI_OK;
}

int g_bwd_gen = 2;              // ttmi_set_option(15, 1): the round-3 backward kernel (register-staged prefetch, LDS image skew) for A/B measurements
void flash_set_bwd_gen(int v) { g_bwd_gen = v; }
int flash_attn_bwd(const FlashParams& p, hipStream_t st) {
    TTMI_REQUIRE((p.qu || p.e16) && p.k && p.v && (p.bd || p.e16) && p.o && p.lse && p.dO && p.delta && p.dS16 && p.dG16 && p.dK && p.dV, "flash_attn_bwd: null pointer");
    TTMI_REQUIRE(!p.e16 || (p.qp && p.cT && p.u && p.ld_qp % 8 == 0 && p.ld_e % 8 == 0 && aligned16(p.qp) && aligned16(p.e16)),
                 "flash_attn_bwd: in-kernel position term needs 16-byte aligned q / E with pitches %% 8 == 0");
    TTMI_REQUIRE(p.ldp >= p.L && p.ldp % 8 == 0 && (long)p.L * (p.L + 1) < (1L << 31), "flash_attn_bwd: bad dS pitch / L too large");
    TTMI_REQUIRE(flash_supported(p.Dh, p.e16 ? p.ld_qp : p.ld_qu, p.ld_kv, p.ld_o) && p.ld_dkv % 4 == 0, "flash_attn_bwd: unsupported head dim %d / pitches", p.Dh);
    TTMI_REQUIRE((p.e16 || aligned16(p.qu)) && aligned16(p.k) && aligned16(p.v) && aligned16(p.dO) && aligned16(p.dK) && aligned16(p.dV), "flash_attn_bwd: alignment");
    const long n = (long)p.B * p.L * p.H;
    dim3 grid(cdiv(p.L, 128), p.B * p.H);
    FlashParams q = p;
    // band / interval masks that leave at most half of the query tiles to a key block: zero the two bf16 slabs once (they must hold zeros
    // wherever the mask cuts) and let the kernel skip the rest - 256 MB of memset against 3/4 of the kernel's loads, MFMAs and 2-byte stores
    // (kind 4: callers that know nothing pass -1 / -1; 0 / 0 from callers written before the bounds existed is treated the same way)
    const bool reach_known = p.mask_kind == 2 || (p.mask_kind == 4 && p.mask_left >= 0 && p.mask_right >= 0 && p.mask_left + p.mask_right > 0);
    if (reach_known && ((long)p.mask_left + p.mask_right + 1 + 128 + 64) * 2 <= p.L && !(p.debug & 4)) {
        const size_t bytes = (size_t)p.B * p.H * p.slab16 * sizeof(bf16_t);
        if (fill_zero(p.dS16, bytes, st) != TTMI_OK || fill_zero(p.dG16, bytes, st) != TTMI_OK) return TTMI_EINVAL;
        q.bwd_skip = 1;
    }
    bf16_t* zg = p.zero_dg_row0 ? p.dG16 : nullptr;
    if (p.Dh == 64) hipLaunchKernelGGL(flash_delta_kernel<64>, dim3(cdiv(n * 8, 256)), dim3(256), 0, st, p.dO, p.o, p.ld_o, p.B, p.L, p.H, p.delta, p.zero_f32, p.zero_f32 ? p.zero_n : 0L, zg, p.slab16, (int)p.ldp);
    else hipLaunchKernelGGL(flash_delta_kernel<32>, dim3(cdiv(n * 4, 256)), dim3(256), 0, st, p.dO, p.o, p.ld_o, p.B, p.L, p.H, p.delta, p.zero_f32, p.zero_f32 ? p.zero_n : 0L, zg, p.slab16, (int)p.ldp);
    // in-kernel position term: (64 + 32 + 64 + 256) tile rows + cext ring + lse / delta + lo / hi + 4 private images of [64][36] bf16
#define BWD_LAUNCH(MKV) do { \
        if (p.e16 && p.Dh == 64 && g_bwd_gen >= 2) { \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(flash_bwd_rel2_kernel<MKV>), hipFuncAttributeMaxDynamicSharedMemorySize, BWD2_LDS) != hipSuccess) { ttmi_set_error("flash_attn_bwd: LDS attribute"); return TTMI_EINVAL; } \
            hipLaunchKernelGGL((flash_bwd_rel2_kernel<MKV>), grid, dim3(256), BWD2_LDS, st, q); \
        } else if (p.e16) { \
            if (p.Dh == 64) { const int lds = 416 * 128 + 256 * 4 + 256 + 256 + 4 * 64 * 36 * 2 + 256; \
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(flash_bwd_rel_kernel<64, MKV>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) { ttmi_set_error("flash_attn_bwd: LDS attribute"); return TTMI_EINVAL; } \
                hipLaunchKernelGGL((flash_bwd_rel_kernel<64, MKV>), grid, dim3(256), lds, st, q); } \
            else hipLaunchKernelGGL((flash_bwd_rel_kernel<32, MKV>), grid, dim3(256), 416 * 64 + 256 * 4 + 256 + 256 + 4 * 64 * 36 * 2 + 256, st, q); \
        } else if (p.Dh == 64) hipLaunchKernelGGL((flash_bwd_kernel<64, MKV>), grid, dim3(256), 64 * 128 + 256 + 8192 + 256, st, q); \
        else hipLaunchKernelGGL((flash_bwd_kernel<32, MKV>), grid, dim3(256), 64 * 64 + 256 + 8192 + 256, st, q); } while (0)
    const bool probe = p.L >= 256;                   // timing probe 3: the audio encoder's backward kernel (the label encoder's is tiny)
    if (probe) ttmi_probe_begin(3, st);
    switch (p.mask_kind) {
        case 1: BWD_LAUNCH(1); break;
        case 2: BWD_LAUNCH(2); break;
        case 3: BWD_LAUNCH(3); break;
        case 4: BWD_LAUNCH(4); break;
        default: BWD_LAUNCH(0); break;
    }
#undef BWD_LAUNCH
    if (probe) ttmi_probe_end(3, st);
    TTMI_LAUNCH_CHECK("flash_bwd_kernel");
    return TTMI_OK;
}
