// Throughput GEMMs for the bf16 pipeline (gfx950).  Operands are bf16 in HBM, f32 accumulate.
//
//   gemm_nt_bf16   C[M,N] = A[M,K] . B[N,K]^T (+bias)      forward / dgrad (weights pre-transposed, tiny)
//   gemm_tn_bf16   C[M,N] += A[K,M]^T . B[K,N]             wgrad: both operands reduction-major; the transposed
//                                                           fragments come from ds_read_b64_tr_b16, no transposed copies
//
// Structure (both): 128x128x64 block tile, 4 waves (2x2) x (2x2 v_mfma_f32_32x32x16_bf16), operands staged
// HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip), double-buffered, one barrier per K-step with
// the next tile's loads in flight under the MFMAs.  LDS images are linear per wave-instruction (a glds
// constraint) and XOR-swizzled through the per-lane SOURCE address so every fragment read is conflict-free:
//   NT: 128-B rows [row][k]:   chunk' = chunk ^ ((row >> 1) & 7)      (ds_read_b128)
//   TN: 256-B rows [k][col]:   chunk' = chunk ^ ((k & 3) << 2)        (ds_read_b64_tr_b16)
// Workgroups are renumbered so the 8 that share an XCD (ids equal mod 8) walk a compact 8x8-tile window
// (A and B panels of ~2 MB each stay in that XCD's 4 MB L2).
#include "gemm_fast.h"
#include "ttmi.h"
#include <mutex>

void ttmi_probe_begin(int slot, hipStream_t st);
void ttmi_probe_end(int slot, hipStream_t st);

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int TM = 128, TN_ = 128, TK = 64, NTH = 256;
constexpr int TILE_B = 128 * 64 * 2;        // 16 KiB per operand tile
constexpr int GROUP_M = 8;

// 16 zero bytes in global memory: glds source for out-of-range K rows/chunks (the source address is per lane)
__device__ __attribute__((aligned(16))) const uint4 g_zero16 = {0u, 0u, 0u, 0u};

struct FP {
    const bf16_t* A;
    const bf16_t* B;
    void* C;
    const float* bias;
    const float* addend;
    const bf16_t* mask;
    int mask_mode;
    int nt;                                 // persistent kernels: 1 = streaming (nontemporal) 16-byte bf16 output stores
    int relu;
    float scale;
    DropSpec drop;
    int M, N, K;
    long lda, ldb, ldc;
    int tiles_m, tiles_n, splitk, ksteps;   // ksteps per split
    int atomic;
    int gm;                                 // TN: tiles per co-resident group along M
    float* colsum;                          // TN: if set, colsum[m] += sum_k A[k][m] (taken from the LDS tiles by the tn == 0 blocks)
    int nz2;                                // batch z = blockIdx.y = z1 * nz2 + z2
    long sA1, sA2, sB1, sB2, sC1, sC2, sV1, sV2;
    // NT only: optional second operand pair accumulated into the same tile, C = A.B^T + A2.B2^T (A2 has A's batch strides)
    const bf16_t* A2;
    const bf16_t* B2;
    int K2;
    int walk = 0;                           // v8 NT kernel: 1 = every XCD walks whole GROUP_M row groups of its own (groups xcd, xcd + 8, ...): an A panel enters ONE L2
    int kwrap = 0;                          // persistent NT kernels: K-tiles [kwrap, K / 64) re-read A's K-tiles [0, ..) against B2 (same pitch as B): C = A.(B + B2)^T, 0 = off
    long lda2, ldb2, sB1b, sB2b;
    float* colsum_mid;                      // column sums of the FIRST product (rows < M), atomically added; batch strides sV1 / sV2
    // v8 NT, LEAN 3 ("exp store"): C = bf16(exp(acc + bias - exp_shift)), zeros in columns [N, ldc); rowsum[(tile_n * 4 + wave column) * M + m] =
    // sum of the row's 64 stored-before-rounding values of that wave (the caller adds the nparts partials of a row)
    float* rowsum;
    int nparts;
    const float* exp_shift;                 // device scalar (nullable = 0): chosen by the caller without a host round trip
    // v8 NT, LEAN 4: tanh' mask (mask_mode 1) and a per-row factor rowscale[m] on the result; v8 TN: colsum weighted by csw[k] (bf16) instead of ones
    const float* rowscale;
    const bf16_t* csw;
};

__device__ __forceinline__ void glds4(const void* gsrc, char* lds_wave_base) {       // 4 bytes per lane: LDS address = base + lane * 4
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}
__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// tile id -> (tm, tn): XCD-aware bijective renumbering + grouped (GROUP_M tall) ordering
__device__ __forceinline__ void tile_of(int bid, int nwg, int tiles_m, int tiles_n, int& tm, int& tn) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    const int per_group = GROUP_M * tiles_n;
    const int group = id / per_group, in = id % per_group;
    const int first = group * GROUP_M;
    const int gsz = min(tiles_m - first, GROUP_M);
    tm = first + in % gsz;
    tn = in / gsz;
}

template <typename TC>
__device__ __forceinline__ void store_tile(const f32x16 (&acc)[2][2], const FP& p, TC* C, int bm, int bn, int wm, int wn,
                                           int lane, bool add_bias) {
    // plain epilogue (every wgrad, most forward GEMMs): decided once per workgroup, not per element
    if (!add_bias && !p.addend && !p.relu && !p.mask && p.drop.p <= 0.f) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = bn + wn * 64 + j * 32 + (lane & 31);
                if (n >= p.N) continue;
                const int m0 = bm + wm * 64 + i * 32 + 4 * (lane >> 5);
                TC* c0 = C + (long)m0 * p.ldc + n;
                const bool full = m0 + 28 < p.M;             // rows m0 + {0..3} + 8 {0..3}
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dm = (r & 3) + 8 * (r >> 2);
                    if (!full && m0 + dm >= p.M) continue;
                    if constexpr (sizeof(TC) == 4) {
                        if (p.atomic) atomicAdd(reinterpret_cast<float*>(c0) + (long)dm * p.ldc, acc[i][j][r]);
                        else reinterpret_cast<float*>(c0)[(long)dm * p.ldc] = acc[i][j][r];
                    } else {
                        reinterpret_cast<bf16_t*>(c0)[(long)dm * p.ldc] = f32_to_bf16(acc[i][j][r]);
                    }
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = bn + wn * 64 + j * 32 + (lane & 31);
            if (n >= p.N) continue;
            const float bv = add_bias ? p.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = bm + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= p.M) continue;
                float v = acc[i][j][r] + bv;
                const long ci = (long)m * p.ldc + n;
                if (p.addend) v += p.addend[ci];
                if (p.relu) v = fmaxf(v, 0.f);
                if (p.mask) {
                    const float mv = bf16_to_f32(p.mask[ci]);
                    v = p.mask_mode ? v * (1.f - mv * mv) : (mv > 0.f ? v * p.scale : 0.f);
                }
                v *= drop_mult(p.drop, (unsigned long long)ci);
                if constexpr (sizeof(TC) == 4) {
                    if (p.atomic) atomicAdd(reinterpret_cast<float*>(C) + ci, v);
                    else reinterpret_cast<float*>(C)[ci] = v;
                } else {
                    reinterpret_cast<bf16_t*>(C)[ci] = f32_to_bf16(v);
                }
            }
        }
}

// ------------------------------------------------------------------ NT: A[M,K], B[N,K], K contiguous in both
template <typename TC, int NBUF, bool PIPE, bool DUAL = false>
__global__ __launch_bounds__(NTH, NBUF == 1 ? 4 : 2) void gemm_nt_bf16_kernel(const FP p_) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [NBUF buffers][A 16K | B 16K]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    FP p = p_;
    p.drop = drop_live(p.drop);
    int tm, tn;
    tile_of(blockIdx.x, gridDim.x, p.tiles_m, p.tiles_n, tm, tn);
    const int bm = tm * TM, bn = tn * TN_;
    {
        const int z1 = blockIdx.y / p.nz2, z2 = blockIdx.y % p.nz2;
        p.A += z1 * p.sA1 + z2 * p.sA2;
        p.B += z1 * p.sB1 + z2 * p.sB2;
        const long co = z1 * p.sC1 + z2 * p.sC2;
        p.C = static_cast<char*>(p.C) + co * (long)sizeof(TC);
        if (p.addend) p.addend += co;
        if (p.mask) p.mask += co;
        if (p.bias) p.bias += z1 * p.sV1 + z2 * p.sV2;
        if (DUAL) {
            p.A2 += z1 * p.sA1 + z2 * p.sA2;
            p.B2 += z1 * p.sB1b + z2 * p.sB2b;
            if (p.colsum_mid) p.colsum_mid += z1 * p.sV1 + z2 * p.sV2;
        }
    }

    // staging: wave w, instruction j covers tile rows R = (4w + j) * 8 + (lane >> 3), LDS slot = lane & 7,
    // which must hold source chunk slot ^ ((R >> 1) & 7)
    const bf16_t* asrc[4];
    const bf16_t* bsrc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int R = (wave * 4 + j) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((R >> 1) & 7);
        const int ra = min(bm + R, p.M - 1), rb = min(bn + R, p.N - 1);
        asrc[j] = p.A + (long)ra * p.lda + chunk * 8;
        bsrc[j] = p.B + (long)rb * p.ldb + chunk * 8;
    }
    int kchunk[4];                                   // source chunk (k offset / 8) of this lane per instruction
#pragma unroll
    for (int j = 0; j < 4; ++j) kchunk[j] = (lane & 7) ^ ((((wave * 4 + j) * 8 + (lane >> 3)) >> 1) & 7);
    const void* zsrc = &g_zero16;
    auto issue = [&](int buf, int kt) {
        char* base = smem + buf * 2 * TILE_B + wave * 4096;
        const bool tail = (kt + 1) * TK > p.K;       // wave-uniform: only the last K-step can be partial
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool zero = tail && (kt * TK + kchunk[j] * 8 >= p.K);
            glds16(zero ? zsrc : (const void*)(asrc[j] + (long)kt * TK), base + j * 1024);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool zero = tail && (kt * TK + kchunk[j] * 8 >= p.K);
            glds16(zero ? zsrc : (const void*)(bsrc[j] + (long)kt * TK), base + TILE_B + j * 1024);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (p.K + TK - 1) / TK;
    const int sw = (lane >> 1) & 7;                 // ((row >> 1) & 7) for row = ... + (lane & 31)
    const int rowa = wm * 64 + (lane & 31), rowb = wn * 64 + (lane & 31);
    auto frag = [&](const char* la, const char* lb, int kk, bf16x8 (&af)[2], bf16x8 (&bf)[2]) {
        const int c = ((kk * 2 + (lane >> 5)) ^ sw) << 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            af[i] = *reinterpret_cast<const bf16x8*>(la + (rowa + i * 32) * 128 + c);
            bf[i] = *reinterpret_cast<const bf16x8*>(lb + (rowb + i * 32) * 128 + c);
        }
    };
    auto mma = [&](const bf16x8 (&af)[2], const bf16x8 (&bf)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    };
    auto compute = [&](const char* la, const char* lb) {
        if constexpr (PIPE) {
            bf16x8 a0[2], b0[2], a1[2], b1[2];
            frag(la, lb, 0, a0, b0);
            frag(la, lb, 1, a1, b1);
            __builtin_amdgcn_s_setprio(1);
            mma(a0, b0);
            __builtin_amdgcn_s_setprio(0);
            frag(la, lb, 2, a0, b0);
            __builtin_amdgcn_s_setprio(1);
            mma(a1, b1);
            __builtin_amdgcn_s_setprio(0);
            frag(la, lb, 3, a1, b1);
            __builtin_amdgcn_s_setprio(1);
            mma(a0, b0);
            mma(a1, b1);
            __builtin_amdgcn_s_setprio(0);
        } else {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                bf16x8 af[2], bf[2];
                frag(la, lb, kk, af, bf);
                mma(af, bf);
            }
        }
    };
    if constexpr (NBUF == 2) {
        issue(0, 0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) issue(cur ^ 1, kt + 1);
            const char* la = smem + cur * 2 * TILE_B;
            compute(la, la + TILE_B);
            __syncthreads();       // drains the prefetch (vmcnt(0)) and fences the buffer swap
        }
    } else {
        // single 32 KiB buffer, two barriers per K-step: overlap comes from 4 co-resident workgroups per CU
        for (int kt = 0; kt < nk; ++kt) {
            issue(0, kt);
            __syncthreads();
            compute(smem, smem + TILE_B);
            __syncthreads();
        }
        if constexpr (DUAL) {
            // column sums of the first product over the tile's valid rows (attention backward: d r_w_bias = sum_i dq_content[i])
            if (p.colsum_mid) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float cs = 0.f;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (bm + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) < p.M) cs += acc[i][j][r];
                    cs += __shfl_xor(cs, 32, 64);
                    const int n = bn + wn * 64 + j * 32 + (lane & 31);
                    if (lane < 32 && n < p.N) atomicAdd(p.colsum_mid + n, cs);
                }
            }
            // second operand pair into the same accumulators
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int R = (wave * 4 + j) * 8 + (lane >> 3);
                const int chunk = (lane & 7) ^ ((R >> 1) & 7);
                asrc[j] = p.A2 + (long)min(bm + R, p.M - 1) * p.lda2 + chunk * 8;
                bsrc[j] = p.B2 + (long)min(bn + R, p.N - 1) * p.ldb2 + chunk * 8;
            }
            p.K = p.K2;
            const int nk2 = (p.K2 + TK - 1) / TK;
            for (int kt = 0; kt < nk2; ++kt) {
                issue(0, kt);
                __syncthreads();
                compute(smem, smem + TILE_B);
                __syncthreads();
            }
        }
    }
    store_tile<TC>(acc, p, reinterpret_cast<TC*>(p.C), bm, bn, wm, wn, lane, p.bias != nullptr);
}

// ------------------------------------------------------------------ TN: A[K,M], B[K,N], reduction index is the slow one
__device__ __forceinline__ bf16x4 ds_read_tr16(const char* lds_addr) {
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)lds_addr);
    return __builtin_bit_cast(bf16x4, v);
}
// The same read as asm.  In front of every ds_read_tr INTRINSIC the compiler puts s_waitcnt vmcnt(0) while LDS-DMA loads are in flight
// (it cannot tell their destination from the buffer being read), which drains a multi-tile prefetch at the top of every K-tile.  The
// pipelined kernels order LDS traffic themselves (counted vmcnt + barrier per tile), so they issue the read opaquely; the consumer
// must sit behind an explicit s_waitcnt lgkmcnt (LDS_TR_WAIT) - the compiler does not know these results are in flight.
#define LDS_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define LDS_TR_WAIT() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

template <int NBUF, bool CS>
__global__ __launch_bounds__(NTH, NBUF == 1 ? 4 : 2) void gemm_tn_bf16_kernel(const FP p_) {
    FP p = p_;
    {
        const int z1 = blockIdx.y / p.nz2, z2 = blockIdx.y % p.nz2;
        p.A += z1 * p.sA1 + z2 * p.sA2;
        p.B += z1 * p.sB1 + z2 * p.sB2;
        p.C = static_cast<char*>(p.C) + (z1 * p.sC1 + z2 * p.sC2) * (long)sizeof(float);
        if (p.colsum) p.colsum += z1 * p.sV1 + z2 * p.sV2;
    }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // Workgroups are dispatched round-robin over the 8 XCDs, so ids equal mod 8 share an L2.  With splitk a multiple of 8,
    // ks = id % splitk puts ONE K-range on each XCD: its workgroups stream the same rows of A and B (wgrad: tiny output,
    // huge K), instead of every XCD re-streaming most of both operands for its share of output tiles (PMC: 33 -> GB fetched).
    int ks, tm, tn;
    if (p.splitk % 8 == 0) {
        const int tid_flat = blockIdx.x / p.splitk;
        ks = blockIdx.x % p.splitk;
        // an XCD holds 32 CUs x 4 (NBUF 1) or 2 (NBUF 2) workgroups: make one co-resident wave of tiles GM tall x tiles_n wide
        const int GM = p.gm;
        const int per_group = GM * p.tiles_n;
        const int group = tid_flat / per_group, in = tid_flat % per_group;
        const int first = group * GM;
        const int gsz = min(p.tiles_m - first, GM);
        tm = first + in % gsz;
        tn = in / gsz;
    } else {
        const int ntile = p.tiles_m * p.tiles_n;
        ks = blockIdx.x / ntile;
        tile_of(blockIdx.x % ntile, ntile, p.tiles_m, p.tiles_n, tm, tn);
    }
    const int bm = tm * TM, bn = tn * TN_;

    // staging: one wave-instruction = 4 k-rows x 256 B.  wave w, instruction j: k-row kr = (4w + j) * 4 + (lane >> 4),
    // LDS slot = lane & 15 holds source chunk slot ^ ((kr & 3) << 2).  Column chunks are clamped into the row
    // (columns beyond M/N feed only output rows/cols that are never stored).
    const long k0 = (long)ks * p.ksteps * TK;
    const bf16_t* asrc[4];
    const bf16_t* bsrc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int kr = (wave * 4 + j) * 4 + (lane >> 4);
        const int chunk = (lane & 15) ^ ((kr & 3) << 2);
        const long ca = min((long)bm + chunk * 8, p.lda - 8), cb = min((long)bn + chunk * 8, p.ldb - 8);
        asrc[j] = p.A + (k0 + kr) * p.lda + ca;
        bsrc[j] = p.B + (k0 + kr) * p.ldb + cb;
    }
    const void* zsrc = &g_zero16;
    auto issue = [&](int buf, int kt) {
        char* base = smem + buf * 2 * TILE_B + wave * 4096;
        const long kbase = k0 + (long)kt * TK;
        const bool tail = kbase + TK > p.K;          // wave-uniform
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool zero = tail && (kbase + (wave * 4 + j) * 4 + (lane >> 4) >= p.K);
            glds16(zero ? zsrc : (const void*)(asrc[j] + (long)kt * TK * p.lda), base + j * 1024);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool zero = tail && (kbase + (wave * 4 + j) * 4 + (lane >> 4) >= p.K);
            glds16(zero ? zsrc : (const void*)(bsrc[j] + (long)kt * TK * p.ldb), base + TILE_B + j * 1024);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // transposed fragment read: 16-lane group g = lane >> 4 reads the 4 x 16 block (k rows 8h + 4s + q, 16 columns),
    // lane 4q + p of the group supplies the address of row q, columns 4p..4p+3; lane i receives column i.
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3, h = g >> 1;
    const int cola = wm * 64 + 16 * (g & 1) + 4 * pp, colb = wn * 64 + 16 * (g & 1) + 4 * pp;
    const int nk = min(p.ksteps, (int)((p.K - k0 + TK - 1) / TK));
    auto frag = [&](const char* la, const char* lb, int kk, bf16x8 (&af)[2], bf16x8 (&bf)[2]) {
        const int kr = kk * 16 + 8 * h + q;                    // (kr & 3) == q
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ca = cola + i * 32, cb = colb + i * 32;
            const int oa = (((ca >> 3) ^ (q << 2)) << 4) + (ca & 7) * 2, ob = (((cb >> 3) ^ (q << 2)) << 4) + (cb & 7) * 2;
            const bf16x4 a0 = ds_read_tr16(la + kr * 256 + oa), a1 = ds_read_tr16(la + (kr + 4) * 256 + oa);
            const bf16x4 b0 = ds_read_tr16(lb + kr * 256 + ob), b1 = ds_read_tr16(lb + (kr + 4) * 256 + ob);
            af[i] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
            bf[i] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    };
    auto mma = [&](const bf16x8 (&af)[2], const bf16x8 (&bf)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    };
    auto compute = [&](const char* la, const char* lb) {
        if constexpr (NBUF == 1) {
            // 4 workgroups per CU already overlap each other's LDS latency; one fragment set keeps the kernel under the
            // 128-VGPR budget of that occupancy without spills
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                bf16x8 a0[2], b0[2];
                frag(la, lb, kk, a0, b0);
                __builtin_amdgcn_s_setprio(1);
                mma(a0, b0);
                __builtin_amdgcn_s_setprio(0);
            }
        } else {
            bf16x8 a0[2], b0[2], a1[2], b1[2];
            frag(la, lb, 0, a0, b0);
            frag(la, lb, 1, a1, b1);
            __builtin_amdgcn_s_setprio(1);
            mma(a0, b0);
            __builtin_amdgcn_s_setprio(0);
            frag(la, lb, 2, a0, b0);
            __builtin_amdgcn_s_setprio(1);
            mma(a1, b1);
            __builtin_amdgcn_s_setprio(0);
            frag(la, lb, 3, a1, b1);
            __builtin_amdgcn_s_setprio(1);
            mma(a0, b0);
            mma(a1, b1);
            __builtin_amdgcn_s_setprio(0);
        }
    };
    // optional column sums of A (the bias gradient of the Linear whose wgrad this is), taken from the staged LDS tile.  The
    // tiles_n blocks that share an A tile split the K-steps between them (kt % tiles_n == tn), so no block is slower than the rest
    float cs = 0.f;
    auto colsum_tile = [&](const char* la, int kt) {
        if (CS && (kt % p.tiles_n) == tn) {
            const int col = tid & 127, r0 = (tid >> 7) * 32;
#pragma unroll 8
            for (int rr = 0; rr < 32; ++rr) {
                const int kr = r0 + rr;
                const float a = bf16_to_f32(*reinterpret_cast<const bf16_t*>(la + kr * 256 + ((((col >> 3) ^ ((kr & 3) << 2))) << 4) + (col & 7) * 2));
                // weighted form (the ragged M % 256 strip of the joint wgrad in the exp-domain loss form): weight of reduction row k0 + kt*64 + kr;
                // rows beyond K were staged as zeros, their weights are not read
                const long kg = k0 + (long)kt * TK + kr;
                cs += p.csw ? (kg < p.K ? a * bf16_to_f32(p.csw[kg]) : 0.f) : a;
            }
        }
    };
    if constexpr (NBUF == 2) {
        issue(0, 0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) issue(cur ^ 1, kt + 1);
            const char* la = smem + cur * 2 * TILE_B;
            compute(la, la + TILE_B);
            colsum_tile(la, kt);
            __syncthreads();
        }
    } else {
        for (int kt = 0; kt < nk; ++kt) {
            issue(0, kt);
            __syncthreads();
            compute(smem, smem + TILE_B);
            colsum_tile(smem, kt);
            __syncthreads();
        }
    }
    if (CS && bm + (tid & 127) < p.M) atomicAdd(p.colsum + bm + (tid & 127), cs);
    store_tile<float>(acc, p, reinterpret_cast<float*>(p.C), bm, bn, wm, wn, lane, false);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// epilogue of the 256-wide kernels for 4 consecutive columns n0..n0+3 of row m: bias, residual addend, ReLU, mask (ReLU' or tanh'),
// dropout, then one 16-byte (f32) / 8-byte (bf16) store (scalar fallback on ragged or unaligned edges)
template <typename TC>
__device__ __forceinline__ void epi_store4(const FP& p, TC* C, int m, int n0, f32x4 x, bool vec, bool bias_added = false) {
    if (m >= p.M || n0 >= p.N) return;
    const long ci = (long)m * p.ldc + n0;
    float v[4] = {x[0], x[1], x[2], x[3]};
    if (vec && n0 + 3 < p.N) {
        if (p.bias && !bias_added) {
            const float4 bv = *reinterpret_cast<const float4*>(p.bias + n0);   // n0 % 4 == 0; bias from hipMalloc/torch: 16-B aligned rows
            v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
        }
        if (p.addend) {
            const float4 a = *reinterpret_cast<const float4*>(p.addend + ci);
            v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
        }
        if (p.relu) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        if (p.mask) {
            const uint2 mk = *reinterpret_cast<const uint2*>(p.mask + ci);
            const unsigned short ms[4] = {(unsigned short)(mk.x & 0xffff), (unsigned short)(mk.x >> 16),
                                          (unsigned short)(mk.y & 0xffff), (unsigned short)(mk.y >> 16)};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float mv = bf16_to_f32(ms[j]);
                v[j] = p.mask_mode ? v[j] * (1.f - mv * mv) : (mv > 0.f ? v[j] * p.scale : 0.f);
            }
        }
        if (p.drop.p > 0.f) {
            float dm[4];
            drop_mult4(p.drop, (unsigned long long)ci, dm);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= dm[j];
        }
        if constexpr (sizeof(TC) == 4) {
            const f32x4 o = {v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(C) + ci) = o;
        } else {
            u32x2 o;
            o.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
            o.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
            *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(C) + ci) = o;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (n0 + j >= p.N) continue;
            float y = v[j] + (p.bias ? p.bias[n0 + j] : 0.f);
            if (p.addend) y += p.addend[ci + j];
            if (p.relu) y = fmaxf(y, 0.f);
            if (p.mask) {
                const float mv = bf16_to_f32(p.mask[ci + j]);
                y = p.mask_mode ? y * (1.f - mv * mv) : (mv > 0.f ? y * p.scale : 0.f);
            }
            y *= drop_mult(p.drop, (unsigned long long)(ci + j));
            if constexpr (sizeof(TC) == 4) reinterpret_cast<float*>(C)[ci + j] = y;
            else reinterpret_cast<bf16_t*>(C)[ci + j] = f32_to_bf16(y);
        }
    }
}

// the vector path of epi_store4 without the store: bias, residual, ReLU, mask, dropout on 4 consecutive columns (caller guarantees the
// alignment conditions of `vec`, m < M and n0 + 3 < N)
__device__ __forceinline__ void epi_math4(const FP& p, int m, int n0, f32x4 x, float (&v)[4]) {
    const long ci = (long)m * p.ldc + n0;
    v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
    if (p.bias) {
        const float4 bv = *reinterpret_cast<const float4*>(p.bias + n0);
        v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
    }
    if (p.addend) {
        const float4 a = *reinterpret_cast<const float4*>(p.addend + ci);
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
    }
    if (p.relu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
    }
    if (p.mask) {
        const uint2 mk = *reinterpret_cast<const uint2*>(p.mask + ci);
        const unsigned short ms[4] = {(unsigned short)(mk.x & 0xffff), (unsigned short)(mk.x >> 16),
                                      (unsigned short)(mk.y & 0xffff), (unsigned short)(mk.y >> 16)};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float mv = bf16_to_f32(ms[j]);
            v[j] = p.mask_mode ? v[j] * (1.f - mv * mv) : (mv > 0.f ? v[j] * p.scale : 0.f);
        }
    }
    if (p.drop.p > 0.f) {
        float dm[4];
        drop_mult4(p.drop, (unsigned long long)ci, dm);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= dm[j];
    }
}

// bf16 output, 8 consecutive columns of row m from two image chunks: one 16-byte store on the aligned interior, epi_store4 twice otherwise
__device__ __forceinline__ void epi_store8_bf16(const FP& p, bf16_t* C, int m, int n0, f32x4 x0, f32x4 x1, bool vec) {
    if (m < p.M && n0 + 7 < p.N) {
        float a[4], b[4];
        epi_math4(p, m, n0, x0, a);
        epi_math4(p, m, n0 + 4, x1, b);
        uint4 o;
        o.x = pack_bf16x2(a[0], a[1]); o.y = pack_bf16x2(a[2], a[3]); o.z = pack_bf16x2(b[0], b[1]); o.w = pack_bf16x2(b[2], b[3]);
        // outputs far larger than the L2 (the joint's logits, 7 GB) leave as streaming stores: they would only evict the operand panels
        // (joint forward 7.47 -> 7.2 ms)
        // (the scope bits only hurt: sc1 / sc0 sc1 / sc1 nt / sc0 sc1 nt 7.8-7.9 ms against 7.65 plain and 7.3 nt on the same box)
        if (p.nt) __builtin_nontemporal_store(u32x4{o.x, o.y, o.z, o.w}, reinterpret_cast<u32x4*>(C + (long)m * p.ldc + n0));
        else *reinterpret_cast<uint4*>(C + (long)m * p.ldc + n0) = o;
    } else {
        epi_store4<bf16_t>(p, C, m, n0, x0, vec);
        epi_store4<bf16_t>(p, C, m, n0 + 4, x1, vec);
    }
}

// =====================================================================================================================
// v8: persistent 256x256x64 kernel, one 512-thread workgroup per CU, 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 = 8 x 4
// v_mfma_f32_16x16x32_bf16 tiles (128 accumulator registers), 2 x 64 KiB operand buffers + 32 KiB epilogue staging in LDS.
// A K-tile is staged as four 16 KiB half-tiles (A rows with (row>>6)&1 = h, B columns with (col>>5)&1 = h) and consumed in
// two phases of 32 MFMAs (half the wave tile's rows x K=64):
//     phase A: read A(h0) B(h0) B(h1) | mma rows h0      phase B: read A(h1) | mma rows h1
// Every phase = [LDS reads + global_load_lds prefetch + s_waitcnt lgkmcnt(0)] s_barrier [32 MFMAs] s_barrier.  Waves 4-7 (the
// SIMD partners of waves 0-3) run one barrier behind, so in every barrier interval one wave per SIMD issues MFMAs while its
// partner reads LDS / issues the prefetch.  Loads are never drained inside the loop: a half-tile region is re-staged in the
// phase after its last read (A(h1) of K-tile t+1 in phase A, the other three of K-tile t+2 in phase B), and one counted
// s_waitcnt vmcnt(6) per K-tile (phase B: three half-tiles stay in flight) followed by a barrier orders the LDS-DMA of tile
// t+1 before its first read in the next phase.  (Stores of the previous output tile may still be counted by vmcnt: loads
// retire in order among themselves, so a counted wait can only be conservative.)
// Measured on MI355X (random operands): 8192^3 1505 TFLOP/s, joint dgrad (K=4352) 1264, joint forward (K=1024, 7 GB of bf16
// output) 957; with four phases of 16 MFMAs (8 barriers per K-tile) 1314 / 1190 / 933; v4 128x128: 1052 / 939 / 736.
// Output: the MFMA takes the B fragment as its row operand, so a lane holds 4 consecutive columns of one C row; each wave
// passes its accumulators 16 rows at a time through a private 4 KiB f32 LDS image (XOR-swizzled) and writes whole 128-byte
// (bf16) / 256-byte (f32) row segments.  The next output tile's first seven half-tiles are already in flight while the
// epilogue runs, and its stores drain under the next tile's MFMAs.
// =====================================================================================================================
constexpr int T8 = 256, NTH8 = 512, HT8 = 128 * 64 * 2, BUF8 = 4 * HT8;   // buffer: [A h0 | A h1 | B h0 | B h1]
constexpr int LDS8 = 2 * BUF8 + 8 * 4096;
constexpr int V8_STAGE_DMA = HT8 / (NTH8 * 16);            // LDS-DMA instructions per wave and half-tile stage (16 bytes per lane each)
constexpr int V8_INFLIGHT = 3 * V8_STAGE_DMA;              // what the counted waits of the v8 kernels leave in flight: three half-tiles
static_assert(V8_STAGE_DMA == 2 && V8_INFLIGHT == 6, "v8: the counted vmcnt waits assume 2 LDS-DMA instructions per half-tile stage");

// LEAN: 1 = the instance for bias-only bf16 outputs (the joint forward), 2 = for mask-only ones (the joint dgrad's tanh', ReLU'), 5 = bias + ReLU + dropout
// (the FFN's first Linear): their epilogues
// carry no residual / ReLU / dropout code and test nothing per store, which
// costs the main loop registers in the general instance
// KW: the K loop runs twice over A's K-tiles, the second time against p.B2 (two-term weights, NtEpilogue::B_lo) - its own instances, so that the
// default ones carry no trace of it (the scalar selects in stage() cost the joint's three GEMMs 0.1 - 0.2 ms per C2 step when they were unconditional)
template <typename TC, int LEAN, bool KW>
__device__ __forceinline__ void gemm_nt_v8_body(const FP& p_) {
    FP p = p_;
    p.drop = drop_live(p.drop);
    constexpr bool MASKED = LEAN == 2 || LEAN == 4;       // epilogues that read the mask operand (same layout as the output)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int nk = (p.K + TK - 1) / TK;

    // persistent tile walk: in every round the 32 workgroups of one XCD (ids equal mod 8) take 32 consecutive ids = an
    // 8-tall x 4-wide window of the GROUP_M-grouped order
    // p.walk = 1 (TTMI_TILE_WALK): the XCD owns the row groups xcd, xcd + 8, ... and its 32 workgroups walk them tile by tile - the 8 A panels of a group
    // are then fetched into one L2 instead of the four or five that share a group in the default walk (2.28 x the algorithmic traffic on the forward)
    const int w_nloc = gridDim.x >> 3, w_xcd = blockIdx.x & 7, w_loc = blockIdx.x >> 3;
    const int w_per_group = GROUP_M * p.tiles_n, w_ngroups = (p.tiles_m + GROUP_M - 1) / GROUP_M;
    const int w_own = w_ngroups > w_xcd ? (w_ngroups - w_xcd + 7) >> 3 : 0;
    const bool w_last = w_own > 0 && ((w_ngroups - 1) & 7) == w_xcd;      // the (possibly short) last group is this XCD's
    const long w_owned = (long)w_own * w_per_group - (w_last ? (long)(w_ngroups * GROUP_M - p.tiles_m) * p.tiles_n : 0);
    auto tile_id = [&](int it) -> long {
        if (p.walk) {
            const long s = (long)it * w_nloc + w_loc;
            if (s >= w_owned) return (long)p.tiles_m * p.tiles_n;          // past this XCD's share
            return ((long)w_xcd + 8 * (s / w_per_group)) * w_per_group + s % w_per_group;
        }
        // (grids that are no multiple of 8 - data-parallel backward leaves 256 - r CUs: the first gridDim.x & 7 XCDs hold one workgroup more)
        return (long)it * gridDim.x + (blockIdx.x & 7) * (gridDim.x >> 3) + min((int)(blockIdx.x & 7), (int)(gridDim.x & 7)) + (blockIdx.x >> 3);
    };
    auto coords = [&](long id, int& bm, int& bn) {
        const int per_group = GROUP_M * p.tiles_n;
        const int group = (int)(id / per_group), in = (int)(id % per_group);
        const int first = group * GROUP_M;
        const int gsz = min(p.tiles_m - first, GROUP_M);
        bm = (first + in % gsz) * T8;
        bn = (in / gsz) * T8;
    };
    // staging: instruction j of wave w fills half-tile rows rho = (2w + j) * 8 + (lane >> 3), 16-byte slot lane & 7, which must
    // hold source chunk slot ^ ((rho >> 1) & 7)
    // sources are a wave-uniform tile base (SGPRs) + a 32-bit per-lane byte offset: the saddr form of global_load_lds, no 64-bit VALU
    unsigned oA[2][2], oB[2][2];
    const char* baseA = nullptr;
    const char* baseB = nullptr;
    const char* baseB2 = nullptr;
    int kch[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) kch[j] = (lane & 7) ^ ((((wave * 2 + j) * 8 + (lane >> 3)) >> 1) & 7);
    auto sources = [&](int bm, int bn) {
        baseA = reinterpret_cast<const char*>(p.A + (long)bm * p.lda);
        baseB = reinterpret_cast<const char*>(p.B + (long)bn * p.ldb);
        if constexpr (KW) baseB2 = reinterpret_cast<const char*>(p.B2 + (long)bn * p.ldb);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rho = (wave * 2 + j) * 8 + (lane >> 3);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ra = min((rho >> 6) * 128 + h * 64 + (rho & 63), p.M - 1 - bm);
                const int rb = min((rho >> 5) * 64 + h * 32 + (rho & 31), p.N - 1 - bn);
                oA[h][j] = (unsigned)(((long)ra * p.lda + kch[j] * 8) * 2);
                oB[h][j] = (unsigned)(((long)rb * p.ldb + kch[j] * 8) * 2);
            }
        }
    };
    // kind: 0 = A h0, 1 = A h1, 2 = B h0, 3 = B h1 (also the region index inside a buffer)
    auto stage = [&](int kind, int buf, int kt) {
        char* dst = smem + buf * BUF8 + kind * HT8 + wave * 2048;
        const bool second = KW && kt >= p.kwrap;                                   // second weight term: A's K-tiles again, against B2
        const char* base = (kind < 2 ? baseA : second ? baseB2 : baseB) + (long)(second ? kt - p.kwrap : kt) * (TK * 2);      // K % 64 == 0 (checked by the launcher): no tail
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned o = kind == 0 ? oA[0][j] : kind == 1 ? oA[1][j] : kind == 2 ? oB[0][j] : oB[1][j];
            glds16(base + o, dst + j * 1024);
        }
    };
    auto prologue = [&]() {                       // K-tile 0 complete + three half-tiles of K-tile 1
        stage(0, 0, 0); stage(2, 0, 0); stage(3, 0, 0); stage(1, 0, 0);
        // K-tile 0 must have landed at the loop's first wait; tile 1's three half-tiles may fly (one K-tile only: that wait is a full drain)
        if (nk > 1) { TTMI_VM_GUARD("v8"); stage(0, 1, 1); stage(2, 1, 1); stage(3, 1, 1); }
    };

    // fragment addresses: row rho = wr*64 + mt*16 + (lane & 15) (A) / wc*32 + nt*16 + (lane & 15) (B); (rho >> 1) & 7 = (lane >> 1) & 7
    const int sw = (lane >> 1) & 7;
    int aoff[2], boff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int c = ((ks * 4 + (lane >> 4)) ^ sw) << 4;
        aoff[ks] = (wr * 64 + (lane & 15)) * 128 + c;
        boff[ks] = (wc * 32 + (lane & 15)) * 128 + c;
    }
    f32x4 acc[8][4];
    bf16x8 af[4][2], bfr[2][2][2];
    auto read_a = [&](const char* base, int h) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) af[mt][ks] = *reinterpret_cast<const bf16x8*>(base + h * HT8 + aoff[ks] + mt * 2048);
    };
    auto read_b = [&](const char* base, int h) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                bfr[h][nt][ks] = *reinterpret_cast<const bf16x8*>(base + (2 + h) * HT8 + boff[ks] + nt * 2048);
    };
    auto mma = [&](int mh, int nh) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mh * 4 + mt][nh * 2 + nt] =
                        __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nh][nt][ks], af[mt][ks], acc[mh * 4 + mt][nh * 2 + nt], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
#define V8_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define V8_BAR() __builtin_amdgcn_s_barrier()

    const int rounds = p.walk ? (int)((w_owned + w_nloc - 1) / w_nloc) : (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
    // (a per-XCD start stagger lived here: with the lean epilogue and streaming stores it measures as a 0.05 ms loss - phase offsets do not
    // persist, see DESIGN.md - and is gone)
    int bm = 0, bn = 0;
    bool live = tile_id(0) < ntiles;               // the ids of one round are a permutation of it*grid .. it*grid + grid - 1
    // LEAN 1 / 3 (bias and exp-store epilogues): the accumulators START as the bias of their column - nothing is added in the epilogue -
    // read from a per-wave LDS row (64 floats) that the previous tile's epilogue filled for this tile's columns.  Columns beyond N hold
    // -3e38 in the exp-store form: exp2 of it is the exact zero the padded pitch needs.
    constexpr bool BIAS_INIT = LEAN == 1 || LEAN == 3 || LEAN == 5;
    float* brow = reinterpret_cast<float*>(smem + 2 * BUF8 + 16384 + wave * 256);
    auto bias_of = [&](int bn_) -> float {
        const int col = bn_ + wc * 64 + lane;
        return col < p.N ? (p.bias ? p.bias[col] : 0.f) : (LEAN == 3 ? -3.0e38f : 0.f);
    };
    if (live) { coords(tile_id(0), bm, bn); sources(bm, bn); prologue(); }
    if constexpr (BIAS_INIT) { if (live) brow[lane] = bias_of(bn); }
    const float eshift2 = (LEAN == 3 && p.exp_shift ? *p.exp_shift : 0.f) * 1.4426950408889634f;
    for (int it = 0; it < rounds && live; ++it) {
        if constexpr (BIAS_INIT) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(brow + j * 16 + (lane >> 4) * 4);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i][j] = b4;
            }
        } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (nk > 1) TTMI_VM_WAIT("v8", V8_INFLIGHT);
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        V8_BAR();
        if (wr == 1) V8_BAR();                     // waves 4-7 run one barrier behind
        for (int t = 0; t < nk; ++t) {
            const int d = t & 1;
            const char* base = smem + d * BUF8;
            const bool more = t + 2 < nk;
            // phase A: rows h0 of the wave tile x all 64 columns
            read_a(base, 0);
            read_b(base, 0);
            read_b(base, 1);
            if (t + 1 < nk) stage(1, d ^ 1, t + 1);
            V8_LGKM0();                            // A(h0), B(h0), B(h1) are re-staged next phase: their reads must have retired
            V8_BAR();
            mma(0, 0);
            mma(0, 1);
            V8_BAR();
            // phase B: rows h1
            read_a(base, 1);
            if (more) {
                TTMI_VM_GUARD("v8");                // everything of K-tile t + 1 (A(h1) issued in phase A, the rest one tile ago) is older than this point
                stage(0, d, t + 2); stage(2, d, t + 2); stage(3, d, t + 2);
                TTMI_VM_WAIT("v8", V8_INFLIGHT);
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            V8_LGKM0();
            V8_BAR();
            mma(1, 1);
            mma(1, 0);
            V8_BAR();
        }
        if (wr == 0) V8_BAR();                     // realign: every wave has finished reading the operand buffers

        const int cbm = bm, cbn = bn;
        live = tile_id(it + 1) < ntiles;
        if (live) {
            coords(tile_id(it + 1), bm, bn);
            float nb = 0.f;
            if constexpr (BIAS_INIT) nb = bias_of(bn);     // requested before the operand prefetch, landed by the time that is issued
            sources(bm, bn);
            prologue();
            if constexpr (BIAS_INIT) brow[lane] = nb;      // this tile's row was consumed when its accumulators were set
        }

        // epilogue of (cbm, cbn): acc[mi][ni][j] = C[cbm + wr*128 + mi*16 + (lane & 15)][cbn + wc*64 + ni*16 + (lane >> 4)*4 + j]
        // -> private f32 image [16 rows][64 cols], 16-byte chunk c of row r at slot c ^ r -> rows of 256 B
        TC* C = reinterpret_cast<TC*>(p.C);
        char* img = smem + 2 * BUF8 + wave * 4096;
        const bool vec = (p.ldc % 4 == 0) && ((reinterpret_cast<size_t>(p.C) & 15) == 0) &&
                         (!p.addend || (reinterpret_cast<size_t>(p.addend) & 15) == 0) &&
                         (!p.mask || (reinterpret_cast<size_t>(p.mask) & 7) == 0) && (!p.bias || (reinterpret_cast<size_t>(p.bias) & 15) == 0);
        const int wrow = lane & 15, wq = lane >> 4;
        const bool plain8 = vec && p.ldc % 8 == 0;       // bf16 rows in 16-byte pieces (any epilogue)
        // the slab loop stays rolled (one copy of the epilogue code); the accumulators are picked by a wave-uniform switch so that
        // they are never indexed dynamically (which would put all 128 of them in scratch)
#define V8_SLAB(I) case I: _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) \
            *reinterpret_cast<f32x4*>(img + wrow * 256 + (((ni * 4 + wq) ^ wrow) << 4)) = acc[I][ni]; break;
        // LEAN: everything that does not depend on the slab is formed once per tile - the lane's column test and its row-0 output address
        const int ln0 = cbn + wc * 64 + (lane & 7) * 8;
        const bool lfull = ln0 + 7 < p.N;
        const int lm0 = cbm + wr * 128 + (lane >> 3);
        bf16_t* lrow0 = reinterpret_cast<bf16_t*>(p.C) + (long)lm0 * p.ldc + ln0;
        // LEAN 2: the mask vectors of slab mi + 1 are requested before slab mi goes through its LDS transposes and stores
        u32x4 mkc[2] = {u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}}, mkn[2] = {u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}};
        const bf16_t* lmask0 = p.mask + (long)lm0 * p.ldc + ln0;
        float rsc_c[2] = {0.f, 0.f}, rsc_n[2] = {0.f, 0.f};   // LEAN 4: the row factors travel with the mask vectors
        auto mask_fetch = [&](int mi, u32x4 (&dst)[2], float (&rs)[2]) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (lm0 + mi * 16 + q * 8 < p.M) {
                    if (lfull) dst[q] = *reinterpret_cast<const u32x4*>(lmask0 + (long)(mi * 16 + q * 8) * p.ldc);
                    if constexpr (LEAN == 4) rs[q] = p.rowscale[lm0 + mi * 16 + q * 8];
                }
        };
        if constexpr (LEAN == 1 || LEAN == 3 || LEAN == 5) {
            // bias / exp / bias + ReLU + dropout epilogues (the K = 1024 projection, where the un-overlapped epilogue is a third of the tile): the bias is already
            // in the accumulators, the values are rounded to bf16 BEFORE the transposing trip through LDS (half the LDS traffic of
            // the f32 image), and the trip of slab s overlaps the arithmetic of slab s + 1 (DS operations of a wave execute in order: the
            // next slab's writes cannot overtake this slab's reads of the same image).
            // Image: 16 rows x 128 B; the 8-byte piece c = 4 ni + g of row r sits at piece c ^ (r & ~1): the 32 lanes of a half wave write
            // 64 distinct banks, the 8 lanes that read one row back take its 128 bytes.
            const int g = lane >> 4;
            char* img = smem + 2 * BUF8 + wave * 2048;
            char* wimg = img + wrow * 128;
            const int wsw = wrow & ~1;
            const int rr = lane >> 3, c8 = lane & 7;
            const long part = (long)((cbn / T8) * 4 + wc) * p.M;
            u32x4 o[2];
#pragma unroll
            for (int sl = 0; sl <= 8; ++sl) {
                if (sl > 0) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int r = q * 8 + rr;
                        o[q] = *reinterpret_cast<const u32x4*>(img + r * 128 + (((2 * c8) ^ (r & ~1)) << 3));
                    }
                }
                if (sl < 8) {
                    f32x4 v[4] = {acc[sl & 7][0], acc[sl & 7][1], acc[sl & 7][2], acc[sl & 7][3]};     // unrolled: a static index
                    float rs = 0.f;
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) {
                        if constexpr (LEAN == 3) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                v[ni][j] = __builtin_amdgcn_exp2f(v[ni][j] * 1.4426950408889634f - eshift2);
                                rs += v[ni][j];
                            }
                        }
                        if constexpr (LEAN == 5) {        // ReLU, then the dropout mask of the lane's four consecutive columns (one hash word)
                            float dm[4];
                            drop_mult4(p.drop, (unsigned long long)(cbm + wr * 128 + sl * 16 + wrow) * p.ldc + (cbn + wc * 64 + ni * 16 + g * 4), dm);
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[ni][j] = fmaxf(v[ni][j], 0.f) * dm[j];
                        }
                        const uint2 w = {pack_bf16x2(v[ni][0], v[ni][1]), pack_bf16x2(v[ni][2], v[ni][3])};
                        *reinterpret_cast<uint2*>(wimg + (((ni * 4 + g) ^ wsw) << 3)) = w;
                    }
                    if constexpr (LEAN == 3) {
                        rs += __shfl_xor(rs, 16, 64);
                        rs += __shfl_xor(rs, 32, 64);
                        const int m = cbm + wr * 128 + sl * 16 + wrow;
                        if (g == 0 && m < p.M) p.rowsum[part + m] = rs;       // part-major: 16 consecutive rows per store
                    }
                }
                if (sl > 0) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int m = lm0 + (sl - 1) * 16 + q * 8;
                        if (m < p.M) {
                            bf16_t* dst = lrow0 + (long)((sl - 1) * 16 + q * 8) * p.ldc;
                            if (ln0 + 7 < (LEAN == 3 ? (int)p.ldc : p.N)) {
                                if (p.nt) __builtin_nontemporal_store(o[q], reinterpret_cast<u32x4*>(dst));
                                else *reinterpret_cast<u32x4*>(dst) = o[q];
                            } else {
#pragma unroll
                                for (int j = 0; j < 8; ++j)
                                    if (ln0 + j < (LEAN == 3 ? (int)p.ldc : p.N)) dst[j] = (bf16_t)(o[q][j >> 1] >> ((j & 1) * 16));
                            }
                        }
                    }
                }
            }
        } else {
        if constexpr (MASKED) mask_fetch(0, mkc, rsc_c);
        // general epilogue: the lane's four bias vectors (its 4-column chunk depends on the row of the 16-row slab only, not on the slab) are fetched ONCE per tile,
        // together - inside epi_store4 each of the 32 calls per lane and tile loaded its own and waited for it alone (round 6: the f32-output projection of the
        // bf16x3 mode spent 18.5 ms on the flops its dgrad does in 14)
        f32x4 gbias[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        bool ghoist[4] = {false, false, false, false};
        if constexpr (!LEAN) {
            if (p.bias && vec) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n0 = cbn + wc * 64 + (((lane & 15) ^ (q * 4 + (lane >> 4))) << 2);
                    ghoist[q] = n0 + 3 < p.N;
                    gbias[q] = *reinterpret_cast<const f32x4*>(p.bias + (ghoist[q] ? n0 : 0));
                    if (!ghoist[q]) gbias[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
#pragma unroll 1
        for (int mi = 0; mi < 8; ++mi) {
            if constexpr (MASKED) { if (mi + 1 < 8) mask_fetch(mi + 1, mkn, rsc_n); }
            switch (mi) { V8_SLAB(0) V8_SLAB(1) V8_SLAB(2) V8_SLAB(3) V8_SLAB(4) V8_SLAB(5) V8_SLAB(6) V8_SLAB(7) }
            if (LEAN || (sizeof(TC) == 2 && plain8)) {
                // bf16 output: 8 columns per lane, one 16-byte store - 8 rows x 128 B per instruction
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int r = q * 8 + (lane >> 3);
                    const int c8 = lane & 7;
                    const f32x4 x0 = *reinterpret_cast<const f32x4*>(img + r * 256 + (((2 * c8) ^ r) << 4));
                    const f32x4 x1 = *reinterpret_cast<const f32x4*>(img + r * 256 + (((2 * c8 + 1) ^ r) << 4));
                    if constexpr (MASKED) {
                        const int m = lm0 + mi * 16 + q * 8;
                        if (lfull) {
                            if (m < p.M) {
                                float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
                                if constexpr (LEAN == 2) {         // one 16-byte read of the mask operand (same layout as the output)
                                    const u32x4 mk = mkc[q];
#pragma unroll
                                    for (int j = 0; j < 4; ++j) {
                                        const float lo = __uint_as_float(mk[j] << 16), hi = __uint_as_float(mk[j] & 0xffff0000u);
                                        v[2 * j] = p.mask_mode ? v[2 * j] * (1.f - lo * lo) : (lo > 0.f ? v[2 * j] * p.scale : 0.f);
                                        v[2 * j + 1] = p.mask_mode ? v[2 * j + 1] * (1.f - hi * hi) : (hi > 0.f ? v[2 * j + 1] * p.scale : 0.f);
                                    }
                                }
                                if constexpr (LEAN == 4) {         // tanh' mask and a per-row factor; the mask operand leaves scaled by the same factor
                                    const u32x4 mk = mkc[q];
                                    const float rsc = rsc_c[q];
                                    u32x4 ms;
#pragma unroll
                                    for (int j = 0; j < 4; ++j) {
                                        const float lo = __uint_as_float(mk[j] << 16), hi = __uint_as_float(mk[j] & 0xffff0000u);
                                        v[2 * j] *= (1.f - lo * lo) * rsc;
                                        v[2 * j + 1] *= (1.f - hi * hi) * rsc;
                                        ms[j] = pack_bf16x2(lo * rsc, hi * rsc);
                                    }
                                    *reinterpret_cast<u32x4*>(const_cast<bf16_t*>(p.mask) + (long)m * p.ldc + ln0) = ms;
                                }
                                const u32x4 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
                                bf16_t* dst = lrow0 + (long)(mi * 16 + q * 8) * p.ldc;
                                if (p.nt) __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(dst));
                                else *reinterpret_cast<u32x4*>(dst) = o;
                            }
                        } else if (m < p.M) {            // the ragged last 8-column group of the matrix
#pragma unroll
                            for (int j = 0; j < 8; ++j)
                                if (ln0 + j < p.N)
                                {
                                    float y = j < 4 ? x0[j] : x1[j - 4];
                                    if constexpr (LEAN == 2) {
                                        const float mv = bf16_to_f32(p.mask[(long)m * p.ldc + ln0 + j]);
                                        y = p.mask_mode ? y * (1.f - mv * mv) : (mv > 0.f ? y * p.scale : 0.f);
                                    }
                                    if constexpr (LEAN == 4) {
                                        bf16_t* mp = const_cast<bf16_t*>(p.mask) + (long)m * p.ldc + ln0 + j;
                                        const float mv = bf16_to_f32(*mp), rsc = rsc_c[q];
                                        y *= (1.f - mv * mv) * rsc;
                                        *mp = f32_to_bf16(mv * rsc);
                                    }
                                    reinterpret_cast<bf16_t*>(C)[(long)m * p.ldc + ln0 + j] = f32_to_bf16(y);
                                }
                        }
                    } else if constexpr (sizeof(TC) == 2)
                        epi_store8_bf16(p, reinterpret_cast<bf16_t*>(C), cbm + wr * 128 + mi * 16 + r, cbn + wc * 64 + c8 * 8, x0, x1, vec);
                }
                if constexpr (MASKED) { mkc[0] = mkn[0]; mkc[1] = mkn[1]; rsc_c[0] = rsc_n[0]; rsc_c[1] = rsc_n[1]; }
            } else if constexpr (!LEAN) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = q * 4 + (lane >> 4);                    // row of the 16-row slab
                const int c = (lane & 15) ^ r;                        // logical 4-column chunk held by slot lane & 15
                f32x4 x = *reinterpret_cast<const f32x4*>(img + r * 256 + ((lane & 15) << 4));
                const int m = cbm + wr * 128 + mi * 16 + r;
                const int n0 = cbn + wc * 64 + c * 4;
                x += gbias[q];
                epi_store4<TC>(p, C, m, n0, x, vec, ghoist[q]);
            }
            }
        }
        }
    }
#undef V8_SLAB
#undef V8_LGKM0
#undef V8_BAR
}

template <typename TC, int LEAN = 0>
__global__ __launch_bounds__(NTH8, 1) void gemm_nt_bf16_v8_kernel(const FP p) { gemm_nt_v8_body<TC, LEAN, false>(p); }
template <typename TC, int LEAN = 0>          // two-term weights (NtEpilogue::B_lo): the K loop runs twice over A's K-tiles
__global__ __launch_bounds__(NTH8, 1) void gemm_nt_bf16_v8_kw_kernel(const FP p) { gemm_nt_v8_body<TC, LEAN, true>(p); }

// =====================================================================================================================
// v9: the v8 schedule for mid-sized outputs (the encoder GEMMs: 16000 rows x 512..2048 columns, K = 512..2048), where 256x256
// tiles leave CUs idle.  Persistent, 256 (M) x 128 (N) x 64 tile, 8 waves as 4 x 2, wave tile 64 x 64 = 4 x 4 MFMA tiles (64
// accumulator registers), THREE 48 KiB LDS stages and ONE phase per K-tile:
//     [16 ds_read_b128 + prefetch of K-tile t+2 + lgkmcnt(0) + vmcnt(6)] s_barrier [32 MFMAs] s_barrier
// waves 4-7 one barrier behind.  K-tile t+2 goes into the stage read in phase t-1 (retired before that phase's barrier), and the
// counted wait at the end of phase t leaves exactly its six loads in flight, so K-tile t+1 has landed before phase t+1 reads it.
// The epilogue images (4 KiB per wave, as in v8) live in stage 2, which the next output tile's prologue (K-tiles 0 and 1) does not
// touch; the barrier that opens the next tile's loop orders them before K-tile 2 is staged there.
// =====================================================================================================================
constexpr int T9M = 256, T9N = 128, STG9 = (T9M + T9N) * 64 * 2, LDS9 = 3 * STG9;
constexpr int V9_INFLIGHT = STG9 / (512 * 16);         // LDS-DMA instructions per wave (of 8) and stage = what the counted waits of the 3-stage kernels leave in flight
static_assert(V9_INFLIGHT == 6, "v9: the counted vmcnt waits assume 6 LDS-DMA instructions per stage");

// LEAN (as in v8): 1 = plain or bias-only epilogue (either output type), 2 = mask-only (bf16 output), 3 = f32 output + residual addend
// (+ bias); launcher-checked alignment
// F32IN: both operands are f32 in memory (p.A / p.B point at floats, lda / ldb count floats): a stage row is the same 128 bytes = 32 floats, every
// fragment read the same ds_read_b128 (4 consecutive k of one row), consumed by four v_mfma_f32_16x16x4_f32 - step s takes component s of both
// operands, i.e. the reduction runs in the order k = 16 c + 4 (lane >> 4) + s over the steps s of chunk c (any pairing of k is valid as long as A and
// B agree).  The exact-f32 path of the fp32 mode and of greedy decoding (csrc/gemm.hip routes its large NT problems here).
template <typename TC, int LEAN, bool F32IN, bool KW>            // KW: as in v8
__device__ __forceinline__ void gemm_nt_v9_body(const FP& p_) {
    constexpr int ES = F32IN ? 4 : 2, TKE = 128 / ES;      // operand element size, k per stage
    FP p = p_;
    p.drop = drop_live(p.drop);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave & 3, wc = wave >> 2;               // waves w and w+4 (SIMD partners) differ in the column half
    const int grp = wave >> 2;                             // 1: runs one barrier behind
    const int ntiles = p.tiles_m * p.tiles_n;
    const int nk = (p.K + TKE - 1) / TKE;

    // workgroups with equal id mod 8 share an XCD: they take consecutive tiles of every round; a grid that is no multiple of 8 (data-parallel backward leaves
    // 256 - r CUs to this kernel) has one workgroup more on its first gridDim.x & 7 XCDs
    auto tile_id = [&](int it) { return (long)it * gridDim.x + (blockIdx.x & 7) * (gridDim.x >> 3) + min((int)(blockIdx.x & 7), (int)(gridDim.x & 7)) + (blockIdx.x >> 3); };
    auto coords = [&](long id, int& bm, int& bn) {
        const int per_group = GROUP_M * p.tiles_n;
        const int group = (int)(id / per_group), in = (int)(id % per_group);
        const int first = group * GROUP_M;
        const int gsz = min(p.tiles_m - first, GROUP_M);
        bm = (first + in % gsz) * T9M;
        bn = (in / gsz) * T9N;
    };
    // staging: A = 32 units of 8 rows (wave w: units 4w..4w+3), B = 16 units (wave w: 2w, 2w+1); slot lane & 7 <- chunk slot ^ ((row >> 1) & 7)
    // (sources = wave-uniform tile base + 32-bit per-lane byte offsets: saddr form of global_load_lds; K % 64 == 0, no tail)
    unsigned oA[4], oB[2];
    const char* baseA = nullptr;
    const char* baseB = nullptr;
    const char* baseB2 = nullptr;
    auto sources = [&](int bm, int bn) {
        baseA = reinterpret_cast<const char*>(p.A) + (long)bm * p.lda * ES;
        baseB = reinterpret_cast<const char*>(p.B) + (long)bn * p.ldb * ES;
        if constexpr (KW) baseB2 = reinterpret_cast<const char*>(p.B2) + (long)bn * p.ldb * ES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = (wave * 4 + j) * 8 + (lane >> 3);
            oA[j] = (unsigned)(((long)min(r, p.M - 1 - bm) * p.lda * ES + ((lane & 7) ^ ((r >> 1) & 7)) * 16));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = (wave * 2 + j) * 8 + (lane >> 3);
            oB[j] = (unsigned)(((long)min(r, p.N - 1 - bn) * p.ldb * ES + ((lane & 7) ^ ((r >> 1) & 7)) * 16));
        }
    };
    auto stage = [&](int stg, int kt) {
        char* dst = smem + stg * STG9;
        const bool second = KW && kt >= p.kwrap;           // the weight's second bf16 term: the same A columns again, against B2
        const int ka = second ? kt - p.kwrap : kt;
        const char* ba = baseA + (long)ka * (TK * 2);
        const char* bb = (second ? baseB2 : baseB) + (long)ka * (TK * 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(ba + oA[j], dst + (wave * 4 + j) * 1024);
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(bb + oB[j], dst + T9M * 128 + (wave * 2 + j) * 1024);
    };
    const int sw = (lane >> 1) & 7;
    int aoff[2], boff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int c = ((ks * 4 + (lane >> 4)) ^ sw) << 4;
        aoff[ks] = (wr * 64 + (lane & 15)) * 128 + c;
        boff[ks] = T9M * 128 + (wc * 64 + (lane & 15)) * 128 + c;
    }
    f32x4 acc[4][4];
    using Frag = typename std::conditional<F32IN, f32x4, bf16x8>::type;
    Frag af[4][2], bfr[4][2];
#define V9_BAR() __builtin_amdgcn_s_barrier()

    int bm = 0, bn = 0;
    bool live = tile_id(0) < ntiles;
    if (live) { coords(tile_id(0), bm, bn); sources(bm, bn); stage(0, 0); if (nk > 1) { TTMI_VM_GUARD("v9"); stage(1, 1); } }
    for (int it = 0; live; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (nk > 1) TTMI_VM_WAIT("v9", V9_INFLIGHT);
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        V9_BAR();
        if (grp == 1) V9_BAR();
        int stg = 0;                                       // t % 3
        for (int t = 0; t < nk; ++t) {
            const char* base = smem + stg * STG9;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i][ks] = *reinterpret_cast<const Frag*>(base + aoff[ks] + i * 2048);
#pragma unroll
                for (int i = 0; i < 4; ++i) bfr[i][ks] = *reinterpret_cast<const Frag*>(base + boff[ks] + i * 2048);
            }
            if (t + 2 < nk) {
                TTMI_VM_GUARD("v9");                   // K-tile t + 1 (staged one tile ago) is older than this point
                stage(stg == 0 ? 2 : stg - 1, t + 2);      // (t + 2) % 3
                TTMI_VM_WAIT("v9", V9_INFLIGHT);
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            V9_BAR();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        if constexpr (F32IN) {
#pragma unroll
                            for (int s4 = 0; s4 < 4; ++s4)
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bfr[nt][ks][s4], af[mt][ks][s4], acc[mt][nt], 0, 0, 0);
                        } else {
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nt][ks], af[mt][ks], acc[mt][nt], 0, 0, 0);
                        }
                    }
            __builtin_amdgcn_s_setprio(0);
            V9_BAR();
            stg = stg == 2 ? 0 : stg + 1;
        }
        if (grp == 0) V9_BAR();

        const int cbm = bm, cbn = bn;
        live = tile_id(it + 1) < ntiles;
        if (live) { coords(tile_id(it + 1), bm, bn); sources(bm, bn); stage(0, 0); if (nk > 1) { TTMI_VM_GUARD("v9"); stage(1, 1); } }

        TC* C = reinterpret_cast<TC*>(p.C);
        char* img = smem + 2 * STG9 + wave * 4096;
        const bool vec = (p.ldc % 4 == 0) && ((reinterpret_cast<size_t>(p.C) & 15) == 0) &&
                         (!p.addend || (reinterpret_cast<size_t>(p.addend) & 15) == 0) &&
                         (!p.mask || (reinterpret_cast<size_t>(p.mask) & 7) == 0) && (!p.bias || (reinterpret_cast<size_t>(p.bias) & 15) == 0);
        const bool plain8 = vec && p.ldc % 8 == 0;       // bf16 rows in 16-byte pieces (any epilogue)
        const int wrow = lane & 15, wq = lane >> 4;
        // rolled slab loop with a wave-uniform switch over the accumulators (one copy of the epilogue code, no dynamic register indexing)
#define V9_SLAB(I) case I: _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) \
            *reinterpret_cast<f32x4*>(img + wrow * 256 + (((ni * 4 + wq) ^ wrow) << 4)) = acc[I][ni]; break;
        if constexpr (LEAN != 0) {
            // per-tile constants: the lane's columns (8 for bf16, 4 for f32 output), their bias, the column test, the row-0 address
            constexpr int NC = sizeof(TC) == 2 ? 8 : 4;
            const int ln0 = cbn + wc * 64 + (sizeof(TC) == 2 ? (lane & 7) * 8 : (lane & 15) * 4);
            const bool lfull = ln0 + NC - 1 < p.N;
            float lb[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j) lb[j] = 0.f;
            if ((LEAN == 1 || LEAN == 3) && p.bias && lfull) {
#pragma unroll
                for (int j = 0; j < NC; j += 4) {
                    const float4 b = *reinterpret_cast<const float4*>(p.bias + ln0 + j);
                    lb[j] = b.x; lb[j + 1] = b.y; lb[j + 2] = b.z; lb[j + 3] = b.w;
                }
            }
            const int lm0 = cbm + wr * 64 + (sizeof(TC) == 2 ? (lane >> 3) : (lane >> 4));
            TC* lrow0 = C + (long)lm0 * p.ldc + ln0;
            if constexpr (LEAN == 3 && sizeof(TC) == 4) {
                // residual epilogue (round 6): ALL 16 addend vectors of the lane are requested before the first slab goes through LDS - their latency is paid once, together with
                // the wait for the next tile's prefetch, instead of once per slab inside the rolled loop (these are one-round launches: FFN2 forward 16000 x 512 x 1024 took
                // 50 - 54 us against 21 - 29 us for the same shape without the addend).  The slab loop is unrolled here so that the 64 registers are indexed statically.
                f32x4 ad[4][4];
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        ad[mi][q] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (lfull && lm0 + mi * 16 + q * 4 < p.M)
                            ad[mi][q] = *reinterpret_cast<const f32x4*>(p.addend + (long)(lm0 + mi * 16 + q * 4) * p.ldc + ln0);
                    }
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) *reinterpret_cast<f32x4*>(img + wrow * 256 + (((ni * 4 + wq) ^ wrow) << 4)) = acc[mi][ni];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int r = q * 4 + (lane >> 4);
                        const f32x4 x = *reinterpret_cast<const f32x4*>(img + r * 256 + (((lane & 15) ^ r) << 4));
                        const int m = lm0 + mi * 16 + q * 4;
                        if (lfull) {
                            if (m < p.M) {
                                f32x4 o = {x[0] + lb[0], x[1] + lb[1], x[2] + lb[2], x[3] + lb[3]};
                                o += ad[mi][q];
                                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(lrow0) + (long)(mi * 16 + q * 4) * p.ldc) = o;
                            }
                        } else {
                            epi_store4<TC>(p, C, m, ln0, x, vec);
                        }
                    }
                }
            } else {
#pragma unroll 1
            for (int mi = 0; mi < 4; ++mi) {
                switch (mi) { V9_SLAB(0) V9_SLAB(1) V9_SLAB(2) V9_SLAB(3) }
                if constexpr (sizeof(TC) == 2) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int r = q * 8 + (lane >> 3);
                        const int c8 = lane & 7;
                        const f32x4 x0 = *reinterpret_cast<const f32x4*>(img + r * 256 + (((2 * c8) ^ r) << 4));
                        const f32x4 x1 = *reinterpret_cast<const f32x4*>(img + r * 256 + (((2 * c8 + 1) ^ r) << 4));
                        const int m = lm0 + mi * 16 + q * 8;
                        if (lfull) {
                            if (m < p.M) {
                                float v[8] = {x0[0] + lb[0], x0[1] + lb[1], x0[2] + lb[2], x0[3] + lb[3], x1[0] + lb[4], x1[1] + lb[5], x1[2] + lb[6], x1[3] + lb[7]};
                                if constexpr (LEAN == 2) {
                                    const u32x4 mk = *reinterpret_cast<const u32x4*>(p.mask + (long)m * p.ldc + ln0);
#pragma unroll
                                    for (int j = 0; j < 4; ++j) {
                                        const float lo = __uint_as_float(mk[j] << 16), hi = __uint_as_float(mk[j] & 0xffff0000u);
                                        v[2 * j] = p.mask_mode ? v[2 * j] * (1.f - lo * lo) : (lo > 0.f ? v[2 * j] * p.scale : 0.f);
                                        v[2 * j + 1] = p.mask_mode ? v[2 * j + 1] * (1.f - hi * hi) : (hi > 0.f ? v[2 * j + 1] * p.scale : 0.f);
                                    }
                                }
                                const u32x4 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
                                bf16_t* dst = reinterpret_cast<bf16_t*>(lrow0) + (long)(mi * 16 + q * 8) * p.ldc;
                                if (p.nt) __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(dst));
                                else *reinterpret_cast<u32x4*>(dst) = o;
                            }
                        } else {                       // the ragged last column group: the general element-wise path
                            epi_store4<bf16_t>(p, reinterpret_cast<bf16_t*>(C), m, ln0, x0, vec);
                            epi_store4<bf16_t>(p, reinterpret_cast<bf16_t*>(C), m, ln0 + 4, x1, vec);
                        }
                    }
                } else {
                    // f32 output: the lane keeps ONE 4-column chunk (c = lane & 15) for every row, read from slot c ^ r
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int r = q * 4 + (lane >> 4);
                        const f32x4 x = *reinterpret_cast<const f32x4*>(img + r * 256 + (((lane & 15) ^ r) << 4));
                        const int m = lm0 + mi * 16 + q * 4;
                        if (lfull) {
                            if (m < p.M) {
                                f32x4 o = {x[0] + lb[0], x[1] + lb[1], x[2] + lb[2], x[3] + lb[3]};
                                const long off = (long)(mi * 16 + q * 4) * p.ldc;
                                if constexpr (LEAN == 3) o += *reinterpret_cast<const f32x4*>(p.addend + (long)lm0 * p.ldc + ln0 + off);
                                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(lrow0) + off) = o;
                            }
                        } else {
                            epi_store4<TC>(p, C, m, ln0, x, vec);
                        }
                    }
                }
            }
            }
        } else {
#pragma unroll 1
        for (int mi = 0; mi < 4; ++mi) {
            switch (mi) { V9_SLAB(0) V9_SLAB(1) V9_SLAB(2) V9_SLAB(3) }
            if (sizeof(TC) == 2 && plain8) {
                // bf16 output: 8 columns per lane, one 16-byte store (8 rows x 128 B per instruction)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int r = q * 8 + (lane >> 3);
                    const int c8 = lane & 7;
                    const f32x4 x0 = *reinterpret_cast<const f32x4*>(img + r * 256 + (((2 * c8) ^ r) << 4));
                    const f32x4 x1 = *reinterpret_cast<const f32x4*>(img + r * 256 + (((2 * c8 + 1) ^ r) << 4));
                    if constexpr (sizeof(TC) == 2)
                        epi_store8_bf16(p, reinterpret_cast<bf16_t*>(C), cbm + wr * 64 + mi * 16 + r, cbn + wc * 64 + c8 * 8, x0, x1, vec);
                }
            } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = q * 4 + (lane >> 4);
                const int c = (lane & 15) ^ r;
                const f32x4 x = *reinterpret_cast<const f32x4*>(img + r * 256 + ((lane & 15) << 4));
                epi_store4<TC>(p, C, cbm + wr * 64 + mi * 16 + r, cbn + wc * 64 + c * 4, x, vec);
            }
            }
        }
    }
        }
#undef V9_SLAB
#undef V9_BAR
}

template <typename TC, int LEAN = 0>
__global__ __launch_bounds__(NTH8, 1) void gemm_nt_bf16_v9_kernel(const FP p) { gemm_nt_v9_body<TC, LEAN, false, false>(p); }
template <typename TC, int LEAN = 0>          // two-term weights
__global__ __launch_bounds__(NTH8, 1) void gemm_nt_bf16_v9_kw_kernel(const FP p) { gemm_nt_v9_body<TC, LEAN, false, true>(p); }
template <int LEAN = 0>                       // f32 operands, f32 output (exact-f32 path)
__global__ __launch_bounds__(NTH8, 1) void gemm_nt_f32_v9_kernel(const FP p) { gemm_nt_v9_body<float, LEAN, true, false>(p); }

// =====================================================================================================================
// Mid-sized exact-f32 NT products (greedy decoding: the label encoder on alive x history = 200 .. 2000 rows; d = 512 .. 1536 columns): too few
// 256 x 128 tiles for the persistent kernel to fill the chip, too many rows for the 32 x 32-tile kernel of csrc/gemm.hip, which re-reads every
// operand row from L2 once per 32 output columns (31 - 34 TFLOP/s).  64 x 64 tiles, 4 waves as 2 x 2 (32 x 32 each = 2 x 2 v_mfma_f32_16x16x4_f32
// tiles, operands swapped so a lane owns 4 consecutive columns of a row), three 16 KiB LDS stages of 32 floats of K filled by LDS-DMA with the
// 256 x 128 kernel's swizzle and fragment reads, ONE barrier per stage, counted vmcnt; 48 KiB of LDS and < 128 registers: three workgroups per CU.
// =====================================================================================================================
constexpr int TMID = 64, STGM = 2 * TMID * 128, LDSM = 3 * STGM;
constexpr int MID_INFLIGHT = STGM / (256 * 16);        // LDS-DMA instructions per wave (of 4) and stage
static_assert(MID_INFLIGHT == 4, "mid f32: the counted vmcnt waits assume 4 LDS-DMA instructions per stage");

// F32IN: f32 operands (stage = 32 floats of K, four v_mfma_f32_16x16x4_f32 per fragment pair) or bf16 operands (stage = 64 bf16 of K, one
// v_mfma_f32_16x16x32_bf16): the label encoder's products of the bf16 TRAINING step (1632 rows: 52 - 156 tiles of 128 x 128, one K-step at a time per
// lone workgroup, 45 - 67 us per launch) are the same mid-sized problem as the decoder's f32 ones.
template <bool F32IN, typename TC>
__device__ __forceinline__ void gemm_nt_mid_body(const FP& p) {
    constexpr int ES = F32IN ? 4 : 2, TKE = 128 / ES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave & 1, wc = wave >> 1;
    const int bn = blockIdx.x * TMID, bm = blockIdx.y * TMID;
    const int nk = p.K / TKE;                              // launcher: K % TKE == 0
    unsigned oA[2], oB[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (wave * 2 + j) * 8 + (lane >> 3);
        const unsigned slot = (unsigned)(((lane & 7) ^ ((r >> 1) & 7)) * 16);
        oA[j] = (unsigned)((long)min(r, p.M - 1 - bm) * p.lda * ES) + slot;
        oB[j] = (unsigned)((long)min(r, p.N - 1 - bn) * p.ldb * ES) + slot;
    }
    const char* baseA = reinterpret_cast<const char*>(p.A) + (long)bm * p.lda * ES;
    const char* baseB = reinterpret_cast<const char*>(p.B) + (long)bn * p.ldb * ES;
    auto stage = [&](int stg, int kt) {
        char* dst = smem + stg * STGM;
        const char* ba = baseA + (long)kt * 128;
        const char* bb = baseB + (long)kt * 128;
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(ba + oA[j], dst + (wave * 2 + j) * 1024);
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(bb + oB[j], dst + TMID * 128 + (wave * 2 + j) * 1024);
    };
    const int sw = (lane >> 1) & 7;
    int aoff[2], boff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int c = ((ks * 4 + (lane >> 4)) ^ sw) << 4;
        aoff[ks] = (wr * 32 + (lane & 15)) * 128 + c;
        boff[ks] = TMID * 128 + (wc * 32 + (lane & 15)) * 128 + c;
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    using Frag = typename std::conditional<F32IN, f32x4, bf16x8>::type;
    Frag af[2][2], bfr[2][2];

    stage(0, 0);
    if (nk > 1) { TTMI_VM_GUARD("mid"); stage(1, 1); TTMI_VM_WAIT("mid", MID_INFLIGHT); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int stg = 0;
    for (int t = 0; t < nk; ++t) {
        const char* base = smem + stg * STGM;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i][ks] = *reinterpret_cast<const Frag*>(base + aoff[ks] + i * 2048);
#pragma unroll
            for (int i = 0; i < 2; ++i) bfr[i][ks] = *reinterpret_cast<const Frag*>(base + boff[ks] + i * 2048);
        }
        if (t + 2 < nk) {
            TTMI_VM_GUARD("mid");                          // K-tile t + 1 (staged one tile ago) is older than this point
            stage(stg == 0 ? 2 : stg - 1, t + 2);          // (t + 2) % 3: the stage read in iteration t - 1 (retired before that iteration's barrier)
            TTMI_VM_WAIT("mid", MID_INFLIGHT);
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // K-tile t + 1 has landed for every wave; every wave's reads of K-tile t have retired
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    if constexpr (F32IN) {
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bfr[nt][ks][s4], af[mt][ks][s4], acc[mt][nt], 0, 0, 0);
                    } else {
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nt][ks], af[mt][ks], acc[mt][nt], 0, 0, 0);
                    }
                }
        __builtin_amdgcn_s_setprio(0);
        stg = stg == 2 ? 0 : stg + 1;
    }
    // epilogue: acc[mt][nt] of lane l = row wr*32 + mt*16 + (l & 15), columns wc*32 + nt*16 + 4*(l >> 4) .. +3: the general 4-column store
    // (bias, residual addend, ReLU, ReLU / tanh mask, dropout; 16-byte or 8-byte stores on aligned interiors)
    TC* C = reinterpret_cast<TC*>(p.C);
    const bool vec = (p.ldc % 4 == 0) && ((reinterpret_cast<size_t>(p.C) & 15) == 0) && (!p.addend || (reinterpret_cast<size_t>(p.addend) & 15) == 0) &&
                     (!p.mask || (reinterpret_cast<size_t>(p.mask) & 7) == 0) && (!p.bias || (reinterpret_cast<size_t>(p.bias) & 15) == 0);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
            epi_store4<TC>(p, C, bm + wr * 32 + mt * 16 + (lane & 15), bn + wc * 32 + nt * 16 + 4 * (lane >> 4), acc[mt][nt], vec);
}
__global__ __launch_bounds__(256, 3) void gemm_nt_f32_mid_kernel(const FP p_) {
    FP p = p_;
    p.drop = drop_live(p.drop);
    gemm_nt_mid_body<true, float>(p);
}
template <typename TC>
__global__ __launch_bounds__(256, 3) void gemm_nt_bf16_mid_kernel(const FP p_) {
    FP p = p_;
    p.drop = drop_live(p.drop);
    gemm_nt_mid_body<false, TC>(p);
}

// =====================================================================================================================
// v8 TN (wgrad with a huge reduction: C[M,N] += A[K,M]^T B[K,N], K >> M, N): the NT v8 schedule (256x256x64 tile, 8 waves as
// 2 x 4, two phases of 32 v_mfma_f32_16x16x32_bf16 per K-tile, waves 4-7 one barrier behind, counted vmcnt) with both operands
// reduction-major in LDS and every fragment read through ds_read_b64_tr_b16.
// LDS regions (16 KiB each, 2 buffers): A(h) = [64 k][128 m], m = block rows h*128..+127; B(h) = [64 k][128 n] likewise; 256-byte
// k-rows whose 16-byte chunks are XOR-swizzled by sw(k) = (k & 3) << 2 | ((k >> 3) & 1) << 1 through the glds SOURCE address, so
// the 4 k-rows x 2 k-octets a half-wave's transposed read touches fall on 8 distinct 32-byte bank groups.
// The wave tile is two 64-row strips (rows h*128 + wr*64 ..) x 64 columns: phase A needs only A(h0), phase B only A(h1).
// Work: grid = one workgroup per CU; the reduction is cut into 8*S ranges, XCD x (workgroup ids equal mod 8) owns ranges
// x*S..x*S+S-1 and its 32 workgroups walk the item list (range-major, then output tile), so the workgroups that run together
// on an XCD stream the SAME rows of A and B through that XCD's L2.  Results are added to C with f32 atomics.
// Fused bias gradient: colsum[m] += sum_k A[k][m] comes out of one extra MFMA per wave and K-tile against an all-ones fragment -
// the 4 column tiles x 4 waves that read the same A rows each take one of the 16 (strip, 16-row tile, k-half) pieces.
// =====================================================================================================================
constexpr int LDS8T = 2 * BUF8 + 8 * 4096;      // operand buffers + the epilogue's per-wave images

template <int CS>           // column sums of A: 0 = none, 1 = plain (all-ones fragment), 2 = weighted by p.csw[k] (exp-domain loss: bias gradient)
__global__ __launch_bounds__(NTH8, 1) void gemm_tn_bf16_v8_kernel(const FP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int S = p.splitk;                                // K-ranges per XCD
    const int xcd = blockIdx.x & 7, cu = blockIdx.x >> 3, ncu = gridDim.x >> 3;
    // p.gm = Q > 0: the M % 256 rows form one more tile row whose p.tiles_n tiles are cut into Q pieces along the XCD's whole K share and
    // appended to the item list, one piece per workgroup (tiles_n * Q = workgroups per XCD): every CU gets the same extra 1/Q tile
    const int nmain = S * ntiles;
    const int nitems = nmain + (p.gm > 0 ? p.tiles_n * p.gm : 0);
    const int nkt = (p.K + TK - 1) / TK;                   // K-tiles in all
    const int per = p.ksteps;                              // K-tiles per range

    // staging: unit u = 2*wave + j covers k-rows 4u + (lane >> 4) of a region, 16-byte slot lane & 15 <- source chunk slot ^ sw(k)
    int kst[2], cst[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        kst[j] = (wave * 2 + j) * 4 + (lane >> 4);
        cst[j] = (lane & 15) ^ (((kst[j] & 3) << 2) | (((kst[j] >> 3) & 1) << 1));
    }
    // sources = wave-uniform (item, K-tile) base + 32-bit per-lane byte offsets (saddr form of global_load_lds); K % 64 == 0, no tail
    unsigned oA[2][2], oB[2][2];
    const char* baseA = nullptr;
    const char* baseB = nullptr;
    long kbeg = 0;                                         // first reduction row of the current item
    int nk = 0;
    auto sources = [&](int bm, int bn, long k0) {
        baseA = reinterpret_cast<const char*>(p.A + k0 * p.lda);
        baseB = reinterpret_cast<const char*>(p.B + k0 * p.ldb);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const long ca = min((long)bm + h * 128 + cst[j] * 8, p.lda - 8), cb = min((long)bn + h * 128 + cst[j] * 8, p.ldb - 8);
                oA[h][j] = (unsigned)((kst[j] * p.lda + ca) * 2);
                oB[h][j] = (unsigned)((kst[j] * p.ldb + cb) * 2);
            }
    };
    auto stage = [&](int kind, int buf, int kt) {          // kind: 0 = A h0, 1 = A h1, 2 = B h0, 3 = B h1
        char* dst = smem + buf * BUF8 + kind * HT8 + wave * 2048;
        const char* base = kind < 2 ? baseA + (long)kt * TK * p.lda * 2 : baseB + (long)kt * TK * p.ldb * 2;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned o = kind == 0 ? oA[0][j] : kind == 1 ? oA[1][j] : kind == 2 ? oB[0][j] : oB[1][j];
            glds16(base + o, dst + j * 1024);
        }
    };
    auto prologue = [&]() {
        stage(0, 0, 0); stage(2, 0, 0); stage(3, 0, 0); stage(1, 0, 0);
        // K-tile 0 must have landed at the loop's first wait; tile 1's three half-tiles may fly (one K-tile only: that wait is a full drain)
        if (nk > 1) { TTMI_VM_GUARD("v8"); stage(0, 1, 1); stage(2, 1, 1); stage(3, 1, 1); }
    };
    // item i of this XCD -> (range, tile); tiles in GROUP_M-grouped order so that co-resident tiles share A / B panels
    auto item = [&](int i, int& bm, int& bn, int& tn) {
        if (i >= nmain) {                                  // a piece of a strip tile
            const int si = i - nmain;
            tn = si % p.tiles_n;
            const int piece = si / p.tiles_n;
            bm = p.tiles_m * T8;
            bn = tn * T8;
            const long share = (long)S * per, pp = (share + p.gm - 1) / p.gm;       // K-tiles of this XCD, per piece
            const long k0t = (long)xcd * share + piece * pp;
            kbeg = k0t * TK;
            nk = (int)max(0L, min(min(pp, share - piece * pp), (long)nkt - k0t));
            return;
        }
        const int s = i / ntiles, t = i % ntiles;
        const int per_group = GROUP_M * p.tiles_n;
        const int group = t / per_group, in = t % per_group;
        const int first = group * GROUP_M;
        const int gsz = min(p.tiles_m - first, GROUP_M);
        bm = (first + in % gsz) * T8;
        tn = in / gsz;
        bn = tn * T8;
        const long r = (long)xcd * S + s;
        kbeg = r * per * TK;
        nk = (int)max(0L, min((long)per, (long)nkt - r * per));
    };

    // transposed fragment reads: group g = lane >> 4 (the k-octet), lane 4q + pp of the group addresses k-row q, columns 4pp..4pp+3
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int swz = (q << 2) | ((g & 1) << 1);
    int aoff[4], boff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ca = wr * 64 + i * 16 + 4 * pp, cb = (wc & 1) * 64 + i * 16 + 4 * pp;
        aoff[i] = (g * 8 + q) * 256 + (((ca >> 3) ^ swz) << 4) + (ca & 7) * 2;
        boff[i] = (2 + (wc >> 1)) * HT8 + (g * 8 + q) * 256 + (((cb >> 3) ^ swz) << 4) + (cb & 7) * 2;
    }
    f32x4 acc[8][4];
    f32x4 cs = {0.f, 0.f, 0.f, 0.f};
    bf16x8 af[4][2], bfr[4][2];
    // transposed reads as asm (see LDS_TR): the counted vmcnt + barrier at the end of a phase cover the tiles, V8_LGKM0 these reads
#define TN8_TR(dst, addr, off) LDS_TR(dst, addr, off)
    const unsigned lds0 = (unsigned)(size_t)smem;          // LDS byte offset = low half of the flat shared address
    auto read_a = [&](const char* base, int h) {
        const unsigned b0 = lds0 + (unsigned)(base - smem) + h * HT8;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const unsigned a = b0 + aoff[mt];
                bf16x4 lo, hi;
                if (ks == 0) { TN8_TR(lo, a, 0); TN8_TR(hi, a, 1024); }
                else { TN8_TR(lo, a, 8192); TN8_TR(hi, a, 9216); }
                af[mt][ks] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
    };
    auto read_b = [&](const char* base) {
        const unsigned b0 = lds0 + (unsigned)(base - smem);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const unsigned a = b0 + boff[nt];
                bf16x4 lo, hi;
                if (ks == 0) { TN8_TR(lo, a, 0); TN8_TR(hi, a, 1024); }
                else { TN8_TR(lo, a, 8192); TN8_TR(hi, a, 9216); }
                bfr[nt][ks] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
    };
    auto mma = [&](int mh) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mh * 4 + mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nt][ks], af[mt][ks], acc[mh * 4 + mt][nt], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.0f;
    long kbeg_cur = 0;                                     // first reduction row of the item being computed
    int cs_unit = -1;                                      // (mh, mt, ks) piece of the column sums this wave owns for the current item
    // weighted column sums (CS == 2): colsum[m] += sum_k csw[k] A[k][m] - the all-ones fragment becomes the 8 weights of this lane's k-octet
    // (k = K-tile base + 32 ks + 8 (lane >> 4) + 0..7, the k order of the transposed A fragments).  The 64 weights of K-tile t travel like an
    // operand tile: ONE LDS-DMA instruction of wave 0 (32 lanes x 4 bytes) into slot t & 1 at the head of its epilogue image - free during
    // the K loop - issued with the A(h1) stage of the same tile, i.e. older than the six loads the counted vmcnt leaves in flight, so the wait
    // and barrier that publish that stage publish the weights too.  (A vector load into registers would sit in the in-order vmcnt queue
    // in front of the staged tiles: +35 %; through the scalar cache every wave's lgkmcnt(0) waited out a miss per K-tile: +12 %.)
    bf16x8 wcur;
    auto wload = [&](int t) {
        if constexpr (CS == 2) {
            if (wave == 0 && lane < 32) glds4(p.csw + kbeg_cur + (long)t * TK + lane * 2, smem + 2 * BUF8 + (t & 1) * 128);
        }
    };
    auto fetch_w = [&](int t, int mh) {                   // with the phase's fragment reads, behind the same lgkmcnt(0)
        if constexpr (CS == 2) {
            if (cs_unit >= 0 && (cs_unit >> 3) == mh) {
                const unsigned a = lds0 + 2 * BUF8 + (t & 1) * 128 + (cs_unit & 1) * 64 + (lane >> 4) * 16;
                asm volatile("ds_read_b128 %0, %1" : "=v"(wcur) : "v"(a) : "memory");
            }
        }
    };
    auto colsum_mma = [&](int mh) {
        if (!CS) return;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                if (cs_unit == mh * 8 + mt * 2 + ks) {
                    if constexpr (CS == 2) cs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wcur, af[mt][ks], cs, 0, 0, 0);
                    else cs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[mt][ks], cs, 0, 0, 0);
                }
    };
#define V8_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define V8_BAR() __builtin_amdgcn_s_barrier()

    int bm = 0, bn = 0, tn = 0;
    int it = cu;
    bool live = it < nitems;
    if (live) { item(it, bm, bn, tn); sources(bm, bn, kbeg); prologue(); }
    while (live) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        cs = f32x4{0.f, 0.f, 0.f, 0.f};
        cs_unit = CS && tn < 4 ? __builtin_amdgcn_readfirstlane(tn * 4 + wc) : -1;     // column tiles 0..3 share the 16 pieces
        const int cnk = nk;
        if constexpr (CS == 2) { kbeg_cur = kbeg; if (cnk > 0) wload(0); }      // the image area is free again: the previous item's epilogue is done
        if (cnk > 1 && CS != 2) TTMI_VM_WAIT("v8", V8_INFLIGHT);
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // (CS == 2: tile 0's weights are the newest load - once per item)
        V8_BAR();
        if (wr == 1) V8_BAR();
        for (int t = 0; t < cnk; ++t) {
            const int d = t & 1;
            const char* base = smem + d * BUF8;
            const bool more = t + 2 < cnk;
            fetch_w(t, 0);
            read_a(base, 0);
            read_b(base);
            if (t + 1 < cnk) { wload(t + 1); stage(1, d ^ 1, t + 1); }
            V8_LGKM0();
            V8_BAR();
            mma(0);
            colsum_mma(0);
            V8_BAR();
            fetch_w(t, 1);
            read_a(base, 1);
            if (more) {
                TTMI_VM_GUARD("v8");                // everything of K-tile t + 1 (A(h1) issued in phase A, the rest one tile ago) is older than this point
                stage(0, d, t + 2); stage(2, d, t + 2); stage(3, d, t + 2);
                TTMI_VM_WAIT("v8", V8_INFLIGHT);
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            V8_LGKM0();
            V8_BAR();
            mma(1);
            colsum_mma(1);
            V8_BAR();
        }
        if (wr == 0) V8_BAR();

        const int cbm = bm, cbn = bn, cunit = cs_unit;
        it += ncu;
        live = it < nitems;
        if (live) { item(it, bm, bn, tn); sources(bm, bn, kbeg); prologue(); }

        if (cnk > 0) {
            // memory-side atomics want whole contiguous segments per wave-instruction: 16 rows at a time through a private LDS image
            // (behind the operand buffers), one 256-byte row per instruction (see TN v9)
            float* C = reinterpret_cast<float*>(p.C);
            float* img = reinterpret_cast<float*>(smem + 2 * BUF8 + wave * 4096);
            const int n = cbn + wc * 64 + lane;
#define TN8_SLAB(I) case I: _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) \
            *reinterpret_cast<f32x4*>(img + (lane & 15) * 64 + ((ni * 4 + (lane >> 4)) ^ (lane & 15)) * 4) = acc[I][ni]; break;
#pragma unroll 1
            for (int mi = 0; mi < 8; ++mi) {
                switch (mi) { TN8_SLAB(0) TN8_SLAB(1) TN8_SLAB(2) TN8_SLAB(3) TN8_SLAB(4) TN8_SLAB(5) TN8_SLAB(6) TN8_SLAB(7) }
                const int m0 = cbm + (mi >> 2) * 128 + wr * 64 + (mi & 3) * 16;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = img[r * 64 + (((lane >> 2) ^ r) << 2) + (lane & 3)];
                    if (m0 + r < p.M && n < p.N) atomicAdd(C + (long)(m0 + r) * p.ldc + n, v);
                }
            }
#undef TN8_SLAB
            if (CS && cunit >= 0 && lane < 16) {
                const int m = cbm + (cunit >> 3) * 128 + wr * 64 + ((cunit >> 1) & 3) * 16 + lane;
                if (m < p.M) atomicAdd(p.colsum + m, cs[0]);
            }
        }
    }
#undef V8_LGKM0
#undef V8_BAR
}

// =====================================================================================================================
// v9 TN: the encoder wgrads (C[M,N] += A[K,M]^T B[K,N], M, N = 512..2048, K = B*T = 16000): TN v8's operand handling (both
// operands reduction-major in LDS, every fragment by ds_read_b64_tr_b16, same sw(k) swizzle) on v9's schedule (256 x 128 tile,
// 8 waves as 4 x 2, wave tile 64 x 64, three 48 KiB stages, one phase per K-tile, waves 4-7 one barrier behind, counted
// vmcnt(6)).  A stage = A [64 k][256 m] (512-byte k-rows, 32 staging units of 2 k-rows) | B [64 k][128 n] (256-byte k-rows, 16
// units of 4 k-rows).  Work = TN v8's item list: the reduction is cut into 8*S ranges, XCD x owns S of them, its workgroups
// walk (range, tile) items; f32 atomics into C; optional column sums of A from one extra all-ones MFMA per wave and K-tile
// (needs tiles_n >= 4: column tiles 0..3 x 2 wave columns that read the same A rows share its 8 (16-row tile, k-half) pieces).
// The 128x128 kernel this replaces spent 150 of its 221 us per layer in the main loops (4 workgroups per CU, single-buffered).
// =====================================================================================================================
template <bool CS>
__global__ __launch_bounds__(NTH8, 1) void gemm_tn_bf16_v9_kernel(const FP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)smem;          // LDS byte offset = low half of the flat shared address (asm reads)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave & 3, wc = wave >> 2;               // waves w and w+4 (SIMD partners) differ in the column half
    const int grp = wave >> 2;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int S = p.splitk;
    const int xcd = blockIdx.x & 7, cu = blockIdx.x >> 3, ncu = gridDim.x >> 3;
    const int nitems = S * ntiles;
    const int nkt = (p.K + TK - 1) / TK;
    const int per = p.ksteps;

    // staging: A unit u = 4*wave + j: k-rows 2u + (lane >> 5), 16-byte slot lane & 31; B unit u = 2*wave + j: k-rows 4u + (lane >> 4), slot lane & 15
    unsigned oA[4], oB[2];
    const char* baseA = nullptr;
    const char* baseB = nullptr;
    long kbeg = 0;
    int nk = 0;
    auto sources = [&](int bm, int bn, long k0) {
        baseA = reinterpret_cast<const char*>(p.A + k0 * p.lda);
        baseB = reinterpret_cast<const char*>(p.B + k0 * p.ldb);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = (wave * 4 + j) * 2 + (lane >> 5);
            const int c = (lane & 31) ^ (((k & 3) << 2) | (((k >> 3) & 1) << 1));
            oA[j] = (unsigned)((k * p.lda + min((long)bm + c * 8, p.lda - 8)) * 2);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = (wave * 2 + j) * 4 + (lane >> 4);
            const int c = (lane & 15) ^ (((k & 3) << 2) | (((k >> 3) & 1) << 1));
            oB[j] = (unsigned)((k * p.ldb + min((long)bn + c * 8, p.ldb - 8)) * 2);
        }
    };
    auto stage = [&](int stg, int kt) {
        char* dst = smem + stg * STG9;
        const char* ba = baseA + (long)kt * TK * p.lda * 2;
        const char* bb = baseB + (long)kt * TK * p.ldb * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(ba + oA[j], dst + (wave * 4 + j) * 1024);
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(bb + oB[j], dst + T9M * 128 + (wave * 2 + j) * 1024);
    };
    auto item = [&](int i, int& bm, int& bn, int& tn) {
        const int s = i / ntiles, t = i % ntiles;
        const int per_group = GROUP_M * p.tiles_n;
        const int group = t / per_group, in = t % per_group;
        const int first = group * GROUP_M;
        const int gsz = min(p.tiles_m - first, GROUP_M);
        bm = (first + in % gsz) * T9M;
        tn = in / gsz;
        bn = tn * T9N;
        const long r = (long)xcd * S + s;
        kbeg = r * per * TK;
        nk = (int)max(0L, min((long)per, (long)nkt - r * per));
    };

    // transposed fragment reads: group g = lane >> 4 (the k-octet), lane 4q + pp of the group addresses k-row q, columns 4pp..4pp+3
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int swz = (q << 2) | ((g & 1) << 1);
    int aoff[4], boff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ca = wr * 64 + i * 16 + 4 * pp, cb = wc * 64 + i * 16 + 4 * pp;
        aoff[i] = (g * 8 + q) * 512 + (((ca >> 3) ^ swz) << 4) + (ca & 7) * 2;
        boff[i] = T9M * 128 + (g * 8 + q) * 256 + (((cb >> 3) ^ swz) << 4) + (cb & 7) * 2;
    }
    f32x4 acc[4][4];
    f32x4 cs = {0.f, 0.f, 0.f, 0.f};
    bf16x8 af[4][2], bfr[4][2];
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.0f;
#define V9_BAR() __builtin_amdgcn_s_barrier()

    int bm = 0, bn = 0, tn = 0;
    int it = cu;
    bool live = it < nitems;
    if (live) { item(it, bm, bn, tn); sources(bm, bn, kbeg); stage(0, 0); if (nk > 1) { TTMI_VM_GUARD("v9"); stage(1, 1); } }
    while (live) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        cs = f32x4{0.f, 0.f, 0.f, 0.f};
        const int cs_unit = CS && tn < 4 ? __builtin_amdgcn_readfirstlane(tn * 2 + wc) : -1;     // (mt, ks) = (unit >> 1, unit & 1); column tiles 0..3 share the 8 pieces
        const int cnk = nk;
        if (cnk > 1) TTMI_VM_WAIT("v9", V9_INFLIGHT);
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        V9_BAR();
        if (grp == 1) V9_BAR();
        int stg = 0;
        for (int t = 0; t < cnk; ++t) {
            const unsigned base = lds0 + stg * STG9;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned a = base + aoff[i];
                    bf16x4 lo, hi;
                    if (ks == 0) { LDS_TR(lo, a, 0); LDS_TR(hi, a, 4 * 512); }
                    else { LDS_TR(lo, a, 32 * 512); LDS_TR(hi, a, 36 * 512); }
                    af[i][ks] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned bq = base + boff[i];
                    bf16x4 lo, hi;
                    if (ks == 0) { LDS_TR(lo, bq, 0); LDS_TR(hi, bq, 4 * 256); }
                    else { LDS_TR(lo, bq, 32 * 256); LDS_TR(hi, bq, 36 * 256); }
                    bfr[i][ks] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            }
            if (t + 2 < cnk) {
                TTMI_VM_GUARD("v9");                   // K-tile t + 1 (staged one tile ago) is older than this point
                stage(stg == 0 ? 2 : stg - 1, t + 2);
                TTMI_VM_WAIT("v9", V9_INFLIGHT);
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            V9_BAR();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nt][ks], af[mt][ks], acc[mt][nt], 0, 0, 0);
            if (CS) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
                        if (cs_unit == mt * 2 + ks) cs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[mt][ks], cs, 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
            V9_BAR();
            stg = stg == 2 ? 0 : stg + 1;
        }
        if (grp == 0) V9_BAR();

        const int cbm = bm, cbn = bn, cunit = cs_unit;
        it += ncu;
        live = it < nitems;
        if (live) { item(it, bm, bn, tn); sources(bm, bn, kbeg); stage(0, 0); if (nk > 1) { TTMI_VM_GUARD("v9"); stage(1, 1); } }

        if (cnk > 0) {
            // atomics run at the memory side and want whole contiguous segments per wave-instruction: the accumulators (a lane holds 4
            // columns of one row) go 16 rows at a time through a private LDS image in stage 2 (free until the next loop's K-tile 2) and
            // leave as one 256-byte row per instruction - straight from the registers the same adds were 4-byte pieces in 16 rows
            float* C = reinterpret_cast<float*>(p.C);
            float* img = reinterpret_cast<float*>(smem + 2 * STG9 + wave * 4096);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    *reinterpret_cast<f32x4*>(img + (lane & 15) * 64 + ((ni * 4 + (lane >> 4)) ^ (lane & 15)) * 4) = acc[mi][ni];
                const int n = cbn + wc * 64 + lane;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = cbm + wr * 64 + mi * 16 + r;
                    const float v = img[r * 64 + (((lane >> 2) ^ r) << 2) + (lane & 3)];
                    if (m < p.M && n < p.N) atomicAdd(C + (long)m * p.ldc + n, v);
                }
            }
            if (CS && cunit >= 0 && lane < 16) {
                const int m = cbm + wr * 64 + (cunit >> 1) * 16 + lane;
                if (m < p.M) atomicAdd(p.colsum + m, cs[0]);
            }
        }
    }
#undef V9_BAR
}

// =====================================================================================================================
// Grouped TN: the weight-gradient GEMMs of several encoder layers in ONE launch (C_q[M_q,N_q] += A_q[K_q,M_q]^T B_q[K_q,N_q], q < 16).
// A layer's four wgrads are 64 tiles of 256 x 128 - a quarter of the chip - which is why they used to be cut along K into ranges added
// up with f32 atomics (a third of each launch, and a result that depends on the order the atomics land in).  Four layers together are
// 256 tiles: every workgroup takes ONE tile over its whole reduction (K = B*T: 250 K-tiles of TN v9's schedule, no split, no atomics),
// adds it to C with plain read-modify-write (one writer per element: bit-identical from run to run and across ranks), and the tiles of
// one problem sit on one XCD so that its L2 serves each operand panel to all the tiles that share it.
// Optional column sums of A_q (the bias gradient that belongs to the wgrad): the all-ones MFMA of TN v9, dealt out so that every row
// has ONE owner - 16-row tile mt of wave row wr goes to the wave with tn * 2 + wc == mt (column tiles 0 and 1) - and added with plain
// stores as well.  Needs N_q / 128 >= 2.
// =====================================================================================================================
struct TnGroupProb {
    const bf16_t* A; const bf16_t* B; float* C; float* colsum;
    int M, N, K, tiles_m;
    long lda, ldb, ldc;
    int tile0;                      // first tile of this problem in the launch's tile list
};
struct TnGroup {
    TnGroupProb pr[16];
    int n, total;
    int pieces;                     // != 0: an XCD whose tile list is a few tiles longer than a whole number of rounds cuts the surplus tiles along K, one piece per
                                    // workgroup, added atomically (data-parallel backward on 256 - r CUs: 256 tiles would otherwise take two rounds)
};

__global__ __launch_bounds__(NTH8, 1) void gemm_tn_bf16_group_kernel(const TnGroup g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave & 3, wc = wave >> 2;
    const int grp = wave >> 2;
    const int xcd = blockIdx.x & 7, cu = blockIdx.x >> 3, ncu = gridDim.x >> 3;
    const int per_x = (g.total + 7) / 8;                   // XCD x owns tiles [x * per_x, (x + 1) * per_x): consecutive tiles = one problem's
    const int lo_x = min(xcd * per_x, g.total), hi = min((xcd + 1) * per_x, g.total);
    // items of this XCD: its tiles, whole - unless (g.pieces) the last round would be a small one: then its s tiles are cut into P = ncu / s pieces along K each
    // (every workgroup ends with one piece of 1 / P tile instead of s workgroups running a whole second tile)
    const int n_x = hi - lo_x, n_rounds = n_x / ncu, n_sur = n_x - n_rounds * ncu;
    const int P = (g.pieces && n_sur > 0 && 2 * n_sur <= ncu) ? ncu / n_sur : 0;
    const int n_whole = P ? n_rounds * ncu : n_x;
    const int n_items = n_whole + (P ? n_sur * P : 0);

    unsigned oA[4], oB[2];
    const char* baseA = nullptr;
    const char* baseB = nullptr;
    long lda = 0, ldb = 0;
    int nk = 0, q = 0;
    bool piece = false;                                    // the current item is a K-piece of a tile (added atomically)
    auto item = [&](int v, int& bm, int& bn, int& tn) {    // v: index into this XCD's item list
        piece = v >= n_whole;
        const int it = lo_x + (piece ? n_whole + (v - n_whole) / P : v);
        q = 0;
        while (q + 1 < g.n && it >= g.pr[q + 1].tile0) ++q;
        const TnGroupProb& pr = g.pr[q];
        const int t = it - pr.tile0;
        bm = (t % pr.tiles_m) * T9M;
        tn = t / pr.tiles_m;
        bn = tn * T9N;
        lda = pr.lda; ldb = pr.ldb;
        nk = pr.K / TK;
        baseA = reinterpret_cast<const char*>(pr.A);
        baseB = reinterpret_cast<const char*>(pr.B);
        if (piece) {                                       // K-tiles [k0, k1) of the tile (host: every problem has >= 2 P K-tiles when pieces are on)
            const int part = (v - n_whole) % P;
            const long k0 = (long)part * nk / P, k1 = (long)(part + 1) * nk / P;
            baseA += k0 * TK * lda * 2;
            baseB += k0 * TK * ldb * 2;
            nk = (int)(k1 - k0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = (wave * 4 + j) * 2 + (lane >> 5);
            const int c = (lane & 31) ^ (((k & 3) << 2) | (((k >> 3) & 1) << 1));
            oA[j] = (unsigned)((k * lda + bm + c * 8) * 2);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = (wave * 2 + j) * 4 + (lane >> 4);
            const int c = (lane & 15) ^ (((k & 3) << 2) | (((k >> 3) & 1) << 1));
            oB[j] = (unsigned)((k * ldb + bn + c * 8) * 2);
        }
    };
    auto stage = [&](int stg, int kt) {
        char* dst = smem + stg * STG9;
        const char* ba = baseA + (long)kt * TK * lda * 2;
        const char* bb = baseB + (long)kt * TK * ldb * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(ba + oA[j], dst + (wave * 4 + j) * 1024);
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(bb + oB[j], dst + T9M * 128 + (wave * 2 + j) * 1024);
    };

    const int gq = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
    const int swz = (qq << 2) | ((gq & 1) << 1);
    int aoff[4], boff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ca = wr * 64 + i * 16 + 4 * pp, cb = wc * 64 + i * 16 + 4 * pp;
        aoff[i] = (gq * 8 + qq) * 512 + (((ca >> 3) ^ swz) << 4) + (ca & 7) * 2;
        boff[i] = T9M * 128 + (gq * 8 + qq) * 256 + (((cb >> 3) ^ swz) << 4) + (cb & 7) * 2;
    }
    f32x4 acc[4][4];
    f32x4 cs = {0.f, 0.f, 0.f, 0.f};
    bf16x8 af[4][2], bfr[4][2];
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.0f;
#define V9_BAR() __builtin_amdgcn_s_barrier()

    int bm = 0, bn = 0, tn = 0;
    int it = cu;
    bool live = it < n_items;
    if (live) { item(it, bm, bn, tn); stage(0, 0); if (nk > 1) { TTMI_VM_GUARD("v9"); stage(1, 1); } }
    while (live) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        cs = f32x4{0.f, 0.f, 0.f, 0.f};
        const TnGroupProb cur = g.pr[q];
        const bool cpiece = piece;
        const int cs_mt = (cur.colsum && tn < 2) ? __builtin_amdgcn_readfirstlane(tn * 2 + wc) : -1;
        const int cnk = nk;
        if (cnk > 1) TTMI_VM_WAIT("v9", V9_INFLIGHT);
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        V9_BAR();
        if (grp == 1) V9_BAR();
        int stg = 0;
        for (int t = 0; t < cnk; ++t) {
            const unsigned base = lds0 + stg * STG9;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned a = base + aoff[i];
                    bf16x4 lo, hi2;
                    if (ks == 0) { LDS_TR(lo, a, 0); LDS_TR(hi2, a, 4 * 512); }
                    else { LDS_TR(lo, a, 32 * 512); LDS_TR(hi2, a, 36 * 512); }
                    af[i][ks] = __builtin_shufflevector(lo, hi2, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned bq = base + boff[i];
                    bf16x4 lo, hi2;
                    if (ks == 0) { LDS_TR(lo, bq, 0); LDS_TR(hi2, bq, 4 * 256); }
                    else { LDS_TR(lo, bq, 32 * 256); LDS_TR(hi2, bq, 36 * 256); }
                    bfr[i][ks] = __builtin_shufflevector(lo, hi2, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            }
            if (t + 2 < cnk) {
                TTMI_VM_GUARD("v9");                   // K-tile t + 1 (staged one tile ago) is older than this point
                stage(stg == 0 ? 2 : stg - 1, t + 2);
                TTMI_VM_WAIT("v9", V9_INFLIGHT);
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            V9_BAR();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nt][ks], af[mt][ks], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                if (cs_mt == mt) {
                    cs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[mt][0], cs, 0, 0, 0);
                    cs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[mt][1], cs, 0, 0, 0);
                }
            __builtin_amdgcn_s_setprio(0);
            V9_BAR();
            stg = stg == 2 ? 0 : stg + 1;
        }
        if (grp == 0) V9_BAR();

        const int cbm = bm, cbn = bn;
        it += ncu;
        live = it < n_items;
        if (live) { item(it, bm, bn, tn); stage(0, 0); if (nk > 1) { TTMI_VM_GUARD("v9"); stage(1, 1); } }

        // C += tile: 16 rows at a time through a private LDS image in stage 2 (free until the next tile's K-tile 2), one 256-byte row per
        // instruction; this workgroup is the only writer of these elements - unless the item was a K-piece: its P writers add atomically
        float* img = reinterpret_cast<float*>(smem + 2 * STG9 + wave * 4096);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                *reinterpret_cast<f32x4*>(img + (lane & 15) * 64 + ((ni * 4 + (lane >> 4)) ^ (lane & 15)) * 4) = acc[mi][ni];
            float* crow = cur.C + (long)(cbm + wr * 64 + mi * 16) * cur.ldc + cbn + wc * 64 + lane;
            if (cpiece) {
#pragma unroll
                for (int r = 0; r < 16; ++r) atomicAdd(crow + (long)r * cur.ldc, img[r * 64 + (((lane >> 2) ^ r) << 2) + (lane & 3)]);
            } else {
                float old[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) old[r] = crow[(long)r * cur.ldc];
#pragma unroll
                for (int r = 0; r < 16; ++r) crow[(long)r * cur.ldc] = old[r] + img[r * 64 + (((lane >> 2) ^ r) << 2) + (lane & 3)];
            }
        }
        if (cs_mt >= 0 && lane < 16) {
            float* c = cur.colsum + cbm + wr * 64 + cs_mt * 16 + lane;
            if (cpiece) atomicAdd(c, cs[0]);
            else *c = *c + cs[0];
        }
    }
#undef V9_BAR
}

template <typename K>
int enable_lds(K kernel, int bytes) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) { ttmi_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    return TTMI_OK;
}

// ttmi_set_option(1, v) - A/B measurements: 1 = 128x128 double-buffered, 3 = 1 + software-pipelined fragment reads,
// 4 (default) = single 32 KiB buffer, 4 workgroups per CU, one fragment set (stays under 128 VGPRs without spills; NT +15 % at
// the joint shapes; TN +8 % once each XCD owns a K-range) with the persistent 256x256 kernel (v8) for outputs of >= 1024 tiles,
// 5 = 4 without v8, 8 = v8 wherever it fits, 9 = the persistent 256x128 kernel (v9) wherever it fits.
// Measured and dropped (same box, joint projection M=816000 N=4334 K=1024, v4 = 720 TFLOP/s): 256x128 3-stage ring with
// counted vmcnt 621; persistent 256x256 with a 4-slice ring that never drains 668; the double-buffered 256x256 kernel without the wave stagger (v6, rounds 1 - 5);
// 4 waves x 128x128 wave tiles with AGPR accumulators (v10, rounds 1 - 5: 1.36 vs 1.55 PFLOP/s at 8192^3); round 6 (commit 70a15b4, profiles/r06_joint_gemm_variants.txt):
// accumulators stored straight from a store-shaped layout without the LDS transpose (+1 .. +3.5 % time), two free-running 4-wave workgroups per CU
// on 256x128 tiles (1.5 - 1.7 x the time), the 8 LDS-DMA instructions of a K-tile spread 4 + 4 over the two phases (+-1 %).
int g_gemm_fast_version = 4;
int g_f32_fast = 1;              // ttmi_set_option(17, v): 0 = f32 NT products stay on the kernels of csrc/gemm.hip (A/B); 2 = the 64x64-tile kernel wherever it can run; 3 = the persistent kernel only
int g_bf16_mid = 32;             // set_version(16 + n): bf16 NT problems the persistent kernels leave go to the 64 x 64-tile kernel from n of its tiles on (16 = never)
int g_nt_stores = 1;             // streaming stores for bf16 outputs >= 256 MB (set_version(14 / 15) = off / on, generation unchanged)
int g_num_cus = 0;
int g_tn_group_pieces = 1;        // ttmi_set_option(20, 0): grouped weight gradients never cut surplus tiles into K-pieces (A/B; see TnGroup::pieces)
int g_reserved_cus = 0;           // ttmi_set_option(6, n): process-wide default of the per-stream reservation below (measurement switch)
// CUs the mid-sized persistent GEMMs leave to concurrently running communication kernels: PER-STREAM state (ttmi_stream_reserve_cus), so a
// data-parallel backward pass on one stream does not change what another stream or thread launches.  Library-owned fork streams answer
// for the caller stream they serve (gemm_fast_alias_stream).
struct StreamRes { int dev; hipStream_t st, parent; int n; bool has_n; };
constexpr int MAX_RES = 64;
StreamRes g_res[MAX_RES];
int g_nres = 0;
std::mutex g_res_mu;

StreamRes* res_entry(int dev, hipStream_t st, bool create) {
    for (int i = 0; i < g_nres; ++i)
        if (g_res[i].dev == dev && g_res[i].st == st) return &g_res[i];
    if (!create || g_nres == MAX_RES) return nullptr;
    g_res[g_nres] = StreamRes{dev, st, st, 0, false};
    return &g_res[g_nres++];
}

int reserved_cus(hipStream_t st) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(g_res_mu);
    StreamRes* e = res_entry(dev, st, false);
    if (e && e->parent != st) e = res_entry(dev, e->parent, false);
    return e && e->has_n ? e->n : g_reserved_cus;
}
int g_tn_target_blocks = 512;     // split-K aims at this many workgroups for small outputs (ttmi_set_option(4, n))


}  // namespace

bool gemm_fast_nt_ok(const void* A, const void* B, const void* C, int M, int N, int K, long lda, long ldb) {
    return aligned16(A) && aligned16(B) && C && M > 0 && N > 0 && K >= 8 && K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 &&
           lda >= K && ldb >= K;
}

static void fill_batch(FP& p, const FastBatch& b) {
    p.nz2 = b.nz2; p.sA1 = b.sA1; p.sA2 = b.sA2; p.sB1 = b.sB1; p.sB2 = b.sB2; p.sC1 = b.sC1; p.sC2 = b.sC2; p.sV1 = b.sV1; p.sV2 = b.sV2;
}

static void ensure_num_cus() {
    if (g_num_cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
            g_num_cus = n / 8 * 8;        // the persistent tile walks assume a multiple of 8 (one share per XCD)
        if (g_num_cus <= 0) g_num_cus = 256;
    }
}
// persistent 256x256 NT kernel: needs several rounds of tiles per CU to amortise its pipeline fill and tail
static bool nt_v8_eligible(int M, int N, int K) {
    ensure_num_cus();
    const long t9 = (long)cdiv(M, T9M) * cdiv(N, T9N);
    return M >= 1024 && N >= 256 && K >= 128 && K % TK == 0 &&
           ((g_gemm_fast_version == 8) || (g_gemm_fast_version == 4 && t9 >= g_num_cus * 3 / 4));
}
// persistent 256x256 TN kernel (huge-reduction wgrad); S = K-ranges per XCD, strip_in = the M % 256 rows ride inside the kernel
static bool tn_v8_eligible(int M, int N, int K, bool colsum, int* S_out, bool* strip_in) {
    ensure_num_cus();
    const bool tn8 = K % TK == 0 && N >= 256 && N % 256 == 0 && (!colsum || N / 256 >= 4) &&
                     ((g_gemm_fast_version == 4 && K >= 32768 && M >= 1024) || (g_gemm_fast_version == 8 && K >= 2048 && M >= 256));
    if (!tn8) return false;
    const int ncu_x = g_num_cus / 8, tiles_n = N / T8, ntile = (M / T8) * tiles_n;
    int S = 1;                                          // fill every CU of the XCD, then balance the rounds
    while (S * ntile < ncu_x || ((S * ntile) % ncu_x != 0 && S * ntile < 8 * ncu_x)) ++S;
    // the strip rides along inside the kernel when its pieces deal out evenly (one per workgroup of the XCD) behind balanced main
    // items; otherwise it goes to the 128x128 kernel
    if (S_out) *S_out = S;
    if (strip_in) *strip_in = M % T8 != 0 && ncu_x % tiles_n == 0 && (S * ntile) % ncu_x == 0 && g_gemm_fast_version != 8;
    return true;
}
// can the fused joint + loss fast path run at this size?  (projection forward with the exp store, dgrad with the row factor,
// wgrad with the weighted column sums - all three on the persistent 256x256 kernels)
bool gemm_fast_joint_exp_ok(int M, int V, int J, long ldv, bool fwd_only) {
    if (M <= 0 || V <= 0 || J <= 0 || ldv < V) return false;
    if (!nt_v8_eligible(M, V, J)) return false;
    if (fwd_only) return true;                                      // (evaluation: only the exp-store projection runs)
    if (!nt_v8_eligible(M, J, (int)ldv)) return false;
    // the wgrad's reduction runs over the lattice rows PADDED to its 64-row K-tile (the caller's buffers have room for the pad rows and
    // joint_bwd_impl zero-fills them): any row count of a training-sized batch qualifies, not only multiples of 64
    return tn_v8_eligible(V, J, (M + TK - 1) / TK * TK, true, nullptr, nullptr);       // (a ragged V % 256 strip rides inside the kernel or takes the 128x128 kernel: both weight their column sums)
}

int gemm_nt_bf16(const bf16_t* A, const bf16_t* B, void* C, int c_dtype, const NtEpilogue& epi, int M, int N, int K, long lda,
                 long ldb, long ldc, hipStream_t st, const FastBatch& batch) {
    TTMI_REQUIRE(gemm_fast_nt_ok(A, B, C, M, N, K, lda, ldb), "gemm_nt_bf16: shape/alignment not supported (M=%d N=%d K=%d)", M,
                 N, K);
    FP p;
    p.A = A; p.B = B; p.C = C; p.bias = epi.bias; p.addend = epi.addend; p.mask = epi.mask; p.mask_mode = epi.mask_mode; p.nt = (g_nt_stores && c_dtype == 1 && (long)M * ldc * 2 >= (256L << 20)) ? 1 : 0; p.relu = epi.relu; p.scale = epi.scale; p.drop = epi.drop;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.tiles_m = cdiv(M, TM); p.tiles_n = cdiv(N, TN_); p.splitk = 1; p.ksteps = cdiv(K, TK); p.atomic = 0; p.gm = GROUP_M; p.colsum = nullptr;
    p.A2 = epi.A2; p.B2 = epi.B2; p.K2 = epi.K2; p.lda2 = epi.lda2; p.ldb2 = epi.ldb2; p.sB1b = epi.sB1b; p.sB2b = epi.sB2b; p.colsum_mid = epi.colsum_mid;
    p.rowsum = epi.rowsum; p.nparts = epi.nparts; p.exp_shift = epi.exp_shift; p.rowscale = epi.rowscale; p.csw = nullptr;
    const bool dual = epi.A2 != nullptr;
    // second weight term (epi.B_lo, same shape, pitch and batch strides as B): the persistent kernels run the K loop twice over the same A tiles
    // (kwrap), the 128x128 kernel takes (A, B_lo) as its second operand pair
    const bool two_term = epi.B_lo != nullptr;
    TTMI_REQUIRE(!two_term || (!dual && aligned16(epi.B_lo) && (g_gemm_fast_version == 4 || g_gemm_fast_version == 8 || g_gemm_fast_version == 9)),
                 "gemm_nt_bf16: a second weight term needs a 16-byte aligned B_lo, no second operand pair and the default kernel generations");
    const int k_lo = two_term ? (epi.K_lo ? epi.K_lo : K) : 0;      // columns of A that meet B_lo
    TTMI_REQUIRE(!two_term || (k_lo > 0 && k_lo <= K && (k_lo == K || k_lo % TK == 0)), "gemm_nt_bf16: K_lo = %d must be a multiple of %d inside K = %d", k_lo, TK, K);
    if (dual) TTMI_REQUIRE(epi.B2 && epi.K2 >= 8 && epi.K2 % 8 == 0 && aligned16(epi.A2) && aligned16(epi.B2) && epi.lda2 % 8 == 0 && epi.ldb2 % 8 == 0 &&
                           c_dtype == 1, "gemm_nt_bf16: bad second operand pair");
    fill_batch(p, batch);
    const int nbatch = batch.nz1 * batch.nz2;
    TTMI_REQUIRE(nbatch >= 1 && nbatch <= 65535, "gemm_nt_bf16: bad batch count %d", nbatch);
    // 256x256 tiles pay off when the K loop is long enough to amortise the un-overlapped epilogue of a one-workgroup-per-CU
    // kernel and there are enough tiles to fill the chip (joint dgrad: K = 4352 -> 954 vs 880 TFLOP/s; short-K forward: worse)
    ensure_num_cus();
    // persistent kernels: 256x256 tiles (v8) are ~1.8x the cost of 256x128 tiles (v9); take whichever needs less time for its
    // whole number of rounds over the CUs (joint: v8; encoder N = 512 / 1536: v9; N = 2048: v8), the 128x128 kernel for small outputs
    const long t9 = (long)cdiv(M, T9M) * cdiv(N, T9N), t8 = (long)cdiv(M, T8) * cdiv(N, T8);
    const bool pers = nbatch == 1 && M >= 1024 && N >= 128 && K >= 128 && K % TK == 0 && !dual;
    const int reserved = reserved_cus(st);
    // CUs the NT kernels may fill: all of them, or - while communication kernels hold `reserved` CUs - EXACTLY the rest (round 6: rounded down to a multiple of 8 a
    // reservation of 1 ... 8 CUs left 248, and the encoder's 252-tile backward GEMMs took two rounds instead of one: +1.3 ms per step for any reservation at all)
    const int cus_avail = reserved > 0 ? std::max(8, g_num_cus - reserved) : g_num_cus;
    const double cost9 = (double)cdiv(t9, cus_avail), cost8 = N >= 256 ? 1.8 * cdiv(t8, t8 < 1024 ? cus_avail : g_num_cus) : 1e30;
    // the exp-store / row-scale epilogues exist on the 256x256 kernel only: callers ask gemm_fast_joint_exp_ok() first
    const bool needs8 = epi.rowsum || epi.rowscale;
    const bool v8 = nbatch == 1 && !dual && nt_v8_eligible(M, N, K);
    TTMI_REQUIRE(!needs8 || (v8 && c_dtype == 1), "gemm_nt_bf16: the exp-store / row-scale epilogues exist on the persistent 256x256 kernel only (M=%d N=%d K=%d)", M, N, K);
    // (measured and dropped, round 6: under a CU reservation, handing a problem of barely more tiles than workgroups - the encoder's 252 on 224 - to the 64 x 64-tile
    // kernel instead of a second persistent round: +0.4 ms per step, profiles/r06_reserve_cus_single_gpu.txt)
    const bool v9 = !needs8 && pers && ((g_gemm_fast_version == 9) || (g_gemm_fast_version == 4 && t9 >= g_num_cus * 3 / 4 && cost9 <= cost8));
    if (v9) {
        if (two_term) { p.B2 = epi.B_lo; p.kwrap = K / TK; p.K = K + k_lo; }
        p.tiles_m = cdiv(M, T9M); p.tiles_n = cdiv(N, T9N);
        // a persistent grid larger than the CUs that are actually free runs its surplus workgroups AFTER the others (twice the time):
        // with gradient all-reduce kernels resident during backward, leave them room (multi-GPU runs set option 6)
        const int grid9 = reserved > 0 ? (int)std::min<long>(t9, cus_avail) : (int)((std::min<long>(t9, cus_avail) + 7) / 8 * 8);
        // lean epilogue instances (nothing tested per store): plain / bias-only for both output types, mask-only for bf16
        const bool base_ok = !p.addend && !p.relu && p.drop.p <= 0.f && aligned16(C) && (!p.bias || aligned16(p.bias));
        const bool lean1 = base_ok && !p.mask && ldc % (c_dtype == 0 ? 4 : 8) == 0;
        const bool lean2 = base_ok && p.mask && !p.bias && c_dtype == 1 && ldc % 8 == 0 && aligned16(p.mask);
#define V9_LAUNCH(...) do { if (int rc = enable_lds((gemm_nt_bf16_v9_kernel<__VA_ARGS__>), LDS9)) return rc; \
            hipLaunchKernelGGL((gemm_nt_bf16_v9_kernel<__VA_ARGS__>), dim3((unsigned)grid9), dim3(NTH8), LDS9, st, p); } while (0)
        if (p.kwrap) {                                     // two-term weights (option 13): the plain / bias-only f32 instance (o_net, CoreNet.3: the default since round 6), else the general epilogue
            if (c_dtype == 0 && lean1) {
                if (int rc = enable_lds((gemm_nt_bf16_v9_kw_kernel<float, 1>), LDS9)) return rc;
                hipLaunchKernelGGL((gemm_nt_bf16_v9_kw_kernel<float, 1>), dim3((unsigned)grid9), dim3(NTH8), LDS9, st, p);
            } else if (c_dtype == 0) {
                if (int rc = enable_lds((gemm_nt_bf16_v9_kw_kernel<float>), LDS9)) return rc;
                hipLaunchKernelGGL((gemm_nt_bf16_v9_kw_kernel<float>), dim3((unsigned)grid9), dim3(NTH8), LDS9, st, p);
            } else {
                if (int rc = enable_lds((gemm_nt_bf16_v9_kw_kernel<bf16_t>), LDS9)) return rc;
                hipLaunchKernelGGL((gemm_nt_bf16_v9_kw_kernel<bf16_t>), dim3((unsigned)grid9), dim3(NTH8), LDS9, st, p);
            }
        } else if (c_dtype == 0) {
            if (lean1) V9_LAUNCH(float, 1);
            else if (p.addend && !p.mask && !p.relu && p.drop.p <= 0.f && ldc % 4 == 0 && aligned16(C) && aligned16(p.addend) && (!p.bias || aligned16(p.bias))) V9_LAUNCH(float, 3);
            else V9_LAUNCH(float, 0);
        } else {
            if (lean1) V9_LAUNCH(bf16_t, 1);
            else if (lean2) V9_LAUNCH(bf16_t, 2);
            else V9_LAUNCH(bf16_t, 0);
        }
#undef V9_LAUNCH
        TTMI_LAUNCH_CHECK("gemm_nt_bf16_v9_kernel");
        return TTMI_OK;
    }
    // persistent 256x256 kernel: needs several rounds of tiles per CU to amortise its pipeline fill and tail
    if (v8) {
        if (two_term) { p.B2 = epi.B_lo; p.kwrap = K / TK; p.K = K + k_lo; }
        p.tiles_m = cdiv(M, T8); p.tiles_n = cdiv(N, T8);
        static const int walk_env = [] { const char* e = getenv("TTMI_TILE_WALK"); return e ? atoi(e) : 0; }();
        p.walk = walk_env;
        const long nwg8 = (long)p.tiles_m * p.tiles_n;
        const int cus8 = nwg8 < 1024 ? cus_avail : g_num_cus;   // encoder-sized problems only (see v9)
        const int grid8 = reserved > 0 && nwg8 < 1024 ? (int)std::min<long>(nwg8, cus8) : (int)((std::min<long>(nwg8, cus8) + 7) / 8 * 8);   // multiple of 8: one share of every round per XCD
        if (grid8 & 7) p.walk = 0;
if (p.kwrap) {
            TTMI_REQUIRE(!needs8, "gemm_nt_bf16: no second weight term with the exp-store / row-scale epilogues");
            if (c_dtype == 0) {
                if (int rc = enable_lds((gemm_nt_bf16_v8_kw_kernel<float>), LDS8)) return rc;
                hipLaunchKernelGGL((gemm_nt_bf16_v8_kw_kernel<float>), dim3((unsigned)grid8), dim3(NTH8), LDS8, st, p);
            } else {
                if (int rc = enable_lds((gemm_nt_bf16_v8_kw_kernel<bf16_t>), LDS8)) return rc;
                hipLaunchKernelGGL((gemm_nt_bf16_v8_kw_kernel<bf16_t>), dim3((unsigned)grid8), dim3(NTH8), LDS8, st, p);
            }
        } else if (c_dtype == 0) {
            if (int rc = enable_lds(gemm_nt_bf16_v8_kernel<float>, LDS8)) return rc;
            hipLaunchKernelGGL(gemm_nt_bf16_v8_kernel<float>, dim3((unsigned)grid8), dim3(NTH8), LDS8, st, p);
        } else if (p.rowsum) {
            TTMI_REQUIRE(!p.addend && !p.mask && !p.relu && p.drop.p <= 0.f && ldc % 8 == 0 && aligned16(C) && p.nparts >= 4 * p.tiles_n,
                         "gemm_nt_bf16: exp store needs a plain bf16 output with pitch %% 8 == 0 and nparts >= 4 * column tiles");
            if (int rc = enable_lds((gemm_nt_bf16_v8_kernel<bf16_t, 3>), LDS8)) return rc;
            hipLaunchKernelGGL((gemm_nt_bf16_v8_kernel<bf16_t, 3>), dim3((unsigned)grid8), dim3(NTH8), LDS8, st, p);
        } else if (!p.addend && !p.mask && !p.relu && p.drop.p <= 0.f && ldc % 8 == 0 && aligned16(C) && (!p.bias || aligned16(p.bias))) {
            if (int rc = enable_lds((gemm_nt_bf16_v8_kernel<bf16_t, 1>), LDS8)) return rc;
            hipLaunchKernelGGL((gemm_nt_bf16_v8_kernel<bf16_t, 1>), dim3((unsigned)grid8), dim3(NTH8), LDS8, st, p);
        } else if (!p.addend && !p.mask && p.relu && !p.rowscale && ldc % 8 == 0 && aligned16(C) && (!p.bias || aligned16(p.bias))) {
            if (int rc = enable_lds((gemm_nt_bf16_v8_kernel<bf16_t, 5>), LDS8)) return rc;
            hipLaunchKernelGGL((gemm_nt_bf16_v8_kernel<bf16_t, 5>), dim3((unsigned)grid8), dim3(NTH8), LDS8, st, p);
        } else if (p.rowscale) {
            TTMI_REQUIRE(!p.addend && p.mask && p.mask_mode == 1 && !p.bias && !p.relu && p.drop.p <= 0.f && ldc % 8 == 0 && aligned16(C) && aligned16(p.mask),
                         "gemm_nt_bf16: the row factor needs the tanh-mask bf16 epilogue");
            if (int rc = enable_lds((gemm_nt_bf16_v8_kernel<bf16_t, 4>), LDS8)) return rc;
            hipLaunchKernelGGL((gemm_nt_bf16_v8_kernel<bf16_t, 4>), dim3((unsigned)grid8), dim3(NTH8), LDS8, st, p);
        } else if (!p.addend && p.mask && !p.bias && !p.relu && p.drop.p <= 0.f && ldc % 8 == 0 && aligned16(C) && aligned16(p.mask)) {
            if (int rc = enable_lds((gemm_nt_bf16_v8_kernel<bf16_t, 2>), LDS8)) return rc;
            hipLaunchKernelGGL((gemm_nt_bf16_v8_kernel<bf16_t, 2>), dim3((unsigned)grid8), dim3(NTH8), LDS8, st, p);
        } else {
            if (int rc = enable_lds(gemm_nt_bf16_v8_kernel<bf16_t>, LDS8)) return rc;
            hipLaunchKernelGGL(gemm_nt_bf16_v8_kernel<bf16_t>, dim3((unsigned)grid8), dim3(NTH8), LDS8, st, p);
        }
        TTMI_LAUNCH_CHECK("gemm_nt_bf16_v8_kernel");
        return TTMI_OK;
    }
    // mid-sized problems (whatever the persistent kernels did not take: fewer than 1024 rows or too few of their tiles - the label encoder's 1632 rows):
    // 64 x 64 tiles through a 3-stage LDS-DMA pipeline instead of 128 x 128 tiles that walk K one barrier-separated step at a time
    if (g_bf16_mid && (g_gemm_fast_version == 4 || g_gemm_fast_version == 5) && nbatch == 1 && !dual && !two_term && K >= 64 && K % 64 == 0 && (long)cdiv(M, TMID) * cdiv(N, TMID) >= g_bf16_mid &&
        cdiv(M, TMID) <= 65535 && (long)63 * std::max(lda, ldb) * 2 + 128 < (1L << 32)) {
        p.tiles_m = cdiv(M, TMID); p.tiles_n = cdiv(N, TMID);
        if (c_dtype == 0) hipLaunchKernelGGL(gemm_nt_bf16_mid_kernel<float>, dim3((unsigned)p.tiles_n, (unsigned)p.tiles_m), dim3(256), LDSM, st, p);
        else hipLaunchKernelGGL(gemm_nt_bf16_mid_kernel<bf16_t>, dim3((unsigned)p.tiles_n, (unsigned)p.tiles_m), dim3(256), LDSM, st, p);
        TTMI_LAUNCH_CHECK("gemm_nt_bf16_mid_kernel");
        return TTMI_OK;
    }
    const long nwg = (long)p.tiles_m * p.tiles_n;
    TTMI_REQUIRE(nwg < (1L << 31), "gemm_nt_bf16: too many tiles");
    const int ver = g_gemm_fast_version;
#define NT_LAUNCH(TCT, NB, PP) hipLaunchKernelGGL((gemm_nt_bf16_kernel<TCT, NB, PP>), dim3((unsigned)nwg, nbatch), dim3(NTH), 2 * NB * TILE_B, st, p)
    if (two_term) {
        p.A2 = A; p.B2 = epi.B_lo; p.K2 = k_lo; p.lda2 = lda; p.ldb2 = ldb; p.sB1b = p.sB1; p.sB2b = p.sB2; p.colsum_mid = nullptr;
    }
    if (dual || two_term) {
        if (c_dtype == 1) hipLaunchKernelGGL((gemm_nt_bf16_kernel<bf16_t, 1, false, true>), dim3((unsigned)nwg, nbatch), dim3(NTH), 2 * TILE_B, st, p);
        else hipLaunchKernelGGL((gemm_nt_bf16_kernel<float, 1, false, true>), dim3((unsigned)nwg, nbatch), dim3(NTH), 2 * TILE_B, st, p);
    } else if (c_dtype == 0) {
        if (ver >= 4) NT_LAUNCH(float, 1, false);
        else if (ver == 3) NT_LAUNCH(float, 2, true);
        else NT_LAUNCH(float, 2, false);
    } else {
        if (ver >= 4) NT_LAUNCH(bf16_t, 1, false);
        else if (ver == 3) NT_LAUNCH(bf16_t, 2, true);
        else NT_LAUNCH(bf16_t, 2, false);
    }
#undef NT_LAUNCH
    TTMI_LAUNCH_CHECK("gemm_nt_bf16_kernel");
    return TTMI_OK;
}

// ---- mid-sized exact-f32 NT products: 64 x 64 tiles (gemm_nt_f32_mid_kernel); bias / accumulate / ReLU epilogue
bool gemm_nt_f32_mid_ok(const void* A, const void* B, const void* C, int M, int N, int K, long lda, long ldb, long ldc) {
    return g_f32_fast && aligned16(A) && aligned16(B) && C && M >= 1 && N >= 1 && K >= 32 && K % 32 == 0 && lda % 4 == 0 && ldb % 4 == 0 && lda >= K &&
           ldb >= K && ldc >= N && (long)63 * std::max(lda, ldb) * 4 + 128 < (1L << 32) && cdiv(M, TMID) <= 65535;
}
int gemm_nt_f32_mid(const float* A, const float* B, float* C, const float* bias, int accumulate, int relu, int M, int N, int K, long lda, long ldb,
                    long ldc, hipStream_t st) {
    TTMI_REQUIRE(gemm_nt_f32_mid_ok(A, B, C, M, N, K, lda, ldb, ldc), "gemm_nt_f32_mid: shape/alignment not supported (M=%d N=%d K=%d)", M, N, K);
    FP p;
    p.A = reinterpret_cast<const bf16_t*>(A); p.B = reinterpret_cast<const bf16_t*>(B); p.C = C; p.bias = bias; p.addend = accumulate ? C : nullptr;
    p.mask = nullptr; p.mask_mode = 0; p.nt = 0; p.relu = relu; p.scale = 1.f; p.drop = DropSpec();
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.tiles_m = cdiv(M, TMID); p.tiles_n = cdiv(N, TMID); p.splitk = 1; p.ksteps = K / 32; p.atomic = 0; p.gm = GROUP_M; p.colsum = nullptr;
    p.A2 = nullptr; p.B2 = nullptr; p.K2 = 0; p.lda2 = p.ldb2 = p.sB1b = p.sB2b = 0; p.colsum_mid = nullptr;
    p.rowsum = nullptr; p.nparts = 0; p.exp_shift = nullptr; p.rowscale = nullptr; p.csw = nullptr;
    fill_batch(p, FastBatch());
    hipLaunchKernelGGL(gemm_nt_f32_mid_kernel, dim3((unsigned)p.tiles_n, (unsigned)p.tiles_m), dim3(256), LDSM, st, p);
    TTMI_LAUNCH_CHECK("gemm_nt_f32_mid_kernel");
    return TTMI_OK;
}

// ---- exact-f32 NT products on the persistent 256x128 kernel (F32IN): C[M,N] (f32) = epi(A[M,K] . B[N,K]^T), f32 operands
bool gemm_nt_f32_ok(const void* A, const void* B, const void* C, int M, int N, int K, long lda, long ldb, long ldc) {
    return g_f32_fast && aligned16(A) && aligned16(B) && C && M >= 1024 && N >= 128 && K >= 64 && K % 32 == 0 && lda % 4 == 0 && ldb % 4 == 0 &&
           lda >= K && ldb >= K && ldc >= N && (long)255 * std::max(lda, ldb) * 4 + 128 < (1L << 32);
}
int gemm_nt_f32(const float* A, const float* B, float* C, const NtEpilogue& epi, int M, int N, int K, long lda, long ldb, long ldc, hipStream_t st) {
    TTMI_REQUIRE(gemm_nt_f32_ok(A, B, C, M, N, K, lda, ldb, ldc), "gemm_nt_f32: shape/alignment not supported (M=%d N=%d K=%d)", M, N, K);
    TTMI_REQUIRE(!epi.A2 && !epi.B_lo && !epi.rowsum && !epi.rowscale && !epi.mask, "gemm_nt_f32: bias / addend / ReLU / dropout epilogues only");
    FP p;
    p.A = reinterpret_cast<const bf16_t*>(A); p.B = reinterpret_cast<const bf16_t*>(B); p.C = C; p.bias = epi.bias; p.addend = epi.addend; p.mask = nullptr;
    p.mask_mode = 0; p.nt = 0; p.relu = epi.relu; p.scale = epi.scale; p.drop = epi.drop;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.tiles_m = cdiv(M, T9M); p.tiles_n = cdiv(N, T9N); p.splitk = 1; p.ksteps = K / 32; p.atomic = 0; p.gm = GROUP_M; p.colsum = nullptr;
    p.A2 = nullptr; p.B2 = nullptr; p.K2 = 0; p.lda2 = p.ldb2 = p.sB1b = p.sB2b = 0; p.colsum_mid = nullptr;
    p.rowsum = nullptr; p.nparts = 0; p.exp_shift = nullptr; p.rowscale = nullptr; p.csw = nullptr;
    fill_batch(p, FastBatch());
    ensure_num_cus();
    const long t9 = (long)p.tiles_m * p.tiles_n;
    const int cus9 = std::max(8, (g_num_cus - reserved_cus(st)) / 8 * 8);
    const int grid9 = (int)((std::min<long>(t9, cus9) + 7) / 8 * 8);
    const bool base_ok = !p.relu && p.drop.p <= 0.f && aligned16(C) && ldc % 4 == 0 && (!p.bias || aligned16(p.bias));
#define V9F_LAUNCH(L) do { if (int rc = enable_lds((gemm_nt_f32_v9_kernel<L>), LDS9)) return rc; \
            hipLaunchKernelGGL((gemm_nt_f32_v9_kernel<L>), dim3((unsigned)grid9), dim3(NTH8), LDS9, st, p); } while (0)
    if (base_ok && !p.addend) V9F_LAUNCH(1);
    else if (base_ok && aligned16(p.addend)) V9F_LAUNCH(3);
    else V9F_LAUNCH(0);
#undef V9F_LAUNCH
    TTMI_LAUNCH_CHECK("gemm_nt_f32_v9_kernel");
    return TTMI_OK;
}

bool gemm_fast_tn_ok(const void* A, const void* B, const void* C, int M, int N, int K, long lda, long ldb) {
    return aligned16(A) && aligned16(B) && C && M > 0 && N > 0 && K >= 1 && lda % 8 == 0 && ldb % 8 == 0 &&
           lda >= ((M + 7) & ~7) && ldb >= ((N + 7) & ~7);
}

// C (f32) += A^T B by atomics when splitk > 1 or accumulate != 0, else C = A^T B
int gemm_tn_bf16(const bf16_t* A, const bf16_t* B, float* C, int M, int N, int K, long lda, long ldb, long ldc, int accumulate,
                 hipStream_t st, float* colsum_a, const FastBatch& batch, const bf16_t* colsum_w) {
    TTMI_REQUIRE(gemm_fast_tn_ok(A, B, C, M, N, K, lda, ldb), "gemm_tn_bf16: shape/alignment not supported (M=%d N=%d K=%d)", M,
                 N, K);
    FP p;
    p.A = A; p.B = B; p.C = C; p.bias = nullptr; p.addend = nullptr; p.mask = nullptr; p.mask_mode = 0; p.nt = 0; p.relu = 0; p.scale = 1.f; p.drop = DropSpec();
    p.colsum = colsum_a;
    p.rowsum = nullptr; p.nparts = 0; p.exp_shift = nullptr; p.rowscale = nullptr; p.csw = colsum_w;
    TTMI_REQUIRE(!colsum_w || (colsum_a && aligned16(colsum_w)), "gemm_tn_bf16: weighted column sums need colsum_a and 16-byte aligned weights");
    fill_batch(p, batch);
    const int nbatch = batch.nz1 * batch.nz2;
    TTMI_REQUIRE(nbatch >= 1 && nbatch <= 65535, "gemm_tn_bf16: bad batch count %d", nbatch);
    if (g_num_cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
            g_num_cus = n / 8 * 8;
        if (g_num_cus <= 0) g_num_cus = 256;
    }
    // encoder-sized wgrad: persistent 256x128 kernel when its tiles fill the output exactly and the reduction is long enough
    // (weighted column sums exist on the 256x256 kernel only: such a call never takes this branch)
    const bool tn9 = !colsum_w && (g_gemm_fast_version == 4 || g_gemm_fast_version == 9) && nbatch == 1 && accumulate && K % TK == 0 && K >= 4096 && K < 32768 &&
                     M % T9M == 0 && N % T9N == 0 && (!colsum_a || N / T9N >= 4) && (long)(M / T9M) * (N / T9N) >= 8;
    if (tn9) {
        p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
        p.tiles_m = M / T9M; p.tiles_n = N / T9N;
        const int cus = std::max(8, (g_num_cus - reserved_cus(st)) / 8 * 8), ncu_x = cus / 8, ntile = p.tiles_m * p.tiles_n;
        // ranges per XCD: keep every item >= ~16 K-tiles long (atomics per output element grow with the number of ranges) and fill the XCD's
        // workgroups as evenly as the tile count allows
        const int nkt = cdiv(K, TK);
        int S = std::max(1, ncu_x / ntile);
        while (S > 1 && nkt / (8 * S) < 12) --S;
        p.splitk = S; p.ksteps = cdiv(nkt, 8 * S); p.atomic = 1; p.gm = 0;
        if (colsum_a) {
            if (int rc = enable_lds(gemm_tn_bf16_v9_kernel<true>, LDS9)) return rc;
            hipLaunchKernelGGL(gemm_tn_bf16_v9_kernel<true>, dim3((unsigned)cus), dim3(NTH8), LDS9, st, p);
        } else {
            if (int rc = enable_lds(gemm_tn_bf16_v9_kernel<false>, LDS9)) return rc;
            hipLaunchKernelGGL(gemm_tn_bf16_v9_kernel<false>, dim3((unsigned)cus), dim3(NTH8), LDS9, st, p);
        }
        TTMI_LAUNCH_CHECK("gemm_tn_bf16_v9_kernel");
        return TTMI_OK;
    }
    // huge-reduction wgrad (the joint projection: K = B*T*U1 rows): persistent 256x256 kernel on the 256-row tiles that are
    // full, the 128x128 kernel below on the remaining M % 256 rows
    int S = 1;
    bool strip_in = false;
    const bool tn8 = nbatch == 1 && accumulate && tn_v8_eligible(M, N, K, colsum_a != nullptr, &S, &strip_in);
    TTMI_REQUIRE(!colsum_w || tn8, "gemm_tn_bf16: weighted column sums exist on the persistent 256x256 kernel only (M=%d N=%d K=%d)", M, N, K);
    if (tn8) {
        const int ncu_x = g_num_cus / 8;
        p.tiles_m = M / T8; p.tiles_n = N / T8;             // full tile rows; the M % 256 strip: see below
        const int nkt = cdiv(K, TK);
        const int Mfull = strip_in ? M : M / T8 * T8;
        p.M = Mfull; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
        p.splitk = S; p.ksteps = cdiv(nkt, 8 * S); p.atomic = 1; p.gm = strip_in ? ncu_x / p.tiles_n : 0;
        if (colsum_a && p.csw) {
            if (int rc = enable_lds(gemm_tn_bf16_v8_kernel<2>, LDS8T)) return rc;
            hipLaunchKernelGGL(gemm_tn_bf16_v8_kernel<2>, dim3((unsigned)g_num_cus), dim3(NTH8), LDS8T, st, p);
        } else if (colsum_a) {
            if (int rc = enable_lds(gemm_tn_bf16_v8_kernel<1>, LDS8T)) return rc;
            hipLaunchKernelGGL(gemm_tn_bf16_v8_kernel<1>, dim3((unsigned)g_num_cus), dim3(NTH8), LDS8T, st, p);
        } else {
            if (int rc = enable_lds(gemm_tn_bf16_v8_kernel<0>, LDS8T)) return rc;
            hipLaunchKernelGGL(gemm_tn_bf16_v8_kernel<0>, dim3((unsigned)g_num_cus), dim3(NTH8), LDS8T, st, p);
        }
        TTMI_LAUNCH_CHECK("gemm_tn_bf16_v8_kernel");
        if (Mfull == M) return TTMI_OK;
        A += Mfull; C += (long)Mfull * ldc; M -= Mfull;     // the strip: same call, remaining rows of C
        if (colsum_a) colsum_a += Mfull;
        p.A = A; p.C = C; p.colsum = colsum_a;
    }
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.tiles_m = cdiv(M, TM); p.tiles_n = cdiv(N, TN_);
    const long tiles = (long)p.tiles_m * p.tiles_n;
    const int ksteps_total = cdiv(K, TK);
    // split-K: enough workgroups to fill the chip, but every split adds a tile of f32 atomics (measured on the encoder wgrads:
    // aiming at 512 workgroups beats 2048 by 10 % of the layer time).  Long reductions always get >= 8 splits, in multiples of
    // 8, so that each XCD owns a K-range (see the kernel).
    int splitk = 1;
    if (tiles * nbatch < 1024) {
        // every split adds a tile of contended f32 atomics: a 512x512 output (16 tiles) pays more for 32 adds per element than it
        // gains from 512 workgroups (measured 37 -> 29 us at 256)
        const long target = tiles * nbatch <= 16 && ksteps_total < 1024 ? std::min(g_tn_target_blocks, 256) : g_tn_target_blocks;
        splitk = (int)((target + tiles * nbatch - 1) / (tiles * nbatch));
        if (ksteps_total >= 1024 && splitk < 8) splitk = 8;
        if (splitk > ksteps_total / 4) splitk = ksteps_total / 4;
        if (splitk < 1) splitk = 1;
    }
    if (splitk >= 8) splitk = splitk / 8 * 8;      // one K-range per XCD (see the kernel)
    p.ksteps = cdiv(ksteps_total, splitk);         // the last split may be short; rows >= K are zero-filled
    if (splitk < 8) splitk = cdiv(ksteps_total, p.ksteps);
    p.splitk = splitk;
    p.atomic = (splitk > 1 || accumulate) ? 1 : 0;
    p.gm = GROUP_M;                                 // 8 x tiles_n co-resident tiles per K-range measured best (16: -7 %)
#define TN_LAUNCH(NB, CSF) hipLaunchKernelGGL((gemm_tn_bf16_kernel<NB, CSF>), dim3((unsigned)(tiles * splitk), nbatch), dim3(NTH), 2 * NB * TILE_B, st, p)
        if (g_gemm_fast_version >= 4) {
            if (colsum_a) TN_LAUNCH(1, true);
            else TN_LAUNCH(1, false);
        } else {
            if (colsum_a) TN_LAUNCH(2, true);
            else TN_LAUNCH(2, false);
        }
#undef TN_LAUNCH
    TTMI_LAUNCH_CHECK("gemm_tn_bf16_kernel");
    return TTMI_OK;
}

void gemm_fast_set_version(int v) {
    if (v == 14 || v == 15) { g_nt_stores = v == 15; return; }
    if (v >= 16) { g_bf16_mid = v - 16; return; }
    g_gemm_fast_version = v;
}
void gemm_fast_set_tn_target(int n) { g_tn_target_blocks = n; }
void gemm_fast_set_f32(int on) { g_f32_fast = on; }
int gemm_fast_f32_mode() { return g_f32_fast; }
void gemm_fast_set_reserved_cus(int n) { g_reserved_cus = n < 0 ? 0 : n; }
void gemm_fast_set_tn_group_pieces(int on) { g_tn_group_pieces = on; }
void gemm_fast_stream_reserve_cus(hipStream_t st, int n) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(g_res_mu);
    if (StreamRes* e = res_entry(dev, st, true)) { e->n = n < 0 ? 0 : n; e->has_n = true; }
}
void gemm_fast_alias_stream(hipStream_t child, hipStream_t parent) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(g_res_mu);
    if (StreamRes* e = res_entry(dev, child, true)) e->parent = parent;
}

// Grouped weight gradients (see gemm_tn_bf16_group_kernel).  Problems that do not fit the kernel's tiling run through gemm_tn_bf16 one by one
// (same results up to the order of their atomics); the rest go out in launches of up to 16 problems.
bool gemm_tn_group_fits(const TnProblem& q) {
    return q.A && q.B && q.C && q.M > 0 && q.N > 0 && q.M % T9M == 0 && q.N % T9N == 0 && q.K % TK == 0 && q.K >= 2 * TK && q.lda % 8 == 0 && q.ldb % 8 == 0 && q.lda >= q.M && q.ldb >= q.N && q.ldc >= q.N &&
           aligned16(q.A) && aligned16(q.B) && (!q.colsum || q.N / T9N >= 2) && (long)q.K * q.lda * 2 < (1L << 32) && (long)q.K * q.ldb * 2 < (1L << 32);
}
int gemm_tn_bf16_group(const TnProblem* probs, int n, hipStream_t st) {
    TTMI_REQUIRE(probs && n >= 0, "gemm_tn_bf16_group: bad arguments");
    if (g_num_cus == 0) {
        int dev = 0, cnt = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cnt, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cnt > 0)
            g_num_cus = cnt / 8 * 8;
        if (g_num_cus <= 0) g_num_cus = 256;
    }
    if (int rc = enable_lds(gemm_tn_bf16_group_kernel, LDS9)) return rc;
    TnGroup g;
    g.n = 0; g.total = 0; g.pieces = 0;
    int min_nkt = 1 << 30;
    // data-parallel backward: leave the CUs reserved on this stream to RCCL's kernels, like the v8 / v9 launches do (a persistent grid
    // larger than the CUs that are free runs its surplus workgroups after the others anyway)
    const int reserved = reserved_cus(st);
    const int cus = std::max(8, (g_num_cus - reserved) / 8 * 8);
    auto flush = [&]() -> int {
        if (g.n == 0) return TTMI_OK;
        // K-pieces for an XCD's surplus tiles only while CUs are reserved (the unreserved launches stay free of atomics: bit-reproducible) and only if the shortest
        // reduction still gives every piece a few K-tiles (P <= cus / 8 pieces per tile)
        g.pieces = reserved > 0 && min_nkt >= 2 * (cus / 8) && g_tn_group_pieces;
        min_nkt = 1 << 30;
        const bool probe = g.total >= 128;      // timing probe 4: a grouped launch that fills at least half the chip
        if (probe) ttmi_probe_begin(4, st);
        hipLaunchKernelGGL(gemm_tn_bf16_group_kernel, dim3((unsigned)cus), dim3(NTH8), LDS9, st, g);
        if (probe) ttmi_probe_end(4, st);
        TTMI_LAUNCH_CHECK("gemm_tn_bf16_group_kernel");
        g.n = 0; g.total = 0;
        return TTMI_OK;
    };
    for (int i = 0; i < n; ++i) {
        const TnProblem& q = probs[i];
        if (!gemm_tn_group_fits(q)) {
            if (int rc = gemm_tn_bf16(q.A, q.B, q.C, q.M, q.N, q.K, q.lda, q.ldb, q.ldc, 1, st, q.colsum)) return rc;
            continue;
        }
        TnGroupProb& d = g.pr[g.n];
        d.A = q.A; d.B = q.B; d.C = q.C; d.colsum = q.colsum; d.M = q.M; d.N = q.N; d.K = q.K; d.tiles_m = q.M / T9M;
        d.lda = q.lda; d.ldb = q.ldb; d.ldc = q.ldc; d.tile0 = g.total;
        g.total += d.tiles_m * (q.N / T9N);
        min_nkt = std::min(min_nkt, q.K / TK);
        if (++g.n == 16) { if (int rc = flush()) return rc; }
    }
    return flush();
}

extern "C" {
// Grouped weight gradients: descs is a HOST array of n problems C[M,N] += A[K,M]^T B[K,N] (bf16 A / B, f32 C; colsum nullable: += column
// sums of A).  Problems with M % 256 == 0, N % 128 == 0, K % 64 == 0 share launches of the no-atomics grouped kernel (bit-reproducible);
// anything else runs through ttmi_gemm_tn_bf16's path.
int ttmi_wgrad_group(const ttmi_wgrad_desc* descs, int n, void* stream) {
    static_assert(sizeof(ttmi_wgrad_desc) == sizeof(TnProblem), "ttmi_wgrad_desc and TnProblem must share one layout");
    return gemm_tn_bf16_group(reinterpret_cast<const TnProblem*>(descs), n, static_cast<hipStream_t>(stream));
}
// bring-up / test entry points (dtype codes 0 = f32, 1 = bf16)
int ttmi_gemm_nt_bf16(const void* A, const void* B, void* C, int c_dtype, const float* bias, int M, int N, int K, long lda,
                      long ldb, long ldc, void* stream) {
    NtEpilogue e;
    e.bias = bias;
    return gemm_nt_bf16(static_cast<const bf16_t*>(A), static_cast<const bf16_t*>(B), C, c_dtype, e, M, N, K, lda, ldb, ldc,
                        static_cast<hipStream_t>(stream), FastBatch());
}
// two-term weight (see NtEpilogue::B_lo): C = epi(A.(B + B_lo)^T) in one launch of a persistent kernel; relu != 0: ReLU after the bias
int ttmi_gemm_nt_bf16_two_term(const void* A, const void* B, const void* B_lo, void* C, int c_dtype, const float* bias, int relu, int M, int N, int K,
                               long lda, long ldb, long ldc, void* stream) {
    NtEpilogue e;
    e.bias = bias; e.relu = relu; e.B_lo = static_cast<const bf16_t*>(B_lo);
    return gemm_nt_bf16(static_cast<const bf16_t*>(A), static_cast<const bf16_t*>(B), C, c_dtype, e, M, N, K, lda, ldb, ldc,
                        static_cast<hipStream_t>(stream), FastBatch());
}
int ttmi_gemm_nt_bf16_two_term_klo(const void* A, const void* B, const void* B_lo, void* C, int c_dtype, const float* bias, int relu, int M, int N, int K,
                                   int k_lo, long lda, long ldb, long ldc, void* stream) {
    NtEpilogue e;
    e.bias = bias; e.relu = relu; e.B_lo = static_cast<const bf16_t*>(B_lo); e.K_lo = k_lo;
    return gemm_nt_bf16(static_cast<const bf16_t*>(A), static_cast<const bf16_t*>(B), C, c_dtype, e, M, N, K, lda, ldb, ldc,
                        static_cast<hipStream_t>(stream), FastBatch());
}
// exp store (see NtEpilogue): C = bf16(exp(A.B^T + bias - shift)), rowsum [M, nparts] partial row sums
int ttmi_gemm_nt_bf16_exp(const void* A, const void* B, void* C, const float* bias, float* rowsum, int nparts, const float* shift, int M, int N, int K,
                          long lda, long ldb, long ldc, void* stream) {
    NtEpilogue e;
    e.bias = bias; e.rowsum = rowsum; e.nparts = nparts; e.exp_shift = shift;
    return gemm_nt_bf16(static_cast<const bf16_t*>(A), static_cast<const bf16_t*>(B), C, 1, e, M, N, K, lda, ldb, ldc,
                        static_cast<hipStream_t>(stream), FastBatch());
}
// row factor (see NtEpilogue::rowscale): C = bf16((A.B^T) * (1 - mask^2) * rowscale[m]), mask <- rowscale[m] * mask in place
int ttmi_gemm_nt_bf16_rowscale(const void* A, const void* B, void* C, void* mask, const float* rowscale, int M, int N, int K, long lda, long ldb, long ldc,
                               void* stream) {
    NtEpilogue e;
    e.mask = static_cast<const bf16_t*>(mask); e.mask_mode = 1; e.rowscale = rowscale;
    return gemm_nt_bf16(static_cast<const bf16_t*>(A), static_cast<const bf16_t*>(B), C, 1, e, M, N, K, lda, ldb, ldc,
                        static_cast<hipStream_t>(stream), FastBatch());
}
int ttmi_gemm_tn_bf16(const void* A, const void* B, float* C, int M, int N, int K, long lda, long ldb, long ldc, int accumulate,
                      float* colsum_a, void* stream) {
    return gemm_tn_bf16(static_cast<const bf16_t*>(A), static_cast<const bf16_t*>(B), C, M, N, K, lda, ldb, ldc, accumulate,
                        static_cast<hipStream_t>(stream), colsum_a, FastBatch());
}
int ttmi_gemm_tn_bf16_wsum(const void* A, const void* B, float* C, int M, int N, int K, long lda, long ldb, long ldc, float* colsum_a,
                           const void* colsum_w, void* stream) {
    return gemm_tn_bf16(static_cast<const bf16_t*>(A), static_cast<const bf16_t*>(B), C, M, N, K, lda, ldb, ldc, 1,
                        static_cast<hipStream_t>(stream), colsum_a, FastBatch(), static_cast<const bf16_t*>(colsum_w));
}
}
