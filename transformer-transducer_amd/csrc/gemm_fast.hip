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
};

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
                if (p.mask) v = bf16_to_f32(p.mask[ci]) > 0.f ? v * p.scale : 0.f;
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
template <typename TC, int NBUF, bool PIPE>
__global__ __launch_bounds__(NTH, NBUF == 1 ? 4 : 2) void gemm_nt_bf16_kernel(const FP p_) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [NBUF buffers][A 16K | B 16K]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    FP p = p_;
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
    }
    store_tile<TC>(acc, p, reinterpret_cast<TC*>(p.C), bm, bn, wm, wn, lane, p.bias != nullptr);
}

// ------------------------------------------------------------------ TN: A[K,M], B[K,N], reduction index is the slow one
__device__ __forceinline__ bf16x4 ds_read_tr16(const char* lds_addr) {
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)lds_addr);
    return __builtin_bit_cast(bf16x4, v);
}

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
                cs += bf16_to_f32(*reinterpret_cast<const bf16_t*>(la + kr * 256 + ((((col >> 3) ^ ((kr & 3) << 2))) << 4) + (col & 7) * 2));
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

// =====================================================================================================================
// v6: 256x256x64 tile, 8 waves (4x2), each wave 64x128 (2x4 MFMA tiles, 128 accumulator registers), double-buffered
// 2 x 64 KiB LDS, one workgroup per CU.  Half the L2->LDS bytes per FLOP of the 128x128 kernels.
// =====================================================================================================================
constexpr int T6 = 256, NTH6 = 512, STAGE6 = 2 * 256 * 64 * 2;   // 64 KiB per stage (A 32 KiB | B 32 KiB)

template <typename TC>
__global__ __launch_bounds__(NTH6, 1) void gemm_nt_bf16_v6_kernel(const FP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                       // 4 x 2 waves, wave tile 64 (M) x 128 (N)
    int tm, tn;
    tile_of(blockIdx.x, gridDim.x, p.tiles_m, p.tiles_n, tm, tn);
    const int bm = tm * T6, bn = tn * T6;

    const bf16_t* asrc[4];
    const bf16_t* bsrc[4];
    int chunkk[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int R = (wave * 4 + j) * 8 + (lane >> 3);            // 0..255
        chunkk[j] = (lane & 7) ^ ((R >> 1) & 7);
        asrc[j] = p.A + (long)min(bm + R, p.M - 1) * p.lda + chunkk[j] * 8;
        bsrc[j] = p.B + (long)min(bn + R, p.N - 1) * p.ldb + chunkk[j] * 8;
    }
    const void* zsrc = &g_zero16;
    auto issue = [&](int buf, int kt) {
        char* base = smem + buf * STAGE6;
        const bool tail = (kt + 1) * TK > p.K;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool zero = tail && (kt * TK + chunkk[j] * 8 >= p.K);
            glds16(zero ? zsrc : (const void*)(asrc[j] + (long)kt * TK), base + (wave * 4 + j) * 1024);
            glds16(zero ? zsrc : (const void*)(bsrc[j] + (long)kt * TK), base + STAGE6 / 2 + (wave * 4 + j) * 1024);
        }
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (p.K + TK - 1) / TK;
    const int sw = (lane >> 1) & 7;
    const int rowa = wm * 64 + (lane & 31), rowb = wn * 128 + (lane & 31);
    auto frag = [&](const char* la, const char* lb, int kk, bf16x8 (&af)[2], bf16x8 (&bf)[4]) {
        const int c = ((kk * 2 + (lane >> 5)) ^ sw) << 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const bf16x8*>(la + (rowa + i * 32) * 128 + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(lb + (rowb + j * 32) * 128 + c);
    };
    auto mma = [&](const bf16x8 (&af)[2], const bf16x8 (&bf)[4]) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    };
    issue(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) issue(cur ^ 1, kt + 1);
        const char* la = smem + cur * STAGE6;
        const char* lb = la + STAGE6 / 2;
        bf16x8 a0[2], b0[4], a1[2], b1[4];
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
        __syncthreads();
    }
    // epilogue (wave tile 64 x 128)
    TC* C = reinterpret_cast<TC*>(p.C);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = bn + wn * 128 + j * 32 + (lane & 31);
            if (n >= p.N) continue;
            const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = bm + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= p.M) continue;
                float v = acc[i][j][r] + bv;
                const long ci = (long)m * p.ldc + n;
                if (p.addend) v += p.addend[ci];
                if (p.relu) v = fmaxf(v, 0.f);
                if (p.mask) v = bf16_to_f32(p.mask[ci]) > 0.f ? v * p.scale : 0.f;
                v *= drop_mult(p.drop, (unsigned long long)ci);
                if constexpr (sizeof(TC) == 4) reinterpret_cast<float*>(C)[ci] = v;
                else reinterpret_cast<bf16_t*>(C)[ci] = f32_to_bf16(v);
            }
        }
}

template <typename K>
int enable_lds(K kernel, int bytes) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) { ttmi_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    return TTMI_OK;
}

// ttmi_set_option(1, v) - A/B measurements: 1 = 128x128 double-buffered, 3 = 1 + software-pipelined fragment reads,
// 4 (default) = single 32 KiB buffer, 4 workgroups per CU, one fragment set (stays under 128 VGPRs without spills; NT +15 % at
// the joint shapes; TN +8 % once each XCD owns a K-range) with the 256x256 kernel for long-K shapes, 6 = 256x256 wherever it fits.
// Measured and dropped (same box, joint projection M=816000 N=4334 K=1024, v4 = 720 TFLOP/s): 256x128 3-stage ring with
// counted vmcnt 621; persistent 256x256 with a 4-slice ring that never drains 668 (dgrad K=4352: 904 vs 938 for v6).
int g_gemm_fast_version = 4;
int g_tn_target_blocks = 512;     // split-K aims at this many workgroups for small outputs (ttmi_set_option(4, n))


}  // namespace

bool gemm_fast_nt_ok(const void* A, const void* B, const void* C, int M, int N, int K, long lda, long ldb) {
    return aligned16(A) && aligned16(B) && C && M > 0 && N > 0 && K >= 8 && K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 &&
           lda >= K && ldb >= K;
}

static void fill_batch(FP& p, const FastBatch& b) {
    p.nz2 = b.nz2; p.sA1 = b.sA1; p.sA2 = b.sA2; p.sB1 = b.sB1; p.sB2 = b.sB2; p.sC1 = b.sC1; p.sC2 = b.sC2; p.sV1 = b.sV1; p.sV2 = b.sV2;
}

int gemm_nt_bf16(const bf16_t* A, const bf16_t* B, void* C, int c_dtype, const NtEpilogue& epi, int M, int N, int K, long lda,
                 long ldb, long ldc, hipStream_t st, const FastBatch& batch) {
    TTMI_REQUIRE(gemm_fast_nt_ok(A, B, C, M, N, K, lda, ldb), "gemm_nt_bf16: shape/alignment not supported (M=%d N=%d K=%d)", M,
                 N, K);
    FP p;
    p.A = A; p.B = B; p.C = C; p.bias = epi.bias; p.addend = epi.addend; p.mask = epi.mask; p.relu = epi.relu; p.scale = epi.scale; p.drop = epi.drop;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.tiles_m = cdiv(M, TM); p.tiles_n = cdiv(N, TN_); p.splitk = 1; p.ksteps = cdiv(K, TK); p.atomic = 0; p.gm = GROUP_M; p.colsum = nullptr;
    fill_batch(p, batch);
    const int nbatch = batch.nz1 * batch.nz2;
    TTMI_REQUIRE(nbatch >= 1 && nbatch <= 65535, "gemm_nt_bf16: bad batch count %d", nbatch);
    // 256x256 tiles pay off when the K loop is long enough to amortise the un-overlapped epilogue of a one-workgroup-per-CU
    // kernel and there are enough tiles to fill the chip (joint dgrad: K = 4352 -> 954 vs 880 TFLOP/s; short-K forward: worse)
    const bool big = (g_gemm_fast_version == 6) || (g_gemm_fast_version == 4 && K >= 2048 && (long)cdiv(M, T6) * cdiv(N, T6) >= 1024);
    if (big && M >= 1024 && N >= 256 && nbatch == 1) {
        p.tiles_m = cdiv(M, T6); p.tiles_n = cdiv(N, T6);
        const long nwg6 = (long)p.tiles_m * p.tiles_n;
        if (c_dtype == 0) {
            if (int rc = enable_lds(gemm_nt_bf16_v6_kernel<float>, 2 * STAGE6)) return rc;
            hipLaunchKernelGGL(gemm_nt_bf16_v6_kernel<float>, dim3((unsigned)nwg6), dim3(NTH6), 2 * STAGE6, st, p);
        } else {
            if (int rc = enable_lds(gemm_nt_bf16_v6_kernel<bf16_t>, 2 * STAGE6)) return rc;
            hipLaunchKernelGGL(gemm_nt_bf16_v6_kernel<bf16_t>, dim3((unsigned)nwg6), dim3(NTH6), 2 * STAGE6, st, p);
        }
        TTMI_LAUNCH_CHECK("gemm_nt_bf16_v6_kernel");
        return TTMI_OK;
    }
    const long nwg = (long)p.tiles_m * p.tiles_n;
    TTMI_REQUIRE(nwg < (1L << 31), "gemm_nt_bf16: too many tiles");
    const int ver = g_gemm_fast_version;
#define NT_LAUNCH(TCT, NB, PP) hipLaunchKernelGGL((gemm_nt_bf16_kernel<TCT, NB, PP>), dim3((unsigned)nwg, nbatch), dim3(NTH), 2 * NB * TILE_B, st, p)
    if (c_dtype == 0) {
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

bool gemm_fast_tn_ok(const void* A, const void* B, const void* C, int M, int N, int K, long lda, long ldb) {
    return aligned16(A) && aligned16(B) && C && M > 0 && N > 0 && K >= 1 && lda % 8 == 0 && ldb % 8 == 0 &&
           lda >= ((M + 7) & ~7) && ldb >= ((N + 7) & ~7);
}

// C (f32) += A^T B by atomics when splitk > 1 or accumulate != 0, else C = A^T B
int gemm_tn_bf16(const bf16_t* A, const bf16_t* B, float* C, int M, int N, int K, long lda, long ldb, long ldc, int accumulate,
                 hipStream_t st, float* colsum_a, const FastBatch& batch) {
    TTMI_REQUIRE(gemm_fast_tn_ok(A, B, C, M, N, K, lda, ldb), "gemm_tn_bf16: shape/alignment not supported (M=%d N=%d K=%d)", M,
                 N, K);
    FP p;
    p.A = A; p.B = B; p.C = C; p.bias = nullptr; p.addend = nullptr; p.mask = nullptr; p.relu = 0; p.scale = 1.f; p.drop = DropSpec();
    p.colsum = colsum_a;
    fill_batch(p, batch);
    const int nbatch = batch.nz1 * batch.nz2;
    TTMI_REQUIRE(nbatch >= 1 && nbatch <= 65535, "gemm_tn_bf16: bad batch count %d", nbatch);
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.tiles_m = cdiv(M, TM); p.tiles_n = cdiv(N, TN_);
    const long tiles = (long)p.tiles_m * p.tiles_n;
    const int ksteps_total = cdiv(K, TK);
    // split-K: enough workgroups to fill the chip, but every split adds a tile of f32 atomics (measured on the encoder wgrads:
    // aiming at 512 workgroups beats 2048 by 10 % of the layer time).  Long reductions always get >= 8 splits, in multiples of
    // 8, so that each XCD owns a K-range (see the kernel).
    int splitk = 1;
    if (tiles * nbatch < 1024) {
        splitk = (int)((g_tn_target_blocks + tiles * nbatch - 1) / (tiles * nbatch));
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

void gemm_fast_set_version(int v) { g_gemm_fast_version = v; }
void gemm_fast_set_tn_target(int n) { g_tn_target_blocks = n; }

extern "C" {
// bring-up / test entry points (dtype codes 0 = f32, 1 = bf16)
int ttmi_gemm_nt_bf16(const void* A, const void* B, void* C, int c_dtype, const float* bias, int M, int N, int K, long lda,
                      long ldb, long ldc, void* stream) {
    NtEpilogue e;
    e.bias = bias;
    return gemm_nt_bf16(static_cast<const bf16_t*>(A), static_cast<const bf16_t*>(B), C, c_dtype, e, M, N, K, lda, ldb, ldc,
                        static_cast<hipStream_t>(stream), FastBatch());
}
int ttmi_gemm_tn_bf16(const void* A, const void* B, float* C, int M, int N, int K, long lda, long ldb, long ldc, int accumulate,
                      float* colsum_a, void* stream) {
    return gemm_tn_bf16(static_cast<const bf16_t*>(A), static_cast<const bf16_t*>(B), C, M, N, K, lda, ldb, ldc, accumulate,
                        static_cast<hipStream_t>(stream), colsum_a, FastBatch());
}
}
