// Generic MFMA GEMM for gfx950:  C[z] = alpha * op(A[z]) . op(B[z]) + beta * C[z]  (+bias, ReLU, mask, atomics)
//
// One kernel family serves every dense contraction on the Transformer-Transducer path
// (QKV / output / FFN / joint projections, the attention score products and all their
// backward forms): operands may be K-major or M/N-major (so dgrad and wgrad need no
// transposed copies), f32 or bf16 in memory, batched with two-level strides.
//
//   compute = bf16 : v_mfma_f32_32x32x16_bf16, f32 accumulate   (throughput path)
//   compute = f32  : v_mfma_f32_32x32x2_f32, bit-exact fmaf chain (parity path, 1e-4 gate)
//
// Block tile 128x128, 4 waves (2x2), each wave 64x64 = 2x2 MFMA tiles of 32x32.  Operands are
// register-staged (global -> VGPR -> convert -> LDS) so any source layout/dtype lands in the
// same K-contiguous LDS image; M/N-major sources are transposed in registers (4x4 blocks) on the
// way in.  LDS rows are padded (80 B for bf16, 17 dwords for f32) so the ds_read_b128 / ds_read_b32
// fragment reads are bank-conflict free.  Global loads for tile k+1 are issued before the MFMAs of
// tile k (double-buffered LDS, one barrier per K-step).
#include "gemm.h"
#include "gemm_fast.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BM = 128, BN = 128, NT = 256;

template <bool BF16C>
struct Cfg {
    static constexpr int BK = BF16C ? 32 : 16;
    static constexpr int LD = BF16C ? 40 : 17;                 // LDS row pitch in elements
    static constexpr int ESZ = BF16C ? 2 : 4;
    static constexpr int TILE_BYTES = 128 * LD * ESZ;          // one operand tile
};

struct KParams {
    const void* A;
    const void* B;
    void* C;
    const float* bias;
    const float* aux;
    int M, N, K;
    long lda, ldb, ldc;
    int nz2;
    long sA1, sA2, sB1, sB2, sC1, sC2, sBias1, sBias2;
    float alpha, beta;
    int flags;
    int splitk, kchunk;
    int vecA, vecB;
    DropSpec drop;
};

template <typename S>
__device__ __forceinline__ float ld1(const S* p);
template <>
__device__ __forceinline__ float ld1<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ld1<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }

// 4 consecutive source elements starting at p (nvalid of them in range), as floats
template <typename S>
__device__ __forceinline__ float4 load4(const S* p, int nvalid, bool vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nvalid >= 4 && vec) {
        if constexpr (sizeof(S) == 4) {
            v = *reinterpret_cast<const float4*>(p);
        } else {
            const uint2 w = *reinterpret_cast<const uint2*>(p);
            v.x = __uint_as_float(w.x << 16);
            v.y = __uint_as_float(w.x & 0xffff0000u);
            v.z = __uint_as_float(w.y << 16);
            v.w = __uint_as_float(w.y & 0xffff0000u);
        }
    } else {
        if (nvalid > 0) v.x = ld1<S>(p);
        if (nvalid > 1) v.y = ld1<S>(p + 1);
        if (nvalid > 2) v.z = ld1<S>(p + 2);
        if (nvalid > 3) v.w = ld1<S>(p + 3);
    }
    return v;
}

// Staging of one 128 x BK operand tile.  KMAJOR: element (r,k) at src[r*ld + k]; else src[k*ld + r].
// X3 (BF16C only, f32 sources): a tile is staged TWICE - hi = bf16(x) at `lds`, lo = bf16(x - hi) at `lds` + 2 * TILE_BYTES - for the three-term
// product of the bf16x3 parity mode
template <typename S, bool KMAJOR, bool BF16C, bool X3 = false>
struct Stager {
    using C = Cfg<BF16C>;
    static constexpr int NV = KMAJOR ? (128 * C::BK / 4) / NT : ((C::BK / 4) * 32 + NT - 1) / NT;   // float4 groups / thread
    float4 v[KMAJOR ? NV : 4 * NV];

    __device__ __forceinline__ void load(const S* src, long ld, int r0, int R, int k0, int Kend, bool vec, int tid) {
        if constexpr (KMAJOR) {
#pragma unroll
            for (int s = 0; s < NV; ++s) {
                const int idx = tid + NT * s;
                const int row = idx / (C::BK / 4), kq = idx % (C::BK / 4);
                const int gr = r0 + row, gk = k0 + kq * 4;
                const int nvalid = (gr < R) ? (Kend - gk) : 0;
                v[s] = load4<S>(src + (long)gr * ld + gk, nvalid, vec);
            }
        } else {
#pragma unroll
            for (int s = 0; s < NV; ++s) {
                const int idx = tid + NT * s;
                const int kb = idx / 32, rb = idx % 32;
                const bool on = idx < (C::BK / 4) * 32;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int gk = k0 + kb * 4 + kk, gr = r0 + rb * 4;
                    const int nvalid = (on && gk < Kend) ? (R - gr) : 0;
                    v[s * 4 + kk] = load4<S>(src + (long)gk * ld + gr, nvalid, vec);
                }
            }
        }
    }

    __device__ __forceinline__ void store(char* lds, int tid) const {
        if constexpr (KMAJOR) {
#pragma unroll
            for (int s = 0; s < NV; ++s) {
                const int idx = tid + NT * s;
                const int row = idx / (C::BK / 4), kq = idx % (C::BK / 4);
                if constexpr (BF16C) {
                    uint2 w;
                    w.x = pack_bf16x2(v[s].x, v[s].y);
                    w.y = pack_bf16x2(v[s].z, v[s].w);
                    *reinterpret_cast<uint2*>(lds + (row * C::LD + kq * 4) * 2) = w;
                    if constexpr (X3) {
                        uint2 l;
                        l.x = pack_bf16x2(v[s].x - __uint_as_float(w.x << 16), v[s].y - __uint_as_float(w.x & 0xffff0000u));
                        l.y = pack_bf16x2(v[s].z - __uint_as_float(w.y << 16), v[s].w - __uint_as_float(w.y & 0xffff0000u));
                        *reinterpret_cast<uint2*>(lds + 2 * C::TILE_BYTES + (row * C::LD + kq * 4) * 2) = l;
                    }
                } else {
                    float* d = reinterpret_cast<float*>(lds) + row * C::LD + kq * 4;
                    d[0] = v[s].x; d[1] = v[s].y; d[2] = v[s].z; d[3] = v[s].w;
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < NV; ++s) {
                const int idx = tid + NT * s;
                const int kb = idx / 32, rb = idx % 32;
                if (idx < (C::BK / 4) * 32) {
                    const float4 a = v[s * 4 + 0], b = v[s * 4 + 1], c = v[s * 4 + 2], d = v[s * 4 + 3];
                    const float t0[4] = {a.x, b.x, c.x, d.x}, t1[4] = {a.y, b.y, c.y, d.y};
                    const float t2[4] = {a.z, b.z, c.z, d.z}, t3[4] = {a.w, b.w, c.w, d.w};
                    const float* tr[4] = {t0, t1, t2, t3};
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int row = rb * 4 + rr;
                        if constexpr (BF16C) {
                            uint2 w;
                            w.x = pack_bf16x2(tr[rr][0], tr[rr][1]);
                            w.y = pack_bf16x2(tr[rr][2], tr[rr][3]);
                            *reinterpret_cast<uint2*>(lds + (row * C::LD + kb * 4) * 2) = w;
                            if constexpr (X3) {
                                uint2 l;
                                l.x = pack_bf16x2(tr[rr][0] - __uint_as_float(w.x << 16), tr[rr][1] - __uint_as_float(w.x & 0xffff0000u));
                                l.y = pack_bf16x2(tr[rr][2] - __uint_as_float(w.y << 16), tr[rr][3] - __uint_as_float(w.y & 0xffff0000u));
                                *reinterpret_cast<uint2*>(lds + 2 * C::TILE_BYTES + (row * C::LD + kb * 4) * 2) = l;
                            }
                        } else {
                            float* dd = reinterpret_cast<float*>(lds) + row * C::LD + kb * 4;
                            dd[0] = tr[rr][0]; dd[1] = tr[rr][1]; dd[2] = tr[rr][2]; dd[3] = tr[rr][3];
                        }
                    }
                }
            }
        }
    }
};

template <typename SA, typename SB, typename TC, bool AK, bool BKM, bool BF16C, bool X3 = false>
__global__ __launch_bounds__(NT) void gemm_kernel(const KParams p_) {
    KParams p = p_;
    p.drop = drop_live(p.drop);
    using C = Cfg<BF16C>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // layout: [buf0: A | B][buf1: A | B]; X3: [buf: A_hi | B_hi | A_lo | B_lo]
    constexpr int PER_BUF = X3 ? 4 : 2;
    auto ldsA = [&](int buf) -> char* { return smem + buf * PER_BUF * C::TILE_BYTES; };
    auto ldsB = [&](int buf) -> char* { return smem + buf * PER_BUF * C::TILE_BYTES + C::TILE_BYTES; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int bm = blockIdx.y * BM, bn = blockIdx.x * BN;
    const int z = blockIdx.z / p.splitk, ks = blockIdx.z % p.splitk;
    const int z1 = z / p.nz2, z2 = z % p.nz2;
    const SA* A = reinterpret_cast<const SA*>(p.A) + z1 * p.sA1 + z2 * p.sA2;
    const SB* B = reinterpret_cast<const SB*>(p.B) + z1 * p.sB1 + z2 * p.sB2;
    TC* Cp = reinterpret_cast<TC*>(p.C) + z1 * p.sC1 + z2 * p.sC2;
    const int kbeg = ks * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);
    const int nk = (kend - kbeg + C::BK - 1) / C::BK;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    Stager<SA, AK, BF16C, X3> sa;
    Stager<SB, BKM, BF16C, X3> sb;
    if (nk > 0) {
        sa.load(A, p.lda, bm, p.M, kbeg, kend, p.vecA, tid);
        sb.load(B, p.ldb, bn, p.N, kbeg, kend, p.vecB, tid);
        sa.store(ldsA(0), tid);
        sb.store(ldsB(0), tid);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            sa.load(A, p.lda, bm, p.M, kbeg + (kt + 1) * C::BK, kend, p.vecA, tid);
            sb.load(B, p.ldb, bn, p.N, kbeg + (kt + 1) * C::BK, kend, p.vecB, tid);
        }
        const char* la = ldsA(cur);
        const char* lb = ldsB(cur);
        if constexpr (BF16C) {
#pragma unroll
            for (int kk = 0; kk < C::BK / 16; ++kk) {
                bf16x8 af[2], bf[2];
                const int kof = kk * 16 + 8 * (lane >> 5);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = wm * 64 + i * 32 + (lane & 31);
                    af[i] = *reinterpret_cast<const bf16x8*>(la + (row * C::LD + kof) * 2);
                    const int col = wn * 64 + i * 32 + (lane & 31);
                    bf[i] = *reinterpret_cast<const bf16x8*>(lb + (col * C::LD + kof) * 2);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
                if constexpr (X3) {                               // + lo . hi + hi . lo (the lo tiles sit 2 tiles behind their hi tiles)
                    bf16x8 al[2], bl[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int row = wm * 64 + i * 32 + (lane & 31);
                        al[i] = *reinterpret_cast<const bf16x8*>(la + 2 * C::TILE_BYTES + (row * C::LD + kof) * 2);
                        const int col = wn * 64 + i * 32 + (lane & 31);
                        bl[i] = *reinterpret_cast<const bf16x8*>(lb + 2 * C::TILE_BYTES + (col * C::LD + kof) * 2);
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bf[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bl[j], acc[i][j], 0, 0, 0);
                        }
                }
            }
        } else {
            const float* fa = reinterpret_cast<const float*>(la);
            const float* fb = reinterpret_cast<const float*>(lb);
#pragma unroll
            for (int kk = 0; kk < C::BK / 2; ++kk) {
                float af[2], bf[2];
                const int kof = kk * 2 + (lane >> 5);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    af[i] = fa[(wm * 64 + i * 32 + (lane & 31)) * C::LD + kof];
                    bf[i] = fb[(wn * 64 + i * 32 + (lane & 31)) * C::LD + kof];
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
        if (kt + 1 < nk) {
            sa.store(ldsA(cur ^ 1), tid);
            sb.store(ldsB(cur ^ 1), tid);
        }
        __syncthreads();
    }

    // epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const float* bias = (p.flags & GEMM_BIAS) ? p.bias + z1 * p.sBias1 + z2 * p.sBias2 : nullptr;
    const float* aux = (p.flags & GEMM_MASK_AUX) ? p.aux + z1 * p.sC1 + z2 * p.sC2 : nullptr;
    const bool first_split = (ks == 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = bn + wn * 64 + j * 32 + (lane & 31);
            if (n >= p.N) continue;
            const float bv = (bias && first_split) ? bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = bm + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= p.M) continue;
                const long ci = (long)m * p.ldc + n;
                float v = p.alpha * acc[i][j][r] + bv;
                if constexpr (sizeof(TC) == 4) {
                    if (p.flags & GEMM_ATOMIC) {
                        atomicAdd(reinterpret_cast<float*>(Cp) + ci, v);
                        continue;
                    }
                    if (p.beta != 0.f) v += p.beta * reinterpret_cast<const float*>(Cp)[ci];
                } else {
                    if (p.beta != 0.f) v += p.beta * bf16_to_f32(reinterpret_cast<const bf16_t*>(Cp)[ci]);
                }
                if (p.flags & GEMM_RELU) v = fmaxf(v, 0.f);
                if (aux) v = aux[ci] > 0.f ? v : 0.f;
                v *= drop_mult(p.drop, (unsigned long long)ci);
                if constexpr (sizeof(TC) == 4) reinterpret_cast<float*>(Cp)[ci] = v;
                else reinterpret_cast<bf16_t*>(Cp)[ci] = f32_to_bf16(v);
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// Skinny exact-f32 GEMM (M <= 128 rows, A K-major): the greedy decoder runs the label encoder on ONE history of L tokens and
// the joint on a block of <= 64 frames, so every product has a handful of rows.  The 128x128 kernel above spends its whole
// K loop (K/16 barrier-separated steps of one workgroup per 128 columns) on a mostly empty tile: 67 us per launch, 88 % of a
// decode step.  Here a workgroup owns a 32 x 32 output tile and its 4 waves split the reduction: wave w takes the 8-wide
// k-chunks w, w+4, ...; operands go straight from global memory (L2-resident at these sizes) into the MFMA registers - lane
// (r = lane & 31, g = lane >> 5) loads 4 consecutive k of row r at chunk offset 4g, MFMA c of the chunk consumes component c
// of both operands (any pairing of k indices is a valid reduction order as long as A and B agree) - then the four partial
// tiles meet in LDS in a fixed order (deterministic) and all 256 threads run the epilogue with row-contiguous stores.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int SK_T = 32, SK_PITCH = 33;

template <bool BKM>
__global__ __launch_bounds__(NT) void gemm_skinny_f32_kernel(const KParams p_) {
    KParams p = p_;
    p.drop = drop_live(p.drop);
    __shared__ float red[4][SK_T][SK_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, g = lane >> 5;
    const int bm = blockIdx.y * SK_T, bn = blockIdx.x * SK_T;
    const int z = blockIdx.z, z1 = z / p.nz2, z2 = z % p.nz2;
    const float* A = reinterpret_cast<const float*>(p.A) + z1 * p.sA1 + z2 * p.sA2;
    const float* B = reinterpret_cast<const float*>(p.B) + z1 * p.sB1 + z2 * p.sB2;
    float* Cp = reinterpret_cast<float*>(p.C) + z1 * p.sC1 + z2 * p.sC2;
    // rows / columns beyond the matrix are clamped for the loads and never stored
    const float* arow = A + (long)min(bm + r, p.M - 1) * p.lda;
    const int ncol = min(bn + r, p.N - 1);
    const float* brow = BKM ? B + (long)ncol * p.ldb : B + ncol;
    const int nchunk = (p.K + 7) >> 3;

    auto load_a = [&](int k) { return load4<float>(arow + k, p.K - k, p.vecA); };
    auto load_b = [&](int k) {
        if constexpr (BKM) {
            return load4<float>(brow + k, p.K - k, p.vecB);
        } else {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < p.K) v.x = brow[(long)k * p.ldb];
            if (k + 1 < p.K) v.y = brow[(long)(k + 1) * p.ldb];
            if (k + 2 < p.K) v.z = brow[(long)(k + 2) * p.ldb];
            if (k + 3 < p.K) v.w = brow[(long)(k + 3) * p.ldb];
            return v;
        }
    };

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    constexpr int UN = 4;                                  // chunks in flight per wave: 8 float4 loads, then 16 MFMAs
    for (int c0 = wave; c0 < nchunk; c0 += 4 * UN) {
        float4 av[UN], bv[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int k = (c0 + 4 * u) * 8 + 4 * g;        // chunks past the end load zeros (nvalid <= 0)
            av[u] = load_a(k);
            bv[u] = load_b(k);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].x, bv[u].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].y, bv[u].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].z, bv[u].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].w, bv[u].w, acc, 0, 0, 0);
        }
    }
    // C/D map of the 32x32 MFMA: col = lane & 31, row = (i & 3) + 8 (i >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int i = 0; i < 16; ++i) red[wave][(i & 3) + 8 * (i >> 2) + 4 * g][r] = acc[i];
    __syncthreads();

    const float* bias = (p.flags & GEMM_BIAS) ? p.bias + z1 * p.sBias1 + z2 * p.sBias2 : nullptr;
    const float* aux = (p.flags & GEMM_MASK_AUX) ? p.aux + z1 * p.sC1 + z2 * p.sC2 : nullptr;
#pragma unroll
    for (int i = 0; i < SK_T * SK_T / NT; ++i) {
        const int idx = tid + NT * i, row = idx >> 5, col = idx & 31;
        const int m = bm + row, n = bn + col;
        if (m >= p.M || n >= p.N) continue;
        const long ci = (long)m * p.ldc + n;
        float v = p.alpha * (((red[0][row][col] + red[1][row][col]) + red[2][row][col]) + red[3][row][col]);
        if (bias) v += bias[n];
        if (p.beta != 0.f) v += p.beta * Cp[ci];
        if (p.flags & GEMM_RELU) v = fmaxf(v, 0.f);
        if (aux) v = aux[ci] > 0.f ? v : 0.f;
        v *= drop_mult(p.drop, (unsigned long long)ci);
        Cp[ci] = v;
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// TTMI_PRECISION=bf16x3: the attention core's batched products (round 5).  Every one of them streams one [L, L] f32 slab per (batch, head)
// - 256 MB at B = 32, H = 8, L = 500 - against operands of Dh = 64 columns.  Through the 128 x 128 kernel above they ran at 1.2 TB/s of that
// slab whatever the arithmetic (exact f32 and three-term bf16 within 3 %): a tile walks its 128 slab rows in 128-byte strips, one barrier-separated
// step per strip with two workgroups per CU, so the bytes in flight - not the MFMA - set the pace (tools/debug/bench_generic_gemm.py; the library's
// f32 batched product on the same operands: 95 us reading, 179 us writing the slab).  Two shapes cover them:
//   x3_panel64_kernel   C[M, 64] (+)= A[M, K] B[K, 64]    (P V, dS K, dG E: A k-major; P^T dO, dS^T (q + u), dG^T q: A m-major).  The slab operand A has
//       no reuse across waves (a wave's 32 rows meet all 64 columns), so it goes from global memory straight into the MFMA A fragments - split
//       into bf16 hi + lo in registers, the next 32-wide strip already in flight - and only the small B strip (32 x 64, L2-resident) passes
//       through LDS, transposed and split once per workgroup: one barrier per strip, 20 KB of LDS, four to five workgroups per CU.
//   x3_rows_nt64_kernel C[M, N] (+)= A[M, 64] B[N, 64]^T (+ bias)   (q E^T, (q + u) k^T, dO V^T).  A workgroup owns 32 COMPLETE rows of the slab
//       - one contiguous 64 KB region - and its waves take the 32-column blocks in turn: both operands straight from global memory (k-major,
//       L2-resident) into fragments, no LDS, no barrier.
// Both: three bf16 MFMA terms hi.hi + lo.hi + hi.lo per fragment pair, f32 accumulation.
// ---------------------------------------------------------------------------------------------------------------------
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split8(const float4 a, const float4 b, bf16x8& hi, bf16x8& lo) {
    u32x4 h, l;
    h.x = pack_bf16x2(a.x, a.y); h.y = pack_bf16x2(a.z, a.w); h.z = pack_bf16x2(b.x, b.y); h.w = pack_bf16x2(b.z, b.w);
    l.x = pack_bf16x2(a.x - __uint_as_float(h.x << 16), a.y - __uint_as_float(h.x & 0xffff0000u));
    l.y = pack_bf16x2(a.z - __uint_as_float(h.y << 16), a.w - __uint_as_float(h.y & 0xffff0000u));
    l.z = pack_bf16x2(b.x - __uint_as_float(h.z << 16), b.y - __uint_as_float(h.z & 0xffff0000u));
    l.w = pack_bf16x2(b.z - __uint_as_float(h.w << 16), b.w - __uint_as_float(h.w & 0xffff0000u));
    hi = __builtin_bit_cast(bf16x8, h);
    lo = __builtin_bit_cast(bf16x8, l);
}

constexpr int P64_BM = 128;      // rows per workgroup; KC = strip width, KC + 8 = LDS pitch of a B column (bf16 elements)

template <bool AK, int KC>
__global__ __launch_bounds__(NT) void x3_panel64_kernel(const KParams p) {
    constexpr int P64_KC = KC, P64_LD = KC + 8, NKK = KC / 16, NB = KC / 32;
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2][2][64 * P64_LD];      // [stage][hi | lo][column n][k]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, g = lane >> 5;
    const int bm = blockIdx.x * P64_BM;
    const int z = blockIdx.y, z1 = z / p.nz2, z2 = z % p.nz2;
    const float* A = reinterpret_cast<const float*>(p.A) + z1 * p.sA1 + z2 * p.sA2;
    const float* B = reinterpret_cast<const float*>(p.B) + z1 * p.sB1 + z2 * p.sB2;
    float* Cp = reinterpret_cast<float*>(p.C) + z1 * p.sC1 + z2 * p.sC2;
    const int row = bm + 32 * wave + r;
    const int rowc = min(row, p.M - 1);                    // rows past the matrix are loaded from its last row and never stored
    const int nc = (p.K + P64_KC - 1) / P64_KC;

    // lane (r, g) of a 32x32x16 MFMA supplies k = 8 g ... 8 g + 7 of its row: two float4 per 16-wide block, two blocks per strip
    auto load_a = [&](int k0, float4* d) {
        if constexpr (AK) {
            const float* ap = A + (long)rowc * p.lda;
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const int k = k0 + 16 * kk + 8 * g;
                d[2 * kk] = load4<float>(ap + k, p.K - k, p.vecA);
                d[2 * kk + 1] = load4<float>(ap + k + 4, p.K - k - 4, p.vecA);
            }
        } else {                                           // A stored [K, M]: 32 consecutive rows per load instruction
            const float* ap = A + rowc;
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const int k = k0 + 16 * kk + 8 * g;
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = (k + i < p.K) ? ap[(long)(k + i) * p.lda] : 0.f;
                d[2 * kk] = make_float4(v[0], v[1], v[2], v[3]);
                d[2 * kk + 1] = make_float4(v[4], v[5], v[6], v[7]);
            }
        }
    };
    // B strip [KC k][64 n] (n contiguous): thread (kp = tid >> 4, n4 = 4 (tid & 15)) takes rows 2 kp, 2 kp + 1 (+ 32 per further block) of four columns
    const int kp = tid >> 4, n4 = (tid & 15) * 4;
    auto load_b = [&](int k0, float4* d) {
#pragma unroll
        for (int s2 = 0; s2 < 2 * NB; ++s2) {
            const int k = k0 + 32 * (s2 >> 1) + 2 * kp + (s2 & 1);
            d[s2] = load4<float>(B + (long)k * p.ldb + n4, k < p.K ? 4 : 0, p.vecB);
        }
    };
    auto store_b = [&](int st, const float4* d) {
        unsigned* hi = reinterpret_cast<unsigned*>(Bs[st][0]);
        unsigned* lo = reinterpret_cast<unsigned*>(Bs[st][1]);
#pragma unroll
        for (int b2 = 0; b2 < NB; ++b2) {
            const float x0[4] = {d[2 * b2].x, d[2 * b2].y, d[2 * b2].z, d[2 * b2].w}, x1[4] = {d[2 * b2 + 1].x, d[2 * b2 + 1].y, d[2 * b2 + 1].z, d[2 * b2 + 1].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned h = pack_bf16x2(x0[i], x1[i]);
                const unsigned l = pack_bf16x2(x0[i] - __uint_as_float(h << 16), x1[i] - __uint_as_float(h & 0xffff0000u));
                hi[(n4 + i) * (P64_LD / 2) + 16 * b2 + kp] = h;
                lo[(n4 + i) * (P64_LD / 2) + 16 * b2 + kp] = l;
            }
        }
    };

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    float4 ra[2 * NKK], rn[2 * NKK], rb[2 * NB];
    if (nc > 0) {
        load_a(0, ra);
        load_b(0, rb);
        store_b(0, rb);
    }
    __syncthreads();
    for (int c = 0; c < nc; ++c) {
        const int cur = c & 1;
        const bool more = c + 1 < nc;
        if (more) {
            load_a((c + 1) * P64_KC, rn);
            load_b((c + 1) * P64_KC, rb);
        }
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            bf16x8 ah, al;
            split8(ra[2 * kk], ra[2 * kk + 1], ah, al);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int off = (32 * j + r) * P64_LD + 16 * kk + 8 * g;
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&Bs[cur][0][off]);
                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(&Bs[cur][1][off]);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[j], 0, 0, 0);
            }
        }
        if (more) store_b(cur ^ 1, rb);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2 * NKK; ++i) ra[i] = rn[i];
    }
    // C/D map of the 32x32 MFMA: col = lane & 31, row = (i & 3) + 8 (i >> 2) + 4 (lane >> 5)
    const bool atomic = p.flags & GEMM_ATOMIC;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = bm + 32 * wave + (i & 3) + 8 * (i >> 2) + 4 * g;
            if (m >= p.M) continue;
            float* cp = Cp + (long)m * p.ldc + 32 * j + r;
            if (atomic) atomicAdd(cp, acc[j][i]);
            else *cp = p.beta != 0.f ? acc[j][i] + *cp : acc[j][i];
        }
}

// The m-major slab form (A stored [K, M]: P^T dO, dS^T (q + u), dG^T q) with 64 rows per wave.  In x3_panel64_kernel<false> a load instruction covers 32
// consecutive m of two k rows - 128-byte pieces of 2000-byte rows, each straddling two cache lines that the neighbouring wave asks for again (TCC_REQ 2 x
// the slab's lines, 434 MB fetched for 256 MB).  Here lane l of a wave is row m0 + l for 64 rows: one load per k row, 256 contiguous bytes, three lines
// where the two 128-byte pieces took four.  The 32 x 32 x 16 MFMA wants lanes 32 - 63 to hold k + 8 ... k + 15 of rows 0 - 31, which sit in lanes 0 - 31 of
// other registers: v_permlane32_swap of the registers of k and k + 8 yields the fragment element of BOTH 32-row tiles in one instruction.
template <int DUMMY>
__global__ __launch_bounds__(NT) void x3_panel64_m64_kernel(const KParams p) {
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2][2][64 * 40];      // [stage][hi | lo][column n][k]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, g = lane >> 5;
    const int m0 = blockIdx.x * 256 + 64 * wave;
    const int z = blockIdx.y, z1 = z / p.nz2, z2 = z % p.nz2;
    const float* A = reinterpret_cast<const float*>(p.A) + z1 * p.sA1 + z2 * p.sA2;
    const float* B = reinterpret_cast<const float*>(p.B) + z1 * p.sB1 + z2 * p.sB2;
    float* Cp = reinterpret_cast<float*>(p.C) + z1 * p.sC1 + z2 * p.sC2;
    const float* ap = A + min(m0 + lane, p.M - 1);              // rows past the matrix: the last row's values, never stored
    const int nc = (p.K + 31) / 32;
    auto load_a = [&](int k0, float* d) {
#pragma unroll
        for (int k = 0; k < 32; ++k) d[k] = (k0 + k < p.K) ? ap[(long)(k0 + k) * p.lda] : 0.f;
    };
    const int kp = tid >> 4, n4 = (tid & 15) * 4;
    auto load_b = [&](int k0, float4* d) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int k = k0 + 2 * kp + s2;
            d[s2] = load4<float>(B + (long)k * p.ldb + n4, k < p.K ? 4 : 0, p.vecB);
        }
    };
    auto store_b = [&](int st, const float4* d) {
        unsigned* hi = reinterpret_cast<unsigned*>(Bs[st][0]);
        unsigned* lo = reinterpret_cast<unsigned*>(Bs[st][1]);
        const float x0[4] = {d[0].x, d[0].y, d[0].z, d[0].w}, x1[4] = {d[1].x, d[1].y, d[1].z, d[1].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned h = pack_bf16x2(x0[i], x1[i]);
            const unsigned l = pack_bf16x2(x0[i] - __uint_as_float(h << 16), x1[i] - __uint_as_float(h & 0xffff0000u));
            hi[(n4 + i) * 20 + kp] = h;
            lo[(n4 + i) * 20 + kp] = l;
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][j][i] = 0.f;
    float ra[32], rn[32];
    float4 rb[2];
    if (nc > 0) {
        load_a(0, ra);
        load_b(0, rb);
        store_b(0, rb);
    }
    __syncthreads();
    for (int c = 0; c < nc; ++c) {
        const int cur = c & 1;
        const bool more = c + 1 < nc;
        if (more) {
            load_a((c + 1) * 32, rn);
            load_b((c + 1) * 32, rb);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            float t0[8], t1[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_int(ra[16 * kk + i]), __float_as_int(ra[16 * kk + 8 + i]), false, false);
                t0[i] = __int_as_float(sw[0]);                  // lanes 0 - 31: rows 0 - 31 at k + i; lanes 32 - 63: rows 0 - 31 at k + 8 + i
                t1[i] = __int_as_float(sw[1]);                  // the same of rows 32 - 63
            }
            bf16x8 ah[2], al[2];
            split8(make_float4(t0[0], t0[1], t0[2], t0[3]), make_float4(t0[4], t0[5], t0[6], t0[7]), ah[0], al[0]);
            split8(make_float4(t1[0], t1[1], t1[2], t1[3]), make_float4(t1[4], t1[5], t1[6], t1[7]), ah[1], al[1]);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int off = (32 * j + r) * 40 + 16 * kk + 8 * g;
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&Bs[cur][0][off]);
                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(&Bs[cur][1][off]);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], bh, acc[t][j], 0, 0, 0);
                    acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[t], bh, acc[t][j], 0, 0, 0);
                    acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], bl, acc[t][j], 0, 0, 0);
                }
            }
        }
        if (more) store_b(cur ^ 1, rb);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 32; ++i) ra[i] = rn[i];
    }
    const bool atomic = p.flags & GEMM_ATOMIC;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = m0 + 32 * t + (i & 3) + 8 * (i >> 2) + 4 * g;
                if (m >= p.M) continue;
                float* cp = Cp + (long)m * p.ldc + 32 * j + r;
                if (atomic) atomicAdd(cp, acc[t][j][i]);
                else *cp = p.beta != 0.f ? acc[t][j][i] + *cp : acc[t][j][i];
            }
}

// The k-major slab form with LINE-ALIGNED strips.  A 128-byte strip of a row whose pitch is 2000 bytes straddles two cache lines, and in
// x3_panel64_kernel<true> the second half of each is gone from the L2 by the time the next strip asks for it: TCC_MISS = 2 x the slab's lines,
// 515 MB fetched for 256 MB (rocprofv3 --pmc, profiles/r05_bf16x3_attention_core_kernels.txt).  The line phase of a row start repeats every 8 rows
// when the pitch is a multiple of 4 floats, so a wave takes the 32 rows {r0 + 8 i} - ONE phase phi - and its strips are k in [32 c - phi, 32 c - phi + 32):
// every load instruction reads whole lines, every line once.  The four waves of a workgroup hold four of the eight phases of a 256-row range (two
// workgroups per range), each reading the B strip at its own offset: B sits in LDS as a ring of four 32-k strips (strip c + 1 being written while
// c - 1 and c are read; strip -1 = zeros for the k < 0 head of phase-shifted rows).
constexpr int PK_LD = 40;
__global__ __launch_bounds__(NT) void x3_panel64_kphase_kernel(const KParams p) {
    __shared__ __attribute__((aligned(16))) unsigned short Bs[4][2][64 * PK_LD];      // [ring slot][hi | lo][column n][k within the strip]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, g = lane >> 5;
    const int R0 = 256 * (blockIdx.x >> 1), rho = 4 * (blockIdx.x & 1) + wave;
    const int z = blockIdx.y, z1 = z / p.nz2, z2 = z % p.nz2;
    const float* A = reinterpret_cast<const float*>(p.A) + z1 * p.sA1 + z2 * p.sA2;
    const float* B = reinterpret_cast<const float*>(p.B) + z1 * p.sB1 + z2 * p.sB2;
    float* Cp = reinterpret_cast<float*>(p.C) + z1 * p.sC1 + z2 * p.sC2;
    const int row = R0 + rho + 8 * r;
    const float* ap = A + (long)min(row, p.M - 1) * p.lda;      // rows past the matrix: loaded from its last row (another phase: slower, not wrong), never stored
    const int phi = (int)((reinterpret_cast<uintptr_t>(A + (long)min(R0 + rho, p.M - 1) * p.lda) >> 2) & 31);      // floats between the line start and the wave's row starts
    const int nc = (p.K + 31 + 31) / 32;                        // strips c with 32 c - phi < K for the largest phase (31): the same count for every wave (barriers)

    auto load_a = [&](int c, float4* d) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int k = 32 * c - phi + 16 * kk + 8 * g;       // a multiple of 4 (the slab is 16-byte aligned, its pitch a multiple of 4 floats)
            d[2 * kk] = load4<float>(ap + k, k >= 0 ? p.K - k : 0, 1);
            d[2 * kk + 1] = load4<float>(ap + k + 4, k + 4 >= 0 ? p.K - k - 4 : 0, 1);
        }
    };
    const int kp = tid >> 4, n4 = (tid & 15) * 4;
    auto load_b = [&](int s, float4* d) {                        // strip s: rows 32 s + 2 kp, + 1 of four columns (rows >= K: zeros)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int k = 32 * s + 2 * kp + s2;
            d[s2] = load4<float>(B + (long)k * p.ldb + n4, k < p.K ? 4 : 0, p.vecB);
        }
    };
    auto store_b = [&](int slot, const float4* d) {
        unsigned* hi = reinterpret_cast<unsigned*>(Bs[slot][0]);
        unsigned* lo = reinterpret_cast<unsigned*>(Bs[slot][1]);
        const float x0[4] = {d[0].x, d[0].y, d[0].z, d[0].w}, x1[4] = {d[1].x, d[1].y, d[1].z, d[1].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned h = pack_bf16x2(x0[i], x1[i]);
            const unsigned l = pack_bf16x2(x0[i] - __uint_as_float(h << 16), x1[i] - __uint_as_float(h & 0xffff0000u));
            hi[(n4 + i) * (PK_LD / 2) + kp] = h;
            lo[(n4 + i) * (PK_LD / 2) + kp] = l;
        }
    };
    // four consecutive k (a multiple of 4: inside one strip) of column n from the ring
    auto ring4 = [&](int plane, int n, int k) -> uint2 {
        return *reinterpret_cast<const uint2*>(&Bs[(k >> 5) & 3][plane][n * PK_LD + (k & 31)]);      // k = -phi ... -1: strip -1 = slot 3, zeroed below
    };

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    float4 ra[4], rn[4], rb[2];
    for (int i = tid; i < 2 * 64 * PK_LD / 2; i += NT) reinterpret_cast<unsigned*>(Bs[3][0])[i] = 0u;      // both planes of slot 3 (contiguous)
    load_a(0, ra);
    load_b(0, rb);
    store_b(0, rb);
    __syncthreads();
    for (int c = 0; c < nc; ++c) {
        const bool more = c + 1 < nc;
        if (more) {
            load_a(c + 1, rn);
            load_b(c + 1, rb);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 ah, al;
            split8(ra[2 * kk], ra[2 * kk + 1], ah, al);
            const int k = 32 * c - phi + 16 * kk + 8 * g;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = 32 * j + r;
                const uint2 h0 = ring4(0, n, k), h1 = ring4(0, n, k + 4), l0 = ring4(1, n, k), l1 = ring4(1, n, k + 4);
                const u32x4 hv = {h0.x, h0.y, h1.x, h1.y}, lv = {l0.x, l0.y, l1.x, l1.y};
                const bf16x8 bh = __builtin_bit_cast(bf16x8, hv), bl = __builtin_bit_cast(bf16x8, lv);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[j], 0, 0, 0);
            }
        }
        if (more) store_b((c + 1) & 3, rb);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = rn[i];
    }
    // C/D map of the 32x32 MFMA: col = lane & 31, tile row = (i & 3) + 8 (i >> 2) + 4 (lane >> 5); tile row q is matrix row R0 + rho + 8 q
    const bool atomic = p.flags & GEMM_ATOMIC;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = R0 + rho + 8 * ((i & 3) + 8 * (i >> 2) + 4 * g);
            if (m >= p.M) continue;
            float* cp = Cp + (long)m * p.ldc + 32 * j + r;
            if (atomic) atomicAdd(cp, acc[j][i]);
            else *cp = p.beta != 0.f ? acc[j][i] + *cp : acc[j][i];
        }
}

constexpr int RN_CG = 256, RN_NP = RN_CG + 8;      // columns per pass through the LDS tile, its row pitch (4 RN_NP = 32 mod 64 banks: the two row groups of a store miss each other)

__global__ __launch_bounds__(NT, 4) void x3_rows_nt64_kernel(const KParams p) {
    // The 32 x 32 accumulators have lanes along COLUMNS: stored directly, every instruction leaves two 128-byte row pieces that straddle cache
    // lines (row pitch 2000 bytes).  The workgroup's 32 rows pass through an LDS tile instead and leave row by row in line-aligned pieces, 16 bytes
    // per lane where a row piece is 16-byte aligned (158 -> 145 us at 512 columns per pass).
    // 256 columns per pass (34 KB of LDS, four workgroups per CU): 135 us at B = 32, H = 8, L = 500, 192 us with beta = 1 (512 columns, two
    // workgroups per CU: 150 / 200; 128 columns: 155 / 231).  With 512-column passes the kernel took 110 us with the write-out skipped and 77 us
    // with the products skipped (tools/debug/bench_generic_gemm.py).  Measured and NOT faster: the operand rows in whole lines through a
    // wave-private LDS image instead of fragment-shaped loads (145), two column blocks of operand rows in flight (146), eight waves with every
    // load of a pass in flight and one contiguous run of 16-byte stores (167, spills); no LDS tile at all - two adjacent column blocks per wave, v_permlane32_swap of their
    // accumulators so that a store instruction is one row's 64 consecutive columns (146; 147 with two row tiles per wave, i.e. half the B reads).  SQ counters of the
    // tile form (tools/debug/pmc_rows_kernel.sh): 1364 VALU + 609 SALU + 40 load + 26 store + 90 LDS instructions per wave, a wave lives 31 us at 85 % of the
    // resident-wave capacity and waits on memory for 64 % of it.  The library's batched f32 product: 179 us.
    extern __shared__ __attribute__((aligned(16))) float rn_tile[];      // [32][RN_NP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, g = lane >> 5;
    // workgroups are dealt to the 8 XCDs round-robin, and every row block of a (batch, head) slab reads ALL of its B rows: numbered as launched,
    // each XCD's L2 fetches every B once (FETCH_SIZE: 288 MB = A + 8 x B at B = 32, L = 500).  Renumbered (p.kchunk = 1: the launcher's switch)
    // so that the row blocks of one slab run on ONE XCD, next to each other in time: same box, alternating, 200 / 204 -> 192 / 193 us with beta = 1,
    // 137 +- 2 either way without; the C2 step 113.86 / 113.92 -> 113.49 / 113.68 ms.
    int bx = blockIdx.x, z = blockIdx.y;
    if (p.kchunk == 1) {
        const unsigned total = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
        if ((total & 7) == 0) {
            const unsigned t = (lin & 7) * (total >> 3) + (lin >> 3);
            bx = t % gridDim.x;
            z = t / gridDim.x;
        }
    }
    const int bm = bx * 32;
    const int z1 = z / p.nz2, z2 = z % p.nz2;
    const float* A = reinterpret_cast<const float*>(p.A) + z1 * p.sA1 + z2 * p.sA2;
    const float* B = reinterpret_cast<const float*>(p.B) + z1 * p.sB1 + z2 * p.sB2;
    float* Cp = reinterpret_cast<float*>(p.C) + z1 * p.sC1 + z2 * p.sC2;
    const float* bias = (p.flags & GEMM_BIAS) ? p.bias + z1 * p.sBias1 + z2 * p.sBias2 : nullptr;
    bf16x8 ah[4], al[4];
    {
        const float4* ap = reinterpret_cast<const float4*>(A + (long)min(bm + r, p.M - 1) * p.lda + 8 * g);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) split8(ap[4 * kk], ap[4 * kk + 1], ah[kk], al[kk]);
    }
    auto load_b = [&](int j, float4* d) {
        const float4* bp = reinterpret_cast<const float4*>(B + (long)min(32 * j + r, p.N - 1) * p.ldb + 8 * g);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            d[2 * kk] = bp[4 * kk];
            d[2 * kk + 1] = bp[4 * kk + 1];
        }
    };
    float4 bc[8], bn[8];
    for (int cg = 0; cg < p.N; cg += RN_CG) {
        const int ncols = min(p.N - cg, RN_CG), ncb = (ncols + 31) >> 5, j0 = cg >> 5;
        if (wave < ncb) load_b(j0 + wave, bc);
        for (int j = wave; j < ncb; j += 4) {
            if (j + 4 < ncb) load_b(j0 + j + 4, bn);
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                bf16x8 bh, bl;
                split8(bc[2 * kk], bc[2 * kk + 1], bh, bl);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kk], bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[kk], bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kk], bl, acc, 0, 0, 0);
            }
            const int nl = 32 * j + r;
            if (nl < ncols) {
                const float bv = bias ? bias[cg + nl] : 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) rn_tile[((i & 3) + 8 * (i >> 2) + 4 * g) * RN_NP + nl] = acc[i] + bv;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) bc[i] = bn[i];
        }
        __syncthreads();
        for (int rr = wave; rr < 32 && bm + rr < p.M; rr += 4) {
            float* crow = Cp + (long)(bm + rr) * p.ldc + cg;
            const int mis = (int)((reinterpret_cast<uintptr_t>(crow) >> 2) & 31);      // floats past the 128-byte line the row piece starts in
            const float* trow = rn_tile + rr * RN_NP;
            // (beta = 1 - the content scores added onto the shifted position term: all of the row's reads in flight before its first store; a load
            // per store serialised on the aliasing stores and took 405 us against 135 us without the accumulation)
            if ((reinterpret_cast<uintptr_t>(crow) & 15) == 0 && (ncols & 3) == 0) {
                // 16-byte-aligned row piece (the pitch-L views): 16 bytes per lane, instruction boundaries still on 128-byte lines - three
                // instructions a row instead of nine
                const int mis4 = mis >> 2, n4 = ncols >> 2;
                float4 cv[RN_CG / 256 + 1];
#pragma unroll
                for (int it = 0; it < RN_CG / 256 + 1; ++it) {
                    const int c4 = lane - mis4 + 64 * it;
                    cv[it] = (p.beta != 0.f && c4 >= 0 && c4 < n4) ? reinterpret_cast<const float4*>(crow)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int it = 0; it < RN_CG / 256 + 1; ++it) {
                    const int c4 = lane - mis4 + 64 * it;
                    if (c4 < 0 || c4 >= n4) continue;
                    const float4 t = reinterpret_cast<const float4*>(trow)[c4];
                    reinterpret_cast<float4*>(crow)[c4] = make_float4(t.x + cv[it].x, t.y + cv[it].y, t.z + cv[it].z, t.w + cv[it].w);
                }
            } else if (p.beta != 0.f) {
                float cv[RN_CG / 64 + 1];
#pragma unroll
                for (int it = 0; it < RN_CG / 64 + 1; ++it) {
                    const int c = lane - mis + 64 * it;
                    cv[it] = (c >= 0 && c < ncols) ? crow[c] : 0.f;
                }
#pragma unroll
                for (int it = 0; it < RN_CG / 64 + 1; ++it) {
                    const int c = lane - mis + 64 * it;
                    if (c >= 0 && c < ncols) crow[c] = trow[c] + cv[it];
                }
            } else {
                for (int c = lane - mis; c < ncols; c += 64)
                    if (c >= 0) crow[c] = trow[c];
            }
        }
        if (cg + RN_CG < p.N) __syncthreads();
    }
}

static int g_x3_attn_kernels = 1;      // TTMI_X3_ATTN_KERNELS=0: the attention core of the bf16x3 mode back on the 128 x 128 kernel (A/B runs)

template <typename SA, typename SB, typename TC, bool AK, bool BKM, bool BF16C, bool X3 = false>
int launch_t(const KParams& p, dim3 grid, hipStream_t st) {
    const size_t lds = (X3 ? 8 : 4) * Cfg<BF16C>::TILE_BYTES;
    if constexpr (X3) {                                           // 80 KB of dynamic LDS: above the default limit
        static bool once = false;
        if (!once) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<SA, SB, TC, AK, BKM, BF16C, X3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
                ttmi_set_error("gemm: LDS attribute");
                return TTMI_EINVAL;
            }
            once = true;
        }
    }
    hipLaunchKernelGGL((gemm_kernel<SA, SB, TC, AK, BKM, BF16C, X3>), grid, dim3(NT), lds, st, p);
    TTMI_LAUNCH_CHECK("gemm_kernel");
    return TTMI_OK;
}

template <typename SA, typename SB, typename TC, bool BF16C, bool X3 = false>
int launch_layout(const KParams& p, dim3 grid, hipStream_t st) {
    const bool ak = p.flags & GEMM_A_KMAJOR, bk = p.flags & GEMM_B_KMAJOR;
    if (ak && bk) return launch_t<SA, SB, TC, true, true, BF16C, X3>(p, grid, st);
    if (ak && !bk) return launch_t<SA, SB, TC, true, false, BF16C, X3>(p, grid, st);
    if (!ak && !bk) return launch_t<SA, SB, TC, false, false, BF16C, X3>(p, grid, st);
    return launch_t<SA, SB, TC, false, true, BF16C, X3>(p, grid, st);
}

}  // namespace

static int g_pers_min_tiles = 512;   // f32 NT: the persistent 256 x 128 kernel from two full rounds of its tiles on, 64 x 64 tiles below (profiles/r04_f32_gemm_kernels.txt)
static int g_mid_min_tiles = 72;     // ... and the 64 x 64-tile kernel from this many of ITS tiles on (fewer: the 32 x 32-tile kernel, whose waves split K)
static int g_skinny_rows = 128;     // f32 products with at most this many rows go to gemm_skinny_f32_kernel (0 = never)
void ttmi_gemm_set_skinny_rows(int rows) { g_skinny_rows = rows < 0 ? 0 : rows; }

int ttmi_launch_gemm(const GemmDesc& d, hipStream_t st) {
    TTMI_REQUIRE(d.A && d.B && d.C, "gemm: null operand");
    TTMI_REQUIRE(d.M > 0 && d.N > 0 && d.K >= 0, "gemm: bad shape M=%d N=%d K=%d", d.M, d.N, d.K);
    TTMI_REQUIRE(d.nz1 > 0 && d.nz2 > 0 && d.splitk > 0, "gemm: bad batch/splitk");
    TTMI_REQUIRE(!(d.flags & GEMM_BIAS) || d.bias, "gemm: GEMM_BIAS without bias pointer");
    TTMI_REQUIRE(!(d.flags & GEMM_MASK_AUX) || d.aux, "gemm: GEMM_MASK_AUX without aux pointer");
    TTMI_REQUIRE(!(d.flags & GEMM_ATOMIC) || d.c_dtype == DT_F32, "gemm: atomic epilogue needs f32 C");
    TTMI_REQUIRE(d.splitk == 1 || (d.flags & GEMM_ATOMIC), "gemm: splitk>1 needs GEMM_ATOMIC");
    const bool bf16c = d.flags & GEMM_BF16_MFMA;
    TTMI_REQUIRE(bf16c || (d.a_dtype == DT_F32 && d.b_dtype == DT_F32), "gemm: f32 compute needs f32 operands");
    KParams p;
    p.A = d.A; p.B = d.B; p.C = d.C; p.bias = d.bias; p.aux = d.aux;
    p.M = d.M; p.N = d.N; p.K = d.K; p.lda = d.lda; p.ldb = d.ldb; p.ldc = d.ldc;
    p.nz2 = d.nz2;
    p.sA1 = d.sA1; p.sA2 = d.sA2; p.sB1 = d.sB1; p.sB2 = d.sB2; p.sC1 = d.sC1; p.sC2 = d.sC2;
    p.sBias1 = d.sBias1; p.sBias2 = d.sBias2;
    p.alpha = d.alpha; p.beta = d.beta; p.flags = d.flags; p.splitk = d.splitk; p.drop = d.drop;
    const int bk = bf16c ? 32 : 16;
    int kchunk = (d.K + d.splitk - 1) / d.splitk;
    kchunk = (kchunk + bk - 1) / bk * bk;
    p.kchunk = kchunk > 0 ? kchunk : bk;
    const size_t ea = d.a_dtype == DT_F32 ? 4 : 2, eb = d.b_dtype == DT_F32 ? 4 : 2;
    auto vec_ok = [](const void* ptr, size_t es, long ld, long s1, long s2) {
        const size_t q = 4;   // 4 elements per vector access
        return ((reinterpret_cast<uintptr_t>(ptr) % (q * es)) == 0) && (ld % q == 0) && (s1 % q == 0) && (s2 % q == 0);
    };
    p.vecA = vec_ok(d.A, ea, d.lda, d.sA1, d.sA2);
    p.vecB = vec_ok(d.B, eb, d.ldb, d.sB1, d.sB2);
    // exact-f32 NT products with plain epilogues leave this file when they are large enough:
    //  - >= 1024 rows and enough 256 x 128 tiles: the persistent LDS-DMA kernel with f32 operands (fp32 mode: encoder / joint forward; greedy decoding: the
    //    joint over a block of frames) - 1.3 - 2 x the register-staged 128 x 128 kernel below;
    //  - a few hundred to a few thousand rows (greedy decoding: the label encoder on alive x history rows): 64 x 64 tiles through LDS
    const int f32_mode = gemm_fast_f32_mode();
    if (!bf16c && f32_mode != 0 && (d.flags & GEMM_A_KMAJOR) && (d.flags & GEMM_B_KMAJOR) && !(d.flags & (GEMM_ATOMIC | GEMM_MASK_AUX)) && d.splitk == 1 &&
        d.nz1 * d.nz2 == 1 && d.c_dtype == DT_F32 && d.alpha == 1.f && (d.beta == 0.f || d.beta == 1.f)) {
        const long t9 = (long)cdiv(d.M, 256) * cdiv(d.N, 128), t64 = (long)cdiv(d.M, 64) * cdiv(d.N, 64);
        const bool pers_ok = f32_mode != 2 && gemm_nt_f32_ok(d.A, d.B, d.C, d.M, d.N, d.K, d.lda, d.ldb, d.ldc);
        const bool mid_ok = f32_mode != 3 && d.drop.p <= 0.f && gemm_nt_f32_mid_ok(d.A, d.B, d.C, d.M, d.N, d.K, d.lda, d.ldb, d.ldc);
        const bool want_pers = pers_ok && (f32_mode == 3 || !mid_ok || t9 >= g_pers_min_tiles);
        const bool want_mid = mid_ok && !want_pers && (f32_mode == 2 || t64 >= g_mid_min_tiles);
        if (want_pers) {
            NtEpilogue e;
            e.bias = (d.flags & GEMM_BIAS) ? d.bias : nullptr;
            e.addend = d.beta == 1.f ? static_cast<const float*>(d.C) : nullptr;
            e.relu = (d.flags & GEMM_RELU) ? 1 : 0;
            e.drop = d.drop;
            return gemm_nt_f32(static_cast<const float*>(d.A), static_cast<const float*>(d.B), static_cast<float*>(d.C), e, d.M, d.N, d.K, d.lda, d.ldb, d.ldc, st);
        }
        if (want_mid)
            return gemm_nt_f32_mid(static_cast<const float*>(d.A), static_cast<const float*>(d.B), static_cast<float*>(d.C), (d.flags & GEMM_BIAS) ? d.bias : nullptr,
                                   d.beta == 1.f, (d.flags & GEMM_RELU) ? 1 : 0, d.M, d.N, d.K, d.lda, d.ldb, d.ldc, st);
    }
    // ... and products of up to 2048 rows whose 128 x 128 tiling would leave more than half of the CUs without a workgroup: the batched greedy
    // decoder's label-encoder calls (B x history rows, 512 ... 1536 columns: 16 ... 48 such tiles, each walking K in 16-wide barrier steps
    // for ~100 us; 32 x 32 tiles with the reduction split over a workgroup's waves: ~10 us.  decode_batch at 8 utterances: 26 -> 63 utt/s)
    const long tiles128 = (long)cdiv(d.N, BN) * cdiv(d.M, BM) * d.nz1 * d.nz2;
    const bool few_tiles = d.M <= 2048 && tiles128 < 128;
    if (!bf16c && g_skinny_rows > 0 && (d.M <= g_skinny_rows || few_tiles) && (d.flags & GEMM_A_KMAJOR) && !(d.flags & GEMM_ATOMIC) && d.splitk == 1 &&
        d.c_dtype == DT_F32 && d.K > 0 && (long)d.nz1 * d.nz2 <= 65535) {
        dim3 sgrid(cdiv(d.N, SK_T), cdiv(d.M, SK_T), d.nz1 * d.nz2);
        if (d.flags & GEMM_B_KMAJOR) hipLaunchKernelGGL(gemm_skinny_f32_kernel<true>, sgrid, dim3(NT), 0, st, p);
        else hipLaunchKernelGGL(gemm_skinny_f32_kernel<false>, sgrid, dim3(NT), 0, st, p);
        TTMI_LAUNCH_CHECK("gemm_skinny_f32_kernel");
        return TTMI_OK;
    }
    dim3 grid(cdiv(d.N, BN), cdiv(d.M, BM), d.nz1 * d.nz2 * d.splitk);
    TTMI_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "gemm: grid too large (M tiles %u, batch %u)", grid.y, grid.z);
    const int key = (bf16c ? 8 : 0) | (d.a_dtype << 2) | (d.b_dtype << 1) | d.c_dtype;
    if (key == 0 && (d.flags & GEMM_BF16X3)) {
        static const bool env_once = [] {
            const char* e = getenv("TTMI_X3_ATTN_KERNELS");
            if (e && e[0] == '0') g_x3_attn_kernels = 0;
            return true;
        }();
        (void)env_once;
        const bool plain = d.alpha == 1.f && (d.beta == 0.f || d.beta == 1.f) && d.splitk == 1 && d.drop.p <= 0.f && (long)d.nz1 * d.nz2 <= 65535;
        const bool ak = d.flags & GEMM_A_KMAJOR, bkm = d.flags & GEMM_B_KMAJOR;
        // (a k-major slab that is not 16-byte aligned - the pitch-(L+1) view of dG - loads its fragments in single floats: 214 -> 178 us.  Tried: lanes along k,
        // one 256-byte load per row at any alignment, the 32 x 64 piece transposed into fragments through a wave-private LDS image: correct, 213 us at 200 registers)
        if (g_x3_attn_kernels && plain && d.N == 64 && !bkm && p.vecB && !(d.flags & ~(GEMM_BF16X3 | GEMM_A_KMAJOR | GEMM_ATOMIC)) &&
            (!(d.flags & GEMM_ATOMIC) || d.beta == 0.f)) {
            dim3 pg(cdiv(d.M, P64_BM), d.nz1 * d.nz2);
            static const int kphase_env = [] { const char* e = getenv("TTMI_X3_KPHASE"); return e ? atoi(e) : 1; }();
            if (ak && p.vecA && kphase_env) {
                hipLaunchKernelGGL(x3_panel64_kphase_kernel, dim3(2 * cdiv(d.M, 256), d.nz1 * d.nz2), dim3(NT), 0, st, p);
                TTMI_LAUNCH_CHECK("x3_panel64_kphase_kernel");
                return TTMI_OK;
            }
            // (64-wide strips were measured: k-major 120 -> 125 us, m-major 106 -> 97 us at three workgroups per CU instead of four; 32 stays)
            static const int m64_env = [] { const char* e = getenv("TTMI_X3_M64"); return e ? atoi(e) : 1; }();
            if (ak) hipLaunchKernelGGL((x3_panel64_kernel<true, 32>), pg, dim3(NT), 0, st, p);
            else if (m64_env) hipLaunchKernelGGL((x3_panel64_m64_kernel<0>), dim3(cdiv(d.M, 256), d.nz1 * d.nz2), dim3(NT), 0, st, p);
            else hipLaunchKernelGGL((x3_panel64_kernel<false, 32>), pg, dim3(NT), 0, st, p);
            TTMI_LAUNCH_CHECK("x3_panel64_kernel");
            return TTMI_OK;
        }
        if (g_x3_attn_kernels && plain && d.K == 64 && ak && bkm && p.vecA && p.vecB &&
            !(d.flags & ~(GEMM_BF16X3 | GEMM_A_KMAJOR | GEMM_B_KMAJOR | GEMM_BIAS))) {
            constexpr int rn_lds = 32 * RN_NP * 4;             // (the attribute call matters from 512-column passes on: 66.5 KB)
            static bool rn_attr = false;
            if (!rn_attr) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(x3_rows_nt64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, rn_lds) != hipSuccess) {
                    ttmi_set_error("gemm: LDS attribute (x3_rows_nt64_kernel)");
                    return TTMI_EINVAL;
                }
                rn_attr = true;
            }
            static const int rn_xcd = [] { const char* e = getenv("TTMI_X3_XCD"); return e ? atoi(e) : 1; }();
            p.kchunk = rn_xcd;                                 // (K = 64 is fixed in this kernel: the field carries the numbering switch)
            hipLaunchKernelGGL(x3_rows_nt64_kernel, dim3(cdiv(d.M, 32), d.nz1 * d.nz2), dim3(NT), rn_lds, st, p);
            TTMI_LAUNCH_CHECK("x3_rows_nt64_kernel");
            return TTMI_OK;
        }
        // bf16x3 parity mode: the 128 x 128 kernel with every f32 tile staged as bf16 hi + lo and three bf16 MFMA terms per fragment pair (K-steps of 32:
        // the split-K chunking above assumed 16-wide steps - re-round it)
        int kc = (d.K + d.splitk - 1) / d.splitk;
        kc = (kc + 31) / 32 * 32;
        p.kchunk = kc > 0 ? kc : 32;
        return launch_layout<float, float, float, true, true>(p, grid, st);
    }
    switch (key) {
        case 0: return launch_layout<float, float, float, false>(p, grid, st);
        case 8: return launch_layout<float, float, float, true>(p, grid, st);
        case 8 | 1: return launch_layout<float, float, bf16_t, true>(p, grid, st);
        case 8 | 4: return launch_layout<bf16_t, float, float, true>(p, grid, st);
        case 8 | 2: return launch_layout<float, bf16_t, float, true>(p, grid, st);
        case 8 | 6: return launch_layout<bf16_t, bf16_t, float, true>(p, grid, st);
        case 8 | 7: return launch_layout<bf16_t, bf16_t, bf16_t, true>(p, grid, st);
        case 8 | 3: return launch_layout<float, bf16_t, bf16_t, true>(p, grid, st);
        case 8 | 5: return launch_layout<bf16_t, float, bf16_t, true>(p, grid, st);
        default: break;
    }
    ttmi_set_error("gemm: unsupported dtype combination a=%d b=%d c=%d bf16=%d", d.a_dtype, d.b_dtype, d.c_dtype, (int)bf16c);
    return TTMI_EINVAL;
}

extern "C" {
// Test/bring-up entry point for the generic GEMM (dtype codes: 0 = f32, 1 = bf16; flags = GemmFlags).
int ttmi_gemm(const void* A, const void* B, void* C, const float* bias, const float* aux, int a_dtype, int b_dtype,
              int c_dtype, int M, int N, int K, long lda, long ldb, long ldc, int nz1, int nz2, long sA1, long sA2,
              long sB1, long sB2, long sC1, long sC2, float alpha, float beta, int flags, int splitk, void* stream) {
    GemmDesc d;
    d.A = A; d.B = B; d.C = C; d.bias = bias; d.aux = aux;
    d.a_dtype = a_dtype; d.b_dtype = b_dtype; d.c_dtype = c_dtype;
    d.M = M; d.N = N; d.K = K; d.lda = lda; d.ldb = ldb; d.ldc = ldc;
    d.nz1 = nz1; d.nz2 = nz2; d.sA1 = sA1; d.sA2 = sA2; d.sB1 = sB1; d.sB2 = sB2; d.sC1 = sC1; d.sC2 = sC2;
    d.alpha = alpha; d.beta = beta; d.flags = flags; d.splitk = splitk;
    return ttmi_launch_gemm(d, static_cast<hipStream_t>(stream));
}
}
