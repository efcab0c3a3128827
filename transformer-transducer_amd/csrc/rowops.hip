// HBM-bound row kernels of the Transformer-Transducer path (LayerNorm, masked softmax over the
// relative-position score view, bias/column reductions, relative-position table gather/scatter,
// embedding, joint tanh).  One wave (64 lanes) per row, 16-byte accesses where alignment allows,
// wave reductions by __shfl_xor; cross-row reductions finish with float atomics (agent scope).
// Reference arithmetic: tt/transformer.py:52-58,148-175, tt/decoder.py:26,39, tt/model.py:33-37.
#include "rowops.h"
#include <type_traits>
#ifndef LN_BWD_GRID
#define LN_BWD_GRID 384
#endif
#include <algorithm>

namespace {

constexpr int WPB = 4;   // waves per block for the wave-per-row kernels

__device__ __forceinline__ long wave_row() { return (long)blockIdx.x * WPB + (threadIdx.x >> 6); }

// ------------------------------------------------------------------ LayerNorm
__global__ __launch_bounds__(WPB * 64) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                          const float* __restrict__ g, const float* __restrict__ b, long rows,
                                                          int d, float eps, float* __restrict__ s_out, float* __restrict__ y,
                                                          float* __restrict__ mean_o, float* __restrict__ rstd_o, int vec,
                                                          bf16_t* __restrict__ y16, DropSpec rdrop, DropSpec odrop) {
    rdrop = drop_live(rdrop);
    odrop = drop_live(odrop);
    const long r = wave_row();
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + r * d;
    const float* rr = res ? res + r * d : nullptr;
    float* sr = s_out ? s_out + r * d : nullptr;
    float* yr = y ? y + r * d : nullptr;
    bf16_t* y16r = y16 ? y16 + r * d : nullptr;
    float sum = 0.f;
    const unsigned long long base = (unsigned long long)r * d;
    auto load4 = [&](int i) -> float4 {
        float4 v = *reinterpret_cast<const float4*>(xr + i);
        if (rr) {
            const float4 w = *reinterpret_cast<const float4*>(rr + i);
            v.x += w.x * drop_mult(rdrop, base + i);
            v.y += w.y * drop_mult(rdrop, base + i + 1);
            v.z += w.z * drop_mult(rdrop, base + i + 2);
            v.w += w.w * drop_mult(rdrop, base + i + 3);
        }
        return v;
    };
    auto val = [&](int i) -> float { return xr[i] + (rr ? rr[i] * drop_mult(rdrop, base + i) : 0.f); };
    if (vec) {
        for (int i = lane * 4; i < d; i += 256) {
            const float4 v = load4(i);
            if (sr) *reinterpret_cast<float4*>(sr + i) = v;
            sum += (v.x + v.y) + (v.z + v.w);
        }
    } else {
        for (int i = lane; i < d; i += 64) {
            const float v = val(i);
            if (sr) sr[i] = v;
            sum += v;
        }
    }
    const float mean = wave_sum(sum) / d;
    float sq = 0.f;
    if (vec) {
        for (int i = lane * 4; i < d; i += 256) {
            const float4 v = load4(i);
            const float a = v.x - mean, bq = v.y - mean, c = v.z - mean, e = v.w - mean;
            sq += (a * a + bq * bq) + (c * c + e * e);
        }
    } else {
        for (int i = lane; i < d; i += 64) {
            const float a = val(i) - mean;
            sq += a * a;
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / d + eps);
    if (lane == 0) {
        if (mean_o) mean_o[r] = mean;
        if (rstd_o) rstd_o[r] = rstd;
    }
    if (vec) {
        for (int i = lane * 4; i < d; i += 256) {
            const float4 v = load4(i);
            const float4 gg = *reinterpret_cast<const float4*>(g + i);
            const float4 bb = *reinterpret_cast<const float4*>(b + i);
            float4 o;
            o.x = ((v.x - mean) * rstd * gg.x + bb.x) * drop_mult(odrop, base + i);
            o.y = ((v.y - mean) * rstd * gg.y + bb.y) * drop_mult(odrop, base + i + 1);
            o.z = ((v.z - mean) * rstd * gg.z + bb.z) * drop_mult(odrop, base + i + 2);
            o.w = ((v.w - mean) * rstd * gg.w + bb.w) * drop_mult(odrop, base + i + 3);
            if (yr) *reinterpret_cast<float4*>(yr + i) = o;
            if (y16r) {
                uint2 w;
                w.x = pack_bf16x2(o.x, o.y);
                w.y = pack_bf16x2(o.z, o.w);
                *reinterpret_cast<uint2*>(y16r + i) = w;
            }
        }
    } else {
        for (int i = lane; i < d; i += 64) {
            const float o = ((val(i) - mean) * rstd * g[i] + b[i]) * drop_mult(odrop, base + i);
            if (yr) yr[i] = o;
            if (y16r) y16r[i] = f32_to_bf16(o);
        }
    }
}

// d <= 512, d % 4 == 0 (every LayerNorm of the path): the wave keeps its row in registers - ONE pass over x and res, one dropout word per
// four elements, mean and variance from the registers (two-pass formula, as above), every output written once.  The three-pass kernel above
// re-read its operands from the cache and re-drew the residual branch's dropout mask in every pass: at p > 0 it was bound by the mask's
// integer multiplies, not by HBM (33.6 us for 128 MB at C2).
// Optional second norm (g2 != nullptr): h16 = bf16(LN(y; g2, b2)) with its statistics - the FFN's pre-norm of the value this call produces
// (tt/transformer.py:54-58 applies it to the attention sub-layer's output), so that y is not read back for it.
template <int KV>
__global__ __launch_bounds__(WPB * 64) void ln_fwd_row_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                              const float* __restrict__ g, const float* __restrict__ b, long rows,
                                                              int d, float eps, float* __restrict__ s_out, float* __restrict__ y,
                                                              float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                              bf16_t* __restrict__ y16, DropSpec rdrop, DropSpec odrop,
                                                              const float* __restrict__ g2, const float* __restrict__ b2,
                                                              bf16_t* __restrict__ h16, float* __restrict__ mean2_o,
                                                              float* __restrict__ rstd2_o) {
    rdrop = drop_live(rdrop);
    odrop = drop_live(odrop);
    const long r = wave_row();
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const unsigned long long base = (unsigned long long)r * d;
    float v[KV][4];
    float sum = 0.f;
    // all of the row's loads first, unconditionally (a chunk past d reads the row's first columns and is dropped): under `if (c0 < d)` every chunk's loads sat in a
    // block of their own with s_waitcnt vmcnt(0) behind them - two 16-byte loads in flight per lane instead of four (round 6, the ISA)
    float4 xa[KV], xw[KV];
#pragma unroll
    for (int k = 0; k < KV; ++k) xa[k] = *reinterpret_cast<const float4*>(x + base + (k * 256 + lane * 4 < d ? k * 256 + lane * 4 : 0));
    if (res) {
#pragma unroll
        for (int k = 0; k < KV; ++k) xw[k] = *reinterpret_cast<const float4*>(res + base + (k * 256 + lane * 4 < d ? k * 256 + lane * 4 : 0));
    }
#pragma unroll
    for (int k = 0; k < KV; ++k) {
        const int c0 = k * 256 + lane * 4;
        if (c0 < d) {
            const float4 a = xa[k];
            v[k][0] = a.x; v[k][1] = a.y; v[k][2] = a.z; v[k][3] = a.w;
            if (res) {
                const float4 w = xw[k];
                float m[4];
                drop_mult4(rdrop, base + c0, m);
                v[k][0] += w.x * m[0]; v[k][1] += w.y * m[1]; v[k][2] += w.z * m[2]; v[k][3] += w.w * m[3];
            }
            if (s_out) *reinterpret_cast<float4*>(s_out + base + c0) = make_float4(v[k][0], v[k][1], v[k][2], v[k][3]);
            sum += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
        } else {
            v[k][0] = v[k][1] = v[k][2] = v[k][3] = 0.f;
        }
    }
    const float invd = 1.f / d;
    const float mean = wave_sum(sum) * invd;
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < KV; ++k)
        if (k * 256 + lane * 4 < d) {
            const float a = v[k][0] - mean, bq = v[k][1] - mean, c = v[k][2] - mean, e = v[k][3] - mean;
            sq += (a * a + bq * bq) + (c * c + e * e);
        }
    const float rstd = rsqrtf(wave_sum(sq) * invd + eps);
    if (lane == 0) {
        if (mean_o) mean_o[r] = mean;
        if (rstd_o) rstd_o[r] = rstd;
    }
    float sum2 = 0.f;
#pragma unroll
    for (int k = 0; k < KV; ++k) {
        const int c0 = k * 256 + lane * 4;
        if (c0 < d) {
            const float4 gg = *reinterpret_cast<const float4*>(g + c0);
            const float4 bb = *reinterpret_cast<const float4*>(b + c0);
            float m[4];
            drop_mult4(odrop, base + c0, m);
            v[k][0] = ((v[k][0] - mean) * rstd * gg.x + bb.x) * m[0];
            v[k][1] = ((v[k][1] - mean) * rstd * gg.y + bb.y) * m[1];
            v[k][2] = ((v[k][2] - mean) * rstd * gg.z + bb.z) * m[2];
            v[k][3] = ((v[k][3] - mean) * rstd * gg.w + bb.w) * m[3];
            if (y) *reinterpret_cast<float4*>(y + base + c0) = make_float4(v[k][0], v[k][1], v[k][2], v[k][3]);
            if (y16) {
                uint2 w;
                w.x = pack_bf16x2(v[k][0], v[k][1]);
                w.y = pack_bf16x2(v[k][2], v[k][3]);
                *reinterpret_cast<uint2*>(y16 + base + c0) = w;
            }
            sum2 += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
        }
    }
    if (!g2) return;                                                   // kernel argument: wave-uniform
    const float mean2 = wave_sum(sum2) * invd;
    float sq2 = 0.f;
#pragma unroll
    for (int k = 0; k < KV; ++k)
        if (k * 256 + lane * 4 < d) {
            const float a = v[k][0] - mean2, bq = v[k][1] - mean2, c = v[k][2] - mean2, e = v[k][3] - mean2;
            sq2 += (a * a + bq * bq) + (c * c + e * e);
        }
    const float rstd2 = rsqrtf(wave_sum(sq2) * invd + eps);
    if (lane == 0) {
        mean2_o[r] = mean2;
        rstd2_o[r] = rstd2;
    }
#pragma unroll
    for (int k = 0; k < KV; ++k) {
        const int c0 = k * 256 + lane * 4;
        if (c0 < d) {
            const float4 gg = *reinterpret_cast<const float4*>(g2 + c0);
            const float4 bb = *reinterpret_cast<const float4*>(b2 + c0);
            uint2 w;
            w.x = pack_bf16x2((v[k][0] - mean2) * rstd2 * gg.x + bb.x, (v[k][1] - mean2) * rstd2 * gg.y + bb.y);
            w.y = pack_bf16x2((v[k][2] - mean2) * rstd2 * gg.z + bb.z, (v[k][3] - mean2) * rstd2 * gg.w + bb.w);
            *reinterpret_cast<uint2*>(h16 + base + c0) = w;
        }
    }
}

__global__ __launch_bounds__(WPB * 64) void ln_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ s,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const float* __restrict__ g, const float* __restrict__ dadd,
                                                             long rows, int d, float* __restrict__ dx, DropSpec ddrop) {
    ddrop = drop_live(ddrop);
    const long r = wave_row();
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const float mu = mean[r], rs = rstd[r];
    const float* dyr = dy + r * d;
    const float* sr = s + r * d;
    float m1 = 0.f, m2 = 0.f;
    const unsigned long long base = (unsigned long long)r * d;
    for (int i = lane; i < d; i += 64) {
        const float dxh = dyr[i] * drop_mult(ddrop, base + i) * g[i];
        const float xh = (sr[i] - mu) * rs;
        m1 += dxh;
        m2 += dxh * xh;
    }
    m1 = wave_sum(m1) / d;
    m2 = wave_sum(m2) / d;
    for (int i = lane; i < d; i += 64) {
        const float dxh = dyr[i] * drop_mult(ddrop, base + i) * g[i];
        const float xh = (sr[i] - mu) * rs;
        float v = rs * (dxh - m1 - xh * m2);
        if (dadd) v += dadd[r * d + i];
        dx[r * d + i] = v;
    }
}

constexpr int LNP_ROWS = 64;
__global__ __launch_bounds__(256) void ln_bwd_params_kernel(const float* __restrict__ dy, const float* __restrict__ s,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            long rows, int d, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, DropSpec ddrop) {
    ddrop = drop_live(ddrop);
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= d) return;
    const long r0 = (long)blockIdx.y * LNP_ROWS;
    const long r1 = r0 + LNP_ROWS < rows ? r0 + LNP_ROWS : rows;
    float ag = 0.f, ab = 0.f;
    for (long r = r0; r < r1; ++r) {
        const float v = dy[r * d + c] * drop_mult(ddrop, (unsigned long long)r * d + c);
        ag += v * (s[r * d + c] - mean[r]) * rstd[r];
        ab += v;
    }
    atomicAdd(dgamma + c, ag);
    atomicAdd(dbeta + c, ab);
}

// dx, dgamma and dbeta in ONE pass over dy and s (d <= 512, d % 4 == 0): a wave keeps its row's dy*drop and x-hat in registers
// between the row statistics and the dx formula, and carries per-column partial sums for gamma / beta over the rows it walks;
// they are combined over the block's 4 waves in LDS and added with one atomic per column and block.
template <int KV>
__global__ __launch_bounds__(256) void ln_bwd_fused_kernel(const float* __restrict__ dy, const float* __restrict__ s,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ g, const float* __restrict__ dadd, long rows,
                                                           int d, float* __restrict__ dx, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, DropSpec ddrop, bf16_t* __restrict__ dx16,
                                                           DropSpec xdrop, float* __restrict__ dx16_colsum) {
    ddrop = drop_live(ddrop);
    xdrop = drop_live(xdrop);
    // optional second output for the bf16 pipeline: dx16 = bf16(dx * dropout(xdrop)) (the masked gradient the following
    // GEMMs consume) and its column sums (the bias gradient of the Linear in front of the dropout)
    __shared__ float red[3][4][KV * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float gam[KV][4], ag[KV][4], ab[KV][4], ac[KV][4];
#pragma unroll
    for (int k = 0; k < KV; ++k) {
        const int c0 = k * 256 + lane * 4;
        const float4 gv = c0 < d ? *reinterpret_cast<const float4*>(g + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
        gam[k][0] = gv.x; gam[k][1] = gv.y; gam[k][2] = gv.z; gam[k][3] = gv.w;
#pragma unroll
        for (int c = 0; c < 4; ++c) ag[k][c] = ab[k][c] = ac[k][c] = 0.f;
    }
    const float invd = 1.f / d;
    // the row loop is software-pipelined by hand: the loads of the wave's NEXT row are in flight while the current row goes through
    // its two wave reductions and its stores (8 waves per CU at this grid cannot hide a row's latency chain on their own)
    struct RowIn {
        float4 a[KV], b[KV], e[KV];
        float mu, rs;
    };
    const long stride = (long)gridDim.x * 4;
    auto fetch = [&](long r, RowIn& in) {
        const unsigned long long base = (unsigned long long)r * d;
        in.mu = mean[r]; in.rs = rstd[r];
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            const int c0 = k * 256 + lane * 4;
            if (c0 < d) {
                in.a[k] = *reinterpret_cast<const float4*>(dy + base + c0);
                in.b[k] = *reinterpret_cast<const float4*>(s + base + c0);
                if (dadd) in.e[k] = *reinterpret_cast<const float4*>(dadd + base + c0);
            }
        }
    };
    long r = (long)blockIdx.x * 4 + wave;
    RowIn cur, nxt;
    if (r < rows) fetch(r, cur);
    for (; r < rows; r += stride) {
        if (r + stride < rows) fetch(r + stride, nxt);
        const float mu = cur.mu, rs = cur.rs;
        const unsigned long long base = (unsigned long long)r * d;
        float v[KV][4], xh[KV][4];
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            const int c0 = k * 256 + lane * 4;
            if (c0 < d) {
                const float4 a = cur.a[k], b = cur.b[k];
                const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
                float dm[4];
                drop_mult4(ddrop, base + c0, dm);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    v[k][c] = av[c] * dm[c];
                    xh[k][c] = (bv[c] - mu) * rs;
                    const float dxh = v[k][c] * gam[k][c];
                    m1 += dxh;
                    m2 += dxh * xh[k][c];
                    ag[k][c] += v[k][c] * xh[k][c];
                    ab[k][c] += v[k][c];
                }
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[k][c] = xh[k][c] = 0.f;
            }
        }
        m1 = wave_sum(m1) * invd;
        m2 = wave_sum(m2) * invd;
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            const int c0 = k * 256 + lane * 4;
            if (c0 < d) {
                float o[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = rs * (v[k][c] * gam[k][c] - m1 - xh[k][c] * m2);
                if (dadd) {
                    const float4 e = cur.e[k];
                    o[0] += e.x; o[1] += e.y; o[2] += e.z; o[3] += e.w;
                }
                *reinterpret_cast<float4*>(dx + base + c0) = make_float4(o[0], o[1], o[2], o[3]);
                if (dx16) {
                    float m[4];
                    drop_mult4(xdrop, base + c0, m);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        m[c] *= o[c];
                        ac[k][c] += m[c];
                    }
                    uint2 w;
                    w.x = pack_bf16x2(m[0], m[1]);
                    w.y = pack_bf16x2(m[2], m[3]);
                    *reinterpret_cast<uint2*>(dx16 + base + c0) = w;
                }
            }
        }
        cur = nxt;
    }
#pragma unroll
    for (int k = 0; k < KV; ++k)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            red[0][wave][k * 256 + lane * 4 + c] = ag[k][c];
            red[1][wave][k * 256 + lane * 4 + c] = ab[k][c];
            red[2][wave][k * 256 + lane * 4 + c] = ac[k][c];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += 256) {
        atomicAdd(dgamma + c, red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
        atomicAdd(dbeta + c, red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
        if (dx16_colsum) atomicAdd(dx16_colsum + c, red[2][0][c] + red[2][1][c] + red[2][2][c] + red[2][3][c]);
    }
}

// The two LayerNorm backward passes that meet at a layer's middle (tt/transformer.py:54-58 and :172-175), in ONE pass over the rows:
//     dy  = LN1'(dh; y, mean1, rstd1, g1) + dres1          the FFN's pre-norm, plus the FFN's residual branch  (= gradient of the layer's y)
//     dx  = LN2'(dy; s, mean2, rstd2, g2)                  the attention sub-layer's post-norm             (= residual gradient of the layer's x)
//     dx16 = bf16(dx * dropout(xdrop))                     the o_net GEMMs' operand
// dy never exists in memory (one 32 MB write and read per layer at C2), both norms' gamma / beta sums ride along as in ln_bwd_fused_kernel.
template <int KV>
__global__ __launch_bounds__(256) void ln_bwd_pair_kernel(const float* __restrict__ dh, const float* __restrict__ y,
                                                          const float* __restrict__ mean1, const float* __restrict__ rstd1,
                                                          const float* __restrict__ g1, const float* __restrict__ dres1,
                                                          const float* __restrict__ s, const float* __restrict__ mean2,
                                                          const float* __restrict__ rstd2, const float* __restrict__ g2, long rows, int d,
                                                          float* __restrict__ dx, float* __restrict__ dgamma1, float* __restrict__ dbeta1,
                                                          float* __restrict__ dgamma2, float* __restrict__ dbeta2,
                                                          bf16_t* __restrict__ dx16, DropSpec xdrop) {
    xdrop = drop_live(xdrop);
    __shared__ float red[4][4][KV * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float gam1[KV][4], gam2[KV][4], ag1[KV][4], ab1[KV][4], ag2[KV][4], ab2[KV][4];
#pragma unroll
    for (int k = 0; k < KV; ++k) {
        const int c0 = k * 256 + lane * 4;
        const float4 a = c0 < d ? *reinterpret_cast<const float4*>(g1 + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 b = c0 < d ? *reinterpret_cast<const float4*>(g2 + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
        gam1[k][0] = a.x; gam1[k][1] = a.y; gam1[k][2] = a.z; gam1[k][3] = a.w;
        gam2[k][0] = b.x; gam2[k][1] = b.y; gam2[k][2] = b.z; gam2[k][3] = b.w;
#pragma unroll
        for (int c = 0; c < 4; ++c) ag1[k][c] = ab1[k][c] = ag2[k][c] = ab2[k][c] = 0.f;
    }
    const float invd = 1.f / d;
    struct RowIn {
        float4 a[KV], b[KV], e[KV], f[KV];
        float mu1, rs1, mu2, rs2;
    };
    const long stride = (long)gridDim.x * 4;
    auto fetch = [&](long r, RowIn& in) {
        const unsigned long long base = (unsigned long long)r * d;
        in.mu1 = mean1[r]; in.rs1 = rstd1[r];
        in.mu2 = mean2[r]; in.rs2 = rstd2[r];
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            const int c0 = k * 256 + lane * 4;
            if (c0 < d) {
                in.a[k] = *reinterpret_cast<const float4*>(dh + base + c0);
                in.b[k] = *reinterpret_cast<const float4*>(y + base + c0);
                in.e[k] = *reinterpret_cast<const float4*>(dres1 + base + c0);
                in.f[k] = *reinterpret_cast<const float4*>(s + base + c0);
            }
        }
    };
    long r = (long)blockIdx.x * 4 + wave;
    RowIn cur, nxt;
    if (r < rows) fetch(r, cur);
    for (; r < rows; r += stride) {
        if (r + stride < rows) fetch(r + stride, nxt);
        const unsigned long long base = (unsigned long long)r * d;
        float v[KV][4], xh[KV][4];
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            if (k * 256 + lane * 4 < d) {
                const float4 a = cur.a[k], b = cur.b[k];
                const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    v[k][c] = av[c];
                    xh[k][c] = (bv[c] - cur.mu1) * cur.rs1;
                    const float dxh = v[k][c] * gam1[k][c];
                    m1 += dxh;
                    m2 += dxh * xh[k][c];
                    ag1[k][c] += v[k][c] * xh[k][c];
                    ab1[k][c] += v[k][c];
                }
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[k][c] = xh[k][c] = 0.f;
            }
        }
        m1 = wave_sum(m1) * invd;
        m2 = wave_sum(m2) * invd;
        float n1 = 0.f, n2 = 0.f;
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            if (k * 256 + lane * 4 < d) {
                const float4 e = cur.e[k], f = cur.f[k];
                const float ev[4] = {e.x, e.y, e.z, e.w}, fv[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    v[k][c] = cur.rs1 * (v[k][c] * gam1[k][c] - m1 - xh[k][c] * m2) + ev[c];        // dy
                    xh[k][c] = (fv[c] - cur.mu2) * cur.rs2;
                    const float dxh = v[k][c] * gam2[k][c];
                    n1 += dxh;
                    n2 += dxh * xh[k][c];
                    ag2[k][c] += v[k][c] * xh[k][c];
                    ab2[k][c] += v[k][c];
                }
            }
        }
        n1 = wave_sum(n1) * invd;
        n2 = wave_sum(n2) * invd;
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            const int c0 = k * 256 + lane * 4;
            if (c0 < d) {
                float o[4], m[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = cur.rs2 * (v[k][c] * gam2[k][c] - n1 - xh[k][c] * n2);
                *reinterpret_cast<float4*>(dx + base + c0) = make_float4(o[0], o[1], o[2], o[3]);
                drop_mult4(xdrop, base + c0, m);
                uint2 w;
                w.x = pack_bf16x2(o[0] * m[0], o[1] * m[1]);
                w.y = pack_bf16x2(o[2] * m[2], o[3] * m[3]);
                *reinterpret_cast<uint2*>(dx16 + base + c0) = w;
            }
        }
        cur = nxt;
    }
#pragma unroll
    for (int k = 0; k < KV; ++k)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            red[0][wave][k * 256 + lane * 4 + c] = ag1[k][c];
            red[1][wave][k * 256 + lane * 4 + c] = ab1[k][c];
            red[2][wave][k * 256 + lane * 4 + c] = ag2[k][c];
            red[3][wave][k * 256 + lane * 4 + c] = ab2[k][c];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += 256) {
        atomicAdd(dgamma1 + c, red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
        atomicAdd(dbeta1 + c, red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
        atomicAdd(dgamma2 + c, red[2][0][c] + red[2][1][c] + red[2][2][c] + red[2][3][c]);
        atomicAdd(dbeta2 + c, red[3][0][c] + red[3][1][c] + red[3][2][c] + red[3][3][c]);
    }
}

// ------------------------------------------------------------------ masked softmax on the score view
__device__ __forceinline__ bool masked_at(const MaskDesc& m, int b, int i, int j) {
    switch (m.kind) {
        case MASK_CAUSAL: return j > i;
        case MASK_BAND: return (j > i + m.right) || (j < i - m.left);
        case MASK_TENSOR: return m.ptr[(long)b * m.sb + (long)i * m.si + j] != 0;
        case MASK_INTERVAL: {
            const int* r = reinterpret_cast<const int*>(m.ptr) + (long)b * m.sb + 2 * i;
            return j < r[0] || j > r[1];
        }
        default: return false;
    }
}

__global__ __launch_bounds__(WPB * 64) void softmax_fwd_kernel(float* __restrict__ S, long nrows, int nh, int L, long ld,
                                                               long slab, float scale, const MaskDesc m) {
    const long r = wave_row();
    if (r >= nrows) return;
    const int lane = threadIdx.x & 63;
    const int i = (int)(r % L);
    const long sl = r / L;
    const int b = (int)(sl / nh);
    float* row = S + sl * slab + (long)i * ld;
    float mx = -INFINITY;
    for (int j = lane; j < L; j += 64) {
        const float v = masked_at(m, b, i, j) ? -INFINITY : row[j] * scale;
        mx = fmaxf(mx, v);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < L; j += 64) {
        const float v = masked_at(m, b, i, j) ? -INFINITY : row[j] * scale;
        sum += __expf(v - mx);
    }
    const float inv = 1.f / wave_sum(sum);
    for (int j = lane; j < L; j += 64) {
        const float v = masked_at(m, b, i, j) ? -INFINITY : row[j] * scale;
        row[j] = __expf(v - mx) * inv;
    }
}

// rows of up to 512 columns (every C2 / C4 attention): the row stays in registers - one read, one write, one mask test and one exp per element where
// the kernel above walks the row three times (147 -> 84 us at B = 32, H = 8, L = 500, the fp32 / bf16x3 modes' attention)
__global__ __launch_bounds__(WPB * 64) void softmax_fwd_reg_kernel(float* __restrict__ S, long nrows, int nh, int L, long ld, long slab, float scale,
                                                                   const MaskDesc m) {
    const long r = wave_row();
    if (r >= nrows) return;
    const int lane = threadIdx.x & 63;
    const int i = (int)(r % L);
    const long sl = r / L;
    const int b = (int)(sl / nh);
    float* row = S + sl * slab + (long)i * ld;
    float v[8];
    float mx = -INFINITY;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int j = lane + 64 * q;
        v[q] = (j < L && !masked_at(m, b, i, j)) ? row[j] * scale : -INFINITY;
        mx = fmaxf(mx, v[q]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        v[q] = __expf(v[q] - mx);
        sum += v[q];
    }
    const float inv = 1.f / wave_sum(sum);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int j = lane + 64 * q;
        if (j < L) row[j] = v[q] * inv;
    }
}

__global__ __launch_bounds__(WPB * 64) void softmax_bwd_kernel(float* __restrict__ dP, const float* __restrict__ P, long nrows,
                                                               int L, long ld, long slab, float scale) {
    const long r = wave_row();
    if (r >= nrows) return;
    const int lane = threadIdx.x & 63;
    const int i = (int)(r % L);
    const long sl = r / L;
    float* drow = dP + sl * slab + (long)i * ld;
    const float* prow = P + sl * slab + (long)i * ld;
    float dot = 0.f;
    for (int j = lane; j < L; j += 64) dot += drow[j] * prow[j];
    dot = wave_sum(dot);
    for (int j = lane; j < L; j += 64) drow[j] = prow[j] * (drow[j] - dot) * scale;
}

// ------------------------------------------------------------------ small elementwise / reductions
__global__ void add_row_bias_kernel(const float* __restrict__ in, long ldi, const float* __restrict__ bias, long rows, int cols,
                                    float* __restrict__ out, long ldo) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * cols) return;
    const long r = idx / cols;
    const int c = (int)(idx % cols);
    out[r * ldo + c] = in[r * ldi + c] + bias[c];
}

__global__ void add_row_bias_bf16_kernel(const bf16_t* __restrict__ in, long ldi, const float* __restrict__ bias, long rows,
                                         int cols, bf16_t* __restrict__ out, long ldo) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * cols) return;
    const long r = idx / cols;
    const int c = (int)(idx % cols);
    out[r * ldo + c] = f32_to_bf16(bf16_to_f32(in[r * ldi + c]) + bias[c]);
}

constexpr int CS_ROWS = 128;
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ in, long ld, long rows, int cols, int nz2,
                                                     long si1, long si2, long so1, long so2, float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const int z = blockIdx.z, z1 = z / nz2, z2 = z % nz2;
    const float* p = in + z1 * si1 + z2 * si2;
    const long r0 = (long)blockIdx.y * CS_ROWS;
    const long r1 = r0 + CS_ROWS < rows ? r0 + CS_ROWS : rows;
    float a = 0.f;
    for (long r = r0; r < r1; ++r) a += p[r * ld + c];
    atomicAdd(out + z1 * so1 + z2 * so2 + c, a);
}

// vector form (cols % 4 == 0, 16-byte aligned rows): thread = 4 columns x every 4th row of a 64-row slab, four loads in flight,
// the block's 4 row-lanes combined in LDS, one atomic per column and block (the scalar kernel above ran at 0.85 TB/s)
__global__ __launch_bounds__(256) void colsum_vec_kernel(const float* __restrict__ in, long ld, long rows, int cols,
                                                         float* __restrict__ out) {
    __shared__ float red[4][256];
    const int cg = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 256 + cg * 4;
    const long r0 = (long)blockIdx.y * 64;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (c0 < cols) {
#pragma unroll 4
        for (int k = 0; k < 16; ++k) {
            const long r = r0 + rl + 4 * k;
            if (r < rows) {
                const float4 v = *reinterpret_cast<const float4*>(in + r * ld + c0);
                a[0] += v.x; a[1] += v.y; a[2] += v.z; a[3] += v.w;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) red[rl][cg * 4 + c] = a[c];
    __syncthreads();
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < cols) atomicAdd(out + c, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ void relpos_gather_kernel(const float* __restrict__ r_emb, const float* __restrict__ r_bias, int K, int L, int H,
                                     int Dh, float* __restrict__ E, float* __restrict__ cT, bf16_t* __restrict__ E16) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = (long)L * H * Dh;
    if (idx < n) {
        const int p = (int)(idx / (H * Dh));
        const int e = max(0, p + K - L);
        const float v = r_emb[(long)e * H * Dh + idx % (H * Dh)];
        E[idx] = v;
        if (E16) E16[idx] = f32_to_bf16(v);                    // the bf16 copy the fused attention kernels read, from the same pass
    }
    if (idx < (long)L * H) {
        const int h = (int)(idx / L), p = (int)(idx % L);
        const int e = max(0, p + K - L);
        cT[idx] = r_bias[(long)e * H + h];
    }
}

__global__ void relpos_scatter_kernel(const float* __restrict__ dE, const float* __restrict__ dcT, int K, int L, int H, int Dh,
                                      float* __restrict__ g_emb, float* __restrict__ g_bias, int folded) {
    // rows p >= L - K of the effective table map one to one onto table rows e = p + K - L; the L - K rows below (sequences longer than the
    // table) all fold onto row 0.  folded: relpos_fold_row0_kernel sums those in chunks (as atomics from here they serialise on one row:
    // 0.19 ms per audio layer at L = 2000, K = 410); a few dozen such rows (C2: 90) are cheaper as atomics than as a second launch
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = (long)L * H * Dh;
    if (idx < n) {
        const int p = (int)(idx / (H * Dh));
        const int e = p + K - L;
        if (e >= 0 || !folded) atomicAdd(g_emb + (long)max(e, 0) * H * Dh + idx % (H * Dh), dE[idx]);
    }
    if (idx < (long)L * H) {
        const int h = (int)(idx / L), p = (int)(idx % L);
        const int e = p + K - L;
        if (e >= 0 || !folded) atomicAdd(g_bias + (long)max(e, 0) * H + h, dcT[idx]);
    }
}

constexpr int FOLD_ROWS = 32;
__global__ __launch_bounds__(256) void relpos_fold_row0_kernel(const float* __restrict__ dE, const float* __restrict__ dcT, int nclamp, int L,
                                                               int H, int Dh, float* __restrict__ g_emb, float* __restrict__ g_bias) {
    const int c = blockIdx.x * 256 + threadIdx.x;                 // column of the [L, H*Dh] table gradient
    const int p0 = blockIdx.y * FOLD_ROWS, p1 = min(p0 + FOLD_ROWS, nclamp);
    const int HD = H * Dh;
    if (c < HD) {
        float acc = 0.f;
        for (int p = p0; p < p1; ++p) acc += dE[(long)p * HD + c];
        atomicAdd(g_emb + c, acc);
    }
    if (blockIdx.x == 0 && threadIdx.x < H) {
        float acc = 0.f;
        for (int p = p0; p < p1; ++p) acc += dcT[(long)threadIdx.x * L + p];
        atomicAdd(g_bias + threadIdx.x, acc);
    }
}

__global__ void embed_fwd_kernel(const long* __restrict__ tok, const float* __restrict__ W, long n, int d, int V,
                                 float* __restrict__ out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * d) return;
    const long r = idx / d;
    const long t = tok[r];
    // an id outside [0, V) is a caller error (nn.Embedding raises): it must not pass silently as some other row, so it reads as NaN
    out[idx] = (t < 0 || t >= V) ? __builtin_nanf("") : W[t * d + idx % d];
}

__global__ void embed_bwd_kernel(const long* __restrict__ tok, const float* __restrict__ dout, long n, int d, int V, int pad,
                                 float* __restrict__ gW) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * d) return;
    const long r = idx / d;
    const long t = tok[r];
    if (t == pad || t < 0 || t >= V) return;
    atomicAdd(gW + t * d + idx % d, dout[idx]);
}

// ------------------------------------------------------------------ joint: H = tanh(PE[b,t] + PD[b,u] + bias)
template <typename TH>
__global__ __launch_bounds__(256) void joint_tanh_fwd_kernel(const float* __restrict__ PE, const float* __restrict__ PD,
                                                             const float* __restrict__ bias, int T, int U1, int J,
                                                             TH* __restrict__ H) {
    const long bt = blockIdx.x;            // (b, t)
    const int b = (int)(bt / T);
    const float* pe = PE + bt * J;
    for (int j = threadIdx.x; j < J; j += 256) {
        const float e = pe[j] + bias[j];
        for (int u = 0; u < U1; ++u) {
            const float v = tanhf(e + PD[((long)b * U1 + u) * J + j]);
            const long o = (bt * U1 + u) * J + j;
            if constexpr (sizeof(TH) == 4) H[o] = v;
            else H[o] = f32_to_bf16(v);
        }
    }
}

// bf16 pipeline (J % 4 == 0): 4 columns per thread, 8-byte stores; tanh(x) = 1 - 2 / (1 + e^(2x)) on v_exp_f32 / v_rcp_f32 (relative
// error ~1e-6, three orders below the bf16 rounding of H) - the libm tanhf made this kernel instruction-bound at 1.7 TB/s
// (round 5: v_exp_f32 / v_rcp_f32 themselves - `__expf` + `__frcp_rn` compiled to the IEEE division sequence, two v_div_scale, v_div_fmas,
// v_div_fixup and four fmas around the v_rcp: 17 vector instructions per element where 5 do; both hardware ops are good to 1 ulp)
__device__ __forceinline__ float fast_tanh(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);      // e^(2x): inf for large x -> 1 - 0; 0 for very negative x -> 1 - 2
    return fmaf(-2.f, __builtin_amdgcn_rcpf(1.f + e), 1.f);
}
__global__ __launch_bounds__(256) void joint_tanh_fwd_bf16x4_kernel(const float* __restrict__ PE, const float* __restrict__ PD,
                                                                    const float* __restrict__ bias, int T, int U1, int J,
                                                                    bf16_t* __restrict__ H) {
    const long bt = blockIdx.x;
    const int b = (int)(bt / T);
    for (int j = threadIdx.x * 4; j < J; j += 1024) {
        const float4 pe = *reinterpret_cast<const float4*>(PE + bt * J + j), bi = *reinterpret_cast<const float4*>(bias + j);
        const float e0 = pe.x + bi.x, e1 = pe.y + bi.y, e2 = pe.z + bi.z, e3 = pe.w + bi.w;
        const float* pd = PD + (long)b * U1 * J + j;
        bf16_t* h = H + bt * U1 * J + j;
#pragma unroll 3
        for (int u = 0; u < U1; ++u) {
            const float4 d = *reinterpret_cast<const float4*>(pd + (long)u * J);
            uint2 w;
            w.x = pack_bf16x2(fast_tanh(e0 + d.x), fast_tanh(e1 + d.y));
            w.y = pack_bf16x2(fast_tanh(e2 + d.z), fast_tanh(e3 + d.w));
            *reinterpret_cast<uint2*>(h + (long)u * J) = w;
        }
    }
}

// J in {256, 512, 1024, 2048}: 8 columns per thread (16-byte stores), the block's 256 / (J/8) thread groups take every groups-th u.
// EMIS (exp-domain loss form): the two logits per lattice row that the loss reads - blank and next label - also leave in f32,
// emis[row] = (h16 . Wp16[blank] + bp[blank], h16 . Wp16[y] + bp[y]) with the SAME bf16 operands the projection GEMM multiplies (so they
// agree with its f32 accumulators up to summation order) instead of being read back from the bf16-rounded exp store.  The kernel is
// bound by its 2 bytes per element of output; the dot products ride in its idle VALU slots, the label rows of Wp16 come from L2.
struct JointEmis {
    const bf16_t* Wp16 = nullptr;   // [V, J] bf16, the projection's B operand
    const float* Wp32 = nullptr;    // [V, J] f32, the master weight the bf16 copy was rounded from
    const float* bp = nullptr;      // [V]
    const int* labels = nullptr;    // [B, U1 - 1]
    float* out = nullptr;           // [B * T * U1, 4]: blank / label logit from the GEMM's operands, blank / label logit from f32 operands
    int V = 0, blank = 0;
};
__device__ __forceinline__ void unpack_bf16x8(const uint4& w, float* f) {
    f[0] = __uint_as_float(w.x << 16); f[1] = __uint_as_float(w.x & 0xffff0000u);
    f[2] = __uint_as_float(w.y << 16); f[3] = __uint_as_float(w.y & 0xffff0000u);
    f[4] = __uint_as_float(w.z << 16); f[5] = __uint_as_float(w.z & 0xffff0000u);
    f[6] = __uint_as_float(w.w << 16); f[7] = __uint_as_float(w.w & 0xffff0000u);
}
// sum over the lanes of a row of the joint block by DPP moves only (no LDS crossbar: __shfl_xor is a ds_bpermute): after the two quad
// permutes and the two mirrors every lane holds its 16-lane row's sum; row_bcast:15 (rows 1, 3) and row_bcast:31 (rows 2, 3) carry the
// totals up, so lane 31 of each half holds the sum of its 32 lanes after the first and lane 63 the sum of all 64 after the second
template <bool FULL64>
__device__ __forceinline__ float dpp_row_sum(float v) {
    auto mv = [](float x, auto ctrl, auto rmask) {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, decltype(rmask)::value, 0xF, false));
    };
    v += mv(v, std::integral_constant<int, 0xB1>(), std::integral_constant<int, 0xF>());      // quad_perm [1,0,3,2]
    v += mv(v, std::integral_constant<int, 0x4E>(), std::integral_constant<int, 0xF>());      // quad_perm [2,3,0,1]
    v += mv(v, std::integral_constant<int, 0x141>(), std::integral_constant<int, 0xF>());     // row_half_mirror
    v += mv(v, std::integral_constant<int, 0x140>(), std::integral_constant<int, 0xF>());     // row_mirror
    v += mv(v, std::integral_constant<int, 0x142>(), std::integral_constant<int, 0xA>());     // row_bcast:15 into rows 1 and 3
    if constexpr (FULL64) v += mv(v, std::integral_constant<int, 0x143>(), std::integral_constant<int, 0xC>());   // row_bcast:31 into rows 2 and 3
    return v;
}
// FOUR sums at once (the EMIS dot products of one lattice row): a two-step transpose inside every quad leaves lane l with value l & 3
// summed over its quad (9 instructions), two row rotates and the row / half swaps finish it: EVERY lane holds value l & 3 summed over the
// 64 lanes (FULL64; else over its half of 32) - 13 cross-lane instructions where four dpp_row_sum chains took 24
// DPP moves + 24 adds (round 5: the kernel was bound by its vector instructions, not by its stores)
template <bool FULL64>
__device__ __forceinline__ float dpp_row_sum4(float a0, float a1, float a2, float a3, int lane) {
    auto mv = [](float x, auto ctrl, auto rmask) {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, decltype(rmask)::value, 0xF, false));
    };
    const bool p = lane & 1, q = lane & 2;
    // pairs: the even lane ends with values {0, 2}, the odd one with {1, 3}
    const float b0 = (p ? a1 : a0) + mv(p ? a0 : a1, std::integral_constant<int, 0xB1>(), std::integral_constant<int, 0xF>());      // quad_perm [1,0,3,2]
    const float b1 = (p ? a3 : a2) + mv(p ? a2 : a3, std::integral_constant<int, 0xB1>(), std::integral_constant<int, 0xF>());
    // pairs of pairs: lane l of the quad ends with value l
    float v = (q ? b1 : b0) + mv(q ? b0 : b1, std::integral_constant<int, 0x4E>(), std::integral_constant<int, 0xF>());              // quad_perm [2,3,0,1]
    v += mv(v, std::integral_constant<int, 0x124>(), std::integral_constant<int, 0xF>());     // row_ror:4
    v += mv(v, std::integral_constant<int, 0x128>(), std::integral_constant<int, 0xF>());     // row_ror:8: every lane holds its row's sum of value l & 3
    // across the rows of 16 the lane-in-row (= value index) must be kept: gfx950's row / half swaps (row_bcast would hand every lane of the
    // next row lane 15's value).  v_permlane16_swap D, S: rows 1, 3 of D <-> rows 0, 2 of S; with D = S = v, D + S = rows (0 + 1) and (2 + 3)
    auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_int(v), __float_as_int(v), false, false);
    v = __int_as_float(s16[0]) + __int_as_float(s16[1]);
    if constexpr (FULL64) {                                                                    // v_permlane32_swap: upper half of D <-> lower half of S
        auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
        v = __int_as_float(s32[0]) + __int_as_float(s32[1]);
    }
    return v;                                                                                  // every lane: value l & 3 summed over the 64 (32) lanes
}
constexpr int JT_TT_DEFAULT = 4, JT_TT = JT_TT_DEFAULT;             // frames per block: the label-encoder rows PD[b, u, :] (and, with EMIS, the rows of Wp16 of b's labels) are
                                     // read once per u and used for four frames - these L2 reads, not the 2 bytes per element of output, held the kernel at 3.2 TB/s
template <bool EMIS, int JT_TT = JT_TT_DEFAULT>
__global__ __launch_bounds__(256) void joint_tanh_fwd_bf16x8_kernel(const float* __restrict__ PE, const float* __restrict__ PD,
                                                                    const float* __restrict__ bias, int T, int U1, int J,
                                                                    bf16_t* __restrict__ H, JointEmis em) {
    extern __shared__ float part[];        // EMIS: [JT_TT][U1][waves per row][4] partial dot products
    const int tblocks = (T + JT_TT - 1) / JT_TT;
    const int b = blockIdx.x / tblocks, t0 = (blockIdx.x % tblocks) * JT_TT;
    const int nt = min(JT_TT, T - t0);
    const long bt0 = (long)b * T + t0;
    const int tpr = J >> 3, grp = threadIdx.x / tpr, ngrp = 256 / tpr;
    const int tin = threadIdx.x - grp * tpr;
    const int j = tin * 8;
    float e[JT_TT][8];
    {
        const float4 b0 = *reinterpret_cast<const float4*>(bias + j), b1 = *reinterpret_cast<const float4*>(bias + j + 4);
#pragma unroll
        for (int tt = 0; tt < JT_TT; ++tt) {
            const long r = bt0 + (tt < nt ? tt : 0);
            const float4 p0 = *reinterpret_cast<const float4*>(PE + r * J + j), p1 = *reinterpret_cast<const float4*>(PE + r * J + j + 4);
            e[tt][0] = p0.x + b0.x; e[tt][1] = p0.y + b0.y; e[tt][2] = p0.z + b0.z; e[tt][3] = p0.w + b0.w;
            e[tt][4] = p1.x + b1.x; e[tt][5] = p1.y + b1.y; e[tt][6] = p1.z + b1.z; e[tt][7] = p1.w + b1.w;
        }
    }
    const float* pd = PD + (long)b * U1 * J + j;
    float wb[8], wb32[8];
    const int nw = tpr >= 64 ? tpr >> 6 : 1;            // waves that share one row
    if constexpr (EMIS) {
        unpack_bf16x8(*reinterpret_cast<const uint4*>(em.Wp16 + (long)em.blank * J + j), wb);
        const float4 a0 = *reinterpret_cast<const float4*>(em.Wp32 + (long)em.blank * J + j), a1 = *reinterpret_cast<const float4*>(em.Wp32 + (long)em.blank * J + j + 4);
        wb32[0] = a0.x; wb32[1] = a0.y; wb32[2] = a0.z; wb32[3] = a0.w; wb32[4] = a1.x; wb32[5] = a1.y; wb32[6] = a1.z; wb32[7] = a1.w;
    }
#pragma unroll 2
    for (int u = grp; u < U1; u += ngrp) {
        const float4 d0 = *reinterpret_cast<const float4*>(pd + (long)u * J), d1 = *reinterpret_cast<const float4*>(pd + (long)u * J + 4);
        float wl[8], wl32[8];
        if constexpr (EMIS) {
            int y = em.blank;
            if (u < U1 - 1) {
                y = em.labels[(long)b * (U1 - 1) + u];
                y = y < 0 ? 0 : (y >= em.V ? em.V - 1 : y);
            }
            unpack_bf16x8(*reinterpret_cast<const uint4*>(em.Wp16 + (long)y * J + j), wl);
            const float4 a0 = *reinterpret_cast<const float4*>(em.Wp32 + (long)y * J + j), a1 = *reinterpret_cast<const float4*>(em.Wp32 + (long)y * J + j + 4);
            wl32[0] = a0.x; wl32[1] = a0.y; wl32[2] = a0.z; wl32[3] = a0.w; wl32[4] = a1.x; wl32[5] = a1.y; wl32[6] = a1.z; wl32[7] = a1.w;
        }
#pragma unroll
        for (int tt = 0; tt < JT_TT; ++tt) {
            if (tt < nt) {                                  // (block-uniform)
                const float th[8] = {fast_tanh(e[tt][0] + d0.x), fast_tanh(e[tt][1] + d0.y), fast_tanh(e[tt][2] + d0.z), fast_tanh(e[tt][3] + d0.w),
                                     fast_tanh(e[tt][4] + d1.x), fast_tanh(e[tt][5] + d1.y), fast_tanh(e[tt][6] + d1.z), fast_tanh(e[tt][7] + d1.w)};
                uint4 w;
                w.x = pack_bf16x2(th[0], th[1]);
                w.y = pack_bf16x2(th[2], th[3]);
                w.z = pack_bf16x2(th[4], th[5]);
                w.w = pack_bf16x2(th[6], th[7]);
                *reinterpret_cast<uint4*>(H + ((bt0 + tt) * U1 + u) * J + j) = w;
                if constexpr (EMIS) {
                    // two logits twice: from the operands the GEMM multiplies (bf16 h, bf16 weights: what P and its row sum contain) and from the
                    // unrounded h with the f32 master weights (what the emission log-probs should be made of)
                    float hv[8];
                    unpack_bf16x8(w, hv);
                    float db = 0.f, dl = 0.f, db32 = 0.f, dl32 = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        db = fmaf(hv[i], wb[i], db); dl = fmaf(hv[i], wl[i], dl);
                        db32 = fmaf(th[i], wb32[i], db32); dl32 = fmaf(th[i], wl32[i], dl32);
                    }
                    // sum over the lanes of this wave that hold the row (all 64, or an aligned group of tpr = 32): EVERY lane of the group ends with the sum
                    // of value index lane & 3; the group's first four lanes store them
                    const float sv = tpr >= 64 ? dpp_row_sum4<true>(db, dl, db32, dl32, tin) : dpp_row_sum4<false>(db, dl, db32, dl32, tin);
                    if ((tin & (tpr >= 64 ? 63 : 31)) < 4)
                        part[(((long)tt * U1 + u) * nw + (tin >> 6)) * 4 + (tin & 3)] = sv;
                }
            }
        }
    }
    if constexpr (EMIS) {
        __syncthreads();
        for (int i = threadIdx.x; i < nt * U1; i += 256) {
            const int tt = i / U1, u = i - tt * U1;
            float sb = 0.f, sl = 0.f, sb32 = 0.f, sl32 = 0.f;
            for (int w = 0; w < nw; ++w) {                                              // fixed order
                const float4 q = *reinterpret_cast<const float4*>(part + ((long)i * nw + w) * 4);
                sb += q.x; sl += q.y; sb32 += q.z; sl32 += q.w;
            }
            int y = em.blank;
            if (u < U1 - 1) {
                y = em.labels[(long)b * (U1 - 1) + u];
                y = y < 0 ? 0 : (y >= em.V ? em.V - 1 : y);
            }
            const float bb = em.bp[em.blank], by = em.bp[y];
            *reinterpret_cast<float4*>(em.out + ((bt0 + tt) * U1 + u) * 4) = make_float4(sb + bb, sl + by, sb32 + bb, sl32 + by);
        }
    }
}

// any J: one wave per lattice row reads H16 back (shapes the 8-column kernel does not take)
__global__ __launch_bounds__(256) void joint_emis_kernel(const bf16_t* __restrict__ H, long rows, int T, int U1, int J, JointEmis em) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const int u = (int)(row % U1);
    const int b = (int)(row / U1 / T);
    int y = em.blank;
    if (u < U1 - 1) {
        y = em.labels[(long)b * (U1 - 1) + u];
        y = y < 0 ? 0 : (y >= em.V ? em.V - 1 : y);
    }
    const bf16_t* h = H + row * J;
    const bf16_t *w0 = em.Wp16 + (long)em.blank * J, *w1 = em.Wp16 + (long)y * J;
    const float *f0 = em.Wp32 + (long)em.blank * J, *f1 = em.Wp32 + (long)y * J;
    float db = 0.f, dl = 0.f, db32 = 0.f, dl32 = 0.f;
    for (int j = lane; j < J; j += 64) {
        const float hv = bf16_to_f32(h[j]);                 // (only the rounded h exists here: the second pair corrects the weights' rounding alone)
        db = fmaf(hv, bf16_to_f32(w0[j]), db);
        dl = fmaf(hv, bf16_to_f32(w1[j]), dl);
        db32 = fmaf(hv, f0[j], db32);
        dl32 = fmaf(hv, f1[j], dl32);
    }
    db = wave_sum(db);
    dl = wave_sum(dl);
    db32 = wave_sum(db32);
    dl32 = wave_sum(dl32);
    if (lane == 0) {
        const float bb = em.bp[em.blank], by = em.bp[y];
        *reinterpret_cast<float4*>(em.out + row * 4) = make_float4(db + bb, dl + by, db32 + bb, dl32 + by);
    }
}

constexpr int JT_TC = 16;
template <typename TH>
__global__ __launch_bounds__(256) void joint_tanh_bwd_kernel(const TH* __restrict__ dH, const TH* __restrict__ H, int T, int U1,
                                                             int J, float* __restrict__ dPE, float* __restrict__ dPD) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= J) return;
    const int t0 = blockIdx.y * JT_TC;
    const int b = blockIdx.z;
    float accE[JT_TC];
#pragma unroll
    for (int i = 0; i < JT_TC; ++i) accE[i] = 0.f;
    for (int u = 0; u < U1; ++u) {
        float accD = 0.f;
        // (all 32 loads unconditional, frames past T clamped and dropped: under `if (t < T)` each pair waited alone - see joint_sum_bwd_bf16x4_part_kernel)
        TH hv[JT_TC], gv[JT_TC];
#pragma unroll
        for (int tt = 0; tt < JT_TC; ++tt) {
            const long o = (((long)b * T + min(t0 + tt, T - 1)) * U1 + u) * J + j;
            hv[tt] = H[o];
            gv[tt] = dH[o];
        }
#pragma unroll
        for (int tt = 0; tt < JT_TC; ++tt) {
            float h, g;
            if constexpr (sizeof(TH) == 4) { h = hv[tt]; g = gv[tt]; }
            else { h = bf16_to_f32(hv[tt]); g = bf16_to_f32(gv[tt]); }
            const float v = t0 + tt < T ? g * (1.f - h * h) : 0.f;
            accE[tt] += v;
            accD += v;
        }
        atomicAdd(dPD + ((long)b * U1 + u) * J + j, accD);
    }
#pragma unroll
    for (int tt = 0; tt < JT_TC; ++tt)
        if (t0 + tt < T) dPE[((long)b * T + t0 + tt) * J + j] = accE[tt];
}

// bf16x3 (round 5): H never exists in f32.  It leaves the forward kernel as the two-block row [hi | lo] (hi = bf16(h), lo = bf16(h - hi); pitch
// 2 Jp bf16, columns [J, Jp) of both blocks zero) - the A operand of the three-term projection as that GEMM reads it (the hi block twice: NtEpilogue::K_lo)
// and the planes of the weight gradient - and the backward forms 1 - (hi + lo)^2.  Saves the split pass over H in the forward and the one in the
// backward (2 x 6.6 GB at C2).  tanh through v_exp_f32 / v_rcp_f32: absolute error ~1e-7, below the 2^-17 of the split itself.
__global__ __launch_bounds__(256) void joint_tanh_fwd_x3_kernel(const float* __restrict__ PE, const float* __restrict__ PD,
                                                                const float* __restrict__ bias, int T, int U1, int J, int Jp,
                                                                bf16_t* __restrict__ H3) {
    const long bt = blockIdx.x;
    const int b = (int)(bt / T);
    for (int j = threadIdx.x * 4; j < Jp; j += 1024) {
        bf16_t* h = H3 + bt * U1 * 2 * Jp + j;
        if (j >= J) {                                   // pad columns (J % 4 == 0: a thread's four columns are all real or all pad)
            for (int u = 0; u < U1; ++u) {
                bf16_t* row = h + (long)u * 2 * Jp;
                *reinterpret_cast<uint2*>(row) = make_uint2(0u, 0u);
                *reinterpret_cast<uint2*>(row + Jp) = make_uint2(0u, 0u);
            }
            continue;
        }
        const float4 pe = *reinterpret_cast<const float4*>(PE + bt * J + j), bi = *reinterpret_cast<const float4*>(bias + j);
        const float e0 = pe.x + bi.x, e1 = pe.y + bi.y, e2 = pe.z + bi.z, e3 = pe.w + bi.w;
        const float* pd = PD + (long)b * U1 * J + j;
#pragma unroll 3
        for (int u = 0; u < U1; ++u) {
            const float4 d = *reinterpret_cast<const float4*>(pd + (long)u * J);
            const float t0 = fast_tanh(e0 + d.x), t1 = fast_tanh(e1 + d.y), t2 = fast_tanh(e2 + d.z), t3 = fast_tanh(e3 + d.w);
            uint2 hi, lo;
            hi.x = pack_bf16x2(t0, t1);
            hi.y = pack_bf16x2(t2, t3);
            lo.x = pack_bf16x2(t0 - __uint_as_float(hi.x << 16), t1 - __uint_as_float(hi.x & 0xffff0000u));
            lo.y = pack_bf16x2(t2 - __uint_as_float(hi.y << 16), t3 - __uint_as_float(hi.y & 0xffff0000u));
            bf16_t* row = h + (long)u * 2 * Jp;
            *reinterpret_cast<uint2*>(row) = hi;
            *reinterpret_cast<uint2*>(row + Jp) = lo;
        }
    }
}

// dpre = dH (1 - H^2) with H = hi + lo from the two-block rows; dPE[b,t,:] = sum_u dpre, dPD[b,u,:] += sum_t dpre.  Four columns per thread (16-byte
// loads of dH, 8-byte loads of the two H blocks), the column sums of a wave re-dealt through LDS for lane-contiguous atomics - the structure of
// joint_sum_bwd_bf16x4_kernel below (the one-column form of the f32 kernel took 1.80 ms at C2)
constexpr int JT_TCX = 16;
__global__ __launch_bounds__(256) void joint_tanh_bwd_x3_kernel(const float* __restrict__ dH, const bf16_t* __restrict__ H3, int T, int U1, int J,
                                                                int Jp, float* __restrict__ dPE, float* __restrict__ dPD) {
    __shared__ float xch[4 * 256];
    const int j = (blockIdx.x * 256 + threadIdx.x) * 4;
    const bool act = j < J;                            // (whole waves stay alive: the exchange is per wave)
    const int jc = act ? j : 0;
    const int t0 = blockIdx.y * JT_TCX;
    const int b = blockIdx.z;
    float accE[JT_TCX][4];
#pragma unroll
    for (int i = 0; i < JT_TCX; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) accE[i][c] = 0.f;
    for (int u = 0; u < U1; ++u) {
        float accD[4] = {0.f, 0.f, 0.f, 0.f};
        // (loads unconditional in two batches of eight frames, frames past T / columns past J clamped and dropped: under `if (t < T && act)` every frame's three
        // loads waited alone - see joint_sum_bwd_bf16x4_part_kernel)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float4 g8[JT_TCX / 2];
            uint2 hi8[JT_TCX / 2], lo8[JT_TCX / 2];
#pragma unroll
            for (int k = 0; k < JT_TCX / 2; ++k) {
                const long row = ((long)b * T + min(t0 + half * (JT_TCX / 2) + k, T - 1)) * U1 + u;
                g8[k] = *reinterpret_cast<const float4*>(dH + row * J + jc);
                hi8[k] = *reinterpret_cast<const uint2*>(H3 + row * 2 * Jp + jc);
                lo8[k] = *reinterpret_cast<const uint2*>(H3 + row * 2 * Jp + Jp + jc);
            }
#pragma unroll
            for (int k = 0; k < JT_TCX / 2; ++k) {
                const int tt = half * (JT_TCX / 2) + k;
                const uint2 hi = hi8[k], lo = lo8[k];
                const float h[4] = {__uint_as_float(hi.x << 16) + __uint_as_float(lo.x << 16), __uint_as_float(hi.x & 0xffff0000u) + __uint_as_float(lo.x & 0xffff0000u),
                                    __uint_as_float(hi.y << 16) + __uint_as_float(lo.y << 16), __uint_as_float(hi.y & 0xffff0000u) + __uint_as_float(lo.y & 0xffff0000u)};
                const float gv[4] = {g8[k].x, g8[k].y, g8[k].z, g8[k].w};
                const bool ok = t0 + tt < T && act;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float v = ok ? gv[c] * (1.f - h[c] * h[c]) : 0.f;
                    accE[tt][c] += v;
                    accD[c] += v;
                }
            }
        }
        float* xw = xch + (threadIdx.x >> 6) * 256;
        *reinterpret_cast<float4*>(xw + (threadIdx.x & 63) * 4) = make_float4(accD[0], accD[1], accD[2], accD[3]);
        float* d = dPD + ((long)b * U1 + u) * J + (blockIdx.x * 256 + (threadIdx.x & ~63)) * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col = 64 * c + (threadIdx.x & 63);
            if ((blockIdx.x * 256 + (threadIdx.x & ~63)) * 4 + col < J) atomicAdd(d + col, xw[col]);
        }
    }
#pragma unroll
    for (int tt = 0; tt < JT_TCX; ++tt)
        if (t0 + tt < T && act)
            *reinterpret_cast<float4*>(dPE + ((long)b * T + t0 + tt) * J + j) = make_float4(accE[tt][0], accE[tt][1], accE[tt][2], accE[tt][3]);
}

constexpr int JT_TC4 = 16;     // frames per block of the kernel below (32: fewer atomics but 0.52 -> 0.87 ms, too few blocks in flight; measured round 3)
// bf16 pipeline: dP = dH * (1 - H^2) was already formed in the dgrad GEMM's epilogue; this only reduces it:
// dPE[b,t,:] = sum_u dP[b,t,u,:], dPD[b,u,:] += sum_t dP[b,t,u,:].  4 columns per thread, 8-byte loads.
__global__ __launch_bounds__(256) void joint_sum_bwd_bf16x4_kernel(const bf16_t* __restrict__ dP, int T, int U1, int J,
                                                                   float* __restrict__ dPE, float* __restrict__ dPD) {
    __shared__ float xch[4 * 256];                     // per wave: 256 column sums on their way to lane-contiguous atomics
    const int j = (blockIdx.x * 256 + threadIdx.x) * 4;
    const bool act = j < J;                            // (whole waves stay alive: the exchange is per wave)
    const int t0 = blockIdx.y * JT_TC4;
    const int b = blockIdx.z;
    float accE[JT_TC4][4];
#pragma unroll
    for (int i = 0; i < JT_TC4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) accE[i][c] = 0.f;
    for (int u = 0; u < U1; ++u) {
        float accD[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tt = 0; tt < JT_TC4; ++tt) {
            const int t = t0 + tt;
            if (t < T && act) {
                const uint2 w = *reinterpret_cast<const uint2*>(dP + (((long)b * T + t) * U1 + u) * J + j);
                const float v[4] = {__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16),
                                    __uint_as_float(w.y & 0xffff0000u)};
#pragma unroll
                for (int c = 0; c < 4; ++c) { accE[tt][c] += v[c]; accD[c] += v[c]; }
            }
        }
        // memory-side atomics want contiguous segments per wave-instruction: the wave's 256 column sums (4 consecutive per lane) are
        // re-dealt through LDS so that lane l adds column 64c + l (one 256-byte segment per instruction instead of 4-byte pieces)
        float* xw = xch + (threadIdx.x >> 6) * 256;
        *reinterpret_cast<float4*>(xw + (threadIdx.x & 63) * 4) = make_float4(accD[0], accD[1], accD[2], accD[3]);
        float* d = dPD + ((long)b * U1 + u) * J + (blockIdx.x * 256 + (threadIdx.x & ~63)) * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col = 64 * c + (threadIdx.x & 63);
            if ((blockIdx.x * 256 + (threadIdx.x & ~63)) * 4 + col < J) atomicAdd(d + col, xw[col]);
        }
    }
#pragma unroll
    for (int tt = 0; tt < JT_TC4; ++tt)
        if (t0 + tt < T && act)
            *reinterpret_cast<float4*>(dPE + ((long)b * T + t0 + tt) * J + j) = make_float4(accE[tt][0], accE[tt][1], accE[tt][2], accE[tt][3]);
}

// Two-pass form of the kernel above (round 6): the block's sums over its 16 frames go to part[b, t / 16, u, :] by plain 16-byte stores instead of 53 M f32 atomics on
// dPD, and joint_dpd_reduce_kernel adds the ceil(T / 16) partial rows of every label state in a fixed order - dPD (and with it the gradient of the label states)
// is the same bits in every run.  214 MB written and read again at C2 against the atomics' memory-side serialisation.
__global__ __launch_bounds__(256) void joint_sum_bwd_bf16x4_part_kernel(const bf16_t* __restrict__ dP, int T, int U1, int J,
                                                                        float* __restrict__ dPE, float* __restrict__ part) {
    const int j = (blockIdx.x * 256 + threadIdx.x) * 4;
    const bool act = j < J;
    const int jc = act ? j : 0;
    const int t0 = blockIdx.y * JT_TC4;
    const int b = blockIdx.z;
    float accE[JT_TC4][4];
#pragma unroll
    for (int i = 0; i < JT_TC4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) accE[i][c] = 0.f;
    float* prow = part + (((long)b * gridDim.y + blockIdx.y) * U1) * J + jc;
    // every load unconditional (frames past T and columns past J read a clamped address and are dropped by a select): under `if (t < T && act)` each of the sixteen
    // loads got a block of its own with s_waitcnt vmcnt(0) behind it - ONE 8-byte load in flight per lane, 3.0 TB/s (the ISA of rounds 3 - 6's kernel above)
    const bf16_t* src = dP + ((long)b * T * U1) * J + jc;
    long roff[JT_TC4];
#pragma unroll
    for (int tt = 0; tt < JT_TC4; ++tt) roff[tt] = (long)min(t0 + tt, T - 1) * U1 * J;
    for (int u = 0; u < U1; ++u) {
        uint2 w[JT_TC4];
#pragma unroll
        for (int tt = 0; tt < JT_TC4; ++tt) w[tt] = *reinterpret_cast<const uint2*>(src + roff[tt] + (long)u * J);
        float accD[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tt = 0; tt < JT_TC4; ++tt) {
            const bool ok = t0 + tt < T;                     // (block-uniform)
            const float v[4] = {ok ? __uint_as_float(w[tt].x << 16) : 0.f, ok ? __uint_as_float(w[tt].x & 0xffff0000u) : 0.f,
                                ok ? __uint_as_float(w[tt].y << 16) : 0.f, ok ? __uint_as_float(w[tt].y & 0xffff0000u) : 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) { accE[tt][c] += v[c]; accD[c] += v[c]; }
        }
        if (act) *reinterpret_cast<float4*>(prow + (long)u * J) = make_float4(accD[0], accD[1], accD[2], accD[3]);
    }
#pragma unroll
    for (int tt = 0; tt < JT_TC4; ++tt)
        if (t0 + tt < T && act)
            *reinterpret_cast<float4*>(dPE + ((long)b * T + t0 + tt) * J + j) = make_float4(accE[tt][0], accE[tt][1], accE[tt][2], accE[tt][3]);
}
// dPD[b, u, :] = sum over the nt partial rows, in order (uj4 = U1 * J / 4 float4 per utterance)
__global__ __launch_bounds__(256) void joint_dpd_reduce_kernel(const float4* __restrict__ part, int nt, long uj4, long n4, float4* __restrict__ dPD) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const long b = i / uj4, r = i - b * uj4;
    const float4* src = part + b * nt * uj4 + r;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int k = 0;
    for (; k + 4 <= nt; k += 4) {
        const float4 a0 = src[(long)k * uj4], a1 = src[(long)(k + 1) * uj4], a2 = src[(long)(k + 2) * uj4], a3 = src[(long)(k + 3) * uj4];
        s.x += a0.x; s.y += a0.y; s.z += a0.z; s.w += a0.w;
        s.x += a1.x; s.y += a1.y; s.z += a1.z; s.w += a1.w;
        s.x += a2.x; s.y += a2.y; s.z += a2.z; s.w += a2.w;
        s.x += a3.x; s.y += a3.y; s.z += a3.z; s.w += a3.w;
    }
    for (; k < nt; ++k) {
        const float4 a = src[(long)k * uj4];
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
    dPD[i] = s;
}

// lo[i] = bf16(src[i] - float(hi[i])): the second term of a two-term bf16 split of an f32 weight (hi = bf16(src))
__global__ void bf16_residual_kernel(const float* __restrict__ src, const bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) lo[i] = f32_to_bf16(src[i] - bf16_to_f32(hi[i]));
}

// r[i] = src[i] - float(bf16(src[i])) in f32 (what a kernel that rounds its operands while staging turns into the same second split term)
__global__ void bf16_residual_f32_kernel(const float* __restrict__ src, float* __restrict__ r, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) r[i] = src[i] - bf16_to_f32(f32_to_bf16(src[i]));
}

// ---- TTMI_PRECISION=bf16x3 (round 5): an f32 operand as THREE bf16 blocks along its reduction dimension.  With x = hi + lo + O(2^-17 |x|)
// (hi = bf16(x), lo = bf16(x - hi)), A . B^T ~ A_hi B_hi + A_lo B_hi + A_hi B_lo: the "A-like" operand is laid out [hi | lo | hi], the "B-like"
// one [hi | hi | lo], and ONE bf16 MFMA GEMM over the tripled reduction gives the product to ~2^-16 relative per term (the dropped lo . lo term),
// at three sixteenths of the exact-f32 MFMA's cycles.  dst [R, 3 Cp] bf16, block b of row r at dst[r * 3 Cp + b * Cp ...], columns [C, Cp) zero.
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ src, long ld, long R, int C, int Cp, int hhl,
                                                     bf16_t* __restrict__ dst) {
    const long q4 = Cp >> 2, n = R * q4;
    const bool vec = (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
        const long r = idx / q4;
        const int c = (int)(idx - r * q4) * 4;
        float x[4];
        const float* sp = src + r * ld + c;
        if (vec && c + 3 < C) {
            const float4 v = *reinterpret_cast<const float4*>(sp);
            x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i] = c + i < C ? sp[i] : 0.f;
        }
        uint2 hi, lo;
        hi.x = pack_bf16x2(x[0], x[1]);
        hi.y = pack_bf16x2(x[2], x[3]);
        lo.x = pack_bf16x2(x[0] - __uint_as_float(hi.x << 16), x[1] - __uint_as_float(hi.x & 0xffff0000u));
        lo.y = pack_bf16x2(x[2] - __uint_as_float(hi.y << 16), x[3] - __uint_as_float(hi.y & 0xffff0000u));
        bf16_t* d = dst + r * (hhl == 2 ? 2 : 3) * Cp + c;
        *reinterpret_cast<uint2*>(d) = hi;
        *reinterpret_cast<uint2*>(d + Cp) = hhl == 1 ? hi : lo;
        if (hhl != 2) *reinterpret_cast<uint2*>(d + 2 * Cp) = hhl ? lo : hi;     // (mode 2: two blocks [hi | lo], the planes of a TN product)
    }
}
// the same of the TRANSPOSE: src f32 [R, C] (pitch ld) -> dst [C, 3 Rp], dst[c * 3 Rp + b * Rp + r], rows r in [R, Rp) zero (weights only: small)
__global__ __launch_bounds__(256) void split3_transpose_kernel(const float* __restrict__ src, long ld, int R, int C, int Rp, int hhl,
                                                               bf16_t* __restrict__ dst) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        tile[ty + 8 * k][tx] = (r < R && c < C) ? src[(long)r * ld + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;
        if (c < C && r < Rp) {
            const float x = tile[tx][ty + 8 * k];
            const bf16_t hi = f32_to_bf16(x), lo = f32_to_bf16(x - bf16_to_f32(hi));
            bf16_t* d = dst + (long)c * 3 * Rp + r;
            d[0] = hi;
            d[Rp] = hhl ? hi : lo;
            d[2 * Rp] = hhl ? lo : hi;
        }
    }
}

__global__ void convert_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        const float4 v = *reinterpret_cast<const float4*>(src + i);
        uint2 w;
        w.x = pack_bf16x2(v.x, v.y);
        w.y = pack_bf16x2(v.z, v.w);
        *reinterpret_cast<uint2*>(dst + i) = w;
    } else {
        for (long k = i; k < n; ++k) dst[k] = f32_to_bf16(src[k]);
    }
}

__global__ void dropout_apply_kernel(const float* __restrict__ in, long n, DropSpec ds, float* __restrict__ o32,
                                     bf16_t* __restrict__ o16) {
    ds = drop_live(ds);
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = in[i] * drop_mult(ds, (unsigned long long)i);
    if (o32) o32[i] = v;
    if (o16) o16[i] = f32_to_bf16(v);
}

// 32x32 tile transpose through LDS
__global__ __launch_bounds__(256) void transpose_convert_kernel(const float* __restrict__ src, int R, int C,
                                                                bf16_t* __restrict__ dst, long ldd, bf16_t* __restrict__ plain) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        const float v = (r < R && c < C) ? src[(long)r * C + c] : 0.f;
        tile[i][tx] = v;
        if (plain && r < R && c < C) plain[(long)r * C + c] = f32_to_bf16(v);      // the untransposed bf16 copy from the same read
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < C && r < ldd) dst[(long)c * ldd + r] = f32_to_bf16(tile[tx][i]);
    }
}

// the same tile transpose for MANY weights in one launch (the bf16 shadows of ttmi.train.FlatModel, refreshed once per optimiser step):
// table row = (src f32*, R, C, dstT bf16*, ldd, plain bf16*, first tile, tiles along C); block b serves the weight whose tile range holds b
__global__ __launch_bounds__(256) void shadow_refresh_kernel(const long* __restrict__ table, int n, long lo_delta) {
    __shared__ float tile[32][33];
    __shared__ long ent[8];
    if (threadIdx.x == 0) {
        int lo = 0, hi = n - 1;
        while (lo < hi) {                                   // last entry whose first tile <= blockIdx.x
            const int mid = (lo + hi + 1) >> 1;
            if (table[mid * 8 + 6] <= (long)blockIdx.x) lo = mid; else hi = mid - 1;
        }
        for (int k = 0; k < 8; ++k) ent[k] = table[lo * 8 + k];
    }
    __syncthreads();
    const float* src = reinterpret_cast<const float*>(ent[0]);
    const int R = (int)ent[1], C = (int)ent[2];
    bf16_t* dst = reinterpret_cast<bf16_t*>(ent[3]);
    const long ldd = ent[4];
    bf16_t* plain = reinterpret_cast<bf16_t*>(ent[5]);
    const int t = (int)((long)blockIdx.x - ent[6]), tx_n = (int)ent[7];
    const int r0 = (t / tx_n) * 32, c0 = (t % tx_n) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        const float v = (r < R && c < C) ? src[(long)r * C + c] : 0.f;
        tile[i][tx] = v;
        if (plain && r < R && c < C) {
            const bf16_t h = f32_to_bf16(v);
            plain[(long)r * C + c] = h;
            if (lo_delta) plain[lo_delta + (long)r * C + c] = f32_to_bf16(v - bf16_to_f32(h));     // second term of the weight's bf16 split (round 6: option 13's operand, once per step)
        }
    }
    __syncthreads();
    if (dst)
        for (int i = ty; i < 32; i += 8) {
            const int c = c0 + i, r = r0 + tx;
            if (c < C && r < ldd) dst[(long)c * ldd + r] = f32_to_bf16(tile[tx][i]);     // rows [R, ldd) of the pitch receive zeros
        }
}

constexpr int CSB_ROWS = 128;
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ in_, long ld, long rows, int cols,
                                                          float* __restrict__ out_, int nz2, long si1, long si2, long so2) {
    const bf16_t* in = in_ + (blockIdx.z / nz2) * si1 + (blockIdx.z % nz2) * si2;
    float* out = out_ + (blockIdx.z % nz2) * so2;
    // each thread owns 2 adjacent columns (one 4-byte load); a wave covers 128 columns = 256 B per row
    const int c = (blockIdx.x * 256 + threadIdx.x) * 2;
    if (c >= cols) return;
    const long r0 = (long)blockIdx.y * CSB_ROWS;
    const long r1 = r0 + CSB_ROWS < rows ? r0 + CSB_ROWS : rows;
    float a0 = 0.f, a1 = 0.f;
    const bool pair = (c + 1 < cols) && ((ld & 1) == 0);
    long r = r0;
    if (pair) {                                    // 4 independent loads in flight per thread
        float b0 = 0.f, b1 = 0.f, c0 = 0.f, c1 = 0.f, d0 = 0.f, d1 = 0.f;
        for (; r + 3 < r1; r += 4) {
            const uint32_t w0 = *reinterpret_cast<const uint32_t*>(in + r * ld + c);
            const uint32_t w1 = *reinterpret_cast<const uint32_t*>(in + (r + 1) * ld + c);
            const uint32_t w2 = *reinterpret_cast<const uint32_t*>(in + (r + 2) * ld + c);
            const uint32_t w3 = *reinterpret_cast<const uint32_t*>(in + (r + 3) * ld + c);
            a0 += __uint_as_float(w0 << 16); a1 += __uint_as_float(w0 & 0xffff0000u);
            b0 += __uint_as_float(w1 << 16); b1 += __uint_as_float(w1 & 0xffff0000u);
            c0 += __uint_as_float(w2 << 16); c1 += __uint_as_float(w2 & 0xffff0000u);
            d0 += __uint_as_float(w3 << 16); d1 += __uint_as_float(w3 & 0xffff0000u);
        }
        a0 += (b0 + c0) + d0;
        a1 += (b1 + c1) + d1;
    }
    for (; r < r1; ++r) {
        if (pair) {
            const uint32_t w = *reinterpret_cast<const uint32_t*>(in + r * ld + c);
            a0 += __uint_as_float(w << 16);
            a1 += __uint_as_float(w & 0xffff0000u);
        } else {
            a0 += bf16_to_f32(in[r * ld + c]);
            if (c + 1 < cols) a1 += bf16_to_f32(in[r * ld + c + 1]);
        }
    }
    atomicAdd(out + c, a0);
    if (c + 1 < cols) atomicAdd(out + c + 1, a1);
}

template <typename TS>
__global__ __launch_bounds__(256) void transpose_batched_kernel(const TS* __restrict__ src, long ld, int nz2, long s1, long s2, int R,
                                                                int C, bf16_t* __restrict__ dst, long ldd) {
    __shared__ float tile[32][33];
    const int z = blockIdx.z, z1 = z / nz2, z2 = z % nz2;
    const TS* sp = src + z1 * s1 + z2 * s2;
    bf16_t* dp = dst + (long)z * C * ldd;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        float v = 0.f;
        if (r < R && c < C) {
            if constexpr (sizeof(TS) == 4) v = sp[(long)r * ld + c];
            else v = bf16_to_f32(sp[(long)r * ld + c]);
        }
        tile[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < C && r < ldd) dp[(long)c * ldd + r] = f32_to_bf16(tile[tx][i]);
    }
}

// greedy scan: rows of logits (f32 or bf16, pitch ld); out[0] = first row whose argmax != blank (or n), out[1] = that argmax.
// One wave per row computes the argmax (first maximal index, like torch.argmax); the rows' results go through a single
// atomicMin on the packed (row << 32 | token) key, so one 8-byte D2H read tells the host where the next symbol is.
template <typename TL>
__global__ __launch_bounds__(256) void greedy_scan_kernel(const TL* __restrict__ logits, long ld, int n, int V, int blank,
                                                          unsigned long long* __restrict__ out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n) return;
    const TL* r = logits + (long)row * ld;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int v = lane; v < V; v += 64) {
        float x;
        if constexpr (sizeof(TL) == 4) x = r[v];
        else x = bf16_to_f32(r[v]);
        if (x > best) { best = x; bi = v; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0 && bi != blank) atomicMin(out, ((unsigned long long)row << 32) | (unsigned)bi);
}

// ---- batched greedy decoding in lockstep over SYMBOL steps (tt/model.py:70-108, every utterance of a batch at once): after step s every
// utterance still decoding holds exactly s + 1 tokens, so ONE label-encoder call of length s + 1 serves the whole batch exactly (the
// relative-position term depends on the sequence length: utterances of different history lengths cannot share a padded call).
// greedy_scan_batch: logits [B, n, V] = the joint of frames t_b .. t_b + n - 1 of every utterance against its own label state; per utterance
// the first frame (inside its length) whose argmax is not blank -> key[b] = frame offset << 32 | symbol (atomicMin; key[b] = n << 32 before).
template <typename TL>
__global__ __launch_bounds__(256) void greedy_scan_batch_kernel(const TL* __restrict__ logits, long ld, int B, int n, int V, int blank,
                                                                const int* __restrict__ t, const int* __restrict__ T_len,
                                                                const int* __restrict__ need, unsigned long long* __restrict__ key) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B * n) return;
    const int b = row / n, r = row - b * n;
    if (!need[b] || t[b] + r >= T_len[b]) return;       // (wave-uniform) this utterance has its symbol already, or the frame does not exist
    const TL* p = logits + (long)row * ld;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int v = lane; v < V; v += 64) {
        float x;
        if constexpr (sizeof(TL) == 4) x = p[v];
        else x = bf16_to_f32(p[v]);
        if (x > best) { best = x; bi = v; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0 && bi != blank) atomicMin(key + b, ((unsigned long long)r << 32) | (unsigned)bi);
}
// one thread per utterance: consume key[b].  A symbol found: append it to the history (column n_hist), move past its frame, this utterance
// is served for the step; none in these n frames: move on n frames, finished when the utterance has no frames left.  flags[0] = utterances
// that still need a symbol in this step (the host scans another block), flags[1] = utterances not finished; key is reset for the next scan.
__global__ void greedy_advance_kernel(unsigned long long* __restrict__ key, int B, int n, int n_hist, long* __restrict__ hist, long ld_hist,
                                      int* __restrict__ t, const int* __restrict__ T_len, int* __restrict__ need, int* __restrict__ done,
                                      int* __restrict__ count, int* __restrict__ flags) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const unsigned long long k = key[b];
    key[b] = (unsigned long long)n << 32;
    if (need[b]) {
        const int row = (int)(k >> 32);
        if (row < n) {
            hist[(long)b * ld_hist + n_hist] = (long)(unsigned)(k & 0xffffffffu);
            t[b] += row + 1;                                       // the emitting frame is consumed (at most one symbol per frame)
            count[b] += 1;
            need[b] = 0;
        } else {
            t[b] += n;
            if (t[b] >= T_len[b]) { need[b] = 0; done[b] = 1; }
        }
    }
    if (need[b]) atomicAdd(flags, 1);
    if (!done[b]) atomicAdd(flags + 1, 1);
}

}  // namespace

int greedy_scan_batch(const void* logits, int dtype, long ld, int B, int n, int V, int blank, const int* t, const int* T_len, const int* need,
                      unsigned long long* key, hipStream_t st) {
    TTMI_REQUIRE(logits && t && T_len && need && key && B > 0 && n > 0 && V > 0 && ld >= V, "greedy_scan_batch: bad arguments");
    if (dtype == 0)
        hipLaunchKernelGGL(greedy_scan_batch_kernel<float>, dim3(cdiv((long)B * n, 4)), dim3(256), 0, st, static_cast<const float*>(logits), ld, B, n, V,
                           blank, t, T_len, need, key);
    else
        hipLaunchKernelGGL(greedy_scan_batch_kernel<bf16_t>, dim3(cdiv((long)B * n, 4)), dim3(256), 0, st, static_cast<const bf16_t*>(logits), ld, B, n,
                           V, blank, t, T_len, need, key);
    TTMI_LAUNCH_CHECK("greedy_scan_batch_kernel");
    return TTMI_OK;
}

int greedy_advance(unsigned long long* key, int B, int n, int n_hist, long* hist, long ld_hist, int* t, const int* T_len, int* need, int* done,
                   int* count, int* flags, hipStream_t st) {
    TTMI_REQUIRE(key && hist && t && T_len && need && done && count && flags && B > 0 && n > 0 && n_hist >= 1 && n_hist < ld_hist,
                 "greedy_advance: bad arguments");
    if (fill_zero(flags, 2 * sizeof(int), st) != TTMI_OK) return TTMI_EINVAL;
    hipLaunchKernelGGL(greedy_advance_kernel, dim3(cdiv(B, 64)), dim3(64), 0, st, key, B, n, n_hist, hist, ld_hist, t, T_len, need, done, count, flags);
    TTMI_LAUNCH_CHECK("greedy_advance_kernel");
    return TTMI_OK;
}

int ln_fwd(const float* x, const float* res, const float* g, const float* b, long rows, int d, float eps, float* s_out,
           float* y, float* mean, float* rstd, hipStream_t st, bf16_t* y16, DropSpec res_drop, DropSpec out_drop, const LnPreNorm* pre) {
    TTMI_REQUIRE(x && g && b && (y || y16) && rows > 0 && d > 0, "ln_fwd: bad arguments");
    TTMI_REQUIRE(!pre || y, "ln_fwd: the second norm reads the f32 output row");
    const int vec = (d % 4 == 0) && aligned16(x) && (!y || aligned16(y)) && aligned16(g) && aligned16(b) && (!res || aligned16(res)) &&
                    (!s_out || aligned16(s_out)) && (!y16 || (reinterpret_cast<uintptr_t>(y16) & 7) == 0);
    TTMI_REQUIRE(!pre || (pre->g && pre->b && pre->h16 && pre->mean && pre->rstd), "ln_fwd: incomplete second norm");
    if (vec && d <= 512 && (!pre || (aligned16(pre->g) && aligned16(pre->b) && (reinterpret_cast<uintptr_t>(pre->h16) & 7) == 0))) {
        const float* g2 = pre ? pre->g : nullptr;
        const float* b2 = pre ? pre->b : nullptr;
        bf16_t* h16 = pre ? pre->h16 : nullptr;
        float* m2 = pre ? pre->mean : nullptr;
        float* r2 = pre ? pre->rstd : nullptr;
        if (d <= 256)
            hipLaunchKernelGGL(ln_fwd_row_kernel<1>, dim3(cdiv(rows, WPB)), dim3(WPB * 64), 0, st, x, res, g, b, rows, d, eps, s_out, y, mean,
                               rstd, y16, res_drop, out_drop, g2, b2, h16, m2, r2);
        else
            hipLaunchKernelGGL(ln_fwd_row_kernel<2>, dim3(cdiv(rows, WPB)), dim3(WPB * 64), 0, st, x, res, g, b, rows, d, eps, s_out, y, mean,
                               rstd, y16, res_drop, out_drop, g2, b2, h16, m2, r2);
        TTMI_LAUNCH_CHECK("ln_fwd_row_kernel");
        return TTMI_OK;
    }
    hipLaunchKernelGGL(ln_fwd_kernel, dim3(cdiv(rows, WPB)), dim3(WPB * 64), 0, st, x, res, g, b, rows, d, eps, s_out, y, mean,
                       rstd, vec, y16, res_drop, out_drop);
    TTMI_LAUNCH_CHECK("ln_fwd_kernel");
    if (pre)                                                // shapes the row kernel does not take: the second norm as its own launch
        return ln_fwd(y, nullptr, pre->g, pre->b, rows, d, eps, nullptr, nullptr, pre->mean, pre->rstd, st, pre->h16);
    return TTMI_OK;
}

int g_ln_bwd_grid = LN_BWD_GRID;
int ln_bwd(const float* dy, const float* s, const float* mean, const float* rstd, const float* g, const float* dadd, long rows,
           int d, float* dx, float* dgamma, float* dbeta, hipStream_t st, DropSpec dy_drop, bf16_t* dx16, DropSpec dx16_drop,
           float* dx16_colsum) {
    TTMI_REQUIRE(dy && s && mean && rstd && g && dx && dgamma && dbeta && rows > 0 && d > 0, "ln_bwd: bad arguments");
    const bool fused = d % 4 == 0 && d <= 512 && aligned16(dy) && aligned16(s) && aligned16(g) && aligned16(dx) && (!dadd || aligned16(dadd)) &&
                       (!dx16 || (reinterpret_cast<uintptr_t>(dx16) & 7) == 0);
    if (fused) {
        const int grid = (int)std::min<long>(cdiv(rows, 4), g_ln_bwd_grid);   // rows are walked grid-stride
        if (d <= 256)
            hipLaunchKernelGGL(ln_bwd_fused_kernel<1>, dim3(grid), dim3(256), 0, st, dy, s, mean, rstd, g, dadd, rows, d, dx, dgamma, dbeta, dy_drop, dx16, dx16_drop, dx16_colsum);
        else
            hipLaunchKernelGGL(ln_bwd_fused_kernel<2>, dim3(grid), dim3(256), 0, st, dy, s, mean, rstd, g, dadd, rows, d, dx, dgamma, dbeta, dy_drop, dx16, dx16_drop, dx16_colsum);
        TTMI_LAUNCH_CHECK("ln_bwd_fused_kernel");
        return TTMI_OK;
    }
    hipLaunchKernelGGL(ln_bwd_dx_kernel, dim3(cdiv(rows, WPB)), dim3(WPB * 64), 0, st, dy, s, mean, rstd, g, dadd, rows, d, dx,
                       dy_drop);
    TTMI_LAUNCH_CHECK("ln_bwd_dx_kernel");
    hipLaunchKernelGGL(ln_bwd_params_kernel, dim3(cdiv(d, 256), cdiv(rows, LNP_ROWS)), dim3(256), 0, st, dy, s, mean, rstd, rows,
                       d, dgamma, dbeta, dy_drop);
    TTMI_LAUNCH_CHECK("ln_bwd_params_kernel");
    if (dx16) {                                             // shapes the fused kernel does not take: same results by separate passes
        if (int rc = dropout_apply(dx, rows * d, dx16_drop, nullptr, dx16, st)) return rc;
        if (dx16_colsum) return colsum_bf16(dx16, d, rows, d, dx16_colsum, st, 1, 1, 0, 0, 0);
    }
    return TTMI_OK;
}

bool ln_bwd_pair_supported(int d) { return d % 4 == 0 && d <= 512; }
int ln_bwd_pair(const float* dh, const float* y, const float* mean1, const float* rstd1, const float* g1, const float* dres1, const float* s,
                const float* mean2, const float* rstd2, const float* g2, long rows, int d, float* dx, float* dgamma1, float* dbeta1,
                float* dgamma2, float* dbeta2, bf16_t* dx16, DropSpec dx16_drop, hipStream_t st) {
    TTMI_REQUIRE(dh && y && mean1 && rstd1 && g1 && dres1 && s && mean2 && rstd2 && g2 && dx && dgamma1 && dbeta1 && dgamma2 && dbeta2 && dx16 &&
                 rows > 0, "ln_bwd_pair: bad arguments");
    TTMI_REQUIRE(ln_bwd_pair_supported(d) && aligned16(dh) && aligned16(y) && aligned16(dres1) && aligned16(s) && aligned16(g1) && aligned16(g2) &&
                 aligned16(dx) && (reinterpret_cast<uintptr_t>(dx16) & 7) == 0, "ln_bwd_pair: d %% 4 == 0, d <= 512 and 16-byte aligned rows required");
    const int grid = (int)std::min<long>(cdiv(rows, 4), g_ln_bwd_grid);
    if (d <= 256)
        hipLaunchKernelGGL(ln_bwd_pair_kernel<1>, dim3(grid), dim3(256), 0, st, dh, y, mean1, rstd1, g1, dres1, s, mean2, rstd2, g2, rows, d, dx,
                           dgamma1, dbeta1, dgamma2, dbeta2, dx16, dx16_drop);
    else
        hipLaunchKernelGGL(ln_bwd_pair_kernel<2>, dim3(grid), dim3(256), 0, st, dh, y, mean1, rstd1, g1, dres1, s, mean2, rstd2, g2, rows, d, dx,
                           dgamma1, dbeta1, dgamma2, dbeta2, dx16, dx16_drop);
    TTMI_LAUNCH_CHECK("ln_bwd_pair_kernel");
    return TTMI_OK;
}

int softmax_fwd(float* S, int nb, int nh, int L, long ld, long slab, float scale, const MaskDesc& m, hipStream_t st) {
    TTMI_REQUIRE(S && nb > 0 && nh > 0 && L > 0, "softmax_fwd: bad arguments");
    TTMI_REQUIRE((m.kind != MASK_TENSOR && m.kind != MASK_INTERVAL) || m.ptr, "softmax_fwd: tensor mask without pointer");
    const long nrows = (long)nb * nh * L;
    if (L <= 512) hipLaunchKernelGGL(softmax_fwd_reg_kernel, dim3(cdiv(nrows, WPB)), dim3(WPB * 64), 0, st, S, nrows, nh, L, ld, slab, scale, m);
    else hipLaunchKernelGGL(softmax_fwd_kernel, dim3(cdiv(nrows, WPB)), dim3(WPB * 64), 0, st, S, nrows, nh, L, ld, slab, scale, m);
    TTMI_LAUNCH_CHECK("softmax_fwd_kernel");
    return TTMI_OK;
}

int softmax_bwd(float* dP, const float* P, int nb, int L, long ld, long slab, float scale, hipStream_t st) {
    TTMI_REQUIRE(dP && P && nb > 0 && L > 0, "softmax_bwd: bad arguments");
    const long nrows = (long)nb * L;
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3(cdiv(nrows, WPB)), dim3(WPB * 64), 0, st, dP, P, nrows, L, ld, slab, scale);
    TTMI_LAUNCH_CHECK("softmax_bwd_kernel");
    return TTMI_OK;
}

int add_row_bias(const float* in, long ldi, const float* bias, long rows, int cols, float* out, long ldo, hipStream_t st) {
    TTMI_REQUIRE(in && bias && out && rows > 0 && cols > 0, "add_row_bias: bad arguments");
    hipLaunchKernelGGL(add_row_bias_kernel, dim3(cdiv(rows * cols, 256)), dim3(256), 0, st, in, ldi, bias, rows, cols, out, ldo);
    TTMI_LAUNCH_CHECK("add_row_bias_kernel");
    return TTMI_OK;
}

int colsum(const float* in, long ld, long rows, int cols, int nz1, int nz2, long si1, long si2, long so1, long so2, float* out,
           hipStream_t st) {
    TTMI_REQUIRE(in && out && rows > 0 && cols > 0 && nz1 > 0 && nz2 > 0, "colsum: bad arguments");
    if (nz1 * nz2 == 1 && cols % 4 == 0 && ld % 4 == 0 && aligned16(in) && cdiv(rows, 64) <= 65535) {
        hipLaunchKernelGGL(colsum_vec_kernel, dim3(cdiv(cols, 256), cdiv(rows, 64)), dim3(256), 0, st, in, ld, rows, cols, out);
        TTMI_LAUNCH_CHECK("colsum_vec_kernel");
        return TTMI_OK;
    }
    dim3 grid(cdiv(cols, 256), cdiv(rows, CS_ROWS), nz1 * nz2);
    TTMI_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "colsum: grid too large");
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, st, in, ld, rows, cols, nz2, si1, si2, so1, so2, out);
    TTMI_LAUNCH_CHECK("colsum_kernel");
    return TTMI_OK;
}

int relpos_gather(const float* r_emb, const float* r_bias, int K, int L, int H, int Dh, float* E, float* cT, hipStream_t st, bf16_t* E16) {
    TTMI_REQUIRE(r_emb && r_bias && E && cT && K > 0 && L > 0, "relpos_gather: bad arguments");
    hipLaunchKernelGGL(relpos_gather_kernel, dim3(cdiv((long)L * H * Dh, 256)), dim3(256), 0, st, r_emb, r_bias, K, L, H, Dh, E,
                       cT, E16);
    TTMI_LAUNCH_CHECK("relpos_gather_kernel");
    return TTMI_OK;
}

int relpos_scatter(const float* dE, const float* dcT, int K, int L, int H, int Dh, float* g_r_emb, float* g_r_bias,
                   hipStream_t st) {
    TTMI_REQUIRE(dE && dcT && g_r_emb && g_r_bias, "relpos_scatter: bad arguments");
    const int folded = L - K >= 256;
    hipLaunchKernelGGL(relpos_scatter_kernel, dim3(cdiv((long)L * H * Dh, 256)), dim3(256), 0, st, dE, dcT, K, L, H, Dh, g_r_emb,
                       g_r_bias, folded);
    TTMI_LAUNCH_CHECK("relpos_scatter_kernel");
    if (folded) {
        hipLaunchKernelGGL(relpos_fold_row0_kernel, dim3(cdiv(H * Dh, 256), cdiv(L - K, FOLD_ROWS)), dim3(256), 0, st, dE, dcT, L - K, L, H, Dh,
                           g_r_emb, g_r_bias);
        TTMI_LAUNCH_CHECK("relpos_fold_row0_kernel");
    }
    return TTMI_OK;
}

int embed_fwd(const long* tokens, const float* W, long n, int d, int V, float* out, hipStream_t st) {
    TTMI_REQUIRE(tokens && W && out && n > 0 && d > 0 && V > 0, "embed_fwd: bad arguments");
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(cdiv(n * d, 256)), dim3(256), 0, st, tokens, W, n, d, V, out);
    TTMI_LAUNCH_CHECK("embed_fwd_kernel");
    return TTMI_OK;
}

int embed_bwd(const long* tokens, const float* dout, long n, int d, int V, int padding_idx, float* gW, hipStream_t st) {
    TTMI_REQUIRE(tokens && dout && gW && n > 0 && d > 0 && V > 0, "embed_bwd: bad arguments");
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(cdiv(n * d, 256)), dim3(256), 0, st, tokens, dout, n, d, V, padding_idx, gW);
    TTMI_LAUNCH_CHECK("embed_bwd_kernel");
    return TTMI_OK;
}

int joint_tanh_fwd(const float* PE, const float* PD, const float* bias, int B, int T, int U1, int J, void* H, int h_dtype,
                   hipStream_t st) {
    TTMI_REQUIRE(PE && PD && bias && H && B > 0 && T > 0 && U1 > 0 && J > 0, "joint_tanh_fwd: bad arguments");
    if (h_dtype == 0)
        hipLaunchKernelGGL(joint_tanh_fwd_kernel<float>, dim3(B * T), dim3(256), 0, st, PE, PD, bias, T, U1, J,
                           static_cast<float*>(H));
    else if ((J == 256 || J == 512 || J == 1024 || J == 2048) && aligned16(PE) && aligned16(PD) && aligned16(bias) && aligned16(H))
        hipLaunchKernelGGL(joint_tanh_fwd_bf16x8_kernel<false>, dim3(B * cdiv(T, JT_TT)), dim3(256), 0, st, PE, PD, bias, T, U1, J,
                           static_cast<bf16_t*>(H), JointEmis());
    else if (J % 4 == 0 && aligned16(PE) && aligned16(PD) && aligned16(bias) && (reinterpret_cast<uintptr_t>(H) & 7) == 0)
        hipLaunchKernelGGL(joint_tanh_fwd_bf16x4_kernel, dim3(B * T), dim3(256), 0, st, PE, PD, bias, T, U1, J, static_cast<bf16_t*>(H));
    else
        hipLaunchKernelGGL(joint_tanh_fwd_kernel<bf16_t>, dim3(B * T), dim3(256), 0, st, PE, PD, bias, T, U1, J,
                           static_cast<bf16_t*>(H));
    TTMI_LAUNCH_CHECK("joint_tanh_fwd_kernel");
    return TTMI_OK;
}

// bf16 H plus the f32 blank / label logits of every lattice row (exp-domain loss form): emis [B*T*U1, 2]
int joint_tanh_fwd_emis(const float* PE, const float* PD, const float* bias, int B, int T, int U1, int J, bf16_t* H, const bf16_t* Wp16,
                        const float* Wp32, const float* bp, const int* labels, int V, int blank, float* emis, hipStream_t st) {
    TTMI_REQUIRE(PE && PD && bias && H && Wp16 && Wp32 && bp && emis && (labels || U1 == 1) && B > 0 && T > 0 && U1 > 0 && J > 0 && V > 0 &&
                 blank >= 0 && blank < V, "joint_tanh_fwd_emis: bad arguments");
    TTMI_REQUIRE(aligned16(emis), "joint_tanh_fwd_emis: emis must be 16-byte aligned");
    JointEmis em;
    em.Wp16 = Wp16; em.Wp32 = Wp32; em.bp = bp; em.labels = labels; em.out = emis; em.V = V; em.blank = blank;
    static int tt_env = -1;
    if (tt_env < 0) { const char* e = getenv("TTMI_JOINT_TT"); tt_env = e ? atoi(e) : 8; }
    // 8 frames per block where there are frames to spare: the label rows of Wp16 / Wp32 and the PD row are read once per u for eight frames
    // (570 -> 545 us at C2, same box; TTMI_JOINT_TT=4 for A/B)
    const int TT = (tt_env == 8 && T >= 64) ? 8 : JT_TT;
    const size_t part_bytes = (size_t)TT * U1 * (J / 8 >= 64 ? J / 8 / 64 : 1) * 4 * sizeof(float);
    if ((J == 256 || J == 512 || J == 1024 || J == 2048) && aligned16(PE) && aligned16(PD) && aligned16(bias) && aligned16(H) && aligned16(Wp16) &&
        aligned16(Wp32) && part_bytes <= 60 * 1024) {
        if (TT == 8)
            hipLaunchKernelGGL((joint_tanh_fwd_bf16x8_kernel<true, 8>), dim3(B * cdiv(T, 8)), dim3(256), part_bytes, st, PE, PD, bias, T, U1, J, H, em);
        else
            hipLaunchKernelGGL((joint_tanh_fwd_bf16x8_kernel<true, JT_TT_DEFAULT>), dim3(B * cdiv(T, JT_TT)), dim3(256), part_bytes, st,
                           PE, PD, bias, T, U1, J, H, em);
        TTMI_LAUNCH_CHECK("joint_tanh_fwd_bf16x8_kernel<emis>");
        return TTMI_OK;
    }
    if (int rc = joint_tanh_fwd(PE, PD, bias, B, T, U1, J, H, 1, st)) return rc;
    const long rows = (long)B * T * U1;
    hipLaunchKernelGGL(joint_emis_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, H, rows, T, U1, J, em);
    TTMI_LAUNCH_CHECK("joint_emis_kernel");
    return TTMI_OK;
}

int joint_tanh_bwd(const void* dH, const void* H, int h_dtype, int B, int T, int U1, int J, float* dPE, float* dPD,
                   hipStream_t st) {
    TTMI_REQUIRE(dH && dPE && dPD && B > 0 && T > 0 && U1 > 0 && J > 0, "joint_tanh_bwd: bad arguments");
    if (!H) {     // dH already holds dH * (1 - H^2) (formed in the dgrad epilogue)
        TTMI_REQUIRE(h_dtype == 1 && J % 4 == 0 && aligned16(dPE) && (reinterpret_cast<uintptr_t>(dH) & 7) == 0, "joint_tanh_bwd: pre-multiplied input needs bf16, J %% 4 == 0");
        hipLaunchKernelGGL(joint_sum_bwd_bf16x4_kernel, dim3(cdiv(J, 1024), cdiv(T, JT_TC4), B), dim3(256), 0, st,
                           static_cast<const bf16_t*>(dH), T, U1, J, dPE, dPD);
        TTMI_LAUNCH_CHECK("joint_sum_bwd_bf16 kernel");
        return TTMI_OK;
    }
    dim3 grid(cdiv(J, 256), cdiv(T, JT_TC), B);
    if (h_dtype == 0)
        hipLaunchKernelGGL(joint_tanh_bwd_kernel<float>, grid, dim3(256), 0, st, static_cast<const float*>(dH),
                           static_cast<const float*>(H), T, U1, J, dPE, dPD);
    else
        hipLaunchKernelGGL(joint_tanh_bwd_kernel<bf16_t>, grid, dim3(256), 0, st, static_cast<const bf16_t*>(dH),
                           static_cast<const bf16_t*>(H), T, U1, J, dPE, dPD);
    TTMI_LAUNCH_CHECK("joint_tanh_bwd_kernel");
    return TTMI_OK;
}

// C[i, j] += T[i, j] + T[i, Np + j] + T[Mp + i, j]: the three terms of a bf16x3 weight gradient out of the one-launch block product (layers.hip x3_tn)
__global__ __launch_bounds__(256) void x3_fold_blocks_kernel(const float* __restrict__ T, int Mp, int Np, float* __restrict__ C, int M, int N, long ldc) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)M * N) return;
    const int r = (int)(i / N), c = (int)(i - (long)r * N);
    const long ldt = 2L * Np;
    C[(long)r * ldc + c] += T[r * ldt + c] + T[r * ldt + Np + c] + T[(long)(Mp + r) * ldt + c];
}
int x3_fold_blocks(const float* T, int Mp, int Np, float* C, int M, int N, long ldc, hipStream_t st) {
    TTMI_REQUIRE(T && C && M > 0 && N > 0 && Mp >= M && Np >= N && ldc >= N, "x3_fold_blocks: bad arguments");
    hipLaunchKernelGGL(x3_fold_blocks_kernel, dim3((unsigned)cdiv((long)M * N, 256L)), dim3(256), 0, st, T, Mp, Np, C, M, N, ldc);
    TTMI_LAUNCH_CHECK("x3_fold_blocks_kernel");
    return TTMI_OK;
}

size_t joint_sum_bwd_part_floats(int B, int T, int U1, int J) { return (size_t)B * cdiv(T, JT_TC4) * U1 * J; }
// dPE / dPD from dP = dH (1 - H^2) (bf16, formed in the dgrad epilogue) without atomics: part = joint_sum_bwd_part_floats() floats of scratch, 16-byte aligned
int joint_sum_bwd_two_pass(const bf16_t* dP, int B, int T, int U1, int J, float* dPE, float* dPD, float* part, hipStream_t st) {
    TTMI_REQUIRE(dP && dPE && dPD && part && B > 0 && T > 0 && U1 > 0 && J > 0 && J % 4 == 0 && aligned16(dPE) && aligned16(dPD) && aligned16(part) &&
                 (reinterpret_cast<uintptr_t>(dP) & 7) == 0, "joint_sum_bwd_two_pass: bad arguments");
    const int nt = cdiv(T, JT_TC4);
    hipLaunchKernelGGL(joint_sum_bwd_bf16x4_part_kernel, dim3(cdiv(J, 1024), nt, B), dim3(256), 0, st, dP, T, U1, J, dPE, part);
    TTMI_LAUNCH_CHECK("joint_sum_bwd_bf16x4_part_kernel");
    const long uj4 = (long)U1 * J / 4, n4 = (long)B * uj4;
    hipLaunchKernelGGL(joint_dpd_reduce_kernel, dim3((unsigned)cdiv(n4, 256L)), dim3(256), 0, st, reinterpret_cast<const float4*>(part), nt, uj4, n4,
                       reinterpret_cast<float4*>(dPD));
    TTMI_LAUNCH_CHECK("joint_dpd_reduce_kernel");
    return TTMI_OK;
}

int joint_tanh_fwd_x3(const float* PE, const float* PD, const float* bias, int B, int T, int U1, int J, int Jp, bf16_t* H3, hipStream_t st) {
    TTMI_REQUIRE(PE && PD && bias && H3 && B > 0 && T > 0 && U1 > 0 && J > 0 && J % 4 == 0 && Jp >= J && Jp % 4 == 0 && aligned16(PE) && aligned16(PD) &&
                 aligned16(bias) && aligned16(H3), "joint_tanh_fwd_x3: bad arguments");
    hipLaunchKernelGGL(joint_tanh_fwd_x3_kernel, dim3(B * T), dim3(256), 0, st, PE, PD, bias, T, U1, J, Jp, H3);
    TTMI_LAUNCH_CHECK("joint_tanh_fwd_x3_kernel");
    return TTMI_OK;
}
int joint_tanh_bwd_x3(const float* dH, const bf16_t* H3, int B, int T, int U1, int J, int Jp, float* dPE, float* dPD, hipStream_t st) {
    TTMI_REQUIRE(dH && H3 && dPE && dPD && B > 0 && T > 0 && U1 > 0 && J > 0 && J % 4 == 0 && Jp >= J && Jp % 4 == 0 && aligned16(dH) && aligned16(dPE) &&
                 (reinterpret_cast<uintptr_t>(H3) & 7) == 0, "joint_tanh_bwd_x3: bad arguments");
    hipLaunchKernelGGL(joint_tanh_bwd_x3_kernel, dim3(cdiv(J, 1024), cdiv(T, JT_TCX), B), dim3(256), 0, st, dH, H3, T, U1, J, Jp, dPE, dPD);
    TTMI_LAUNCH_CHECK("joint_tanh_bwd_x3_kernel");
    return TTMI_OK;
}

// Zero fills are KERNELS, never hipMemsetAsync / hipMemset2DAsync (round 5): a memset NODE inside a captured HIP graph is not ordered
// against its neighbouring kernel nodes by ROCm 7.2's packet-capture graph launch (DEBUG_CLR_GRAPH_PACKET_CAPTURE, on by default) - the
// 2-D fill of the position slab's column 0 (attn_fwd_impl, fp32 path) landed before OR after the GEMMs around it from replay to replay, and
// the greedy decoder's label-encoder graphs came back with other token sets in a third of the processes (tools/debug/replay_divergence.py:
// hipMemset2DAsync 7 / 7 bad passes, this kernel 0 / 4, packet capture off 0 / 4).  TTMI_MEMSET_KERNEL=0 restores the
// runtime's memsets (to reproduce the failure only).
__global__ __launch_bounds__(256) void zero_bytes_kernel(char* __restrict__ p, size_t head, size_t n16, size_t tail) {
    // [p, p + head) bytes up to the first 16-byte boundary, then n16 16-byte pieces, then `tail` bytes
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x, gsz = (size_t)gridDim.x * 256;
    uint4* q = reinterpret_cast<uint4*>(p + head);
    const uint4 z = {0u, 0u, 0u, 0u};
    for (size_t i = gid; i < n16; i += gsz) q[i] = z;
    if (gid < head) p[gid] = 0;
    if (gid < tail) p[head + n16 * 16 + gid] = 0;
}
// rows of `width2` 2-byte elements every `pitch2` elements
__global__ __launch_bounds__(256) void zero2d_kernel(unsigned short* __restrict__ p, size_t pitch2, size_t width2, size_t height) {
    const size_t n = width2 * height;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t r = i / width2, c = i - r * width2;
        p[r * pitch2 + c] = 0;
    }
}
// the same for odd widths / pitches / addresses: single bytes
__global__ __launch_bounds__(256) void zero2d_bytes_kernel(unsigned char* __restrict__ p, size_t pitch, size_t width, size_t height) {
    const size_t n = width * height;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t r = i / width, c = i - r * width;
        p[r * pitch + c] = 0;
    }
}
static int g_memset_kernel = -1;
static bool memset_by_kernel() {
    if (g_memset_kernel < 0) {
        const char* e = getenv("TTMI_MEMSET_KERNEL");
        g_memset_kernel = e ? atoi(e) : 1;
    }
    return g_memset_kernel != 0;
}

int fill_zero(void* p, size_t bytes, hipStream_t st) {
    if (bytes == 0) return TTMI_OK;
    TTMI_REQUIRE(p, "fill_zero: null pointer");
    if (memset_by_kernel()) {
        const uintptr_t a = reinterpret_cast<uintptr_t>(p);
        size_t head = (16 - (a & 15)) & 15;
        if (head > bytes) head = bytes;
        const size_t n16 = (bytes - head) / 16, tail = bytes - head - n16 * 16;
        size_t nb = (n16 + 255) / 256;
        if (nb < 1) nb = 1;
        if (nb > 8192) nb = 8192;
        hipLaunchKernelGGL(zero_bytes_kernel, dim3((unsigned)nb), dim3(256), 0, st, static_cast<char*>(p), head, n16, tail);
        TTMI_LAUNCH_CHECK("zero_bytes_kernel");
        return TTMI_OK;
    }
    hipError_t e = hipMemsetAsync(p, 0, bytes, st);
    if (e != hipSuccess) {
        ttmi_set_error("fill_zero: %s", hipGetErrorString(e));
        return (int)e;
    }
    return TTMI_OK;
}

// `height` rows of `width` bytes, `pitch` bytes apart
int fill_zero2d(void* p, size_t pitch, size_t width, size_t height, hipStream_t st) {
    if (width == 0 || height == 0) return TTMI_OK;
    TTMI_REQUIRE(p, "fill_zero2d: null pointer");
    if (memset_by_kernel()) {                        // never a runtime memset NODE unless TTMI_MEMSET_KERNEL=0 asks for it (unordered under graph replay)
        const bool even = pitch % 2 == 0 && width % 2 == 0 && (reinterpret_cast<uintptr_t>(p) & 1) == 0;
        const size_t n = (even ? width / 2 : width) * height;
        size_t nb = (n + 255) / 256;
        if (nb > 8192) nb = 8192;
        if (even) hipLaunchKernelGGL(zero2d_kernel, dim3((unsigned)nb), dim3(256), 0, st, static_cast<unsigned short*>(p), pitch / 2, width / 2, height);
        else hipLaunchKernelGGL(zero2d_bytes_kernel, dim3((unsigned)nb), dim3(256), 0, st, static_cast<unsigned char*>(p), pitch, width, height);
        TTMI_LAUNCH_CHECK("zero2d_kernel");
        return TTMI_OK;
    }
    hipError_t e = hipMemset2DAsync(p, pitch, 0, width, height, st);
    if (e != hipSuccess) {
        ttmi_set_error("fill_zero2d: %s", hipGetErrorString(e));
        return (int)e;
    }
    return TTMI_OK;
}

int bf16_residual(const float* src, const bf16_t* hi, bf16_t* lo, long n, hipStream_t st) {
    TTMI_REQUIRE(src && hi && lo && n > 0, "bf16_residual: bad arguments");
    hipLaunchKernelGGL(bf16_residual_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, src, hi, lo, n);
    TTMI_LAUNCH_CHECK("bf16_residual_kernel");
    return TTMI_OK;
}

int bf16_residual_f32(const float* src, float* r, long n, hipStream_t st) {
    TTMI_REQUIRE(src && r && n > 0, "bf16_residual_f32: bad arguments");
    hipLaunchKernelGGL(bf16_residual_f32_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, src, r, n);
    TTMI_LAUNCH_CHECK("bf16_residual_f32_kernel");
    return TTMI_OK;
}

int convert_bf16(const float* src, bf16_t* dst, long n, hipStream_t st) {
    TTMI_REQUIRE(src && dst && n > 0 && aligned16(src) && (reinterpret_cast<uintptr_t>(dst) & 7) == 0, "convert_bf16: bad arguments");
    hipLaunchKernelGGL(convert_bf16_kernel, dim3(cdiv((n + 3) / 4, 256)), dim3(256), 0, st, src, dst, n);
    TTMI_LAUNCH_CHECK("convert_bf16_kernel");
    return TTMI_OK;
}

__global__ __launch_bounds__(256) void relu_mask_scale_kernel(float* __restrict__ g, const float* __restrict__ a, long n, float scale) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) g[i] = a[i] > 0.f ? g[i] * scale : 0.f;
}
// g[i] = a[i] > 0 ? g[i] * scale : 0 (the generic f32 GEMM's GEMM_MASK_AUX epilogue as a pass of its own: bf16x3 mode's FFN dgrad)
int relu_mask_scale(float* g, const float* a, long n, float scale, hipStream_t st) {
    TTMI_REQUIRE(g && a && n > 0, "relu_mask_scale: bad arguments");
    long nb = (n + 255) / 256;
    if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(relu_mask_scale_kernel, dim3((unsigned)nb), dim3(256), 0, st, g, a, n, scale);
    TTMI_LAUNCH_CHECK("relu_mask_scale_kernel");
    return TTMI_OK;
}

int split3_bf16(const float* src, long ld, long R, int C, int Cp, int hhl, bf16_t* dst, hipStream_t st) {
    TTMI_REQUIRE(src && dst && R > 0 && C > 0 && Cp >= C && Cp % 8 == 0 && ld >= C && aligned16(dst), "split3_bf16: bad arguments");
    long nb = (R * (Cp / 4) + 255) / 256;
    if (nb > 65536) nb = 65536;
    hipLaunchKernelGGL(split3_kernel, dim3((unsigned)nb), dim3(256), 0, st, src, ld, R, C, Cp, hhl, dst);
    TTMI_LAUNCH_CHECK("split3_kernel");
    return TTMI_OK;
}
int split3_transpose_bf16(const float* src, long ld, int R, int C, int Rp, bool hhl, bf16_t* dst, hipStream_t st) {
    TTMI_REQUIRE(src && dst && R > 0 && C > 0 && Rp >= R && Rp % 8 == 0 && ld >= C && aligned16(dst), "split3_transpose_bf16: bad arguments");
    hipLaunchKernelGGL(split3_transpose_kernel, dim3(cdiv(C, 32), cdiv(Rp, 32)), dim3(256), 0, st, src, ld, R, C, Rp, hhl ? 1 : 0, dst);
    TTMI_LAUNCH_CHECK("split3_transpose_kernel");
    return TTMI_OK;
}

int transpose_convert_bf16(const float* src, int R, int C, bf16_t* dst, long ldd, hipStream_t st, bf16_t* plain) {
    TTMI_REQUIRE(src && dst && R > 0 && C > 0 && ldd >= R, "transpose_convert_bf16: bad arguments");
    hipLaunchKernelGGL(transpose_convert_kernel, dim3(cdiv(C, 32), cdiv(ldd, 32)), dim3(256), 0, st, src, R, C, dst, ldd, plain);
    TTMI_LAUNCH_CHECK("transpose_convert_kernel");
    return TTMI_OK;
}

int shadow_refresh(const long* table, int n, long total_tiles, hipStream_t st, long lo_delta) {
    TTMI_REQUIRE(table && n > 0 && total_tiles > 0 && total_tiles < (1L << 31) && lo_delta >= 0, "shadow_refresh: bad arguments");
    hipLaunchKernelGGL(shadow_refresh_kernel, dim3((unsigned)total_tiles), dim3(256), 0, st, table, n, lo_delta);
    TTMI_LAUNCH_CHECK("shadow_refresh_kernel");
    return TTMI_OK;
}

int colsum_bf16(const bf16_t* in, long ld, long rows, int cols, float* out, hipStream_t st, int nz1, int nz2, long si1, long si2,
                long so2) {
    TTMI_REQUIRE(in && out && rows > 0 && cols > 0 && (reinterpret_cast<uintptr_t>(in) & 3) == 0, "colsum_bf16: bad arguments");
    dim3 grid(cdiv(cols, 512), cdiv(rows, CSB_ROWS), nz1 * nz2);
    TTMI_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "colsum_bf16: grid too large");
    hipLaunchKernelGGL(colsum_bf16_kernel, grid, dim3(256), 0, st, in, ld, rows, cols, out, nz2, si1, si2, so2);
    TTMI_LAUNCH_CHECK("colsum_bf16_kernel");
    return TTMI_OK;
}

int add_row_bias_bf16(const bf16_t* in, long ldi, const float* bias, long rows, int cols, bf16_t* out, long ldo, hipStream_t st) {
    TTMI_REQUIRE(in && bias && out && rows > 0 && cols > 0, "add_row_bias_bf16: bad arguments");
    hipLaunchKernelGGL(add_row_bias_bf16_kernel, dim3(cdiv(rows * cols, 256)), dim3(256), 0, st, in, ldi, bias, rows, cols, out,
                       ldo);
    TTMI_LAUNCH_CHECK("add_row_bias_bf16_kernel");
    return TTMI_OK;
}

int dropout_apply(const float* in, long n, DropSpec ds, float* out32, bf16_t* out16, hipStream_t st) {
    TTMI_REQUIRE(in && (out32 || out16) && n > 0, "dropout_apply: bad arguments");
    hipLaunchKernelGGL(dropout_apply_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, in, n, ds, out32, out16);
    TTMI_LAUNCH_CHECK("dropout_apply_kernel");
    return TTMI_OK;
}

int greedy_scan(const void* logits, int dtype, long ld, int n, int V, int blank, unsigned long long* out, hipStream_t st) {
    TTMI_REQUIRE(logits && out && n > 0 && V > 0 && ld >= V, "greedy_scan: bad arguments");
    const unsigned long long none = ((unsigned long long)n << 32);
    hipError_t e = hipMemcpyAsync(out, &none, 8, hipMemcpyHostToDevice, st);      // pageable 8-byte copy: staged immediately
    if (e != hipSuccess) { ttmi_set_error("greedy_scan: %s", hipGetErrorString(e)); return (int)e; }
    if (dtype == 0)
        hipLaunchKernelGGL(greedy_scan_kernel<float>, dim3(cdiv(n, 4)), dim3(256), 0, st, static_cast<const float*>(logits), ld, n, V, blank, out);
    else
        hipLaunchKernelGGL(greedy_scan_kernel<bf16_t>, dim3(cdiv(n, 4)), dim3(256), 0, st, static_cast<const bf16_t*>(logits), ld, n, V, blank, out);
    TTMI_LAUNCH_CHECK("greedy_scan_kernel");
    return TTMI_OK;
}

int transpose_bf16_batched(const void* src, int src_dtype, long ld, int nz1, int nz2, long s1, long s2, int R, int C, bf16_t* dst,
                           long ldd, hipStream_t st) {
    TTMI_REQUIRE(src && dst && R > 0 && C > 0 && ldd >= R && nz1 > 0 && nz2 > 0, "transpose_bf16_batched: bad arguments");
    dim3 grid(cdiv(C, 32), cdiv(ldd, 32), nz1 * nz2);
    TTMI_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "transpose_bf16_batched: grid too large");
    if (src_dtype == 0)
        hipLaunchKernelGGL(transpose_batched_kernel<float>, grid, dim3(256), 0, st, static_cast<const float*>(src), ld, nz2, s1, s2, R, C, dst, ldd);
    else
        hipLaunchKernelGGL(transpose_batched_kernel<bf16_t>, grid, dim3(256), 0, st, static_cast<const bf16_t*>(src), ld, nz2, s1, s2, R, C, dst, ldd);
    TTMI_LAUNCH_CHECK("transpose_batched_kernel");
    return TTMI_OK;
}
