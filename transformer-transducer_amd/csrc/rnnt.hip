// RNN-T loss for gfx950: log-softmax gather, wavefront-diagonal alpha/beta
// lattice, gradient w.r.t. the logits.
//
// Replaces the external `warprnnt_pytorch.RNNTLoss` the reference calls at
// train.py:53,231.
//
//   rnnt_lse_kernel        HBM-bound: one wave per (b,t,u) row of V logits,
//                          16-B loads, online log-sum-exp, wave reduction by
//                          __shfl_xor; emits lse and the two emission
//                          log-probs the lattice needs, stored DIAGONAL-MAJOR
//                          (index (t+u)*U1 + u) so the lattice reads are coalesced.
//   rnnt_alphabeta_kernel  latency-bound dynamic programme: one wave per
//                          (utterance, direction).  Lane l, slot r owns label
//                          u = 64 r + l; the anti-diagonal frontier lives in
//                          registers, the neighbour cell arrives by a one-lane
//                          DPP wave rotate, emission rows are prefetched two
//                          chunks ahead.  No MFMA: it is not a contraction.
//   rnnt_grad_kernel       HBM-bound: one wave per row, reads the logits once
//                          and writes the gradient once (in place allowed).
#include "common.h"

namespace {

constexpr float NEG = -1e30f;
constexpr int LSE_WAVES = 4;

// The frontier is carried in fp64: |alpha| grows to hundreds/thousands, where an fp32 ulp
// (6e-5 at 600) accumulated over T+U steps costs 1e-4 relative in exp(alpha+beta-ll).  Only the
// bounded correction log(1+exp(-|a-b|)) in [0, ln 2] is evaluated in fp32 (v_exp_f32/v_log_f32).
typedef double acc_t;
__device__ __forceinline__ acc_t lae(acc_t a, acc_t b) {
    const acc_t m = a > b ? a : b;
    const float d = -(float)fabs(a - b);
    return m + (acc_t)__logf(1.0f + __expf(d));
}

template <int CTRL>
__device__ __forceinline__ acc_t dpp_rot(acc_t x) {
    const long long v = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(v & 0xffffffffLL), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(v >> 32), CTRL, 0xF, 0xF, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ acc_t rot_r1(acc_t x) { return dpp_rot<0x13C>(x); }   // lane l <- l-1 (0 <- 63): wave_ror:1
__device__ __forceinline__ acc_t rot_l1(acc_t x) { return dpp_rot<0x134>(x); }   // lane l <- l+1 (63 <- 0): wave_rol:1

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// size in floats of one utterance's diagonal-major array
__host__ __device__ __forceinline__ long diag_stride(int T, int U1) { return (long)(T + U1 - 1) * U1; }

// ------------------------------------------------------------------ lse + gather
__global__ __launch_bounds__(LSE_WAVES * 64) void rnnt_lse_kernel(
    const float* __restrict__ logits, const int* __restrict__ labels, const int* __restrict__ act_lens,
    const int* __restrict__ label_lens, int B, int T, int U1, int V, int blank, int vec_ok,
    float* __restrict__ lse, float* __restrict__ lpb_d, float* __restrict__ lpl_d) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * LSE_WAVES + (threadIdx.x >> 6);
    if (row >= (long)B * T * U1) return;
    const int u = (int)(row % U1);
    const long bt = row / U1;
    const int t = (int)(bt % T);
    const int b = (int)(bt / T);
    const int Tb = clampi(act_lens[b], 1, T), Ub = clampi(label_lens[b], 0, U1 - 1);
    if (t >= Tb || u > Ub) return;
    const float* r = logits + row * V;
    float m = NEG, s = 0.f;
    int head = vec_ok ? (int)((4 - ((reinterpret_cast<uintptr_t>(r) >> 2) & 3)) & 3) : V;
    if (head > V) head = V;
    if (lane < head) { m = r[lane]; s = 1.f; }
    for (int i = 64 + lane; i < head; i += 64) {      // only when !vec_ok (scalar path)
        float x = r[i];
        float mn = fmaxf(m, x);
        s = s * __expf(m - mn) + __expf(x - mn);
        m = mn;
    }
    const int nvec = (V - head) >> 2;
    const float4* rv = reinterpret_cast<const float4*>(r + head);
    for (int i = lane; i < nvec; i += 64) {
        float4 x = rv[i];
        float mn = fmaxf(fmaxf(fmaxf(x.x, x.y), fmaxf(x.z, x.w)), m);
        s = s * __expf(m - mn) + (__expf(x.x - mn) + __expf(x.y - mn)) + (__expf(x.z - mn) + __expf(x.w - mn));
        m = mn;
    }
    const int tail0 = head + (nvec << 2);
    if (tail0 + lane < V) {
        float x = r[tail0 + lane];
        float mn = fmaxf(m, x);
        s = s * __expf(m - mn) + __expf(x - mn);
        m = mn;
    }
    const float M = wave_max(m);
    s = wave_sum(s * __expf(m - M));
    if (lane == 0) {
        const float l = M + __logf(s);
        lse[row] = l;
        const long di = (long)b * diag_stride(T, U1) + (long)(t + u) * U1 + u;
        lpb_d[di] = r[blank] - l;
        float pl = NEG;
        if (u < Ub) {
            int y = labels[(long)b * (U1 - 1) + u];
            y = clampi(y, 0, V - 1);
            pl = r[y] - l;
        }
        lpl_d[di] = pl;
    }
}

// ------------------------------------------------------------------ alpha / beta
// PF = emission rows per prefetch chunk (shrinks as the slots-per-lane R grows, to stay in registers)
template <int R, int PF>
struct RowBuf {
    float pb[PF][R];
    float pl[PF][R];
};

// alpha chunk: rows needed for steps d = base .. base+PF-1 are rows d-1
template <int R, int PF>
__device__ __forceinline__ void load_alpha_rows(RowBuf<R, PF>& buf, const float* lpb, const float* lpl, int base, int D,
                                                int U1, int Ub, int lane) {
#pragma unroll
    for (int s = 0; s < PF; ++s) {
        int row = base + s - 1;
        row = row < D - 1 ? row : D - 1;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int u = r * 64 + lane;
            buf.pb[s][r] = (u <= Ub) ? lpb[(long)row * U1 + u] : NEG;
            buf.pl[s][r] = (u <= Ub && u > 0) ? lpl[(long)row * U1 + u - 1] : NEG;
        }
    }
}

template <int R, int PF>
__device__ __forceinline__ void alpha_steps(acc_t (&a)[R], const RowBuf<R, PF>& buf, int base, int D, int U1, int Tb, int Ub,
                                            int lane, acc_t* __restrict__ alpha) {
#pragma unroll
    for (int s = 0; s < PF; ++s) {
        const int d = base + s;
        if (d < D) {   // wave-uniform
            acc_t left[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                acc_t nb = rot_r1(a[r]);
                acc_t wrap = (r > 0) ? rot_r1(a[r > 0 ? r - 1 : 0]) : (acc_t)NEG;
                left[r] = (lane == 0) ? wrap : nb;
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int u = r * 64 + lane;
                const int t = d - u;
                const bool valid = (u <= Ub) && (t >= 0) && (t < Tb);
                const acc_t tt = (t > 0) ? a[r] + (acc_t)buf.pb[s][r] : (acc_t)NEG;
                const acc_t tu = (u > 0) ? left[r] + (acc_t)buf.pl[s][r] : (acc_t)NEG;
                const acc_t v = valid ? lae(tt, tu) : (acc_t)NEG;
                a[r] = v;
                if (valid) alpha[(long)d * U1 + u] = v;
            }
        }
    }
}

// beta chunk walks d downward: steps d = base, base-1, ..., base-PF+1 use row d
template <int R, int PF>
__device__ __forceinline__ void load_beta_rows(RowBuf<R, PF>& buf, const float* lpb, const float* lpl, int base, int U1, int Ub,
                                               int lane) {
#pragma unroll
    for (int s = 0; s < PF; ++s) {
        int row = base - s;
        row = row > 0 ? row : 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int u = r * 64 + lane;
            buf.pb[s][r] = (u <= Ub) ? lpb[(long)row * U1 + u] : NEG;
            buf.pl[s][r] = (u <= Ub) ? lpl[(long)row * U1 + u] : NEG;
        }
    }
}

template <int R, int PF>
__device__ __forceinline__ void beta_steps(acc_t (&bt)[R], const RowBuf<R, PF>& buf, int base, int U1, int Tb, int Ub, int lane,
                                           acc_t* __restrict__ beta) {
#pragma unroll
    for (int s = 0; s < PF; ++s) {
        const int d = base - s;
        if (d >= 0) {   // wave-uniform
            acc_t right[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                acc_t nb = rot_l1(bt[r]);
                acc_t wrap = (r + 1 < R) ? rot_l1(bt[r + 1 < R ? r + 1 : r]) : (acc_t)NEG;
                right[r] = (lane == 63) ? wrap : nb;
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int u = r * 64 + lane;
                const int t = d - u;
                const bool valid = (u <= Ub) && (t >= 0) && (t < Tb);
                const acc_t tt = (t < Tb - 1) ? bt[r] + (acc_t)buf.pb[s][r] : (acc_t)NEG;
                const acc_t tu = (u < Ub) ? right[r] + (acc_t)buf.pl[s][r] : (acc_t)NEG;
                const acc_t v = valid ? lae(tt, tu) : (acc_t)NEG;
                bt[r] = v;
                if (valid) beta[(long)d * U1 + u] = v;
            }
        }
    }
}

template <int R, int PF>
__global__ __launch_bounds__(64) void rnnt_alphabeta_kernel(const float* __restrict__ lpb_d, const float* __restrict__ lpl_d,
                                                            const int* __restrict__ act_lens,
                                                            const int* __restrict__ label_lens, int T, int U1,
                                                            acc_t* __restrict__ alpha_d, acc_t* __restrict__ beta_d,
                                                            acc_t* __restrict__ ll, float* __restrict__ costs) {
    const int b = blockIdx.x >> 1;
    const bool do_beta = blockIdx.x & 1;
    const int lane = threadIdx.x;
    const int Tb = clampi(act_lens[b], 1, T), Ub = clampi(label_lens[b], 0, U1 - 1);
    const int D = Tb + Ub;   // diagonals 0 .. D-1
    const long off = (long)b * diag_stride(T, U1);
    const float* lpb = lpb_d + off;
    const float* lpl = lpl_d + off;
    RowBuf<R, PF> A, Bq;
    if (!do_beta) {
        acc_t* alpha = alpha_d + off;
        acc_t a[R];
#pragma unroll
        for (int r = 0; r < R; ++r) a[r] = (r == 0 && lane == 0) ? 0.0 : (acc_t)NEG;
        if (lane == 0) alpha[0] = 0.0;
        load_alpha_rows<R, PF>(A, lpb, lpl, 1, D, U1, Ub, lane);
        for (int base = 1; base < D; base += 2 * PF) {
            load_alpha_rows<R, PF>(Bq, lpb, lpl, base + PF, D, U1, Ub, lane);
            alpha_steps<R, PF>(a, A, base, D, U1, Tb, Ub, lane, alpha);
            load_alpha_rows<R, PF>(A, lpb, lpl, base + 2 * PF, D, U1, Ub, lane);
            alpha_steps<R, PF>(a, Bq, base + PF, D, U1, Tb, Ub, lane, alpha);
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (r * 64 + lane == Ub) {
                const acc_t v = a[r] + (acc_t)lpb[(long)(D - 1) * U1 + Ub];
                ll[b * 2 + 0] = v;
                costs[b] = (float)(-v);
            }
    } else {
        acc_t* beta = beta_d + off;
        acc_t bt[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool term = (r * 64 + lane == Ub);
            bt[r] = term ? (acc_t)lpb[(long)(D - 1) * U1 + Ub] : (acc_t)NEG;
            if (term) beta[(long)(D - 1) * U1 + Ub] = bt[r];
        }
        load_beta_rows<R, PF>(A, lpb, lpl, D - 2, U1, Ub, lane);
        for (int base = D - 2; base >= 0; base -= 2 * PF) {
            load_beta_rows<R, PF>(Bq, lpb, lpl, base - PF, U1, Ub, lane);
            beta_steps<R, PF>(bt, A, base, U1, Tb, Ub, lane, beta);
            load_beta_rows<R, PF>(A, lpb, lpl, base - 2 * PF, U1, Ub, lane);
            beta_steps<R, PF>(bt, Bq, base - PF, U1, Tb, Ub, lane, beta);
        }
        if (lane == 0) ll[b * 2 + 1] = bt[0];
    }
}

// ------------------------------------------------------------------ gradient
__global__ __launch_bounds__(LSE_WAVES * 64) void rnnt_grad_kernel(
    const float* logits, const int* __restrict__ labels, const int* __restrict__ act_lens,
    const int* __restrict__ label_lens, int B, int T, int U1, int V, int blank, int vec_ok, const float* __restrict__ lse,
    const acc_t* __restrict__ alpha_d, const acc_t* __restrict__ beta_d, const acc_t* __restrict__ ll,
    const float* __restrict__ grad_out, int grad_out_stride, float scale, float* grad) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * LSE_WAVES + (threadIdx.x >> 6);
    if (row >= (long)B * T * U1) return;
    const int u = (int)(row % U1);
    const long bt = row / U1;
    const int t = (int)(bt % T);
    const int b = (int)(bt / T);
    const int Tb = clampi(act_lens[b], 1, T), Ub = clampi(label_lens[b], 0, U1 - 1);
    const bool valid = (t < Tb) && (u <= Ub);
    const float* r = logits + row * V;
    float* g = grad + row * V;
    float c = 0.f, eb = 0.f, el = 0.f, gs = 0.f;
    int yv = -1;
    if (valid) {
        const acc_t* al = alpha_d + (long)b * diag_stride(T, U1);
        const acc_t* be = beta_d + (long)b * diag_stride(T, U1);
        const long di = (long)(t + u) * U1 + u;
        const float l = lse[row];
        const acc_t a = al[di] - ll[b * 2];                                     // alpha - ll, fp64
        c = (float)(a + be[di]) - l;
        gs = scale * grad_out[(long)b * grad_out_stride];
        const float lpb = r[blank] - l;
        if (t == Tb - 1 && u == Ub) eb = __expf((float)a + lpb);
        else if (t < Tb - 1) eb = __expf((float)(a + be[di + U1]) + lpb);       // beta[t+1,u]
        if (u < Ub) {
            yv = clampi(labels[(long)b * (U1 - 1) + u], 0, V - 1);
            el = __expf((float)(a + be[di + U1 + 1]) + (r[yv] - l));            // beta[t,u+1]
        }
    }
    // every lane must have read r[blank], r[yv] before any lane overwrites them (in-place use)
    __builtin_amdgcn_wave_barrier();
    auto f = [&](float x, int v) -> float {
        float e = __expf(x + c);
        e -= (v == blank) ? eb : 0.f;
        e -= (v == yv) ? el : 0.f;
        return valid ? gs * e : 0.f;
    };
    int head = vec_ok ? (int)((4 - ((reinterpret_cast<uintptr_t>(r) >> 2) & 3)) & 3) : V;
    if (head > V) head = V;
    for (int i = lane; i < head; i += 64) g[i] = f(r[i], i);
    const int nvec = (V - head) >> 2;
    const float4* rv = reinterpret_cast<const float4*>(r + head);
    float4* gv = reinterpret_cast<float4*>(g + head);
    for (int i = lane; i < nvec; i += 64) {
        float4 x = valid ? rv[i] : make_float4(0, 0, 0, 0);
        const int v0 = head + (i << 2);
        float4 o;
        o.x = f(x.x, v0);
        o.y = f(x.y, v0 + 1);
        o.z = f(x.z, v0 + 2);
        o.w = f(x.w, v0 + 3);
        gv[i] = o;
    }
    const int tail0 = head + (nvec << 2);
    if (tail0 + lane < V) g[tail0 + lane] = f(r[tail0 + lane], tail0 + lane);
}

template <int R, int PF>
void launch_alphabeta(hipStream_t st, int B, const float* lpb, const float* lpl, const int* al, const int* ll_, int T, int U1,
                      acc_t* a, acc_t* b, acc_t* ll, float* costs) {
    hipLaunchKernelGGL((rnnt_alphabeta_kernel<R, PF>), dim3(2 * B), dim3(64), 0, st, lpb, lpl, al, ll_, T, U1, a, b, ll, costs);
}

struct Ws {
    acc_t *alpha, *beta, *ll;
    float *lse, *lpb, *lpl;
};
Ws carve(void* ws, int B, int T, int U1) {
    const long n = (long)B * T * U1, nd = (long)B * diag_stride(T, U1);
    Ws w;
    acc_t* q = static_cast<acc_t*>(ws);   // fp64 part first (workspace must be 8-byte aligned)
    w.alpha = q; q += nd;
    w.beta = q; q += nd;
    w.ll = q; q += 2 * (long)B;
    float* p = reinterpret_cast<float*>(q);
    w.lse = p; p += n;
    w.lpb = p; p += nd;
    w.lpl = p;
    return w;
}

}  // namespace

extern "C" {

// bytes of caller-provided workspace shared by ttmi_rnnt_loss_fwd / _bwd
size_t ttmi_rnnt_workspace_bytes(int B, int T, int U1) {
    return sizeof(float) * ((size_t)B * T * U1 + 2 * (size_t)B * diag_stride(T, U1)) +
           sizeof(acc_t) * (2 * (size_t)B * diag_stride(T, U1) + 2 * (size_t)B) + 64;
}

// Forward: per-utterance costs[B] = -log P(y|x).  logits f32 [B,T,U1,V] contiguous,
// labels i32 [B,U1-1], act_lens/label_lens i32 [B] (all device pointers).  Fills the
// workspace (lse, alpha, beta, ll) that ttmi_rnnt_loss_bwd consumes.
int ttmi_rnnt_loss_fwd(const float* logits, const int* labels, const int* act_lens, const int* label_lens, int B, int T,
                       int U1, int V, int blank, void* workspace, float* costs, void* stream) {
    TTMI_REQUIRE(logits && (labels || U1 == 1) && act_lens && label_lens && workspace && costs, "rnnt_loss_fwd: null pointer");
    TTMI_REQUIRE(B > 0 && T > 0 && U1 > 0 && V > 0, "rnnt_loss_fwd: bad shape B=%d T=%d U1=%d V=%d", B, T, U1, V);
    TTMI_REQUIRE(blank >= 0 && blank < V, "rnnt_loss_fwd: blank %d outside [0,%d)", blank, V);
    TTMI_REQUIRE(U1 <= 1024, "rnnt_loss_fwd: U+1=%d > 1024 unsupported", U1);
    TTMI_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 7) == 0, "rnnt_loss_fwd: workspace must be 8-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    Ws w = carve(workspace, B, T, U1);
    const long rows = (long)B * T * U1;
    const int vec_ok = aligned16(logits) ? 1 : 0;
    hipLaunchKernelGGL(rnnt_lse_kernel, dim3(cdiv(rows, LSE_WAVES)), dim3(LSE_WAVES * 64), 0, st, logits, labels, act_lens,
                       label_lens, B, T, U1, V, blank, vec_ok, w.lse, w.lpb, w.lpl);
    TTMI_LAUNCH_CHECK("rnnt_lse_kernel");
    if (U1 <= 64) launch_alphabeta<1, 8>(st, B, w.lpb, w.lpl, act_lens, label_lens, T, U1, w.alpha, w.beta, w.ll, costs);
    else if (U1 <= 128) launch_alphabeta<2, 8>(st, B, w.lpb, w.lpl, act_lens, label_lens, T, U1, w.alpha, w.beta, w.ll, costs);
    else if (U1 <= 256) launch_alphabeta<4, 4>(st, B, w.lpb, w.lpl, act_lens, label_lens, T, U1, w.alpha, w.beta, w.ll, costs);
    else if (U1 <= 512) launch_alphabeta<8, 2>(st, B, w.lpb, w.lpl, act_lens, label_lens, T, U1, w.alpha, w.beta, w.ll, costs);
    else launch_alphabeta<16, 1>(st, B, w.lpb, w.lpl, act_lens, label_lens, T, U1, w.alpha, w.beta, w.ll, costs);
    TTMI_LAUNCH_CHECK("rnnt_alphabeta_kernel");
    return TTMI_OK;
}

// Backward: grad[b,t,u,v] = scale * grad_out[b*grad_out_stride] * d cost_b / d logits.
// grad may alias logits (in-place).  Cells outside [0,T_b) x [0,U_b] get exact zeros.
int ttmi_rnnt_loss_bwd(const float* logits, const int* labels, const int* act_lens, const int* label_lens, int B, int T,
                       int U1, int V, int blank, const void* workspace, const float* grad_out, int grad_out_stride,
                       float scale, float* grad, void* stream) {
    TTMI_REQUIRE(logits && (labels || U1 == 1) && act_lens && label_lens && workspace && grad_out && grad,
                 "rnnt_loss_bwd: null pointer");
    TTMI_REQUIRE(B > 0 && T > 0 && U1 > 0 && V > 0, "rnnt_loss_bwd: bad shape");
    TTMI_REQUIRE(blank >= 0 && blank < V, "rnnt_loss_bwd: blank %d outside [0,%d)", blank, V);
    hipStream_t st = static_cast<hipStream_t>(stream);
    Ws w = carve(const_cast<void*>(workspace), B, T, U1);
    const long rows = (long)B * T * U1;
    const int vec_ok = (aligned16(logits) && aligned16(grad)) ? 1 : 0;
    hipLaunchKernelGGL(rnnt_grad_kernel, dim3(cdiv(rows, LSE_WAVES)), dim3(LSE_WAVES * 64), 0, st, logits, labels, act_lens,
                       label_lens, B, T, U1, V, blank, vec_ok, w.lse, w.alpha, w.beta, w.ll, grad_out, grad_out_stride, scale,
                       grad);
    TTMI_LAUNCH_CHECK("rnnt_grad_kernel");
    return TTMI_OK;
}

}  // extern "C"
