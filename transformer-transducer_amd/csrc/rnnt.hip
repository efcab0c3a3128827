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

// ------------------------------------------------------------------ row access helpers (f32 or bf16 logits)
typedef unsigned u32x4n __attribute__((ext_vector_type(4)));
template <typename TL>
struct Vec16 {                                   // one 16-byte access = NV elements
    static constexpr int NV = 16 / sizeof(TL);
    float f[NV];
    template <bool NT = true>
    __device__ __forceinline__ void load(const TL* p) {
        // the logits / gradient stream through once per pass (7 - 56 GB): streaming accesses in the gradient pass (its loads and stores
        // together: 4.19 -> 4.06 ms for the loss op); the log-sum-exp pass reads faster with plain loads (1.38 against 1.53 ms)
        uint4 w;
        if constexpr (NT) {
            const u32x4n wn = __builtin_nontemporal_load(reinterpret_cast<const u32x4n*>(p));
            w = make_uint4(wn.x, wn.y, wn.z, wn.w);
        } else {
            w = *reinterpret_cast<const uint4*>(p);
        }
        if constexpr (sizeof(TL) == 4) {
            f[0] = __uint_as_float(w.x); f[1] = __uint_as_float(w.y); f[2] = __uint_as_float(w.z); f[3] = __uint_as_float(w.w);
        } else {
            const uint32_t u[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f[2 * i] = __uint_as_float(u[i] << 16);
                f[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u);
            }
        }
    }
    __device__ __forceinline__ void store(TL* p) const {
        uint4 w;
        if constexpr (sizeof(TL) == 4) {
            w.x = __float_as_uint(f[0]); w.y = __float_as_uint(f[1]); w.z = __float_as_uint(f[2]); w.w = __float_as_uint(f[3]);
        } else {
            w.x = pack_bf16x2(f[0], f[1]); w.y = pack_bf16x2(f[2], f[3]);
            w.z = pack_bf16x2(f[4], f[5]); w.w = pack_bf16x2(f[6], f[7]);
        }
        __builtin_nontemporal_store(u32x4n{w.x, w.y, w.z, w.w}, reinterpret_cast<u32x4n*>(p));
    }
};
template <typename TL>
__device__ __forceinline__ float ldf(const TL* p) {
    if constexpr (sizeof(TL) == 4) return *p;
    else return bf16_to_f32(*p);
}
template <typename TL>
__device__ __forceinline__ void stf(TL* p, float v) {
    if constexpr (sizeof(TL) == 4) *p = v;
    else *p = f32_to_bf16(v);
}
// elements before the first 16-byte boundary of row pointer r (all V when vectors are not allowed)
template <typename TL>
__device__ __forceinline__ int row_head(const TL* r, int V, int vec_ok) {
    constexpr int NV = 16 / sizeof(TL);
    int head = vec_ok ? (int)((NV - ((reinterpret_cast<uintptr_t>(r) / sizeof(TL)) % NV)) % NV) : V;
    return head > V ? V : head;
}

// ------------------------------------------------------------------ lse + gather
template <typename TL>
__global__ __launch_bounds__(LSE_WAVES * 64) void rnnt_lse_kernel(
    const TL* __restrict__ logits, long ldv, const int* __restrict__ labels, const int* __restrict__ act_lens,
    const int* __restrict__ label_lens, int B, int T, int U1, int V, int blank, int vec_ok,
    float* __restrict__ lse, float* __restrict__ lpb_d, float* __restrict__ lpl_d) {
    constexpr int NV = 16 / sizeof(TL);
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * LSE_WAVES + (threadIdx.x >> 6);
    if (row >= (long)B * T * U1) return;
    const int u = (int)(row % U1);
    const long bt = row / U1;
    const int t = (int)(bt % T);
    const int b = (int)(bt / T);
    const int Tb = clampi(act_lens[b], 1, T), Ub = clampi(label_lens[b], 0, U1 - 1);
    if (t >= Tb || u > Ub) return;
    const TL* r = logits + row * ldv;
    float m = NEG, s = 0.f;
    auto upd = [&](float x) {
        const float mn = fmaxf(m, x);
        s = s * __expf(m - mn) + __expf(x - mn);
        m = mn;
    };
    const int head = row_head<TL>(r, V, vec_ok);
    for (int i = lane; i < head; i += 64) upd(ldf<TL>(r + i));
    const int nvec = (V - head) / NV;
    for (int i = lane; i < nvec; i += 64) {
        Vec16<TL> x;
        x.template load<false>(r + head + i * NV);
        float mx = x.f[0];
#pragma unroll
        for (int k = 1; k < NV; ++k) mx = fmaxf(mx, x.f[k]);
        const float mn = fmaxf(m, mx);
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) acc += __expf(x.f[k] - mn);
        s = s * __expf(m - mn) + acc;
        m = mn;
    }
    for (int i = head + nvec * NV + lane; i < V; i += 64) upd(ldf<TL>(r + i));
    const float M = wave_max(m);
    s = wave_sum(s * __expf(m - M));
    if (lane == 0) {
        const float l = M + __logf(s);
        lse[row] = l;
        const long di = (long)b * diag_stride(T, U1) + (long)(t + u) * U1 + u;
        lpb_d[di] = ldf<TL>(r + blank) - l;
        float pl = NEG;
        if (u < Ub) {
            int y = labels[(long)b * (U1 - 1) + u];
            y = clampi(y, 0, V - 1);
            pl = ldf<TL>(r + y) - l;
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


// ------------------------------------------------------------------ alpha / beta, one workgroup per (utterance, direction)
// The serial chain of the recursion is one log-add-exp per diagonal; everything else is kept off it:
//   * one COMPUTE wave per 64 labels (lane = label u), all walking the same diagonals: the frontier cell of the neighbouring wave (label
//     64 w - 1 for alpha, 64 (w + 1) for beta) crosses through an LDS ring of 64 slots.  A slot is ONE 16-byte {value, step tag} granule,
//     written by one ds_write_b128 of one lane and read by one ds_read_b128 (a lane's 16 bytes move in one LDS cycle), so there is no
//     separate flag and no barrier per diagonal; the read for step g + 1 is issued at the top of step g and has landed when it is needed
//     (a consumer that once had to spin stays one step behind its producer).  U + 1 = 201 is 4 waves side by side instead of 4 slots
//     walked serially by one wave;
//   * a HELPER wave does all the global traffic: the emission rows of chunk c + 1 (CH diagonals = ONE contiguous span of the diagonal-major
//     arrays) go straight into LDS by LDS-DMA while the compute waves walk chunk c out of the other buffer, and the alpha / beta rows of
//     chunk c - 1, which the compute waves left in an LDS ring, are streamed out as whole rows.  The compute waves issue no vector-memory
//     instruction inside the loop (the one-wave kernel waited on vmcnt for its own prefetch loads and cell stores every few steps), and
//     their LDS reads of step s + 1 are issued before the arithmetic of step s;
//   * one workgroup barrier per CH diagonals swaps the buffers.
// Frontier in fp64 as before (lae).  Cells outside the ragged lattice are written as -1e30 (they are never read).
constexpr int LAT_NB = 64;            // boundary ring slots; > CH + 1 (a neighbour runs at most one chunk ahead between two barriers)
struct __attribute__((aligned(16))) LatSlot {
    acc_t v;
    int tag;        // steps completed when v was written (g + 1); slot LAT_NB - 1 starts as {initial cell, 0}
    int pad;
};
static_assert(sizeof(LatSlot) == 16 && alignof(LatSlot) == 16, "a boundary slot is one 16-byte LDS granule: {value, tag} travel in one ds_read/write_b128");
// byte offset of the boundary ring inside the kernel's dynamic LDS (launch-time check: must be a multiple of 16)
constexpr size_t lat_ring_offset(int CHR, int CH, int U1) { return (size_t)4 * CHR * 4 + (size_t)2 * CH * U1 * 8; }
__device__ __forceinline__ void lat_glds4(const void* gsrc, float* lds_wave_base) {       // 4 bytes per lane: LDS address = base + lane * 4
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}
typedef int lat_i4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ LatSlot lat_slot_read(const LatSlot* p) {          // one ds_read_b128, never split or cached by the compiler
    const lat_i4 w = *(const volatile __attribute__((address_space(3))) lat_i4*)(p);       // (a generic volatile access would be a flat_load)
    LatSlot r;
    r.v = __longlong_as_double(((long long)w.y << 32) | (unsigned)w.x);
    r.tag = w.z;
    r.pad = 0;
    return r;
}
__device__ __forceinline__ void lat_slot_write(LatSlot* p, acc_t v, int tag) {
    const long long b = __double_as_longlong(v);
    lat_i4 w;
    w.x = (int)(b & 0xffffffffLL); w.y = (int)(b >> 32); w.z = tag; w.w = 0;
    *(volatile __attribute__((address_space(3))) lat_i4*)(p) = w;
}

template <bool BETA>
__device__ __forceinline__ void lat_walk(const float* __restrict__ Eb, const float* __restrict__ El, acc_t* __restrict__ A, LatSlot* __restrict__ bnd,
                                         int CHR, int CH, int U1, int W, int wave, int lane, int Tb, int Ub, int D, int nsteps, int NC,
                                         acc_t& a) {
    const int u = wave * 64 + lane;
    const int uc = u < U1 ? u : U1 - 1;                       // in-range column for the (unconditional) LDS reads
    const int ul = BETA ? uc : (uc > 0 ? uc - 1 : 0);         // label log-prob column: alpha reads the cell on the left
    // neighbour hand-off: alpha takes the cell of the wave on the left (its lane 63) into lane 0, beta the one on the right into lane 63
    const bool consume = BETA ? (wave + 1 < W) : (wave > 0);
    const bool produce = BETA ? (wave > 0) : (wave + 1 < W);
    const int edge = BETA ? 63 : 0, pedge = BETA ? 0 : 63;
    const LatSlot* cring = bnd + (BETA ? wave + 1 : wave - 1) * LAT_NB;
    LatSlot* pring = bnd + wave * LAT_NB;
    LatSlot pre;                                              // the neighbour's cell for the NEXT step, read one step ahead
    pre.v = (acc_t)NEG; pre.tag = 0; pre.pad = 0;
    if (consume) pre = lat_slot_read(cring + LAT_NB - 1);     // step 0 uses the initial cell (tag 0)
    for (int c = 0; c < NC; ++c) {
        const int n = min(CH, nsteps - c * CH);
        const float* eb = Eb + (c & 1) * CHR;
        const float* el = El + (c & 1) * CHR;
        acc_t* ao = A + (long)(c & 1) * CH * U1;
        // rows of the chunk sit in memory order; alpha walks them upwards (row of step s = s), beta downwards (n - 1 - s)
        int r = BETA ? n - 1 : 0;
        float pb = eb[r * U1 + uc], pl = el[r * U1 + ul];
        for (int s = 0; s < n; ++s) {
            const int g = c * CH + s;
            // issue the LDS reads of the next step before this step's arithmetic: emission row and the neighbour's next cell
            const int rn = BETA ? (s + 1 < n ? r - 1 : r) : (s + 1 < n ? r + 1 : r);
            const float pb_n = eb[rn * U1 + uc], pl_n = el[rn * U1 + ul];
            LatSlot nxt;
            nxt.v = (acc_t)NEG; nxt.tag = 0; nxt.pad = 0;
            if (consume) nxt = lat_slot_read(cring + (g & (LAT_NB - 1)));        // wanted at step g + 1 with tag g + 1
            acc_t nb = BETA ? rot_l1(a) : rot_r1(a);
            if (lane == edge) nb = (acc_t)NEG;
            if (consume) {
                while (pre.tag != g) {                        // not published yet (rare after the first steps): poll the slot of step g - 1
                    __builtin_amdgcn_s_sleep(1);
                    pre = lat_slot_read(cring + ((g + LAT_NB - 1) & (LAT_NB - 1)));
                }
                if (lane == edge) nb = pre.v;
            }
            const int d = BETA ? D - 2 - g : 1 + g;
            const int t = d - u;
            const bool valid = (u <= Ub) && (t >= 0) && (t < Tb);
            const bool has_t = BETA ? (t < Tb - 1) : (t > 0);
            const bool has_u = BETA ? (u < Ub) : (u > 0);
            const acc_t tt = has_t ? a + (acc_t)pb : (acc_t)NEG;
            const acc_t tu = has_u ? nb + (acc_t)pl : (acc_t)NEG;
            a = valid ? lae(tt, tu) : (acc_t)NEG;
            if (u < U1) ao[r * U1 + u] = a;
            if (produce && lane == pedge) lat_slot_write(pring + (g & (LAT_NB - 1)), a, g + 1);
            pb = pb_n; pl = pl_n; pre = nxt; r = rn;
            // a use of the prefetched registers at the END of the step: the compiler's wait for them lands here, counted (the two LDS
            // writes above stay in flight), after a whole step of arithmetic - at the first use in the next step it would wait for everything
            {
                int lo = (int)(__double_as_longlong(pre.v) & 0xffffffffLL), hi = (int)(__double_as_longlong(pre.v) >> 32);
                asm volatile("" : "+v"(pb), "+v"(pl), "+v"(lo), "+v"(hi), "+v"(pre.tag));
                pre.v = __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void rnnt_lattice_lds_kernel(const float* __restrict__ lpb_d, const float* __restrict__ lpl_d,
                                                                const int* __restrict__ act_lens, const int* __restrict__ label_lens, int T,
                                                                int U1, int W, int CH, acc_t* __restrict__ alpha_d,
                                                                acc_t* __restrict__ beta_d, acc_t* __restrict__ ll, float* __restrict__ costs) {
    extern __shared__ __attribute__((aligned(16))) char lat_smem[];
    const int b = blockIdx.x >> 1;
    const bool do_beta = blockIdx.x & 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool helper = wave == W;
    const int Tb = clampi(act_lens[b], 1, T), Ub = clampi(label_lens[b], 0, U1 - 1);
    const int D = Tb + Ub;                       // diagonals 0 .. D-1
    const long off = (long)b * diag_stride(T, U1);
    const float* lpb = lpb_d + off;
    const float* lpl = lpl_d + off;
    acc_t* out = (do_beta ? beta_d : alpha_d) + off;
    const int CHR = (CH * U1 + 63) / 64 * 64;   // floats per emission buffer (whole 64-float LDS-DMA pieces)
    float* Eb = reinterpret_cast<float*>(lat_smem);      // [2][CHR] blank log-probs of the chunk's diagonals
    float* El = Eb + 2 * CHR;                            // [2][CHR] label log-probs
    acc_t* A = reinterpret_cast<acc_t*>(El + 2 * CHR);   // [2][CH * U1] the chunk's alpha / beta rows
    // [W][LAT_NB].  16-byte aligned by construction: the emission buffers are 4 CHR floats (CHR a multiple of 64) and A holds 2 CH U1 doubles
    // = a multiple of 16 bytes whatever the parity of CH U1.  (Round 3 added ((CH U1) & 1) doubles here, which moved the ring 8 bytes OFF
    // alignment whenever CH U1 is odd - the default C5 shape, U1 = 201, CH = 19 - and with it split every {value, tag} granule the tag
    // protocol needs to move as one ds_read_b128 / ds_write_b128.)
    LatSlot* bnd = reinterpret_cast<LatSlot*>(A + 2 * (long)CH * U1);
    const int nsteps = D - 1;
    const int NC = (nsteps + CH - 1) / CH;
    // chunk c = steps c CH .. c CH + n - 1.  alpha: step g forms diagonal 1 + g from emission row g; beta: step g forms diagonal D - 2 - g
    // from emission row D - 2 - g.  row0 = lowest emission row of the chunk (its rows are contiguous in memory either way).
    auto chunk = [&](int c, int& row0, int& n) {
        n = min(CH, nsteps - c * CH);
        row0 = do_beta ? (D - 2 - c * CH - (n - 1)) : c * CH;
    };
    auto issue_loads = [&](int c) {              // helper wave: emission rows of chunk c -> LDS (asynchronous)
        int row0, n;
        chunk(c, row0, n);
        const int cnt = n * U1;
        const float* gb = lpb + (long)row0 * U1;
        const float* gl = lpl + (long)row0 * U1;
        float* eb = Eb + (c & 1) * CHR;
        float* el = El + (c & 1) * CHR;
        for (int i = 0; i < cnt; i += 64) {
            const int k = min(i + lane, cnt - 1);          // the last piece re-reads the span's last element instead of running past it
            lat_glds4(gb + k, eb + i);
            lat_glds4(gl + k, el + i);
        }
    };
    auto drain = [&](int c) {                    // helper wave: the chunk's alpha / beta rows, LDS -> global, whole rows
        int row0, n;
        chunk(c, row0, n);
        const int cnt = n * U1;
        const acc_t* src = A + (long)(c & 1) * CH * U1;
        acc_t* dst = out + (long)(do_beta ? row0 : row0 + 1) * U1;
        int i = lane;
        for (; i + 192 < cnt; i += 256) {        // four reads in flight per wait
            const acc_t v0 = src[i], v1 = src[i + 64], v2 = src[i + 128], v3 = src[i + 192];
            dst[i] = v0; dst[i + 64] = v1; dst[i + 128] = v2; dst[i + 192] = v3;
        }
        for (; i < cnt; i += 64) dst[i] = src[i];
    };
    const int u = wave * 64 + lane;
    acc_t a = (acc_t)NEG;                        // this lane's frontier cell
    if (!helper) {
        if (!do_beta) {
            a = (u == 0) ? 0.0 : (acc_t)NEG;
            if (u == 0) out[0] = 0.0;
            if (wave + 1 < W && lane == 63) lat_slot_write(bnd + wave * LAT_NB + LAT_NB - 1, a, 0);
        } else {
            const bool term = (u == Ub);
            if (term) {
                a = (acc_t)lpb[(long)(D - 1) * U1 + Ub];
                out[(long)(D - 1) * U1 + Ub] = a;
            }
            if (wave > 0 && lane == 0) lat_slot_write(bnd + wave * LAT_NB + LAT_NB - 1, a, 0);
        }
    } else if (NC > 0) {
        issue_loads(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (helper) {
        for (int c = 0; c < NC; ++c) {
            if (c + 1 < NC) issue_loads(c + 1);
            if (c > 0) drain(c - 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (NC > 0) drain(NC - 1);
        return;
    }
    if (!do_beta) {
        lat_walk<false>(Eb, El, A, bnd, CHR, CH, U1, W, wave, lane, Tb, Ub, D, nsteps, NC, a);
        if (u == Ub) {
            const acc_t v = a + (acc_t)lpb[(long)(D - 1) * U1 + Ub];
            ll[b * 2 + 0] = v;
            costs[b] = (float)(-v);
        }
    } else {
        lat_walk<true>(Eb, El, A, bnd, CHR, CH, U1, W, wave, lane, Tb, Ub, D, nsteps, NC, a);
        if (u == 0) ll[b * 2 + 1] = a;
    }
}

// ------------------------------------------------------------------ gradient
template <typename TL>
__global__ __launch_bounds__(LSE_WAVES * 64) void rnnt_grad_kernel(
    const TL* logits, long ldv, const int* __restrict__ labels, const int* __restrict__ act_lens,
    const int* __restrict__ label_lens, int B, int T, int U1, int V, int blank, int vec_ok, const float* __restrict__ lse,
    const acc_t* __restrict__ alpha_d, const acc_t* __restrict__ beta_d, const acc_t* __restrict__ ll,
    const float* __restrict__ grad_out, int grad_out_stride, float scale, TL* grad, long ldg) {
    constexpr int NV = 16 / sizeof(TL);
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * LSE_WAVES + (threadIdx.x >> 6);
    if (row >= (long)B * T * U1) return;
    const int u = (int)(row % U1);
    const long bt = row / U1;
    const int t = (int)(bt % T);
    const int b = (int)(bt / T);
    const int Tb = clampi(act_lens[b], 1, T), Ub = clampi(label_lens[b], 0, U1 - 1);
    const bool valid = (t < Tb) && (u <= Ub);
    const TL* r = logits + row * ldv;
    TL* g = grad + row * ldg;
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
        const float lpb = ldf<TL>(r + blank) - l;
        if (t == Tb - 1 && u == Ub) eb = __expf((float)a + lpb);
        else if (t < Tb - 1) eb = __expf((float)(a + be[di + U1]) + lpb);       // beta[t+1,u]
        if (u < Ub) {
            yv = clampi(labels[(long)b * (U1 - 1) + u], 0, V - 1);
            el = __expf((float)(a + be[di + U1 + 1]) + (ldf<TL>(r + yv) - l));  // beta[t,u+1]
        }
    }
    // every lane has read r[blank], r[yv] before any lane overwrites them (in-place use)
    __builtin_amdgcn_wave_barrier();
    auto f = [&](float x, int v) -> float {
        float e = __expf(x + c);
        e -= (v == blank) ? eb : 0.f;
        e -= (v == yv) ? el : 0.f;
        return valid ? gs * e : 0.f;
    };
    const int head = row_head<TL>(r, V, vec_ok);
    for (int i = lane; i < head; i += 64) stf<TL>(g + i, f(valid ? ldf<TL>(r + i) : 0.f, i));
    const int nvec = (V - head) / NV;
    for (int i = lane; i < nvec; i += 64) {
        Vec16<TL> x;
        const int v0 = head + i * NV;
        if (valid) x.load(r + v0);
#pragma unroll
        for (int k = 0; k < NV; ++k) x.f[k] = f(valid ? x.f[k] : 0.f, v0 + k);
        x.store(g + v0);
    }
    for (int i = head + nvec * NV + lane; i < V; i += 64) stf<TL>(g + i, f(valid ? ldf<TL>(r + i) : 0.f, i));
    // padded pitch: columns [V, ldg) are zeroed so that the joint's dgrad GEMM may run over the full pitch
    for (int i = V + lane; i < ldg; i += 64) stf<TL>(g + i, 0.f);
}


// bf16x3 mode (round 6): the gradient of f32 logits written IN PLACE as the two bf16 planes the joint's backward multiplies - row r becomes [hi(0 .. ldv) | lo(0 .. ldv)],
// hi = bf16(g), lo = bf16(g - hi): 2 ldv bf16 in the bytes of ldv f32 - so that nothing has to read the 14 GB of d logits again only to split them (5.3 ms per C2 step).
// One wave per row; the row is read completely (KV 16-byte vectors per lane, unconditional loads of a clamped index: all in flight) and the wave waits for ALL of its
// loads before its first store: the planes overwrite columns other lanes have not consumed yet.  Columns [V, ldv) leave as zeros in both planes.  ldv % 4 == 0, rows
// 16-byte aligned, ldv <= 256 KV.
template <int KV>
__global__ __launch_bounds__(LSE_WAVES * 64) void rnnt_grad_split_kernel(
    float* logits, long ldv, const int* __restrict__ labels, const int* __restrict__ act_lens, const int* __restrict__ label_lens, int B, int T,
    int U1, int V, int blank, const float* __restrict__ lse, const acc_t* __restrict__ alpha_d, const acc_t* __restrict__ beta_d,
    const acc_t* __restrict__ ll, const float* __restrict__ grad_out, int grad_out_stride, float scale) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * LSE_WAVES + (threadIdx.x >> 6);
    if (row >= (long)B * T * U1) return;
    const int u = (int)(row % U1);
    const long bt = row / U1;
    const int t = (int)(bt % T);
    const int b = (int)(bt / T);
    const int Tb = clampi(act_lens[b], 1, T), Ub = clampi(label_lens[b], 0, U1 - 1);
    const bool valid = (t < Tb) && (u <= Ub);
    float* r = logits + row * ldv;
    const int nvec = (int)(ldv >> 2);
    float4 x[KV];
#pragma unroll
    for (int k = 0; k < KV; ++k) x[k] = *reinterpret_cast<const float4*>(r + 4 * min(lane + 64 * k, nvec - 1));
    float c = 0.f, eb = 0.f, el = 0.f, gs = 0.f;
    int yv = -1;
    if (valid) {
        const acc_t* al = alpha_d + (long)b * diag_stride(T, U1);
        const acc_t* be = beta_d + (long)b * diag_stride(T, U1);
        const long di = (long)(t + u) * U1 + u;
        const float l = lse[row];
        const acc_t a = al[di] - ll[b * 2];
        c = (float)(a + be[di]) - l;
        gs = scale * grad_out[(long)b * grad_out_stride];
        const float lpb = r[blank] - l;
        if (t == Tb - 1 && u == Ub) eb = __expf((float)a + lpb);
        else if (t < Tb - 1) eb = __expf((float)(a + be[di + U1]) + lpb);
        if (u < Ub) {
            yv = clampi(labels[(long)b * (U1 - 1) + u], 0, V - 1);
            el = __expf((float)(a + be[di + U1 + 1]) + (r[yv] - l));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the whole row (and r[blank], r[yv]) is in registers before any lane overwrites a byte of it
    __builtin_amdgcn_wave_barrier();
    auto f = [&](float v, int col) -> float {
        float e = __expf(v + c);
        e -= (col == blank) ? eb : 0.f;
        e -= (col == yv) ? el : 0.f;
        return (valid && col < V) ? gs * e : 0.f;
    };
    bf16_t* hi = reinterpret_cast<bf16_t*>(r);
    bf16_t* lo = hi + ldv;
#pragma unroll
    for (int k = 0; k < KV; ++k) {
        const int vi = lane + 64 * k;
        if (vi < nvec) {
            const float g[4] = {f(x[k].x, 4 * vi), f(x[k].y, 4 * vi + 1), f(x[k].z, 4 * vi + 2), f(x[k].w, 4 * vi + 3)};
            bf16_t h[4];
            uint2 wh, wl;
#pragma unroll
            for (int j = 0; j < 4; ++j) h[j] = f32_to_bf16(g[j]);
            wh.x = (unsigned)h[0] | ((unsigned)h[1] << 16);
            wh.y = (unsigned)h[2] | ((unsigned)h[3] << 16);
            wl.x = (unsigned)f32_to_bf16(g[0] - bf16_to_f32(h[0])) | ((unsigned)f32_to_bf16(g[1] - bf16_to_f32(h[1])) << 16);
            wl.y = (unsigned)f32_to_bf16(g[2] - bf16_to_f32(h[2])) | ((unsigned)f32_to_bf16(g[3] - bf16_to_f32(h[3])) << 16);
            *reinterpret_cast<uint2*>(hi + 4 * vi) = wh;
            *reinterpret_cast<uint2*>(lo + 4 * vi) = wl;
        }
    }
}

// ------------------------------------------------------------------ exp-domain variant (fused joint + loss fast path)
// The joint's projection GEMM stored P[row, v] = exp(logit - shift) in bf16 plus per-row partial sums of the unrounded values
// (gemm_fast.hip, "exp store" epilogue).  Nothing below walks the rows: the forward needs S = sum of the partials and two entries
// per row, the backward leaves the row factor s_r = gscale * exp(alpha + beta - ll - log S) to the consuming GEMMs
// (d logits[r, v] = s_r * P[r, v] everywhere but at the blank / label columns, which are patched in P here).
constexpr float EXP_SHIFT_MARGIN = 40.f;
// emis (nullable): f32 [rows, 4] from ttmi_joint_fwd_exp - the logits of the blank and of the row's next label, first as the GEMM formed them
// (bf16 operands), then from f32 operands: the emission log-probs are logit_f32 - shift - log S', S' = S with those two columns' terms exchanged.
// flag (nullable, device int): bit 0 is set when a row of the lattice lost its sum - every exp(logit - shift) underflowed, or the sum overflowed:
// the shift in use no longer fits the logits.  Such a row's log-sum-exp becomes NaN, so the utterance's cost and every gradient of the step are
// NaN rather than finite and wrong; callers drop the step and re-seed the shift (ttmi_rnnt_shift_seed).
__global__ __launch_bounds__(256) void rnnt_prep_exp_kernel(
    const bf16_t* __restrict__ P, long ldv, const float* __restrict__ rowsum, int nparts, const int* __restrict__ labels,
    const int* __restrict__ act_lens, const int* __restrict__ label_lens, int B, int T, int U1, int V, int blank,
    float* __restrict__ lse, float* __restrict__ lpb_d, float* __restrict__ lpl_d, const float* __restrict__ shift_cur,
    float* __restrict__ shift_next, const float* __restrict__ emis, int* __restrict__ flag) {
    const long row = (long)blockIdx.x * 256 + threadIdx.x;
    if (row >= (long)B * T * U1) return;
    const int u = (int)(row % U1);
    const long bt = row / U1;
    const int t = (int)(bt % T);
    const int b = (int)(bt / T);
    const int Tb = clampi(act_lens[b], 1, T), Ub = clampi(label_lens[b], 0, U1 - 1);
    if (t >= Tb || u > Ub) return;
    const long rows = (long)B * T * U1;
    float S = 0.f;
    {   // part-major: coalesced across the block's rows.  Eight partial sums in flight per row (a plain runtime loop issued one load per wait); the sum is
        // taken in the same order as before: ((((0 + p0) + p1) + ...) - the loads are hoisted, not the adds
        const float* rp = rowsum + row;
        int i = 0;
        for (; i + 8 <= nparts; i += 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = rp[(long)(i + k) * rows];
#pragma unroll
            for (int k = 0; k < 8; ++k) S += v[k];
        }
        for (; i < nparts; ++i) S += rp[(long)i * rows];
    }
    const float cur = shift_cur ? *shift_cur : 0.f;
    float zb = 0.f, zy = 0.f;
    if (emis) {
        // the two columns the loss reads (blank, next label) enter the row's softmax with their f32-operand logits: their terms of the sum,
        // which the GEMM formed from bf16 operands, are exchanged for the accurate ones.  The bf16 rounding of the projection weights is the
        // same number in every row, so the blank column's error does not average out along an alignment - it was most of the loss error.
        const float4 z = *reinterpret_cast<const float4*>(emis + row * 4);
        zb = z.z;
        zy = z.w;
        S += __expf(zb - cur) - __expf(z.x - cur);
        if (u < Ub && clampi(labels[(long)b * (U1 - 1) + u], 0, V - 1) != blank) S += __expf(zy - cur) - __expf(z.y - cur);
    }
    const bool good = S > 1.0e-37f && S < 3.0e38f;                        // false also for NaN
    if (!good && flag) atomicOr(flag, 1);
    const float l = good ? __logf(S) : __int_as_float(0x7fc00000);
    lse[row] = l;
    if (shift_next && good) {
        // the shift the NEXT step should use: this row's log-sum-exp in logit units, less a margin that keeps exp() far from both
        // ends of the f32 / bf16 range.  Non-negative floats order like their bit patterns, so an integer atomic max does it.
        const float cand = l + cur - EXP_SHIFT_MARGIN;
        if (cand > 0.f && cand < 3.0e38f) atomicMax(reinterpret_cast<int*>(shift_next), __float_as_int(cand));
    }
    const bf16_t* r = P + row * ldv;
    const long di = (long)b * diag_stride(T, U1) + (long)(t + u) * U1 + u;
    if (emis) {
        lpb_d[di] = zb - cur - l;
        lpl_d[di] = u < Ub ? zy - cur - l : NEG;
        return;
    }
    const float pb = ldf<bf16_t>(r + blank);
    lpb_d[di] = (pb > 0.f ? __logf(pb) : NEG) - l;
    float pl = NEG;
    if (u < Ub) {
        const int y = clampi(labels[(long)b * (U1 - 1) + u], 0, V - 1);
        const float py = ldf<bf16_t>(r + y);
        pl = (py > 0.f ? __logf(py) : NEG) - l;
    }
    lpl_d[di] = pl;
}

// max over the lattice rows of (log-sum-exp - margin) into *shift_next, from the workspace a PLAIN ttmi_rnnt_loss_fwd left behind (its lse
// array, logit units): how the exp-domain form gets its first shift - and a new one after a flagged step - without a host synchronisation
__global__ __launch_bounds__(256) void rnnt_shift_seed_kernel(const float* __restrict__ lse, const int* __restrict__ act_lens,
                                                              const int* __restrict__ label_lens, int B, int T, int U1,
                                                              float* __restrict__ shift_next) {
    const long row = (long)blockIdx.x * 256 + threadIdx.x;
    float cand = 0.f;
    if (row < (long)B * T * U1) {
        const int u = (int)(row % U1);
        const long bt = row / U1;
        const int t = (int)(bt % T);
        const int b = (int)(bt / T);
        const int Tb = clampi(act_lens[b], 1, T), Ub = clampi(label_lens[b], 0, U1 - 1);
        if (t < Tb && u <= Ub) cand = lse[row] - EXP_SHIFT_MARGIN;
    }
    cand = wave_max(cand > 0.f && cand < 3.0e38f ? cand : 0.f);
    if ((threadIdx.x & 63) == 0 && cand > 0.f) atomicMax(reinterpret_cast<int*>(shift_next), __float_as_int(cand));
}

__global__ __launch_bounds__(256) void rnnt_scale_exp_kernel(
    bf16_t* __restrict__ P, long ldv, const int* __restrict__ labels, const int* __restrict__ act_lens,
    const int* __restrict__ label_lens, int B, int T, int U1, int V, int blank, const float* __restrict__ lse,
    const acc_t* __restrict__ alpha_d, const acc_t* __restrict__ beta_d, const acc_t* __restrict__ ll,
    const float* __restrict__ grad_out, int grad_out_stride, float scale, float* __restrict__ srow, bf16_t* __restrict__ srow16) {
    const long row = (long)blockIdx.x * 256 + threadIdx.x;
    if (row >= (long)B * T * U1) return;
    const int u = (int)(row % U1);
    const long bt = row / U1;
    const int t = (int)(bt % T);
    const int b = (int)(bt / T);
    const int Tb = clampi(act_lens[b], 1, T), Ub = clampi(label_lens[b], 0, U1 - 1);
    float sr = 0.f;
    if (t < Tb && u <= Ub) {
        const acc_t* al = alpha_d + (long)b * diag_stride(T, U1);
        const acc_t* be = beta_d + (long)b * diag_stride(T, U1);
        const long di = (long)(t + u) * U1 + u;
        const acc_t bcur = be[di];
        sr = scale * grad_out[(long)b * grad_out_stride] * __expf((float)(al[di] - ll[b * 2] + bcur) - lse[row]);
        // emission terms: e_blank / (s P_blank) = exp(beta[t+1,u] - beta[t,u]) (exp(-beta) at the terminal cell), e_label likewise
        float rb = 0.f, rl = 0.f;
        if (t == Tb - 1 && u == Ub) rb = __expf((float)(-bcur));
        else if (t < Tb - 1) rb = __expf((float)(be[di + U1] - bcur));
        bf16_t* r = P + row * ldv;
        int yv = -1;
        if (u < Ub) {
            yv = clampi(labels[(long)b * (U1 - 1) + u], 0, V - 1);
            rl = __expf((float)(be[di + U1 + 1] - bcur));
        }
        const float pb = ldf<bf16_t>(r + blank);
        if (yv == blank) stf<bf16_t>(r + blank, pb * (1.f - rb - rl));
        else {
            stf<bf16_t>(r + blank, pb * (1.f - rb));
            if (yv >= 0) stf<bf16_t>(r + yv, ldf<bf16_t>(r + yv) * (1.f - rl));
        }
    }
    srow[row] = sr;
    stf<bf16_t>(srow16 + row, sr);
}

template <int R, int PF>
void launch_alphabeta(hipStream_t st, int B, const float* lpb, const float* lpl, const int* al, const int* ll_, int T, int U1,
                      acc_t* a, acc_t* b, acc_t* ll, float* costs) {
    hipLaunchKernelGGL((rnnt_alphabeta_kernel<R, PF>), dim3(2 * B), dim3(64), 0, st, lpb, lpl, al, ll_, T, U1, a, b, ll, costs);
}


int g_lattice_version = 0;      // ttmi_set_option(9, 1): the round-1 kernel (one wave per utterance and direction) for A/B measurements
// alpha + beta of B utterances.  Default: rnnt_lattice_lds_kernel (one workgroup per utterance and direction: one wave per 64 labels + a helper
// wave); the one-wave kernel for U + 1 > 960 (17 waves do not fit a workgroup) and on request.
int launch_lattice(hipStream_t st, int B, const float* lpb, const float* lpl, const int* al, const int* ll_, int T, int U1, acc_t* a, acc_t* b,
                   acc_t* ll, float* costs) {
    const int W = (U1 + 63) / 64;
    if (g_lattice_version == 0 && W <= 15) {
        int CH = (120 * 1024) / (32 * U1);
        CH = CH > 32 ? 32 : (CH < 2 ? 2 : CH);
        const int CHR = (CH * U1 + 63) / 64 * 64;
        const size_t lds = lat_ring_offset(CHR, CH, U1) + (size_t)W * LAT_NB * sizeof(LatSlot) + 64;
        if (lat_ring_offset(CHR, CH, U1) % 16 != 0) {
            ttmi_set_error("rnnt lattice: boundary ring at LDS offset %zu is not 16-byte aligned (CH %d, U1 %d)", lat_ring_offset(CHR, CH, U1), CH, U1);
            return TTMI_EINVAL;
        }
        static size_t enabled = 0;
        if (lds > enabled) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rnnt_lattice_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) {
                ttmi_set_error("rnnt lattice: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
                return (int)e;
            }
            enabled = lds;
        }
        hipLaunchKernelGGL(rnnt_lattice_lds_kernel, dim3(2 * B), dim3((W + 1) * 64), lds, st, lpb, lpl, al, ll_, T, U1, W, CH, a, b, ll, costs);
        return TTMI_OK;
    }
    if (U1 <= 64) launch_alphabeta<1, 8>(st, B, lpb, lpl, al, ll_, T, U1, a, b, ll, costs);
    else if (U1 <= 128) launch_alphabeta<2, 8>(st, B, lpb, lpl, al, ll_, T, U1, a, b, ll, costs);
    else if (U1 <= 256) launch_alphabeta<4, 4>(st, B, lpb, lpl, al, ll_, T, U1, a, b, ll, costs);
    else if (U1 <= 512) launch_alphabeta<8, 2>(st, B, lpb, lpl, al, ll_, T, U1, a, b, ll, costs);
    else launch_alphabeta<16, 1>(st, B, lpb, lpl, al, ll_, T, U1, a, b, ll, costs);
    return TTMI_OK;
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

void ttmi_rnnt_set_lattice_version(int v) { g_lattice_version = v; }
void ttmi_probe_begin(int point, hipStream_t st);      // optim.hip: HIP-event timing probes (point 1 = loss forward, 2 = loss backward)
void ttmi_probe_end(int point, hipStream_t st);

extern "C" {

// bytes of caller-provided workspace shared by ttmi_rnnt_loss_fwd / _bwd
size_t ttmi_rnnt_workspace_bytes(int B, int T, int U1) {
    return sizeof(float) * ((size_t)B * T * U1 + 2 * (size_t)B * diag_stride(T, U1)) +
           sizeof(acc_t) * (2 * (size_t)B * diag_stride(T, U1) + 2 * (size_t)B) + 64;
}

// Forward: per-utterance costs[B] = -log P(y|x).  logits [B,T,U1,V] (dtype 0 = f32, 1 = bf16) with row pitch ldv >= V
// elements (rows (b,t,u) at logits + ((b*T+t)*U1+u)*ldv), labels i32 [B,U1-1], act_lens/label_lens i32 [B] (device
// pointers).  Fills the workspace (lse, alpha, beta, ll) that ttmi_rnnt_loss_bwd consumes.
int ttmi_rnnt_loss_fwd(const void* logits, int dtype, long ldv, const int* labels, const int* act_lens, const int* label_lens,
                       int B, int T, int U1, int V, int blank, void* workspace, float* costs, void* stream) {
    TTMI_REQUIRE(logits && (labels || U1 == 1) && act_lens && label_lens && workspace && costs, "rnnt_loss_fwd: null pointer");
    TTMI_REQUIRE(B > 0 && T > 0 && U1 > 0 && V > 0, "rnnt_loss_fwd: bad shape B=%d T=%d U1=%d V=%d", B, T, U1, V);
    TTMI_REQUIRE(ldv >= V && (dtype == 0 || dtype == 1), "rnnt_loss_fwd: bad pitch/dtype");
    TTMI_REQUIRE(blank >= 0 && blank < V, "rnnt_loss_fwd: blank %d outside [0,%d)", blank, V);
    TTMI_REQUIRE(U1 <= 1024, "rnnt_loss_fwd: U+1=%d > 1024 unsupported", U1);
    TTMI_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 7) == 0, "rnnt_loss_fwd: workspace must be 8-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    Ws w = carve(workspace, B, T, U1);
    const long rows = (long)B * T * U1;
    const size_t es = dtype == 0 ? 4 : 2;
    const int vec_ok = ((reinterpret_cast<uintptr_t>(logits) % es) == 0) ? 1 : 0;
    ttmi_probe_begin(1, st);
    if (dtype == 0)
        hipLaunchKernelGGL(rnnt_lse_kernel<float>, dim3(cdiv(rows, LSE_WAVES)), dim3(LSE_WAVES * 64), 0, st,
                           static_cast<const float*>(logits), ldv, labels, act_lens, label_lens, B, T, U1, V, blank, vec_ok,
                           w.lse, w.lpb, w.lpl);
    else
        hipLaunchKernelGGL(rnnt_lse_kernel<bf16_t>, dim3(cdiv(rows, LSE_WAVES)), dim3(LSE_WAVES * 64), 0, st,
                           static_cast<const bf16_t*>(logits), ldv, labels, act_lens, label_lens, B, T, U1, V, blank, vec_ok,
                           w.lse, w.lpb, w.lpl);
    TTMI_LAUNCH_CHECK("rnnt_lse_kernel");
    const int lat_rc = launch_lattice(st, B, w.lpb, w.lpl, act_lens, label_lens, T, U1, w.alpha, w.beta, w.ll, costs);
    ttmi_probe_end(1, st);
    if (lat_rc) return lat_rc;
    TTMI_LAUNCH_CHECK("rnnt lattice kernel");
    return TTMI_OK;
}

// Backward: grad[b,t,u,v] = scale * grad_out[b*grad_out_stride] * d costs[b] / d logits, same dtype as the logits, row
// pitch ldg (columns [V, ldg) are written as zeros).  grad may alias logits (in-place) when ldg == ldv.  Cells outside
// [0,T_b) x [0,U_b] get exact zeros.
int ttmi_rnnt_loss_bwd(const void* logits, int dtype, long ldv, const int* labels, const int* act_lens, const int* label_lens,
                       int B, int T, int U1, int V, int blank, const void* workspace, const float* grad_out,
                       int grad_out_stride, float scale, void* grad, long ldg, void* stream) {
    TTMI_REQUIRE(logits && (labels || U1 == 1) && act_lens && label_lens && workspace && grad_out && grad,
                 "rnnt_loss_bwd: null pointer");
    TTMI_REQUIRE(B > 0 && T > 0 && U1 > 0 && V > 0, "rnnt_loss_bwd: bad shape");
    TTMI_REQUIRE(ldv >= V && ldg >= V && (dtype == 0 || dtype == 1), "rnnt_loss_bwd: bad pitch/dtype");
    TTMI_REQUIRE(blank >= 0 && blank < V, "rnnt_loss_bwd: blank %d outside [0,%d)", blank, V);
    hipStream_t st = static_cast<hipStream_t>(stream);
    Ws w = carve(const_cast<void*>(workspace), B, T, U1);
    const long rows = (long)B * T * U1;
    const size_t es = dtype == 0 ? 4 : 2;
    // vector path needs logits and grad rows to share their 16-byte phase
    const int vec_ok = ((reinterpret_cast<uintptr_t>(logits) % 16) == (reinterpret_cast<uintptr_t>(grad) % 16) &&
                        ((ldv - ldg) * (long)es) % 16 == 0 && (reinterpret_cast<uintptr_t>(logits) % es) == 0) ? 1 : 0;
    ttmi_probe_begin(2, st);
    if (dtype == 0)
        hipLaunchKernelGGL(rnnt_grad_kernel<float>, dim3(cdiv(rows, LSE_WAVES)), dim3(LSE_WAVES * 64), 0, st,
                           static_cast<const float*>(logits), ldv, labels, act_lens, label_lens, B, T, U1, V, blank, vec_ok,
                           w.lse, w.alpha, w.beta, w.ll, grad_out, grad_out_stride, scale, static_cast<float*>(grad), ldg);
    else
        hipLaunchKernelGGL(rnnt_grad_kernel<bf16_t>, dim3(cdiv(rows, LSE_WAVES)), dim3(LSE_WAVES * 64), 0, st,
                           static_cast<const bf16_t*>(logits), ldv, labels, act_lens, label_lens, B, T, U1, V, blank, vec_ok,
                           w.lse, w.alpha, w.beta, w.ll, grad_out, grad_out_stride, scale, static_cast<bf16_t*>(grad), ldg);
    ttmi_probe_end(2, st);
    TTMI_LAUNCH_CHECK("rnnt_grad_kernel");
    return TTMI_OK;
}

// bf16x3 mode: the gradient of f32 logits [rows, ldv] written over them as two bf16 planes per row, [hi | lo] with pitch 2 ldv bf16 - the operand layout of the joint's
// three-term backward (ttmi_joint_bwd_split), which then skips its own split pass over d logits.  Same workspace and scale semantics as ttmi_rnnt_loss_bwd.
int ttmi_rnnt_loss_bwd_split_ok(long ldv, const void* logits) {
    return ldv % 64 == 0 && ldv <= 256L * 32 && (reinterpret_cast<uintptr_t>(logits) & 15) == 0;
}
int ttmi_rnnt_loss_bwd_split(void* logits, long ldv, const int* labels, const int* act_lens, const int* label_lens, int B, int T, int U1, int V,
                             int blank, const void* workspace, const float* grad_out, int grad_out_stride, float scale, void* stream) {
    TTMI_REQUIRE(logits && (labels || U1 == 1) && act_lens && label_lens && workspace && grad_out, "rnnt_loss_bwd_split: null pointer");
    TTMI_REQUIRE(B > 0 && T > 0 && U1 > 0 && V > 0 && ldv >= V && blank >= 0 && blank < V, "rnnt_loss_bwd_split: bad shape");
    TTMI_REQUIRE(ttmi_rnnt_loss_bwd_split_ok(ldv, logits), "rnnt_loss_bwd_split: rows must be 16-byte aligned with a pitch %% 64 == 0 of at most 8192 floats (pitch %ld)", ldv);
    hipStream_t st = static_cast<hipStream_t>(stream);
    Ws w = carve(const_cast<void*>(workspace), B, T, U1);
    const long rows = (long)B * T * U1;
    const int kv = (int)cdiv(ldv / 4, 64L);
    ttmi_probe_begin(2, st);
#define SPLIT_LAUNCH(KV) hipLaunchKernelGGL(rnnt_grad_split_kernel<KV>, dim3(cdiv(rows, LSE_WAVES)), dim3(LSE_WAVES * 64), 0, st, static_cast<float*>(logits), ldv, labels, \
                           act_lens, label_lens, B, T, U1, V, blank, w.lse, w.alpha, w.beta, w.ll, grad_out, grad_out_stride, scale)
    if (kv <= 8) SPLIT_LAUNCH(8);
    else if (kv <= 17) SPLIT_LAUNCH(17);
    else if (kv <= 26) SPLIT_LAUNCH(26);
    else SPLIT_LAUNCH(32);
#undef SPLIT_LAUNCH
    ttmi_probe_end(2, st);
    TTMI_LAUNCH_CHECK("rnnt_grad_split_kernel");
    return TTMI_OK;
}

// Exp-domain forms for the fused joint + loss fast path (see rnnt_prep_exp_kernel).  P bf16 [rows, ldv] = exp(logit - shift) and
// rowsum f32 [nparts, rows] come from ttmi_joint_fwd_exp; same workspace as above.  shift_cur (device scalar, nullable = 0) is the
// shift P was formed with; shift_next (device scalar, nullable) receives max(itself, max over rows of log-sum-exp - 40): callers
// feed it to the next step's forward so that exp() stays inside the bf16 range whatever the scale of the logits.  The backward patches the blank / label entries
// of P in place and writes the per-row factors srow (f32) and srow16 (bf16): d logits = srow[r] * P[r, :].
int ttmi_rnnt_loss_fwd_exp(const void* P, long ldv, const float* rowsum, int nparts, const int* labels, const int* act_lens,
                           const int* label_lens, int B, int T, int U1, int V, int blank, void* workspace, float* costs,
                           const float* shift_cur, float* shift_next, const float* emis, int* flag, void* stream) {
    TTMI_REQUIRE(P && rowsum && (labels || U1 == 1) && act_lens && label_lens && workspace && costs, "rnnt_loss_fwd_exp: null pointer");
    TTMI_REQUIRE(B > 0 && T > 0 && U1 > 0 && V > 0 && nparts > 0 && ldv >= V, "rnnt_loss_fwd_exp: bad shape");
    TTMI_REQUIRE(blank >= 0 && blank < V, "rnnt_loss_fwd_exp: blank %d outside [0,%d)", blank, V);
    TTMI_REQUIRE(U1 <= 1024, "rnnt_loss_fwd_exp: U+1=%d > 1024 unsupported", U1);
    TTMI_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 7) == 0, "rnnt_loss_fwd_exp: workspace must be 8-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    Ws w = carve(workspace, B, T, U1);
    const long rows = (long)B * T * U1;
    ttmi_probe_begin(1, st);
    hipLaunchKernelGGL(rnnt_prep_exp_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, st, static_cast<const bf16_t*>(P), ldv, rowsum,
                       nparts, labels, act_lens, label_lens, B, T, U1, V, blank, w.lse, w.lpb, w.lpl, shift_cur, shift_next, emis, flag);
    TTMI_LAUNCH_CHECK("rnnt_prep_exp_kernel");
    const int lat_rc = launch_lattice(st, B, w.lpb, w.lpl, act_lens, label_lens, T, U1, w.alpha, w.beta, w.ll, costs);
    ttmi_probe_end(1, st);
    if (lat_rc) return lat_rc;
    TTMI_LAUNCH_CHECK("rnnt lattice kernel");
    return TTMI_OK;
}

// *shift_next = max(*shift_next, max over the lattice rows of log-sum-exp - 40) from the workspace of a plain ttmi_rnnt_loss_fwd on the same
// (B, T, U1): seeds the exp-domain form's shift on the device (first step, new weights, or after a flagged step)
int ttmi_rnnt_shift_seed(const void* workspace, const int* act_lens, const int* label_lens, int B, int T, int U1, float* shift_next,
                         void* stream) {
    TTMI_REQUIRE(workspace && act_lens && label_lens && shift_next && B > 0 && T > 0 && U1 > 0, "rnnt_shift_seed: bad arguments");
    Ws w = carve(const_cast<void*>(workspace), B, T, U1);
    const long rows = (long)B * T * U1;
    hipLaunchKernelGGL(rnnt_shift_seed_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), w.lse, act_lens,
                       label_lens, B, T, U1, shift_next);
    TTMI_LAUNCH_CHECK("rnnt_shift_seed_kernel");
    return TTMI_OK;
}

int ttmi_rnnt_loss_bwd_exp(void* P, long ldv, const int* labels, const int* act_lens, const int* label_lens, int B, int T, int U1,
                           int V, int blank, const void* workspace, const float* grad_out, int grad_out_stride, float scale,
                           float* srow, void* srow16, void* stream) {
    TTMI_REQUIRE(P && (labels || U1 == 1) && act_lens && label_lens && workspace && grad_out && srow && srow16,
                 "rnnt_loss_bwd_exp: null pointer");
    TTMI_REQUIRE(B > 0 && T > 0 && U1 > 0 && V > 0 && ldv >= V, "rnnt_loss_bwd_exp: bad shape");
    TTMI_REQUIRE(blank >= 0 && blank < V, "rnnt_loss_bwd_exp: blank %d outside [0,%d)", blank, V);
    hipStream_t st = static_cast<hipStream_t>(stream);
    Ws w = carve(const_cast<void*>(workspace), B, T, U1);
    const long rows = (long)B * T * U1;
    ttmi_probe_begin(2, st);
    hipLaunchKernelGGL(rnnt_scale_exp_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, st, static_cast<bf16_t*>(P), ldv, labels,
                       act_lens, label_lens, B, T, U1, V, blank, w.lse, w.alpha, w.beta, w.ll, grad_out, grad_out_stride, scale,
                       srow, static_cast<bf16_t*>(srow16));
    ttmi_probe_end(2, st);
    TTMI_LAUNCH_CHECK("rnnt_scale_exp_kernel");
    return TTMI_OK;
}

}  // extern "C"
