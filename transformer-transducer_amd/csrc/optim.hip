// Training-step tail on flat f32 buffers: global grad-norm, clip, SGD(momentum)/Adam update.
// Mirrors train.py:62-65 (clip_grad_norm_(max_grad_norm) then optimizer.step()) and tt/optim.py:57-73
// (SGD momentum / Adam betas (0.9, 0.98), eps 1e-8).  HBM-bound streaming kernels, 16-byte accesses.
// The clip coefficient is computed ON DEVICE from the reduced norm: no host sync in the step.
#include "common.h"

namespace {

// Two-stage, fixed-order reduction: every rank must obtain the bit-identical norm from identical (all-reduced)
// gradients, otherwise the clip coefficient - and with it the replicas - drift apart.  No atomics here.
constexpr int SUMSQ_BLOCKS = 1024;
__device__ float g_sumsq_partials[SUMSQ_BLOCKS];

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long n) {
    float acc = 0.f;
    const long n4 = n >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = x4[i];
        acc += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) acc += x[i] * x[i];
    acc = wave_sum(acc);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) g_sumsq_partials[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

__global__ __launch_bounds__(64) void sumsq_final_kernel(int nblocks, float* __restrict__ out) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 64) acc += g_sumsq_partials[i];
    acc = wave_sum(acc);
    if (threadIdx.x == 0) *out += acc;
}

// A step whose gradient norm is not finite is DROPPED (parameters and optimiser state untouched): with clip_grad_norm_ the reference
// would turn every weight into NaN at this point (train.py:62-65); here it is the exp-domain loss form's way of failing loudly - a step
// whose shift no longer fitted the logits has NaN costs and gradients (rnnt_prep_exp_kernel) and must not reach the weights.
// The drop does not depend on clipping being on (max_norm): whoever passes `normsq` gets it (FusedOptimizer always does; round 3 skipped the
// norm when max_norm == 0 - tt.optim.Optimizer, where train.py clips by itself - and a flagged exp-domain step then reached the weights).
__device__ __forceinline__ bool step_dropped(const float* normsq) {
    return normsq && !(*normsq < 3.0e38f);          // inf or NaN
}
// Hyper-parameters that change between steps live in a small DEVICE array the kernels read at run time (`hyper`, may be NULL = use the
// by-value arguments): hyper[0] = learning rate, hyper[1] = number of optimiser steps taken so far (Adam's bias-correction exponent,
// advanced on the device by optim_tick_kernel).  A step captured into a HIP graph (ttmi.train.GraphedStep) therefore follows
// Optimizer.decay_lr() (tt/optim.py:30-33, train.py:257) and Adam's step count under replay - by-value arguments are frozen at capture.
// A DROPPED step (non-finite norm) does not consume a step count - Adam's bias corrections would otherwise run ahead of m and v, which
// the drop leaves untouched - and is counted in hyper[2], which the host may poll without synchronising the step.
__global__ void optim_tick_kernel(float* __restrict__ hyper, const float* __restrict__ normsq) {
    if (step_dropped(normsq)) hyper[2] += 1.f;
    else hyper[1] += 1.f;
}
// coef = grad_scale * min(1, max_norm / (grad_scale * sqrt(normsq) + 1e-6))   (torch.nn.utils.clip_grad_norm_)
__device__ __forceinline__ float clip_coef(const float* normsq, float max_norm, float grad_scale) {
    if (!normsq || max_norm <= 0.f) return grad_scale;
    const float c = max_norm / (grad_scale * sqrtf(*normsq) + 1e-6f);
    return grad_scale * fminf(c, 1.f);
}

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ mom,
                                                  long n, float lr, float momentum, float wd, int nesterov, float max_norm,
                                                  const float* __restrict__ normsq, float grad_scale, const float* __restrict__ hyper) {
    if (step_dropped(normsq)) return;
    if (hyper) lr = hyper[0];
    const float coef = clip_coef(normsq, max_norm, grad_scale);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float gi = g[i] * coef + wd * p[i];
        if (momentum != 0.f) {
            const float b = momentum * mom[i] + gi;
            mom[i] = b;
            gi = nesterov ? gi + momentum * b : b;
        }
        p[i] -= lr * gi;
    }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                   float wd, float bc1, float bc2, float max_norm,
                                                   const float* __restrict__ normsq, float grad_scale, const float* __restrict__ hyper) {
    if (step_dropped(normsq)) return;
    if (hyper) {                                    // (hyper[1] was advanced by optim_tick_kernel just before this launch: counts from 1)
        lr = hyper[0];
        bc1 = 1.f - powf(b1, hyper[1]);
        bc2 = 1.f - powf(b2, hyper[1]);
    }
    const float coef = clip_coef(normsq, max_norm, grad_scale);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gi = g[i] * coef + wd * p[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= (lr / bc1) * mi / (sqrtf(vi) / sqrtf(bc2) + eps);
    }
}

// torch.optim.Adadelta (tt/optim.py:74-81): sq = rho sq + (1 - rho) g^2; delta = sqrt(acc + eps) / sqrt(sq + eps) g; acc = rho acc + (1 - rho) delta^2;
// p -= lr delta
__global__ __launch_bounds__(256) void adadelta_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ sq,
                                                       float* __restrict__ acc, long n, float lr, float rho, float eps, float wd,
                                                       float max_norm, const float* __restrict__ normsq, float grad_scale,
                                                       const float* __restrict__ hyper) {
    if (step_dropped(normsq)) return;
    if (hyper) lr = hyper[0];
    const float coef = clip_coef(normsq, max_norm, grad_scale);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gi = g[i] * coef + wd * p[i];
        const float s = rho * sq[i] + (1.f - rho) * gi * gi;
        const float a = acc[i];
        const float delta = sqrtf(a + eps) / sqrtf(s + eps) * gi;
        sq[i] = s;
        acc[i] = rho * a + (1.f - rho) * delta * delta;
        p[i] -= lr * delta;
    }
}

int grid_for(long n) {
    long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

extern "C" {

// *out += sum(x^2), deterministic (fixed reduction order; calls on one stream only - the partials buffer is shared)
int ttmi_sumsq(const float* x, long n, float* out, void* stream) {
    TTMI_REQUIRE(x && out && n > 0, "sumsq: bad arguments");
    TTMI_REQUIRE(aligned16(x), "sumsq: x must be 16-byte aligned");
    int nb = grid_for(n >> 2);
    if (nb > SUMSQ_BLOCKS) nb = SUMSQ_BLOCKS;
    hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), x, n);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), nb, out);
    TTMI_LAUNCH_CHECK("sumsq_kernel");
    return TTMI_OK;
}

// torch.optim.SGD step on a flat buffer with the gradient clip folded in (normsq may be NULL = no clipping).
// The effective gradient is g * grad_scale (e.g. 1/world_size after a SUM all-reduce), clipped to max_norm.
// hyper (device float[3], may be NULL): see optim_tick_kernel above - hyper[0] replaces `lr` at run time, hyper[1] counts the steps TAKEN,
// hyper[2] the steps dropped.  A non-finite *normsq drops the step.
int ttmi_sgd_step(float* p, const float* g, float* mom, long n, float lr, float momentum, float weight_decay, int nesterov,
                  float max_norm, const float* normsq, float grad_scale, float* hyper, void* stream) {
    TTMI_REQUIRE(p && g && n > 0 && (mom || momentum == 0.f), "sgd_step: bad arguments");
    if (hyper) hipLaunchKernelGGL(optim_tick_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), hyper, normsq);
    hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), p, g, mom, n, lr,
                       momentum, weight_decay, nesterov, max_norm, normsq, grad_scale, hyper);
    TTMI_LAUNCH_CHECK("sgd_kernel");
    return TTMI_OK;
}

// torch.optim.Adam step (step counts from 1)
int ttmi_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int step, float max_norm, const float* normsq, float grad_scale, float* hyper, void* stream) {
    TTMI_REQUIRE(p && g && m && v && n > 0 && (step > 0 || hyper), "adam_step: bad arguments");
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);      // (replaced on the device when hyper is given)
    if (hyper) hipLaunchKernelGGL(optim_tick_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), hyper, normsq);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), p, g, m, v, n, lr, beta1,
                       beta2, eps, weight_decay, bc1, bc2, max_norm, normsq, grad_scale, hyper);
    TTMI_LAUNCH_CHECK("adam_kernel");
    return TTMI_OK;
}

// torch.optim.Adadelta step (square_avg, acc_delta state buffers)
int ttmi_adadelta_step(float* p, const float* g, float* square_avg, float* acc_delta, long n, float lr, float rho, float eps,
                       float weight_decay, float max_norm, const float* normsq, float grad_scale, float* hyper, void* stream) {
    TTMI_REQUIRE(p && g && square_avg && acc_delta && n > 0 && rho >= 0.f && rho <= 1.f && eps > 0.f, "adadelta_step: bad arguments");
    if (hyper) hipLaunchKernelGGL(optim_tick_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), hyper, normsq);
    hipLaunchKernelGGL(adadelta_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), p, g, square_avg, acc_delta, n, lr,
                       rho, eps, weight_decay, max_norm, normsq, grad_scale, hyper);
    TTMI_LAUNCH_CHECK("adadelta_kernel");
    return TTMI_OK;
}

// ---- timing probes: HIP events recorded on the launch stream around named launches --------------------------
// Probe points: 0 = joint vocabulary projection GEMM (forward), 1 = RNN-T loss forward (log-sum-exp pass + lattice), 2 = RNN-T loss
// backward (gradient pass), 3 = fused attention backward kernel of an audio-sized layer (L >= 256), 4 = a grouped weight-gradient launch
// (ttmi_wgrad_group, >= 128 tiles).  Events are owned by the library.  Every point records into whichever of the 64 event pairs was armed
// last, so a timing loop can arm pair i in step i and read them all after its final fence - no host synchronisation inside the
// timed region.
constexpr int NPROBE = 64, NPOINT = 5;
static hipEvent_t g_probe[NPOINT][NPROBE][2];
static int g_probe_state[NPOINT][NPROBE];     // 0 idle, 1 armed, 2 recorded
static int g_probe_cur = -1;

int ttmi_probe_arm(int slot) {
    TTMI_REQUIRE(slot >= 0 && slot < NPROBE, "probe: bad slot");
    for (int pt = 0; pt < NPOINT; ++pt) {
        if (!g_probe[pt][slot][0]) {
            (void)hipEventCreate(&g_probe[pt][slot][0]);
            (void)hipEventCreate(&g_probe[pt][slot][1]);
        }
        g_probe_state[pt][slot] = 1;
    }
    g_probe_cur = slot;
    return TTMI_OK;
}
// blocks until the stop event has completed; returns elapsed milliseconds (or <0 if the probe never fired)
float ttmi_probe_point_read_ms(int point, int slot) {
    if (point < 0 || point >= NPOINT || slot < 0 || slot >= NPROBE || !g_probe[point][slot][0] || g_probe_state[point][slot] != 2) return -1.f;
    float ms = -1.f;
    (void)hipEventSynchronize(g_probe[point][slot][1]);
    (void)hipEventElapsedTime(&ms, g_probe[point][slot][0], g_probe[point][slot][1]);
    g_probe_state[point][slot] = 0;
    return ms;
}
float ttmi_probe_read_ms(int slot) { return ttmi_probe_point_read_ms(0, slot); }
}

void ttmi_probe_begin(int pt, hipStream_t st) {
    if (g_probe_cur >= 0 && g_probe_state[pt][g_probe_cur] == 1) (void)hipEventRecord(g_probe[pt][g_probe_cur][0], st);
}
void ttmi_probe_end(int pt, hipStream_t st) {
    if (g_probe_cur >= 0 && g_probe_state[pt][g_probe_cur] == 1) {
        (void)hipEventRecord(g_probe[pt][g_probe_cur][1], st);
        g_probe_state[pt][g_probe_cur] = 2;
    }
}
