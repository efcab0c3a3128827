// Feature front-end on the GPU: waveform -> log-mel -> frame stacking -> subsampling (+ zero padding) and the in-loop time /
// frequency masks.  Replaces the per-utterance numpy code of the reference's data loader and training loop:
//   get_feature / get_feature2      tt/utils.py:182-207  (librosa.feature.melspectrogram(y, sr, n_fft=512, hop_length=160, n_mels) + log)
//   concat_frame                    tt/utils.py:120-143
//   subsampling                     tt/utils.py:146-151
//   Dataset.pad                     tt/dataset.py:40-57
//   frequency_/time_mask_augment    tt/utils.py:297-329  (train.py:41-44)
//
// The STFT is a dense contraction with a fixed [n_fft x 2(n_fft/2+1)] basis (the Hann window folded into its rows), so it runs on the
// exact-f32 MFMA GEMM (v_mfma_f32_32x32x2_f32) instead of an FFT: one gather kernel lays the reflect-padded frames out as a matrix,
// two GEMMs (DFT, mel filterbank) and two elementwise kernels (power, log) do the rest, batched over the utterances of a step.
// Everything else here is HBM-bound row copying: one thread per 4 output floats, coalesced 16-byte stores.
#include "gemm.h"

namespace {

// frames[b, f, k] = y_b[reflect(f * hop + k - n_fft / 2)] for f < 1 + n_b / hop, else 0   (librosa.stft: center=True, pad_mode='reflect')
__global__ __launch_bounds__(256) void frames_kernel(const short* __restrict__ wave, long pitch, const int* __restrict__ n_samples,
                                                     int Fmax, int n_fft, int hop, float* __restrict__ frames) {
    const int b = blockIdx.y, f = blockIdx.x;
    const int n = n_samples[b];
    const int nf = n > n_fft / 2 ? 1 + n / hop : 0;        // np.pad(reflect) needs more samples than the pad width
    float* out = frames + ((long)b * Fmax + f) * n_fft;
    const short* y = wave + (long)b * pitch;
    for (int k = threadIdx.x; k < n_fft; k += 256) {
        float v = 0.f;
        if (f < nf) {
            int i = f * hop + k - n_fft / 2;
            if (i < 0) i = -i;
            if (i >= n) i = 2 * (n - 1) - i;
            v = (i >= 0 && i < n) ? (float)y[i] : 0.f;
        }
        out[k] = v;
    }
}

// power[r, k] = re^2 + im^2 from spec[r, 2k], spec[r, 2k + 1]
__global__ __launch_bounds__(256) void power_kernel(const float* __restrict__ spec, long rows, int nbin, float* __restrict__ power) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * nbin) return;
    const long r = idx / nbin;
    const int k = (int)(idx % nbin);
    const float2 z = *reinterpret_cast<const float2*>(spec + r * (2L * nbin) + 2 * k);
    power[idx] = z.x * z.x + z.y * z.y;
}

// in place: mode 0 = natural log where > 0 and 0 elsewhere (np.ma.log(...).filled(0), tt/utils.py:191-192); mode 1 = log10 with exact zeros
// replaced by the float64 machine epsilon first (tt/utils.py:205-206).  Rows beyond an utterance's frame count are zero padding.
__global__ __launch_bounds__(256) void log_kernel(float* __restrict__ mel, const int* __restrict__ n_samples, int Fmax, int n_mels, int hop,
                                                  int n_fft, int mode, long total) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const long row = idx / n_mels;
    const int b = (int)(row / Fmax), f = (int)(row % Fmax);
    const int n = n_samples[b];
    const int nf = n > n_fft / 2 ? 1 + n / hop : 0;
    const float s = mel[idx];
    float v = 0.f;
    if (f < nf) {
        if (mode == 0) v = s > 0.f ? logf(s) : 0.f;
        else v = log10f(s == 0.f ? 2.220446049250313e-16f : s);
    }
    mel[idx] = v;
}

// out[b, t', blk * F + c] = feat[b, s * t' + shift(blk), c], zero off the ends of the utterance and for t' >= ceil(n_b / s).
// Blocks 0..left hold the frames t - left .. t.  Future frame t + i + 1 goes to block RIGHT + i + 1 exactly as tt/utils.py:138-141 writes it
// (not left + i + 1; later writes win, which is the order the loops below reproduce).
__global__ __launch_bounds__(256) void stack_subsample_kernel(const float* __restrict__ feat, const int* __restrict__ n_frames, int Tin,
                                                              int F, int left, int right, int sub, int Tout, float* __restrict__ out,
                                                              int* __restrict__ out_lens) {
    const int b = blockIdx.y, tp = blockIdx.x;
    const int n = n_frames ? min(n_frames[b], Tin) : Tin;
    const int nout = (n + sub - 1) / sub;
    if (tp == 0 && threadIdx.x == 0 && out_lens) out_lens[b] = min(nout, Tout);
    const int W = F * (1 + left + right);
    float* o = out + ((long)b * Tout + tp) * W;
    const float* x = feat + (long)b * Tin * F;
    const int t = tp * sub;
    for (int c = threadIdx.x; c < W; c += 256) {
        float v = 0.f;
        if (tp < nout) {
            const int blk = c / F, col = c - blk * F;
            if (blk <= left) {
                const int src = t - (left - blk);
                if (src >= 0 && src < n) v = x[(long)src * F + col];
            }
            const int i = blk - right - 1;                   // a future frame written onto this block?  (overwrites, like the reference's loop)
            if (i >= 0 && i < right && t + i + 1 < n) v = x[(long)(t + i + 1) * F + col];
        }
        o[c] = v;
    }
}

constexpr int MAX_SPANS = 32;
struct Spans {
    int n_time, n_freq;
    int t0[MAX_SPANS], tw[MAX_SPANS], f0[MAX_SPANS], fw[MAX_SPANS];
};

// x[:, t0:t0+tw, :] = 0 and x[:, :, f0:f0+fw] = 0 for every span, whole batch (tt/utils.py:311,327), one pass, stores only
__global__ __launch_bounds__(256) void spec_mask_kernel(float* __restrict__ x, int T, int F, Spans sp) {
    const int t = blockIdx.x;
    float* row = x + ((long)blockIdx.y * T + t) * F;
    bool whole = false;
    for (int i = 0; i < sp.n_time; ++i) whole |= (t >= sp.t0[i] && t < sp.t0[i] + sp.tw[i]);
    if (whole) {
        for (int c = threadIdx.x; c < F; c += 256) row[c] = 0.f;
        return;
    }
    for (int i = 0; i < sp.n_freq; ++i)
        for (int c = sp.f0[i] + threadIdx.x; c < sp.f0[i] + sp.fw[i] && c < F; c += 256) row[c] = 0.f;
}

inline size_t al64(size_t n) { return (n + 63) & ~size_t(63); }

}  // namespace

extern "C" {

// scratch for ttmi_logmel: frames [B*Fmax, n_fft] + spectrum [B*Fmax, 2*(n_fft/2+1)] + power [B*Fmax, n_fft/2+1], Fmax = 1 + nmax / hop
size_t ttmi_logmel_ws_floats(int B, int nmax, int n_fft, int hop) {
    if (B <= 0 || nmax <= 0 || n_fft <= 0 || hop <= 0) return 0;
    const size_t rows = (size_t)B * (1 + nmax / hop), nbin = n_fft / 2 + 1;
    return al64(rows * n_fft) + al64(rows * 2 * nbin) + al64(rows * nbin) + 64;
}

// wave i16 [B, pitch >= nmax] (zero padded), n_samples i32 [B] (device) -> out f32 [B, Fmax, n_mels], Fmax = 1 + nmax / hop; frames beyond
// 1 + n_b / hop are zero.  dft [2*(n_fft/2+1), n_fft]: rows 2k / 2k+1 = w[n] cos(2 pi k n / n_fft) / -w[n] sin(...) with the analysis
// window w folded in; mel_w [n_mels, n_fft/2+1].  log_mode 0 = get_feature, 1 = get_feature2, 2 = no log (mel power).
int ttmi_logmel(const short* wave, long pitch, const int* n_samples, int B, int nmax, int n_fft, int hop, int n_mels, const float* dft,
                const float* mel_w, int log_mode, float* ws, float* out, void* stream) {
    TTMI_REQUIRE(wave && n_samples && dft && mel_w && ws && out, "logmel: null pointer");
    TTMI_REQUIRE(B > 0 && B <= 65535 && nmax > 0 && pitch >= nmax && n_fft >= 16 && n_fft % 4 == 0 && hop > 0 && n_mels > 0, "logmel: bad dims");
    TTMI_REQUIRE(log_mode >= 0 && log_mode <= 2, "logmel: bad log mode %d", log_mode);
    TTMI_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255) == 0, "logmel: workspace must be 256-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int Fmax = 1 + nmax / hop, nbin = n_fft / 2 + 1;
    const long rows = (long)B * Fmax;
    TTMI_REQUIRE(rows < (1L << 31) / 4, "logmel: too many frames");
    float* frames = ws;
    float* spec = frames + al64((size_t)rows * n_fft);
    float* power = spec + al64((size_t)rows * 2 * nbin);
    hipLaunchKernelGGL(frames_kernel, dim3(Fmax, B), dim3(256), 0, st, wave, pitch, n_samples, Fmax, n_fft, hop, frames);
    TTMI_LAUNCH_CHECK("frames_kernel");
    GemmDesc g;                                            // spectrum = frames . dft^T   (exact f32)
    g.A = frames; g.B = dft; g.C = spec; g.M = (int)rows; g.N = 2 * nbin; g.K = n_fft; g.lda = n_fft; g.ldb = n_fft; g.ldc = 2 * nbin;
    int rc = ttmi_launch_gemm(g, st);
    if (rc) return rc;
    hipLaunchKernelGGL(power_kernel, dim3(cdiv(rows * nbin, 256)), dim3(256), 0, st, spec, rows, nbin, power);
    TTMI_LAUNCH_CHECK("power_kernel");
    GemmDesc m;                                            // mel = power . mel_w^T
    m.A = power; m.B = mel_w; m.C = out; m.M = (int)rows; m.N = n_mels; m.K = nbin; m.lda = nbin; m.ldb = nbin; m.ldc = n_mels;
    rc = ttmi_launch_gemm(m, st);
    if (rc) return rc;
    if (log_mode != 2) {
        hipLaunchKernelGGL(log_kernel, dim3(cdiv(rows * n_mels, 256)), dim3(256), 0, st, out, n_samples, Fmax, n_mels, hop, n_fft, log_mode,
                           rows * n_mels);
        TTMI_LAUNCH_CHECK("log_kernel");
    }
    return TTMI_OK;
}

// feat f32 [B, Tin, F]; n_frames i32 [B] (device, nullable = Tin everywhere) -> out f32 [B, Tout, F*(1+left+right)], rows >= ceil(n_b/subsample)
// zero; out_lens (nullable, device i32 [B]) = min(ceil(n_b / subsample), Tout)
int ttmi_stack_subsample(const float* feat, const int* n_frames, int B, int Tin, int F, int left, int right, int subsample, int Tout,
                         float* out, int* out_lens, void* stream) {
    TTMI_REQUIRE(feat && out, "stack_subsample: null pointer");
    TTMI_REQUIRE(B > 0 && B <= 65535 && Tin > 0 && F > 0 && left >= 0 && right >= 0 && subsample > 0 && Tout > 0, "stack_subsample: bad dims");
    hipLaunchKernelGGL(stack_subsample_kernel, dim3(Tout, B), dim3(256), 0, static_cast<hipStream_t>(stream), feat, n_frames, Tin, F, left,
                       right, subsample, Tout, out, out_lens);
    TTMI_LAUNCH_CHECK("stack_subsample_kernel");
    return TTMI_OK;
}

// in place on x f32 [B, T, F]: zero rows [t0, t0+tw) and columns [f0, f0+fw) of every utterance.  The span lists are HOST arrays of (start,
// width) pairs (at most 32 each) and travel as kernel arguments: no device copy, no synchronisation.
int ttmi_spec_mask(float* x, int B, int T, int F, const int* time_spans, int n_time, const int* freq_spans, int n_freq, void* stream) {
    TTMI_REQUIRE(x && B > 0 && B <= 65535 && T > 0 && F > 0, "spec_mask: bad arguments");
    TTMI_REQUIRE(n_time >= 0 && n_time <= MAX_SPANS && n_freq >= 0 && n_freq <= MAX_SPANS && (n_time == 0 || time_spans) && (n_freq == 0 || freq_spans),
                 "spec_mask: at most %d spans per axis", MAX_SPANS);
    Spans sp;
    sp.n_time = n_time; sp.n_freq = n_freq;
    for (int i = 0; i < n_time; ++i) {
        sp.t0[i] = time_spans[2 * i]; sp.tw[i] = time_spans[2 * i + 1];
        TTMI_REQUIRE(sp.t0[i] >= 0 && sp.tw[i] >= 0, "spec_mask: negative time span");
    }
    for (int i = 0; i < n_freq; ++i) {
        sp.f0[i] = freq_spans[2 * i]; sp.fw[i] = freq_spans[2 * i + 1];
        TTMI_REQUIRE(sp.f0[i] >= 0 && sp.fw[i] >= 0, "spec_mask: negative frequency span");
    }
    if (n_time + n_freq == 0) return TTMI_OK;
    hipLaunchKernelGGL(spec_mask_kernel, dim3(T, B), dim3(256), 0, static_cast<hipStream_t>(stream), x, T, F, sp);
    TTMI_LAUNCH_CHECK("spec_mask_kernel");
    return TTMI_OK;
}

}  // extern "C"
