"""Feature front-end on the GPU: waveform -> log-mel -> frame stacking -> subsampling -> zero padding, and the
training loop's time / frequency masks, as libttmi kernels (csrc/frontend.hip).  The reference does this per utterance in numpy inside
12 DataLoader workers (tt/dataset.py:84-106, tt/utils.py:120-151,182-214) and masks with 20 slice assignments per step
(train.py:41-44, tt/utils.py:297-329); here a whole batch is a handful of launches and never leaves the device.

No CPU path: every function needs the HIP library and a GPU.
"""
import ctypes
import math
import random

import numpy as np
import torch

from . import check, lib
from .ops import _p, _stream, scratch

c_int, c_long = ctypes.c_int, ctypes.c_long
_tables = {}


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, math.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_hz / f_sp + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, f / f_sp)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, math.log(6.4) / 27.0
    return np.where(m >= min_log_hz / f_sp, min_log_hz * np.exp(logstep * (m - min_log_hz / f_sp)), f_sp * m)


def tables(sr, n_fft, n_mels, device):
    """(dft [2*(n_fft/2+1), n_fft], mel_w [n_mels, n_fft/2+1]) on `device`, cached: the DFT basis with the periodic Hann window folded
    into its rows and librosa 0.8's Slaney-scale, area-normalised triangular filterbank (htk=False, norm='slaney', fmin=0, fmax=sr/2)"""
    key = (sr, n_fft, n_mels, str(device))
    t = _tables.get(key)
    if t is None:
        n = np.arange(n_fft, dtype=np.float64)
        w = (0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)).astype(np.float32).astype(np.float64)     # the window is float32 in librosa
        k = np.arange(n_fft // 2 + 1, dtype=np.float64)[:, None]
        ang = 2.0 * np.pi * k * n[None, :] / n_fft
        dft = np.empty((2 * (n_fft // 2 + 1), n_fft), dtype=np.float64)
        dft[0::2], dft[1::2] = np.cos(ang) * w, -np.sin(ang) * w
        fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
        mel_f = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(sr / 2.0), n_mels + 2))
        fdiff = np.diff(mel_f)
        ramps = mel_f[:, None] - fftfreqs[None, :]
        melw = np.stack([np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1])) for i in range(n_mels)])
        melw *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
        t = _tables[key] = (torch.tensor(dft, dtype=torch.float32, device=device), torch.tensor(melw, dtype=torch.float32, device=device))
    return t


def log_mel(wave, n_samples, sr=16000, n_mels=128, mode="ln", n_fft=512, hop=160):
    """wave int16 [B, nmax] (zero padded, device), n_samples int32 [B] (device) -> f32 [B, 1 + nmax // hop, n_mels]; rows beyond an
    utterance's 1 + n // hop frames are zero.  mode 'ln' = get_feature, 'log10' = get_feature2, 'power' = no log."""
    if not wave.is_cuda or wave.dtype is not torch.int16 or wave.dim() != 2:
        raise ValueError("log_mel: wave must be a 2-D int16 device tensor [B, samples]")
    if n_samples.dtype is not torch.int32 or not n_samples.is_cuda:
        raise ValueError("log_mel: n_samples must be an int32 device tensor")
    wave = wave.contiguous()
    B, nmax = wave.shape
    dft, melw = tables(sr, n_fft, n_mels, wave.device)
    L = lib()
    L.ttmi_logmel_ws_floats.restype = ctypes.c_size_t
    ws = scratch(L.ttmi_logmel_ws_floats(c_int(B), c_int(nmax), c_int(n_fft), c_int(hop)), wave.device)
    out = torch.empty(B, 1 + nmax // hop, n_mels, dtype=torch.float32, device=wave.device)
    check(L.ttmi_logmel(_p(wave), c_long(wave.stride(0)), _p(n_samples), c_int(B), c_int(nmax), c_int(n_fft), c_int(hop), c_int(n_mels),
                        _p(dft), _p(melw), c_int({"ln": 0, "log10": 1, "power": 2}[mode]), _p(ws), _p(out), _stream()), "ttmi_logmel")
    return out


def stack_subsample(feat, n_frames=None, left=3, right=0, subsample=3, out_len=None):
    """feat f32 [B, T, F] (device), n_frames int32 [B] (device) or None -> (f32 [B, Tout, F*(1+left+right)], int32 [B] lengths): concat_frame
    + subsampling + zero padding to Tout = out_len or ceil(T / subsample) (tt/utils.py:120-151, tt/dataset.py:52-54)"""
    if not feat.is_cuda or feat.dtype is not torch.float32 or feat.dim() != 3:
        raise ValueError("stack_subsample: feat must be a 3-D float32 device tensor [B, T, F]")
    feat = feat.contiguous()
    B, T, F = feat.shape
    Tout = out_len if out_len is not None else (T + subsample - 1) // subsample
    out = torch.empty(B, Tout, F * (1 + left + right), dtype=torch.float32, device=feat.device)
    lens = torch.empty(B, dtype=torch.int32, device=feat.device)
    check(lib().ttmi_stack_subsample(_p(feat), _p(n_frames), c_int(B), c_int(T), c_int(F), c_int(left), c_int(right), c_int(subsample),
                                     c_int(Tout), _p(out), _p(lens), _stream()), "ttmi_stack_subsample")
    return out, lens


def draw_spans(length, max_width, mask_num):
    """the reference's RNG protocol (tt/utils.py:306-311,322-327): width from numpy's global generator, start from python's `random`,
    alternately - so seeding both as a reference run does reproduces its masks exactly"""
    spans = []
    for _ in range(mask_num):
        w = int(np.random.uniform(low=0.0, high=max_width))
        spans.append((random.randint(0, length - w), w))
    return spans


def spec_mask_(x, time_spans=(), freq_spans=()):
    """in place on x f32 [B, T, F] (device): zero the given (start, width) row / column spans for the whole batch, one launch"""
    if not x.is_cuda or x.dtype is not torch.float32 or x.dim() != 3 or not x.is_contiguous():
        raise ValueError("spec_mask_: x must be a contiguous 3-D float32 device tensor")
    B, T, F = x.shape
    ts = (ctypes.c_int * (2 * len(time_spans)))(*[v for s in time_spans for v in s])
    fs = (ctypes.c_int * (2 * len(freq_spans)))(*[v for s in freq_spans for v in s])
    check(lib().ttmi_spec_mask(_p(x), c_int(B), c_int(T), c_int(F), ts, c_int(len(time_spans)), fs, c_int(len(freq_spans)), _stream()),
          "ttmi_spec_mask")
    return x


class FeaturePipeline:
    """The data path of one training batch on the device (tt/dataset.py:84-106 + train.py:32-44): padded int16 waveforms + sample counts
    -> features [B, Tmax, n_mels * (1 + left + right)], their lengths, optionally masked.  YAML keys as in config.data
    (feature_dim, left_context_width, right_context_width, subsample)."""

    def __init__(self, feature_dim=128, left_context_width=3, right_context_width=0, subsample=3, sample_rate=16000, mode="log10"):
        self.n_mels, self.left, self.right, self.sub, self.sr, self.mode = feature_dim, left_context_width, right_context_width, subsample, sample_rate, mode

    def __call__(self, wave, n_samples, augment=False, max_mask_time=5, max_mask_frequency=5, mask_num=10):
        mel = log_mel(wave, n_samples, self.sr, self.n_mels, self.mode)
        n_frames = 1 + torch.div(n_samples, 160, rounding_mode="floor").int()
        feats, lens = stack_subsample(mel, n_frames, self.left, self.right, self.sub)
        if augment:                                          # train.py:41-44: frequency spans first, then time spans
            fs = draw_spans(feats.shape[2], max_mask_frequency, mask_num)
            ts = draw_spans(feats.shape[1], max_mask_time, mask_num)
            spec_mask_(feats, ts, fs)
        return feats, lens
