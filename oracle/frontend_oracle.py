"""CPU oracle for the feature front-end and the sliding-window streaming recogniser.  TEST INFRASTRUCTURE ONLY
(same rules as oracle/tt_oracle.py: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it).

Reference lines followed (relative to /root/reference):
  concat_frame            tt/utils.py:120-143   (frame stacking: `left` past frames | frame | `right` future frames, zeros off the ends)
  subsampling             tt/utils.py:146-151   (every `subsample`-th row, starting at row 0)
  frequency / time masks  tt/utils.py:297-329   (SpecAugment-style zeroing over the WHOLE batch; RNG protocol below)
  Dataset.pad             tt/dataset.py:40-57   (zero rows up to max_input_length)
  get_feature(2)          tt/utils.py:182-207   (log-mel: librosa.feature.melspectrogram(y, sr, n_fft=512, hop_length=160, n_mels))
  streaming loop          audio/streamRec_unlimit_dynamic_window.py:111-216

Parity status
  * stacking / subsampling / masks / padding / the streaming window loop: PINNED against the imported reference
    (tools/gen_golden_r2.py ran the reference's own functions and its StreamRec.start_rec loop on CPU; fixtures tests/golden/frontend.npz
    and tests/golden/streaming.npz, checked by tests/test_oracle_frontend.py).
  * log-mel: the arithmetic lives in the third-party dependency librosa (requirements.txt:5 `librosa~=0.8.0`), absent from
    /root/reference and not installed.  `mel_filterbank` / `stft_power` / `log_mel` restate librosa 0.8.0's published algorithm
    (stft: centred frames, reflect padding, periodic Hann window of n_fft samples, power = |X|^2; filters.mel: Slaney scale,
    htk=False, norm='slaney', fmin=0, fmax=sr/2) as called at tt/utils.py:190,204.  The reference holds no vector for it:
    "parity unpinned" for this stage; the STFT half is cross-checked against scipy.signal.stft, the filterbank against its defining
    properties (tests/test_oracle_frontend.py).
"""
import numpy as np


# ---------------------------------------------------------------------------------------------------- stacking / subsampling
def concat_frame(features, left, right):
    T, F = features.shape
    out = np.zeros((T, F * (1 + left + right)), dtype=np.float32)
    for j in range(left + 1):                         # block j holds frame t - (left - j); block `left` is the frame itself
        s = j - left
        lo, hi = max(0, -s), min(T, T - s)
        if hi > lo:
            out[lo:hi, j * F:(j + 1) * F] = features[lo + s:hi + s]
    for i in range(right):                            # frame t + i + 1 goes to block RIGHT + i + 1 (tt/utils.py:138-141) - not left + i + 1:
        blk = right + i + 1                           # with left != right the future frames land on other blocks (the middle one included) and the
        out[0:T - i - 1, blk * F:(blk + 1) * F] = features[i + 1:T]      # last blocks stay zero.  Every shipped config has right = 0.
    return out


def subsampling(features, subsample=3):
    return features[::subsample].copy()


def pad_rows(features, max_length):
    out = np.zeros((max_length, features.shape[1]), dtype=features.dtype)
    out[:features.shape[0]] = features
    return out


# ---------------------------------------------------------------------------------------------------- SpecAugment-style masks
def draw_masks(np_rng_uniform, py_randint, length, max_width, mask_num):
    """the reference's RNG protocol (tt/utils.py:306-311,322-327): per mask, width = int(np.random.uniform(0, max_width)), then
    start = random.randint(0, length - width) (inclusive).  The two callables stand for the two generators."""
    spans = []
    for _ in range(mask_num):
        w = int(np_rng_uniform(0.0, max_width))
        spans.append((py_randint(0, length - w), w))
    return spans


def apply_masks(inputs, time_spans=(), freq_spans=()):
    out = np.array(inputs, copy=True)
    for f0, f in freq_spans:
        out[:, :, f0:f0 + f] = 0
    for t0, t in time_spans:
        out[:, t0:t0 + t, :] = 0
    return out


# ---------------------------------------------------------------------------------------------------- log-mel (librosa 0.8.0 restated)
def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin=0, fmax=sr/2, htk=False, norm='slaney') -> float32 [n_mels, 1 + n_fft/2]"""
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(sr / 2.0), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        w[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


def hann_periodic(n):
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)).astype(np.float32)


def stft_power(y, n_fft=512, hop=160):
    """|STFT|^2 of librosa.stft(y, n_fft, hop_length=hop, win_length=n_fft, window='hann', center=True, pad_mode='reflect'):
    [1 + n_fft/2, 1 + len(y)//hop] in float32"""
    y = np.asarray(y, dtype=np.float32)
    yp = np.pad(y, n_fft // 2, mode="reflect")
    n_frames = 1 + (len(yp) - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    frames = yp[idx] * hann_periodic(n_fft)[None, :]
    spec = np.fft.rfft(frames.astype(np.float64), axis=1)
    return (spec.real ** 2 + spec.imag ** 2).T.astype(np.float32)


def log_mel(wave, sr=16000, n_mels=128, mode="ln"):
    """tt/utils.py:182-207.  mode 'ln' = get_feature (np.ma.log: log where > 0, 0 elsewhere); 'log10' = get_feature2 (zeros replaced by
    the float64 epsilon before log10).  wave: int16 or float samples -> [frames, n_mels] float32"""
    S = mel_filterbank(sr, 512, n_mels) @ stft_power(np.asarray(wave).astype(np.float32))
    if mode == "ln":
        out = np.zeros_like(S)
        np.log(S, out=out, where=S > 0)
    else:
        out = np.log10(np.where(S == 0, np.finfo(float).eps, S)).astype(np.float32)
    return np.ascontiguousarray(out.T)


def final_feature(wave, sr=16000, n_mels=128, left=3, right=0, subsample=3, mode="ln"):
    """tt/utils.py:210-214 (get_final_feature) / tt/dataset.py:92-95"""
    return subsampling(concat_frame(log_mel(wave, sr, n_mels, mode), left, right), subsample)


# ---------------------------------------------------------------------------------------------------- streaming recogniser
class StreamingOracle:
    """audio/streamRec_unlimit_dynamic_window.py:111-216 from the log-mel stage on, one call per audio window.

    State: all log-mel rows so far, all stacked rows, all subsampled rows, the position of the next effective frame, the emitted
    tokens and the label-encoder state.  Per window: drop the last 3 log-mel rows (incomplete frames, :128), stack with 3 rows of
    history (:134-137), subsample continuing the global phase (:142-148), then - once more than `right_len` frames lie beyond the
    position, or on the last window - run the encoder on [position - left_len, end) under context_mask(left, right) and keep the
    centre (:166-179); greedy loop over the kept frames with the label history capped at 40 tokens (:181-204)."""

    def __init__(self, sd, left_context, right_context, n_layer, n_mels, max_history=40):
        from . import tt_oracle as O
        self.O, self.sd = O, sd
        self.left, self.right = left_context, right_context
        self.left_len, self.right_len = n_layer * left_context, n_layer * right_context
        self.max_history = max_history
        self.log_mel = np.empty((0, n_mels), dtype=np.float32)
        self.concat = np.empty((0, 4 * n_mels), dtype=np.float32)
        self.sub = np.empty((0, 4 * n_mels), dtype=np.float32)
        self.pos = 0
        self.result = []
        self.windows = []                       # (start, end, left_frame, right_frame) of every encoder call
        self.blank_frame = 0
        self.breaks = []                        # result indices in front of which the reference starts a new line (blank_frame >= 15)
        self.dec_state = O.decoder_fwd(np.array([[0]]), sd, None)[0][0, -1]           # :113-115 (a [1,1,d] state there; same numbers)

    def feed(self, win_log_mel, last=False):
        O = self.O
        feat = np.asarray(win_log_mel, dtype=np.float32)[:-3]
        n = feat.shape[0]
        self.log_mel = np.concatenate([self.log_mel, feat], 0)
        stacked = concat_frame(self.log_mel[-3 - n:], 3, 0)[3:]
        before = self.concat.shape[0]
        self.concat = np.concatenate([self.concat, stacked], 0)
        skip = (0, 2, 1)[before % 3]
        fresh = self.concat[before + skip:]
        if fresh.shape[0] > 0:                  # (the reference's np.row_stack of an empty list raises; its windows never get there)
            self.sub = np.concatenate([self.sub, subsampling(fresh, 3)], 0)
        total = self.sub.shape[0]
        emitted = []
        if total - self.pos > self.right_len or last:
            left_frame, right_frame = self.left_len, self.right_len
            start = self.pos - left_frame
            if start < 0:
                left_frame, start = self.pos, 0
            if last:
                right_frame = 0
            win = self.sub[start:total][None]
            enc = O.encoder_fwd(win, self.sd, O.context_mask(win.shape[1], self.left, self.right)[:, :, None])[0]
            # :177-179 verbatim: effect_end = -right_frame, so with right_frame == 0 (the last window, or right_context = 0) the slice is
            # [left_frame:0] = EMPTY - the reference never decodes the frames of its last window.  Mirrored, not repaired.
            eff = enc[:, left_frame:-right_frame, :]
            self.windows.append((start, total, left_frame, right_frame))
            for t in range(eff.shape[1]):
                z, _ = O.joint_fwd(eff[0, t], self.dec_state, self.sd)
                pred = int(np.argmax(z))
                if pred != 0:
                    if self.blank_frame >= 15:
                        self.breaks.append(len(self.result))
                    self.result.append(pred)
                    emitted.append(pred)
                    hist = self.result[-self.max_history:]
                    self.dec_state = O.decoder_fwd(np.array([hist]), self.sd, None)[0][0, -1]
                    self.blank_frame = 0
                elif self.result:
                    self.blank_frame += 1
            self.pos += eff.shape[1]
        return emitted
