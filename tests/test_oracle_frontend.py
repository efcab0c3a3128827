"""Pin oracle/frontend_oracle.py (CPU only): frame stacking / subsampling / masks / padding and the streaming window loop against
fixtures produced by the imported reference (tools/gen_golden_r2.py); the log-mel stage (librosa 0.8.0, absent: "parity unpinned")
against scipy's STFT and the defining properties of the Slaney filterbank."""
import os
import random

import numpy as np
import pytest

from conftest import GOLDEN, rel_err
from oracle import frontend_oracle as F


@pytest.fixture(scope="module")
def fz():
    return np.load(os.path.join(GOLDEN, "frontend.npz"))


@pytest.fixture(scope="module")
def sz():
    z = np.load(os.path.join(GOLDEN, "streaming.npz"))
    return z, {k[3:]: z[k] for k in z.files if k.startswith("sd/")}


def test_stacking_and_subsampling_match_the_reference(fz):
    for left, right in ((3, 0), (2, 1), (0, 0)):
        assert np.array_equal(F.concat_frame(fz["feat"], left, right), fz["concat_%d_%d" % (left, right)])
    assert np.array_equal(F.concat_frame(fz["short"], 3, 0), fz["short_concat_3_0"])
    st = F.concat_frame(fz["feat"], 3, 0)
    for s in (3, 2, 1):
        assert np.array_equal(F.subsampling(st, s), fz["sub_%d" % s])
    assert np.array_equal(F.pad_rows(fz["sub_3"], 30), fz["padded"])


def test_mask_rng_protocol_matches_the_reference(fz):
    """same seeds for numpy's and python's generators -> same spans -> same zeroed batch (train.py:41-44: frequency masks first)"""
    np.random.seed(11)
    random.seed(12)
    B, T, Fd = fz["batch"].shape
    fs = F.draw_masks(np.random.uniform, random.randint, Fd, 5, 10)
    ts = F.draw_masks(np.random.uniform, random.randint, T, 5, 10)
    assert np.array_equal(F.apply_masks(fz["batch"], ts, fs), fz["batch_masked_seed_11_12"])
    np.random.seed(21)
    random.seed(22)
    ts = F.draw_masks(np.random.uniform, random.randint, T, 9, 4)
    assert np.array_equal(F.apply_masks(fz["batch"], ts), fz["batch_time_masked_seed_21_22"])


def test_streaming_loop_matches_the_reference_run(sz):
    """the reference's StreamRec.start_rec ran on these 8 audio windows (its own loop, tools/gen_golden_r2.py): same encoder window
    lengths, same encoder outputs, same 91 tokens - including the dropped first 3 frames, the 40-token label history and the last
    window that is encoded but never decoded"""
    z, sd = sz
    o = F.StreamingOracle(sd, int(z["left_context"]), int(z["right_context"]), int(z["n_layer"]), 128)
    n = int(z["n_windows"])
    from oracle import tt_oracle as O
    real = O.encoder_fwd
    outs = []

    def spy(x, sd_, mask=None):
        y = real(x, sd_, mask)
        outs.append(y[0])
        return y

    O.encoder_fwd = spy
    try:
        for i in range(n):
            o.feed(z["win%d" % i], last=(i == n - 1))
    finally:
        O.encoder_fwd = real
    assert [w[1] - w[0] for w in o.windows] == z["enc_call_lengths"].tolist()
    for i in (0, 3):
        assert rel_err(outs[i], z["enc_call%d" % i]) < 1e-5
    assert o.result == z["tokens"].tolist() and len(o.result) > 40
    assert o.windows[0][:3] == (0, 32, 0) and o.windows[-1][3] == 0


def test_stft_power_agrees_with_scipy():
    from scipy import signal
    rng = np.random.default_rng(3)
    y = (rng.normal(size=4000) * 2000).astype(np.int16).astype(np.float32)
    P = F.stft_power(y)
    assert P.shape == (257, 1 + 4000 // 160)
    yp = np.pad(y, 256, mode="reflect")
    _, _, Z = signal.stft(yp, window=signal.get_window("hann", 512, fftbins=True), nperseg=512, noverlap=512 - 160, nfft=512,
                          boundary=None, padded=False)
    Z = Z * signal.get_window("hann", 512, fftbins=True).sum()          # scipy normalises by the window sum, librosa does not
    assert rel_err(P, (np.abs(Z) ** 2)[:, :P.shape[1]]) < 1e-5


def test_mel_filterbank_properties():
    """Slaney filters: triangles on the FFT grid, non-negative, each peaking inside its band, centres linear below 1 kHz and
    geometric above, area-normalised (2 / bandwidth)"""
    w = F.mel_filterbank(16000, 512, 128).astype(np.float64)
    assert w.shape == (128, 257) and (w >= 0).all()
    freqs = np.linspace(0, 8000, 257)
    centres = F._mel_to_hz(np.linspace(F._hz_to_mel(0.0), F._hz_to_mel(8000.0), 130))
    assert abs(F._mel_to_hz(F._hz_to_mel(3000.0)) - 3000.0) < 1e-9 and abs(F._hz_to_mel(1000.0) - 15.0) < 1e-12
    lin = centres[centres < 1000]
    assert np.allclose(np.diff(lin), np.diff(lin)[0])
    geo = centres[centres > 1000]
    assert np.allclose(geo[1:] / geo[:-1], (geo[1] / geo[0]))
    for i in (5, 40, 100, 127):
        nz = np.nonzero(w[i])[0]
        assert freqs[nz[0]] > centres[i] - 1e-9 and freqs[nz[-1]] < centres[i + 2] + 1e-9
        # the continuous triangle has area 1 under the Slaney normalisation: its samples on the 31.25 Hz grid integrate to ~1
        if len(nz) > 8:
            assert abs(w[i].sum() * (freqs[1] - freqs[0]) - 1.0) < 0.05


def test_log_mel_modes():
    rng = np.random.default_rng(4)
    y = (rng.normal(size=3200) * 1000).astype(np.int16)
    a, b = F.log_mel(y, mode="ln"), F.log_mel(y, mode="log10")
    assert a.shape == b.shape == (21, 128) and a.dtype == np.float32
    assert rel_err(a, b * np.log(10.0)) < 1e-6
    z = F.log_mel(np.zeros(1600, dtype=np.int16), mode="ln")
    assert (z == 0).all()                                               # np.ma.log of zeros -> masked -> filled with 0 (tt/utils.py:191-192)
    z2 = F.log_mel(np.zeros(1600, dtype=np.int16), mode="log10")
    assert np.allclose(z2, np.log10(np.finfo(float).eps))               # tt/utils.py:205-206
    ff = F.final_feature(y)
    assert ff.shape == (7, 512)
