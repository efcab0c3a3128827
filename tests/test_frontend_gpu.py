"""GPU front-end (csrc/frontend.hip) and the streaming recogniser (ttmi/streaming.py) through the C ABI: against the fixtures the
imported reference produced (tests/golden/frontend.npz, streaming.npz) and against the oracle (oracle/frontend_oracle.py)."""
import os
import random

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_err
from oracle import frontend_oracle as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fz():
    return np.load(os.path.join(GOLDEN, "frontend.npz"))


def test_stacking_subsampling_padding_bit_exact(fz):
    """pure data movement: bit-exact against the reference's concat_frame / subsampling / Dataset.pad, through the reference-named
    functions (device tensor in, device tensor out) and through the fused batch kernel"""
    from tt import utils as U
    from ttmi import frontend
    dev = lambda a: torch.tensor(a, device="cuda")
    for left, right in ((3, 0), (2, 1), (0, 0)):
        assert np.array_equal(U.concat_frame(dev(fz["feat"]), left, right).cpu().numpy(), fz["concat_%d_%d" % (left, right)])
    assert np.array_equal(U.concat_frame(dev(fz["short"]), 3, 0).cpu().numpy(), fz["short_concat_3_0"])
    st = U.concat_frame(dev(fz["feat"]), 3, 0)
    for s in (3, 2, 1):
        assert np.array_equal(U.subsampling(st, s).cpu().numpy(), fz["sub_%d" % s])
    # host data belongs to the reference's own numpy code (DataLoader workers, tt/dataset.py): without the reference checkout on sys.path
    # the call fails loudly instead of touching the GPU from what may be a forked worker
    if not os.path.isdir("/root/reference/tt"):
        with pytest.raises(ValueError, match="reference"):
            U.concat_frame(fz["feat"], 3, 0)
    # fused: two utterances of different lengths in one launch, padded to 30 rows like Dataset.pad
    feat = torch.zeros(2, 53, 24, device="cuda")
    feat[0] = torch.tensor(fz["feat"])
    feat[1, :20] = torch.tensor(fz["feat"][:20])
    feat[1, 20:] = 99.0                                      # beyond the second utterance's length: must never show up
    out, lens = frontend.stack_subsample(feat, torch.tensor([53, 20], dtype=torch.int32, device="cuda"), 3, 0, 3, out_len=30)
    assert lens.tolist() == [18, 7]
    assert np.array_equal(out[0].cpu().numpy(), fz["padded"])
    want1 = F.pad_rows(F.subsampling(F.concat_frame(fz["feat"][:20], 3, 0), 3), 30)
    assert np.array_equal(out[1].cpu().numpy(), want1)


def test_masks_follow_the_reference_rng_protocol(fz):
    from tt import utils as U
    x = torch.tensor(fz["batch"]).cuda()
    np.random.seed(11)
    random.seed(12)
    y = U.time_mask_augment(U.frequency_mask_augment(x.clone(), max_mask_frequency=5, mask_num=10), max_mask_time=5, mask_num=10)
    assert np.array_equal(y.cpu().numpy(), fz["batch_masked_seed_11_12"])
    np.random.seed(21)
    random.seed(22)
    y = U.time_mask_augment(x.clone(), max_mask_time=9, mask_num=4)
    assert np.array_equal(y.cpu().numpy(), fz["batch_time_masked_seed_21_22"])
    with pytest.raises(ValueError):
        U.time_mask_augment(x.cpu())


@pytest.mark.parametrize("mode", ["ln", "log10"])
def test_log_mel_vs_oracle(mode):
    """a ragged batch of synthetic recordings (one all-zero): GPU log-mel (DFT and filterbank as exact-f32 MFMA GEMMs) against the numpy
    restatement of librosa 0.8's melspectrogram.  The log makes absolute error the meaningful figure: an f32 DFT (here a 512-term f32
    dot product per bin, in librosa a complex64 FFT) leaves ~1e-4 relative error on bins 60 dB below the strongest one, i.e. ~1e-3 on
    their logarithm; the oracle transforms in float64"""
    from ttmi import frontend
    rng = np.random.default_rng(9)
    lens = [16000, 9999, 16000, 700]
    waves = np.zeros((4, 16000), dtype=np.int16)
    for b, n in enumerate(lens):
        t = np.arange(n)
        waves[b, :n] = (2500 * np.sin(2 * np.pi * (300 + 200 * b) * t / 16000.0) + 300 * rng.normal(size=n)).astype(np.int16)
    waves[2] = 0
    got = frontend.log_mel(torch.tensor(waves).cuda(), torch.tensor(lens, dtype=torch.int32).cuda(), 16000, 128, mode).cpu().numpy()
    assert got.shape == (4, 101, 128)
    for b, n in enumerate(lens):
        want = F.log_mel(waves[b, :n], 16000, 128, mode)
        nf = want.shape[0]
        assert nf == 1 + n // 160
        assert np.abs(got[b, :nf] - want).max() < 3e-3, (b, np.abs(got[b, :nf] - want).max())
        assert rel_err(got[b, :nf], want) < 2e-5
        assert (got[b, nf:] == 0).all()
    from tt import utils as U
    w0 = torch.tensor(waves[0], device="cuda")
    one = U.get_feature(w0, 16000, 128) if mode == "ln" else U.get_feature2(w0, 16000, 128)
    assert one.is_cuda and np.abs(one.cpu().numpy() - got[0]).max() < 1e-3        # (one utterance alone takes other GEMM tiles: f32 rounding differences)
    ff = U.get_final_feature(torch.tensor(waves[1, :9999], device="cuda"))
    assert rel_err(ff.cpu().numpy(), F.final_feature(waves[1, :9999])) < 1e-5


def test_feature_pipeline_batch():
    from ttmi.frontend import FeaturePipeline
    rng = np.random.default_rng(2)
    lens = [8000, 4801]
    waves = np.zeros((2, 8000), dtype=np.int16)
    for b, n in enumerate(lens):
        waves[b, :n] = (rng.normal(size=n) * 1500).astype(np.int16)
    pipe = FeaturePipeline(feature_dim=128, left_context_width=3, right_context_width=0, subsample=3, mode="log10")
    feats, flen = pipe(torch.tensor(waves).cuda(), torch.tensor(lens, dtype=torch.int32).cuda())
    assert feats.shape == (2, 17, 512) and flen.tolist() == [17, 11]
    for b, n in enumerate(lens):
        want = F.final_feature(waves[b, :n], mode="log10")
        assert np.abs(feats[b, :want.shape[0]].cpu().numpy() - want).max() < 3e-3
        assert (feats[b, want.shape[0]:] == 0).all()
    np.random.seed(3)
    random.seed(4)
    masked, _ = pipe(torch.tensor(waves).cuda(), torch.tensor(lens, dtype=torch.int32).cuda(), augment=True)
    np.random.seed(3)
    random.seed(4)
    fs = F.draw_masks(np.random.uniform, random.randint, 512, 5, 10)
    ts = F.draw_masks(np.random.uniform, random.randint, 17, 5, 10)
    assert np.array_equal(masked.cpu().numpy(), F.apply_masks(feats.cpu().numpy(), ts, fs))


def _streaming_model(z):
    from tt.model import Transducer
    from tt.utils import AttrDict
    cfg = AttrDict(dict(enc=dict(n_layer=2, d_model=512, n_head=2, d_head=8, d_inner=16, max_input_length=48, left_context=6, right_context=2),
                        dec=dict(n_layer=1, d_model=512, n_head=2, d_head=8, d_inner=16, max_target_length=16),
                        joint=dict(input_size=1024, inner_size=16), vocab_size=40, dropout=0.0))
    model = Transducer(cfg).cuda().eval()
    sd = {k[3:]: torch.tensor(z[k]) for k in z.files if k.startswith("sd/")}
    model.load_state_dict(sd, strict=True)
    return model


@pytest.mark.parametrize("block", [64, 5])
def test_streaming_recogniser_matches_the_reference_run(block):
    """the 8 audio windows on which the reference's own StreamRec.start_rec loop ran (tools/gen_golden_r2.py): identical 91 tokens, same
    encoder windows, same encoder outputs"""
    from ttmi.streaming import StreamingRecognizer
    z = np.load(os.path.join(GOLDEN, "streaming.npz"))
    rec = StreamingRecognizer(_streaming_model(z), block=block)
    assert (rec.left, rec.right, rec.left_len, rec.right_len) == (6, 2, 12, 4)
    n = int(z["n_windows"])
    per_window = [rec.feed(z["win%d" % i], last=(i == n - 1)) for i in range(n)]
    assert [w[1] - w[0] for w in rec.windows] == z["enc_call_lengths"].tolist()
    for i in (0, 3):
        assert rel_err(rec.windows[i][4][0].cpu().numpy(), z["enc_call%d" % i][0]) < 1e-4
    assert rec.result == z["tokens"].tolist()
    assert sum(per_window, []) == rec.result and per_window[-1] == []          # the last window is encoded but never decoded
    assert rec.sub.shape[0] <= 12 + 4 + 34                                      # the feature buffer stays bounded
    # and against the oracle's state machine, including where the GUI would break lines
    sd = {k[3:]: z[k] for k in z.files if k.startswith("sd/")}
    o = F.StreamingOracle(sd, 6, 2, 2, 128)
    for i in range(n):
        o.feed(z["win%d" % i], last=(i == n - 1))
    assert o.result == rec.result and o.breaks == rec.breaks
    assert [w[:4] for w in rec.windows] == o.windows


def test_streaming_from_audio_runs_the_whole_loop():
    """recording -> windows -> GPU log-mel -> recogniser, against the oracle fed its own log-mel of the same windows (the log-mel stage
    itself is 'parity unpinned': librosa is absent)"""
    from ttmi.streaming import StreamingRecognizer
    z = np.load(os.path.join(GOLDEN, "streaming.npz"))
    model = _streaming_model(z)
    wave = z["wave"][:15519 * 3 + 7000]
    rec = StreamingRecognizer(model)
    got = rec.recognize(wave)
    sd = {k[3:]: z[k] for k in z.files if k.startswith("sd/")}
    o = F.StreamingOracle(sd, 6, 2, 2, 128)
    p = 0
    while True:
        last = p + 15999 >= len(wave)
        o.feed(F.log_mel(wave[p:] if last else wave[p:p + 15999], 16000, 128, "ln"), last)
        if last:
            break
        p += 15519
    assert len(rec.windows) == len(o.windows) == 4 and [w[:4] for w in rec.windows] == o.windows
    assert got == o.result and len(got) > 0
