#!/usr/bin/env python3
"""Round-2 fixtures, produced by running the REFERENCE on CPU in the build container (needs /root/reference; never runs on the GPU box):

* tests/golden/frontend.npz  - the reference's own tt.utils.concat_frame / subsampling / frequency_mask_augment / time_mask_augment
  (tt/utils.py:120-151,297-329) and Dataset.pad (tt/dataset.py:40-57) on seeded inputs, including the RNG protocol of the masks
  (numpy's global generator for the widths, `random` for the starts).
* tests/golden/streaming.npz - the reference's sliding-window recogniser, `StreamRec.start_rec`
  (audio/streamRec_unlimit_dynamic_window.py:99-218), run AS IS on a synthetic recording: the class is imported from the reference and
  its own loop executes; only its I/O is replaced - placeholder modules for pyaudio / tkinter (microphone, GUI: no arithmetic), a
  temporary checkpoint + vocabulary written by this script, and `get_feature` (librosa, absent here) replaced by a deterministic
  log-mel function whose outputs are stored in the fixture, so every later stage sees exactly the recorded numbers.

Nothing from the reference is copied: inputs go in, tensors come out.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_r2.py
"""
import os
import random
import sys
import tempfile
import types
from unittest import mock

sys.dont_write_bytecode = True
REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REF)
for _m in ("librosa", "editdistance", "pyaudio"):
    sys.modules.setdefault(_m, types.ModuleType(_m))
sys.modules["pyaudio"].PyAudio = mock.MagicMock          # microphone handle (start_rec only opens and starts it)
sys.modules["pyaudio"].paContinue, sys.modules["pyaudio"].paComplete = 0, 1
_tk = mock.MagicMock()                                    # GUI widgets: every call is a no-op
sys.modules["tkinter"] = _tk
sys.modules["tkinter.font"] = _tk.font

import numpy as np                                        # noqa: E402
import torch                                              # noqa: E402
import yaml                                               # noqa: E402

import tt.utils as ref_utils                              # noqa: E402
from tt.utils import AttrDict                             # noqa: E402
from tt.model import Transducer                           # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def frontend_fixture():
    out = {}
    rng = np.random.default_rng(7)
    feat = rng.normal(size=(53, 24)).astype(np.float32)
    out["feat"] = feat
    for left, right in ((3, 0), (2, 1), (0, 0)):
        out["concat_%d_%d" % (left, right)] = ref_utils.concat_frame(feat, left, right)
    stacked = ref_utils.concat_frame(feat, 3, 0)
    for s in (3, 2, 1):
        out["sub_%d" % s] = ref_utils.subsampling(stacked, s)
    one = rng.normal(size=(2, 24)).astype(np.float32)         # shorter than the context: zeros everywhere off the ends
    out["short"] = one
    out["short_concat_3_0"] = ref_utils.concat_frame(one, 3, 0)
    # masks: the training loop's call order (train.py:41-44: frequency first, then time), torch tensors [B, T, F]
    batch = torch.tensor(rng.normal(size=(3, 40, 96)).astype(np.float32))
    out["batch"] = batch.numpy().copy()
    np.random.seed(11)
    random.seed(12)
    masked = ref_utils.time_mask_augment(ref_utils.frequency_mask_augment(batch.clone(), max_mask_frequency=5, mask_num=10),
                                         max_mask_time=5, mask_num=10)
    out["batch_masked_seed_11_12"] = masked.numpy()
    np.random.seed(21)
    random.seed(22)
    out["batch_time_masked_seed_21_22"] = ref_utils.time_mask_augment(batch.clone(), max_mask_time=9, mask_num=4).numpy()
    # Dataset.pad (2-D branch): zero rows up to max_input_length
    import pandas  # noqa: F401  (tt.dataset imports it)
    sys.modules.setdefault("augment", types.ModuleType("augment"))
    aa = types.ModuleType("augment.audio_augment")
    aa.audio_augment = lambda x: x
    sys.modules.setdefault("augment.audio_augment", aa)
    from tt.dataset import Dataset
    ds = Dataset.__new__(Dataset)
    ds.max_input_length, ds.max_target_length, ds.ignore_id = 30, 9, 0
    out["padded"] = ds.pad(out["sub_3"]).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "frontend.npz"), **out)
    print("frontend.npz: %d arrays" % len(out))


def synthetic_log_mel(wave, rate, dim):
    """stand-in for tt.utils.get_feature (librosa): a deterministic [1 + len//160, dim] feature of the audio window.  Its outputs are
    recorded, so its formula is irrelevant to the parity claim; it only has to vary with the audio so that tokens are emitted."""
    n = 1 + len(wave) // 160
    w = np.asarray(wave, dtype=np.float32) / 3000.0
    frames = np.stack([np.resize(w[max(0, i * 160 - 80):i * 160 + 80], 160) for i in range(n)])
    basis = np.cos(np.outer(np.arange(160), np.arange(1, dim + 1)) * np.pi / 160).astype(np.float32)
    return np.log1p((frames @ basis) ** 2).astype(np.float32)


def streaming_fixture():
    cfg = yaml.load(open(os.path.join(REF, "config", "joint_streaming.yaml")), Loader=yaml.FullLoader)
    m = cfg["model"]
    # the recogniser hard-codes 128 log-mel bins and 512-d stacked frames (:66-68): the model keeps d_model = 512 and is made small elsewhere
    m["enc"].update(n_layer=2, d_model=512, n_head=2, d_head=8, d_inner=16, max_input_length=48, left_context=6, right_context=2)
    m["dec"].update(n_layer=1, d_model=512, n_head=2, d_head=8, d_inner=16, max_target_length=16)
    m["joint"].update(input_size=1024, inner_size=16)
    m["vocab_size"] = 40
    m["dropout"] = 0.0
    tmp = tempfile.mkdtemp()
    torch.manual_seed(77)
    model = Transducer(AttrDict(m)).eval()
    with torch.no_grad():
        model.joint.project_layer.bias[0] += 2.2           # blank ahead: most frames do not emit
        model.joint.project_layer.weight.mul_(6.0)
    torch.save({"encoder": model.encoder.state_dict(), "decoder": model.decoder.state_dict(), "joint": model.joint.state_dict()},
               os.path.join(tmp, "m.chkpt"))
    with open(os.path.join(tmp, "vocab"), "w") as f:
        for i in range(40):
            f.write("w%d %d\n" % (i, i))
    cfg["training"]["load_model"] = os.path.join(tmp, "m.chkpt")
    cfg["data"]["vocab"] = os.path.join(tmp, "vocab")

    import audio.streamRec_unlimit_dynamic_window as S
    feats = []

    def recording_get_feature(wave, rate, dim):
        f = synthetic_log_mel(wave, rate, dim)
        feats.append(f)
        return f

    enc_calls = []
    with mock.patch.object(torch.Tensor, "cuda", lambda self, *a, **k: self), \
            mock.patch.object(torch.nn.Module, "cuda", lambda self, *a, **k: self), \
            mock.patch.object(S, "get_feature", recording_get_feature):
        rec = S.StreamRec(config=AttrDict(cfg))
        real_enc = rec.model.encoder.forward

        def spy(x, mask=None):
            y = real_enc(x, mask)
            enc_calls.append((x.shape[1], y.detach().numpy().copy()))
            return y

        rec.model.encoder.forward = spy
        rng = np.random.default_rng(5)
        n = 15519 * 7 + 9000                               # eight windows: first (no history), six full ones, a short last clip
        t = np.arange(n)
        wave = (3000 * np.sin(2 * np.pi * t * (200 + 150 * np.sin(t / 9000.0)) / 16000.0) + 500 * rng.normal(size=n)).astype(np.int16)
        rec.audio_data, rec.frame_num, rec.recording = wave, n, False     # the whole recording is already there and the mic has stopped
        result_holder = []
        real_reset = rec.reset_parameter
        rec.reset_parameter = lambda: (result_holder.append(list(rec.result)), real_reset())
        with torch.no_grad():
            rec.start_rec()
    tokens = result_holder[0]
    out = {"tokens": np.array(tokens, dtype=np.int64), "n_windows": np.array(len(feats)),
           "left_context": np.array(6), "right_context": np.array(2), "n_layer": np.array(2), "wave": wave}
    for i, f in enumerate(feats):
        out["win%d" % i] = f
    out["enc_call_lengths"] = np.array([c[0] for c in enc_calls])
    for i in (0, 3):                                       # the first window (no history) and a full one: the encoder outputs themselves
        out["enc_call%d" % i] = enc_calls[i][1]
    for pre, mod in (("encoder.", model.encoder), ("decoder.", model.decoder), ("joint.", model.joint)):
        for k, v in mod.state_dict().items():
            out["sd/" + pre + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "streaming.npz"), **out)
    print("streaming.npz: %d windows, %d encoder calls (lengths %s), %d tokens: %s, %.0f KB" % (
        len(feats), len(enc_calls), [c[0] for c in enc_calls], len(tokens), tokens, os.path.getsize(os.path.join(OUT, "streaming.npz")) / 1024))


def sparse_greedy_fixture():
    """the reference's own Transducer.recognize (tt/model.py:92-108) on the tiny fixture nets with the blank logit raised, so that most frames
    do NOT emit (round 1's greedy fixtures emitted on nearly every frame and barely exercised the blank branch): weights = the committed
    tiny_klong / tiny_kshort state_dicts + a blank bias, inputs = fresh seeded features of 120 frames (histories beyond both table lengths)"""
    out = {}
    for name, k_enc, k_dec in (("tiny_klong", 64, 12), ("tiny_kshort", 16, 4)):
        z = np.load(os.path.join(OUT, name + ".npz"))
        cfg = yaml.load(open(os.path.join(REF, "config", "aishell.yaml")), Loader=yaml.FullLoader)
        m = cfg["model"]
        for side, k in (("enc", k_enc), ("dec", k_dec)):
            m[side].update(n_layer=2, d_model=96, n_head=4, d_head=24, d_inner=160)
        m["enc"]["max_input_length"], m["dec"]["max_target_length"] = k_enc, k_dec
        m["joint"].update(input_size=192, inner_size=80)
        m["vocab_size"], m["dropout"] = 48, 0.0
        model = Transducer(AttrDict(m)).eval()
        for pre, mod in (("encoder.", model.encoder), ("decoder.", model.decoder), ("joint.", model.joint)):
            mod.load_state_dict({k[len("sd/" + pre):]: torch.tensor(z[k]) for k in z.files if k.startswith("sd/" + pre)})
        bias = 0.75
        with torch.no_grad():
            model.joint.project_layer.bias[0] += bias
        gen = torch.Generator().manual_seed(99)
        x = torch.randn(3, 120, 96, generator=gen)
        lens = torch.tensor([120, 77, 120])
        with torch.no_grad():
            hyp = model.recognize(x, lens)
        out[name + "/inputs"], out[name + "/lens"], out[name + "/blank_bias"] = x.numpy(), lens.numpy(), np.array(bias, dtype=np.float32)
        for b, h in enumerate(hyp):
            out["%s/tokens%d" % (name, b)] = np.array(h, dtype=np.int64)
        print(name, "sparse greedy:", [len(h) for h in hyp], "symbols for", lens.tolist(), "frames")
        # the reference's beam search (tt/model.py:110-198, width 5) on the first 48 / 31 frames of the same inputs
        blens = torch.tensor([48, 31, 48])
        with torch.no_grad():
            beams = model.recognize_beam_search(x[:, :48], blens)
        out[name + "/beam_lens"] = blens.numpy()
        for b, h in enumerate(beams):
            out["%s/beam_tokens%d" % (name, b)] = np.array(h, dtype=np.int64)
        print(name, "beam search:", [len(h) for h in beams], "symbols")
    np.savez_compressed(os.path.join(OUT, "greedy_sparse.npz"), **out)


if __name__ == "__main__":
    torch.set_num_threads(4)
    frontend_fixture()
    streaming_fixture()
    sparse_greedy_fixture()
