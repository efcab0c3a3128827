"""The reference's tt/utils.py names that belong to the accelerated path: AttrDict (tt/utils.py:11-27), the mask builders
(:233-251), the feature front-end (:120-151,182-214,297-329, on the GPU: ttmi.frontend) and save_model (:80-91).  Logging, scoring and
file helpers of that file are outside it and resolve to the reference's own module when it is on sys.path."""
import importlib.util
import os
import sys

import torch

_REF = None


def _reference_utils():
    """the reference's own tt/utils.py (another `tt/utils.py` further down sys.path), loaded on first use"""
    global _REF
    if _REF is None:
        here = os.path.dirname(os.path.abspath(__file__))
        for entry in sys.path:
            cand = os.path.join(entry or ".", "tt", "utils.py")
            if os.path.isfile(cand) and os.path.dirname(os.path.abspath(cand)) != here:
                spec = importlib.util.spec_from_file_location("tt._reference_utils", cand)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)     # needs the reference's own dependencies (librosa, editdistance, ...)
                _REF = mod
                break
        else:
            raise ImportError("no reference tt/utils.py found on sys.path behind the ttmi overlay")
    return _REF


def __getattr__(name):
    """PEP 562: names this hot-path subset does not define (init_logger, save_model, get_feature, computer_cer, ...) are
    served by the reference's tt/utils.py, so `from tt.utils import AttrDict, init_logger, ...` keeps working unchanged"""
    if name.startswith("__"):
        raise AttributeError(name)
    try:
        return getattr(_reference_utils(), name)
    except ImportError as e:
        raise AttributeError("tt.utils.%s is not part of the accelerated path and the reference's tt/utils.py could not be "
                             "loaded (%s)" % (name, e)) from e



class AttrDict(dict):
    """dict with attribute access; ABSENT KEYS READ AS None (callers rely on it, e.g.
    config.share_embedding); nested dicts are wrapped lazily on first access."""

    def __getattr__(self, item):
        if item not in self:
            return None
        value = self[item]
        if type(value) is dict:
            value = self[item] = AttrDict(value)
        return value

    def __setattr__(self, item, value):
        self.__dict__[item] = value


def look_ahead_mask(label):
    """bool [U, U], True above the diagonal (label position i may not see j > i)."""
    n = label.size(1)
    return torch.ones(n, n, dtype=label.dtype, device=label.device).triu(1).bool()


def context_mask(audio, left_context=10, right_context=2):
    """0/1 [T, T] in audio's dtype (not bool, as in the reference): 1 where frame i may NOT see j,
    i.e. j > i + right_context or j < i - left_context."""
    n = audio.size(1)
    ones = torch.ones(n, n, dtype=audio.dtype, device=audio.device)
    return ones.triu(right_context + 1) + ones.tril(-left_context - 1)


def chunk_mask(audio, chunk=16, left_context=64):
    """Block-streaming mask (no reference counterpart; BASELINE config 4): frame i sees its own block of
    `chunk` frames plus `left_context` frames before the block."""
    n = audio.size(1)
    i = torch.arange(n, device=audio.device)[:, None]
    j = torch.arange(n, device=audio.device)[None, :]
    lo = (i // chunk) * chunk - left_context
    hi = (i // chunk + 1) * chunk - 1
    return ((j < lo) | (j > hi)).to(audio.dtype)


# ---- feature front-end with the reference's names (tt/utils.py:120-151,182-214,297-329).  DEVICE tensors run on the HIP kernels of
# ttmi.frontend (device tensor in -> device tensor out).  Host inputs (numpy arrays, CPU tensors) are what the reference's own data
# loading hands these names: tt/dataset.py calls get_feature2 + concat_frame inside AudioDataset.__getitem__, which train.py:174-184 runs
# in fork-started DataLoader workers AFTER model.cuda() - a forked child must never touch HIP.  Those calls are served by the reference's
# own numpy functions (the overlay's usual forwarding, see __getattr__); this repo holds no CPU arithmetic for them.  The batched GPU
# front-end is the opt-in ttmi.frontend.FeaturePipeline.
def _on_device(x):
    """True: run the HIP kernels.  Host data, DataLoader workers and forked children of a process that initialised the GPU never do."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda):
        return False
    from torch.utils.data import get_worker_info
    if get_worker_info() is not None or torch.cuda._is_in_bad_fork():
        raise RuntimeError("tt.utils front-end: a device tensor inside a DataLoader worker / forked child; hand the worker host data "
                           "(served by the reference's numpy code) or run ttmi.frontend.FeaturePipeline in the training process")
    return True


def _host(name):
    """the reference's own implementation of `name` for host data"""
    try:
        return getattr(_reference_utils(), name)
    except ImportError as e:
        raise ValueError("tt.utils.%s: host input (numpy / CPU tensor) is served by the reference's tt/utils.py, which could not be loaded "
                         "(%s); the MI355X path takes device tensors (there is no CPU path in this package)" % (name, e)) from e


def concat_frame(features, left_context_width, right_context_width):
    """[T, F] -> [T, F * (1 + left + right)]: `left` past frames | frame | future frames (placed as the reference places them)"""
    if not _on_device(features):
        return _host("concat_frame")(features, left_context_width, right_context_width)
    from ttmi import frontend
    return frontend.stack_subsample(features.float()[None], None, left_context_width, right_context_width, 1)[0][0]


def subsampling(features, subsample=3):
    """every `subsample`-th row from row 0 (a strided copy; the fused path is ttmi.frontend.stack_subsample)"""
    if isinstance(features, torch.Tensor):
        return features[::subsample].contiguous()
    return features[::subsample].copy()


def _log_mel(name, wave_data, framerate, feature_dim, mode):
    if not _on_device(wave_data):
        return _host(name)(wave_data, framerate, feature_dim)
    from ttmi import frontend
    w = wave_data.to(torch.int16)
    n = torch.tensor([w.numel()], dtype=torch.int32, device=w.device)
    return frontend.log_mel(w.reshape(1, -1), n, framerate, feature_dim, mode)[0]


def get_feature(wave_data, framerate, feature_dim=128):
    """int16 samples -> natural-log mel spectrogram [1 + n // 160, feature_dim] (zeros where the power is 0)"""
    return _log_mel("get_feature", wave_data, framerate, feature_dim, "ln")


def get_feature2(wave_data, framerate, feature_dim=128):
    """int16 samples -> log10 mel spectrogram (zero power -> log10 of the float64 epsilon)"""
    return _log_mel("get_feature2", wave_data, framerate, feature_dim, "log10")


def get_final_feature(samples, sample_rate=16000, feature_dim=128, left=3, right=0, subsample=3):
    if not _on_device(samples):
        return _host("get_final_feature")(samples, sample_rate, feature_dim, left, right, subsample)
    from ttmi import frontend
    w = samples.to(torch.int16)
    n = torch.tensor([w.numel()], dtype=torch.int32, device=w.device)
    mel = frontend.log_mel(w.reshape(1, -1), n, sample_rate, feature_dim, "ln")
    return frontend.stack_subsample(mel, None, left, right, subsample)[0][0]


def time_mask_augment(inputs, max_mask_time=5, mask_num=10):
    """in place on [B, T, F]: `mask_num` row spans zeroed for the whole batch; widths / starts drawn exactly as the reference draws them"""
    from ttmi import frontend
    if not (isinstance(inputs, torch.Tensor) and inputs.is_cuda):
        raise ValueError("time_mask_augment: the MI355X path masks device tensors (train.py:37-44 moves the batch to the GPU first)")
    return frontend.spec_mask_(inputs, time_spans=frontend.draw_spans(inputs.shape[1], max_mask_time, mask_num))


def frequency_mask_augment(inputs, max_mask_frequency=5, mask_num=10):
    from ttmi import frontend
    if not (isinstance(inputs, torch.Tensor) and inputs.is_cuda):
        raise ValueError("frequency_mask_augment: the MI355X path masks device tensors (train.py:37-44 moves the batch to the GPU first)")
    return frontend.spec_mask_(inputs, freq_spans=frontend.draw_spans(inputs.shape[2], max_mask_frequency, mask_num))


def save_model(model, optimizer, config, save_name):
    """the reference's checkpoint layout (tt/utils.py:80-91)"""
    from ttmi.train import save_checkpoint
    save_checkpoint(model, optimizer, save_name, multi_gpu=config.training.num_gpu > 1)


def count_parameters(model):
    """(total, audio-encoder, label-encoder) parameter counts as train.py:221-228 unpacks them (tt/utils.py:57-66: names containing
    'encoder' / 'decoder')"""
    n_params = sum(p.nelement() for p in model.parameters())
    enc = sum(p.nelement() for n, p in model.named_parameters() if "encoder" in n)
    dec = sum(p.nelement() for n, p in model.named_parameters() if "decoder" in n and "encoder" not in n)
    return n_params, enc, dec
