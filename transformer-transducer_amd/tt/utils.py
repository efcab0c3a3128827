"""Hot-path subset of the reference's tt/utils.py: AttrDict (tt/utils.py:11-27) and the two mask
builders (tt/utils.py:233-251).  Feature extraction / logging / checkpoint helpers of that file are
outside the accelerated path."""
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


def count_parameters(model):
    total = sum(p.numel() for p in model.parameters())
    return total, {n: p.numel() for n, p in model.named_parameters()}
