"""Hot-path subset of the reference's tt/utils.py: AttrDict (tt/utils.py:11-27) and the two mask
builders (tt/utils.py:233-251).  Feature extraction / logging / checkpoint helpers of that file are
outside the accelerated path."""
import torch


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
