"""Drop-in for `warprnnt_pytorch` (HawkAaron/warp-transducer pytorch_binding), the
third-party loss the reference imports at train.py:13 and calls at train.py:53,231:

    criterion = RNNTLoss()                      # blank=0, reduction='mean'
    loss = criterion(logits, targets.int(), inputs_length.int(), targets_length.int())

Same call contract: un-normalised logits in, softmax taken
internally, int32 labels/lengths, gradient w.r.t. the logits, 'mean' divides by
the batch and returns shape [1].  The arithmetic runs in libttmi's HIP kernels
(csrc/rnnt.hip); there is no CPU path here.
"""
import os
import weakref

import torch

from ttmi import ops

__all__ = ["RNNTLoss", "rnnt_loss", "check_lengths"]


_max_cache = {}     # id(tensor) -> (weakref, tensor._version, max): the length check costs a device-to-host sync; a lengths tensor that
                    # is the SAME object with the SAME version counter as last time has the same contents, so its maximum is reused


def _cached_max(t):
    ent = _max_cache.get(id(t))
    if ent is not None and ent[0]() is t and ent[1] == t._version:
        return ent[2]
    v = int(t.max())
    if len(_max_cache) > 64:
        _max_cache.clear()
    _max_cache[id(t)] = (weakref.ref(t), t._version, v)
    return v


def _certify(acts, labels, act_lens, label_lens, check_lengths):
    for name, t in (("labels", labels), ("label_lengths", label_lens), ("lengths", act_lens)):
        if t.dtype is not torch.int32:
            raise TypeError("%s must be int32" % name)
        if not t.is_contiguous():
            raise ValueError("%s must be contiguous" % name)
    if acts.dtype not in (torch.float32, torch.bfloat16):       # bf16: the MI355X bf16 pipeline's own logits
        raise TypeError("acts must be float32 (or the bf16 logits of the bf16 pipeline)")
    if acts.dim() != 4:
        raise ValueError("acts must have 4 dimensions (batch, T, U+1, vocab)")
    if labels.dim() != 2 or act_lens.dim() != 1 or label_lens.dim() != 1:
        raise ValueError("labels must be 2-D, lengths 1-D")
    if act_lens.shape[0] != acts.shape[0] or label_lens.shape[0] != acts.shape[0]:
        raise ValueError("must have a length per example")
    if labels.shape[0] != acts.shape[0] or labels.shape[1] != acts.shape[2] - 1:
        raise ValueError("labels must be [batch, U] with U+1 == acts.shape[2]")
    if check_lengths:   # one tiny D2H sync, as in warp-transducer's certify_inputs
        if _cached_max(act_lens) != acts.shape[1]:
            raise ValueError("Input length mismatch")
        if _cached_max(label_lens) + 1 != acts.shape[2]:
            raise ValueError("Output length mismatch")


def check_lengths(labels, act_lens, label_lens, B, T, U1, check_max):
    """the label / length part of warp-transducer's certify_inputs for callers that never hold an `acts` tensor (the fused joint + loss)"""
    for name, t in (("labels", labels), ("label_lengths", label_lens), ("lengths", act_lens)):
        if t.dtype is not torch.int32:
            raise TypeError("%s must be int32" % name)
        if not t.is_contiguous():
            raise ValueError("%s must be contiguous" % name)
    if labels.dim() != 2 or act_lens.dim() != 1 or label_lens.dim() != 1:
        raise ValueError("labels must be 2-D, lengths 1-D")
    if act_lens.shape[0] != B or label_lens.shape[0] != B:
        raise ValueError("must have a length per example")
    if labels.shape[0] != B or labels.shape[1] != U1 - 1:
        raise ValueError("labels must be [batch, U] with U+1 == acts.shape[2]")
    if check_max:
        if _cached_max(act_lens) != T:
            raise ValueError("Input length mismatch")
        if _cached_max(label_lens) + 1 != U1:
            raise ValueError("Output length mismatch")


class _RNNTLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, acts, labels, act_lens, label_lens, blank, reduction):
        B, T, U1, _ = acts.shape
        acts_c = acts if ops.row_pitch(acts) is not None else acts.contiguous()     # row-padded views are consumed in place
        ws = ops.rnnt_workspace(B, T, U1, acts.device)
        costs = ops.rnnt_loss_fwd(acts_c, labels, act_lens, label_lens, blank, ws)
        ctx.save_for_backward(acts_c, labels, act_lens, label_lens, ws)
        ctx.blank, ctx.reduction = blank, reduction
        if reduction == "none":
            return costs
        out = costs.sum().reshape(1)
        return out / B if reduction == "mean" else out

    @staticmethod
    def backward(ctx, grad_out):
        acts, labels, act_lens, label_lens, ws = ctx.saved_tensors
        B = acts.shape[0]
        go = grad_out.contiguous().float()
        per_utt = ctx.reduction == "none"
        scale = 1.0 / B if ctx.reduction == "mean" else 1.0
        grad = ops.rnnt_loss_bwd(acts, labels, act_lens, label_lens, ctx.blank, ws, go, 1 if per_utt else 0, scale)
        return grad, None, None, None, None, None


def rnnt_loss(acts, labels, act_lens, label_lens, blank=0, reduction="mean", check_lengths=None):
    if check_lengths is None:
        check_lengths = os.environ.get("TTMI_CHECK_LENGTHS", "1") != "0"
    if not acts.is_cuda:
        raise ValueError("RNNTLoss: acts must live on the GPU (the MI355X build has no CPU path)")
    labels, act_lens, label_lens = (t.to(acts.device) for t in (labels, act_lens, label_lens))
    _certify(acts, labels, act_lens, label_lens, check_lengths)      # (shape / dtype queries only: a DeferredLogits handle stays a handle)
    fused = getattr(acts, "rnnt_loss", None)
    if fused is not None:
        # `acts` is the handle Transducer.forward returns in the bf16 pipeline (tt.model.DeferredLogits): joint + loss run as one fused op on
        # the encoder states it carries and the logits are never formed - train.py:51-53 as written, on the path of Transducer.loss
        out = fused(labels, act_lens, label_lens, int(blank), reduction)
        if out is not None:
            return out
        acts = acts.materialize()           # (per-utterance costs with gradients, or a handle that was already used as a tensor)
    return _RNNTLossFn.apply(acts, labels, act_lens, label_lens, int(blank), reduction)


class RNNTLoss(torch.nn.Module):
    """RNNTLoss(blank=0, reduction='mean')(acts, labels, act_lens, label_lens)"""

    def __init__(self, blank=0, reduction="mean", check_lengths=None):
        super().__init__()
        if reduction not in ("mean", "sum", "none"):
            raise ValueError("reduction must be 'mean', 'sum' or 'none'")
        self.blank, self.reduction, self.check_lengths = blank, reduction, check_lengths

    def forward(self, acts, labels, act_lens, label_lens):
        return rnnt_loss(acts, labels, act_lens, label_lens, self.blank, self.reduction, self.check_lengths)
