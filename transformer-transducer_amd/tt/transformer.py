"""Layer math of the reference's tt/transformer.py on MI355X.

Same classes and parameter names (`qkv_net`, `o_net`, `layer_norm`, `CoreNet.0/3`), same forward
contracts on time-major [L, B, d] tensors.  The modules are parameter containers; forward dispatches
to one autograd Function per sub-layer whose forward/backward are single C-ABI calls into libttmi
(csrc/layers.hip).  Encoder/decoder stacks call the batch-major entry points directly and skip the
[L,B,d] <-> [B,L,d] transposes the reference performs per call (tt/encoder.py:45,50).
"""
import os

import torch
import torch.nn as nn

from ttmi import ops
from ttmi.ops import MaskSpec


def default_precision():
    """0 = exact-f32 MFMA (parity path), 1 = bf16 MFMA with f32 accumulate (throughput path), 2 = bf16x3: the parity path's f32 data flow
    with its large dense products on the bf16 MFMA in three terms (hi . hi + lo . hi + hi . lo: ~2^-16 relative per product - the quick parity
    mode; read per call from TTMI_PRECISION = fp32 | bf16 | bf16x3)."""
    v = os.environ.get("TTMI_PRECISION", "fp32").lower()
    return 1 if v in ("bf16", "1") else (2 if v in ("bf16x3", "2") else 0)


def label_precision():
    """precision code of the LABEL encoder's layers when it differs from the mode in force, else None.  In bf16 mode TTMI_LABEL_PRECISION = bf16x3 | fp32 runs the label
    stack (1632 rows per layer at C2, on its side stream beside the audio encoder) in a parity mode: one label state meets all T frames of its utterance in the joint,
    so the label encoder's bf16 rounding is ONE pattern in T lattice rows and does not average out along an alignment the way the audio encoder's per-frame rounding
    does (round 6, tools/debug/joint_weight_rounding.py: at a steep training state the label encoder's bf16 arithmetic moves every utterance's cost by -0.02 ... -0.05
    nats of 827 to the same side).  Default: unset (the label stack follows TTMI_PRECISION)."""
    if default_precision() != 1:
        return None
    v = os.environ.get("TTMI_LABEL_PRECISION", "").lower()
    return 2 if v in ("bf16x3", "2") else (0 if v in ("fp32", "0") else None)


def label_value_precision():
    """TTMI_LABEL_VALUE_PRECISION = bf16x3 (default in bf16 mode since round 6) | fp32 | off: the label encoder's states take their VALUE from a second, gradient-free pass
    in that parity mode while the gradient still flows through the bf16 pass (tt.model.Transducer._label_states) - the forward-only form of TTMI_LABEL_PRECISION: the loss
    sees accurate label states, the backward pass costs what it cost (+0.75 ms per C2 step, on the label encoder's side stream).  With the two-term weights of the
    audio encoder's f32-output GEMMs (ttmi_set_option(13, 2), the library's default) the timed mode's batch-mean loss stays within 8.1e-5 of the fp32 mode over 56 states of
    four training trajectories (profiles/r06_loss_error_batch_mean_fixed.log; 2e-4 ... 3e-4 without them).  None = off / not bf16 mode."""
    if default_precision() != 1:
        return None
    v = os.environ.get("TTMI_LABEL_VALUE_PRECISION", "bf16x3").lower()
    return 2 if v in ("bf16x3", "2") else (0 if v in ("fp32",) else None)


_mask_cache = {}     # id(mask tensor) -> (weakref, version, MaskSpec): every layer of a stack gets the same mask tensor


def as_mask_spec(mask, B, L):
    """Reference mask conventions (tt/transformer.py:154-159): None; 2-D = (klen, bsz) key mask broadcast over
    queries; 3-D = (qlen, klen, bsz|1).  Nonzero / True = masked.  Returns a MaskSpec for the kernels.  The conversion (and the
    check whether every row's unmasked keys form one interval - chunk / band / causal masks do - in which case the kernels get
    per-row [lo, hi] ranges instead of L x L bytes) is done once per mask tensor object and version."""
    if mask is None:
        return MaskSpec(0)
    if isinstance(mask, MaskSpec):
        return mask
    ent = _mask_cache.get(id(mask))
    if ent is not None and ent[0]() is mask and ent[1] == mask._version:
        return ent[2]
    m = mask != 0
    if m.dim() == 2:
        t = m.t().unsqueeze(1)                    # [b, 1, j]
    elif m.dim() == 3:
        t = m.permute(2, 0, 1)                    # [b|1, i, j]
    else:
        raise ValueError("attn_mask must be 2-D (klen, bsz) or 3-D (qlen, klen, bsz)")
    spec = MaskSpec(3, tensor=t.to(torch.uint8).contiguous())
    if t.is_cuda and t.shape[1] > 1:
        keep = ~t                                                   # [b|1, i, j], True = attend
        n = keep.sum(-1)
        lo = keep.int().argmax(-1)
        hi = t.shape[-1] - 1 - keep.flip(-1).int().argmax(-1)
        if bool(((n > 0) & (n == hi - lo + 1)).all()):             # one host sync per distinct mask tensor
            # how far the intervals reach from the diagonal (lo_i >= i - left, hi_i <= i + right): lets the backward kernel skip the
            # query tiles a key block never meets (same host sync as the interval test)
            rows = torch.arange(t.shape[1], device=t.device)
            reach = torch.stack([(rows - lo).max(), (hi - rows).max()]).clamp_(min=0).tolist()
            spec = MaskSpec(4, left=int(reach[0]), right=int(reach[1]), tensor=torch.stack([lo, hi], -1).to(torch.int32).contiguous())
    if len(_mask_cache) > 32:
        _mask_cache.clear()
    import weakref
    _mask_cache[id(mask)] = (weakref.ref(mask), mask._version, spec)
    return spec


def grad_targets(params, names, late=()):
    """Gradient buffers for a sub-layer's backward.  Parameters that `ttmi.train.FlatModel` manages (attribute
    `_ttmi_direct`, .grad = view of the flat gradient buffer) are accumulated into IN PLACE by the HIP kernels (their
    gradient semantics are += anyway) and autograd gets None for them; everything else gets a fresh zero buffer that is
    returned to autograd.  Returns (buffers by name, tuple to return to autograd, callbacks to run afterwards); with `late` (names whose
    gradients a deferred grouped launch will write, see ttmi.ops.WgradQueue) the callbacks come back as (now, later)."""
    bufs, rets, after, later = {}, [], [], []
    for n, prm in zip(names, params):
        if getattr(prm, "_ttmi_direct", False) and prm.grad is not None:
            bufs[n] = prm.grad
            rets.append(None)
            cb = getattr(prm, "_ttmi_on_grad", None)
            if cb is not None:
                (later if n in late else after).append(cb)
        else:
            bufs[n] = torch.zeros_like(prm)
            rets.append(bufs[n])
    return (bufs, tuple(rets), after) if not late else (bufs, tuple(rets), after, later)


def _defer_queue(params, names, late, rows, prec, first_layer=False):
    """the grouped-wgrad queue, if this backward pass may use it: queue enabled (FlatModel.enable_grouped_wgrads), an audio-sized layer
    (a layer of fewer rows does not fill its share of the grouped launch) and every deferred gradient written in place into a flat buffer
    (its address must still be valid when the group runs)"""
    q = ops.wgrad_queue
    if q is None or rows < 4096 or prec != 1 or (first_layer and q.immediate_first_layer):
        return None
    for n, prm in zip(names, params):
        if n in late and not (getattr(prm, "_ttmi_direct", False) and prm.grad is not None):
            return None
    return q


class _AttnFn(torch.autograd.Function):
    NAMES = ("qkv_w", "o_w", "ln_g", "ln_b", "r_emb", "r_w_bias", "r_bias")

    LATE = ("qkv_w", "o_w")

    @staticmethod
    def forward(ctx, x, qkv_w, o_w, ln_g, ln_b, r_emb, r_w_bias, r_bias, mask, prec, p_drop, seed, first_layer=False):
        x = x.contiguous()
        p = dict(zip(_AttnFn.NAMES, (t.detach() for t in (qkv_w, o_w, ln_g, ln_b, r_emb, r_w_bias, r_bias))))
        y, saved = ops.attn_fwd(x, p, mask, prec, p_drop, seed)
        ctx.save_for_backward(x, saved, *p.values())
        ctx.prec, ctx.p_drop, ctx.seed, ctx.mask, ctx.first_layer = prec, p_drop, seed, mask, first_layer
        ctx.params = (qkv_w, o_w, ln_g, ln_b, r_emb, r_w_bias, r_bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, saved, *ps = ctx.saved_tensors
        p = dict(zip(_AttnFn.NAMES, ps))
        q = _defer_queue(ctx.params, _AttnFn.NAMES, _AttnFn.LATE, x.shape[0] * x.shape[1], ctx.prec, ctx.first_layer)
        if q is not None and ops.wgrad_defer_supported(x.shape[0] * x.shape[1], x.shape[2], p["r_emb"].shape[1], p["r_emb"].shape[2],
                                                      x.shape[2], ctx.prec):          # (Di = d: this sub-layer has no inner width)
            grads, rets, after, later = grad_targets(ctx.params, _AttnFn.NAMES, _AttnFn.LATE)
            dx = ops.attn_bwd(dy.contiguous(), x, p, saved, ctx.prec, grads, ctx.p_drop, ctx.seed, ctx.mask, defer=q)
            q.add_callbacks(later)
            q.maybe_flush(force=ctx.first_layer)          # a layer boundary: launch if enough layers wait - or if nothing follows
        else:
            grads, rets, after = grad_targets(ctx.params, _AttnFn.NAMES)
            dx = ops.attn_bwd(dy.contiguous(), x, p, saved, ctx.prec, grads, ctx.p_drop, ctx.seed, ctx.mask)
            if ctx.first_layer:
                ops.wgrad_flush()
        for cb in after:
            cb()
        return (dx, *rets, None, None, None, None, None)


class _FFNFn(torch.autograd.Function):
    NAMES = ("ff_w1", "ff_b1", "ff_w2", "ff_b2", "ff_ln_g", "ff_ln_b")

    @staticmethod
    def forward(ctx, y, w1, b1, w2, b2, ln_g, ln_b, prec, p_drop, p_layer, seed, first_layer=False):
        y = y.contiguous()
        p = dict(zip(_FFNFn.NAMES, (t.detach() for t in (w1, b1, w2, b2, ln_g, ln_b))))
        z, saved = ops.ffn_fwd(y, p, prec, p_drop, p_layer, seed)
        ctx.save_for_backward(y, saved, *p.values())
        ctx.prec, ctx.drop, ctx.first_layer = prec, (p_drop, p_layer, seed), first_layer
        ctx.params = (w1, b1, w2, b2, ln_g, ln_b)
        return z

    LATE = ("ff_w1", "ff_b1", "ff_w2")

    @staticmethod
    def backward(ctx, dz):
        y, saved, *ps = ctx.saved_tensors
        p = dict(zip(_FFNFn.NAMES, ps))
        rows = y.numel() // y.shape[-1]
        if ctx.first_layer and ops.wgrad_queue is not None and ops.wgrad_queue.immediate_first_layer:
            ops.wgrad_flush()          # first node of the first layer's backward: this layer keeps its own launches, the groups behind it go now
        q = _defer_queue(ctx.params, _FFNFn.NAMES, _FFNFn.LATE, rows, ctx.prec, ctx.first_layer)
        if q is not None and ops.wgrad_defer_supported(rows, y.shape[-1], 1, 8, p["ff_w1"].shape[0], ctx.prec):     # (H, Dh = 1, 8: no heads here)
            grads, rets, after, later = grad_targets(ctx.params, _FFNFn.NAMES, _FFNFn.LATE)
            dy = ops.ffn_bwd(dz.contiguous(), y, p, saved, ctx.prec, grads, *ctx.drop, defer=q)
            q.add_callbacks(later)
        else:
            grads, rets, after = grad_targets(ctx.params, _FFNFn.NAMES)
            dy = ops.ffn_bwd(dz.contiguous(), y, p, saved, ctx.prec, grads, *ctx.drop)
        for cb in after:
            cb()
        return (dy, *rets, None, None, None, None, None)


_SUBLAYER_CALLS = bool(os.environ.get("TTMI_SUBLAYER_CALLS"))       # read once: (A/B runs and debugging only)


class _LayerFn(torch.autograd.Function):
    """One encoder layer = ONE C-ABI call per direction (ttmi_layer_fwd / ttmi_layer_bwd): the attention sub-layer and the FFN share the
    passes over the residual stream that two calls force apart (ttmi.h).  x16 / the second output: bf16 copies of the layer's input and
    output, handed from layer to layer by the stacks (never differentiated: they are copies of x and z)."""
    NAMES = _AttnFn.NAMES + _FFNFn.NAMES
    LATE = _AttnFn.LATE + _FFNFn.LATE

    @staticmethod
    def forward(ctx, x, x16, mask, prec, p_attn, seed_attn, p_ffn, p_layer, seed_ffn, first_layer, want16, *params):
        x = x.contiguous()
        p = dict(zip(_LayerFn.NAMES, (t.detach() for t in params)))
        y, z, z16, ctx_a, ctx_f = ops.layer_fwd(x, x16, p, p, mask, prec, p_attn, seed_attn, p_ffn, p_layer, seed_ffn, want16)
        ctx.save_for_backward(x, y, ctx_a, ctx_f, *p.values())
        ctx.x16 = x16
        ctx.args = (prec, mask, p_attn, seed_attn, p_ffn, p_layer, seed_ffn)
        ctx.first_layer, ctx.params = first_layer, params
        ctx.set_materialize_grads(False)               # (no zero-filled gradient for the bf16 copy: it is never differentiated)
        if z16 is None:
            return z, None
        ctx.mark_non_differentiable(z16)
        return z, z16

    @staticmethod
    def backward(ctx, dz, _dz16=None):
        x, y, ctx_a, ctx_f, *ps = ctx.saved_tensors
        if dz is None:                                 # the layer's output did not reach the loss
            return (None,) * (11 + len(ps))
        p = dict(zip(_LayerFn.NAMES, ps))
        prec, mask, p_attn, seed_attn, p_ffn, p_layer, seed_ffn = ctx.args
        rows, d = x.shape[0] * x.shape[1], x.shape[2]
        if ctx.first_layer and ops.wgrad_queue is not None and ops.wgrad_queue.immediate_first_layer:
            ops.wgrad_flush()          # first node of the first layer's backward: this layer keeps its own launches, the groups behind it go now
        q = _defer_queue(ctx.params, _LayerFn.NAMES, _LayerFn.LATE, rows, prec, ctx.first_layer)
        kw = dict(mask=mask, p_attn=p_attn, seed_attn=seed_attn, p_ffn=p_ffn, p_layer=p_layer, seed_ffn=seed_ffn)
        if q is not None and ops.wgrad_defer_supported(rows, d, p["r_emb"].shape[1], p["r_emb"].shape[2], p["ff_w1"].shape[0], prec):
            grads, rets, after, later = grad_targets(ctx.params, _LayerFn.NAMES, _LayerFn.LATE)
            dx = ops.layer_bwd(dz.contiguous(), x, ctx.x16, y, p, p, ctx_a, ctx_f, prec, grads, defer=q, **kw)
            q.add_callbacks(later)
            q.maybe_flush(force=ctx.first_layer)          # a layer boundary: launch if enough layers wait - or if nothing follows
        else:
            grads, rets, after = grad_targets(ctx.params, _LayerFn.NAMES)
            dx = ops.layer_bwd(dz.contiguous(), x, ctx.x16, y, p, p, ctx_a, ctx_f, prec, grads, **kw)
            if ctx.first_layer:
                ops.wgrad_flush()
        for cb in after:
            cb()
        return (dx, None, None, None, None, None, None, None, None, None, None, *rets)


def _drop_p(module, p):
    """dropout probability in effect (nn.Dropout semantics: active only in training mode)"""
    return float(p) if (module.training and p) else 0.0


_seed_salt = 0


def set_seed_salt(rank):
    """data-parallel ranks share one torch seed (identical initial weights, config/aishell.yaml:55) but must not share dropout masks:
    `ttmi.train.GradSync` mixes the rank into every seed drawn below"""
    global _seed_salt
    _seed_salt = (int(rank) * 0x9E3779B1) & 0x7FFFFFFF


def _new_seed(*ps):
    """one 31-bit seed per sub-layer call from torch's CPU generator (reproducible under torch.manual_seed), mixed with the
    data-parallel rank; the HIP kernels derive their counter-based masks from it and regenerate them in backward"""
    return (int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) ^ _seed_salt) if any(ps) else 0


class PositionwiseFF(nn.Module):
    """y = LN(x + W2 drop(relu(W1 LN(x) + b1)) + b2) with ONE LayerNorm used twice (tt/transformer.py:36-58)."""

    def __init__(self, d_model, d_inner, dropout, layer_norm_epsilon=1e-5):
        super().__init__()
        self.d_model, self.d_inner, self.dropout = d_model, d_inner, dropout
        self.CoreNet = nn.Sequential(nn.Linear(d_model, d_inner), nn.ReLU(inplace=True), nn.Dropout(dropout),
                                     nn.Linear(d_inner, d_model), nn.Dropout(dropout))
        self.layer_norm = nn.LayerNorm(d_model, eps=layer_norm_epsilon)

    def forward(self, inp, prec=None, p_layer=0.0):
        """p_layer: dropout the enclosing RelLearnableDecoderLayer applies to this block's output (fused here)"""
        c = self.CoreNet
        p = _drop_p(self, self.dropout)
        return _FFNFn.apply(inp, c[0].weight, c[0].bias, c[3].weight, c[3].bias, self.layer_norm.weight,
                            self.layer_norm.bias, default_precision() if prec is None else prec, p, p_layer,
                            _new_seed(p, p_layer), getattr(self, "first_layer", False))


class RelMultiHeadAttn(nn.Module):
    def __init__(self, n_head, d_model, d_head, dropout, dropatt=0, tgt_len=None, ext_len=None, mem_len=None):
        super().__init__()
        self.n_head, self.d_model, self.d_head, self.dropout = n_head, d_model, d_head, dropout
        self.qkv_net = nn.Linear(d_model, 3 * n_head * d_head, bias=False)
        self.drop = nn.Dropout(dropout)
        self.dropatt = nn.Dropout(dropatt)
        self.o_net = nn.Linear(n_head * d_head, d_model, bias=False)
        self.layer_norm = nn.LayerNorm(d_model)
        self.scale = 1 / (d_head ** 0.5)

    def forward(self, w, r_emb, r_w_bias, r_bias, attn_mask=None):
        raise NotImplementedError


class RelLearnableMultiHeadAttn(RelMultiHeadAttn):
    """Learnable-relative-position attention incl. the reference's exact _rel_shift semantics
    (tt/transformer.py:82-89,106-177)."""

    def forward_bm(self, x, r_emb, r_w_bias, r_bias, mask, prec=None):
        """batch-major [B, L, d] in/out; mask: MaskSpec."""
        p = _drop_p(self, self.dropout)
        return _AttnFn.apply(x, self.qkv_net.weight, self.o_net.weight, self.layer_norm.weight, self.layer_norm.bias,
                             r_emb, r_w_bias, r_bias, mask, default_precision() if prec is None else prec, p, _new_seed(p),
                             getattr(self, "first_layer", False))

    def forward(self, w, r_emb, r_w_bias, r_bias, attn_mask=None):
        """reference contract: w [L, B, d] time-major."""
        L, B = w.size(0), w.size(1)
        y = self.forward_bm(w.transpose(0, 1), r_emb, r_w_bias, r_bias, as_mask_spec(attn_mask, B, L))
        return y.transpose(0, 1)


class RelLearnableDecoderLayer(nn.Module):
    def __init__(self, n_head, d_model, d_head, d_inner, dropout, **kwargs):
        super().__init__()
        self.dec_attn = RelLearnableMultiHeadAttn(n_head, d_model, d_head, dropout, **kwargs)
        self.pos_ff = PositionwiseFF(d_model, d_inner, dropout)
        self.dropout = nn.Dropout(dropout)

    def forward_bm(self, x, r_emb, r_w_bias, r_bias, mask, prec=None, x16=None, want16=False):
        """batch-major [B, L, d] in/out through ONE call per direction (_LayerFn).  x16: the bf16 copy of x the previous layer returned;
        want16: return (z, bf16 copy of z) for the next layer - both only where ops.layer_fused() says the layer runs fused."""
        a, f = self.dec_attn, self.pos_ff
        if _SUBLAYER_CALLS:                            # measurement / debugging: the two sub-layer calls of rounds 1-2 instead of the layer-level one
            return f(a.forward_bm(x, r_emb, r_w_bias, r_bias, mask, prec), prec, _drop_p(self, self.dropout.p))
        prec = default_precision() if prec is None else prec
        pa, pf, pl = _drop_p(a, a.dropout), _drop_p(f, f.dropout), _drop_p(self, self.dropout.p)
        seed_attn = _new_seed(pa)                      # (drawn in the order the two sub-layer calls draw them)
        seed_ffn = _new_seed(pf, pl)
        c = f.CoreNet
        z, z16 = _LayerFn.apply(x, x16, mask, prec, pa, seed_attn, pf, pl, seed_ffn, getattr(a, "first_layer", False), want16,
                                a.qkv_net.weight, a.o_net.weight, a.layer_norm.weight, a.layer_norm.bias, r_emb, r_w_bias, r_bias,
                                c[0].weight, c[0].bias, c[3].weight, c[3].bias, f.layer_norm.weight, f.layer_norm.bias)
        return (z, z16) if want16 else z

    def fused(self, prec=None):
        """does this layer run the fused layer-level kernels (may bf16 copies of the residual stream be handed to / taken from it)?"""
        a, f = self.dec_attn, self.pos_ff
        if _SUBLAYER_CALLS:
            return False
        return ops.layer_fused(a.d_model, a.n_head, a.d_head, f.d_inner, default_precision() if prec is None else prec)

    def forward(self, input, r_emb, r_w_bias, r_bias, attn_mask=None):
        L, B = input.size(0), input.size(1)
        return self.forward_bm(input.transpose(0, 1), r_emb, r_w_bias, r_bias, as_mask_spec(attn_mask, B, L)).transpose(0, 1)
