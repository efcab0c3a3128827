"""Label encoder stack (reference tt/decoder.py): Embedding(V, d, padding_idx=0) named `dec_embedding`
+ N layers of the same relative-position block; BuildDecoder(config)(tokens[B,U], mask=None) -> [B,U,d]."""
import torch
import torch.nn as nn

from ttmi import ops
from tt.transformer import RelLearnableDecoderLayer, as_mask_spec, grad_targets, label_precision


class _EmbedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tokens, weight, padding_idx):
        if tokens.dtype is not torch.long:      # nn.Embedding accepts int32 ids too; the kernels read int64
            if tokens.dtype is not torch.int32:
                raise TypeError("embedding indices must be int64 or int32, got %s" % tokens.dtype)
            tokens = tokens.long()
        tokens = tokens.contiguous()
        ctx.save_for_backward(tokens)
        ctx.shape, ctx.padding_idx, ctx.params = weight.shape, padding_idx, (weight,)
        return ops.embed_fwd(tokens, weight.detach())

    @staticmethod
    def backward(ctx, dout):
        (tokens,) = ctx.saved_tensors
        g, rets, after = grad_targets(ctx.params, ("w",))
        ops.embed_bwd(tokens, dout.contiguous(), ctx.shape[0], ctx.padding_idx, g["w"])
        for cb in after:
            cb()
        return None, rets[0], None


class BaseDecoder(nn.Module):
    def __init__(self, vocab_size, n_layer, k_len, n_head, d_model, d_head, d_inner, dropout, **kwargs):
        super().__init__()
        self.r_emb = nn.Parameter(torch.randn((k_len, n_head, d_head), dtype=torch.float32))
        self.r_w_bias = nn.Parameter(torch.randn((n_head, d_head), dtype=torch.float32))
        self.r_bias = nn.Parameter(torch.randn((k_len, n_head), dtype=torch.float32))
        self.MultiHeadAttention = RelLearnableDecoderLayer(n_head, d_model, d_head, d_inner, dropout, **kwargs)

    def forward_bm(self, x, mask, x16=None, want16=False, prec=None):
        return self.MultiHeadAttention.forward_bm(x, self.r_emb, self.r_w_bias, self.r_bias, mask, prec=prec, x16=x16, want16=want16)

    def forward(self, inputs, attn_mask=None):          # reference contract: time-major [U, B, d]
        return self.MultiHeadAttention(inputs, self.r_emb, self.r_w_bias, self.r_bias, attn_mask)


class BuildDecoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dec_embedding = nn.Embedding(config.vocab_size, config.dec.d_model, padding_idx=0)
        self.layers = nn.ModuleList([
            BaseDecoder(vocab_size=config.vocab_size, n_layer=config.dec.n_layer, k_len=config.dec.max_target_length,
                        n_head=config.dec.n_head, d_model=config.dec.d_model, d_head=config.dec.d_head,
                        d_inner=config.dec.d_inner, dropout=config.dropout)
            for _ in range(config.dec.n_layer)])
        # first layer = last node of this stack's backward pass: launches weight gradients still queued on its stream (tt/encoder.py)
        self.layers[0].MultiHeadAttention.dec_attn.first_layer = True
        self.layers[0].MultiHeadAttention.pos_ff.first_layer = True

    def forward(self, inputs, mask=None, prec=None):
        """prec (not in the reference's signature; None = label_precision() / the mode in force): precision code of this call's layers"""
        ops.weights_fresh()
        spec = as_mask_spec(mask, inputs.size(0), inputs.size(1))
        x = _EmbedFn.apply(inputs, self.dec_embedding.weight, self.dec_embedding.padding_idx)
        x16 = None
        prec = label_precision() if prec is None else prec      # (None = the mode in force; see tt.transformer.label_precision)
        fused = prec is None and x.is_cuda and all(layer.MultiHeadAttention.fused() for layer in self.layers)      # (see tt/encoder.py)
        for i, layer in enumerate(self.layers):
            if fused and i + 1 < len(self.layers):
                x, x16 = layer.forward_bm(x, spec, x16=x16, want16=True)
            else:
                x = layer.forward_bm(x, spec, x16=x16, prec=prec)
        return x
