"""Audio encoder stack (reference tt/encoder.py): per-layer learnable tables r_emb[K,H,Dh],
r_w_bias[H,Dh], r_bias[K,H] (randn init, :18-20) + one RelLearnableDecoderLayer named
`MultiHeadAttention`; BuildEncoder(config)(inputs[B,T,d], mask=None) -> [B,T,d]."""
import torch
import torch.nn as nn

from ttmi import ops
from tt.transformer import RelLearnableDecoderLayer, as_mask_spec


class BaseEncoder(nn.Module):
    def __init__(self, k_len, n_head, d_model, d_head, d_inner, dropout, **kwargs):
        super().__init__()
        self.r_emb = nn.Parameter(torch.randn((k_len, n_head, d_head), dtype=torch.float32))
        self.r_w_bias = nn.Parameter(torch.randn((n_head, d_head), dtype=torch.float32))
        self.r_bias = nn.Parameter(torch.randn((k_len, n_head), dtype=torch.float32))
        self.MultiHeadAttention = RelLearnableDecoderLayer(n_head, d_model, d_head, d_inner, dropout, **kwargs)

    def forward_bm(self, x, mask, x16=None, want16=False):
        return self.MultiHeadAttention.forward_bm(x, self.r_emb, self.r_w_bias, self.r_bias, mask, x16=x16, want16=want16)

    def forward(self, inputs, enc_attn_mask=None):      # reference contract: time-major [T, B, d]
        assert inputs.dim() == 3
        return self.MultiHeadAttention(inputs, self.r_emb, self.r_w_bias, self.r_bias, enc_attn_mask)


class BuildEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.layers = nn.ModuleList([
            BaseEncoder(k_len=config.enc.max_input_length, n_head=config.enc.n_head, d_model=config.enc.d_model,
                        d_head=config.enc.d_head, d_inner=config.enc.d_inner, dropout=config.dropout)
            for _ in range(config.enc.n_layer)])
        # the attention sub-layer of the first layer is the last audio-encoder node of a backward pass: it launches whatever weight
        # gradients are still queued for a grouped launch (ttmi.ops.WgradQueue; plain attribute, not a parameter or buffer)
        self.layers[0].MultiHeadAttention.dec_attn.first_layer = True
        self.layers[0].MultiHeadAttention.pos_ff.first_layer = True

    def forward(self, inputs, mask=None):
        ops.weights_fresh()
        spec = as_mask_spec(mask, inputs.size(0), inputs.size(1))     # converted once, shared by every layer
        # fused layers hand their output to the next one in both forms: f32 (the residual stream) and bf16 (the next layer's GEMM operand)
        x, x16 = inputs, None
        fused = x.is_cuda and all(layer.MultiHeadAttention.fused() for layer in self.layers)
        for i, layer in enumerate(self.layers):
            if fused and i + 1 < len(self.layers):
                x, x16 = layer.forward_bm(x, spec, x16=x16, want16=True)
            else:
                x = layer.forward_bm(x, spec, x16=x16)
        return x
