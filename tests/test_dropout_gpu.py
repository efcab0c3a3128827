"""Training-mode dropout of the fused sub-layers (5 reference sites) vs the numpy oracle fed the EXACT masks the HIP
kernels drew (ttmi_dropout_apply exposes them): forward, input gradient and every parameter gradient."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import tt_oracle as O

pytestmark = pytest.mark.gpu


def _layer(d=64, H=2, Dh=32, Di=96, K=16, p=0.3):
    from tt.encoder import BaseEncoder
    torch.manual_seed(11)
    return BaseEncoder(k_len=K, n_head=H, d_model=d, d_head=Dh, d_inner=Di, dropout=p).cuda().train()


@pytest.mark.parametrize("prec,tol", [("fp32", 1e-4), ("bf16", 6e-2)])
def test_layer_with_dropout_matches_oracle_with_same_masks(prec, tol, monkeypatch):
    from ttmi import ops
    from ttmi.ops import MaskSpec
    monkeypatch.setenv("TTMI_PRECISION", prec)
    p, B, L, d, Di = 0.3, 3, 21, 64, 96
    layer = _layer(p=p)
    x = torch.randn(B, L, d, generator=torch.Generator().manual_seed(3))
    cot = torch.randn(B, L, d, generator=torch.Generator().manual_seed(4))
    torch.manual_seed(77)
    s_attn = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
    s_ffn = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
    torch.manual_seed(77)                                   # the layer will draw the same two seeds
    xg = x.cuda().requires_grad_(True)
    y = layer.forward_bm(xg, MaskSpec(0))
    (y * cot.cuda()).sum().backward()

    def mult(n, seed, shape):
        return ops.dropout_multipliers(n, p, seed, "cuda").cpu().numpy().astype(np.float64).reshape(shape)

    sd = {"encoder.layers.0." + k: v.detach().cpu().numpy().astype(np.float64) for k, v in layer.state_dict().items()}
    prm = O.layer_params(sd, "encoder.", 0)
    prm["drop_attn"] = mult(B * L * d, s_attn ^ 0xA1, (B, L, d))
    prm["drop_ff_in"] = mult(B * L * Di, s_ffn ^ 0xB2, (B, L, Di))
    prm["drop_ff_out"] = mult(B * L * d, s_ffn ^ 0xC3, (B, L, d))
    prm["drop_layer"] = mult(B * L * d, s_ffn ^ 0xD4, (B, L, d))
    for k in ("drop_attn", "drop_ff_in", "drop_ff_out", "drop_layer"):
        frac = float((prm[k] == 0).mean())
        assert abs(frac - p) < 0.03, (k, frac)                # Bernoulli(p) and the four sites are decorrelated
        assert np.allclose(prm[k][prm[k] != 0], 1 / (1 - p))
    assert not np.array_equal(prm["drop_attn"], prm["drop_ff_out"])
    want, cache = O.layer_fwd(x.numpy().astype(np.float64), prm, None)
    dx, g = O.layer_bwd(cot.numpy().astype(np.float64), cache, prm)
    assert rel_err(y.detach().cpu().numpy(), want) < tol
    assert rel_err(xg.grad.cpu().numpy(), dx) < tol
    names = {v: k for k, v in O._LAYER_KEYS.items()}
    for n, prm_t in layer.named_parameters():
        assert rel_err(prm_t.grad.cpu().numpy(), g[names[n]]) < tol, n


def test_eval_mode_is_deterministic_and_dropout_free():
    from ttmi.ops import MaskSpec
    layer = _layer(p=0.5).eval()
    x = torch.randn(2, 9, 64, device="cuda")
    assert torch.equal(layer.forward_bm(x, MaskSpec(0)), layer.forward_bm(x, MaskSpec(0)))
    layer.train()
    a, b = layer.forward_bm(x, MaskSpec(0)), layer.forward_bm(x, MaskSpec(0))
    assert not torch.equal(a, b)                               # fresh seeds per call in training mode


def test_full_size_layer_on_the_persistent_kernels_matches_oracle(monkeypatch):
    """one audio-encoder layer at the C2 shapes (16384 rows, d = 512, Di = 2048, H = 8 x 64, dropout 0.1): the sizes at which the
    persistent 256x256 / 256x128 GEMMs, their ReLU + dropout / ReLU'-mask / residual epilogues, the slab kernel and the LDS-staged
    flash kernels are the ones that run.  Forward, dx and every parameter gradient against the float64 oracle with the same masks."""
    from tt.encoder import BaseEncoder
    from ttmi import ops
    from ttmi.ops import MaskSpec
    monkeypatch.setenv("TTMI_PRECISION", "bf16")
    p, B, L, d, Di, H, Dh = 0.1, 32, 512, 512, 2048, 8, 64
    torch.manual_seed(5)
    layer = BaseEncoder(k_len=512, n_head=H, d_model=d, d_head=Dh, d_inner=Di, dropout=p).cuda().train()
    x = torch.randn(B, L, d, generator=torch.Generator().manual_seed(3))
    cot = torch.randn(B, L, d, generator=torch.Generator().manual_seed(4))
    torch.manual_seed(78)
    s_attn = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
    s_ffn = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
    torch.manual_seed(78)
    xg = x.cuda().requires_grad_(True)
    y = layer.forward_bm(xg, MaskSpec(0))
    (y * cot.cuda()).sum().backward()

    def mult(n, seed, shape):
        return ops.dropout_multipliers(n, p, seed, "cuda").cpu().numpy().astype(np.float32).reshape(shape)

    sd = {"encoder.layers.0." + k: v.detach().cpu().numpy().astype(np.float64) for k, v in layer.state_dict().items()}
    prm = O.layer_params(sd, "encoder.", 0)
    prm["drop_attn"] = mult(B * L * d, s_attn ^ 0xA1, (B, L, d))
    prm["drop_ff_in"] = mult(B * L * Di, s_ffn ^ 0xB2, (B, L, Di))
    prm["drop_ff_out"] = mult(B * L * d, s_ffn ^ 0xC3, (B, L, d))
    prm["drop_layer"] = mult(B * L * d, s_ffn ^ 0xD4, (B, L, d))
    want, cache = O.layer_fwd(x.numpy().astype(np.float64), prm, None)
    dx, g = O.layer_bwd(cot.numpy().astype(np.float64), cache, prm)
    assert rel_err(y.detach().cpu().numpy(), want) < 2e-2
    assert rel_err(xg.grad.cpu().numpy(), dx) < 6e-2
    names = {v: k for k, v in O._LAYER_KEYS.items()}
    for n, prm_t in layer.named_parameters():
        assert rel_err(prm_t.grad.cpu().numpy(), g[names[n]]) < 6e-2, n


@pytest.mark.parametrize("B", [8, 40])
def test_bf16x3_layer_with_dropout_matches_oracle_with_same_masks(monkeypatch, B):
    """TTMI_PRECISION=bf16x3 at sizes where its dense products take the three-term route (1024 rows: operands as [hi | lo | hi] x [hi | hi | lo] in one
    plain launch; 5120 rows: [hi | lo] x [hi | hi] with the hi block walked a second time against the weight's lo block on the persistent kernels,
    one split of dY shared by a sub-layer's weight and input gradients; d = 512, Di = 1024: x3_worth / x3_two_block in csrc/layers.hip) in TRAINING mode: the FFN's bias + ReLU + dropout epilogue on the tripled-K kernel, the ReLU' / dropout mask of its
    dgrad as a pass of its own (relu_mask_scale), the residual accumulation of the qkv dgrad - forward, dx and every parameter gradient
    against the float64 oracle fed the same masks AND the ReLU decisions the HIP path took (one of the 1 M hidden units decided differently -
    a pre-activation within rounding noise of zero - moves the bias gradient by 1e-3 at this size), at the exact-f32 mode's 1e-4."""
    from tt.encoder import BaseEncoder
    from ttmi import ops
    from ttmi.ops import MaskSpec
    monkeypatch.setenv("TTMI_PRECISION", "bf16x3")
    p, L, d, Di, H, Dh = 0.2, 128, 512, 1024, 8, 64
    torch.manual_seed(5)
    layer = BaseEncoder(k_len=96, n_head=H, d_model=d, d_head=Dh, d_inner=Di, dropout=p).cuda().train()
    x = torch.randn(B, L, d, generator=torch.Generator().manual_seed(3))
    cot = torch.randn(B, L, d, generator=torch.Generator().manual_seed(4))
    torch.manual_seed(79)
    s_attn = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
    s_ffn = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
    torch.manual_seed(79)
    xg = x.cuda().requires_grad_(True)
    saved = []
    real_layer_fwd = ops.layer_fwd
    monkeypatch.setattr(ops, "layer_fwd", lambda *a, **k: (saved.append(real_layer_fwd(*a, **k)), saved[-1])[1])
    y = layer.forward_bm(xg, MaskSpec(1))                    # causal mask (the label encoder's)
    (y * cot.cuda()).sum().backward()
    # the ReLU decisions the HIP path took (a1 is stored after the dropout: a1 > 0 <=> active AND kept; a dropped unit's decision is moot),
    # read from the FFN context (csrc/layers.hip FfnCtx, f32 layout: h [rows, d], then a1 [rows, Di], 256-byte aligned)
    rows = B * L
    off = (rows * d * 4 + 255) // 256 * 256 // 4
    a1 = saved[0][4][off:off + rows * Di].view(B, L, Di)

    def mult(n, seed, shape):
        return ops.dropout_multipliers(n, p, seed, "cuda").cpu().numpy().astype(np.float64).reshape(shape)

    sd = {"encoder.layers.0." + k: v.detach().cpu().numpy().astype(np.float64) for k, v in layer.state_dict().items()}
    prm = O.layer_params(sd, "encoder.", 0)
    prm["drop_attn"] = mult(B * L * d, s_attn ^ 0xA1, (B, L, d))
    prm["drop_ff_in"] = mult(B * L * Di, s_ffn ^ 0xB2, (B, L, Di))
    prm["drop_ff_out"] = mult(B * L * d, s_ffn ^ 0xC3, (B, L, d))
    prm["drop_layer"] = mult(B * L * d, s_ffn ^ 0xD4, (B, L, d))
    own = O.layer_fwd(x.numpy().astype(np.float64), prm, O.look_ahead_mask(L)[:, :, None])[1][1]
    kept = prm["drop_ff_in"] != 0
    flips = (own["relu_on"] != (a1 > 0).cpu().numpy()) & kept
    prm["relu_active"] = np.where(kept, (a1 > 0).cpu().numpy(), own["relu_on"])
    want, cache = O.layer_fwd(x.numpy().astype(np.float64), prm, O.look_ahead_mask(L)[:, :, None])
    dx, g = O.layer_bwd(cot.numpy().astype(np.float64), cache, prm)
    tol = 1e-4
    e_y, e_dx = rel_err(y.detach().cpu().numpy(), want), rel_err(xg.grad.cpu().numpy(), dx)
    names = {v: k for k, v in O._LAYER_KEYS.items()}
    worst = max(((rel_err(t.grad.cpu().numpy(), g[names[n]]), n) for n, t in layer.named_parameters()))
    print("bf16x3 layer with dropout: out %.2e, dx %.2e, worst gradient %s %.2e, ReLU units decided differently from float64: %d" % (e_y, e_dx, worst[1], worst[0], int(flips.sum())))
    assert e_y < tol and e_dx < tol and worst[0] < tol, (e_y, e_dx, worst)
