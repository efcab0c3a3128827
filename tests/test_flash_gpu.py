"""Fused attention kernels (csrc/attn_flash.hip) in the bf16 pipeline: one encoder layer vs the float64 oracle and vs
the unfused GEMM+softmax chain, head dims 32/64, lengths off the tile grid, both table branches, every mask kind."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import tt_oracle as O

pytestmark = pytest.mark.gpu


def _run(layer, x, cot, mask):
    layer.zero_grad()
    xg = x.clone().requires_grad_(True)
    y = layer.forward_bm(xg, mask)
    (y * cot).sum().backward()
    return y.detach(), xg.grad.clone(), {n: p.grad.clone() for n, p in layer.named_parameters()}


@pytest.mark.parametrize("Dh,H,L,K,mk", [(64, 2, 70, 128, "none"), (64, 2, 150, 40, "none"), (32, 4, 33, 64, "causal"),
                                          (64, 1, 129, 16, "band"), (32, 2, 96, 200, "tensor"), (64, 2, 500, 410, "none")])
def test_layer_fused_vs_oracle_and_unfused(Dh, H, L, K, mk, monkeypatch):
    from tt.encoder import BaseEncoder
    from ttmi import ops
    from ttmi.ops import MaskSpec
    monkeypatch.setenv("TTMI_PRECISION", "bf16")
    d, Di, B = H * Dh, 96, 2
    torch.manual_seed(L + Dh)
    layer = BaseEncoder(k_len=K, n_head=H, d_model=d, d_head=Dh, d_inner=Di, dropout=0.0).cuda().eval()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, L, d, generator=g).cuda()
    cot = torch.randn(B, L, d, generator=g).cuda()
    omask = None
    if mk == "none":
        mask = MaskSpec(0)
    elif mk == "causal":
        mask, omask = MaskSpec(1), O.look_ahead_mask(L)[:, :, None]
    elif mk == "band":
        mask, omask = MaskSpec(2, left=20, right=3), O.context_mask(L, 20, 3)[:, :, None]
    else:
        m = O.chunk_mask(L, 16, 32)
        mask, omask = MaskSpec(3, tensor=torch.tensor(m[None].astype(np.uint8)).cuda()), m[:, :, None]
    ops.set_option(0, 0)
    y1, dx1, g1 = _run(layer, x, cot, mask)
    ops.set_option(0, 1)
    y0, dx0, g0 = _run(layer, x, cot, mask)            # unfused reference chain, same bf16 pipeline
    ops.set_option(0, 0)
    sd = {"encoder.layers.0." + k: v.detach().cpu().numpy().astype(np.float64) for k, v in layer.state_dict().items()}
    prm = O.layer_params(sd, "encoder.", 0)
    want, cache = O.layer_fwd(x.cpu().numpy().astype(np.float64), prm, omask)
    dxo, go = O.layer_bwd(cot.cpu().numpy().astype(np.float64), cache, prm)
    names = {v: k for k, v in O._LAYER_KEYS.items()}
    e_f = rel_err(y1.cpu().numpy(), want)
    e_u = rel_err(y0.cpu().numpy(), want)
    worst_f = max([rel_err(dx1.cpu().numpy(), dxo)] + [rel_err(g1[n].cpu().numpy(), go[names[n]]) for n in g1])
    worst_u = max([rel_err(dx0.cpu().numpy(), dxo)] + [rel_err(g0[n].cpu().numpy(), go[names[n]]) for n in g0])
    print("fused: out %.2e worst grad %.2e | unfused: out %.2e worst grad %.2e" % (e_f, worst_f, e_u, worst_u))
    assert e_f < 3e-2 and worst_f < 8e-2
    assert e_f < 2.5 * e_u + 1e-3 and worst_f < 2.5 * worst_u + 1e-3     # the fused kernels are as accurate as the unfused chain


@pytest.mark.parametrize("Dh,H,L,K", [(64, 2, 70, 128), (32, 4, 33, 16), (64, 8, 500, 512), (64, 1, 129, 40), (64, 1, 600, 64), (32, 2, 520, 700)])
def test_position_slab_kernel_matches_the_gemm_path(Dh, H, L, K, monkeypatch):
    """relpos_slab_kernel (whole slab rows streamed from LDS) vs the batched-GEMM + memset construction of the same slab:
    same bf16 operands, f32 accumulation over Dh -> the layer outputs agree to f32 rounding"""
    from tt.encoder import BaseEncoder
    from ttmi import ops
    from ttmi.ops import MaskSpec
    monkeypatch.setenv("TTMI_PRECISION", "bf16")
    d, B = H * Dh, 3
    torch.manual_seed(L)
    layer = BaseEncoder(k_len=K, n_head=H, d_model=d, d_head=Dh, d_inner=64, dropout=0.0).cuda().eval()
    x = torch.randn(B, L, d, device="cuda")
    cot = torch.randn(B, L, d, device="cuda")
    ops.set_option(8, 0)                     # the slab design (round 1): position term through HBM
    try:
        y1, dx1, g1 = _run(layer, x, cot, MaskSpec(0))
        ops.set_option(5, 1)
        y0, dx0, g0 = _run(layer, x, cot, MaskSpec(0))
    finally:
        ops.set_option(5, 0)
        ops.set_option(8, 1)
    assert rel_err(y1.cpu().numpy(), y0.cpu().numpy()) < 2e-3       # bf16 activations downstream of an f32-rounding-level change
    assert rel_err(dx1.cpu().numpy(), dx0.cpu().numpy()) < 5e-3


@pytest.mark.parametrize("prec", ["bf16", "fp32"])
def test_interval_structured_masks_take_the_row_range_path(prec, monkeypatch):
    """as_mask_spec turns a mask whose rows are single key intervals (chunk / band masks) into per-row (lo, hi) ranges (mask kind 4); the
    result is identical to the byte-mask path (kind 3) in the fused kernels (bf16) and in the softmax chain (fp32).  A mask with holes
    stays a byte mask."""
    from tt.encoder import BaseEncoder
    from tt.transformer import as_mask_spec
    from ttmi.ops import MaskSpec
    monkeypatch.setenv("TTMI_PRECISION", prec)
    Dh, H, L, K, B = 64, 2, 96, 128, 2
    torch.manual_seed(3)
    layer = BaseEncoder(k_len=K, n_head=H, d_model=H * Dh, d_head=Dh, d_inner=64, dropout=0.0).cuda().eval()
    x = torch.randn(B, L, H * Dh, device="cuda")
    cot = torch.randn(B, L, H * Dh, device="cuda")
    m = torch.tensor(O.chunk_mask(L, 16, 32)).cuda()                 # [L, L] bool-like, True = masked
    ref_style = (m != 0)[:, :, None]                                 # (qlen, klen, 1) as the reference passes it
    spec = as_mask_spec(ref_style, B, L)
    assert spec.kind == 4 and spec.tensor.shape == (1, L, 2)
    assert as_mask_spec(ref_style, B, L) is spec                     # cached per mask tensor
    y4, dx4, g4 = _run(layer, x, cot, spec)
    y3, dx3, g3 = _run(layer, x, cot, MaskSpec(3, tensor=(m != 0)[None].to(torch.uint8).contiguous()))
    assert torch.equal(y4, y3) and torch.equal(dx4, dx3)
    for n in g4:      # parameter gradients are summed by f32 atomics (split-K, column sums): same values, not the same bits
        assert rel_err(g4[n].cpu().numpy(), g3[n].cpu().numpy()) < 1e-5, n
    holes = ref_style.clone()
    holes[5, 3, 0] = True
    holes[5, 1, 0] = False
    holes[5, 5, 0] = False
    assert as_mask_spec(holes, B, L).kind == 3


@pytest.mark.parametrize("L,kind", [(500, "band"), (500, "chunk"), (1100, "band"), (1100, "chunk"), (700, "causal")])
def test_masked_tile_skipping_changes_nothing(L, kind, monkeypatch):
    """structured masks let the fused kernels skip key tiles that are masked for a whole query block (forward) and, for narrow bands on
    long sequences, the query tiles a key block never meets (backward, slabs pre-zeroed): skipped tiles contribute exact zeros, so the
    layer's output and input gradient equal the byte-mask path (kind 3, nothing skipped) bit for bit"""
    from tt.encoder import BaseEncoder
    from tt.transformer import as_mask_spec
    from ttmi.ops import MaskSpec
    monkeypatch.setenv("TTMI_PRECISION", "bf16")
    Dh, H, K, B = 64, 2, 64, 1
    torch.manual_seed(L)
    layer = BaseEncoder(k_len=K, n_head=H, d_model=H * Dh, d_head=Dh, d_inner=64, dropout=0.0).cuda().eval()
    x = torch.randn(B, L, H * Dh, device="cuda")
    cot = torch.randn(B, L, H * Dh, device="cuda")
    if kind == "band":
        m = torch.tensor(O.context_mask(L, 64, 0) != 0).cuda()
        spec = MaskSpec(2, left=64, right=0)
    elif kind == "chunk":
        m = torch.tensor(O.chunk_mask(L, 16, 64) != 0).cuda()
        spec = as_mask_spec(m[:, :, None], B, L)
        assert spec.kind == 4 and spec.left == 64 + 15 and spec.right == 15
    else:
        m = torch.tensor(O.look_ahead_mask(L) != 0).cuda()
        spec = MaskSpec(1)
    y, dx, g = _run(layer, x, cot, spec)
    y3, dx3, g3 = _run(layer, x, cot, MaskSpec(3, tensor=m[None].to(torch.uint8).contiguous()))
    assert torch.equal(y, y3) and torch.equal(dx, dx3)
    for n in g:
        assert rel_err(g[n].cpu().numpy(), g3[n].cpu().numpy()) < 1e-5, n


@pytest.mark.parametrize("Dh,H,L,K,mk", [(64, 2, 70, 128, "none"), (64, 2, 150, 40, "none"), (32, 4, 33, 64, "causal"), (64, 1, 129, 16, "band"),
                                          (32, 2, 96, 200, "chunk"), (64, 8, 500, 410, "none"), (64, 2, 1000, 410, "band"), (64, 1, 257, 300, "causal"),
                                          (64, 1, 1, 8, "none"), (64, 1, 2, 8, "none"), (32, 1, 640, 64, "none")])
def test_position_term_inside_the_kernels_vs_the_slab_design(Dh, H, L, K, mk, monkeypatch):
    """the fused kernels form BD = _rel_shift(q E^T + c) themselves (table window in LDS, MFMA blocks, skew through a private LDS image:
    tt/transformer.py:82-89,143-149) - against the round-1 design that read it from a [B, H, L, L+1] slab: same bf16 operands, f32
    accumulation in a different order, so outputs agree to bf16 rounding; and against the float64 oracle with the same bounds as the slab
    path.  Lengths on and off the tile grid, both table branches (L <= K, L > K), every structured mask, L = 1 and 2 (no upper part)."""
    from tt.encoder import BaseEncoder
    from tt.transformer import as_mask_spec
    from ttmi import ops
    from ttmi.ops import MaskSpec
    monkeypatch.setenv("TTMI_PRECISION", "bf16")
    d, B = H * Dh, 2
    torch.manual_seed(L + Dh)
    layer = BaseEncoder(k_len=K, n_head=H, d_model=d, d_head=Dh, d_inner=64, dropout=0.0).cuda().eval()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, L, d, generator=g).cuda()
    cot = torch.randn(B, L, d, generator=g).cuda()
    omask = None
    if mk == "none":
        mask = MaskSpec(0)
    elif mk == "causal":
        mask, omask = MaskSpec(1), O.look_ahead_mask(L)[:, :, None]
    elif mk == "band":
        mask, omask = MaskSpec(2, left=20, right=3), O.context_mask(L, 20, 3)[:, :, None]
    else:
        m = O.chunk_mask(L, 16, 32)
        mask, omask = as_mask_spec(torch.tensor(m != 0).cuda()[:, :, None], B, L), m[:, :, None]
        assert mask.kind == 4
    y1, dx1, g1 = _run(layer, x, cot, mask)
    ops.set_option(8, 0)
    try:
        y0, dx0, g0 = _run(layer, x, cot, mask)
    finally:
        ops.set_option(8, 1)
    e_y, e_dx = rel_err(y1.cpu().numpy(), y0.cpu().numpy()), rel_err(dx1.cpu().numpy(), dx0.cpu().numpy())
    e_g = max(rel_err(g1[n].cpu().numpy(), g0[n].cpu().numpy()) for n in g1)
    sd = {"encoder.layers.0." + k: v.detach().cpu().numpy().astype(np.float64) for k, v in layer.state_dict().items()}
    prm = O.layer_params(sd, "encoder.", 0)
    want, cache = O.layer_fwd(x.cpu().numpy().astype(np.float64), prm, omask)
    dxo, go = O.layer_bwd(cot.cpu().numpy().astype(np.float64), cache, prm)
    names = {v: k for k, v in O._LAYER_KEYS.items()}
    o_y, o_dx = rel_err(y1.cpu().numpy(), want), rel_err(dx1.cpu().numpy(), dxo)
    o_g = max(rel_err(g1[n].cpu().numpy(), go[names[n]]) for n in g1)
    print("in-kernel vs slab: out %.2e dx %.2e grads %.2e | vs oracle: out %.2e dx %.2e grads %.2e" % (e_y, e_dx, e_g, o_y, o_dx, o_g))
    # (the in-kernel forward raises its running maximum lazily, the slab kernel at every tile: P is rounded to bf16 at a different scale, so
    # the two differ by independent bf16 roundings - measured 3e-4 / 5e-3 / 2e-2 at L=500 - while their distance to the oracle is the same)
    assert e_y < 5e-3 and e_dx < 2e-2 and e_g < 4e-2
    assert o_y < 3e-2 and o_dx < 8e-2 and o_g < 8e-2


def test_streaming_config_chunk_mask_is_cached_and_equals_the_tensor_mask(monkeypatch):
    """config.streaming = {chunk, left}: the model hands the kernels per-row key intervals built once per (T, chunk, left) - the same
    MaskSpec object on every forward, no [T, T] mask tensor - and the encoder output equals the one under tt.utils.chunk_mask given as
    the reference would give it (a [T, T, 1] tensor)"""
    from tt.model import Transducer
    from tt.utils import AttrDict, chunk_mask
    monkeypatch.setenv("TTMI_PRECISION", "bf16")
    side = dict(n_layer=2, d_model=128, n_head=2, d_head=64, d_inner=96)
    cfg = AttrDict(dict(enc=dict(side, max_input_length=64), dec=dict(side, max_target_length=8), joint=dict(input_size=256, inner_size=48),
                        vocab_size=29, dropout=0.0, streaming=dict(chunk=16, left=64)))
    torch.manual_seed(0)
    model = Transducer(cfg).cuda().eval()
    x = torch.randn(2, 300, 128, device="cuda")
    a = model._audio_mask(x)
    assert a.kind == 4 and a.left == 79 and a.right == 15 and model._audio_mask(x) is a
    y1 = model.encoder(x, a)
    y2 = model.encoder(x, chunk_mask(x, 16, 64)[:, :, None])
    assert torch.equal(y1, y2)


@pytest.mark.parametrize("H,L,K,mk,B", [(2, 70, 128, "none", 2), (8, 500, 410, "none", 2), (1, 129, 16, "band", 3), (2, 512, 64, "causal", 1), (1, 1, 8, "none", 2),
                                        (2, 33, 64, "chunk", 2), (2, 700, 64, "none", 2), (1, 1100, 410, "band", 1), (2, 513, 600, "causal", 1)])
def test_one_pass_position_gradients_vs_the_gemm_launches(H, L, K, mk, B, monkeypatch):
    """attn_dqde_kernel (one workgroup per (b, h): dq = dS k + dG E, dE, dc and d r_w_bias from ONE pass over the two bf16 slabs of the
    attention backward kernel, k / E slices resident as MFMA fragments, dE rows resident as accumulators; L > 512: one workgroup per group of
    512 columns, f32 partial dq rows summed by a second launch) against the round-2 launches
    (two transposes, the dual-product dq GEMM, the dE GEMM): the same bf16 operands summed in another order - dq is rounded to bf16 once
    in both - so dx and every gradient agree to bf16 rounding; both sit at the same distance from the float64 oracle."""
    from tt.encoder import BaseEncoder
    from tt.transformer import as_mask_spec
    from ttmi import ops
    from ttmi.ops import MaskSpec
    monkeypatch.setenv("TTMI_PRECISION", "bf16")
    Dh = 64
    d = H * Dh
    torch.manual_seed(3 * L + H)
    layer = BaseEncoder(k_len=K, n_head=H, d_model=d, d_head=Dh, d_inner=64, dropout=0.0).cuda().eval()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, L, d, generator=g).cuda()
    cot = torch.randn(B, L, d, generator=g).cuda()
    omask = None
    if mk == "none":
        mask = MaskSpec(0)
    elif mk == "causal":
        mask, omask = MaskSpec(1), O.look_ahead_mask(L)[:, :, None]
    elif mk == "band":
        mask, omask = MaskSpec(2, left=20, right=3), O.context_mask(L, 20, 3)[:, :, None]
    else:
        m = O.chunk_mask(L, 16, 32)
        mask, omask = as_mask_spec(torch.tensor(m != 0).cuda()[:, :, None], B, L), m[:, :, None]
    y1, dx1, g1 = _run(layer, x, cot, mask)
    ops.set_option(11, 1)
    try:
        y0, dx0, g0 = _run(layer, x, cot, mask)
    finally:
        ops.set_option(11, 0)
    assert torch.equal(y1, y0)
    sd = {"encoder.layers.0." + k: v.detach().cpu().numpy().astype(np.float64) for k, v in layer.state_dict().items()}
    prm = O.layer_params(sd, "encoder.", 0)
    want, cache = O.layer_fwd(x.cpu().numpy().astype(np.float64), prm, omask)
    dxo, go = O.layer_bwd(cot.cpu().numpy().astype(np.float64), cache, prm)
    names = {v: k for k, v in O._LAYER_KEYS.items()}
    worst = {}
    for n in g1:
        e10 = rel_err(g1[n].cpu().numpy(), g0[n].cpu().numpy())
        e1o, e0o = rel_err(g1[n].cpu().numpy(), go[names[n]]), rel_err(g0[n].cpu().numpy(), go[names[n]])
        worst[n] = (e10, e1o, e0o)
        assert e10 < 1e-2 and e1o < max(1.3 * e0o, 5e-3), (n, e10, e1o, e0o)
    e_dx = rel_err(dx1.cpu().numpy(), dx0.cpu().numpy())
    assert e_dx < 1e-2 and rel_err(dx1.cpu().numpy(), dxo) < max(1.3 * rel_err(dx0.cpu().numpy(), dxo), 5e-3)
    pos = [n for n in g1 if n in ("r_emb", "r_bias", "r_w_bias")]
    print("one pass vs GEMM launches (H=%d L=%d %s): dx %.2e; " % (H, L, mk, e_dx) + ", ".join("%s %.1e (oracle %.1e / %.1e)" % ((n,) + worst[n]) for n in pos))


@pytest.mark.parametrize("L,K,mk", [(500, 410, "none"), (257, 300, "none"), (300, 64, "causal"), (448, 410, "band"), (320, 100, "tensor"), (512, 410, "chunk"),
                                    (193, 410, "none")])
def test_resident_forward_kernel_is_bit_identical_to_the_tiled_one(L, K, mk, monkeypatch):
    """flash_fwd_res_kernel (round 4: one workgroup per head, the position table resident in LDS, K / V in phases of 128 keys) against
    flash_fwd_rel_kernel (four workgroups per head, 64-key steps): the same operands in the same accumulation order and the same order of
    key tiles in the online softmax - outputs and every gradient that flows from them agree to the last bit; fully masked tiles that only
    one of the two skips contribute exact zeros.  Every mask kind, both table branches (L <= K, L > K), lengths off the 64- and 128-key
    grids; and against the float64 oracle with the file's bounds."""
    from tt.encoder import BaseEncoder
    from ttmi import ops
    from ttmi.ops import MaskSpec
    monkeypatch.setenv("TTMI_PRECISION", "bf16")
    H, Dh, B = 2, 64, 3
    d = H * Dh
    torch.manual_seed(L)
    layer = BaseEncoder(k_len=K, n_head=H, d_model=d, d_head=Dh, d_inner=96, dropout=0.0).cuda().eval()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, L, d, generator=g).cuda()
    cot = torch.randn(B, L, d, generator=g).cuda()
    omask = None
    if mk == "none":
        mask = MaskSpec(0)
    elif mk == "causal":
        mask, omask = MaskSpec(1), O.look_ahead_mask(L)[:, :, None]
    elif mk == "band":
        mask, omask = MaskSpec(2, left=37, right=5), O.context_mask(L, 37, 5)[:, :, None]
    elif mk == "tensor":
        m = O.chunk_mask(L, 16, 48)
        mask, omask = MaskSpec(3, tensor=torch.tensor(m[None].astype(np.uint8)).cuda()), m[:, :, None]
    else:
        m = O.chunk_mask(L, 16, 64)
        i = np.arange(L)
        lo, hi = np.maximum((i // 16) * 16 - 64, 0), np.minimum((i // 16 + 1) * 16 - 1, L - 1)
        mask = MaskSpec(4, left=64 + 15, right=15, tensor=torch.tensor(np.stack([lo, hi], -1)[None].astype(np.int32)).cuda())
        omask = m[:, :, None]
    try:
        ops.set_option(14, 0)
        y0, dx0, g0 = _run(layer, x, cot, mask)
    finally:
        ops.set_option(14, 1)
    y1, dx1, g1 = _run(layer, x, cot, mask)
    assert torch.equal(y1, y0) and torch.equal(dx1, dx0)
    for n in g0:                                                  # (dE / dc / weight gradients are f32 atomic sums: equal up to their order)
        assert rel_err(g1[n].cpu().numpy(), g0[n].cpu().numpy()) < 1e-5, n
    sd = {"encoder.layers.0." + k: v.detach().cpu().numpy().astype(np.float64) for k, v in layer.state_dict().items()}
    prm = O.layer_params(sd, "encoder.", 0)
    want, cache = O.layer_fwd(x.cpu().numpy().astype(np.float64), prm, omask)
    assert rel_err(y1.cpu().numpy(), want) < 3e-2
