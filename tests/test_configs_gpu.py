"""Parity at the sizes BASELINE.json's configs name, through the C ABI:

* C5 (configs[4]): the lattice at B=8, T=2000, U=200, V=4334 in bf16 (properties at full size + C-oracle spot checks of a full and a
  ragged utterance), an f32 lattice of the same T x U, and one audio-encoder layer at L=2000 (> K=410: the clamped-table branch of
  tt/transformer.py:128-132) against the float64 oracle, unmasked and under a band mask.
* C4 (configs[3], config/joint_streaming.yaml:24-45): the joint at J=2048, V=6485 over 33 600 lattice rows (8 column tiles in the
  wgrad's fused bias sums) against fp32 torch, and an 18-layer-shape encoder layer (Di=2048) under context_mask(64, 0) and the
  chunk(16, 64) mask against the float64 oracle.
* C2 (configs[1]) end to end: the full 12 / 6 model in fp32 at B=2, T=500, U=50: logits, loss and every parameter gradient against
  oracle.transducer_loss_and_grads, 1e-4 rel (north_star's fp32 tolerance).
"""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import tt_oracle as O
from oracle.rnnt_c import rnnt_loss_c

pytestmark = pytest.mark.gpu
TOL = 1e-4


# ----------------------------------------------------------------------------------------------- C5: long-utterance lattice
def test_c5_lattice_bf16_full_size():
    """B=8, T=2000, U=200, V=4334, bf16 logits in the joint's row-padded layout (pitch 4352): every gradient row sums to zero
    (softmax shift invariance), nothing outside a ragged utterance's lattice, pad columns zero; costs and gradients of one full and
    one ragged utterance against the C oracle evaluated on the same bf16 values."""
    from warprnnt_pytorch import RNNTLoss
    B, T, U, V = 8, 2000, 200, 4334
    Vp = (V + 63) // 64 * 64
    g = torch.Generator(device="cuda").manual_seed(11)
    buf = torch.empty(B, T, U + 1, Vp, device="cuda", dtype=torch.bfloat16)
    for b in range(B):                                   # one utterance at a time: no 56 GB f32 temporary
        buf[b] = (torch.randn(T, U + 1, Vp, device="cuda", generator=g) * 1.5).to(torch.bfloat16)
    acts = buf[..., :V].detach().requires_grad_(True)
    lab = torch.randint(1, V, (B, U), device="cuda", generator=g, dtype=torch.int32)
    tl = torch.full((B,), T, dtype=torch.int32, device="cuda")
    ul = torch.full((B,), U, dtype=torch.int32, device="cuda")
    tl[3], ul[3] = 1500, 150
    tl[5], ul[5] = 1999, 200
    raw = []
    acts.register_hook(raw.append)
    costs = RNNTLoss(reduction="none")(acts, lab, tl, ul)
    costs.sum().backward()
    gr = raw[0]
    assert gr.dtype is torch.bfloat16 and gr.stride(-2) == Vp
    full = torch.as_strided(gr, (B, T, U + 1, Vp), gr.stride())
    for b in range(B):
        gb = full[b].float()
        assert float(gb[..., V:].abs().max()) == 0
        assert float(gb[..., :V].sum(-1).abs().max()) < 4e-3            # bf16-rounded rows of magnitude <= 1
    assert float(gr[3, 1500:].float().abs().max()) == 0 and float(gr[3, :, 151:].float().abs().max()) == 0
    assert float(gr[5, 1999:].float().abs().max()) == 0
    for b in (0, 3):
        xb = acts[b:b + 1].detach().float().cpu().numpy()
        _, cb, gb = rnnt_loss_c(xb, lab[b:b + 1].cpu().numpy(), tl[b:b + 1].cpu().numpy(), ul[b:b + 1].cpu().numpy(), reduction="sum")
        assert abs(float(costs[b]) - cb[0]) / cb[0] < 1e-5
        assert rel_err(gr[b].float().cpu().numpy(), gb[0]) < 6e-3       # gradient rounded to bf16 once
        del xb, gb


def test_c5_lattice_f32_long():
    """the same T x U in f32 (B=2, one ragged): costs and gradients against the C oracle at 1e-4"""
    from warprnnt_pytorch import RNNTLoss
    B, T, U, V = 2, 2000, 200, 4334
    g = torch.Generator(device="cuda").manual_seed(12)
    acts = torch.randn(B, T, U + 1, V, device="cuda", generator=g).requires_grad_(True)
    lab = torch.randint(1, V, (B, U), device="cuda", generator=g, dtype=torch.int32)
    tl = torch.tensor([T, 1777], dtype=torch.int32, device="cuda")
    ul = torch.tensor([U, 93], dtype=torch.int32, device="cuda")
    costs = RNNTLoss(reduction="none")(acts, lab, tl, ul)
    costs.sum().backward()
    assert float(acts.grad.sum(-1).abs().max()) < 2e-5
    _, cb, gb = rnnt_loss_c(acts.detach().cpu().numpy(), lab.cpu().numpy(), tl.cpu().numpy(), ul.cpu().numpy(), reduction="sum")
    assert float(np.abs(costs.detach().cpu().numpy() - cb).max() / np.abs(cb).max()) < TOL
    assert rel_err(acts.grad.cpu().numpy(), gb) < TOL


def _layer_vs_oracle(prec, B, L, K, Di, omask, mask, monkeypatch, H=8, Dh=64, seed=0):
    from tt.encoder import BaseEncoder
    monkeypatch.setenv("TTMI_PRECISION", prec)
    d = H * Dh
    torch.manual_seed(seed + L)
    layer = BaseEncoder(k_len=K, n_head=H, d_model=d, d_head=Dh, d_inner=Di, dropout=0.0).cuda().eval()
    gen = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(B, L, d, generator=gen)
    cot = torch.randn(B, L, d, generator=gen)
    xg = x.cuda().requires_grad_(True)
    y = layer.forward_bm(xg, mask)
    (y * cot.cuda()).sum().backward()
    sd = {"encoder.layers.0." + k: v.detach().cpu().numpy().astype(np.float64) for k, v in layer.state_dict().items()}
    prm = O.layer_params(sd, "encoder.", 0)
    want, cache = O.layer_fwd(x.numpy().astype(np.float64), prm, omask)
    dxo, go = O.layer_bwd(cot.numpy().astype(np.float64), cache, prm)
    names = {v: k for k, v in O._LAYER_KEYS.items()}
    e_out = rel_err(y.detach().cpu().numpy(), want)
    e_dx = rel_err(xg.grad.cpu().numpy(), dxo)
    e_g = {n: rel_err(p.grad.cpu().numpy(), go[names[n]]) for n, p in layer.named_parameters()}
    print("%s L=%d: out %.2e dx %.2e worst grad %.2e (%s)" % (prec, L, e_out, e_dx, max(e_g.values()), max(e_g, key=e_g.get)))
    return e_out, e_dx, e_g


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("mk", ["none", "band"])
def test_c5_encoder_layer_L2000(prec, mk, monkeypatch):
    """one C2-shape audio layer (d=512, 8 x 64 heads, Di=1024, K=410) at L=2000: the table is shorter than the sequence, so relative
    positions beyond K-1 clamp onto row 0 and their gradients fold back onto it (tt/transformer.py:128-132)"""
    from ttmi.ops import MaskSpec
    L = 2000
    mask, omask = (MaskSpec(0), None) if mk == "none" else (MaskSpec(2, left=64, right=0), O.context_mask(L, 64, 0)[:, :, None])
    e_out, e_dx, e_g = _layer_vs_oracle(prec, 2, L, 410, 1024, omask, mask, monkeypatch)
    if prec == "fp32":
        assert e_out < TOL and e_dx < TOL and max(e_g.values()) < TOL
    else:
        assert e_out < 3e-2 and e_dx < 8e-2 and max(e_g.values()) < 8e-2


# ----------------------------------------------------------------------------------------------- C4: joint_streaming.yaml dims
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("mk", ["band_64_0", "chunk_16_64"])
def test_c4_encoder_layer_under_streaming_masks(prec, mk, monkeypatch):
    """config/joint_streaming.yaml:27-31 layer (d=512, 8 x 64, Di=2048, K=410) at T=500 under BASELINE configs[3]'s two masks:
    context_mask(left=64, right=0) handed over parametrically and the chunk mask (16-frame blocks + 64 left) handed over as the
    reference would, a [T, T, 1] tensor"""
    from tt.transformer import as_mask_spec
    from ttmi.ops import MaskSpec
    B, L = 4, 500
    if mk == "band_64_0":
        mask, om = MaskSpec(2, left=64, right=0), O.context_mask(L, 64, 0)
    else:
        om = O.chunk_mask(L, 16, 64)
        mask = as_mask_spec(torch.tensor(om != 0).cuda()[:, :, None], B, L)
        assert mask.kind == 4
    e_out, e_dx, e_g = _layer_vs_oracle(prec, B, L, 410, 2048, om[:, :, None], mask, monkeypatch, seed=7)
    if prec == "fp32":
        assert e_out < TOL and e_dx < TOL and max(e_g.values()) < TOL
    else:
        assert e_out < 3e-2 and e_dx < 8e-2 and max(e_g.values()) < 8e-2


def test_c4_joint_dims_vs_torch(monkeypatch):
    """the joint at config/joint_streaming.yaml:43-45 sizes (inner 2048, V=6485) over 33 600 lattice rows in the bf16 pipeline:
    v8 forward (26 column tiles), v8 dgrad with the tanh' epilogue (K = 6528), TN v8 wgrad whose fused bias column sums span 8 column
    tiles of J (the case a first version got wrong, profiles/r01_bench_c4-band_before_colsum_fix_kernel_stats.csv)"""
    from tt.model import JointNet
    monkeypatch.setenv("TTMI_PRECISION", "bf16")
    J, V, B, T, U1, de = 2048, 6485, 4, 400, 21, 512
    torch.manual_seed(J)
    joint = JointNet(2 * de, J, V).cuda()
    enc = torch.randn(B, T, de, device="cuda", requires_grad=True)
    dec = torch.randn(B, U1, de, device="cuda", requires_grad=True)
    cot = torch.randn(B, T, U1, V, device="cuda")
    out = joint(enc, dec)
    assert out.dtype is torch.bfloat16 and out.stride(-2) == 6528
    (out.float() * cot).sum().backward()
    got = [out.detach().float(), enc.grad.clone(), dec.grad.clone()] + [p.grad.clone() for p in joint.parameters()]
    enc.grad = dec.grad = None
    joint.zero_grad()
    Wf, bf, Wp, bp = joint.forward_layer.weight, joint.forward_layer.bias, joint.project_layer.weight, joint.project_layer.bias
    h = torch.tanh((enc @ Wf[:, :de].t())[:, :, None, :] + (dec @ Wf[:, de:].t())[:, None, :, :] + bf)
    ref = h @ Wp.t() + bp
    (ref * cot).sum().backward()
    want = [ref.detach(), enc.grad, dec.grad] + [p.grad for p in joint.parameters()]
    names = ["logits", "denc", "ddec", "g_wf", "g_bf", "g_wp", "g_bp"]
    for n, g, w in zip(names, got, want):
        e = rel_err(g.cpu().numpy(), w.cpu().numpy())
        print(n, "%.2e" % e)
        assert e < 2e-2, n
    # the bias gradient is a plain column sum of the cotangent (rounded to bf16 by the pipeline): tight check on every one of its 6485 entries
    assert rel_err(got[6].cpu().numpy(), cot.to(torch.bfloat16).float().sum((0, 1, 2)).cpu().numpy()) < 1e-4


# ----------------------------------------------------------------------------------------------- C2 end to end
@pytest.mark.parametrize("mode", ["fp32", "bf16x3"])
def test_c2_full_model_fp32_end_to_end(monkeypatch, mode):
    """(mode bf16x3, round 5: the same data flow with the large dense products in three bf16 terms - the quick parity mode must meet the
    SAME bounds as the exact-f32 one.)
    BASELINE configs[1] model (12 audio / 6 label layers, d_model=512, Di=1024, J=1024, V=4334, K=410/42; 48.2 M parameters, random
    init) at B=2, T=500, U=50 in fp32: logits, loss, input gradient and EVERY parameter gradient against the float64 oracle.

    ReLU decisions: among the 12.6 M FFN hidden units of this run a handful have pre-activations within f32 rounding noise of zero
    (smallest |z| here: 7e-8); whether such a unit counts as active is decided by the last bit, and a single flipped unit changes the
    gradients upstream of it by ~3e-4 relative (1 / sqrt(#units)) - in ANY fp32 implementation, the reference's included.  As with the
    dropout masks, the oracle is therefore fed the ON/OFF decisions the HIP path actually took (read from the saved activations of
    ttmi_ffn_fwd); that they differ from the oracle's own decisions only on near-zero units is asserted."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from tt.model import Transducer
    from ttmi import ops
    from warprnnt_pytorch import RNNTLoss
    monkeypatch.setenv("TTMI_PRECISION", mode)
    cfg = bench.c2_config()
    cfg["dropout"] = 0.0
    torch.manual_seed(1)
    model = Transducer(cfg).cuda().eval()
    assert sum(p.numel() for p in model.parameters()) == 48222862
    _end_to_end_vs_oracle(monkeypatch, model, mode, 2, 500, 50, 4334, [500, 431], [50, 37], 12, 6, None)


def test_c2_model_long_utterance_bf16x3_end_to_end(monkeypatch):
    """the same 12 / 6 model at T = 1000, U = 100 (half of C5's lengths, the size the float64 oracle still finishes in a minute) in the quick
    parity mode: both encoders on the L > K branch of tt/transformer.py:128-132 (tables of 410 / 42 rows against 1000 frames / 101 labels: relative positions
    beyond the table clamp onto row 0 and their gradients fold back onto it), the lattice on the multi-wave alpha / beta kernel (U + 1 = 101 > 64).
    Logits, loss, input gradient and every parameter gradient within 1e-4 of the float64 oracle."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from tt.model import Transducer
    monkeypatch.setenv("TTMI_PRECISION", "bf16x3")
    cfg = bench.c2_config()
    cfg["dropout"] = 0.0
    torch.manual_seed(3)
    model = Transducer(cfg).cuda().eval()
    _end_to_end_vs_oracle(monkeypatch, model, "bf16x3", 1, 1000, 100, 4334, [1000], [100], 12, 6, None)


def _end_to_end_vs_oracle(monkeypatch, model, mode, B, T, U, V, tl, ul, n_enc, n_dec, audio_mask):
    """logits, loss, input gradient and every parameter gradient of `model` (already in the wanted TTMI_PRECISION) against the float64 oracle,
    the oracle fed the ReLU decisions the HIP path took (see test_c2_full_model_fp32_end_to_end)"""
    from ttmi import ops
    from warprnnt_pytorch import RNNTLoss
    gen = torch.Generator().manual_seed(1234)
    inp = torch.randn(B, T, 512, generator=gen)
    tgt = torch.randint(1, V, (B, U), generator=gen)
    tl, ul = np.array(tl, dtype=np.int32), np.array(ul, dtype=np.int32)
    saved = []                                               # (FFN ctx, rows, d, Di) of every layer's forward, in host issue order
    real_layer_fwd = ops.layer_fwd

    def spy(x, x16, pa, pf, *a, **k):
        out = real_layer_fwd(x, x16, pa, pf, *a, **k)
        saved.append((out[4], x.numel() // x.shape[-1], x.shape[-1], pf["ff_w1"].shape[0], tuple(x.shape[:-1])))
        return out

    monkeypatch.setattr(ops, "layer_fwd", spy)
    x = inp.cuda().requires_grad_(True)
    logits = model(x, tgt.cuda())
    loss = RNNTLoss()(logits, tgt.int().cuda(), torch.tensor(tl).cuda(), torch.tensor(ul).cuda())
    loss.backward()
    ops.join_side_streams()
    torch.cuda.synchronize()
    sd64 = {k: (v.detach().cpu().numpy().astype(np.float64) if v.dtype == torch.float32 else v.detach().cpu().numpy())
            for k, v in model.state_dict().items()}
    assert len(saved) == n_enc + n_dec                       # the audio layers (queued first), then the label layers
    for n, (ctx, rows, d, Di, lead) in enumerate(saved):
        off = (rows * d * 4 + 255) // 256 * 256 // 4        # FfnCtx (csrc/layers.hip): h [rows, d] f32, then a1 [rows, Di] f32, 256-byte aligned
        a1 = ctx[off:off + rows * Di].view(*lead, Di)
        key = ("encoder.layers.%d" % n) if n < n_enc else ("decoder.layers.%d" % (n - n_enc))
        sd64[key + ".relu_active"] = (a1 > 0).cpu().numpy()
    own = O.transducer_fwd(inp.numpy().astype(np.float64), tgt.numpy(), {k: v for k, v in sd64.items() if not k.endswith("relu_active")}, audio_mask)[1]
    flips = 0
    for i, (ca, cf) in enumerate(own[0]):                    # the oracle's own decisions differ from the HIP path's only on near-zero units
        diff = cf["relu_on"] != sd64["encoder.layers.%d.relu_active" % i]
        flips += int(diff.sum())
        if diff.any():
            z1 = np.abs((cf["h"] @ sd64["encoder.layers.%d.MultiHeadAttention.pos_ff.CoreNet.0.weight" % i].T +
                         sd64["encoder.layers.%d.MultiHeadAttention.pos_ff.CoreNet.0.bias" % i])[diff])
            # a unit may be decided differently only where its pre-activation (O(1) units) is within the mode's own error of zero: exact f32 1e-5;
            # bf16x3 (three-term products in the attention core too: ~4e-6 per product, 18 layers deep) 5e-5
            assert z1.max() < (1e-5 if mode == "fp32" else 5e-5), (i, z1.max())
    want = O.transducer_loss_and_grads(inp.numpy().astype(np.float64), tgt.numpy(), tl, ul, sd64, audio_mask)
    assert rel_err(logits.detach().cpu().numpy(), want["logits"]) < TOL
    assert abs(float(loss.detach()) - want["loss"]) / want["loss"] < TOL
    assert rel_err(x.grad.cpu().numpy(), want["dinputs"]) < TOL
    worst = ("", 0.0)
    for name, p in model.named_parameters():
        e = rel_err(p.grad.cpu().numpy(), want["grads"][name])
        if e > worst[1]:
            worst = (name, e)
        assert e < TOL, (name, e)
    print("end to end " + mode + (" (streaming mask)" if audio_mask is not None else "") + ": loss rel %.2e, dinputs %.2e, worst gradient %s %.2e, ReLU units decided differently from float64: %d"
          % (abs(float(loss.detach()) - want["loss"]) / want["loss"], rel_err(x.grad.cpu().numpy(), want["dinputs"]), worst[0], worst[1], flips))



@pytest.mark.parametrize("streaming,mode", [("band", "fp32"), ("chunk", "bf16x3")])
def test_c4_streaming_model_end_to_end(monkeypatch, streaming, mode):
    """BASELINE configs[3] model end to end (joint_streaming.yaml: 18 audio / 2 label layers, d_inner 2048, joint 2048, V = 6485; 85.6 M
    parameters) under its streaming masks - band context_mask(left=64, right=0) and the 16-frame chunk mask with 64 left frames - at B = 2,
    T = 300, U = 30 against the float64 oracle fed the equivalent mask tensor: logits, loss, input gradient, every parameter gradient
    (C4 had been covered component-wise only).  The band case runs the exact-f32 mode, the chunk case the bf16x3 mode."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from tt.model import Transducer
    monkeypatch.setenv("TTMI_PRECISION", mode)
    cfg = bench.c4_config(streaming)
    cfg["dropout"] = 0.0
    torch.manual_seed(1)
    model = Transducer(cfg).cuda().eval()
    assert sum(p.numel() for p in model.parameters()) == 85605525
    T = 300
    mask = (O.context_mask(T, 64, 0) if streaming == "band" else O.chunk_mask(T, 16, 64))[:, :, None]      # [qlen, klen, 1], tt/transformer.py:154-159
    _end_to_end_vs_oracle(monkeypatch, model, mode, 2, T, 30, 6485, [T, 251], [30, 22], 18, 2, mask)


def test_c2_full_model_bf16_exp_form_end_to_end(monkeypatch):
    """The mode bench.py's headline is quoted in - bf16 pipeline, Transducer.loss(exp_domain=True) - on the BASELINE configs[1] model at
    full depth (12 / 6 layers, 48.2 M parameters), T=500, against the float64 oracle: loss, input gradient and every parameter gradient.
    B=2, U=63: the exp-domain kernels take lattice-row counts that are multiples of their 64-row K-tile (2 x 500 x 64; B=32 x 500 x 51 of
    the bench is one too, B=2 x 500 x 51 is not and would silently run the plain form).

    What bf16 costs at this depth is MEASURED here and bounded with margin (DESIGN.md §2 quotes the printed figures; measured: loss
    2.8e-5, input gradient 2.5e-2, median parameter gradient 5.7e-3, audio-encoder qkv_net / o_net / CoreNet.3 and the joint <= 5.2e-3,
    worst tensor 7.2e-2 = a label-encoder CoreNet.0.weight - 128 rows per layer, where the ReLU decisions that bf16 rounding flips are a
    visible fraction of the gradient; the encoder states carry a relative error of ~2e-3 after 12 layers of bf16 operands,
    tools/debug/bf16_loss_error.py)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from tt.model import Transducer
    from ttmi import ops
    monkeypatch.setenv("TTMI_PRECISION", "bf16")
    cfg = bench.c2_config()
    cfg["dropout"] = 0.0
    torch.manual_seed(1)
    model = Transducer(cfg).cuda().eval()
    B, T, U, V = 2, 500, 63, 4334
    assert ops.joint_exp_supported(B, T, U + 1, 1024, V, 1)
    gen = torch.Generator().manual_seed(1234)
    inp = torch.randn(B, T, 512, generator=gen)
    tgt = torch.randint(1, V, (B, U), generator=gen)
    tl, ul = np.array([T, 431], dtype=np.int32), np.array([U, 37], dtype=np.int32)
    x = inp.cuda().requires_grad_(True)
    st = model.joint.exp_shift_state(x.device)
    st.set(0.0)
    calls = []
    orig = ops.joint_fwd_exp
    monkeypatch.setattr(ops, "joint_fwd_exp", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    loss = model.loss(x, torch.tensor(tl).cuda(), tgt.cuda(), torch.tensor(ul).cuda(), exp_domain=True, chunk=B)
    loss.backward()
    ops.join_side_streams()
    torch.cuda.synchronize()
    assert calls and int(st.flag) == 0
    sd64 = {k: (v.detach().cpu().numpy().astype(np.float64) if v.dtype == torch.float32 else v.detach().cpu().numpy())
            for k, v in model.state_dict().items()}
    want = O.transducer_loss_and_grads(inp.numpy().astype(np.float64), tgt.numpy(), tl, ul, sd64)
    e_loss = abs(float(loss.detach()) - want["loss"]) / want["loss"]
    e_dx = rel_err(x.grad.cpu().numpy(), want["dinputs"])
    errs = {name: rel_err(p.grad.cpu().numpy(), want["grads"][name]) for name, p in model.named_parameters()}
    big = {n: e for n, e in errs.items() if want["grads"][n].size >= 512 * 512}
    order = sorted(errs.items(), key=lambda kv: -kv[1])
    print("C2 end to end bf16 exp form: loss rel %.2e, dinputs %.2e; parameter gradients rel-L2: median %.2e, worst five %s; worst >= 512x512 matrix %s"
          % (e_loss, e_dx, float(np.median(list(errs.values()))), ", ".join("%s %.2e" % kv for kv in order[:5]),
             "%s %.2e" % max(big.items(), key=lambda kv: kv[1])))
    for grp in ("encoder.layers.0.", "encoder.layers.11.", "decoder.layers.0.", "joint."):
        print("   ", grp, ", ".join("%s %.1e" % (n[len(grp):].replace("MultiHeadAttention.", ""), e) for n, e in errs.items() if n.startswith(grp)))
    assert e_loss < 1e-4
    assert e_dx < 5e-2
    assert max(errs.values()) < 1.2e-1, order[0]
    assert float(np.median(list(errs.values()))) < 1e-2
    for n, e in errs.items():       # the matrices that hold 90 % of the parameters: no ReLU decision in front of their gradient
        if n.startswith("joint.") or (n.startswith("encoder.") and n.endswith(("qkv_net.weight", "o_net.weight", "CoreNet.3.weight"))):
            assert e < 1.5e-2, (n, e)


def test_c2_bf16_loss_error_bound_along_a_training_trajectory(monkeypatch):
    """The bf16 mode's loss error is a property of the STATE, not only of the initial weights: the C2 model, 25 SGD steps of
    bench.py's own loop (B = 32, T = 500, U = 50, dropout on, exp-domain loss, clip + SGD), and at the states after 0 / 10 / 25 steps the
    per-utterance costs of a B = 2 sample in eval mode - the timed form (bf16 encoders, exp-domain joint + loss) - against the float64 oracle on
    the same weights.  The state after 10 steps is the hard one: the cost has just fallen from ~4450 to ~650 and the logits have grown to ~24,
    so the same few hundredths of a nat are a several times larger RELATIVE error.  Measured over the rounds' builds (tools/debug/
    bf16_loss_error.py, profiles/r0*_bf16_loss_error*.log): 2e-6 ... 1.9e-4 per utterance, of which the encoders' bf16 operands alone
    (oracle joint + lattice on the GPU's encoder states) carry 1e-6 ... 1.9e-4 - systematic weight rounding that does not average out over
    an alignment's ~550 emissions.  What is asserted: 3e-4 per utterance of the B = 2 oracle sample and - since round 6 - `north_star`'s 1e-4 on the BATCH MEAN of 32 against
    the fp32 mode at every state visited (0 / 2 / 5 / 10 / 25); per utterance 1e-4 is met by TTMI_PRECISION=fp32 / bf16x3 (test_c2_full_model_fp32_end_to_end: <= 1e-6 / 1.3e-7)
    and, in this mode, with ttmi_set_option(13, 1)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from oracle.rnnt_c import rnnt_loss_c
    from tt.model import Transducer, _JointLossFn
    from ttmi.train import FlatModel, FusedOptimizer, GradSync
    monkeypatch.setenv("TTMI_PRECISION", "bf16")
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    model = Transducer(bench.c2_config()).to(dev).train()
    flat = FlatModel(model)
    flat.enable_grouped_wgrads()
    flat.enable_shadows()
    sync = GradSync(flat)
    opt = FusedOptimizer(flat, kind="sgd", lr=0.00025, momentum=0.9, max_grad_norm=200.0)
    B, T, U, V, d = 32, 500, 50, 4334, 512
    g = torch.Generator(device=dev).manual_seed(1234)
    feats = torch.randn(B, T, 80, device=dev, generator=g)
    proj = torch.randn(80, d, device=dev, generator=torch.Generator(device=dev).manual_seed(7)) / 80 ** 0.5
    targets = torch.randint(1, V, (B, U), device=dev, generator=g)
    ilen = torch.full((B,), T, dtype=torch.int32, device=dev)
    tlen = torch.full((B,), U, dtype=torch.int32, device=dev)
    inputs = (feats.reshape(-1, 80) @ proj).reshape(B, T, d).contiguous()

    def errors():
        """(timed form, encoder states only): worst relative cost error of the B = 2 sample against the float64 oracle"""
        n = 2
        model.eval()
        x, y, il, tl = inputs[:n], targets[:n], ilen[:n], tlen[:n]
        sd64 = {k: (v.detach().cpu().numpy().astype(np.float64) if v.dtype == torch.float32 else v.detach().cpu().numpy())
                for k, v in model.state_dict().items()}
        z64, _ = O.transducer_fwd(x.cpu().numpy().astype(np.float64), y.cpu().numpy(), sd64)
        want = rnnt_loss_c(z64.astype(np.float32), y.cpu().numpy(), il.cpu().numpy(), tl.cpu().numpy(), want_grad=False)[1].astype(np.float64)
        with torch.no_grad():
            enc_s, dec_s = model._encode(x, y)
            j = model.joint
            stt = j.exp_shift_state(dev)
            if not stt.valid:
                stt.set(0.0)
            costs = _JointLossFn.apply(enc_s, dec_s, j.forward_layer.weight, j.forward_layer.bias, j.project_layer.weight, j.project_layer.bias,
                                       y.int().contiguous(), il, tl, 1, n, "none", stt, False)
            torch.cuda.synchronize()
            assert int(stt.flag) == 0
            z, _ = O.joint_fwd(enc_s.double().cpu().numpy(), dec_s.double().cpu().numpy(), sd64)
        c_enc = rnnt_loss_c(z.astype(np.float32), y.cpu().numpy(), il.cpu().numpy(), tl.cpu().numpy(), want_grad=False)[1].astype(np.float64)
        # round 6: the quantity BASELINE names is the LOSS of the batch (train.py:53's mean over B = 32), here against TTMI_PRECISION=fp32 on the same weights (the mode
        # test_c2_full_model_fp32_end_to_end holds within 1e-6 of the oracle)
        with torch.no_grad():
            c16 = model.loss(inputs, ilen, targets, tlen, reduction="none", exp_domain=True, check_lengths=False).double()
            monkeypatch.setenv("TTMI_PRECISION", "fp32")
            c32 = model.loss(inputs, ilen, targets, tlen, reduction="none", check_lengths=False).double()
            monkeypatch.setenv("TTMI_PRECISION", "bf16")
        batch = float((c16.mean() - c32.mean()) / c32.mean())
        model.train()
        return (float(np.max(np.abs(costs.double().cpu().numpy() - want) / want)), float(np.max(np.abs(c_enc - want) / want)), float(want.mean()), batch)

    done, seen = 0, []
    for stop in (0, 2, 5, 10, 25):          # (2 and 5: the steepest states of the descent, where the batch mean read 1e-4 ... 3e-4 before round 6's two measures)
        while done < stop:
            flat.zero_grad()
            sync.start_step()
            loss = model.loss(inputs, ilen, targets, tlen, exp_domain=True)
            loss.backward()
            sync.finish()
            opt.step()
            done += 1
        torch.cuda.synchronize()
        seen.append((stop,) + errors())
    print("bf16 loss error along the trajectory (worst of 2 utterances; timed form / encoder states only / oracle cost / batch mean of 32 vs fp32 mode): "
          + "; ".join("step %d: %.2e / %.2e / %.0f / %+.2e" % s for s in seen))
    for stop, e_timed, e_enc, cost, batch in seen:
        assert e_timed < 3e-4 and e_enc < 3e-4, (stop, e_timed, e_enc)
        # round 6: the quantity BASELINE names - the loss of the batch - within north_star's 1e-4 at EVERY state.  What made it 2e-4 ... 3e-4 at steps 2 - 10 was rounding that is
        # ONE pattern in many lattice rows: the label states' (one label state meets all T frames; their second bf16 term enters the joint's input layer, and their VALUE comes
        # from a gradient-free pass of the label encoder in bf16x3: tt.model.Transducer._label_states) and the weights' (the audio encoder's f32-output GEMMs take their weight's
        # second term: ttmi_set_option(13, 2)).  Measured with both over 56 states of four trajectories: batch mean <= 8.1e-5, worst utterance 1.3e-4
        # (profiles/r06_loss_error_batch_mean_fixed.log; this trajectory: <= 4.9e-5)
        assert abs(batch) < 1e-4, (stop, batch)
    assert seen[-1][3] < 0.25 * seen[0][3]              # the loop did train (4450 -> ~430)


# ----------------------------------------------------------------------------------------------- C5: the TIMED form against the oracle
def _oracle_joint_loss_chunked(enc, dec, sd, labels, B_total, frames=64):
    """oracle.joint_fwd (float64) + the C lattice + oracle.joint_bwd for ONE utterance trimmed to its own lattice, evaluated in chunks of `frames`
    frames so that no [T, U+1, V] float64 array exists (7 GB in f32 at T=2000, U=200 is what the C lattice needs).
    -> cost, d enc [T, d], d dec [U+1, d], parameter gradients (of cost / B_total)"""
    T, U1 = enc.shape[0], dec.shape[0]
    V = sd["joint.project_layer.weight"].shape[0]
    z32 = np.empty((1, T, U1, V), dtype=np.float32)
    hs = []
    for t0 in range(0, T, frames):
        z, cache = O.joint_fwd(enc[None, t0:t0 + frames], dec[None], sd)
        z32[0, t0:t0 + frames] = z[0]
        hs.append(cache["h"])
        del z
    _, costs, dz = rnnt_loss_c(z32, labels[None], np.array([T], np.int32), np.array([U1 - 1], np.int32), reduction="sum")
    del z32
    grads, denc, ddec = {}, np.zeros_like(enc), np.zeros_like(dec)
    for i, t0 in enumerate(range(0, T, frames)):
        g = {}
        de, dd = O.joint_bwd(dz[:, t0:t0 + frames].astype(np.float64) / B_total, dict(enc=enc[None, t0:t0 + frames], dec=dec[None], h=hs[i]), sd, g)
        denc[t0:t0 + frames] = de[0]
        ddec += dd[0]
        for k, v in g.items():
            grads[k] = grads.get(k, 0) + v
        hs[i] = None
    return float(costs[0]), denc, ddec, grads


def test_c5_timed_form_vs_oracle(monkeypatch):
    """BASELINE configs[4] in the form bench.py TIMES (VERDICT r5 missing item 3; so far held to the repo's own fp32 pipeline only): exp-domain joint + loss at
    T = 2000, U = 200 - U + 1 = 201 labels: the multi-wave alpha / beta kernel with its LDS hand-off ring; both encoders on the L > K branch of
    tt/transformer.py:128-132 (tables of 410 / 64 rows) - B = 2 with a ragged second utterance, on the GPU's own bf16 encoder states.  Each utterance is
    held to the oracle on its OWN trimmed lattice (what train.py:32-35's trimming means for a batch): oracle.joint_fwd in float64 + the C lattice +
    oracle.joint_bwd.  Bounds: those of test_exp_domain_vs_oracle (costs 8e-5, loss 5e-5, gradients bf16 class)."""
    from tt.model import Transducer, _JointLossFn
    from tt.utils import AttrDict
    import ttmi.ops as ops
    monkeypatch.setenv("TTMI_PRECISION", "bf16")
    side = dict(n_layer=1, d_model=512, n_head=8, d_head=64, d_inner=256)
    cfg = AttrDict(dict(enc=dict(side, max_input_length=410), dec=dict(side, max_target_length=64),
                        joint=dict(input_size=1024, inner_size=1024), vocab_size=4334, dropout=0.0))
    torch.manual_seed(6)
    model = Transducer(cfg).cuda().train()
    B, T, U, V = 2, 2000, 200, 4334
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn(B, T, 512, device="cuda", generator=g)
    y = torch.randint(1, V, (B, U), device="cuda", generator=g)
    al = torch.tensor([T, 1111], dtype=torch.int32, device="cuda")
    ll = torch.tensor([U, 77], dtype=torch.int32, device="cuda")
    y[0, 9] = 0                                              # a label equal to the blank
    with torch.no_grad():
        enc_s, dec_s = model._encode(x, y)
    j = model.joint
    st = j.exp_shift_state(x.device)
    st.set(0.0)
    calls = []
    orig = ops.joint_bwd_exp
    monkeypatch.setattr(ops, "joint_bwd_exp", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    enc_l, dec_l = enc_s.clone().requires_grad_(True), dec_s.clone().requires_grad_(True)
    model.zero_grad()
    chunk = j.default_loss_chunk(B, T, U + 1, True, 1)
    costs = _JointLossFn.apply(enc_l, dec_l, j.forward_layer.weight, j.forward_layer.bias, j.project_layer.weight, j.project_layer.bias,
                               y.int().contiguous(), al, ll, 1, chunk, "none", st, True).detach().cpu().numpy()
    loss = _JointLossFn.apply(enc_l, dec_l, j.forward_layer.weight, j.forward_layer.bias, j.project_layer.weight, j.project_layer.bias,
                              y.int().contiguous(), al, ll, 1, chunk, "mean", st, True)
    loss.backward()
    torch.cuda.synchronize()
    assert calls and int(st.flag) == 0                       # the exp-domain kernels ran, no range flag
    sd = {"joint." + k: v.detach().double().cpu().numpy() for k, v in j.state_dict().items()}
    enc_h, dec_h = enc_s.double().cpu().numpy(), dec_s.double().cpu().numpy()
    want_costs, denc, ddec, grads = [], np.zeros_like(enc_h), np.zeros_like(dec_h), {}
    for b in range(B):
        Tb, Ub = int(al[b]), int(ll[b])
        c, de, dd, gb = _oracle_joint_loss_chunked(enc_h[b, :Tb], dec_h[b, :Ub + 1], sd, y[b, :Ub].int().cpu().numpy(), B)
        want_costs.append(c)
        denc[b, :Tb], ddec[b, :Ub + 1] = de, dd
        for k, v in gb.items():
            grads[k] = grads.get(k, 0) + v
    want_costs = np.array(want_costs)
    ec = np.abs(costs - want_costs) / want_costs
    el = abs(float(loss.detach()) - want_costs.mean()) / want_costs.mean()
    errs = {"denc": rel_err(enc_l.grad.cpu().numpy(), denc), "ddec": rel_err(dec_l.grad.cpu().numpy(), ddec)}
    for k, p in j.named_parameters():
        errs["g_" + k] = rel_err(p.grad.cpu().numpy(), grads["joint." + k])
    print("C5 timed form vs oracle (T=%d U=%d, ragged %d / %d): costs rel %s, loss rel %.2e, %s"
          % (T, U, int(al[1]), int(ll[1]), ["%.2e" % e for e in ec], el, ", ".join("%s %.2e" % kv for kv in errs.items())))
    assert float(enc_l.grad[1, int(al[1]):].abs().max()) == 0 and float(dec_l.grad[1, int(ll[1]) + 1:].abs().max()) == 0       # nothing outside the ragged lattice
    # measured (profiles/r06_c5_timed_form_oracle_test.log): costs 1.0e-5 / 2.6e-5, loss 2.5e-6, d enc 5.3e-3, d dec 2.1e-2, joint weights 1.8e-2 - 1.9e-2, projection bias
    # 2.7e-5: the sums over 2000 frames of bf16-rounded dH rows are what T = 200 (1.5e-2 in test_exp_domain_vs_oracle) has ten times fewer of
    assert ec.max() < 8e-5 and el < 5e-5
    for k, e in errs.items():
        assert e < 3e-2, (k, e)
    assert errs["g_project_layer.bias"] < 5e-3
