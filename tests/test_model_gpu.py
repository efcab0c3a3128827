"""HIP model path (tt.* over libttmi) vs the golden fixtures produced by the imported reference and vs
the numpy oracle.  fp32 tolerance 1e-4 rel (north_star)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import tt_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


def tiny_config(sd):
    from tt.utils import AttrDict
    k_enc = sd["encoder.layers.0.r_emb"].shape[0]
    k_dec = sd["decoder.layers.0.r_emb"].shape[0]
    side = dict(n_layer=2, d_model=96, n_head=4, d_head=24, d_inner=160)
    return AttrDict(dict(enc=dict(side, max_input_length=k_enc), dec=dict(side, max_target_length=k_dec),
                         joint=dict(input_size=192, inner_size=80), vocab_size=48, dropout=0.0))


def build(sd):
    from tt.model import Transducer
    model = Transducer(tiny_config(sd)).cuda().eval()
    missing = model.load_state_dict({k: torch.tensor(v) for k, v in sd.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return model


@pytest.fixture(params=["tiny_klong", "tiny_kshort"])
def gm(request):
    z, sd = load_golden(request.param)
    return z, sd, build(sd)


def test_encoder_layers_forward(gm):
    z, sd, model = gm
    from ttmi.ops import MaskSpec
    x = torch.tensor(z["inputs"], device="cuda")
    for i, layer in enumerate(model.encoder.layers):
        x = layer.forward_bm(x, MaskSpec(0))
        assert rel_err(x.detach().cpu().numpy(), z["enc_layer%d" % i]) < TOL, i


def test_time_major_layer_contract(gm):
    """the reference's layer-level API is time-major [L,B,d] with a [L,L,1] mask tensor"""
    z, sd, model = gm
    from tt.utils import context_mask
    x = torch.tensor(z["inputs"], device="cuda")
    m = context_mask(x, 10, 2)[:, :, None]
    y = x.transpose(0, 1)
    for layer in model.encoder.layers:
        y = layer(y, m)
    assert rel_err(y.transpose(0, 1).detach().cpu().numpy(), z["mask/band_10_2/enc_out"]) < TOL


def test_label_encoder(gm):
    z, sd, model = gm
    from tt.utils import look_ahead_mask
    tg = torch.nn.functional.pad(torch.tensor(z["targets"], device="cuda"), [1, 0, 0, 0], value=0)
    y = model.decoder(tg, look_ahead_mask(tg)[:, :, None])
    assert rel_err(y.detach().cpu().numpy(), z["dec_masked"]) < TOL
    y = model.decoder(tg)
    assert rel_err(y.detach().cpu().numpy(), z["dec_unmasked"]) < TOL


@pytest.mark.parametrize("tag", ["full", "ragged"])
def test_logits_loss_and_every_gradient(gm, tag):
    z, sd, model = gm
    from warprnnt_pytorch import RNNTLoss
    model.zero_grad()
    inp = torch.tensor(z["inputs"], device="cuda", requires_grad=True)
    tgt = torch.tensor(z["targets"], device="cuda")
    logits = model(inp, tgt)
    assert rel_err(logits.detach().cpu().numpy(), z["logits"]) < TOL
    loss = RNNTLoss(check_lengths=False)(logits, tgt.int(), torch.tensor(z[tag + "/act_lens"], device="cuda"),
                                         torch.tensor(z[tag + "/label_lens"], device="cuda"))
    assert abs(float(loss) - float(z[tag + "/loss"])) / float(z[tag + "/loss"]) < TOL
    loss.backward()
    assert rel_err(inp.grad.cpu().numpy(), z[tag + "/dinputs"]) < TOL
    worst = 0.0
    for name, p in model.named_parameters():
        e = rel_err(p.grad.cpu().numpy(), z["%s/grad/%s" % (tag, name)])
        worst = max(worst, e)
        assert e < TOL, (name, e)
    print("worst grad rel err", worst)


@pytest.mark.parametrize("mname", ["band_10_2", "left_8_0", "chunk_8_16"])
def test_streaming_masks(gm, mname):
    z, sd, model = gm
    from ttmi.ops import MaskSpec
    model.zero_grad()
    inp = torch.tensor(z["inputs"], device="cuda", requires_grad=True)
    mask = torch.tensor(z["mask/%s/mask" % mname], device="cuda").float()[:, :, None]
    y = model.encoder(inp, mask)
    assert rel_err(y.detach().cpu().numpy(), z["mask/%s/enc_out" % mname]) < TOL
    (y * torch.tensor(z["enc_cotangent"], device="cuda")).sum().backward()
    assert rel_err(inp.grad.cpu().numpy(), z["mask/%s/dinputs" % mname]) < TOL
    for name, p in model.encoder.named_parameters():
        key = "mask/%s/grad/encoder.%s" % (mname, name)
        if key in z.files:
            assert rel_err(p.grad.cpu().numpy(), z[key]) < TOL, name
    if mname != "chunk_8_16":     # parametric band mask == the same mask given as a tensor
        left, right = (10, 2) if mname == "band_10_2" else (8, 0)
        y2 = model.encoder(inp.detach(), MaskSpec(2, left=left, right=right))
        assert torch.equal(y2, y.detach())


def test_greedy_decode_identical_tokens(gm):
    z, sd, model = gm
    hyp = model.recognize(torch.tensor(z["inputs"], device="cuda"), torch.tensor(z["greedy/lens"]))
    for b, h in enumerate(hyp):
        assert h == z["greedy/tokens%d" % b].tolist()


def test_bf16_mode_close(gm, monkeypatch):
    """throughput path: bf16 MFMA everywhere, bf16 H / logits / dlogits through the glds GEMMs (J = 80, V = 48 -> pitch 64)"""
    z, sd, model = gm
    from warprnnt_pytorch import RNNTLoss
    monkeypatch.setenv("TTMI_PRECISION", "bf16")
    model.zero_grad()
    tgt = torch.tensor(z["targets"], device="cuda")
    logits = model(torch.tensor(z["inputs"], device="cuda"), tgt)
    assert logits.dtype is torch.bfloat16 and logits.stride(-2) == 64
    e = rel_err(logits.detach().float().cpu().numpy(), z["logits"])
    loss = RNNTLoss()(logits, tgt.int(), torch.tensor(z["full/act_lens"], device="cuda"),
                      torch.tensor(z["full/label_lens"], device="cuda"))
    loss.backward()
    el = abs(float(loss.detach()) - float(z["full/loss"])) / float(z["full/loss"])
    worst = max(rel_err(p.grad.cpu().numpy(), z["full/grad/" + n]) for n, p in model.named_parameters())
    print("bf16: logits rel err %.2e, loss rel err %.2e, worst grad rel err %.2e" % (e, el, worst))
    assert e < 2e-2 and el < 2e-3 and worst < 6e-2


def test_bf16_joint_foreign_gradient(monkeypatch):
    """a gradient that did not come from RNNTLoss (dense f32) is repacked into the zero-padded bf16 layout"""
    from tt.model import JointNet
    monkeypatch.setenv("TTMI_PRECISION", "bf16")
    torch.manual_seed(0)
    j = JointNet(64, 72, 50).cuda()
    enc = torch.randn(2, 9, 32, device="cuda", requires_grad=True)
    dec = torch.randn(2, 4, 32, device="cuda", requires_grad=True)
    out = j(enc, dec)
    w = torch.randn(2, 9, 4, 50, device="cuda")
    (out.float() * w).sum().backward()
    sd = {"joint." + k: v.detach().cpu().numpy().astype(np.float64) for k, v in j.state_dict().items()}
    zz, cache = O.joint_fwd(enc.detach().cpu().numpy().astype(np.float64), dec.detach().cpu().numpy().astype(np.float64), sd)
    grads = {}
    de, dd = O.joint_bwd(w.cpu().numpy().astype(np.float64), cache, sd, grads)
    assert rel_err(out.detach().float().cpu().numpy(), zz) < 2e-2
    assert rel_err(enc.grad.cpu().numpy(), de) < 3e-2 and rel_err(dec.grad.cpu().numpy(), dd) < 3e-2
    assert rel_err(j.project_layer.weight.grad.cpu().numpy(), grads["joint.project_layer.weight"]) < 3e-2
    assert rel_err(j.project_layer.bias.grad.cpu().numpy(), grads["joint.project_layer.bias"]) < 3e-2


def test_vs_oracle_odd_shapes():
    """shapes no fixture covers: B=3, T=70 (> one 64-lane row chunk), U=9, both K branches in one model"""
    from tt.model import Transducer
    from tt.utils import AttrDict
    from warprnnt_pytorch import RNNTLoss
    cfg = AttrDict(dict(enc=dict(n_layer=1, d_model=40, n_head=2, d_head=12, d_inner=52, max_input_length=33),
                        dec=dict(n_layer=1, d_model=40, n_head=2, d_head=12, d_inner=52, max_target_length=64),
                        joint=dict(input_size=80, inner_size=36), vocab_size=21, dropout=0.0))
    torch.manual_seed(3)
    model = Transducer(cfg).cuda().eval()
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    B, T, U = 3, 70, 9
    g = torch.Generator().manual_seed(4)
    inp = torch.randn(B, T, 40, generator=g)
    tgt = torch.randint(1, 21, (B, U), generator=g)
    tl, ul = np.array([70, 55, 70], dtype=np.int32), np.array([9, 9, 4], dtype=np.int32)
    sd64 = {k: v.astype(np.float64) if v.dtype == np.float32 else v for k, v in sd.items()}
    want = O.transducer_loss_and_grads(inp.numpy().astype(np.float64), tgt.numpy(), tl, ul, sd64)
    x = inp.cuda().requires_grad_(True)
    logits = model(x, tgt.cuda())
    loss = RNNTLoss()(logits, tgt.int().cuda(), torch.tensor(tl).cuda(), torch.tensor(ul).cuda())
    loss.backward()
    assert rel_err(logits.detach().cpu().numpy(), want["logits"]) < TOL
    assert abs(float(loss) - want["loss"]) / want["loss"] < TOL
    assert rel_err(x.grad.cpu().numpy(), want["dinputs"]) < TOL
    for name, p in model.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), want["grads"][name]) < TOL, name


def test_label_encoder_on_side_stream_same_results(gm):
    """config.overlap_label_encoder runs the label encoder on a side stream: logits and gradients are unchanged"""
    z, sd, model = gm
    from ttmi import ops
    from warprnnt_pytorch import RNNTLoss
    tgt = torch.tensor(z["targets"], device="cuda")
    res = []
    for overlap in (False, True):
        model.config["overlap_label_encoder"] = overlap
        model.zero_grad()
        inp = torch.tensor(z["inputs"], device="cuda", requires_grad=True)
        logits = model(inp, tgt)
        loss = RNNTLoss(check_lengths=False)(logits, tgt.int(), torch.tensor(z["full/act_lens"], device="cuda"),
                                             torch.tensor(z["full/label_lens"], device="cuda"))
        loss.backward()
        ops.join_side_streams()
        torch.cuda.synchronize()
        res.append((logits.detach().clone(), inp.grad.clone(), {n: p.grad.clone() for n, p in model.named_parameters()}))
    model.config["overlap_label_encoder"] = False
    assert torch.equal(res[0][0], res[1][0])
    assert rel_err(res[1][1].cpu().numpy(), res[0][1].cpu().numpy()) < 1e-6
    for n in res[0][2]:
        assert rel_err(res[1][2][n].cpu().numpy(), res[0][2][n].cpu().numpy()) < 1e-5, n


def test_blocked_greedy_decode_matches_frame_by_frame():
    """a model whose blank logit is boosted emits sparsely: the blocked on-device scan must reproduce the reference's
    frame-by-frame loop (restated here with torch ops on the HIP model's own joint/decoder outputs) for every block size"""
    from tt.model import Transducer
    from tt.utils import AttrDict
    side = dict(n_layer=1, d_model=64, n_head=2, d_head=32, d_inner=96)
    cfg = AttrDict(dict(enc=dict(side, max_input_length=16), dec=dict(side, max_target_length=8),
                        joint=dict(input_size=128, inner_size=48), vocab_size=29, dropout=0.0))
    torch.manual_seed(5)
    model = Transducer(cfg).cuda().eval()
    with torch.no_grad():
        model.joint.project_layer.bias[0] += 1.2             # favour blank so that only some frames emit
    x = torch.randn(2, 90, 64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(6))
    lens = [90, 61]
    with torch.no_grad():
        enc = model.encoder(x)
        want = []
        for b in range(2):
            toks = [0]
            dstate = model.decoder(torch.tensor([toks], device="cuda"))[:, -1, :]
            for t in range(lens[b]):
                pred = int(torch.argmax(model.joint(enc[b, t].view(-1), dstate.view(-1))).item())
                if pred != 0:
                    toks.append(pred)
                    dstate = model.decoder(torch.tensor([toks], device="cuda"))[:, -1, :]
            want.append(toks[1:])
        assert 0 < len(want[0]) < 90                         # genuinely sparse emissions
        for block in (1, 7, 64, 500):
            got = [model.decode(enc[b], lens[b], block=block) for b in range(2)]
            assert got == want, block
        assert model.recognize(x, torch.tensor(lens)) == want


def test_decode_graphs_match_eager_and_follow_weight_updates():
    """decode replays one captured graph per history length for the label-encoder re-runs: same tokens as the eager launches
    (config.decode_graphs = False), on first use, on reuse for the next utterance, and after an in-place weight update (graphs hold
    pointers, not values)"""
    from tt.model import Transducer
    from tt.utils import AttrDict
    side = dict(n_layer=2, d_model=64, n_head=2, d_head=32, d_inner=96)
    cfg = AttrDict(dict(enc=dict(side, max_input_length=16), dec=dict(side, max_target_length=8),
                        joint=dict(input_size=128, inner_size=48), vocab_size=29, dropout=0.0))
    torch.manual_seed(9)
    model = Transducer(cfg).cuda().eval()
    model.config["decode_batch_graphs"] = True              # batches take the graph path too (the default again since round 5: tt/model.py decode_batch)
    with torch.no_grad():
        model.joint.project_layer.bias[0] += 0.4             # some blank frames between the emissions
    x = torch.randn(3, 70, 64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(10))
    lens = torch.tensor([70, 55, 70])

    def both():
        model.config["decode_graphs"] = False
        eager = model.recognize(x, lens)
        model.config["decode_graphs"] = True
        return eager, model.recognize(x, lens)

    eager, graphed = both()
    assert len(eager[0]) > 8 and graphed == eager          # histories longer than the label encoder's table (K = 8) included
    sets = model.__dict__["_decode_graphs"]                              # one graph set per batch size (3 here: decode_batch)
    assert sets and all(g.graphs for g in sets.values())                # graphs were captured and are reused below
    assert model.recognize(x, lens) == eager
    with torch.no_grad():
        for p_ in model.decoder.parameters():
            p_.mul_(1.25)
    eager2, graphed2 = both()
    assert graphed2 == eager2 and eager2 != eager


def test_full_size_greedy_decode_tokens_vs_oracle():
    """BASELINE configs[1] model (12 / 6 layers, d_model 512, V = 4334, random init under seed 1) with a blank bias that lets about
    15 % of the frames emit: `recognize` gives the oracle's frame-by-frame token lists (tt/model.py:70-108) exactly, including
    histories longer than the label encoder's table (L > K = 42 needs > 41 symbols: the second utterance runs 400 frames)"""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import bench_decode
    prev = os.environ.get("TTMI_PRECISION")
    try:
        out, model, inputs, lens, _ = bench_decode.run(utts=2, T=400, emit_rate=0.15, precision="fp32")
    finally:
        if prev is None:
            os.environ.pop("TTMI_PRECISION", None)
        else:
            os.environ["TTMI_PRECISION"] = prev
    lens = [150, 400]
    with torch.no_grad():
        hyp = model.recognize(inputs, torch.tensor(lens))
    sd = {k: v.detach().float().cpu().numpy() for k, v in model.state_dict().items()}
    ref = O.recognize(inputs.float().cpu().numpy(), lens, sd)
    assert len(ref[1]) > 42 and len(ref[0]) < 150
    assert hyp == ref


@pytest.mark.parametrize("J,V,B,T,U1", [(512, 200, 2, 37, 9), (1024, 130, 2, 37, 9), (80, 48, 2, 37, 9), (1024, 1500, 4, 400, 21)])
def test_bf16_joint_wide_inner_dims_vs_torch(J, V, B, T, U1, monkeypatch):
    """the 8-column tanh / (t,u)-reduction kernels (J = 512, 1024) and the 4-column ones (other J) of the bf16 joint: logits
    and every gradient against an fp32 torch evaluation of JointNet.forward (tt/model.py:20-39 of the reference)"""
    from tt.model import JointNet
    monkeypatch.setenv("TTMI_PRECISION", "bf16")
    torch.manual_seed(J)
    de = 64                 # the last case (33600 lattice rows) runs on the persistent kernels: v8 forward / dgrad with the tanh' epilogue,
    joint = JointNet(2 * de, J, V).cuda()     # TN v8 wgrad with its all-ones column sums and the M % 256 strip
    enc = torch.randn(B, T, de, device="cuda", requires_grad=True)
    dec = torch.randn(B, U1, de, device="cuda", requires_grad=True)
    cot = torch.randn(B, T, U1, V, device="cuda")
    out = joint(enc, dec)
    (out.float() * cot).sum().backward()
    got = [out.detach().float(), enc.grad.clone(), dec.grad.clone()] + [p.grad.clone() for p in joint.parameters()]
    enc.grad = dec.grad = None
    joint.zero_grad()
    Wf, bf, Wp, bp = joint.forward_layer.weight, joint.forward_layer.bias, joint.project_layer.weight, joint.project_layer.bias
    cat = torch.cat([enc[:, :, None, :].expand(B, T, U1, de), dec[:, None, :, :].expand(B, T, U1, de)], dim=-1)
    ref = torch.tanh(cat @ Wf.t() + bf) @ Wp.t() + bp
    (ref * cot).sum().backward()
    want = [ref.detach(), enc.grad, dec.grad] + [p.grad for p in joint.parameters()]
    for g, w in zip(got, want):
        assert rel_err(g.cpu().numpy(), w.cpu().numpy()) < 2e-2


@pytest.mark.parametrize("name", ["tiny_klong", "tiny_kshort"])
@pytest.mark.parametrize("graphs", [True, False])
def test_sparse_greedy_decode_identical_tokens(name, graphs):
    """the reference's own recognize() with the blank logit raised (tests/golden/greedy_sparse.npz): most frames are blank, histories outgrow
    the label encoder's table; blocked on-device scan + captured label-encoder graphs give the identical token lists"""
    import os
    from conftest import GOLDEN
    z, sd = load_golden(name)
    g = np.load(os.path.join(GOLDEN, "greedy_sparse.npz"))
    model = build(sd)
    with torch.no_grad():
        model.joint.project_layer.bias[0] += float(g[name + "/blank_bias"])
    model.config["decode_graphs"] = graphs
    hyp = model.recognize(torch.tensor(g[name + "/inputs"], device="cuda"), torch.tensor(g[name + "/lens"]))
    for b, h in enumerate(hyp):
        assert h == g["%s/tokens%d" % (name, b)].tolist(), b


@pytest.mark.parametrize("block", [64, 5, 0])
@pytest.mark.parametrize("name", ["tiny_klong", "tiny_kshort"])
def test_beam_search_identical_tokens(name, block, monkeypatch):
    """Transducer.recognize_beam_search (the reference's beam search, tt/model.py:110-198, quirks included) on the HIP model: the token lists
    of the reference run - with the device-side form of round 5 (blocks of frames scored against the lead hypothesis' label state, one
    label-encoder call and one host read per expansion; block = 64 and 5) and with the per-frame restatement of round 2 (block = 0)"""
    from tt.model import Transducer
    real = Transducer.beam_search
    monkeypatch.setattr(Transducer, "beam_search", lambda self, e, l, beam_width=5: real(self, e, l, beam_width=beam_width, block=block))
    import os
    from conftest import GOLDEN
    z, sd = load_golden(name)
    g = np.load(os.path.join(GOLDEN, "greedy_sparse.npz"))
    model = build(sd)
    with torch.no_grad():
        model.joint.project_layer.bias[0] += float(g[name + "/blank_bias"])
    hyp = model.recognize_beam_search(torch.tensor(g[name + "/inputs"][:, :48], device="cuda"), torch.tensor(g[name + "/beam_lens"]))
    for b, h in enumerate(hyp):
        assert h == g["%s/beam_tokens%d" % (name, b)].tolist(), b


@pytest.mark.parametrize("graphs", [True, False])
def test_batched_greedy_decode_equals_one_utterance_at_a_time(graphs):
    """Transducer.decode_batch (round 4: the whole batch in lockstep over symbol steps - one joint call per scanned block, one label-encoder
    call per step, positions / histories / flags on the device) returns what decode() returns utterance by utterance: ragged lengths, an
    utterance that never emits, one of length 1, histories longer than the label encoder's table, every block size"""
    from tt.model import Transducer
    from tt.utils import AttrDict
    side = dict(n_layer=2, d_model=64, n_head=2, d_head=32, d_inner=96)
    cfg = AttrDict(dict(enc=dict(side, max_input_length=16), dec=dict(side, max_target_length=8),
                        joint=dict(input_size=128, inner_size=48), vocab_size=29, dropout=0.0, decode_graphs=graphs))
    torch.manual_seed(9)
    model = Transducer(cfg).cuda().eval()
    with torch.no_grad():
        model.joint.project_layer.bias[0] += 0.9
    x = torch.randn(6, 120, 64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    x[3] = 0.0
    lens = [120, 77, 1, 120, 33, 100]
    with torch.no_grad():
        enc = model.encoder(x)
        enc[3] = enc[3] * 0.0 - 50.0 * model.joint.forward_layer.weight[:, :64].sum(0).sign()     # utterance 3: blank everywhere
        want = [model.decode(enc[b], lens[b]) for b in range(6)]
        assert max(len(w) for w in want) > 8 and min(len(w) for w in want) <= 1
        for block in (1, 5, 64, 300):
            assert model.decode_batch(enc, lens, block=block) == want, block
        assert model.decode_batch(enc[:1], lens[:1]) == want[:1]
