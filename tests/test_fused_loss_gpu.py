"""Fused joint + RNN-T loss, `Transducer.loss(inputs, inputs_length, targets, targets_length)`: the same numbers as
`model(inputs, targets)` + `RNNTLoss()` (train.py:51-53) without ever holding the [B, T, U+1, V] logits - against the reference-run
fixtures, the oracle, and the two-call form."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from test_model_gpu import build

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(params=["tiny_klong", "tiny_kshort"])
def gm(request):
    z, sd = load_golden(request.param)
    return z, sd, build(sd)


@pytest.mark.parametrize("tag", ["full", "ragged"])
@pytest.mark.parametrize("chunk", [1, 2])
def test_loss_and_every_gradient_vs_reference_fixtures(gm, tag, chunk):
    z, sd, model = gm
    model.zero_grad()
    inp = torch.tensor(z["inputs"], device="cuda", requires_grad=True)
    tgt = torch.tensor(z["targets"], device="cuda")
    loss = model.loss(inp, torch.tensor(z[tag + "/act_lens"], device="cuda"), tgt, torch.tensor(z[tag + "/label_lens"], device="cuda"),
                      chunk=chunk, check_lengths=False)
    assert loss.shape == (1,) and abs(float(loss) - float(z[tag + "/loss"])) / float(z[tag + "/loss"]) < TOL
    loss.backward()
    assert rel_err(inp.grad.cpu().numpy(), z[tag + "/dinputs"]) < TOL
    for name, p in model.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), z["%s/grad/%s" % (tag, name)]) < TOL, name


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_same_numbers_as_the_two_call_form(gm, prec, monkeypatch):
    """identical kernels on identical inputs: the loss agrees to the last bit, the gradients up to the order of f32 atomic sums; an
    upstream factor (loss * 3) scales everything"""
    z, sd, model = gm
    from warprnnt_pytorch import RNNTLoss
    monkeypatch.setenv("TTMI_PRECISION", prec)
    tgt = torch.tensor(z["targets"], device="cuda")
    al, ll = torch.tensor(z["ragged/act_lens"], device="cuda"), torch.tensor(z["ragged/label_lens"], device="cuda")
    res = []
    for fused in (False, True):
        model.zero_grad()
        inp = torch.tensor(z["inputs"], device="cuda", requires_grad=True)
        if fused:
            loss = model.loss(inp, al, tgt, ll, chunk=1, check_lengths=False)
        else:
            loss = RNNTLoss(check_lengths=False)(model(inp, tgt), tgt.int(), al, ll)
        (loss * 3.0).backward()
        res.append((loss.detach().clone(), inp.grad.clone(), {n: p.grad.clone() for n, p in model.named_parameters()}))
    assert torch.equal(res[0][0], res[1][0])
    tol = 1e-5 if prec == "fp32" else 1e-2       # bf16: one-utterance chunks run other GEMM variants (bf16 partial products round differently)
    assert rel_err(res[1][1].cpu().numpy(), res[0][1].cpu().numpy()) < tol
    for n in res[0][2]:
        assert rel_err(res[1][2][n].cpu().numpy(), res[0][2][n].cpu().numpy()) < tol, n
    none = model.loss(torch.tensor(z["inputs"], device="cuda"), al, tgt, ll, reduction="none", check_lengths=False)
    assert none.shape == (2,) and abs(float(none.sum() / 2) - float(res[0][0])) < 1e-4 * float(res[0][0])


def test_length_contract(gm):
    z, sd, model = gm
    tgt = torch.tensor(z["targets"], device="cuda")
    inp = torch.tensor(z["inputs"], device="cuda")
    al, ll = torch.tensor(z["full/act_lens"], device="cuda"), torch.tensor(z["full/label_lens"], device="cuda")
    model.loss(inp, al, tgt, ll)                                            # full lengths pass warp-transducer's checks
    with pytest.raises(ValueError):
        model.loss(inp, al - 1, tgt, ll)                                    # T != max(act_lens)
    with pytest.raises(ValueError):
        model.loss(inp, al, tgt, ll - 1)                                    # U + 1 != max(label_lens) + 1
    with pytest.raises(ValueError):
        model.loss(inp, al[:1], tgt, ll)


def test_flat_model_gradients_and_memory(monkeypatch):
    """bf16 pipeline at a size where the persistent GEMMs run (B=8, T=200, U=20, J=1024, V=4334): gradients land in FlatModel's flat
    buffer and match the two-call form; the peak memory of the step drops by about the size of the logits + their gradient"""
    from tt.model import Transducer
    from tt.utils import AttrDict
    from ttmi.train import FlatModel
    from warprnnt_pytorch import RNNTLoss
    monkeypatch.setenv("TTMI_PRECISION", "bf16")
    side = dict(n_layer=1, d_model=512, n_head=8, d_head=64, d_inner=256)
    cfg = AttrDict(dict(enc=dict(side, max_input_length=64), dec=dict(side, max_target_length=16),
                        joint=dict(input_size=1024, inner_size=1024), vocab_size=4334, dropout=0.0))
    torch.manual_seed(2)
    model = Transducer(cfg).cuda().train()
    flat = FlatModel(model)
    B, T, U = 8, 200, 20
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(B, T, 512, device="cuda", generator=g)
    y = torch.randint(1, 4334, (B, U), device="cuda", generator=g)
    al = torch.full((B,), T, dtype=torch.int32, device="cuda")
    ll = torch.full((B,), U, dtype=torch.int32, device="cuda")
    al[2], ll[2] = 150, 11
    out = []
    for fused in (False, True):
        flat.zero_grad()
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        if fused:
            loss = model.loss(x, al, y, ll, chunk=2, check_lengths=False)
        else:
            loss = RNNTLoss(check_lengths=False)(model(x, y), y.int(), al, ll)
        loss.backward()
        torch.cuda.synchronize()
        out.append((float(loss.detach()), flat.grad.clone(), torch.cuda.max_memory_allocated() - base))
        del loss
    assert out[0][0] == out[1][0]
    assert rel_err(out[1][1].cpu().numpy(), out[0][1].cpu().numpy()) < 3e-3
    logits_bytes = B * T * (U + 1) * 4352 * 2
    print("peak above baseline: two-call %.0f MB, fused %.0f MB (logits %.0f MB)" % (out[0][2] / 2 ** 20, out[1][2] / 2 ** 20, logits_bytes / 2 ** 20))
    assert out[0][2] - out[1][2] > 1.2 * logits_bytes        # logits + gradient gone, a chunk's worth (1/4) of one buffer remains


def _training_sized(monkeypatch, prec, J=1024, V=4334):
    from tt.model import Transducer
    from tt.utils import AttrDict
    monkeypatch.setenv("TTMI_PRECISION", prec)
    side = dict(n_layer=1, d_model=512, n_head=8, d_head=64, d_inner=256)
    cfg = AttrDict(dict(enc=dict(side, max_input_length=64), dec=dict(side, max_target_length=16),
                        joint=dict(input_size=1024, inner_size=J), vocab_size=V, dropout=0.0))
    torch.manual_seed(2)
    model = Transducer(cfg).cuda().train()
    B, T, U = 8, 200, 20
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(B, T, 512, device="cuda", generator=g)
    y = torch.randint(1, V, (B, U), device="cuda", generator=g)
    al = torch.full((B,), T, dtype=torch.int32, device="cuda")
    ll = torch.full((B,), U, dtype=torch.int32, device="cuda")
    al[2], ll[2] = 150, 11
    y[5, 3] = 0                                              # a label equal to the blank symbol: both emission terms land on one entry
    return model, x, y, al, ll


def _run(model, x, y, al, ll, **kw):
    model.zero_grad()
    xi = x.clone().requires_grad_(True)
    loss = model.loss(xi, al, y, ll, check_lengths=False, **kw)
    loss.backward()
    return float(loss.detach()), torch.cat([xi.grad.reshape(-1)] + [p.grad.reshape(-1) for p in model.parameters()]).cpu().numpy(), \
        {n: p.grad.cpu().numpy() for n, p in model.named_parameters()}


@pytest.mark.parametrize("J,V", [(1024, 4334), (2048, 6485)])          # C2 and C4 (joint_streaming.yaml) joint dimensions
def test_exp_domain_fast_path(monkeypatch, J, V):
    """B=8, T=200, U=20 in one chunk (33600 lattice rows: the persistent kernels' sizes).  The exp-domain form must be as
    close to the fp32 pipeline as the plain bf16 form is: both are measured against fp32 and printed."""
    import ttmi.ops as ops
    model, x, y, al, ll = _training_sized(monkeypatch, "fp32", J, V)
    ref = _run(model, x, y, al, ll, chunk=8)
    monkeypatch.setenv("TTMI_PRECISION", "bf16")
    plain = _run(model, x, y, al, ll, chunk=8)
    calls = []
    orig = ops.joint_fwd_exp
    monkeypatch.setattr(ops, "joint_fwd_exp", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    fast = _run(model, x, y, al, ll, chunk=8, exp_domain=True)
    assert calls, "the exp-domain kernels did not run"
    e_plain, e_fast = rel_err(plain[1], ref[1]), rel_err(fast[1], ref[1])
    print("loss fp32 %.4f  bf16 %.4f  exp-domain %.4f;  gradient error vs fp32: bf16 %.2e, exp-domain %.2e" % (ref[0], plain[0], fast[0], e_plain, e_fast))
    assert abs(fast[0] - ref[0]) < 2e-4 * ref[0] and abs(plain[0] - ref[0]) < 2e-4 * ref[0]
    assert e_fast < max(1.5 * e_plain, 5e-3)
    for n in ref[2]:
        assert rel_err(fast[2][n], ref[2][n]) < max(2 * rel_err(plain[2][n], ref[2][n]), 5e-3), n
    # small chunks are outside the fast path: silently the plain form, same numbers as without the switch
    small = _run(model, x, y, al, ll, chunk=2, exp_domain=True)
    base = _run(model, x, y, al, ll, chunk=2)
    assert small[0] == base[0]


def test_exp_domain_shift(monkeypatch):
    """the subtracted shift cancels: any value gives the same loss; large logits raise the value gathered for the next step"""
    import ttmi.ops as ops
    from tt.model import _JointLossFn
    model, x, y, al, ll = _training_sized(monkeypatch, "bf16")
    _JointLossFn._shift.clear()
    a = _run(model, x, y, al, ll, chunk=8, exp_domain=True)
    cur, nxt = _JointLossFn._shift[x.device]
    assert float(cur) == 0.0 and float(nxt) == 0.0            # logits of a fresh model are far below the margin
    cur.fill_(7.5)
    b = _run(model, x, y, al, ll, chunk=8, exp_domain=True)
    assert abs(a[0] - b[0]) < 1e-4 * a[0] and rel_err(b[1], a[1]) < 5e-3
    with torch.no_grad():
        model.joint.project_layer.bias[17] += 70.0            # one huge logit in every row: log-sum-exp ~ 70
    _JointLossFn._shift.clear()
    _run(model, x, y, al, ll, chunk=8, exp_domain=True)
    cur, nxt = _JointLossFn._shift[x.device]
    assert 25.0 < float(cur) < 35.0 and float(nxt) == 0.0     # ~70 - 40, handed to the next step
    c = _run(model, x, y, al, ll, chunk=8, exp_domain=True)
    d = _run(model, x, y, al, ll, chunk=8)
    assert np.isfinite(c[0]) and abs(c[0] - d[0]) < 2e-4 * d[0]


def test_exp_domain_long_label_sequences(monkeypatch):
    """U + 1 = 201 labels per lattice column block (C5's label length: the 4-slot variant of the alpha / beta kernel) at T = 320, B = 4
    (257 280 rows in one chunk), ragged lengths: exp-domain form against the plain bf16 form and the fp32 pipeline"""
    from tt.model import Transducer
    from tt.utils import AttrDict
    monkeypatch.setenv("TTMI_PRECISION", "fp32")
    side = dict(n_layer=1, d_model=512, n_head=8, d_head=64, d_inner=256)
    cfg = AttrDict(dict(enc=dict(side, max_input_length=64), dec=dict(side, max_target_length=32),
                        joint=dict(input_size=1024, inner_size=1024), vocab_size=4334, dropout=0.0))
    torch.manual_seed(4)
    model = Transducer(cfg).cuda().train()
    B, T, U = 4, 320, 200
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(B, T, 512, device="cuda", generator=g)
    y = torch.randint(1, 4334, (B, U), device="cuda", generator=g)
    al = torch.tensor([T, T - 37, T, T - 5], dtype=torch.int32, device="cuda")
    ll = torch.tensor([U, U - 60, U - 1, U], dtype=torch.int32, device="cuda")
    ref = _run(model, x, y, al, ll, chunk=B)
    monkeypatch.setenv("TTMI_PRECISION", "bf16")
    plain = _run(model, x, y, al, ll, chunk=B)
    fast = _run(model, x, y, al, ll, chunk=B, exp_domain=True)
    e_plain, e_fast = rel_err(plain[1], ref[1]), rel_err(fast[1], ref[1])
    print("U=200: loss fp32 %.4f  bf16 %.4f  exp-domain %.4f;  gradient error vs fp32: bf16 %.2e, exp-domain %.2e" % (ref[0], plain[0], fast[0], e_plain, e_fast))
    assert abs(fast[0] - ref[0]) < 2e-4 * ref[0] and abs(plain[0] - ref[0]) < 2e-4 * ref[0]
    assert fast[0] != plain[0]                               # (the exp-domain kernels ran: the two forms round differently)
    assert e_fast < max(1.5 * e_plain, 5e-3)
