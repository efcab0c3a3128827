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
    monkeypatch.setenv("TTMI_DEFERRED_LOGITS", "0")          # the two-call form with its logits MATERIALISED is the yardstick here
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
    monkeypatch.setenv("TTMI_DEFERRED_LOGITS", "0")          # (materialised logits: what the fused form's memory is compared with)
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
    seed = _run(model, x, y, al, ll, chunk=8, exp_domain=True)
    assert not calls and seed[0] == plain[0]                 # first use: the plain fused form, which seeds the shift on the device
    fast = _run(model, x, y, al, ll, chunk=8, exp_domain=True)
    assert calls, "the exp-domain kernels did not run"
    e_plain, e_fast = rel_err(plain[1], ref[1]), rel_err(fast[1], ref[1])
    print("loss fp32 %.4f  bf16 %.4f  exp-domain %.4f;  gradient error vs fp32: bf16 %.2e, exp-domain %.2e" % (ref[0], plain[0], fast[0], e_plain, e_fast))
    assert abs(fast[0] - ref[0]) < 1e-4 * ref[0] and abs(plain[0] - ref[0]) < 2e-4 * ref[0]
    assert e_fast < max(1.5 * e_plain, 5e-3)
    for n in ref[2]:
        assert rel_err(fast[2][n], ref[2][n]) < max(2 * rel_err(plain[2][n], ref[2][n]), 5e-3), n
    # small chunks are outside the fast path: silently the plain form, same numbers as without the switch
    small = _run(model, x, y, al, ll, chunk=2, exp_domain=True)
    base = _run(model, x, y, al, ll, chunk=2)
    assert small[0] == base[0]


@pytest.mark.parametrize("J,V", [(1024, 4334), (2048, 6485)])
def test_exp_domain_vs_oracle(monkeypatch, J, V):
    """the form bench.py times, against the ORACLE (not the repo's own fp32 pipeline): joint + loss at the C2 / C4 joint dimensions on the
    encoder states of a bf16 model, B=8, T=200, U=20 (one ragged utterance, one label equal to the blank).  Per-utterance costs, d enc,
    d dec and every joint parameter gradient against oracle.joint_fwd (float64) + the C lattice + oracle.joint_bwd (float64) fed the same
    encoder states.  Bounds: costs 5e-5 rel (f32 emission logits: what is left is the bf16 rounding of H and Wp inside the projection),
    gradients bf16 class."""
    from oracle import tt_oracle as O
    from oracle.rnnt_c import rnnt_loss_c
    from tt.model import _JointLossFn
    model, x, y, al, ll = _training_sized(monkeypatch, "bf16", J, V)
    with torch.no_grad():
        enc_s, dec_s = model._encode(x, y)
    B, T, U1 = enc_s.shape[0], enc_s.shape[1], dec_s.shape[1]
    j = model.joint
    st = j.exp_shift_state(x.device)
    st.set(0.0)                                              # fresh model: logits of order 1
    enc_l, dec_l = enc_s.clone().requires_grad_(True), dec_s.clone().requires_grad_(True)
    model.zero_grad()
    costs = _JointLossFn.apply(enc_l, dec_l, j.forward_layer.weight, j.forward_layer.bias, j.project_layer.weight, j.project_layer.bias,
                               y.int().contiguous(), al, ll, 1, B, "none", st, True)
    assert costs.shape == (B,)
    with pytest.raises(NotImplementedError):
        costs.sum().backward()                               # per-utterance upstream gradients are not part of the fused form
    model.zero_grad()
    loss = _JointLossFn.apply(enc_l, dec_l, j.forward_layer.weight, j.forward_layer.bias, j.project_layer.weight, j.project_layer.bias,
                              y.int().contiguous(), al, ll, 1, B, "mean", st, True)
    loss.backward()
    torch.cuda.synchronize()
    assert int(st.flag) == 0
    sd = {"joint." + k: v.detach().double().cpu().numpy() for k, v in j.state_dict().items()}
    z, cache = O.joint_fwd(enc_s.double().cpu().numpy(), dec_s.double().cpu().numpy(), sd)
    want_loss, want_costs, dz = rnnt_loss_c(z.astype(np.float32), y.int().cpu().numpy(), al.cpu().numpy(), ll.cpu().numpy())
    grads = {}
    denc, ddec = O.joint_bwd(dz.astype(np.float64), cache, sd, grads)
    ec = np.abs(costs.detach().cpu().numpy() - want_costs) / want_costs
    el = abs(float(loss.detach()) - float(want_loss)) / float(want_loss)
    errs = {"denc": rel_err(enc_l.grad.cpu().numpy(), denc), "ddec": rel_err(dec_l.grad.cpu().numpy(), ddec)}
    for k, p in j.named_parameters():
        errs["g_" + k] = rel_err(p.grad.cpu().numpy(), grads["joint." + k])
    print("exp-domain form vs oracle (J=%d V=%d): costs rel max %.2e, loss rel %.2e, %s" % (J, V, ec.max(), el, ", ".join("%s %.2e" % kv for kv in errs.items())))
    assert ec.max() < 8e-5 and el < 5e-5          # (measured 4.8e-5 - 5.1e-5 / 5e-6 - 1.5e-5 over this round's builds: random weights, no structure to average)
    for k, e in errs.items():
        assert e < 1.5e-2, (k, e)
    # the bias gradient of the projection is P^T s: a plain weighted column sum, tighter than the products that round H
    assert errs["g_project_layer.bias"] < 5e-3


def test_exp_domain_shift_protocol(monkeypatch):
    """range control of the exp-domain form (tt.model._ExpShift): state per module, seeded on the device by a plain-form step, handed from
    step to step, invalidated by load_state_dict; the subtracted shift cancels; a shift that no longer fits the logits gives NaN (never a
    finite wrong number), raises the device flag and the next step recovers in the plain form."""
    model, x, y, al, ll = _training_sized(monkeypatch, "bf16")
    st = model.joint.exp_shift_state(x.device)
    assert not st.valid
    plain = _run(model, x, y, al, ll, chunk=8)
    a0 = _run(model, x, y, al, ll, chunk=8, exp_domain=True)     # seeding step = the plain form
    assert st.valid and a0[0] == plain[0] and float(st.cur) == 0.0 and float(st.nxt) == 0.0   # logits of a fresh model are far below the margin
    a = _run(model, x, y, al, ll, chunk=8, exp_domain=True)
    assert abs(a[0] - plain[0]) < 1e-4 * plain[0]
    st.cur.fill_(7.5)
    b = _run(model, x, y, al, ll, chunk=8, exp_domain=True)
    assert abs(a[0] - b[0]) < 2e-5 * a[0] and rel_err(b[1], a[1]) < 5e-3
    assert float(st.cur) == 0.0                                   # the hand-over: this step's rows asked for no shift
    # new weights through load_state_dict: one huge logit in every row (log-sum-exp ~ 70)
    sd = {k: v.clone() for k, v in model.joint.state_dict().items()}
    sd["project_layer.bias"][17] += 70.0
    model.joint.load_state_dict(sd)
    assert not st.valid
    d = _run(model, x, y, al, ll, chunk=8)
    c0 = _run(model, x, y, al, ll, chunk=8, exp_domain=True)      # re-seeds
    assert c0[0] == d[0] and st.valid and 25.0 < float(st.cur) < 35.0 and float(st.nxt) == 0.0       # ~70 - 40
    c = _run(model, x, y, al, ll, chunk=8, exp_domain=True)
    # (gradients: the PLAIN bf16 form is the coarse one here - its logits of ~70 are stored with a bf16 spacing of 0.5)
    assert np.isfinite(c[0]) and abs(c[0] - d[0]) < 1e-4 * d[0] and rel_err(c[1], d[1]) < 1e-1
    # a second model never inherits the first one's state
    other, *_ = _training_sized(monkeypatch, "bf16")
    so = other.joint.exp_shift_state(x.device)
    assert so is not st and not so.valid
    o0 = _run(other, x, y, al, ll, chunk=8, exp_domain=True)
    o1 = _run(other, x, y, al, ll, chunk=8, exp_domain=True)
    assert float(so.cur) == 0.0 and abs(o1[0] - plain[0]) < 1e-4 * plain[0] and o0[0] == plain[0]
    # a shift that does not fit: every exp overflows.  NaN out, flag up; after the flag has been seen the next step is a plain-form one
    import warnings
    st.cur.fill_(-150.0)
    bad = _run(model, x, y, al, ll, chunk=8, exp_domain=True)
    assert not np.isfinite(bad[0]) and not np.isfinite(bad[1]).all()
    torch.cuda.synchronize()
    assert int(st.flag) == 1
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        rec = _run(model, x, y, al, ll, chunk=8, exp_domain=True)
    assert any("under/overflowed" in str(m.message) for m in w) and st.flagged_steps == 1
    assert rec[0] == d[0] and st.valid and 25.0 < float(st.cur) < 35.0 and int(st.flag) == 0
    again = _run(model, x, y, al, ll, chunk=8, exp_domain=True)
    assert abs(again[0] - d[0]) < 1e-4 * d[0]
    # underflow: a shift far above the logits loses every row sum - NaN as well, not the clamped finite numbers of round 2
    st.cur.fill_(400.0)
    bad = _run(model, x, y, al, ll, chunk=8, exp_domain=True)
    assert not np.isfinite(bad[0])


def test_sums_over_frames_in_two_passes_are_reproducible(monkeypatch):
    """round 6: dPD[b, u, :] = sum over t of dP - 32 f32 atomics per entry until now, whose order decided bf16 roundings in front of the d(label states) GEMM and moved
    whole rows of that gradient by 1e-5 from run to run - is written as partial rows and added in a fixed order (option 21 = 0: the atomic kernel).  Both forms give
    the same numbers; with two passes the gradient the label encoder receives is the same BITS in every run"""
    import ttmi.ops as ops
    from tt.model import Transducer
    model, x, y, al, ll = _training_sized(monkeypatch, "bf16")
    monkeypatch.setenv("TTMI_LABEL_VALUE_PRECISION", "off")               # (one label pass: the straight-through form hands the same gradient on)
    seen = {}
    inner = Transducer._label_states

    def spy(self, targets):
        out = inner(self, targets)
        out.register_hook(lambda g: seen.__setitem__("ddec", g.detach().clone()))
        return out
    monkeypatch.setattr(Transducer, "_label_states", spy)

    def once():
        model.joint.exp_shift_state(x.device).set(30.0)                   # (every run with the same shift: the next call's would follow this call's logits)
        out = _run(model, x, y, al, ll, chunk=8, exp_domain=True)
        return out[0], out[1], seen["ddec"].cpu().numpy()
    runs = [once() for _ in range(4)]
    ops.set_option(21, 0)
    try:
        atomic = once()
    finally:
        ops.set_option(21, 1)
    for r in runs[1:]:
        assert r[0] == runs[0][0] and np.array_equal(r[2], runs[0][2])
        assert rel_err(r[1], runs[0][1]) < 1e-6                           # (the parameters' own gradients still meet K-split atomics: 1e-7)
    assert atomic[0] == runs[0][0]
    assert rel_err(atomic[2], runs[0][2]) < 1e-4 and rel_err(atomic[1], runs[0][1]) < 1e-4


@pytest.mark.parametrize("J,V", [(1024, 4334), (2048, 6485)])
def test_bf16x3_loss_gradient_written_as_planes(monkeypatch, J, V):
    """bf16x3 mode, round 6: the loss gradient overwrites the f32 logits as the two bf16 planes [hi | lo] the joint's three-term backward multiplies
    (ttmi_rnnt_loss_bwd_split + ttmi_joint_bwd_split) instead of f32 values that a split pass reads again.  Same planes, same products: every gradient as with
    TTMI_X3_SPLIT_GRAD=0 up to the weight gradients' atomic order; ragged lengths and a label equal to the blank included (_training_sized)"""
    import ttmi.ops as ops
    model, x, y, al, ll = _training_sized(monkeypatch, "bf16x3", J, V)
    seen = []
    inner = ops.rnnt_loss_bwd_split
    monkeypatch.setattr(ops, "rnnt_loss_bwd_split", lambda *a, **k: (seen.append(1), inner(*a, **k))[1])
    planes = _run(model, x, y, al, ll, chunk=4)
    assert len(seen) == 2                                     # two chunks, both through the split form
    monkeypatch.setenv("TTMI_X3_SPLIT_GRAD", "0")
    plain = _run(model, x, y, al, ll, chunk=4)
    assert len(seen) == 2
    assert planes[0] == plain[0]
    assert rel_err(planes[1], plain[1]) < 1e-6
    for n in plain[2]:
        assert rel_err(planes[2][n], plain[2][n]) < 2e-6, n


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
    model.joint.exp_shift_state(x.device).set(0.0)
    fast = _run(model, x, y, al, ll, chunk=B, exp_domain=True)
    e_plain, e_fast = rel_err(plain[1], ref[1]), rel_err(fast[1], ref[1])
    print("U=200: loss fp32 %.4f  bf16 %.4f  exp-domain %.4f;  gradient error vs fp32: bf16 %.2e, exp-domain %.2e" % (ref[0], plain[0], fast[0], e_plain, e_fast))
    assert abs(fast[0] - ref[0]) < 2e-4 * ref[0] and abs(plain[0] - ref[0]) < 2e-4 * ref[0]
    assert fast[0] != plain[0]                               # (the exp-domain kernels ran: the two forms round differently)
    assert e_fast < max(1.5 * e_plain, 5e-3)


@pytest.mark.parametrize("B,T,U", [(7, 187, 29), (5, 333, 21), (3, 487, 42), (33, 50, 19)])
def test_exp_domain_runs_at_any_row_count_vs_oracle(monkeypatch, B, T, U):
    """train.py:32-35 trims every batch to its own maximum lengths, so B * T * (U + 1) is arbitrary; round 3's exp-domain path needed a
    multiple of 64 (the wgrad's K-tile) and silently fell back otherwise.  Now the library pads the reduction:
    odd row counts, ragged lengths - the exp-domain kernels must RUN, and costs / gradients are held to the
    oracle's joint (float64) + C lattice on the same encoder states, with the bounds of test_exp_domain_vs_oracle."""
    import ttmi.ops as ops
    from oracle import tt_oracle as O
    from oracle.rnnt_c import rnnt_loss_c
    from tt.model import _JointLossFn
    model, _, _, _, _ = _training_sized(monkeypatch, "bf16")
    g = torch.Generator(device="cuda").manual_seed(100 * B + U)
    x = torch.randn(B, T, 512, device="cuda", generator=g)
    y = torch.randint(1, 4334, (B, U), device="cuda", generator=g)
    al = torch.randint(T // 2, T + 1, (B,), device="cuda", generator=g).int()
    ll = torch.randint(U // 2, U + 1, (B,), device="cuda", generator=g).int()
    al[0], ll[0] = T, U
    rows = B * T * (U + 1)
    j = model.joint
    chunk = j.default_loss_chunk(B, T, U + 1, True, 1)
    assert (chunk * T * (U + 1)) % 64 != 0 and chunk * T * (U + 1) >= 32768, (rows, chunk)       # the case round 3 could not run
    with torch.no_grad():
        enc_s, dec_s = model._encode(x, y)
    st = j.exp_shift_state(x.device)
    st.set(0.0)
    calls = []
    orig = ops.joint_bwd_exp
    monkeypatch.setattr(ops, "joint_bwd_exp", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    enc_l, dec_l = enc_s.clone().requires_grad_(True), dec_s.clone().requires_grad_(True)
    model.zero_grad()
    loss = _JointLossFn.apply(enc_l, dec_l, j.forward_layer.weight, j.forward_layer.bias, j.project_layer.weight, j.project_layer.bias,
                              y.int().contiguous(), al, ll, 1, chunk, "mean", st, True)
    loss.backward()
    torch.cuda.synchronize()
    assert len(calls) == -(-B // chunk) and int(st.flag) == 0                   # every chunk took the exp-domain kernels
    sd = {"joint." + k: v.detach().double().cpu().numpy() for k, v in j.state_dict().items()}
    z, cache = O.joint_fwd(enc_s.double().cpu().numpy(), dec_s.double().cpu().numpy(), sd)
    want_loss, want_costs, dz = rnnt_loss_c(z.astype(np.float32), y.int().cpu().numpy(), al.cpu().numpy(), ll.cpu().numpy())
    grads = {}
    denc, ddec = O.joint_bwd(dz.astype(np.float64), cache, sd, grads)
    el = abs(float(loss.detach()) - float(want_loss)) / float(want_loss)
    errs = {"denc": rel_err(enc_l.grad.cpu().numpy(), denc), "ddec": rel_err(dec_l.grad.cpu().numpy(), ddec)}
    for k, p in j.named_parameters():
        errs["g_" + k] = rel_err(p.grad.cpu().numpy(), grads["joint." + k])
    print("exp-domain form at %d rows (%d per chunk, %d mod 64): loss rel %.2e, %s" % (rows, chunk * T * (U + 1), (chunk * T * (U + 1)) % 64, el,
                                                                                      ", ".join("%s %.2e" % kv for kv in errs.items())))
    assert el < 5e-5
    for k, e in errs.items():
        assert e < 1.5e-2, (k, e)
    assert errs["g_project_layer.bias"] < 5e-3


def test_exp_domain_warns_once_when_it_cannot_run(monkeypatch):
    """a lattice too small for the persistent kernels: the plain fused form runs (same numbers as without the switch) and says so, once"""
    import warnings
    import tt.model as M
    model, x, y, al, ll = _training_sized(monkeypatch, "bf16")
    M._warned_no_exp.clear()
    base = _run(model, x[:2], y[:2], al[:2].clone().fill_(200), ll[:2].clone().fill_(20))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        a = _run(model, x[:2], y[:2], al[:2].clone().fill_(200), ll[:2].clone().fill_(20), exp_domain=True)
        b = _run(model, x[:2], y[:2], al[:2].clone().fill_(200), ll[:2].clone().fill_(20), exp_domain=True)
    assert a[0] == base[0] == b[0]
    assert sum("outside the persistent" in str(m.message) for m in w) == 1
