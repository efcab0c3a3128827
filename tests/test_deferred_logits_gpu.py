"""train.py:51-53 UNCHANGED on the fused joint + loss path: in the bf16 pipeline `logits = model(inputs, targets)`
is a `tt.model.DeferredLogits` handle; `RNNTLoss` consumes it through `_JointLossFn` (the exp-domain form at training sizes) and anything
else gets the real logits, bit-identical to the eager call."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, rel_err
from test_fused_loss_gpu import _training_sized
from test_model_gpu import build

pytestmark = pytest.mark.gpu


def _grads(model, xi):
    return torch.cat([xi.grad.reshape(-1)] + [p.grad.reshape(-1) for p in model.parameters()]).cpu().numpy()


def test_materialisation_is_bit_identical_to_the_eager_logits(monkeypatch):
    import tt.model as M
    z, sd = load_golden("tiny_klong")
    model = build(sd)
    monkeypatch.setenv("TTMI_PRECISION", "bf16")
    x, y = torch.tensor(z["inputs"], device="cuda"), torch.tensor(z["targets"], device="cuda")
    monkeypatch.setenv("TTMI_DEFERRED_LOGITS", "0")
    eager = model(x, y)
    assert type(eager) is torch.Tensor
    monkeypatch.delenv("TTMI_DEFERRED_LOGITS")
    h = model(x, y)
    assert isinstance(h, M.DeferredLogits) and h.shape == eager.shape and h.dtype is eager.dtype and h.device == eager.device
    assert h.requires_grad and not h.is_materialized
    real = h.float()
    assert h.is_materialized and torch.equal(real, eager.float())
    assert h.stride() == eager.stride() and h.data_ptr() == h.materialize().data_ptr()
    assert torch.equal(h[1, 3:5], eager[1, 3:5]) and torch.equal(F.log_softmax(h.float(), -1), F.log_softmax(eager.float(), -1))
    # the gradient of a foreign loss on the handle = the gradient of the same loss on the eager logits
    res = []
    for deferred in (False, True):
        monkeypatch.setenv("TTMI_DEFERRED_LOGITS", "1" if deferred else "0")
        model.zero_grad()
        xi = x.clone().requires_grad_(True)
        lg = model(xi, y)
        assert isinstance(lg, M.DeferredLogits) == deferred
        (lg.float() ** 2).mean().backward()
        res.append(_grads(model, xi))
    assert rel_err(res[1], res[0]) < 1e-6           # (same kernels; f32 atomics in the weight gradients)


def test_fp32_mode_keeps_eager_logits_unless_asked(monkeypatch):
    import tt.model as M
    from warprnnt_pytorch import RNNTLoss
    z, sd = load_golden("tiny_klong")
    model = build(sd)
    monkeypatch.setenv("TTMI_PRECISION", "fp32")
    x, y = torch.tensor(z["inputs"], device="cuda"), torch.tensor(z["targets"], device="cuda")
    al, ll = torch.tensor(z["ragged/act_lens"], device="cuda"), torch.tensor(z["ragged/label_lens"], device="cuda")
    assert type(model(x, y)) is torch.Tensor
    out = []
    for deferred in (False, True):
        monkeypatch.setenv("TTMI_DEFERRED_LOGITS", "1" if deferred else "0")
        model.zero_grad()
        xi = x.clone().requires_grad_(True)
        lg = model(xi, y)
        loss = RNNTLoss(check_lengths=False)(lg, y.int(), al, ll)
        loss.backward()
        if deferred:
            assert isinstance(lg, M.DeferredLogits) and not lg.is_materialized        # the fused memory form ran
        out.append((float(loss.detach()), _grads(model, xi)))
    assert out[0][0] == out[1][0] and abs(out[0][0] - float(z["ragged/loss"])) < 1e-4 * float(z["ragged/loss"])
    assert rel_err(out[1][1], out[0][1]) < 1e-5


def test_train_py_call_sequence_runs_the_exp_domain_kernels(monkeypatch):
    """B=8, T=200, U=20, J=1024, V=4334 (the persistent kernels' sizes): `criterion(model(x, y), y.int(), al, ll)` gives the loss and
    gradients of `model.loss(..., exp_domain=True)`, never forms the logits and runs the exp-domain projection"""
    import tt.model as M
    import ttmi.ops as ops
    from warprnnt_pytorch import RNNTLoss
    model, x, y, al, ll = _training_sized(monkeypatch, "bf16")
    crit = RNNTLoss(check_lengths=False)
    calls = {"exp": 0, "plain": 0}
    orig_exp, orig_plain = ops.joint_fwd_exp, ops.joint_fwd
    monkeypatch.setattr(ops, "joint_fwd_exp", lambda *a, **k: (calls.__setitem__("exp", calls["exp"] + 1), orig_exp(*a, **k))[1])
    monkeypatch.setattr(ops, "joint_fwd", lambda *a, **k: (calls.__setitem__("plain", calls["plain"] + 1), orig_plain(*a, **k))[1])

    def two_call():
        model.zero_grad()
        xi = x.clone().requires_grad_(True)
        logits = model(xi, y)                                            # train.py:51
        loss = crit(logits, y.int(), al, ll)                             # train.py:53
        loss.backward()
        assert isinstance(logits, M.DeferredLogits) and not logits.is_materialized
        return float(loss.detach()), _grads(model, xi)

    def explicit():
        model.zero_grad()
        xi = x.clone().requires_grad_(True)
        loss = model.loss(xi, al, y, ll, check_lengths=False, exp_domain=True)
        loss.backward()
        return float(loss.detach()), _grads(model, xi)

    seed = two_call()                                                    # first use: the plain fused form seeds the shift on the device
    assert calls == {"exp": 0, "plain": 1}
    a = two_call()
    assert calls == {"exp": 1, "plain": 1}
    b = explicit()
    assert calls == {"exp": 2, "plain": 1}
    assert a[0] == b[0] and rel_err(a[1], b[1]) < 1e-3                   # same kernels on the same operands (f32 atomics in the wgrads)
    assert abs(a[0] - seed[0]) < 1e-4 * seed[0]
    # per-utterance costs without gradients stay on the fused path; with gradients they go through the real logits
    with torch.no_grad():
        lg = model(x, y)
        costs = RNNTLoss(reduction="none", check_lengths=False)(lg, y.int(), al, ll)
    assert costs.shape == (8,) and not lg.is_materialized and abs(float(costs.mean()) - a[0]) < 1e-4 * a[0]
    model.zero_grad()
    xi = x.clone().requires_grad_(True)
    lg = model(xi, y)
    per = RNNTLoss(reduction="none", check_lengths=False)(lg, y.int(), al, ll)
    (per * torch.arange(1, 9, device="cuda")).sum().backward()           # per-utterance upstream gradients
    assert lg.is_materialized and xi.grad is not None and bool(torch.isfinite(xi.grad).all())
    ratio = xi.grad[7].norm() / xi.grad[0].norm()
    assert 1.0 < float(ratio) < 64.0


def test_train_py_shaped_loop_matches_the_explicit_fused_loop(monkeypatch):
    """train.py:46-65 as written - zero_grad, model(), criterion(), backward, clip_grad_norm_, step - with tt.optim.Optimizer in the bf16
    pipeline, against the same loop calling model.loss(exp_domain=True): same loss trajectory"""
    from tt.optim import Optimizer
    from tt.utils import AttrDict
    from warprnnt_pytorch import RNNTLoss
    ocfg = AttrDict(dict(type="sgd", lr=0.002, momentum=0.9, decay_ratio=0.5, weight_decay=0, nesterov=None))
    traj = []
    for explicit in (False, True):
        model, x, y, al, ll = _training_sized(monkeypatch, "bf16")
        opt = Optimizer(model.parameters(), ocfg)
        crit = RNNTLoss(check_lengths=False)
        losses = []
        for step in range(4):
            opt.zero_grad()
            if explicit:
                loss = model.loss(x, al, y, ll, check_lengths=False, exp_domain=True)
            else:
                logits = model(x, y)
                loss = crit(logits, y.int(), al, ll)
            loss.backward()
            losses.append(float(loss))
            torch.nn.utils.clip_grad_norm_(model.parameters(), 200.0)
            opt.step()
            del loss
        traj.append(losses)
    print("two-call (deferred) %s\nexplicit            %s" % (traj[0], traj[1]))
    assert traj[0][0] == traj[1][0]
    assert np.allclose(traj[0], traj[1], rtol=2e-4)
    assert traj[0][-1] < traj[0][0]
