"""The handle protocol of `tt.model.DeferredLogits` (what `model(inputs, targets)` returns in the bf16 pipeline so that train.py:51-53
runs unchanged on the fused joint + loss path) on the CPU: the protocol itself is device-independent - a subclass whose `_produce` forms
the logits with torch ops stands in for the HIP joint.  Metadata never materialises; any other use carries on with exactly the tensor the
eager call returns, inside the same autograd graph; a gradient that reaches the handle itself is passed on."""
import io

import pytest
import torch

import tt.model as M


class _CpuDeferred(M.DeferredLogits):
    def _produce(self):
        j = self._joint
        B, T, U1 = self._enc.shape[0], self._enc.shape[1], self._dec.shape[1]
        x = torch.cat([self._enc[:, :, None, :].expand(B, T, U1, -1), self._dec[:, None, :, :].expand(B, T, U1, -1)], -1)
        return j.project_layer(torch.tanh(j.forward_layer(x)))           # tt/model.py:20-39 of the reference, concat form


@pytest.fixture
def case():
    torch.manual_seed(0)
    j = M.JointNet(8, 6, 5)
    enc = torch.randn(2, 4, 4, requires_grad=True)
    dec = torch.randn(2, 3, 4, requires_grad=True)
    return j, enc, dec


def test_metadata_is_answered_by_the_handle(case):
    j, enc, dec = case
    h = _CpuDeferred(j, enc, dec, 0)
    assert isinstance(h, torch.Tensor) and isinstance(h, M.DeferredLogits)
    assert h.shape == (2, 4, 3, 5) and h.size(1) == 4 and h.dim() == 4 and h.ndim == 4 and len(h) == 2 and h.numel() == 120
    assert h.dtype is torch.float32 and h.device.type == "cpu" and not h.is_cuda and h.requires_grad and h.is_floating_point()
    assert not h.is_materialized
    with torch.no_grad():
        g = _CpuDeferred(j, enc, dec, 0)
    assert not g.requires_grad and not g.is_materialized


def test_foreign_use_materialises_the_eager_tensor(case):
    j, enc, dec = case
    want = _CpuDeferred(j, enc, dec, 0)._produce()
    h = _CpuDeferred(j, enc, dec, 0)
    y = h.float()
    assert h.is_materialized and type(y) is torch.Tensor and torch.equal(y, want)
    assert h.materialize() is h.materialize()                             # produced once, kept
    assert torch.equal(h[0, 1], want[0, 1]) and torch.equal(torch.softmax(h, -1), torch.softmax(want, -1))
    assert torch.equal(h + 1.0, want + 1.0) and torch.equal(torch.cat([h, h], 0), torch.cat([want, want], 0))
    assert h.stride() == want.stride() and h.is_contiguous() == want.is_contiguous() and h.grad_fn is not None
    assert repr(h) == repr(h.materialize())
    buf = io.BytesIO()
    torch.save(h, buf)
    buf.seek(0)
    assert torch.equal(torch.load(buf), want)


def test_autograd_runs_through_the_materialised_logits(case):
    j, enc, dec = case
    want = _CpuDeferred(j, enc, dec, 0)._produce()
    (want * want).sum().backward()
    ge, gd, gw = enc.grad.clone(), dec.grad.clone(), j.project_layer.weight.grad.clone()
    enc.grad = dec.grad = None
    j.zero_grad()
    h = _CpuDeferred(j, enc, dec, 0)
    with torch.no_grad():
        float(h.max())                                                    # first touched where grad mode is off: still produced with its graph
    (h * h).sum().backward()
    assert torch.equal(enc.grad, ge) and torch.equal(dec.grad, gd) and torch.equal(j.project_layer.weight.grad, gw)


def test_gradient_that_reaches_the_handle_is_passed_on(case):
    """a caller below the Python API (a foreign autograd.Function applied to the handle) records the HANDLE as its input"""
    j, enc, dec = case

    class Sum(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            ctx.shape = x.shape
            return x.sum().reshape(1) * 1.0

        @staticmethod
        def backward(ctx, g):
            return g.expand(ctx.shape)

    _CpuDeferred(j, enc, dec, 0)._produce().sum().backward()
    ge = enc.grad.clone()
    enc.grad = None
    Sum.apply(_CpuDeferred(j, enc, dec, 0)).backward()
    assert torch.allclose(enc.grad, ge)


def test_switches(monkeypatch):
    from tt.utils import AttrDict
    cfg = AttrDict({})
    monkeypatch.delenv("TTMI_DEFERRED_LOGITS", raising=False)
    assert M.deferred_logits_enabled(cfg, 1) and not M.deferred_logits_enabled(cfg, 0)          # bf16 pipeline only, by default
    assert not M.deferred_logits_enabled(AttrDict(dict(deferred_logits=False)), 1)
    assert M.deferred_logits_enabled(AttrDict(dict(deferred_logits=True)), 0)
    monkeypatch.setenv("TTMI_DEFERRED_LOGITS", "0")
    assert not M.deferred_logits_enabled(AttrDict(dict(deferred_logits=True)), 1)
    monkeypatch.setenv("TTMI_DEFERRED_LOGITS", "1")
    assert M.deferred_logits_enabled(cfg, 0)


def test_loss_shim_still_refuses_cpu_logits(case):
    from warprnnt_pytorch import RNNTLoss
    j, enc, dec = case
    h = _CpuDeferred(j, enc, dec, 0)
    with pytest.raises(ValueError):
        RNNTLoss()(h, torch.ones(2, 2, dtype=torch.int32), torch.full((2,), 4, dtype=torch.int32), torch.full((2,), 2, dtype=torch.int32))
    assert not h.is_materialized
