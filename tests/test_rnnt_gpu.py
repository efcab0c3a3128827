"""HIP RNN-T loss vs the CPU oracle (numpy float64 / C).  Runs on the GPU box."""
import numpy as np
import pytest
import torch

from conftest import rel_err, load_golden
from oracle import tt_oracle as O
from oracle.rnnt_c import rnnt_loss_c
from test_oracle_rnnt import KAT_COST, KAT_GRAD, KAT_LABELS, KAT_LOGITS

pytestmark = pytest.mark.gpu
TOL = 1e-4      # north_star: fp32 loss / gradients within 1e-4 rel of the reference CPU path


def run_hip(x, y, tl, ul, reduction="mean", blank=0):
    from warprnnt_pytorch import RNNTLoss
    acts = torch.tensor(x, dtype=torch.float32, device="cuda", requires_grad=True)
    loss = RNNTLoss(blank=blank, reduction=reduction)(
        acts, torch.tensor(y, dtype=torch.int32, device="cuda"), torch.tensor(tl, dtype=torch.int32, device="cuda"),
        torch.tensor(ul, dtype=torch.int32, device="cuda"))
    if reduction == "none":
        loss.sum().backward()
    else:
        loss.backward()
    return loss.detach().cpu().numpy(), acts.grad.cpu().numpy()


def test_known_answer_vector():
    loss, g = run_hip(KAT_LOGITS, KAT_LABELS, [2], [2])
    assert loss.shape == (1,)
    assert abs(loss[0] - KAT_COST) < 1e-5
    assert np.abs(g - KAT_GRAD).max() < 1e-5


@pytest.mark.parametrize("B,T,U,V,ragged", [(3, 17, 9, 11, True), (2, 40, 6, 48, False), (4, 33, 70, 37, True),
                                             (2, 20, 130, 19, True), (1, 9, 300, 7, False), (5, 64, 63, 130, True),
                                             (2, 50, 64, 33, True), (1, 1, 0, 5, False), (2, 7, 600, 5, True)])
def test_random_vs_float64_oracle(B, T, U, V, ragged):
    rng = np.random.default_rng(B * 1000 + T + U + V)
    x = (rng.normal(size=(B, T, U + 1, V)) * 2).astype(np.float32)
    y = rng.integers(1, V, size=(B, max(U, 0)))
    tl = np.full(B, T)
    ul = np.full(B, U)
    if ragged and B > 1:
        tl[1:] = rng.integers(1, T + 1, size=B - 1)
        ul[1:] = rng.integers(0, U + 1, size=B - 1)
    want = O.rnnt_loss(x.astype(np.float64), y, tl, ul)
    loss, g = run_hip(x, y.reshape(B, U), tl, ul)
    assert abs(loss[0] - want[0]) / abs(want[0]) < TOL
    assert rel_err(g, want[2]) < TOL
    # exact zeros outside the valid lattice
    for b in range(B):
        assert np.all(g[b, tl[b]:] == 0) and np.all(g[b, :, ul[b] + 1:] == 0)


@pytest.mark.parametrize("reduction", ["sum", "none"])
def test_reductions(reduction):
    rng = np.random.default_rng(11)
    x = rng.normal(size=(3, 12, 5, 9)).astype(np.float32)
    y = rng.integers(1, 9, size=(3, 4))
    want = O.rnnt_loss(x.astype(np.float64), y, [12, 12, 12], [4, 4, 4], reduction="sum")
    loss, g = run_hip(x, y, [12] * 3, [4] * 3, reduction)
    if reduction == "none":
        assert loss.shape == (3,) and rel_err(loss, want[1]) < TOL
    else:
        assert abs(loss[0] - want[0]) / want[0] < TOL
    assert rel_err(g, want[2]) < TOL


def test_nonzero_blank():
    rng = np.random.default_rng(12)
    x = rng.normal(size=(2, 10, 4, 8)).astype(np.float32)
    y = rng.integers(0, 7, size=(2, 3))
    want = O.rnnt_loss(x.astype(np.float64), y, [10, 10], [3, 3], blank=7)
    loss, g = run_hip(x, y, [10, 10], [3, 3], blank=7)
    assert abs(loss[0] - want[0]) / want[0] < TOL and rel_err(g, want[2]) < TOL


def test_golden_logits_costs():
    for name in ("tiny_klong", "tiny_kshort"):
        z, _ = load_golden(name)
        for tag in ("full", "ragged"):
            from warprnnt_pytorch import RNNTLoss
            acts = torch.tensor(z["logits"], device="cuda")
            costs = RNNTLoss(reduction="none", check_lengths=False)(
                acts, torch.tensor(z["targets"], dtype=torch.int32, device="cuda"),
                torch.tensor(z[tag + "/act_lens"], device="cuda"), torch.tensor(z[tag + "/label_lens"], device="cuda"))
            assert rel_err(costs.cpu().numpy(), z[tag + "/costs"]) < TOL


def test_argument_errors():
    from warprnnt_pytorch import RNNTLoss
    acts = torch.zeros(2, 4, 3, 5, device="cuda")
    lab = torch.ones(2, 2, dtype=torch.int32, device="cuda")
    tl = torch.tensor([4, 4], dtype=torch.int32, device="cuda")
    ul = torch.tensor([2, 2], dtype=torch.int32, device="cuda")
    with pytest.raises(TypeError):
        RNNTLoss()(acts, lab.long(), tl, ul)
    with pytest.raises(ValueError):
        RNNTLoss()(acts, lab, torch.tensor([3, 3], dtype=torch.int32, device="cuda"), ul)     # T != max(act_lens)
    with pytest.raises(ValueError):
        RNNTLoss()(acts, lab, tl, torch.tensor([1, 1], dtype=torch.int32, device="cuda"))     # U+1 != max+1
    with pytest.raises(ValueError):
        RNNTLoss()(acts.cpu(), lab, tl, ul)


def test_full_size_properties():
    """BASELINE config 2 lattice (B=32,T=500,U=50,V=4334): size-independent properties +
    C-oracle spot check of two utterances."""
    from warprnnt_pytorch import RNNTLoss
    B, T, U, V = 32, 500, 50, 4334
    g = torch.Generator(device="cuda").manual_seed(5)
    acts = torch.randn(B, T, U + 1, V, device="cuda", generator=g).requires_grad_(True)
    lab = torch.randint(1, V, (B, U), device="cuda", generator=g, dtype=torch.int32)
    tl = torch.full((B,), T, dtype=torch.int32, device="cuda")
    ul = torch.full((B,), U, dtype=torch.int32, device="cuda")
    tl[3], ul[3] = 417, 31
    costs = RNNTLoss(reduction="none")(acts, lab, tl, ul)
    costs.sum().backward()
    gr = acts.grad
    # softmax shift invariance => every gradient row sums to zero
    assert float(gr.sum(-1).abs().max()) < 2e-5
    assert float(gr[3, 417:].abs().max()) == 0 and float(gr[3, :, 32:].abs().max()) == 0
    for b in (0, 3):
        xb = acts[b:b + 1].detach().cpu().numpy()
        _, cb, gb = rnnt_loss_c(xb, lab[b:b + 1].cpu().numpy(), tl[b:b + 1].cpu().numpy(), ul[b:b + 1].cpu().numpy(),
                                reduction="sum")
        assert abs(float(costs[b]) - cb[0]) / cb[0] < TOL
        assert rel_err(gr[b].cpu().numpy(), gb[0]) < TOL


@pytest.mark.parametrize("B,T,U,V", [(2, 30, 7, 48), (3, 21, 12, 131), (1, 9, 3, 4334)])
def test_bf16_row_padded_logits(B, T, U, V):
    """the bf16 pipeline hands the loss a [..., :V] view of a pitch-roundup(V,64) bf16 buffer and gets a gradient
    of the same layout back, pad columns exactly zero"""
    from warprnnt_pytorch import RNNTLoss
    rng = np.random.default_rng(V)
    Vp = (V + 63) // 64 * 64
    buf = torch.zeros(B, T, U + 1, Vp, device="cuda", dtype=torch.bfloat16)
    buf[..., :V] = torch.tensor(rng.normal(size=(B, T, U + 1, V)) * 2, device="cuda").to(torch.bfloat16)
    buf[..., V:] = 77.0                                     # poison the pad: must never be read
    acts = buf[..., :V].detach().requires_grad_(True)
    y = rng.integers(1, V, size=(B, U))
    tl = np.full(B, T, dtype=np.int32)
    ul = np.full(B, U, dtype=np.int32)
    if B > 1:
        tl[1], ul[1] = T - 4, U - 2
    raw = []
    acts.register_hook(raw.append)          # the gradient exactly as the loss's backward hands it to its producer
    loss = RNNTLoss()(acts, torch.tensor(y, dtype=torch.int32, device="cuda"), torch.tensor(tl, device="cuda"),
                      torch.tensor(ul, device="cuda"))
    loss.backward()
    g = raw[0]
    assert g.dtype is torch.bfloat16 and g.stride(-2) == Vp
    want = O.rnnt_loss(buf[..., :V].float().cpu().numpy().astype(np.float64), y, tl, ul)
    assert abs(float(loss) - want[0]) / want[0] < 1e-5       # the loss itself is computed in f32/f64 from the bf16 values
    assert rel_err(g.float().cpu().numpy(), want[2]) < 6e-3  # gradient rounded to bf16 (2^-9 relative)
    full = torch.as_strided(g, (B, T, U + 1, Vp), g.stride())
    assert float(full[..., V:].float().abs().max()) == 0


def test_length_check_cache_follows_tensor_identity_and_version():
    """the max-length validation (a host sync in warp-transducer's certify_inputs) is cached per lengths tensor OBJECT and version
    counter: an in-place change or a different tensor is validated again"""
    from warprnnt_pytorch import RNNTLoss
    B, T, U, V = 2, 5, 3, 7
    acts = torch.randn(B, T, U + 1, V, device="cuda")
    labels = torch.randint(1, V, (B, U), device="cuda", dtype=torch.int32)
    alen = torch.full((B,), T, device="cuda", dtype=torch.int32)
    llen = torch.full((B,), U, device="cuda", dtype=torch.int32)
    crit = RNNTLoss()
    a = crit(acts, labels, alen, llen)
    b = crit(acts, labels, alen, llen)                   # cached path
    assert torch.equal(a, b)
    alen[0] = T - 1
    alen[1] = T - 1                                      # in place: version changes, max is now T-1 != T
    with pytest.raises(ValueError):
        crit(acts, labels, alen, llen)
    with pytest.raises(ValueError):
        crit(acts, labels, torch.full((B,), T + 1, device="cuda", dtype=torch.int32), llen)


def test_timing_probes_fire_around_the_loss_kernels():
    """ttmi_probe_arm(i) makes the next RNN-T loss forward (point 1) and backward (point 2) record HIP events into pair i; an unarmed
    pair reads < 0"""
    from ttmi import ops
    from warprnnt_pytorch import RNNTLoss
    g = torch.Generator(device="cuda").manual_seed(3)
    B, T, U, V = 2, 30, 5, 40
    logits = torch.randn(B, T, U + 1, V, device="cuda", generator=g, requires_grad=True)
    labels = torch.randint(1, V, (B, U), device="cuda", generator=g).int()
    tl = torch.full((B,), T, dtype=torch.int32, device="cuda")
    ul = torch.full((B,), U, dtype=torch.int32, device="cuda")
    ops.probe_arm(7)
    RNNTLoss()(logits, labels, tl, ul).backward()
    torch.cuda.synchronize()
    assert ops.probe_read_ms(7, 1) > 0 and ops.probe_read_ms(7, 2) > 0
    assert ops.probe_read_ms(7, 1) < 0                   # consumed
    assert ops.probe_read_ms(9, 1) < 0                   # never armed


@pytest.mark.parametrize("B,T,U", [(3, 40, 50), (2, 37, 64), (2, 300, 200), (2, 25, 600), (1, 1, 0), (2, 90, 959),
                                   # odd CH * (U + 1) (chunk of CH diagonals x U + 1 labels): the shapes at which round 3's boundary ring sat 8 bytes
                                   # off its 16-byte granules - U + 1 = 201 -> CH 19, 149 -> 25, 255 -> 15, 129 -> 29; many utterances
                                   # per launch so that workgroups of different phase share CUs while the ring is polled
                                   (24, 220, 200), (16, 160, 148), (16, 120, 254), (16, 200, 128)])
def test_lattice_kernels_agree_bit_for_bit(B, T, U):
    """rnnt_lattice_lds_kernel (one workgroup per utterance and direction: a wave per 64 labels, frontier hand-off through LDS, a helper
    wave for the memory traffic) walks the same recursion in the same arithmetic as the one-wave kernel of round 1: costs and gradients
    are identical to the last bit, ragged lengths included; U + 1 = 960 is the widest workgroup (15 + 1 waves)"""
    from ttmi import ops
    rng = np.random.default_rng(7 * T + U)
    V = 6
    x = (rng.normal(size=(B, T, U + 1, V)) * 2).astype(np.float32)
    y = rng.integers(1, V, size=(B, U))
    tl, ul = np.full(B, T), np.full(B, U)
    if B > 1:
        tl[1:] = rng.integers(1, T + 1, size=B - 1)
        ul[1:] = rng.integers(0, U + 1, size=B - 1)
    try:
        ops.set_option(9, 1)
        l_old, g_old = run_hip(x, y.reshape(B, U), tl, ul)
    finally:
        ops.set_option(9, 0)
    l_new, g_new = run_hip(x, y.reshape(B, U), tl, ul)
    assert np.array_equal(l_old, l_new) and np.array_equal(g_old, g_new)
    want = O.rnnt_loss(x.astype(np.float64), y, tl, ul)
    assert abs(l_new[0] - want[0]) / abs(want[0]) < TOL and rel_err(g_new, want[2]) < TOL
