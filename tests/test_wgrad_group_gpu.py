"""Grouped weight gradients (include/ttmi.h, ttmi_wgrad_group): the deferred wgrad GEMMs of several encoder layers in one launch, one
256 x 128 tile per workgroup over the whole reduction - no split along K, no atomics.  Exact-integer problems (every product and sum is an
integer below 2^24, so any summation order gives the same f32) pin the tiling and the column sums; layer-level runs pin the deferred
backward passes against the immediate ones and the run-to-run bit identity the atomics never had."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _ints(shape, g, lo=-2, hi=3):
    return torch.randint(lo, hi, shape, device="cuda", generator=g).to(torch.bfloat16)


def _group(problems):
    from ttmi import ops
    descs = (ops.WgradDesc * len(problems))()
    for d, (A, B, C, col) in zip(descs, problems):
        K, M = A.shape
        d.A, d.B, d.C, d.colsum = A.data_ptr(), B.data_ptr(), C.data_ptr(), (col.data_ptr() if col is not None else 0)
        d.M, d.N, d.K, d.lda, d.ldb, d.ldc = M, B.shape[1], K, A.stride(0), B.stride(0), C.stride(0)
    ops.check(ops.lib().ttmi_wgrad_group(descs, ctypes.c_int(len(problems)), ops._stream()), "ttmi_wgrad_group")


def test_exact_integers_all_shapes_of_a_layer_and_a_misfit():
    g = torch.Generator(device="cuda").manual_seed(11)
    K = 1024
    shapes = [(1536, 512, False), (512, 512, False), (1024, 512, True), (512, 1024, False),      # one audio layer (qkv, o, w1 + b1, w2)
              (256, 256, True), (512, 128, False),                                               # smallest tilings (colsum needs 2 column tiles)
              (384, 512, True)]                                                                  # M % 256 != 0: takes the single-problem path
    probs, want = [], []
    for M, N, cs in shapes * 3:                      # 21 problems: more than one launch of 16
        A, B = _ints((K, M), g), _ints((K, N), g)
        C = torch.randint(-8, 9, (M, N), device="cuda", generator=g).float()
        col = torch.randint(-8, 9, (M,), device="cuda", generator=g).float() if cs else None
        want.append((C + A.float().t() @ B.float(), None if col is None else col + A.float().sum(0)))
        probs.append((A, B, C, col))
    _group(probs)
    torch.cuda.synchronize()
    for (A, B, C, col), (wc, wcol) in zip(probs, want):
        assert torch.equal(C, wc), (A.shape, B.shape)
        if col is not None:
            assert torch.equal(col, wcol), A.shape


@pytest.mark.parametrize("reserve,layers", [(8, 4), (32, 4), (4, 4), (32, 3), (64, 5), (32, 1)])
def test_surplus_tiles_as_k_pieces_under_a_cu_reservation(reserve, layers):
    """data-parallel backward (ttmi_stream_reserve_cus): the four-layer launch is 256 tiles - one round of 256 workgroups, two of 248 or 224.  The kernel then cuts
    the few surplus tiles of every XCD's list into K-pieces, one per workgroup, added atomically (round 6; exact here: integer sums in any order).  Option 20 = 0
    keeps whole tiles; both give the same numbers, as does the unreserved launch"""
    from ttmi import ops
    g = torch.Generator(device="cuda").manual_seed(13 + reserve + layers)
    K = 8192
    shapes = [(1536, 512, False), (512, 512, False), (1024, 512, True), (512, 1024, False)] * layers
    ops_in = [(_ints((K, M), g, -1, 2), _ints((K, N), g, -1, 2), torch.randint(-8, 9, (M, N), device="cuda", generator=g).float(),
               torch.randint(-8, 9, (M,), device="cuda", generator=g).float() if cs else None) for M, N, cs in shapes]
    want = [(C + A.float().t() @ B.float(), None if col is None else col + A.float().sum(0)) for A, B, C, col in ops_in]
    for pieces in (1, 0):
        probs = [(A, B, C.clone(), None if col is None else col.clone()) for A, B, C, col in ops_in]
        ops.set_option(20, pieces)
        ops.reserve_cus(reserve)
        try:
            _group(probs)
            torch.cuda.synchronize()
        finally:
            ops.reserve_cus(0)
            ops.set_option(20, 1)
        for (A, B, C, col), (wc, wcol) in zip(probs, want):
            assert torch.equal(C, wc), (pieces, A.shape, B.shape)
            if col is not None:
                assert torch.equal(col, wcol), (pieces, A.shape)


def test_padded_pitches_and_long_reduction():
    """operands that are column slices of wider buffers (row pitch > width), K = 16000 as in a C2 step"""
    g = torch.Generator(device="cuda").manual_seed(12)
    K = 16000
    Abig, Bbig = _ints((K, 1536 + 64), g, -1, 2), _ints((K, 512 + 8), g, -1, 2)
    A, B = Abig[:, 64:], Bbig[:, 8:]
    Cbig = torch.zeros(1536, 640, device="cuda")
    C = Cbig[:, 128:]
    _group([(A, B, C, None)])
    assert torch.equal(C, A.float().t() @ B.float())
    assert not Cbig[:, :128].any()


def _layer(monkeypatch, rows_b=8, L=512):
    from tt.encoder import BuildEncoder
    from tt.utils import AttrDict
    from ttmi.train import FlatModel
    monkeypatch.setenv("TTMI_PRECISION", "bf16")
    cfg = AttrDict(dict(enc=dict(n_layer=3, d_model=512, n_head=8, d_head=64, d_inner=1024, max_input_length=64), dropout=0.1))
    torch.manual_seed(5)
    enc = BuildEncoder(cfg).cuda().train()
    flat = FlatModel(enc)
    g = torch.Generator(device="cuda").manual_seed(6)
    x = torch.randn(rows_b, L, 512, device="cuda", generator=g)
    cot = torch.randn(rows_b, L, 512, device="cuda", generator=g)
    return enc, flat, x, cot


def _step(enc, flat, x, cot, seed):
    from ttmi import ops
    flat.zero_grad()
    torch.manual_seed(seed)                                  # same dropout masks in every run
    xi = x.clone().requires_grad_(True)
    (enc(xi) * cot).sum().backward()
    ops.join_side_streams()
    torch.cuda.synchronize()
    return flat.grad.clone(), xi.grad.clone()


def test_deferred_layers_match_immediate_and_repeat_bit_for_bit(monkeypatch):
    """3 audio-sized layers (B*L = 4096 rows): the queue launches once when the first layer's backward ends (12 problems, one launch)"""
    from ttmi import ops
    enc, flat, x, cot = _layer(monkeypatch)
    base_g, base_dx = _step(enc, flat, x, cot, 3)
    launches = []
    orig = ops.WgradQueue.maybe_flush

    def counting(self, force=False):
        n = len(self.descs)
        orig(self, force)
        if n and not self.descs:
            launches.append(n)
    monkeypatch.setattr(ops.WgradQueue, "maybe_flush", counting)
    flat.enable_grouped_wgrads(4)
    try:
        g1, dx1 = _step(enc, flat, x, cot, 3)
        g2, dx2 = _step(enc, flat, x, cot, 3)
    finally:
        flat.disable_grouped_wgrads()
    assert launches == [12, 12]
    assert torch.equal(dx1, base_dx)                                          # the data path is untouched
    assert rel_err(g1.cpu().numpy(), base_g.cpu().numpy()) < 2e-5             # same products, another summation order
    names = [n for n, _ in enc.named_parameters()]
    deferred = [i for i, n in enumerate(names) if n.endswith(("qkv_net.weight", "o_net.weight", "CoreNet.0.weight", "CoreNet.0.bias", "CoreNet.3.weight"))]
    assert len(deferred) == 15
    for i in deferred:                                                        # no atomics: identical from run to run
        o, n = flat.offsets[i], flat.params[i].numel()
        assert torch.equal(g1[o:o + n], g2[o:o + n]), names[i]


def test_gradient_hooks_fire_after_the_group_has_run(monkeypatch):
    """GradSync's per-parameter hooks of the deferred weights must see the finished gradient: they run after the grouped launch"""
    from ttmi import ops
    enc, flat, x, cot = _layer(monkeypatch)
    order = []
    flat.enable_grouped_wgrads(2)
    orig = ops.WgradQueue.maybe_flush
    monkeypatch.setattr(ops.WgradQueue, "maybe_flush", lambda self, force=False: (order.append(("flush?", len(self.descs))), orig(self, force))[1])
    names = dict((id(p), n) for n, p in enc.named_parameters())
    for p in flat.params:
        p._ttmi_on_grad = (lambda p=p: order.append(("hook", names[id(p)], bool(p.grad.abs().sum() > 0))))
    try:
        _step(enc, flat, x, cot, 4)
    finally:
        flat.disable_grouped_wgrads()
        for p in flat.params:
            del p._ttmi_on_grad
    hooks = [e for e in order if e[0] == "hook"]
    assert len(hooks) == len(flat.params) and all(e[2] for e in hooks), [e for e in hooks if not e[2]]
    # with 2 layers per group the queue empties after layer 1 (8 problems) and again after layer 0 (4 problems)
    assert [n for tag, n in (e[:2] for e in order if e[0] == "flush?") if n] == [4, 8, 4]


def test_first_layer_kept_out_of_the_groups_for_data_parallel_runs(monkeypatch):
    """immediate_first_layer: layers 2 and 1 are launched together when layer 0's backward starts, layer 0 keeps its own launches"""
    from ttmi import ops
    enc, flat, x, cot = _layer(monkeypatch)
    base_g, base_dx = _step(enc, flat, x, cot, 3)
    sizes = []
    orig = ops.WgradQueue.maybe_flush
    monkeypatch.setattr(ops.WgradQueue, "maybe_flush", lambda self, force=False: (sizes.append(len(self.descs)) if force and self.descs else None, orig(self, force))[1])
    flat.enable_grouped_wgrads(4, immediate_first_layer=True)
    try:
        g1, dx1 = _step(enc, flat, x, cot, 3)
    finally:
        flat.disable_grouped_wgrads()
    assert sizes == [8]
    assert torch.equal(dx1, base_dx) and rel_err(g1.cpu().numpy(), base_g.cpu().numpy()) < 2e-5


def test_label_encoder_on_its_side_stream_gets_its_own_lane(monkeypatch):
    """a whole Transducer whose label encoder is large enough to be deferred too (B*(U+1) >= 4096 rows) and runs on the side stream: its
    problems are launched on that stream, never together with the audio encoder's"""
    from tt.model import Transducer
    from tt.utils import AttrDict
    from ttmi import ops
    from ttmi.train import FlatModel
    from warprnnt_pytorch import RNNTLoss
    monkeypatch.setenv("TTMI_PRECISION", "bf16")
    side = dict(n_layer=2, d_model=512, n_head=8, d_head=64, d_inner=1024)
    cfg = AttrDict(dict(enc=dict(side, max_input_length=16), dec=dict(side, max_target_length=16),
                        joint=dict(input_size=1024, inner_size=64), vocab_size=29, dropout=0.0, overlap_label_encoder=True))
    torch.manual_seed(8)
    model = Transducer(cfg).cuda().train()
    flat = FlatModel(model)
    B, T, U = 128, 32, 32
    g = torch.Generator(device="cuda").manual_seed(9)
    x = torch.randn(B, T, 512, device="cuda", generator=g)
    y = torch.randint(1, 29, (B, U), device="cuda", generator=g)
    al, ll = torch.full((B,), T, dtype=torch.int32, device="cuda"), torch.full((B,), U, dtype=torch.int32, device="cuda")

    def run():
        flat.zero_grad()
        RNNTLoss()(model(x, y), y.int(), al, ll).backward()
        ops.join_side_streams()
        torch.cuda.synchronize()
        return flat.grad.clone()
    base = run()
    launched = []
    orig = ops.WgradQueue.maybe_flush

    def spy(self, force=False):
        n = len(self.descs)
        orig(self, force)
        if n and not self.descs:
            launched.append((torch.cuda.current_stream().cuda_stream, n))
    monkeypatch.setattr(ops.WgradQueue, "maybe_flush", spy)
    flat.enable_grouped_wgrads(4)
    try:
        got = run()
    finally:
        flat.disable_grouped_wgrads()
    assert sorted(n for _, n in launched) == [8, 8] and launched[0][0] != launched[1][0], launched     # 2 layers x 4 problems per encoder, two streams
    assert rel_err(got.cpu().numpy(), base.cpu().numpy()) < 2e-5
