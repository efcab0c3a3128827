"""Training-step tail and checkpointing: the fused clip + SGD / Adam kernels on the flat buffers against
torch.optim over several steps (so the momentum / moment recurrences are exercised, not just their first step), epoch decay
(tt/optim.py:30-33), and save -> reload -> continue in the reference's `.chkpt` layout (tt/utils.py:80-91, train.py:196-212,234-241)
bit-identical to an uninterrupted run."""
import copy
import os

import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _cfg(dropout=0.0):
    from tt.utils import AttrDict
    side = dict(n_layer=1, d_model=64, n_head=2, d_head=32, d_inner=96)
    return AttrDict(dict(enc=dict(side, max_input_length=16), dec=dict(side, max_target_length=8),
                         joint=dict(input_size=128, inner_size=48), vocab_size=29, dropout=dropout))


def _batch(step):
    g = torch.Generator().manual_seed(100 + step)
    return torch.randn(3, 20, 64, generator=g).cuda(), torch.randint(1, 29, (3, 6), generator=g).cuda()


def _loss(model, x, y):
    from warprnnt_pytorch import RNNTLoss
    B = x.shape[0]
    return RNNTLoss()(model(x, y), y.int(), torch.full((B,), 20, dtype=torch.int32).cuda(), torch.full((B,), 6, dtype=torch.int32).cuda())


@pytest.mark.parametrize("kind,clip", [("sgd", 5.0), ("sgd", 0.0), ("adam", 5.0), ("adam", 0.0), ("adadelta", 5.0), ("adadelta", 0.0)])
def test_fused_update_matches_torch_over_five_steps(kind, clip):
    """same gradients fed to both optimizers for five steps: torch.nn.utils.clip_grad_norm_ + torch.optim.{SGD(momentum .9),
    Adam(betas (.9, .98), eps 1e-8)} (tt/optim.py:57-73) against ttmi_sumsq + ttmi_sgd_step / ttmi_adam_step"""
    from ttmi.train import FlatModel, FusedOptimizer
    torch.manual_seed(3)
    ref = torch.nn.Sequential(torch.nn.Linear(37, 50), torch.nn.Linear(50, 11)).cuda()
    mine = copy.deepcopy(ref)
    flat = FlatModel(mine)
    lr = {"sgd": 0.05, "adam": 0.01, "adadelta": 1.0}[kind]
    opt = FusedOptimizer(flat, kind=kind, lr=lr, momentum=0.9, weight_decay=1e-3, max_grad_norm=clip, rho=0.95)
    topt = {"sgd": lambda: torch.optim.SGD(ref.parameters(), lr=lr, momentum=0.9, weight_decay=1e-3),
            "adam": lambda: torch.optim.Adam(ref.parameters(), lr=lr, betas=(0.9, 0.98), eps=1e-8, weight_decay=1e-3),
            "adadelta": lambda: torch.optim.Adadelta(ref.parameters(), lr=lr, rho=0.95, eps=1e-6, weight_decay=1e-3)}[kind]()
    g = torch.Generator(device="cuda").manual_seed(1)
    for step in range(5):
        for pr, pm in zip(ref.parameters(), mine.parameters()):
            gr = torch.randn(pr.shape, device="cuda", generator=g) * (3.0 if step % 2 else 0.3)     # some steps clip, some do not
            pr.grad = gr.clone()
            pm.grad.copy_(gr)
        if clip:
            norm = torch.nn.utils.clip_grad_norm_(ref.parameters(), clip)
        topt.step()
        opt.step()
        if clip:
            assert abs(float(opt.grad_norm()) - float(norm)) / float(norm) < 1e-5
        for pr, pm in zip(ref.parameters(), mine.parameters()):
            assert rel_err(pm.detach().cpu().numpy(), pr.detach().cpu().numpy()) < 2e-6, (kind, step)
    # and the states agree, in torch's own layout
    sd, tsd = opt.state_dict(), topt.state_dict()
    for i in tsd["state"]:
        for k, v in tsd["state"][i].items():
            if k == "step":
                assert float(sd["state"][i][k]) == float(v)
            else:
                assert rel_err(sd["state"][i][k].cpu().numpy(), v.cpu().numpy()) < 2e-6, k


@pytest.mark.parametrize("kind", ["sgd", "adam"])
def test_checkpoint_resume_is_bit_identical(kind, tmp_path):
    """3 steps, save in the reference's checkpoint layout, 3 more steps; a fresh model + optimizer restored from the file and run for
    the same 3 steps ends with the same bits in every parameter and optimizer state (the HIP model, dropout off: gradients are
    deterministic functions of the weights apart from atomic-order noise, so the batch is fixed and the comparison is on the update)"""
    from tt.model import Transducer
    from ttmi.train import FlatModel, FusedOptimizer, load_checkpoint, save_checkpoint

    def make():
        torch.manual_seed(1)
        model = Transducer(_cfg()).cuda().train()
        flat = FlatModel(model)
        return model, flat, FusedOptimizer(flat, kind=kind, lr=0.01, momentum=0.9, max_grad_norm=5.0, decay_ratio=0.5)

    grads = []                                              # gradients are recorded once and replayed, so both runs see the same bits

    def run(model, flat, opt, steps, record):
        for s in steps:
            flat.zero_grad()
            if record:
                _loss(model, *_batch(s)).backward()
                grads.append(flat.grad.clone())
            else:
                flat.grad.copy_(grads[s])
            opt.step()
            if s == 3:
                opt.epoch()
                opt.decay_lr()

    model, flat, opt = make()
    run(model, flat, opt, range(3), True)
    path = str(tmp_path / "tt.epoch0.chkpt")
    save_checkpoint(model, opt, path)
    run(model, flat, opt, range(3, 6), True)
    ck = torch.load(path)
    assert set(ck) == {"encoder", "decoder", "joint", "optimizer", "epoch", "step"} and ck["step"] == 4 and ck["epoch"] == 0
    assert "layers.0.MultiHeadAttention.dec_attn.qkv_net.weight" in ck["encoder"] and "dec_embedding.weight" in ck["decoder"]

    model2, flat2, opt2 = make()
    with torch.no_grad():
        flat2.flat.add_(1.0)                                # prove the weights come from the file
    load_checkpoint(model2, opt2, path, mode="continue")
    assert opt2.global_step == 4 and opt2.current_epoch == 0
    assert all(p.data_ptr() == flat2.flat.data_ptr() + 4 * o for p, o in zip(flat2.params, flat2.offsets))     # still views
    run(model2, flat2, opt2, range(3, 6), False)
    assert opt2.lr == opt.lr == 0.005 and opt2.global_step == opt.global_step == 7 and opt2.current_epoch == 1
    for a, b in zip(flat2.params, flat.params):              # (the flat buffers' alignment padding is not state)
        assert torch.equal(a, b)
    for s2, s1 in zip(opt2.state, opt.state):
        for a, b in zip(opt2._views(s2), opt._views(s1)):
            assert torch.equal(a, b)


def test_reference_shaped_optimizer_wrapper():
    """tt.optim.Optimizer(model.parameters(), config.optim) with the reference's surface and YAML keys (config/aishell.yaml optim
    section): used the way train.py uses it - zero_grad, backward, clip_grad_norm_, step, epoch, decay_lr - against torch.optim.SGD"""
    from tt.model import Transducer
    from tt.optim import Optimizer
    from tt.utils import AttrDict
    ocfg = AttrDict(dict(type="sgd", lr=0.01, momentum=0.9, decay_ratio=0.5, weight_decay=0, begin_to_adjust_lr=60, nesterov=None))
    torch.manual_seed(1)
    model = Transducer(_cfg()).cuda().train()
    ref = Transducer(_cfg()).cuda().train()
    ref.load_state_dict(model.state_dict())
    opt = Optimizer(model.parameters(), ocfg)
    topt = torch.optim.SGD(ref.parameters(), lr=0.01, momentum=0.9)
    assert opt.global_step == 1 and opt.current_epoch == 0 and opt.lr == 0.01
    for s in range(3):
        for m, o in ((model, opt), (ref, topt)):
            o.zero_grad()
            _loss(m, *_batch(s)).backward()
            torch.nn.utils.clip_grad_norm_(m.parameters(), 200.0)           # train.py:62-63
            o.step()
    assert opt.global_step == 4
    for (n, a), b in zip(model.named_parameters(), ref.parameters()):
        assert rel_err(a.detach().cpu().numpy(), b.detach().cpu().numpy()) < 1e-5, n
    opt.epoch()
    opt.decay_lr()
    assert opt.current_epoch == 1 and opt.lr == 0.005 and opt.param_groups[0]["lr"] == 0.005
    topt.load_state_dict(opt.state_dict())                  # torch.optim accepts our state_dict as it stands
    with pytest.raises(NotImplementedError):
        Optimizer(model.parameters(), AttrDict(dict(type="rmsprop", lr=1.0)))      # build_optimizer's own else-branch (tt/optim.py:83-84)
    # the third type the reference builds (tt/optim.py:74-81): adadelta through the same wrapper against torch.optim.Adadelta
    torch.manual_seed(5)
    m2 = Transducer(_cfg()).cuda().train()
    r2 = Transducer(_cfg()).cuda().train()
    r2.load_state_dict(m2.state_dict())
    o2 = Optimizer(m2.parameters(), AttrDict(dict(type="adadelta", lr=1.0, rho=0.9, eps=1e-6, weight_decay=0.0)))
    t2 = torch.optim.Adadelta(r2.parameters(), lr=1.0, rho=0.9, eps=1e-6, weight_decay=0.0)
    for s in range(3):
        for m, o in ((m2, o2), (r2, t2)):
            o.zero_grad()
            _loss(m, *_batch(s)).backward()
            o.step()
    for (n, a), b in zip(m2.named_parameters(), r2.parameters()):
        assert rel_err(a.detach().cpu().numpy(), b.detach().cpu().numpy()) < 1e-5, n
    t2.load_state_dict(o2.state_dict())


def test_embedding_index_contract():
    """nn.Embedding (tt/decoder.py:26) takes int32 or int64 ids; anything else raises, and an id outside [0, V) does not pass as
    some other row"""
    from ttmi import ops
    W = torch.randn(11, 8, device="cuda")
    ids = torch.tensor([[1, 5, 10]], device="cuda")
    a = ops.embed_fwd(ids, W)
    assert torch.equal(a, W[ids]) and torch.equal(ops.embed_fwd(ids.int(), W), a)
    with pytest.raises(TypeError):
        ops.embed_fwd(ids.float(), W)
    bad = ops.embed_fwd(torch.tensor([[1, 11, -1]], device="cuda"), W)
    assert torch.isnan(bad[0, 1]).all() and torch.isnan(bad[0, 2]).all() and torch.equal(bad[0, 0], W[1])


def test_bf16_weight_shadows_are_transparent(monkeypatch):
    """FlatModel.enable_shadows(): the library takes its bf16 weight copies from buffers rebuilt once per optimiser step instead of
    converting in every call.  Same bits as the converting path: on the first step, after fused updates, after an in-place change made
    through torch (caught by the version check), after an explicit refresh_shadows() following a write through .data, and back to the
    converting path once disabled."""
    from tt.model import Transducer
    from tt.utils import AttrDict
    from ttmi.train import FlatModel, FusedOptimizer
    monkeypatch.setenv("TTMI_PRECISION", "bf16")
    side = dict(n_layer=2, d_model=64, n_head=2, d_head=32, d_inner=96)
    cfg = AttrDict(dict(enc=dict(side, max_input_length=16), dec=dict(side, max_target_length=8),
                        joint=dict(input_size=128, inner_size=72), vocab_size=301, dropout=0.0))     # V = 301: the joint's transposed shadow is padded

    def make(shadows):
        torch.manual_seed(4)
        model = Transducer(cfg).cuda().train()
        flat = FlatModel(model)
        if shadows:
            flat.enable_shadows()
            assert flat.shadow["n"] >= 2 * 4 + 2 * 4 + 2        # qkv / o / W1 / W2 of every layer + the joint's two matrices
        return model, flat, FusedOptimizer(flat, kind="sgd", lr=0.05, momentum=0.9, max_grad_norm=5.0)

    def step(model, flat, opt, s, update=True):
        flat.zero_grad()
        loss = _loss_v(model, s)
        loss.backward()
        g = flat.grad.clone()
        if update:
            opt.step()
        return loss.detach().clone(), g

    def _loss_v(model, s):
        from warprnnt_pytorch import RNNTLoss
        g = torch.Generator().manual_seed(100 + s)
        x, y = torch.randn(3, 20, 64, generator=g).cuda(), torch.randint(1, 301, (3, 6), generator=g).cuda()
        return RNNTLoss()(model(x, y), y.int(), torch.full((3,), 20, dtype=torch.int32).cuda(), torch.full((3,), 6, dtype=torch.int32).cuda())

    a, b = make(False), make(True)
    for s in range(3):
        la, ga = step(*a, s)
        lb, gb = step(*b, s)
        assert torch.equal(la, lb), s
        assert rel_err(gb.cpu().numpy(), ga.cpu().numpy()) < 1e-5, s                 # (f32 atomic order is the only difference)
        # ... and it moves the two models 1e-9 apart per update.  Pure bf16 arithmetic used to round that away; the label states' value now comes from
        # the bf16x3 pass (tt.model._label_states), a smooth function of the parameters, and shows it in the loss's last digits (131.7448 / 131.7449 at
        # s = 1).  The claim here is "same parameters => same bits with or without shadows": every step starts from the SAME parameters and momentum
        with torch.no_grad():
            b[1].flat.copy_(a[1].flat)
            for sb, sa in zip(b[2].state, a[2].state):
                sb.copy_(sa)
        b[1].refresh_shadows()
    with torch.no_grad():
        for m in (a[0], b[0]):
            m.encoder.layers[0].MultiHeadAttention.pos_ff.CoreNet[0].weight.mul_(1.5)  # through torch: version counter moves
    la, _ = step(*a, 7, update=False)
    lb, _ = step(*b, 7, update=False)
    assert torch.equal(la, lb)
    for m in (a[0], b[0]):
        m.joint.project_layer.weight.data.mul_(0.5)                                     # through .data: invisible to the version check
    b[1].refresh_shadows()
    la, _ = step(*a, 8, update=False)
    lb, _ = step(*b, 8, update=False)
    assert torch.equal(la, lb)
    b[1].disable_shadows()
    lb2, _ = step(*b, 8, update=False)
    assert torch.equal(lb2, lb)


def test_training_loop_as_train_py_drives_it(tmp_path):
    """the call sequence of the reference's train() and main() (train.py:21-65,196-263) against the overlay, names and arguments as there:
    Optimizer(model.parameters(), config.optim), optimizer.epoch(), frequency / time masks on the device batch, model(inputs, targets),
    RNNTLoss()(logits, targets.int(), inputs_length.int(), targets_length.int()), clip_grad_norm_, optimizer.step(), the logging reads
    (global_step, lr, count_parameters), save_model(model, optimizer, config, path), decay_lr, and a restart in 'continue' mode"""
    import random
    from tt.model import Transducer
    from tt.optim import Optimizer
    from tt.utils import AttrDict, count_parameters, frequency_mask_augment, save_model, time_mask_augment
    from warprnnt_pytorch import RNNTLoss
    config = AttrDict(dict(model=dict(_cfg(0.1)), optim=dict(type="sgd", lr=0.001, momentum=0.9, decay_ratio=0.5, weight_decay=0, begin_to_adjust_lr=0,
                                                             nesterov=None, step_wise_update=False),
                           training=dict(num_gpu=1, max_grad_norm=200, epochs=2)))
    torch.manual_seed(2)
    model = Transducer(config.model).cuda()
    n_params, enc, dec = count_parameters(model)
    assert n_params == enc + dec + sum(p.numel() for p in model.joint.parameters())
    optimizer = Optimizer(model.parameters(), config.optim)
    criterion = RNNTLoss()
    g = torch.Generator().manual_seed(0)
    data = [(torch.randn(3, 24, 64, generator=g), torch.tensor([24, 20, 24]), torch.randint(1, 29, (3, 6), generator=g), torch.tensor([6, 6, 4]))
            for _ in range(3)]
    np.random.seed(0)
    random.seed(0)
    losses = []
    for epoch in range(2):
        model.train()
        optimizer.epoch()
        for inputs, inputs_length, targets, targets_length in data:
            max_in, max_tg = inputs_length.max(), targets_length.max()
            inputs, targets = inputs[:, :max_in, :], targets[:, :max_tg]
            inputs, inputs_length = inputs.cuda(), inputs_length.cuda()
            targets, targets_length = targets.cuda(), targets_length.cuda()
            inputs = time_mask_augment(frequency_mask_augment(inputs, max_mask_frequency=5, mask_num=10), max_mask_time=5, mask_num=10)
            optimizer.zero_grad()
            logits = model(inputs, targets)
            loss = criterion(logits, targets.int(), inputs_length.int(), targets_length.int())
            loss.backward()
            losses.append(float(loss))
            grad_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), config.training.max_grad_norm)
            optimizer.step()
            assert np.isfinite(grad_norm.item()) and loss.shape == (1,)
        save_name = str(tmp_path / ("m.epoch%d.chkpt" % epoch))
        save_model(model, optimizer, config, save_name)
        if epoch >= config.optim.begin_to_adjust_lr:
            optimizer.decay_lr()
    assert optimizer.global_step == 7 and optimizer.current_epoch == 2 and abs(optimizer.lr - 0.00025) < 1e-12
    assert all(np.isfinite(losses))
    # restart: train.py:196-200,234-241
    checkpoint = torch.load(str(tmp_path / "m.epoch1.chkpt"))
    model2 = Transducer(config.model)
    model2.encoder.load_state_dict(checkpoint["encoder"])
    model2.decoder.load_state_dict(checkpoint["decoder"])
    model2.joint.load_state_dict(checkpoint["joint"])
    model2 = model2.cuda()
    optimizer2 = Optimizer(model2.parameters(), config.optim)
    optimizer2.load_state_dict(checkpoint["optimizer"])
    optimizer2.global_step, optimizer2.current_epoch = checkpoint["step"], checkpoint["epoch"]
    assert optimizer2.global_step == 7 and optimizer2.current_epoch == 2
    for a, b in zip(model.parameters(), model2.parameters()):
        assert torch.equal(a, b)
    for a, b in zip(optimizer._views(optimizer.state[0]), optimizer2._views(optimizer2.state[0])):
        assert torch.equal(a, b)


def test_backward_uses_what_the_forward_used_for_its_weights(monkeypatch):
    """the weight source of a forward call (registered shadow, or the copies it made itself) is recorded per context; the backward call
    takes the same one.  Shadows disabled BETWEEN forward and backward (round 3 then read a transposed copy the forward had never
    written), and enabled between them: same gradients as the undisturbed runs."""
    from tt.model import Transducer
    from tt.utils import AttrDict
    from ttmi.train import FlatModel
    from warprnnt_pytorch import RNNTLoss
    monkeypatch.setenv("TTMI_PRECISION", "bf16")
    side = dict(n_layer=2, d_model=64, n_head=2, d_head=32, d_inner=96)
    cfg = AttrDict(dict(enc=dict(side, max_input_length=16), dec=dict(side, max_target_length=8),
                        joint=dict(input_size=128, inner_size=72), vocab_size=301, dropout=0.0))
    g = torch.Generator().manual_seed(5)
    x, y = torch.randn(3, 20, 64, generator=g).cuda(), torch.randint(1, 301, (3, 6), generator=g).cuda()
    il, tl = torch.full((3,), 20, dtype=torch.int32).cuda(), torch.full((3,), 6, dtype=torch.int32).cuda()

    def run(before, between):
        torch.manual_seed(4)
        model = Transducer(cfg).cuda().train()
        flat = FlatModel(model)
        if before:
            flat.enable_shadows()
        loss = RNNTLoss()(model(x, y), y.int(), il, tl)
        if between == "disable":
            flat.disable_shadows()
            torch.empty(1 << 22, device="cuda").fill_(float("nan"))         # whatever the allocator hands out next is poisoned
        elif between == "enable":
            flat.enable_shadows()
        loss.backward()
        torch.cuda.synchronize()
        out = float(loss.detach()), flat.grad.cpu().numpy().copy()
        flat.disable_shadows()
        return out

    base = run(False, None)
    for before, between in ((True, None), (True, "disable"), (False, "enable")):
        got = run(before, between)
        assert got[0] == base[0] and np.isfinite(got[1]).all()
        assert rel_err(got[1], base[1]) < 1e-5, (before, between)
