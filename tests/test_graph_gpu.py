"""ttmi.train.GraphedStep: the whole training step captured once into a HIP graph and replayed (host work per step).
Replays must walk the same parameter trajectory as eager steps, draw NEW dropout masks on every replay (the seeds themselves are frozen
in the graph: a device word is mixed in at kernel start), and recover from a raised range flag of the exp-domain loss form."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup(monkeypatch, dropout):
    """the step as bench.py drives it, at the smallest sizes at which the exp-domain kernels, the grouped weight gradients and the weight
    shadows are the paths that run (tests/test_dp_nccl_gpu.py)"""
    monkeypatch.setenv("TTMI_PRECISION", "bf16")
    from test_dp_nccl_gpu import _bench_cfg, _bench_data
    from tt.model import Transducer
    from ttmi.train import FlatModel, FusedOptimizer, GradSync
    dev = torch.device("cuda", 0)
    cfg = _bench_cfg()
    cfg["dropout"] = dropout
    torch.manual_seed(1)
    model = Transducer(cfg).to(dev).train()
    flat = FlatModel(model)
    flat.enable_grouped_wgrads()
    flat.enable_shadows()
    sync = GradSync(flat)
    opt = FusedOptimizer(flat, kind="sgd", lr=0.00025, momentum=0.9, max_grad_norm=200.0)
    x, y = _bench_data(0, 0)
    x, y = x.to(dev), y.to(dev)
    il = torch.full((8,), 512, dtype=torch.int32, device=dev)
    tl = torch.full((8,), 7, dtype=torch.int32, device=dev)
    losses = torch.zeros(64, device=dev)
    count = [0]

    def step():
        flat.zero_grad()
        sync.start_step()
        loss = model.loss(x, il, y, tl, exp_domain=True)
        loss.backward()
        sync.finish()
        opt.step()
        return loss.detach()

    return model, flat, opt, step, dev


def _teardown(flat):
    from ttmi import ops
    ops.wgrad_queue = None
    ops.set_dropout_salt(None)
    flat.disable_shadows()


def test_replays_walk_the_eager_trajectory(monkeypatch):
    from ttmi.train import GraphedStep
    torch.manual_seed(11)
    model, flat, opt, step, dev = _setup(monkeypatch, 0.0)
    for _ in range(6):
        step()
    torch.cuda.synchronize()
    want = flat.flat.cpu().numpy().copy()
    _teardown(flat)
    torch.manual_seed(11)
    model, flat, opt, step, dev = _setup(monkeypatch, 0.0)
    g = GraphedStep(step, device=dev, warmup=3, exp_state=model.joint.exp_shift_state(dev), optimizer=opt)
    assert opt.global_step == 1 + 3 and opt.steps_taken == 3     # the capture ran step()'s Python once, not a step: counters restored
    losses = [float(g()) for _ in range(3)]                      # 3 eager warm-up steps + 3 replays = the 6 steps above
    torch.cuda.synchronize()
    got = flat.flat.cpu().numpy()
    assert g.captures == 1 and all(np.isfinite(losses)) and losses[2] < losses[0]
    assert rel_err(got, want) < 5e-6                              # (f32 atomic order in the joint's and the first layer's weight gradients: measured up to 1.9e-6)
    assert opt.global_step == 1 + 6 and opt.steps_taken == 6      # one per executed step (tt/optim.py:8 starts at 1)
    assert float(opt.hyper[1]) == 6.0                             # ... and the device's own count agrees
    # a raised range flag between replays: that replay's step is dropped on the device, the next call is ONE eager step (the plain form
    # re-seeds the shift), the call after it captures again: every batch gets exactly one update
    st = model.joint.exp_shift_state(dev)
    st.cur.fill_(-150.0)
    before = flat.flat.clone()
    bad = float(g())
    torch.cuda.synchronize()
    assert not np.isfinite(bad) and int(st.flag) == 1 and torch.equal(flat.flat, before)       # NaN gradients never reached the weights
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ok = float(g())
    assert np.isfinite(ok) and g.captures == 1 and g.eager_steps == 4 and st.valid
    again = float(g())
    assert np.isfinite(again) and g.captures == 2 and torch.isfinite(flat.flat).all()
    assert opt.global_step == 1 + 9 and opt.steps_taken == 9
    # by-value hyper-parameters are frozen in the graph: changing one is an error until recapture()
    opt.momentum = 0.5
    with pytest.raises(RuntimeError):
        g()
    g.recapture()
    assert np.isfinite(float(g())) and g.captures == 3
    _teardown(flat)


@pytest.mark.parametrize("kind", ["sgd", "adam", "adadelta"])
def test_lr_decay_and_step_count_follow_under_replay(monkeypatch, kind):
    """tt/optim.py:30-33 decay_lr() between replays (train.py:257 calls it every epoch) and Adam's bias-correction step count: the update
    kernels read both from device scalars, so a replayed graph follows them - same trajectory as eager steps with the same schedule"""
    from ttmi.train import GraphedStep
    lr = {"sgd": 0.001, "adam": 0.0005, "adadelta": 1.0}[kind]

    def run(graphed):
        torch.manual_seed(11)
        model, flat, opt, step, dev = _setup(monkeypatch, 0.0)
        opt.__init__(flat, kind=kind, lr=lr, momentum=0.9, max_grad_norm=200.0, decay_ratio=0.5)
        g = GraphedStep(step, device=dev, warmup=2, exp_state=model.joint.exp_shift_state(dev), optimizer=opt) if graphed else None
        if not graphed:
            step(), step()
        lrs = []
        for i in range(6):
            if i in (2, 4):
                opt.decay_lr()                                    # halves opt.lr: the graph must follow
            (g or step)()
            lrs.append(opt.lr)
        torch.cuda.synchronize()
        out = flat.flat.cpu().numpy().copy(), opt.steps_taken, float(opt.hyper[1]), lrs
        _teardown(flat)
        return out

    want, steps_e, dev_e, lrs_e = run(False)
    got, steps_g, dev_g, lrs_g = run(True)
    assert steps_e == steps_g == 8 and dev_e == dev_g == 8.0 and lrs_e == lrs_g and lrs_g[-1] == lr / 4
    # (SGD's update is linear in the gradient: f32 atomic-order noise stays noise.  Adam / Adadelta normalise the update per element, so
    # the same noise moves near-zero gradients' updates by whole steps: two eager runs differ as much)
    assert rel_err(got, want) < (2e-5 if kind == "sgd" else 2e-4)          # (sgd: 7e-6 with round 6's numerics - a ReLU decision at zero between the eager and the replayed run)
    # and the schedule matters: without the decays the parameters end somewhere else (the check above is not vacuous)
    torch.manual_seed(11)
    model, flat, opt, step, dev = _setup(monkeypatch, 0.0)
    opt.__init__(flat, kind=kind, lr=lr, momentum=0.9, max_grad_norm=200.0, decay_ratio=0.5)
    for _ in range(8):
        step()
    torch.cuda.synchronize()
    assert rel_err(flat.flat.cpu().numpy(), want) > 10 * rel_err(got, want)
    _teardown(flat)


def test_every_replay_draws_new_dropout_masks(monkeypatch):
    from ttmi import ops
    from ttmi.train import GraphedStep
    model, flat, opt, step, dev = _setup(monkeypatch, 0.1)
    opt.lr = 0.0                                                  # frozen weights: the loss moves only with the masks
    opt.momentum = 0.0
    g = GraphedStep(step, device=dev, warmup=3, exp_state=model.joint.exp_shift_state(dev))
    losses = [float(g()) for _ in range(4)]
    assert len(set(losses)) == 4, losses                          # same data, same weights, same frozen seeds - different masks
    # the masks follow the salt word and nothing else
    a = ops.dropout_multipliers(4096, 0.3, 77, dev).cpu().numpy()
    g.salt.add_(1)
    b = ops.dropout_multipliers(4096, 0.3, 77, dev).cpu().numpy()
    c = ops.dropout_multipliers(4096, 0.3, 77, dev).cpu().numpy()
    assert np.array_equal(b, c) and not np.array_equal(a, b) and abs((b == 0).mean() - 0.3) < 0.03
    _teardown(flat)
    d = ops.dropout_multipliers(4096, 0.3, 77, dev).cpu().numpy()
    assert not np.array_equal(d, b)                               # salt off again: the seed as passed
