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
    g = GraphedStep(step, device=dev, warmup=3, exp_state=model.joint.exp_shift_state(dev), on_replay=(lambda: setattr(opt, "global_step", opt.global_step + 1),))
    losses = [float(g()) for _ in range(3)]                      # 3 eager warm-up steps + 3 replays = the 6 steps above
    torch.cuda.synchronize()
    got = flat.flat.cpu().numpy()
    assert g.captures == 1 and all(np.isfinite(losses)) and losses[2] < losses[0]
    assert rel_err(got, want) < 1e-6                              # (f32 atomic order in the joint's and the first layer's weight gradients)
    assert opt.global_step == 1 + 3 + 1 + 3                       # 3 warm-ups, the captured call, 3 replays (tt/optim.py:8 starts at 1)
    # a raised range flag between replays: one eager recovery (plain form re-seeds the shift), a new capture, and the trajectory goes on
    st = model.joint.exp_shift_state(dev)
    st.cur.fill_(-150.0)
    bad = float(g())
    torch.cuda.synchronize()
    assert not np.isfinite(bad) and int(st.flag) == 1
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ok = float(g())
    assert np.isfinite(ok) and g.captures == 2 and torch.isfinite(flat.flat).all()      # the NaN step was dropped by the optimiser
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
