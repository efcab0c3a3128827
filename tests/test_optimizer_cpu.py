"""Host logic of ttmi.train.FusedOptimizer that needs no GPU: the reference wrapper's counters (tt/optim.py:8-33) and a state_dict in
torch.optim's own layout, loadable in both directions."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(7, 9), torch.nn.Linear(9, 3))


@pytest.mark.parametrize("kind", ["sgd", "adam"])
def test_state_dict_round_trips_through_torch_optim(kind):
    from ttmi.train import FlatModel, FusedOptimizer
    m = _model()
    flat = FlatModel(m)
    opt = FusedOptimizer(flat, kind=kind, lr=0.01, momentum=0.9)
    assert opt.state_dict()["state"] == {}                  # nothing before the first step, like torch
    # drive torch's optimizer for two steps, load its state here, hand it back
    t = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9) if kind == "sgd" else torch.optim.Adam(m.parameters(), lr=0.01, betas=(0.9, 0.98))
    for s in range(2):
        for p in m.parameters():
            p.grad.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(s)))
        t.step()
    opt.load_state_dict(t.state_dict())
    assert opt.steps_taken == (2 if kind == "adam" else 1)
    sd = opt.state_dict()
    tsd = t.state_dict()
    assert sd["param_groups"][0]["params"] == tsd["param_groups"][0]["params"] == [0, 1, 2, 3]
    for i, ent in tsd["state"].items():
        for k, v in ent.items():
            if k != "step":
                assert torch.equal(sd["state"][i][k], v), (i, k)
    t2 = torch.optim.SGD(m.parameters(), lr=1.0, momentum=0.5) if kind == "sgd" else torch.optim.Adam(m.parameters(), lr=1.0)
    t2.load_state_dict(sd)
    assert t2.param_groups[0]["lr"] == 0.01
    # parameters are still views of the flat buffer and the views of the state line up with them
    for p, o in zip(flat.params, flat.offsets):
        assert p.data_ptr() == flat.flat.data_ptr() + 4 * o


def test_counters_and_decay_follow_the_reference_wrapper():
    from ttmi.train import FlatModel, FusedOptimizer
    opt = FusedOptimizer(FlatModel(_model()), kind="sgd", lr=0.0002, decay_ratio=0.5)
    assert opt.global_step == 1 and opt.current_epoch == 0      # tt/optim.py:8-9
    opt.epoch()
    opt.decay_lr()
    assert opt.current_epoch == 1 and opt.lr == 0.0001          # tt/optim.py:17-18,30-33
    assert opt.state_dict()["param_groups"][0]["lr"] == 0.0001
    with pytest.raises(ValueError):
        opt.load_state_dict({"state": {}, "param_groups": [{"lr": 1.0, "params": [0]}]})


def test_step_refuses_detached_gradients():
    from ttmi.train import FlatModel, FusedOptimizer
    m = _model()
    opt = FusedOptimizer(FlatModel(m), kind="sgd")
    m.zero_grad(set_to_none=True)
    with pytest.raises(RuntimeError):
        opt.step()
