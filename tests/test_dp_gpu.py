"""2 ranks (gloo, both on cuda:0) through the real HIP model: reduced gradients == single-process global-batch
gradients, and the fused clip+SGD tail matches torch's clip_grad_norm_ + SGD."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg():
    from tt.utils import AttrDict
    side = dict(n_layer=1, d_model=64, n_head=2, d_head=32, d_inner=96)
    return AttrDict(dict(enc=dict(side, max_input_length=16), dec=dict(side, max_target_length=8),
                         joint=dict(input_size=128, inner_size=48), vocab_size=29, dropout=0.0))


def _data():
    g = torch.Generator().manual_seed(5)
    return torch.randn(4, 20, 64, generator=g), torch.randint(1, 29, (4, 6), generator=g)


def _grads(model, x, y, fused=False):
    from warprnnt_pytorch import RNNTLoss
    B = x.shape[0]
    al, ll = torch.full((B,), 20, dtype=torch.int32).cuda(), torch.full((B,), 6, dtype=torch.int32).cuda()
    if fused:       # Transducer.loss: what bench.py times; its backward adds the joint's gradients into the flat buffer and fires the same hooks
        loss = model.loss(x.cuda(), al, y.cuda(), ll, exp_domain=True)          # (tiny lattice: the memory form runs, the Python path is the same)
    else:
        loss = RNNTLoss()(model(x.cuda(), y.cuda()), y.int().cuda(), al, ll)
    loss.backward()
    return loss


def _worker(rank, world, port, q, fused=False):
    for p in (ROOT, os.path.join(ROOT, "transformer-transducer_amd")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tt.model import Transducer
    from ttmi.train import FlatModel, FusedOptimizer, GradSync
    torch.manual_seed(1)
    model = Transducer(_cfg()).cuda()
    flat = FlatModel(model)
    sync = GradSync(flat, bucket_mb=0.05)
    opt = FusedOptimizer(flat, kind="sgd", lr=0.01, momentum=0.9, max_grad_norm=5.0, world=world)
    x, y = _data()
    flat.zero_grad()
    sync.start_step()
    _grads(model, x[rank * 2:(rank + 1) * 2], y[rank * 2:(rank + 1) * 2], fused)
    sync.finish()
    gsum = flat.grad.clone()
    opt.step()
    torch.cuda.synchronize()
    q.put((rank, (gsum / world).cpu().numpy(), flat.flat.cpu().numpy(), float(opt.grad_norm())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fused", [False, True])
def test_two_rank_gradients_and_fused_update(fused):
    world, port = 2, 29100 + os.getpid() % 500 + (7 if fused else 0)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, fused)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    from tt.model import Transducer
    from ttmi.train import FlatModel
    torch.manual_seed(1)
    model = Transducer(_cfg()).cuda()
    flat = FlatModel(model)
    x, y = _data()
    _grads(model, x, y)                                     # global batch of 4, mean reduction
    want = flat.grad.cpu().numpy()
    for rank, got, _, _ in res:
        assert rel_err(got, want) < 1e-5, rank
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    # torch's own tail on the same gradients
    params = [p for p in model.parameters()]
    norm = torch.nn.utils.clip_grad_norm_(params, 5.0)
    torch.optim.SGD(params, lr=0.01, momentum=0.9).step()
    assert abs(float(norm) - res[0][3]) / float(norm) < 1e-5
    assert rel_err(res[0][2], flat.flat.cpu().numpy()) < 1e-6
