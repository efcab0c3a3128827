"""The RCCL path (torch.distributed backend "nccl" = RCCL over xGMI), one process per GPU: runs only where >= 2 GPUs are visible (the
driver's multi-GPU node; a one-GPU box skips it).  2 ranks on 2 devices through GradSync with the label encoder on its side stream,
small buckets (so buckets mix side-stream and main-stream gradients - the ordering GradSync._reduce exists for), CUs reserved for the
collective kernels during backward; checks: reduced gradients == single-process global-batch gradients, parameters bit-identical on
both ranks after 3 steps, dropout seeds differ per rank (reference: train.py:55-56,214-219 nn.DataParallel semantics)."""
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
    side = dict(n_layer=2, d_model=64, n_head=2, d_head=32, d_inner=96)
    return AttrDict(dict(enc=dict(side, max_input_length=16), dec=dict(side, max_target_length=8),
                         joint=dict(input_size=128, inner_size=48), vocab_size=29, dropout=0.0, overlap_label_encoder=True))


def _data(step):
    g = torch.Generator().manual_seed(50 + step)
    return torch.randn(4, 20, 64, generator=g), torch.randint(1, 29, (4, 6), generator=g)


def _loss(model, x, y, dev):
    from warprnnt_pytorch import RNNTLoss
    B = x.shape[0]
    return RNNTLoss()(model(x.to(dev), y.to(dev)), y.int().to(dev), torch.full((B,), 20, dtype=torch.int32, device=dev),
                      torch.full((B,), 6, dtype=torch.int32, device=dev))


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "transformer-transducer_amd")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from tt import transformer as tr
    from tt.model import Transducer
    from ttmi import ops
    from ttmi.train import FlatModel, FusedOptimizer, GradSync
    torch.manual_seed(1)
    model = Transducer(_cfg()).to(dev).train()
    flat = FlatModel(model)
    sync = GradSync(flat, bucket_mb=0.05)
    assert len(sync.buckets) > 3
    opt = FusedOptimizer(flat, kind="sgd", lr=0.01, momentum=0.9, max_grad_norm=5.0, world=world)
    torch.manual_seed(7)
    seed = tr._new_seed(0.1)                               # same torch seed on both ranks, rank-mixed dropout seed
    first = None
    for step in range(3):
        x, y = _data(step)
        flat.zero_grad()
        sync.start_step()
        loss = _loss(model, x[rank * 2:(rank + 1) * 2], y[rank * 2:(rank + 1) * 2], dev)
        ops.reserve_cus(32)                                 # the collective's kernels run beside backward
        loss.backward()
        sync.finish()
        ops.reserve_cus(0)
        if step == 0:
            torch.cuda.synchronize()
            first = (flat.grad / world).cpu().numpy()
        opt.step()
    torch.cuda.synchronize()
    q.put((rank, first, flat.flat.cpu().numpy(), seed))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="the RCCL test needs two GPUs (one process per GPU)")
def test_two_ranks_over_rccl():
    world, port = 2, 29300 + os.getpid() % 500
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    from tt.model import Transducer
    from ttmi.train import FlatModel
    torch.manual_seed(1)
    model = Transducer(_cfg()).cuda().train()
    flat = FlatModel(model)
    x, y = _data(0)
    _loss(model, x, y, torch.device("cuda", 0)).backward()
    from ttmi import ops
    ops.join_side_streams()
    torch.cuda.synchronize()
    want = flat.grad.cpu().numpy()
    for rank, got, _, _ in res:
        assert rel_err(got, want) < 1e-5, rank
    assert np.array_equal(res[0][1], res[1][1])             # identical reduced gradients ...
    assert np.array_equal(res[0][2], res[1][2])             # ... and identical parameters after three updates
    assert res[0][3] != res[1][3]                           # but different dropout masks


def _solo(port, q):
    for p in (ROOT, os.path.join(ROOT, "transformer-transducer_amd")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", TTMI_PRECISION="bf16")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from tt.model import Transducer
    from ttmi import ops
    from ttmi.train import FlatModel, FusedOptimizer, GradSync
    torch.manual_seed(1)
    model = Transducer(_cfg()).to(dev).train()
    flat = FlatModel(model).enable_shadows()
    sync = GradSync(flat, bucket_mb=0.05, always_reduce=True)          # a one-rank group still launches every bucket's collective
    opt = FusedOptimizer(flat, kind="sgd", lr=0.01, momentum=0.9, max_grad_norm=5.0, world=1)
    n_works = 0
    for step in range(2):
        x, y = _data(step)
        flat.zero_grad()
        sync.start_step()
        loss = _loss(model, x, y, dev)
        ops.reserve_cus(32)
        loss.backward()
        n_works = max(n_works, len(sync.works))
        sync.finish()
        ops.reserve_cus(0)
        if step == 0:
            torch.cuda.synchronize()
            first = flat.grad.cpu().numpy()
        opt.step()
    dist.barrier()
    t = torch.tensor([1.5], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    torch.cuda.synchronize()
    q.put((first, n_works, len(sync.buckets), float(t)))
    dist.destroy_process_group()


def test_single_rank_rccl_path_on_one_gpu(monkeypatch):
    """what a one-GPU box can execute of the RCCL path: a one-rank `nccl` group, every bucket's all-reduce issued from the gradient
    hooks during backward (async handles, stream ordering against the side stream, the barrier and MAX reduction bench.py uses); the
    'reduced' gradients must equal the plain ones"""
    monkeypatch.setenv("TTMI_PRECISION", "bf16")
    port = 29700 + os.getpid() % 200
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_solo, args=(port, q))
    p.start()
    first, n_works, n_buckets, t = q.get(timeout=600)
    p.join(120)
    assert p.exitcode == 0
    assert n_buckets > 3 and n_works >= n_buckets - 1 and t == 1.5            # collectives were launched from the hooks, during backward
    from tt.model import Transducer
    from ttmi import ops
    from ttmi.train import FlatModel
    torch.manual_seed(1)
    model = Transducer(_cfg()).cuda().train()
    flat = FlatModel(model)
    x, y = _data(0)
    _loss(model, x, y, torch.device("cuda", 0)).backward()
    ops.join_side_streams()
    torch.cuda.synchronize()
    mine = flat.grad.cpu().numpy()
    worst = sorted(((rel_err(first[o:o + p.numel()], mine[o:o + p.numel()]), n) for (n, p), o in zip(model.named_parameters(), flat.offsets)), reverse=True)[:4]
    assert rel_err(first, mine) < 1e-5, worst
