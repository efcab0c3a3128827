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


# ---------------------------------------------------------------------------------------------------------------------------------
# the step exactly as `bench.py --gpus N` drives it (bf16, Transducer.loss(exp_domain=True), grouped weight gradients with the first
# layer kept immediate, bf16 weight shadows, CUs reserved for the collective during backward), at the smallest sizes at which every one
# of those paths is the one that runs: 2 audio layers of 4096 rows (B=8, T=512), d_model=512, J=1024, V=4334, U=7 (32768 lattice rows)
def _bench_cfg():
    from tt.utils import AttrDict
    side = dict(d_model=512, n_head=8, d_head=64, d_inner=1024)
    return AttrDict(dict(enc=dict(side, n_layer=2, max_input_length=64), dec=dict(side, n_layer=1, max_target_length=8),
                         joint=dict(input_size=1024, inner_size=1024), vocab_size=4334, dropout=0.0, overlap_label_encoder=True))


def _bench_data(step, rank):
    g = torch.Generator().manual_seed(1234 + 10 * step + rank)
    return torch.randn(8, 512, 512, generator=g), torch.randint(1, 4334, (8, 7), generator=g)


def _bench_like(dev, world, rank, steps, hooks, graph=False):
    """-> (gradient buffer after the LAST step's backward, parameters after `steps` updates, number of collectives seen in flight,
    whether the exp-domain kernels ran).  graph: the first two steps eager (GraphedStep's warm-up), the rest replayed from ONE captured
    HIP graph that contains the hook-launched all-reduces, the side-stream fan-in and the CU reservation"""
    from tt.model import Transducer
    from ttmi import ops
    from ttmi.train import FlatModel, FusedOptimizer, GradSync, GraphedStep
    torch.manual_seed(1)
    model = Transducer(_bench_cfg()).to(dev).train()
    flat = FlatModel(model)
    flat.enable_grouped_wgrads(immediate_first_layer=hooks)
    flat.enable_shadows()
    sync = GradSync(flat, bucket_mb=4, always_reduce=hooks)
    opt = FusedOptimizer(flat, kind="sgd", lr=0.00025, momentum=0.9, max_grad_norm=200.0, world=world)
    il = torch.full((8,), 512, dtype=torch.int32, device=dev)
    tl = torch.full((8,), 7, dtype=torch.int32, device=dev)
    assert ops.joint_exp_supported(8, 512, 8, 1024, 4334, 1)
    n_works = [0]
    xs, ys = torch.empty(8, 512, 512, device=dev), torch.empty(8, 7, dtype=torch.long, device=dev)      # static inputs (graph replays read them)
    keep = torch.empty_like(flat.grad)

    def one_step():
        flat.zero_grad()
        sync.start_step()
        loss = model.loss(xs, il, ys, tl, exp_domain=True)
        if hooks:
            ops.reserve_cus(32)
        loss.backward()
        n_works[0] = max(n_works[0], len(sync.works))
        sync.finish()
        if hooks:
            ops.reserve_cus(0)
        keep.copy_(flat.grad)                      # the reduced gradients of this step (before the update)
        opt.step()
        return loss.detach()

    gstep = first_replay = None
    for step in range(steps):
        x, y = _bench_data(step, rank)
        xs.copy_(x)
        ys.copy_(y)
        if graph and step == 0:
            gstep = GraphedStep(one_step, device=dev, warmup=2, exp_state=model.joint.exp_shift_state(dev), optimizer=opt)   # steps 0 and 1 (same batch)
        elif graph:
            if step >= 2:
                gstep()
                if step == 2:
                    torch.cuda.synchronize()
                    first_replay = keep.cpu().numpy()
        else:
            one_step()
            if step == 0:
                torch.cuda.synchronize()
                _bench_like.first_step_grad = keep.cpu().numpy()       # the gradient AT the common state (see below)
    torch.cuda.synchronize()
    # graph runs report the gradient of their FIRST replay: bf16 training is chaotic from run to run (f32 atomic-order noise of 1e-7 flips
    # bf16 roundings downstream and reaches 1e-3 after two more steps, 1e-2 after three: tools/debug/eager_repeat3.py), so gradients are
    # compared one step after the common state, parameters at the end
    grad = first_replay if graph else keep.cpu().numpy()
    st = model.joint.exp_shift_state(dev)
    assert int(st.flag) == 0 and torch.isfinite(flat.flat).all()
    if gstep is not None:
        assert gstep.captures == 1 and opt.steps_taken == steps and float(opt.hyper[1]) == steps
        ops.set_dropout_salt(None)
    ops.wgrad_queue = None
    flat.disable_shadows()
    return grad, flat.flat.cpu().numpy(), n_works[0], len(st.pending) > 0 or st.valid, len(sync.buckets)


def _label_encoder_entries(n):
    """-> bool [n]: which entries of the flat gradient buffer of `_bench_cfg()`'s model belong to the label encoder"""
    from tt.model import Transducer
    from ttmi.train import FlatModel
    probe = Transducer(_bench_cfg()).cuda()
    label = np.zeros(n, dtype=bool)
    for (name, prm), off in zip(probe.named_parameters(), FlatModel(probe).offsets):
        if name.startswith("decoder."):
            label[off:off + prm.numel()] = True
    assert label.any() and not label.all()
    return label


def _solo_bench(port, q):
    for p in (ROOT, os.path.join(ROOT, "transformer-transducer_amd")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", TTMI_PRECISION="bf16", NCCL_MAX_NCHANNELS="32")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    out = _bench_like(dev, 1, 0, 3, hooks=True)
    dist.barrier()
    dist.destroy_process_group()
    q.put(out + (_bench_like.first_step_grad,))


def test_single_rank_rccl_bench_step_on_one_gpu(monkeypatch):
    """the RCCL-side step of bench.py, as far as one GPU can run it: every bucket's all-reduce launched from the gradient hooks (incl.
    the deferred hooks of the grouped weight gradients) beside a backward pass whose persistent GEMMs leave the reserved CUs free,
    against the same three steps without hooks, reservation or process group: same gradients, same parameters."""
    monkeypatch.setenv("TTMI_PRECISION", "bf16")
    port = 29900 + os.getpid() % 90
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_solo_bench, args=(port, q))
    p.start()
    grad, params, n_works, exp_ran, n_buckets, grad1 = q.get(timeout=900)
    p.join(120)
    assert p.exitcode == 0
    assert exp_ran and n_buckets >= 4 and n_works >= n_buckets - 1
    want_grad, want_params, _, _, _ = _bench_like(torch.device("cuda", 0), 1, 0, 3, hooks=False)
    want_grad1 = _bench_like.first_step_grad
    # grouped weight gradients are bit-reproducible; the joint's sums over frames and the first layer's K-split sums land in f32 atomic order.  AT the common
    # state the runs' gradients agree to 1e-4: 1e-7 everywhere except the label encoder, where the order noise of dPD (32 atomic partial sums per label
    # state) flips the bf16 rounding of one or two of its 65536 entries on the way into the d(label states) GEMM - one 512-wide row of that gradient then
    # differs by up to 1e-5 of the largest entry, and the label encoder's 64-row bias sums by 2.3e-5 (tools/debug/step0_repro.py; every configuration
    # has this, profiles/r06_step_reproducibility.log).
    # AFTER an update: this fixed-seed test's learning rate is far above the label encoder's stability limit (its gradients' run-to-run distance grows
    # 45-fold per step: 2e-5 -> 1e-3 -> 4.6e-2 in dec_attn.layer_norm.bias, tools/debug/step2_repro.py), and ONE FFN unit of the audio encoder whose
    # pre-activation sits within the runs' 1e-8 parameter distance of zero may be decided differently (worth 1 / sqrt(#units) = 3.5e-4 ... 1e-3 of the
    # gradient; test_c2_full_model_fp32_end_to_end documents the same effect against the oracle).  So two updates later the audio encoder's and the joint's
    # gradients are held to 5e-3, the label encoder's 64-row sums to a sanity bound, and the parameters - which integrate all three gradients - to 1e-5
    label = _label_encoder_entries(grad.size)
    later = rel_err(grad[~label], want_grad[~label]), rel_err(grad[label], want_grad[label])
    print("RCCL-side step vs plain: gradients at the common state %.2e, two updates later %.2e (label encoder %.2e), parameters %.2e"
          % (rel_err(grad1, want_grad1), later[0], later[1], rel_err(params, want_params)))
    assert rel_err(grad1, want_grad1) < 1e-4
    assert later[0] < 5e-3 and later[1] < 0.3
    assert rel_err(params, want_params) < 1e-5


def _solo_bench_graph(port, q):
    for p in (ROOT, os.path.join(ROOT, "transformer-transducer_amd")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", TTMI_PRECISION="bf16", NCCL_MAX_NCHANNELS="32")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    out = _bench_like(dev, 1, 0, 5, hooks=True, graph=True)
    dist.barrier()
    dist.destroy_process_group()
    q.put(out)


def test_single_rank_rccl_graphed_step_on_one_gpu(monkeypatch):
    """ttmi.train.GraphedStep around the RCCL-side step: the captured graph contains GradSync's hook-launched
    all-reduces (one-rank `nccl` group, every bucket's collective issued), the wait_stream fan-in of the label encoder's side stream and
    the CU reservation; 2 eager warm-up steps + 3 replays against five eager steps without hooks or process group: same reduced
    gradients, same parameters."""
    monkeypatch.setenv("TTMI_PRECISION", "bf16")
    port = 29800 + os.getpid() % 90
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_solo_bench_graph, args=(port, q))
    p.start()
    grad, params, n_works, exp_ran, n_buckets = q.get(timeout=900)
    p.join(120)
    assert p.exitcode == 0
    assert exp_ran and n_buckets >= 4 and n_works >= n_buckets - 1
    want_grad, want_params, _, _, _ = _graph_twin(torch.device("cuda", 0), 1, 0, 5)
    print("graphed RCCL step vs eager twin: gradients of the first replay %.2e, parameters after 5 steps %.2e" % (rel_err(grad, want_grad), rel_err(params, want_params)))
    label = _label_encoder_entries(grad.size)
    assert rel_err(grad[~label], want_grad[~label]) < 5e-3     # (7e-7 ... 2e-6: f32 atomic order; 3e-3 when a ReLU unit at zero is decided differently after the two eager warm-up steps, see above)
    assert rel_err(grad[label], want_grad[label]) < 0.3        # (the label encoder's 64-row sums two updates after the common state: 1e-6 ... 5e-2 between two eager runs, see above)
    assert rel_err(params, want_params) < 1e-5                 # (two more steps later: measured 3e-10 .. 1.5e-6, as between two eager runs)


def _graph_twin(dev, world, rank, steps):
    """the eager run a graphed `_bench_like(..., steps, graph=True)` must reproduce: GraphedStep's two warm-up steps both see batch 0 and
    batch 1 is never used, then batches 2 .. steps-1"""
    from tt.model import Transducer
    from ttmi import ops
    from ttmi.train import FlatModel, FusedOptimizer, GradSync
    torch.manual_seed(1)
    model = Transducer(_bench_cfg()).to(dev).train()
    flat = FlatModel(model)
    flat.enable_grouped_wgrads()
    flat.enable_shadows()
    sync = GradSync(flat, bucket_mb=4)
    opt = FusedOptimizer(flat, kind="sgd", lr=0.00025, momentum=0.9, max_grad_norm=200.0, world=world)
    il = torch.full((8,), 512, dtype=torch.int32, device=dev)
    tl = torch.full((8,), 7, dtype=torch.int32, device=dev)
    grad = None
    for i, b in enumerate([0, 0] + list(range(2, steps))):
        x, y = _bench_data(b, rank)
        flat.zero_grad()
        sync.start_step()
        loss = model.loss(x.to(dev), il, y.to(dev), tl, exp_domain=True)
        loss.backward()
        sync.finish()
        if i == 2:
            grad = flat.grad.clone()        # the step a graphed run replays first
        opt.step()
    torch.cuda.synchronize()
    ops.wgrad_queue = None
    flat.disable_shadows()
    return grad.cpu().numpy(), flat.flat.cpu().numpy(), 0, True, len(sync.buckets)


def _worker_bench_graph(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "transformer-transducer_amd")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", TTMI_PRECISION="bf16", NCCL_MAX_NCHANNELS="32")
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    grad, params, n_works, exp_ran, n_buckets = _bench_like(dev, world, rank, 5, hooks=True, graph=True)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, grad, params, n_works, exp_ran))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="the RCCL test needs two GPUs (one process per GPU)")
def test_two_ranks_over_rccl_graphed_step():
    """two ranks replaying captured steps whose graphs contain the bucketed all-reduces: bit-identical reduced gradients and parameters
    on both ranks after 2 eager + 3 replayed steps"""
    world, port = 2, 30500 + os.getpid() % 400
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_bench_graph, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=900) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(r[4] for r in res)
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    assert np.isfinite(res[0][2]).all()


def _worker_bench(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "transformer-transducer_amd")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", TTMI_PRECISION="bf16", NCCL_MAX_NCHANNELS="32")
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    grad, params, n_works, exp_ran, n_buckets = _bench_like(dev, world, rank, 3, hooks=True)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, grad, params, n_works, exp_ran))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="the RCCL test needs two GPUs (one process per GPU)")
def test_two_ranks_over_rccl_bench_step():
    """two ranks, two devices, different utterances: after three bench-like steps both ranks hold bit-identical reduced gradients and
    parameters (the all-reduce delivers the same sums everywhere; clip / SGD are deterministic), and they differ from a one-rank run"""
    world, port = 2, 30100 + os.getpid() % 400
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_bench, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=900) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(r[4] for r in res) and all(r[3] >= 3 for r in res)
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    assert np.isfinite(res[0][2]).all()
