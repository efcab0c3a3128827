#!/usr/bin/env python3
"""Headline benchmark: utterances/sec of the full Transformer-Transducer training step
(forward + RNN-T loss + backward + gradient all-reduce + clip + optimiser) on synthetic 80-d fbank,
T=500, U=50 (BASELINE.json configs[1] / configs[2]), one process per GPU, weak scaling.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement): metric/value/unit + "roofline" for the
dominant kernel (the joint vocabulary-projection MFMA GEMM, timed live with HIP events on its launch
stream) + "cpu_baseline" (the numpy/C oracle timed on the host cores, N=1 only, bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

MFMA_BF16_PEAK_TFLOPS = 2500.0     # dense, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
MFMA_F32_PEAK_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0               # HBM3E, MI355X_MICROARCH.md
ATTN_BWD_KERNEL = "flash_bwd_rel2_kernel<mask kind>"     # what probe 3 brackets (csrc/attn_flash.hip; option 15 = 1 selects the round-3 kernel)


def c2_config(n_enc=12, n_dec=6):
    """config/aishell.yaml model section with enc.n_layer=12, dec.n_layer=6: 48,222,862 params."""
    from tt.utils import AttrDict
    side = dict(d_inner=1024, n_head=8, d_model=512, d_head=64)
    return AttrDict(dict(type="transducer",
                         enc=dict(side, type="attention", max_input_length=410, left_context=10, right_context=2, n_layer=n_enc),
                         dec=dict(side, type="attention", max_target_length=42, n_layer=n_dec),
                         joint=dict(input_size=1024, inner_size=1024), vocab_size=4334, share_weight=False, dropout=0.1,
                         overlap_label_encoder=os.environ.get("TTMI_BENCH_OVERLAP_LABEL", "1") != "0"))


def c4_config(streaming):
    """config/joint_streaming.yaml model section (BASELINE configs[3]): 18 audio / 2 label layers, d_inner 2048, joint inner
    2048, V=6485 (85.6 M params).  streaming: 'band' = context_mask(left=64,right=0), 'chunk' = 16-frame blocks + 64 left."""
    from tt.utils import AttrDict
    side = dict(d_inner=2048, n_head=8, d_model=512, d_head=64)
    st = dict(left=64, right=0) if streaming == "band" else dict(chunk=16, left=64)
    return AttrDict(dict(type="transducer",
                         enc=dict(side, type="attention", max_input_length=410, left_context=10, right_context=2, n_layer=18),
                         dec=dict(side, type="attention", max_target_length=42, n_layer=2),
                         joint=dict(input_size=1024, inner_size=2048), vocab_size=6485, share_weight=False, dropout=0.1,
                         overlap_label_encoder=os.environ.get("TTMI_BENCH_OVERLAP_LABEL", "1") != "0", streaming=st))


def flops_per_utt(cfg, T, U1):
    def layer(L, c):
        d, H, Dh, Di = c["d_model"], c["n_head"], c["d_head"], c["d_inner"]
        return 2 * L * d * 3 * H * Dh + 2 * L * H * Dh * d + 4 * L * d * Di + 6 * L * L * H * Dh
    d, J, V = cfg["enc"]["d_model"], cfg["joint"]["inner_size"], cfg["vocab_size"]
    fwd = cfg["enc"]["n_layer"] * layer(T, cfg["enc"]) + cfg["dec"]["n_layer"] * layer(U1, cfg["dec"]) \
        + 2 * T * d * J + 2 * U1 * d * J + 2 * T * U1 * J * V
    return 3 * fwd


def cpu_baseline(model, feats, proj, targets, T, U, n_utt, gpu_costs, reps):
    """Time the oracle (numpy model + C lattice) on `n_utt` utterances of the same workload, `reps` times (median); also returns the
    relative error of the GPU's per-utterance costs against it."""
    from oracle import tt_oracle as O
    from oracle.rnnt_c import rnnt_loss_c
    sd = {k: v.detach().float().cpu().numpy() for k, v in model.state_dict().items()}
    x = (feats[:n_utt].float().cpu().numpy() @ proj.cpu().numpy())
    y = targets[:n_utt].cpu().numpy()
    tl = np.full(n_utt, T, dtype=np.int32)
    ul = np.full(n_utt, U, dtype=np.int32)
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        z, cache = O.transducer_fwd(x, y, sd)
        loss, costs, dz = rnnt_loss_c(z, y, tl, ul)
        O.transducer_bwd(dz, cache, sd)
        times.append(time.perf_counter() - t0)
        del z, cache, dz
    dt = float(np.median(times))
    rel = float(np.abs(gpu_costs[:n_utt] - costs).max() / np.abs(costs).max())
    return n_utt / dt, dt, rel, times, costs


def encoder_floor(model, enc_states, dec_states, targets, T, U, n_utt, oracle_costs):
    """per-utterance costs of the ORACLE's joint + lattice (float64 joint, C lattice) on the GPU's encoder states, against the all-oracle
    costs: the part of the loss error that is already in the encoder outputs, whatever the joint and the loss do afterwards"""
    from oracle import tt_oracle as O
    from oracle.rnnt_c import rnnt_loss_c
    sd = {k: v.detach().double().cpu().numpy() for k, v in model.state_dict().items() if k.startswith("joint.")}
    z, _ = O.joint_fwd(enc_states, dec_states, sd)
    y = targets[:n_utt].cpu().numpy()
    costs = rnnt_loss_c(z.astype(np.float32), y, np.full(n_utt, T, dtype=np.int32), np.full(n_utt, U, dtype=np.int32), want_grad=False)[1]
    return float(np.abs(costs.astype(np.float64) - oracle_costs).max() / np.abs(oracle_costs).max())


# ---- multi-rank control flow of the driver contract (covered on CPU by tests/test_bench_flow.py over gloo, world size 2)
def rank_seed(rank):
    """every rank draws its own synthetic utterances: generator seed 1234 + rank; the MODEL seed is the same everywhere"""
    return 1234 + int(rank)


def dp_options(world):
    """what a data-parallel run switches on: the first audio layer keeps its own weight-gradient launches (its gradients are the last of
    the step - nothing is left to overlap a grouped launch's late all-reduce with), and backward leaves 32 CUs to RCCL's kernels"""
    # TTMI_BENCH_RESERVE_CUS: the reservation, settable without a rebuild (round 6: the first 8-GPU run can sweep it together with NCCL_MAX_NCHANNELS; on ONE GPU
    # it prices the reservation itself - profiles/r06_reserve_cus_single_gpu.txt).  The channel count follows it unless NCCL_MAX_NCHANNELS is set.
    env = os.environ.get("TTMI_BENCH_RESERVE_CUS")
    return {"immediate_first_layer": world > 1, "reserve_cus": max(0, int(env)) if env not in (None, "") else (32 if world > 1 else 0)}


def fence(world, cuda=True):
    """barrier over the ranks, then drain the device: both sides of every timed region"""
    if world > 1:
        dist.barrier()
    if cuda:
        torch.cuda.synchronize()


def timed_region(step, steps, warmup, world, cuda=True, spread=None):
    """`warmup` untimed steps, fence, EXACTLY `steps` timed ones, fence -> (elapsed, enqueue wall, host CPU) seconds of this rank.
    spread (a list): receives the per-step GPU times in ms - one event per step on the current stream, read after the closing fence
    (median and p10 / p90 beside the mean)"""
    for _ in range(warmup):
        step(False, 0)
    fence(world, cuda)
    marks = []
    if spread is not None and cuda:
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        marks[0].record()
    t0, c0 = time.perf_counter(), time.process_time()
    last = None
    for i in range(steps):
        last = step(True, i)
        if marks:
            marks[i + 1].record()
    enqueue = time.perf_counter() - t0               # wall time the host spent issuing the work: its own CPU time + time blocked on a full queue
    host_cpu = time.process_time() - c0              # CPU time of this process (all threads, incl. autograd's backward thread) over the same window
    fence(world, cuda)
    elapsed = time.perf_counter() - t0
    if marks:
        spread.extend(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
    return elapsed, enqueue, host_cpu, last


def spread_stats(ms):
    """median / p10 / p90 of per-step times (None when nothing was collected)"""
    if not ms:
        return None
    a = np.asarray(ms, dtype=np.float64)
    return {"median": round(float(np.median(a)), 3), "p10": round(float(np.percentile(a, 10)), 3), "p90": round(float(np.percentile(a, 90)), 3),
            "min": round(float(a.min()), 3), "max": round(float(a.max()), 3), "n": int(a.size)}


def max_over_ranks(values, world, device):
    """the slowest rank's figure for every entry (None entries stay None on every rank)"""
    if world <= 1:
        return list(values)
    t = torch.tensor([v or 0.0 for v in values], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(x) if v is not None else None for x, v in zip(t.tolist(), values)]


def pmc_form(path):
    """the loss form the committed PMC passes were taken with (tools/update_pmc_json.py)"""
    try:
        return json.load(open(path)).get("loss_form", "two-call")
    except (OSError, ValueError):
        return None


def _sha16(path):
    import hashlib
    return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]


def self_launch_command(gpus, argv, port=None):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): the command line of the N-rank job this process starts as a CHILD
    (the driver's own form: torch.distributed.run, one node, 127.0.0.1 rendezvous) - decided right after argparse, before anything touches
    the GPU; the parent only waits and passes the exit code on (never an exec of a process that has initialised HIP)"""
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(gpus, argv):
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    proc = subprocess.run(self_launch_command(gpus, argv), env=env)
    if proc.returncode != 0:
        sys.stderr.write("bench.py: the %d-rank job exited with code %d\n" % (gpus, proc.returncode))
    return proc.returncode


def dp_efficiency(value, world, n1_value):
    """(utt/s at N) / (N x utt/s at 1) when the caller supplies the N = 1 figure; the driver computes its own from its runs"""
    if not n1_value or world < 1:
        return None
    return round(float(value) / (world * float(n1_value)), 4)


def decode_mode(args):
    """single GPU: tools/bench_decode.run (product path) + the oracle's frame-by-frame greedy loop on the host as checker and CPU baseline"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_decode
    prec = "fp32" if "--precision" not in sys.argv else args.precision      # token parity is an fp32 claim
    out, model, inputs, lens, hyps = bench_decode.run(args.batch, args.T, args.emit_rate, prec)      # (default: 32 utterances, BASELINE configs[1]'s batch)
    if not args.no_cpu_baseline:
        from oracle import tt_oracle as O
        n = args.cpu_utts
        sd = {k: v.detach().float().cpu().numpy() for k, v in model.state_dict().items()}
        x = inputs[:n].float().cpu().numpy()
        t0 = time.perf_counter()
        ref = O.recognize(x, lens[:n], sd)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(n / dt, 4), "unit": "utt/s", "cores": _blas_threads(), "kind": "port",
                               "sample": "%d utterance(s), oracle/tt_oracle.py recognize (frame-by-frame loop), %.1f s" % (n, dt)}
        out["tokens_identical_to_oracle"] = [r == h for r, h in zip(ref, hyps)]
        out["oracle_symbols"] = [len(r) for r in ref]
    print(json.dumps(out), flush=True)


def _blas_threads():
    """threads the oracle's BLAS calls actually use (threadpoolctl), else the CPUs this process may run on"""
    try:
        from threadpoolctl import threadpool_info
        n = [int(i.get("num_threads", 0)) for i in threadpool_info() if i.get("user_api") == "blas"]
        if n and max(n) > 0:
            return max(n)
    except Exception:
        pass
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count()


def loss_error_trajectory(dev, inputs, targets, ilen, tlen, states=(0, 2, 5, 8, 10), seed=1, label_value=None):
    """the timed bf16 mode's LOSS (train.py:53's batch mean) against TTMI_PRECISION=fp32 on the same weights ALONG a training trajectory: a fresh C2 model,
    this file's own SGD loop, eval-mode losses of the whole batch at the states after `states` steps (the first ten steps are where the error peaks:
    profiles/r06_loss_error_batch_mean.log, tools/debug/loss_error_batch_mean.py) -> worst |relative error| of the batch mean and of a single utterance"""
    from tt.model import Transducer
    from ttmi.train import FlatModel, FusedOptimizer, GradSync
    keep = os.environ.get("TTMI_PRECISION")
    keep_label = os.environ.get("TTMI_LABEL_VALUE_PRECISION")
    os.environ["TTMI_PRECISION"] = "bf16"
    if label_value:
        os.environ["TTMI_LABEL_VALUE_PRECISION"] = label_value
    torch.manual_seed(seed)
    model = Transducer(c2_config()).to(dev).train()
    flat = FlatModel(model)
    flat.enable_grouped_wgrads()
    flat.enable_shadows()
    sync = GradSync(flat)
    opt = FusedOptimizer(flat, kind="sgd", lr=0.00025, momentum=0.9, max_grad_norm=200.0)
    rows, done = [], 0

    def costs(mode):
        os.environ["TTMI_PRECISION"] = mode
        try:
            with torch.no_grad():
                c = model.loss(inputs, ilen, targets, tlen, reduction="none", exp_domain=(mode == "bf16"), check_lengths=False).double()
            torch.cuda.synchronize()
        finally:
            os.environ["TTMI_PRECISION"] = "bf16"
        return c

    try:
        for s in states:
            while done < s:
                flat.zero_grad()
                sync.start_step()
                model.loss(inputs, ilen, targets, tlen, exp_domain=True).backward()
                sync.finish()
                opt.step()
                done += 1
            model.eval()
            ref, got = costs("fp32"), costs("bf16")
            model.train()
            rows.append((s, float(ref.mean()), float((got.mean() - ref.mean()) / ref.mean()), float(((got - ref).abs() / ref).max())))
    finally:
        if keep is None:
            os.environ.pop("TTMI_PRECISION", None)
        else:
            os.environ["TTMI_PRECISION"] = keep
        if keep_label is None:
            os.environ.pop("TTMI_LABEL_VALUE_PRECISION", None)
        else:
            os.environ["TTMI_LABEL_VALUE_PRECISION"] = keep_label
    del model, flat, sync, opt
    torch.cuda.empty_cache()
    return {"states": [r[0] for r in rows], "loss_fp32": [round(r[1], 2) for r in rows], "batch_mean_rel": [float("%.3e" % r[2]) for r in rows],
            "batch_mean_worst": float("%.3e" % max(abs(r[2]) for r in rows)), "worst_utterance": float("%.3e" % max(r[3] for r in rows)), "utterances": int(inputs.shape[0]),
            "note": "timed form (bf16 encoders, exp-domain joint + loss) against TTMI_PRECISION=fp32 on the same weights, eval mode, along %d SGD steps of a fresh model; "
                    "round 6's defaults (label encoder's value pass in bf16x3, two-term weights in the audio encoder's f32-output GEMMs) keep the batch mean within 8.1e-5 over 56 "
                    "states of four trajectories (2e-4 ... 3e-4 without them: loss_rel_err_trajectory_throughput_form); TTMI_PRECISION=bf16x3 / fp32 hold 1e-4 per utterance" % max(states)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU")
    ap.add_argument("--T", type=int, default=500)
    ap.add_argument("--U", type=int, default=50)
    ap.add_argument("--precision", default=os.environ.get("TTMI_PRECISION", "bf16"), choices=["bf16", "fp32", "bf16x3"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-utts", type=int, default=2, help="utterances of the CPU-baseline sample (BASELINE.md §3: B=2)")
    ap.add_argument("--cpu-reps", type=int, default=3, help="repetitions of the CPU-baseline sample (median reported)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL over xGMI)")
    ap.add_argument("--workload", default="c2", choices=["c2", "c4-band", "c4-chunk", "c5"],
                    help="c2 = BASELINE configs[1] (headline, default); c4-* = streaming model configs[3]; c5 = configs[4] "
                         "long-utterance stress (T=2000 U=200 batch 8)")
    ap.add_argument("--mode", default="train", choices=["train", "decode"],
                    help="train = the metric (default); decode = greedy decode of the C2 model (SURVEY §8 A10) with the oracle's loop beside it")
    ap.add_argument("--no-weight-shadows", action="store_true", help="convert the f32 master weights to bf16 in every call (round-1 behaviour; A/B)")
    ap.add_argument("--no-grouped-wgrads", action="store_true", help="launch every encoder weight-gradient GEMM on its own (K-split, f32 atomics; A/B switch)")
    ap.add_argument("--loss-form", default="auto", choices=["auto", "exp", "fused", "two-call", "two-call-eager"],
                    help="how the step gets its loss.  two-call (= auto): train.py:51-53 as written (logits = model(inputs, targets); criterion(logits, ...)); "
                         "in the bf16 pipeline the logits are a deferred handle (tt.model.DeferredLogits) that RNNTLoss consumes through the fused "
                         "exp-domain kernels.  two-call-eager: the same two calls with TTMI_DEFERRED_LOGITS=0 (the logits are materialised, as in rounds 1-3).  "
                         "exp: Transducer.loss(exp_domain=True), the explicit form of the same kernels.  fused: Transducer.loss, chunked memory form.  "
                         "Whatever runs first, the JSON line also carries the other forms' timings.")
    ap.add_argument("--graph", action="store_true", help="run the PRIMARY timed region through ttmi.train.GraphedStep (one HIP-graph launch per step and rank, "
                                                         "any number of ranks: the graph contains the bucketed all-reduces); the per-kernel probes cannot fire in a replay")
    ap.add_argument("--fused-loss", action="store_true", help="same as --loss-form fused")
    ap.add_argument("--no-two-call", action="store_true", help="skip the secondary timings of the other loss forms (profiling runs: one loss form per trace)")
    ap.add_argument("--no-fp32-form", action="store_true", help="skip the secondary timing of the fp32 mode (the 1e-4 parity path)")
    ap.add_argument("--graph-form", action="store_true", help="multi-rank runs: also time the secondary graph_replay_form (every rank captures the step incl. its all-reduces; "
                    "default on one rank only, see the comment at its call site)")
    ap.add_argument("--no-graph-form", action="store_true", help="skip the secondary timing of the step replayed from a captured HIP graph (ttmi.train.GraphedStep)")
    ap.add_argument("--fp32-steps", type=int, default=5, help="steps of the fp32-mode secondary timing (at most --steps)")
    ap.add_argument("--loss-chunk", type=int, default=0, help="utterances per chunk of the fused loss (0 = default: logits chunk <= 2 GB)")
    ap.add_argument("--emit-rate", type=float, default=0.1, help="decode mode: fraction of frames that emit a symbol (blank bias is set for it)")
    ap.add_argument("--n1-value", type=float, default=float(os.environ.get("TTMI_BENCH_N1_VALUE", 0) or 0),
                    help="utt/s of the N = 1 run of the same workload: the line then carries dp_efficiency = value / (N x n1-value)")
    ap.add_argument("--no-trajectory", action="store_true", help="skip loss_rel_err_trajectory (a fresh model, ten SGD steps, five fp32-mode evaluations: ~4 s)")
    ap.add_argument("--no-sync-form", action="store_true", help="skip the secondary timing of train.py's loop body with its host synchronisations "
                                                                "(fresh .int() length tensors, float(loss) every step: train.py:53,60)")
    args = ap.parse_args()
    if args.mode == "decode":
        return decode_mode(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: start the N ranks ourselves (child processes; nothing here has touched the GPU yet) and pass their exit code on
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    os.environ["TTMI_PRECISION"] = args.precision

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    if ndev < 1:
        sys.exit("bench.py: no GPU visible (the MI355X path has no CPU fallback)")
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if local_world > ndev and not os.environ.get("TTMI_BENCH_SHARE_GPUS"):
        # one rank per GPU is the contract: several ranks on one device would hang in RCCL's rendezvous or measure something else
        sys.exit("bench.py: %d ranks on this node but only %d GPU(s) visible; launch with --nproc-per-node <= %d "
                 "(TTMI_BENCH_SHARE_GPUS=1 lets ranks share devices over the gloo backend for rehearsals)" % (local_world, ndev, ndev))
    local = local % ndev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # the collective's workgroups run beside backward on the CUs the encoder-sized persistent GEMMs leave free (ops.reserve_cus(32) below):
        # keep RCCL's channel count within that reservation unless the caller chose otherwise
        os.environ.setdefault("NCCL_MAX_NCHANNELS", str(max(1, dp_options(world)["reserve_cus"] or 32)))
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    from tt.model import Transducer
    from ttmi import ops
    from ttmi.train import FlatModel, FusedOptimizer, GradSync
    from warprnnt_pytorch import RNNTLoss

    if args.workload == "c5":
        args.T, args.U, args.batch = 2000, 200, 8
    cfg = c2_config() if args.workload in ("c2", "c5") else c4_config(args.workload.split("-")[1])
    torch.manual_seed(1)                                   # config/aishell.yaml:55 - same init on every rank
    model = Transducer(cfg).to(dev).train()
    flat = FlatModel(model)
    if args.precision == "bf16" and not args.no_grouped_wgrads:
        flat.enable_grouped_wgrads(immediate_first_layer=dp_options(world)["immediate_first_layer"])   # encoder weight gradients four layers at a time: one tile per CU, no K-split atomics
    if args.precision == "bf16" and not args.no_weight_shadows:
        flat.enable_shadows()                              # bf16 weight copies rebuilt once per optimiser step (one launch) instead of per call
    sync = GradSync(flat)
    opt = FusedOptimizer(flat, kind="sgd", lr=0.00025, momentum=0.9, max_grad_norm=200.0, world=world)
    criterion = RNNTLoss()

    B, T, U, V, d = args.batch, args.T, args.U, cfg["vocab_size"], cfg["enc"]["d_model"]
    g = torch.Generator(device=dev).manual_seed(rank_seed(rank))
    feats = torch.randn(B, T, 80, device=dev, generator=g)
    proj = torch.randn(80, d, device=dev, generator=torch.Generator(device=dev).manual_seed(7)) / 80 ** 0.5
    targets = torch.randint(1, V, (B, U), device=dev, generator=g)
    ilen = torch.full((B,), T, dtype=torch.int32, device=dev)
    tlen = torch.full((B,), U, dtype=torch.int32, device=dev)
    inputs = torch.empty(B, T, d, device=dev)
    gflags = ops.GEMM_A_KMAJOR | (ops.GEMM_BF16_MFMA if args.precision == "bf16" else 0)
    loss_sum = torch.zeros(1, device=dev)
    host_loss = [0.0]
    probe_ms = []

    for kv in os.environ.get("TTMI_OPTIONS", "").split(","):          # measurement switches, e.g. TTMI_OPTIONS=3:0 (no wgrad fork); see include/ttmi.h
        if kv:
            ops.set_option(*(int(v) for v in kv.replace("=", ":").split(":")))          # "k:v" or "k=v" (the library loader reads the same variable)

    form = "fused" if args.fused_loss and args.loss_form == "auto" else args.loss_form
    if form == "auto":
        form = "two-call"           # train.py's own call sequence; in bf16 mode it runs the fused exp-domain kernels (deferred logits)
    os.environ["TTMI_DEFERRED_LOGITS"] = "0" if form == "two-call-eager" else ""

    reserve = dp_options(world)["reserve_cus"]
    # train.py's loop body word for word needs the lengths as the loader hands them over (int64) and converts them per step (train.py:53:
    # `.int()` makes FRESH tensors, so the shim's length check pays its two device-to-host reads every step) and reads the loss on the
    # host (train.py:60 `float(loss)` for logging): the `train_py_sync_form` secondary times exactly that
    ilen64, tlen64, targets64 = ilen.long(), tlen.long(), targets.long()
    sync_form = [False]

    def step(timed, i=0):
        # harness front-end: fixed 80 -> d_model projection (the reference encoder has no input layer)
        ops.gemm(feats, proj, inputs, B * T, d, 80, 80, d, d, gflags if os.environ["TTMI_PRECISION"] == "bf16" else ops.GEMM_A_KMAJOR)
        flat.zero_grad()
        sync.start_step()
        if timed and rank == 0:
            ops.probe_arm(i % 64)                     # this step's joint-projection launch records into event pair i
        if form in ("exp", "fused"):
            loss = model.loss(inputs, ilen, targets, tlen, chunk=args.loss_chunk or None, exp_domain=form == "exp")
        else:
            logits = model(inputs, targets)                             # train.py:51
            if sync_form[0]:
                loss = criterion(logits, targets64.int(), ilen64.int(), tlen64.int())      # train.py:53 verbatim: fresh int32 tensors, two max() reads
            else:
                loss = criterion(logits, targets.int(), ilen, tlen)     # train.py:53
        if reserve:
            ops.reserve_cus(reserve)                  # gradient all-reduce kernels run beside backward: the persistent encoder GEMMs leave them 4 CUs per XCD (per-stream state)
        loss.backward()
        sync.finish()
        if reserve:
            ops.reserve_cus(0)                        # the next forward pass gets the whole chip (read at launch time)
        opt.step()
        if sync_form[0]:
            host_loss[0] += float(loss)               # train.py:60: the loop's own host read of the loss, every step
        else:
            loss_sum.add_(loss.detach())
        return loss

    def is_exp(f):
        """does loss form `f` run the exp-domain kernels on this workload?"""
        if args.precision != "bf16" or f not in ("exp", "two-call"):
            return False
        c = args.loss_chunk or model.default_loss_chunk(B, T, U + 1, True)
        return ops.joint_exp_supported(c, T, U + 1, cfg["joint"]["inner_size"], V, 1)

    def make_graphed():
        from ttmi.train import GraphedStep
        return GraphedStep(lambda: step(False).detach(), device=dev, warmup=2, optimizer=opt,
                           exp_state=model.joint.exp_shift_state(dev) if is_exp(form) else None)

    step_ms = []
    gstep = None
    primary_retry = None
    if args.graph:
        # the primary region replayed from ONE captured HIP graph per rank (forward, loss, backward, bucketed all-reduces, clip + SGD)
        gstep = make_graphed()
        elapsed, enqueue, host_cpu, last = timed_region(lambda timed, i: gstep(), args.steps, args.warmup, world, spread=step_ms)
        run_step = lambda: gstep()
    else:
        try:
            elapsed, enqueue, host_cpu, last = timed_region(step, args.steps, args.warmup, world, spread=step_ms)
        except Exception as exc:
            # one rank only: a step that raises on the host (round 6 saw ONE such failure in ~45 runs, in a secondary: `layer_bwd: null pointer`) is reported and the WHOLE
            # region - warm-up and all K timed steps - is run again from a drained device; a second failure is fatal.  Several ranks must fail together.
            if world > 1:
                raise
            primary_retry = "%s: %s" % (type(exc).__name__, str(exc).splitlines()[0][:300])
            torch.cuda.synchronize()
            try:
                ops.wgrad_flush()
            except Exception:
                pass
            if step_ms is not None:
                del step_ms[:]
            elapsed, enqueue, host_cpu, last = timed_region(step, args.steps, args.warmup, world, spread=step_ms)
        run_step = lambda: step(False, 0)
    # what the host itself needs to issue one step: timed on a drained device (inside the timed region the issue loop runs ahead of the GPU
    # until the HIP queue is full and then waits in the launch calls - spinning, so that wait shows up as CPU time, not as idle time)
    issue = []
    for _ in range(3):
        fence(world)
        t1 = time.perf_counter()
        run_step()
        issue.append(time.perf_counter() - t1)
    fence(world)
    host_issue = float(np.median(issue))
    probe_ms = loss_ms = attn_ms = wgrad_ms = []
    if rank == 0 and not args.graph:                  # the probes are read after the timed region: no host sync inside it
        slots = [i % 64 for i in range(max(0, args.steps - 64), args.steps)]
        probe_ms = [ops.probe_read_ms(i) for i in slots]
        loss_ms = [(ops.probe_read_ms(i, 1), ops.probe_read_ms(i, 2)) for i in slots]
        attn_ms = [ops.probe_read_ms(i, 3) for i in slots]
        wgrad_ms = [ops.probe_read_ms(i, 4) for i in slots]
    if gstep is not None:
        ops.set_dropout_salt(None)
        del gstep

    def secondary(other_form, precision, steps, warmup, spread=None):
        """the same step with another loss form / precision, same model, data and optimiser state, same barrier + synchronize bracket"""
        nonlocal form
        main_form, form = form, other_form
        main_prec, main_def = os.environ["TTMI_PRECISION"], os.environ.get("TTMI_DEFERRED_LOGITS", "")
        os.environ["TTMI_PRECISION"] = precision
        os.environ["TTMI_DEFERRED_LOGITS"] = "0" if other_form == "two-call-eager" else ""
        try:
            return timed_region(lambda timed, i: step(False), steps, warmup, world, spread=spread)[0]
        except Exception as exc:      # a secondary measurement never takes the finished primary one down (one rank; with several the ranks must fail together)
            if world > 1:
                raise
            secondary_errors["%s/%s" % (other_form, precision)] = "%s: %s" % (type(exc).__name__, str(exc).splitlines()[0][:300])
            torch.cuda.synchronize()
            return None
        finally:
            form = main_form
            os.environ["TTMI_PRECISION"] = main_prec
            os.environ["TTMI_DEFERRED_LOGITS"] = main_def

    secondary_errors = {}
    eager_two_call = explicit_form = fp32_form = graph_form = graph_issue = sync_two_call = x3_form = None
    eager_ms, explicit_ms, graph_ms, sync_ms = [], [], [], []
    fp32_steps = max(1, min(args.steps, args.fp32_steps))
    if form == "two-call" and not args.no_sync_form:
        # (0) train.py:51-65 with its host synchronisations left in: the headline region reuses two int32 length
        # tensors (the shim caches their maxima) and never reads the loss; the unchanged script converts per step and logs float(loss)
        sync_form[0] = True
        try:
            sync_two_call = secondary("two-call", args.precision, args.steps, max(1, args.warmup), sync_ms)
        finally:
            sync_form[0] = False
    if args.precision == "bf16" and not args.no_two_call:
        # (1) the same two calls with the logits MATERIALISED (rounds 1-3's two-call form: 7.1 GB of bf16 logits written, walked twice by the
        # loss, 7.1 GB of gradient written and read), timed BEFORE any graph capture;
        # (2) the explicit spelling of the fused form, Transducer.loss(exp_domain=True): the same kernels as the default region
        if form != "two-call-eager":
            eager_two_call = secondary("two-call-eager", args.precision, args.steps, max(1, args.warmup), eager_ms)
        if form != "exp":
            explicit_form = secondary("exp", args.precision, args.steps, max(1, args.warmup), explicit_ms)
    graph_error = None
    # the same step replayed from ONE captured HIP graph (ttmi.train.GraphedStep): one launch per step instead of ~400 from Python; with several
    # ranks the graph holds the bucketed all-reduces as well.  As a SECONDARY it runs on one rank by default: a capture that fails on some
    # rank of a multi-rank job (a backend whose collectives cannot be captured - gloo's host-staged all-reduce raises inside the capture -, a
    # collective library that objects) would take the finished primary measurement down with it or leave the ranks waiting for each other;
    # `--graph-form` asks for it on N ranks (nccl only), `--graph` makes it the primary region.
    want_graph = world == 1 or (args.graph_form and args.backend == "nccl")
    if not args.graph and form != "two-call-eager" and args.precision == "bf16" and not args.no_graph_form and want_graph:
        try:
            gstep = make_graphed()
            graph_form = timed_region(lambda timed, i: gstep(), args.steps, 1, world, spread=graph_ms)[0]
            fence(world)
            t1 = time.perf_counter()
            gstep()
            graph_issue = time.perf_counter() - t1
            fence(world)
            del gstep
        except Exception as e:      # (one rank: report and carry on - the primary region is already measured)
            if world > 1:
                raise
            graph_form, graph_issue, graph_error = None, None, "%s: %s" % (type(e).__name__, str(e).splitlines()[0][:200])
        ops.set_dropout_salt(None)
    if args.precision == "bf16" and not args.no_fp32_form and args.workload == "c2":
        # the fp32 mode: the path that meets north_star's 1e-4 tolerance (tests/test_configs_gpu.py), timed on the same workload
        fp32_form = secondary("two-call", "fp32", fp32_steps, 1)
        # ... and the QUICK parity mode (round 5): the same f32 data flow with its large dense products on the bf16 MFMA in three terms
        # (tests/test_configs_gpu.py holds it to the same 1e-4 bounds against the float64 oracle)
        x3_steps = max(fp32_steps, min(args.steps, 6))          # (a 100 ms step: six of them and two warm-ups cost under a second)
        x3_form = secondary("two-call", "bf16x3", x3_steps, 2)
    label_form = None
    if args.precision == "bf16" and not args.no_fp32_form and args.workload == "c2" and world == 1 and "TTMI_LABEL_VALUE_PRECISION" not in os.environ:
        # the step WITHOUT round 6's two parity measures (the label encoder's gradient-free value pass in bf16x3, the second bf16 term of the audio encoder's f32-output
        # weights): what the same kernels do when the loss may be 2e-4 ... 3e-4 off during the model's first descent - rounds 1 - 5's headline configuration
        os.environ["TTMI_LABEL_VALUE_PRECISION"] = "off"
        ops.set_option(13, 0)
        try:
            label_form = secondary("two-call", args.precision, args.steps, max(2, args.warmup))
        finally:
            del os.environ["TTMI_LABEL_VALUE_PRECISION"]
            ops.set_option(13, 2)
    elapsed, eager_two_call, explicit_form, fp32_form, graph_form, host_issue, sync_two_call, x3_form = max_over_ranks(
        [elapsed, eager_two_call, explicit_form, fp32_form, graph_form, host_issue, sync_two_call, x3_form], world, dev)

    if rank == 0:
        ms_step = 1e3 * elapsed / args.steps
        utt_s = world * B * args.steps / elapsed
        U1, J = U + 1, cfg["joint"]["inner_size"]
        fused_path = form in ("exp", "fused") or (form == "two-call" and args.precision in ("bf16", "bf16x3"))       # bf16 / bf16x3 two-call: deferred logits -> the fused op
        B_launch = B if not fused_path else (args.loss_chunk or model.default_loss_chunk(B, T, U1, form != "fused"))      # utterances per joint-projection launch
        flop_launch = 2.0 * B_launch * T * U1 * J * V                            # one joint-projection launch
        def mean_pos(vals):
            vals = [v for v in vals if v is not None and v > 0]
            return float(np.mean(vals)) if vals else None

        def rate(work, ms, unit_div, digits=2):
            """work / time in the roofline's unit, or None when the probe never fired (never NaN: the line stays strict JSON)"""
            return None if ms is None else round(work / (ms * 1e-3) / unit_div, digits)

        def frac(a, pk):
            return None if a is None else round(a / pk, 4)

        k_ms = mean_pos(probe_ms)
        # bf16x3: every algorithmic multiply-add is three bf16 MFMA terms (hi.hi + lo.hi + hi.lo), so the ceiling for ALGORITHMIC flops is a third of the bf16 peak
        peak = {"bf16": MFMA_BF16_PEAK_TFLOPS, "bf16x3": round(MFMA_BF16_PEAK_TFLOPS / 3.0, 1)}.get(args.precision, MFMA_F32_PEAK_TFLOPS)
        ach = rate(flop_launch, k_ms, 1e12)
        traffic = None          # HBM-side bytes of one launch from the committed PMC passes (only valid for the default workload)
        mfma_busy = None
        pmc = os.path.join(ROOT, "profiles", "pmc_joint_projection.json")
        pmc_build = None
        # (the PMC passes name the KERNEL form they profiled - "exp" = the exp-store projection -; train.py's call sequence runs those kernels through the deferred handle)
        ran_form = "exp" if is_exp(form) else form
        if os.path.exists(pmc) and args.workload == "c2" and (B, T, U) == (32, 500, 50) and args.precision == "bf16" and ran_form == pmc_form(pmc):
            j = json.load(open(pmc))
            # the counters come from separate rocprofv3 --pmc passes (profiles/): valid only for the kernel source they were measured on
            src = os.path.join(ROOT, "transformer-transducer_amd", "csrc", "gemm_fast.hip")
            pmc_build = {"measured_at_commit": j.get("commit"), "gemm_fast_sha16": j.get("gemm_fast_sha16"), "source": j.get("source")}
            if j.get("gemm_fast_sha16") == _sha16(src):
                traffic = (2.0 * j["fetch_size_kb"] + j["write_size_kb"]) * 1024.0
                mfma_busy = j.get("mfma_busy")       # SQ_VALU_MFMA_BUSY_CYCLES / elapsed SIMD cycles of the same launch
            else:
                pmc_build["stale"] = "csrc/gemm_fast.hip changed since the PMC passes: traffic / mfma_busy withheld"
        # the RNN-T loss op at the API boundary: logits read once, gradient written once, alpha / beta / lp_blank / lp_label in f32
        es = 2 if args.precision == "bf16" else 4
        loss_bytes = B_launch * (2.0 * es * T * U1 * V + 16.0 * T * U1)
        lf = mean_pos([a for a, b in loss_ms if a > 0 and b > 0])
        lb = mean_pos([b for a, b in loss_ms if a > 0 and b > 0])
        l_ms = None if lf is None or lb is None else lf + lb
        loss_gbs = rate(loss_bytes, l_ms, 1e9, 1)
        ran_exp = is_exp(form)              # what the fused op actually selected for these chunks
        roof_joint = {"bound": "mfma", "kernel": "gemm_nt_bf16_v8_kernel (joint vocabulary projection%s, M=%d N=%d K=%d)"
                                                 % (", exp-store epilogue" if ran_exp else "", B_launch * T * U1, V, J),
                      "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": frac(ach, peak),
                      "traffic": traffic, "mfma_busy": mfma_busy, "kernel_ms": None if k_ms is None else round(k_ms, 4), "pmc": pmc_build}
        if k_ms is None:
            roof_joint["reason"] = "probe 0 (joint projection launch) never fired in the timed steps"
        # the two weakest kernels of the step, on the record every round
        enc = cfg["enc"]
        H, Dh, dm = enc["n_head"], enc["d_head"], enc["d_model"]
        a_ms = mean_pos(attn_ms)
        attn_bytes = B * T * H * Dh * 2.0 * 6 + B * H * T * 8.0       # (q+u), k, v, dO in; dK, dV out (bf16); lse + delta (f32)
        attn_flops = 10.0 * B * H * T * T * Dh                         # S, dP, dV, dK and (in the following launch) dq products
        roof_attn = {"bound": "hbm", "kernel": "attention backward kernel (%s), one audio layer (B=%d L=%d H=%d Dh=%d): recomputes P incl. the "
                                               "position term, writes dK / dV and dS twice (bf16) for the dq / dE products" % (ATTN_BWD_KERNEL, B, T, H, Dh),
                     "achieved": rate(attn_bytes, a_ms, 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": frac(rate(attn_bytes, a_ms, 1e9, 1), HBM_PEAK_GBS), "traffic": None, "kernel_ms": None if a_ms is None else round(a_ms, 4),
                     "mfma_tflops": rate(0.8 * attn_flops, a_ms, 1e12, 1),
                     "note": "algorithmic bytes only (q, k, v, dO in; dK, dV out); the dS / dG slabs written (2 x B*H*L*L*2 bytes) are design traffic on top; "
                             "kernel_ms is the first audio layer's launch between two events on its stream"}
        w_ms = mean_pos(wgrad_ms)
        Di = enc["d_inner"]
        wg_layer = 2.0 * (B * T) * (3 * H * Dh * dm + H * Dh * dm + 2 * dm * Di)        # qkv_net, o_net, CoreNet.0, CoreNet.3 of one audio layer
        roof_wgrad = {"bound": "mfma", "kernel": "gemm_tn_bf16_group_kernel: the 16 weight-gradient GEMMs of four audio layers in one launch (256 tiles of "
                                                 "256x128, one per CU over K=%d, no atomics); first grouped launch of the step" % (B * T),
                      "achieved": rate(4 * wg_layer, w_ms, 1e12), "peak": peak, "unit": "TFLOP/s",
                      "frac": frac(rate(4 * wg_layer, w_ms, 1e12), peak), "traffic": None, "kernel_ms": None if w_ms is None else round(w_ms, 4)}
        if w_ms is None:
            roof_wgrad["reason"] = "no grouped weight-gradient launch in the timed steps (fp32 mode or --no-grouped-wgrads: per-layer launches)"
        if a_ms is None:
            roof_attn["reason"] = "probe 3 (fused attention backward of an audio-sized layer) never fired"
        if ran_exp:
            # exp-domain form: no pass over the lattice's rows is left - the forward reads the row-sum partials and two entries per row, the
            # backward writes a factor per row and patches two entries; what remains is the alpha / beta recursion (a serial chain per utterance)
            nparts = 4 * ((V + 255) // 256)
            loss_bytes = B_launch * T * U1 * (4.0 * nparts + 2 * 4 + 4 * 4 + 2 * 8 + 2 * 8 + 4 + 2 + 2 * 2 * 2)
            loss_gbs = rate(loss_bytes, l_ms, 1e9, 1)
            roof_loss = {"bound": "hbm", "kernel": "RNN-T loss op, exp-domain form: rnnt_prep_exp_kernel + rnnt_lattice_lds_kernel (%s ms) + rnnt_scale_exp_kernel "
                                                   "(%s ms), P = exp(logits - shift) [%d,%d,%d,%d] bf16 patched at 2 entries per row, emission logits f32 [rows, 2]"
                                                   % (lf and round(lf, 3), lb and round(lb, 3), B_launch, T, U1, V),
                         "achieved": loss_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frac(loss_gbs, HBM_PEAK_GBS),
                         "traffic": None, "kernel_ms": None if l_ms is None else round(l_ms, 4),
                         "note": "latency-bound by the lattice recursion (T+U serial steps per utterance; one workgroup per utterance and direction, one wave per 64 labels); the two-call form's loss op moves "
                                 "%.1f GB per step through rnnt_lse_kernel / rnnt_grad_kernel instead" % (B * (2.0 * es * T * U1 * V) / 1e9)}
        else:
            roof_loss = {"bound": "hbm", "kernel": "RNN-T loss op: rnnt_lse_kernel + rnnt_lattice_lds_kernel (%s ms) + rnnt_grad_kernel (%s ms), "
                                                   "logits [%d,%d,%d,%d] %s" % (lf and round(lf, 3), lb and round(lb, 3), B_launch, T, U1, V, "bf16" if es == 2 else "f32"),
                         "achieved": loss_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frac(loss_gbs, HBM_PEAK_GBS),
                         "traffic": None, "kernel_ms": None if l_ms is None else round(l_ms, 4)}
        # HBM-side traffic of the secondary lines' kernels from the committed PMC passes (tools/update_pmc_json.py), valid for the kernel
        # sources they were measured on and for the default workload
        pmc2 = os.path.join(ROOT, "profiles", "pmc_secondary_kernels.json")
        if os.path.exists(pmc2) and args.workload == "c2" and (B, T, U) == (32, 500, 50) and args.precision == "bf16":
            j2 = json.load(open(pmc2))
            for line, roof in (("attn", roof_attn), ("loss", roof_loss), ("wgrad", roof_wgrad)):
                ent = j2.get("lines", {}).get(line) or {}
                if line == "loss" and j2.get("loss_form") != ran_form:
                    continue
                fresh = all(_sha16(os.path.join(ROOT, "transformer-transducer_amd", "csrc", f)) == h for f, h in (ent.get("sources") or {}).items())
                if ent.get("traffic") is not None and fresh:
                    roof["traffic"] = ent["traffic"]
                    roof["pmc"] = {"measured_at_commit": j2.get("commit"), "source": j2.get("source")}
                elif ent.get("traffic") is not None:
                    roof["pmc"] = {"stale": "kernel source changed since the PMC passes: traffic withheld"}
        lattice_run = args.workload == "c5"             # BASELINE configs[4] is the lattice's HBM-roofline run: the loss op is its dominant-kernel line
        out = {
            "metric": "utterances/sec (fwd+bwd) on 80-d fbank T=%d U=%d" % (T, U), "value": round(utt_s, 3), "unit": "utt/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]: full T-T 12 audio / 6 label layers d_model=512 V=4334 (48.2M params)"
                                    if args.workload in ("c2", "c5") else
                                    "BASELINE configs[3]: joint_streaming.yaml 18/2 layers V=6485 (85.6M params), %s mask"
                                    % args.workload.split("-")[1]) +
                                   ", T=%d U=%d, batch %d/GPU, dropout 0.1, SGD momentum + clip 200" % (T, U, B),
                       "global_batch": world * B, "parallelism": "dp%d" % world,
                       "reserve_cus_in_backward": reserve},
            # host side of a step.  host_issue: wall time to issue ONE step (Python + ctypes + HIP runtime, forward thread and autograd's
            # backward thread) on a drained device - the host's own cost; well below ms_per_step = the step is GPU-bound.  Inside the timed
            # region the issue loop runs ahead until the HIP queue is full and then spins in the launch calls: host_enqueue (wall) and
            # host_cpu (process CPU time, every thread incl. the runtime's) therefore approach ms_per_step and say little by themselves.
            "host_issue_ms_per_step": round(1e3 * host_issue, 3),
            "host_cpu_ms_per_step": round(1e3 * host_cpu / args.steps, 3),
            "host_blocked_ms_per_step": round(max(0.0, 1e3 * (enqueue - host_cpu) / args.steps), 3),
            "host_enqueue_ms_per_step": round(1e3 * enqueue / args.steps, 3), "model_tflops": round(flops_per_utt(cfg, T, U1) * utt_s / 1e12, 2),
            "roofline": roof_loss if lattice_run else roof_joint,
            "roofline_joint" if lattice_run else "roofline_loss": roof_joint if lattice_run else roof_loss,
            "roofline_attn": roof_attn, "roofline_wgrad": roof_wgrad,
            "final_loss": round(float(last.detach()), 4),
        }
        ran = "exp" if ran_exp else ("fused" if fused_path else "two-call")      # the kernels that RAN (exp falls back to the memory form outside the persistent kernels' sizes)
        calls = {"two-call": "logits = model(inputs, targets); RNNTLoss()(logits, ...) as in train.py:51-53", "two-call-eager": "logits = model(inputs, targets); "
                 "RNNTLoss()(logits, ...) with TTMI_DEFERRED_LOGITS=0", "exp": "Transducer.loss(exp_domain=True)", "fused": "Transducer.loss()"}[form]
        kernels = {"two-call": "logits materialised (joint forward, RNN-T loss over the logits, joint backward)",
                   "fused": "fused joint + loss, memory form, %d utterances per chunk" % B_launch,
                   "exp": "fused joint + loss, exp-domain form, %d utterances per chunk" % B_launch}[ran]
        out["config"]["loss"] = calls + (" - the logits are a deferred handle (tt.model.DeferredLogits) consumed by RNNTLoss: " if form == "two-call" and fused_path else " - ") + kernels
        if form in ("exp", "two-call") and args.precision == "bf16" and not ran_exp:
            out["config"]["loss"] += " [chunks of %d x %d x %d lattice rows are outside the exp-domain kernels' sizes]" % (B_launch, T, U1)
        out["config"]["loss_form"] = ran
        out["config"]["call_sequence"] = form
        if args.graph:
            out["config"]["launch"] = "every step replayed from one captured HIP graph per rank (ttmi.train.GraphedStep)"
        st = spread_stats(step_ms)
        if st is not None:
            out["ms_per_step_spread"] = dict(st, note="per-step GPU time between events recorded after each step on the launch stream (this rank)")

        def form_entry(secs, spread, note):
            e = {"ms_per_step": round(1e3 * secs / args.steps, 3), "value": round(world * B * args.steps / secs, 3), "unit": "utt/s", "note": note}
            if spread:
                e["spread"] = spread_stats(spread)
            return e
        if form == "two-call":
            out["two_call_form"] = dict(form_entry(elapsed, step_ms, "train.py:51-53 unchanged IS the primary timed region of this run (the headline above)"),
                                        same_as_headline=True)
        if sync_two_call is not None:
            out["train_py_sync_form"] = form_entry(sync_two_call, sync_ms, "the same %d steps with train.py's own host synchronisations: "
                                                   "criterion(logits, targets.int(), inputs_length.int(), targets_length.int()) on int64 loader tensors (fresh int32 "
                                                   "tensors each step -> the length check's two max() reads, train.py:53) and float(loss) every step (train.py:60)" % args.steps)
        eff = dp_efficiency(utt_s, world, args.n1_value)
        if eff is not None:
            out["dp_efficiency"] = {"value": eff, "n1_value": args.n1_value, "note": "(utt/s at N) / (N x the supplied N = 1 utt/s); target >= 0.90 at N = 8"}
        if eager_two_call is not None:
            out["two_call_materialized_form"] = form_entry(eager_two_call, eager_ms, "the same %d steps with the logits materialised (TTMI_DEFERRED_LOGITS=0: rounds 1-3's "
                                                           "two-call form), timed right after the main region and before any graph capture" % args.steps)
        if explicit_form is not None:
            out["explicit_loss_form"] = form_entry(explicit_form, explicit_ms, "the same %d steps calling Transducer.loss(exp_domain=True) directly (rounds 2-3's headline "
                                                   "form): the kernels the deferred handle routes to" % args.steps)
        if graph_form is not None:
            out["graph_replay_form"] = dict(form_entry(graph_form, graph_ms, "the same %d steps (same call sequence, dropout with fresh masks per step, all-reduce, clip + SGD) "
                                                       "replayed from one captured HIP graph per rank, ttmi.train.GraphedStep, timed after the main region; the headline stays the "
                                                       "eager step because its kernels are timed live by HIP-event probes, which a replayed graph cannot carry" % args.steps),
                                            host_issue_ms_per_step=round(1e3 * graph_issue, 3), host_launches_per_step=1)
        elif graph_error is not None:
            out["graph_replay_form"] = {"error": graph_error, "note": "the capture of the step failed on this run; the primary region is unaffected"}
        elif world > 1 and not args.graph and not args.no_graph_form:
            out["graph_replay_form"] = {"skipped": "secondary graph capture runs on one rank by default; `--graph-form` (nccl) times it on N ranks, `--graph` makes it the primary region"}
        if fp32_form is not None:
            out["fp32_form"] = {"ms_per_step": round(1e3 * fp32_form / fp32_steps, 3), "value": round(world * B * fp32_steps / fp32_form, 3), "unit": "utt/s",
                                "dtype": "f32", "steps": fp32_steps,
                                "note": "TTMI_PRECISION=fp32 (exact-f32 MFMA everywhere, train.py's call sequence): the mode whose loss and gradients are "
                                        "within 1e-4 of the oracle (tests/test_configs_gpu.py::test_c2_full_model_fp32_end_to_end), timed right after the main region"}
        if x3_form is not None:
            out["bf16x3_form"] = {"ms_per_step": round(1e3 * x3_form / x3_steps, 3), "value": round(world * B * x3_steps / x3_form, 3), "unit": "utt/s",
                                  "dtype": "f32 data, bf16 MFMA in three terms", "steps": x3_steps,
                                  "note": "TTMI_PRECISION=bf16x3: the fp32 mode's data flow with its dense and attention-core products as hi.hi + lo.hi + hi.lo on the bf16 MFMA "
                                          "(~2^-16 relative per product); loss and every gradient within 1e-4 of the float64 oracle "
                                          "(tests/test_configs_gpu.py::test_c2_full_model_fp32_end_to_end[bf16x3]): the quick parity mode.  Since round 6 the logits of "
                                          "this mode are a deferred handle too (one fused joint + loss op over the whole batch, the loss gradient written as the bf16 planes "
                                          "the joint's backward multiplies; TTMI_DEFERRED_LOGITS=0 restores the materialised two-call path: +5 ms)"}
        if label_form is not None:
            out["throughput_form"] = {"ms_per_step": round(1e3 * label_form / args.steps, 3), "value": round(world * B * args.steps / label_form, 3), "unit": "utt/s", "steps": args.steps,
                                      "note": "TTMI_LABEL_VALUE_PRECISION=off and ttmi_set_option(13, 0): the timed step without round 6's parity measures (rounds 1 - 5's headline "
                                              "configuration); its loss error along the trajectory: loss_rel_err_trajectory_throughput_form"}
        if secondary_errors:
            out["secondary_errors"] = secondary_errors
        if primary_retry:
            out["primary_retry"] = {"first_attempt_failed_with": primary_retry, "note": "the timed region above is the complete second attempt"}
        if world == 1 and not args.no_cpu_baseline:
            try:
                model.eval()
                with torch.no_grad():
                    os.environ["TTMI_DEFERRED_LOGITS"] = "0"         # the MATERIALISED two-call form on the oracle sample (bf16 logits through the lattice kernels)
                    lg = model(inputs[:args.cpu_utts], targets[:args.cpu_utts])
                    costs = RNNTLoss(reduction="none")(lg, targets[:args.cpu_utts].int(), ilen[:args.cpu_utts], tlen[:args.cpu_utts])
                    del lg
                    os.environ["TTMI_DEFERRED_LOGITS"] = "0" if form == "two-call-eager" else ""
                    timed_costs = lambda n: (model.loss(inputs[:n], ilen[:n], targets[:n], tlen[:n], reduction="none", chunk=args.loss_chunk or None,
                                                        exp_domain=form == "exp") if form in ("exp", "fused") else
                                             RNNTLoss(reduction="none")(model(inputs[:n], targets[:n]), targets[:n].int(), ilen[:n], tlen[:n]))
                    if fused_path:              # the timed loss form on the same sample: its per-utterance costs against the same oracle
                        costs_form = timed_costs(args.cpu_utts)
                    enc_s, dec_s = model._encode(inputs[:args.cpu_utts], targets[:args.cpu_utts])      # the timed precision's encoder states of the same sample
                    enc_s, dec_s = enc_s.double().cpu().numpy(), dec_s.double().cpu().numpy()
                    # the whole batch's loss in the timed form against the fp32 mode of the same model (the mode the tests hold within 1e-6 of the
                    # oracle, tests/test_configs_gpu.py::test_c2_full_model_fp32_end_to_end): what the precision mode does to the step's LOSS - the
                    # mean over the batch that train.py:53 computes - as opposed to the worst single utterance of the oracle sample below
                    batch_rel = None
                    if args.precision == "bf16" and args.workload == "c2":
                        c16 = timed_costs(B).double()
                        os.environ["TTMI_PRECISION"] = "fp32"
                        try:
                            ops.weights_fresh()
                            c32 = RNNTLoss(reduction="none")(model(inputs, targets), targets.int(), ilen, tlen).double()
                        finally:
                            os.environ["TTMI_PRECISION"] = args.precision
                            ops.weights_fresh()
                        batch_rel = float((c16.mean() - c32.mean()).abs() / c32.mean())
                        utt_rel = float(((c16 - c32).abs() / c32).max())
                        del c16, c32
                v, dt, rel, times, oracle_costs = cpu_baseline(model, feats, proj, targets, T, U, args.cpu_utts, costs.float().cpu().numpy(), args.cpu_reps)
                floor = encoder_floor(model, enc_s, dec_s, targets, T, U, args.cpu_utts, oracle_costs)
                out["cpu_baseline"] = {"value": round(v, 4), "unit": "utt/s", "cores": _blas_threads(), "kind": "port",
                                       "sample": "%d utterance(s) of the same workload (B=%d as in BASELINE.md §3), fwd+loss+bwd through "
                                                 "oracle/tt_oracle.py (numpy, multithreaded BLAS) + oracle/rnnt_lattice.c, median of %d runs "
                                                 "(%s s)" % (args.cpu_utts, args.cpu_utts, len(times), ", ".join("%.1f" % t for t in times)),
                                       # the reference's OWN PyTorch CPU path, timed once in the survey container (it cannot travel to the GPU box):
                                       "reference_cpu_probe": {"value": 0.33, "unit": "utt/s", "cores": 8, "source": "BASELINE.md §2 (fwd+bwd, lattice excluded, B=2)"}}
                out["loss_rel_err_vs_oracle"] = float("%.3e" % rel)
                # how much of that distance the encoders' precision alone accounts for: the ORACLE's float64 joint + lattice fed the GPU's encoder
                # states of the same sample (bf16 mode: 12 / 6 layers of bf16 GEMM operands leave ~2e-3 relative error on the states)
                out["loss_rel_err_encoder_states_only"] = float("%.3e" % floor)
                if batch_rel is not None:
                    out["loss_rel_err_batch_vs_fp32_mode"] = {"batch_mean": float("%.3e" % batch_rel), "worst_utterance": float("%.3e" % utt_rel), "utterances": B,
                                                              "note": "the timed form's loss of the whole batch against TTMI_PRECISION=fp32 on the same weights and inputs (eval mode)"}
                if batch_rel is not None and not args.no_trajectory:
                    try:
                        out["loss_rel_err_trajectory"] = loss_error_trajectory(dev, inputs, targets, ilen, tlen)
                        out["loss_rel_err_trajectory_worst"] = out["loss_rel_err_trajectory"]["batch_mean_worst"]
                        if label_form is not None:
                            ops.set_option(13, 0)
                            try:
                                lp = loss_error_trajectory(dev, inputs, targets, ilen, tlen, label_value="off")
                            finally:
                                ops.set_option(13, 2)
                            lp["note"] = "the same trajectory with TTMI_LABEL_VALUE_PRECISION=off and option 13 = 0 (throughput_form)"
                            out["loss_rel_err_trajectory_throughput_form"] = lp
                    except Exception as exc:        # a secondary measurement never takes the line down
                        out["loss_rel_err_trajectory"] = {"error": repr(exc)[:300]}
                if fused_path:
                    out["loss_rel_err_vs_oracle_timed_form"] = float("%.3e" % (np.abs(costs_form.float().cpu().numpy() - oracle_costs).max() / np.abs(oracle_costs).max()))
            except Exception as exc:      # the checker legs never take the measured line down: what is missing says why
                out.setdefault("cpu_baseline", {"error": "%s: %s" % (type(exc).__name__, str(exc).splitlines()[0][:300])})
                out["checker_error"] = "%s: %s" % (type(exc).__name__, str(exc).splitlines()[0][:300])

        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
