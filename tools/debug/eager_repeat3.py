#!/usr/bin/env python3
"""per-step gradients of repeated eager runs over a given batch sequence (argv[1], e.g. 0,0,2,3,4), taken as in tests/test_dp_nccl_gpu.py::_graph_twin"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "transformer-transducer_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ["TTMI_PRECISION"] = "bf16"
import numpy as np
import torch
from conftest import rel_err
import test_dp_nccl_gpu as T
from tt.model import Transducer
from ttmi import ops
from ttmi.train import FlatModel, FusedOptimizer, GradSync

dev = torch.device("cuda", 0)
seq = [int(v) for v in sys.argv[1].split(",")]
SYNC = len(sys.argv) > 2 and sys.argv[2] == "sync"


def run():
    torch.manual_seed(1)
    model = Transducer(T._bench_cfg()).to(dev).train()
    flat = FlatModel(model)
    flat.enable_grouped_wgrads()
    flat.enable_shadows()
    sync = GradSync(flat, bucket_mb=4)
    opt = FusedOptimizer(flat, kind="sgd", lr=0.00025, momentum=0.9, max_grad_norm=200.0)
    il = torch.full((8,), 512, dtype=torch.int32, device=dev)
    tl = torch.full((8,), 7, dtype=torch.int32, device=dev)
    grads, losses = [], []
    for b in seq:
        x, y = T._bench_data(b, 0)
        flat.zero_grad()
        sync.start_step()
        loss = model.loss(x.to(dev), il, y.to(dev), tl, exp_domain=True)
        loss.backward()
        sync.finish()
        if SYNC:
            torch.cuda.synchronize()
        grads.append(flat.grad.clone())
        losses.append(loss.detach().clone())
        opt.step()
    torch.cuda.synchronize()
    ops.wgrad_queue = None
    flat.disable_shadows()
    names = [n for n, p in model.named_parameters()]
    return [g.cpu().numpy() for g in grads], names, [p.numel() for p in flat.params], flat.offsets, [float(l) for l in losses]


runs = [run() for _ in range(4)]
for k in range(1, 4):
    ga, names, sizes, offs, la = runs[0]
    gb, lb = runs[k][0], runs[k][4]
    for s in range(len(ga)):
        worst = sorted(((rel_err(gb[s][o:o + n], ga[s][o:o + n]), nm) for nm, n, o in zip(names, sizes, offs)), reverse=True)[:3]
        print("run %d vs 0, step %d (batch %d): grad %.2e loss %.8g vs %.8g  worst: %s" % (k, s, seq[s], rel_err(gb[s], ga[s]), lb[s], la[s],
                                                                                     ", ".join("%s %.1e" % (n, e) for e, n in worst)), flush=True)
