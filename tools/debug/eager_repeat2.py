#!/usr/bin/env python3
"""Where does the eager bench-like step stop being reproducible?  Per-step gradients of repeated runs, with the label encoder on its side
stream (overlap) and in line."""
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
MODE = sys.argv[1] if len(sys.argv) > 1 else "default"


def run(overlap, steps=4, grouped=True, shadows=True, exp=True):
    torch.manual_seed(1)
    cfg = T._bench_cfg()
    cfg["overlap_label_encoder"] = overlap
    model = Transducer(cfg).to(dev).train()
    flat = FlatModel(model)
    if grouped:
        flat.enable_grouped_wgrads()
    if shadows:
        flat.enable_shadows()
    sync = GradSync(flat, bucket_mb=4)
    opt = FusedOptimizer(flat, kind="sgd", lr=0.00025, momentum=0.9, max_grad_norm=200.0)
    il = torch.full((8,), 512, dtype=torch.int32, device=dev)
    tl = torch.full((8,), 7, dtype=torch.int32, device=dev)
    grads = []
    for b in range(steps):
        x, y = T._bench_data(b, 0)
        flat.zero_grad()
        sync.start_step()
        loss = model.loss(x.to(dev), il, y.to(dev), tl, exp_domain=exp)
        loss.backward()
        sync.finish()
        torch.cuda.synchronize()
        grads.append(flat.grad.cpu().numpy().copy())
        opt.step()
    torch.cuda.synchronize()
    ops.wgrad_queue = None
    flat.disable_shadows()
    names = [n for n, p in model.named_parameters()]
    return grads, names, [p.numel() for p in flat.params], flat.offsets


def cmp(tag, a, b):
    ga, names, sizes, offs = a
    gb = b[0]
    for s in range(len(ga)):
        e = rel_err(gb[s], ga[s])
        worst = sorted(((rel_err(gb[s][o:o + n], ga[s][o:o + n]), nm) for nm, n, o in zip(names, sizes, offs)), reverse=True)[:3]
        print("%s step %d: %.2e   worst: %s" % (tag, s, e, ", ".join("%s %.1e" % (n, e) for e, n in worst)), flush=True)


kw = {}
if MODE == "nogroup":
    kw = dict(grouped=False)
elif MODE == "noshadow":
    kw = dict(shadows=False)
elif MODE == "plain":
    kw = dict(exp=False)
r0 = run(True, **kw)
r1 = run(True, **kw)
r2 = run(True, **kw)
cmp("overlap run1 vs run0", r0, r1)
cmp("overlap run2 vs run0", r0, r2)
s0 = run(False, **kw)
s1 = run(False, **kw)
cmp("inline  run1 vs run0", s0, s1)
cmp("inline0 vs overlap0 ", r0, s0)
cmp("inline0 vs overlap1 ", r1, s0)
