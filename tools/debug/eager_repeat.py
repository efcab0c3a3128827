#!/usr/bin/env python3
"""Is the eager bench-like step reproducible from run to run in a FRESH process?  (tests/test_dp_nccl_gpu.py::_graph_twin, repeated;
prints the parameters whose last-step gradients differ most between repetitions)"""
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
from ttmi.train import FlatModel

dev = torch.device("cuda", 0)
runs = [T._graph_twin(dev, 1, 0, 5) for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4)]
torch.manual_seed(1)
model = Transducer(T._bench_cfg()).to(dev)
flat = FlatModel(model)
names = [n for n, p in model.named_parameters()]
for k in range(1, len(runs)):
    g0, g1 = runs[0][0], runs[k][0]
    print("run %d vs 0: gradients %.2e parameters %.2e" % (k, rel_err(g1, g0), rel_err(runs[k][1], runs[0][1])))
    worst = sorted(((rel_err(g1[o:o + p.numel()], g0[o:o + p.numel()]), n) for n, p, o in zip(names, flat.params, flat.offsets)), reverse=True)[:6]
    print("   worst:", ", ".join("%s %.1e" % (n, e) for e, n in worst))
