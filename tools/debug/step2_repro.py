"""repeat the three-step eager run of tests/test_dp_nccl_gpu.py::_bench_like in ONE process and compare the gradient of the last step
between repetitions: how far apart do two runs of the same program end up, and in which parameters?
usage: python tools/debug/step2_repro.py [reps] (env: TTMI_LABEL_VALUE_PRECISION, TTMI_OPTIONS select the configuration)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "transformer-transducer_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ["TTMI_PRECISION"] = "bf16"
import numpy as np, torch
import test_dp_nccl_gpu as T
from conftest import rel_err
from tt.model import Transducer
from ttmi.train import FlatModel
dev = torch.device("cuda", 0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.manual_seed(1)
m = Transducer(T._bench_cfg()).to(dev)
f = FlatModel(m)
names = [(n, o, p.numel()) for (n, p), o in zip(m.named_parameters(), f.offsets)]
del m, f
ref = ref_p = None
for rep in range(reps):
    g, params = T._bench_like(dev, 1, 0, steps, hooks=False)[:2]
    if ref is None:
        ref, ref_p = g, params
        continue
    print("rep %d: last gradient vs rep 0 %.2e, parameters %.2e, step-0 gradient %.2e" % (rep, rel_err(g, ref), rel_err(params, ref_p), 0.0), flush=True)
    rows = sorted(((rel_err(g[o:o + n], ref[o:o + n]) * float(np.abs(ref[o:o + n]).max()) / float(np.abs(ref).max()), rel_err(g[o:o + n], ref[o:o + n]), nm)
                   for nm, o, n in names), reverse=True)[:5]
    for w, e, nm in rows:
        print("      %.2e of the largest gradient (%.2e of its own) %s" % (w, e, nm))
