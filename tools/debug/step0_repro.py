import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/transformer-transducer_amd"); sys.path.insert(0, "/root/repo/tests")
os.environ["TTMI_PRECISION"] = "bf16"
import numpy as np, torch
import test_dp_nccl_gpu as T
from conftest import rel_err
dev = torch.device("cuda", 0)
ref = None
for rep in range(6):
    T._bench_like(dev, 1, 0, 1, hooks=False)
    g = T._bench_like.first_step_grad
    if ref is None: ref = g
    d = np.abs(g - ref)
    print("rep %d: step-0 gradient vs rep 0: rel %.2e, differing entries %d, largest at flat index %d" % (rep, rel_err(g, ref), int((d > 0).sum()), int(d.argmax())), flush=True)
from tt.model import Transducer
from ttmi.train import FlatModel
torch.manual_seed(1)
m = Transducer(T._bench_cfg()).to(dev)
f = FlatModel(m)
for (n, p), o in zip(m.named_parameters(), f.offsets):
    if o <= 7542138 < o + p.numel():
        print("flat index 7542138 is", n, tuple(p.shape), "offset in tensor", 7542138 - o, "-> row", (7542138 - o) // p.shape[-1] if p.dim() > 1 else "-")
# per-parameter differences between outcome X (rep 0) and outcome Y
T._bench_like(dev, 1, 0, 1, hooks=False)
outs = [ref]
for rep in range(8):
    T._bench_like(dev, 1, 0, 1, hooks=False)
    g = T._bench_like.first_step_grad
    if rel_err(g, ref) > 1e-6:
        rows = []
        for (n, p), o in zip(m.named_parameters(), f.offsets):
            a, b = g[o:o + p.numel()], ref[o:o + p.numel()]
            e = rel_err(a, b)
            if e > 1e-6:
                rows.append((e, n, int((np.abs(a - b) > 1e-6 * np.abs(b).max()).sum())))
        for e, n, c in sorted(rows, reverse=True)[:12]:
            print("   %.2e %s (%d entries)" % (e, n, c))
        break
