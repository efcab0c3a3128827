"""repeat ONE eager step of tests/test_dp_nccl_gpu.py::_bench_like in one process: loss, label states handed to the joint, their gradient, and the
step's parameter gradients, each compared with repetition 0.
env: TTMI_LABEL_VALUE_PRECISION / TTMI_OPTIONS select the configuration; DEBUG_NO_OVERLAP=1 keeps the label encoder on the main stream;
DEBUG_POISON=1 fills every scratch arena with NaN bytes before each repetition"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "transformer-transducer_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ["TTMI_PRECISION"] = "bf16"
import numpy as np, torch
import test_dp_nccl_gpu as T
from conftest import rel_err
from tt.model import Transducer
from ttmi import ops
from ttmi.train import FlatModel
dev = torch.device("cuda", 0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
if os.environ.get("DEBUG_NO_OVERLAP") == "1":
    cfg0 = T._bench_cfg
    def cfg():
        c = cfg0(); c["overlap_label_encoder"] = False; c.overlap_label_encoder = False; return c
    T._bench_cfg = cfg
torch.manual_seed(1)
m = Transducer(T._bench_cfg()).to(dev)
f = FlatModel(m)
names = [(n, o, p.numel()) for (n, p), o in zip(m.named_parameters(), f.offsets)]
del m, f
seen = {}
orig = Transducer._label_states
def spy(self, targets):
    out = orig(self, targets)
    seen["dec"] = out.detach().clone()
    out.register_hook(lambda g: seen.__setitem__("ddec", g.detach().clone()))
    return out
Transducer._label_states = spy
orig_loss = Transducer.loss
def spy_loss(self, *a, **k):
    out = orig_loss(self, *a, **k)
    seen["loss"] = out.detach().clone()
    return out
Transducer.loss = spy_loss
ref = None
for rep in range(reps):
    if os.environ.get("DEBUG_POISON") == "1":
        for t in ops._ws_cache.values():
            t.view(torch.uint8).fill_(255)
        torch.cuda.synchronize()
    T._bench_like(dev, 1, 0, 1, hooks=False)
    cur = dict(loss=seen["loss"].cpu().numpy(), g=T._bench_like.first_step_grad, dec=seen["dec"].cpu().numpy(), ddec=seen["ddec"].cpu().numpy())
    if ref is None:
        ref = cur
        continue
    d = np.abs(cur["ddec"] - ref["ddec"])
    print("rep %d vs rep 0: loss %s, label states %.2e, their gradient %.2e (%d of %d entries differ, largest |diff| %.2e of max %.2e), parameter gradients %.2e"
          % (rep, "same bits" if np.array_equal(cur["loss"], ref["loss"]) else "%.2e" % rel_err(cur["loss"], ref["loss"]), rel_err(cur["dec"], ref["dec"]), rel_err(cur["ddec"], ref["ddec"]), int((d > 0).sum()), d.size, d.max(), np.abs(ref["ddec"]).max(), rel_err(cur["g"], ref["g"])), flush=True)
    rows = sorted(((rel_err(cur["g"][o:o + n], ref["g"][o:o + n]), nm) for nm, o, n in names), reverse=True)[:4]
    print("      " + "; ".join("%.1e %s" % (e, nm.replace("MultiHeadAttention.", "")) for e, nm in rows))
