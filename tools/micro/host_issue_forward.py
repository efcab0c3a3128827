"""host time to ISSUE the forward passes of a C2 step (autograd on): audio encoder, label encoder (bf16 pass), label value pass - against their GPU time.
usage: python tools/micro/host_issue_forward.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))
os.environ.setdefault("TTMI_PRECISION", "bf16")
import torch
import bench
from tt.model import Transducer
from tt.transformer import MaskSpec
from ttmi.train import FlatModel
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = Transducer(bench.c2_config()).to(dev).train()
flat = FlatModel(model); flat.enable_shadows()
x = torch.randn(32, 500, 512, device=dev)
y = torch.randint(1, 4334, (32, 51), device=dev)
def measure(name, f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    host = gpu = 0.0
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); out = f(); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        host += t1 - t0; gpu += t2 - t0
        del out
    print("%-34s host issue %.2f ms, issue + drain %.2f ms" % (name, host / reps * 1e3, gpu / reps * 1e3))
measure("audio encoder forward (12 layers)", lambda: model.encoder(x))
measure("label encoder forward (6 layers)", lambda: model.decoder(y, MaskSpec(1)))
def value():
    with torch.no_grad():
        return model.decoder(y, MaskSpec(1), prec=2)
measure("label value pass (bf16x3, no grad)", value)
measure("both label passes (_label_states)", lambda: model._label_states(y))
