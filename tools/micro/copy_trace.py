"""which torch ops issue the device-to-device copies of a training step (torch profiler, one step)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))
os.environ["TTMI_PRECISION"] = "bf16"
import torch
from bench import c2_config
from tt.model import Transducer
from ttmi.train import FlatModel, FusedOptimizer, GradSync
from warprnnt_pytorch import RNNTLoss
dev = torch.device("cuda", 0)
torch.manual_seed(1)
model = Transducer(c2_config()).to(dev).train()
flat = FlatModel(model); sync = GradSync(flat)
opt = FusedOptimizer(flat, kind="sgd", lr=0.00025, momentum=0.9, max_grad_norm=200.0, world=1)
crit = RNNTLoss()
B, T, U = 8, 200, 20
x = torch.randn(B, T, 512, device=dev); y = torch.randint(1, 4334, (B, U), device=dev)
il = torch.full((B,), T, dtype=torch.int32, device=dev); tl = torch.full((B,), U, dtype=torch.int32, device=dev)
def step():
    flat.zero_grad(); sync.start_step()
    loss = crit(model(x, y), y.int(), il, tl); loss.backward(); sync.finish(); opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    step(); torch.cuda.synchronize()
rows = [(e.key, e.count) for e in prof.key_averages() if e.key.startswith("aten::") or "Memcpy" in e.key or "Memset" in e.key]
for k, c in sorted(rows, key=lambda r: -r[1])[:25]: print("%-40s %d" % (k, c))
