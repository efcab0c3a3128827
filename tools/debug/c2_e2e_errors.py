"""C2 end-to-end fp32: HIP path vs float64 oracle and float32 oracle vs float64 oracle (calibration of fp32 rounding noise at depth 12)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
os.environ["TTMI_PRECISION"] = "fp32"
import bench
from conftest import rel_err
from oracle import tt_oracle as O
from tt.model import Transducer
from warprnnt_pytorch import RNNTLoss
cfg = bench.c2_config(); cfg["dropout"] = 0.0
torch.manual_seed(1)
model = Transducer(cfg).cuda().eval()
B, T, U, V = 2, 500, 50, 4334
gen = torch.Generator().manual_seed(1234)
inp = torch.randn(B, T, 512, generator=gen); tgt = torch.randint(1, V, (B, U), generator=gen)
tl, ul = np.array([T, 431], dtype=np.int32), np.array([U, 37], dtype=np.int32)
x = inp.cuda().requires_grad_(True)
logits = model(x, tgt.cuda())
loss = RNNTLoss()(logits, tgt.int().cuda(), torch.tensor(tl).cuda(), torch.tensor(ul).cuda())
loss.backward(); torch.cuda.synchronize()
sd32 = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
sd64 = {k: (v.astype(np.float64) if v.dtype == np.float32 else v) for k, v in sd32.items()}
w64 = O.transducer_loss_and_grads(inp.numpy().astype(np.float64), tgt.numpy(), tl, ul, sd64)
w32 = O.transducer_loss_and_grads(inp.numpy(), tgt.numpy(), tl, ul, sd32)
print("logits  hip %.2e  o32 %.2e" % (rel_err(logits.detach().cpu().numpy(), w64["logits"]), rel_err(w32["logits"], w64["logits"])))
print("loss    hip %.2e  o32 %.2e" % (abs(float(loss) - w64["loss"]) / w64["loss"], abs(w32["loss"] - w64["loss"]) / w64["loss"]))
print("dinputs hip %.2e  o32 %.2e" % (rel_err(x.grad.cpu().numpy(), w64["dinputs"]), rel_err(w32["dinputs"], w64["dinputs"])))
for name, p in model.named_parameters():
    print("%-70s hip %.2e  o32 %.2e" % (name, rel_err(p.grad.cpu().numpy(), w64["grads"][name]), rel_err(w32["grads"][name], w64["grads"][name])))
