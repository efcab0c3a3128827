import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
os.environ["TTMI_PRECISION"] = "fp32"
import bench
from conftest import rel_err
from oracle import tt_oracle as O
from tt.model import Transducer
from ttmi import ops
from warprnnt_pytorch import RNNTLoss
cfg = bench.c2_config(); cfg["dropout"] = 0.0
torch.manual_seed(1)
model = Transducer(cfg).cuda().eval()
B, T, U, V = 2, 500, 50, 4334
gen = torch.Generator().manual_seed(1234)
inp = torch.randn(B, T, 512, generator=gen); tgt = torch.randint(1, V, (B, U), generator=gen)
tl, ul = np.array([T, 431], dtype=np.int32), np.array([U, 37], dtype=np.int32)
sd32 = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
sd64 = {k: (v.astype(np.float64) if v.dtype == np.float32 else v) for k, v in sd32.items()}
w64 = O.transducer_loss_and_grads(inp.numpy().astype(np.float64), tgt.numpy(), tl, ul, sd64)
def run(tag):
    model.zero_grad()
    x = inp.cuda().requires_grad_(True)
    logits = model(x, tgt.cuda())
    loss = RNNTLoss()(logits, tgt.int().cuda(), torch.tensor(tl).cuda(), torch.tensor(ul).cuda())
    loss.backward(); ops.join_side_streams(); torch.cuda.synchronize()
    errs = ["%.1e" % rel_err(model.encoder.layers[i].r_emb.grad.cpu().numpy(), w64["grads"]["encoder.layers.%d.r_emb" % i]) for i in range(12)]
    print(tag, "dinputs %.2e" % rel_err(x.grad.cpu().numpy(), w64["dinputs"]), "r_emb per layer", " ".join(errs), flush=True)
    return {n: p.grad.clone() for n, p in model.named_parameters()}
a = run("overlap on ")
a2 = run("overlap on again")
model.config["overlap_label_encoder"] = False
b = run("overlap off")
ops.set_option(7, 0)
c = run("no skinny kernel")
ops.set_option(7, 128)
worst = max((rel_err(a[n].cpu().numpy(), a2[n].cpu().numpy()), n) for n in a)
print("run-to-run", worst)
