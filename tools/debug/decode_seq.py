#!/usr/bin/env python3
"""which call sequence makes the first pure-replay decode_batch differ?  SEQ=A: capture pass, then batched x3.  SEQ=B: capture pass, singles, batched x3."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))
os.environ["TTMI_PRECISION"] = "fp32"
import torch
from bench import c2_config
from tt.model import Transducer
dev = torch.device("cuda", 0)
cfg = c2_config()
torch.manual_seed(1)
model = Transducer(cfg).to(dev).eval()
model.config["decode_batch_graphs"] = True
g = torch.Generator(device=dev).manual_seed(1234)
feats = torch.randn(32, 500, 80, device=dev, generator=g)
proj = torch.randn(80, 512, device=dev, generator=torch.Generator(device=dev).manual_seed(7)) / 80 ** 0.5
inputs = feats @ proj
lens = [500] * 32
h = lambda x: hashlib.sha256(repr(x).encode()).hexdigest()[:8]
PRE = os.environ.get("PRE", "")
with torch.no_grad():
    if "e" in PRE:
        enc1 = model.encoder(inputs[:1], None)
    if "d" in PRE:
        dec1 = model.decoder(torch.zeros(1, 1, dtype=torch.long, device=dev))
    if "j" in PRE:
        z = model.joint(enc1, dec1)[0, :, 0, :].float()
    if "q" in PRE:
        margin = z[:, 1:].max(dim=1).values - z[:, 0]
        qq = torch.quantile(margin, 0.9)
    model.joint.project_layer.bias[0] += 1.1084104776382446
    if os.environ.get("SEQ", "A") == "C":                  # tools/bench_decode.run()'s own order: recognize (captures), the encoder again, singles, batched
        first = model.recognize(inputs, lens)
        torch.cuda.synchronize()
        enc = model.encoder(inputs, None)
        torch.cuda.synchronize()
    else:
        enc = model.encoder(inputs, None)
        first = model.decode_batch(enc, lens)              # captures the graphs while decoding
    res = [h(first)]
    if os.environ.get("SEQ", "A") in ("B", "C"):
        single = [model.decode(enc[b], lens[b]) for b in range(32)]
        res.append("s:" + h(single))
    for _ in range(3):
        res.append(h(model.decode_batch(enc, lens)))
print(os.environ.get("SEQ", "A"), PRE, " ".join(res))
