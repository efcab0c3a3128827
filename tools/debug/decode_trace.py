#!/usr/bin/env python3
"""Replays tools/bench_decode.run()'s call sequence and records, per symbol step of decode_batch, hashes of the label states and of the history
buffer; runs the pass twice and reports the first step at which the two passes differ (the batched decode was seen to give one of two token sets)."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))
os.environ["TTMI_PRECISION"] = "fp32"
import torch
from bench import c2_config
from tt.model import Transducer
from ttmi import ops
dev = torch.device("cuda", 0)
cfg = c2_config()
torch.manual_seed(1)
model = Transducer(cfg).to(dev).eval()
g = torch.Generator(device=dev).manual_seed(1234)
feats = torch.randn(32, 500, 80, device=dev, generator=g)
proj = torch.randn(80, 512, device=dev, generator=torch.Generator(device=dev).manual_seed(7)) / 80 ** 0.5
inputs = feats @ proj
lens = [500] * 32
h = lambda t: hashlib.sha256(t.detach().float().cpu().numpy().tobytes()).hexdigest()[:8]
trace = []
orig_state = None
def patched_state(self, L, _o=[None]):
    out = _o[0](self, L)
    trace.append((L, h(out), h(self.master[:, :L].float())))
    return out
from tt import model as M
patched_state.__defaults__[0][0] = M._LabelStateGraphs.state
M._LabelStateGraphs.state = patched_state
with torch.no_grad():
    enc1 = model.encoder(inputs[:1], None)
    dec1 = model.decoder(torch.zeros(1, 1, dtype=torch.long, device=dev))
    z = model.joint(enc1, dec1)[0, :, 0, :].float()
    margin = z[:, 1:].max(dim=1).values - z[:, 0]
    model.joint.project_layer.bias[0] += torch.quantile(margin, 0.9)
    model.recognize(inputs, lens)
    torch.cuda.synchronize()
    enc = model.encoder(inputs, None)
    torch.cuda.synchronize()
    del trace[:]
    singles = [model.decode(enc[b], lens[b]) for b in range(32)]
    torch.cuda.synchronize()
    passes = []
    for _ in range(3):
        del trace[:]
        hyp = model.decode_batch(enc, lens)
        passes.append((hashlib.sha256(repr(hyp).encode()).hexdigest()[:8], list(trace)))
print("passes:", [p[0] for p in passes], "single:", hashlib.sha256(repr(singles).encode()).hexdigest()[:8])
a, b = passes[0][1], passes[1][1]
for i, (x, y) in enumerate(zip(a, b)):
    if x != y:
        print("first differing label-state call: #%d  pass0 %s  pass1 %s" % (i, x, y))
        break
else:
    print("label states identical in passes 0 and 1 (%d calls)" % len(a))
