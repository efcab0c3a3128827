#!/usr/bin/env python3
"""Round 5: root-causing the token divergence of decode_batch under label-encoder graph replay (VERDICT r4 'weak' 3).
One process = tools/bench_decode.run()'s call order with every pass's tokens kept, plus (TRACE=1) a check of EVERY replayed label state
against an eager label-encoder call on the same tokens, at the moment of the replay.  Prints one JSON line."""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))
os.environ["TTMI_PRECISION"] = os.environ.get("PREC", "fp32")
import torch
from bench import c2_config
from tt import model as M
from tt.model import Transducer
TRACE = os.environ.get("TRACE", "0") == "1"
UTTS = int(os.environ.get("UTTS", 32))
dev = torch.device("cuda", 0)
cfg = c2_config()
torch.manual_seed(1)
model = Transducer(cfg).to(dev).eval()
model.config["decode_batch_graphs"] = True
g = torch.Generator(device=dev).manual_seed(1234)
feats = torch.randn(UTTS, 500, 80, device=dev, generator=g)
proj = torch.randn(80, 512, device=dev, generator=torch.Generator(device=dev).manual_seed(7)) / 80 ** 0.5
inputs = feats @ proj
lens = [500] * UTTS
hb = lambda t: hashlib.sha256(t.detach().float().cpu().numpy().tobytes()).hexdigest()[:8]
ht = lambda x: "%s/%d" % (hashlib.sha256(repr(x).encode()).hexdigest()[:8], sum(map(len, x)))
events = []
orig_state = M._LabelStateGraphs.state
phase = ["warm"]


def traced_state(self, L):
    captured = L in self.graphs
    out = orig_state(self, L)
    if TRACE and phase[0].startswith("P"):
        a = out.clone()
        toks = self.master[:, :L].contiguous()
        e = self.decoder(toks)[:, -1:, :]
        if not torch.equal(a, e):
            b = orig_state(self, L).clone()
            rows = (a != e).flatten(1).any(1).nonzero().flatten().tolist()
            events.append({"phase": phase[0], "L": L, "captured_before": captured, "rows": rows[:8], "nrows": len(rows),
                           "maxdiff": float((a - e).abs().max()), "second_replay_equals_eager": bool(torch.equal(b, e)),
                           "second_replay_equals_first": bool(torch.equal(b, a))})
    return out


M._LabelStateGraphs.state = traced_state
res = {}
with torch.no_grad():
    enc1 = model.encoder(inputs[:1], None)
    dec1 = model.decoder(torch.zeros(1, 1, dtype=torch.long, device=dev))
    z = model.joint(enc1, dec1)[0, :, 0, :].float()
    margin = z[:, 1:].max(dim=1).values - z[:, 0]
    model.joint.project_layer.bias[0] += torch.quantile(margin, 0.9)
    res["W"] = ht(model.recognize(inputs, lens))
    torch.cuda.synchronize()
    enc = model.encoder(inputs, None)
    torch.cuda.synchronize()
    res["inputs"], res["enc"] = hb(inputs), hb(enc)
    if os.environ.get("SINGLES", "1") == "1":
        phase[0] = "S"
        res["S"] = ht([model.decode(enc[b], lens[b]) for b in range(UTTS)])
        torch.cuda.synchronize()
    for i in (1, 2, 3):
        phase[0] = "P%d" % i
        res["P%d" % i] = ht(model.decode_batch(enc, lens))
    phase[0] = "E"
    model.config["decode_batch_graphs"] = False
    res["E"] = ht(model.decode_batch(enc, lens))
    model.config["decode_batch_shrink"] = False
    res["EN"] = ht(model.decode_batch(enc, lens))
    model.config["decode_batch_shrink"] = True
    model.config["decode_graphs"] = False
    res["SE"] = ht([model.decode(enc[b], lens[b]) for b in range(UTTS)])
res["events"] = events[:6]
res["n_events"] = len(events)
print(json.dumps(res), flush=True)
