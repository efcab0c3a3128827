#!/usr/bin/env python3
"""Does a captured label-encoder graph read scratch memory it has not written?  Replay graph L, poison the graph stream's scratch arena (and the
default stream's), replay again with the same tokens: the two outputs must be bit-identical."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))
os.environ["TTMI_PRECISION"] = "fp32"
import torch
from bench import c2_config
from tt.model import Transducer
from ttmi import ops
dev = torch.device("cuda", 0)
torch.manual_seed(1)
model = Transducer(c2_config()).to(dev).eval()
B = int(os.environ.get("B", 32))
with torch.no_grad():
    gs = model._label_state_graphs(dev, B)
    gs.master.copy_(torch.randint(1, 4334, gs.master.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(3)))
    for L in (1, 2, 7, 33, 41, 44, 45, 52, 60, 61, 64, 65, 100):
        a = gs.state(L).clone()
        for key, t in ops._ws_cache.items():
            t.fill_(float(os.environ.get("POISON", "nan")))
        b = gs.state(L).clone()
        eager = model.decoder(gs.master[:, :L].contiguous())[:, -1:, :]
        print("L %3d: replay == replay after poison: %s   replay == eager: %s   max |diff| %.3g / %.3g  nan %d" %
              (L, torch.equal(a, b), torch.equal(a, eager), float((a - b).abs().max()), float((a - eager).abs().max()), int(torch.isnan(b).sum())))
