"""Only the batched greedy decoder (Transducer.decode_batch) on tools/bench_decode.py's workload, `--passes` times after one warm-up pass: the
process to put under `rocprofv3 --kernel-trace --stats` for the per-kernel picture of a symbol step.
    python3 tools/debug/decode_batch_only.py [--utts 32] [--passes 2]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))
ap = argparse.ArgumentParser()
ap.add_argument("--utts", type=int, default=32)
ap.add_argument("--passes", type=int, default=2)
ap.add_argument("--emit-rate", type=float, default=0.1)
ap.add_argument("--block", type=int, default=None)
a = ap.parse_args()
os.environ["TTMI_PRECISION"] = "fp32"
from bench import c2_config
from tt.model import Transducer

dev = torch.device("cuda", 0)
cfg = c2_config()
torch.manual_seed(1)
model = Transducer(cfg).to(dev).eval()
d = cfg["enc"]["d_model"]
g = torch.Generator(device=dev).manual_seed(1234)
feats = torch.randn(a.utts, 500, 80, device=dev, generator=g)
proj = torch.randn(80, d, device=dev, generator=torch.Generator(device=dev).manual_seed(7)) / 80 ** 0.5
inputs = feats @ proj
lens = [500] * a.utts
with torch.no_grad():
    enc = model.encoder(inputs[:1], None)
    dec = model.decoder(torch.zeros(1, 1, dtype=torch.long, device=dev))
    z = model.joint(enc, dec)[0, :, 0, :].float()
    margin = z[:, 1:].max(dim=1).values - z[:, 0]
    model.joint.project_layer.bias[0] += torch.quantile(margin, 1.0 - a.emit_rate)
    enc_states = model.encoder(inputs, None)
    hyps = model.decode_batch(enc_states, lens, block=a.block)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.passes):
        hyps = model.decode_batch(enc_states, lens, block=a.block)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.passes
steps = max(len(h) for h in hyps)
print("decode_batch: %.1f ms per pass, %d symbol steps (longest hypothesis), %.3f ms per step, mean length %.1f" %
      (1e3 * dt, steps, 1e3 * dt / steps, sum(len(h) for h in hyps) / len(hyps)))
