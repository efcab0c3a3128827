#!/usr/bin/env python3
"""(variant: which ENCODER's bf16 operands carry the loss error?  tools/debug/loss_error_which_encoder.py)
Round 5: does the second bf16 term of the encoder WEIGHTS (ttmi_set_option(13, 1): all four forward GEMMs of a layer, one launch each) bring the timed
bf16 mode's loss under 1e-4 at every training state?  The C2 model, bench.py's own SGD loop (option 13 off while training: ONE trajectory), and at
the states after `--states` steps the per-utterance costs of `--n` utterances in eval mode - the timed form (bf16 encoders, exp-domain joint + loss)
with option 13 = 0 and = 1 on the SAME weights - against the float64 oracle.  Prints worst / mean relative error per state and option.

    python tools/debug/loss_error_stats.py --n 6 --states 0,5,10,15,25,40
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=6)
ap.add_argument("--states", default="0,5,10,15,25,40")
args = ap.parse_args()
os.environ["TTMI_PRECISION"] = "bf16"
import bench
from oracle import tt_oracle as O
from oracle.rnnt_c import rnnt_loss_c
from tt.model import Transducer, _JointLossFn
from ttmi import ops
from ttmi.train import FlatModel, FusedOptimizer, GradSync

dev = torch.device("cuda", 0)
torch.manual_seed(1)
model = Transducer(bench.c2_config()).to(dev).train()
PREC = {"audio": "bf16", "label": "bf16"}


def _with_precision(module, which):
    inner = module.forward

    def fwd(*a, **k):
        prev = os.environ["TTMI_PRECISION"]
        os.environ["TTMI_PRECISION"] = PREC[which]
        try:
            return inner(*a, **k)
        finally:
            os.environ["TTMI_PRECISION"] = prev
    module.forward = fwd


_with_precision(model.encoder, "audio")
_with_precision(model.decoder, "label")
flat = FlatModel(model)
flat.enable_grouped_wgrads()
flat.enable_shadows()
sync = GradSync(flat)
opt = FusedOptimizer(flat, kind="sgd", lr=0.00025, momentum=0.9, max_grad_norm=200.0)
B, T, U, V, d = 32, 500, 50, 4334, 512
g = torch.Generator(device=dev).manual_seed(1234)
feats = torch.randn(B, T, 80, device=dev, generator=g)
proj = torch.randn(80, d, device=dev, generator=torch.Generator(device=dev).manual_seed(7)) / 80 ** 0.5
targets = torch.randint(1, V, (B, U), device=dev, generator=g)
ilen = torch.full((B,), T, dtype=torch.int32, device=dev)
tlen = torch.full((B,), U, dtype=torch.int32, device=dev)
inputs = (feats.reshape(-1, 80) @ proj).reshape(B, T, d).contiguous()


def evaluate(step):
    n = args.n
    model.eval()
    x, y, il, tl = inputs[:n], targets[:n], ilen[:n], tlen[:n]
    sd64 = {k: (v.detach().cpu().numpy().astype(np.float64) if v.dtype == torch.float32 else v.detach().cpu().numpy()) for k, v in model.state_dict().items()}
    z64, _ = O.transducer_fwd(x.cpu().numpy().astype(np.float64), y.cpu().numpy(), sd64)
    want = rnnt_loss_c(z64.astype(np.float32), y.cpu().numpy(), il.cpu().numpy(), tl.cpu().numpy(), want_grad=False)[1].astype(np.float64)
    row = {"step": step, "cost": float(want.mean())}
    for o13, (pa, pl) in enumerate((("bf16", "bf16"), ("bf16", "bf16x3"), ("bf16x3", "bf16"))):
        with torch.no_grad():
            PREC["audio"], PREC["label"] = pa, pl
            enc_s, dec_s = model._encode(x, y)
            PREC["audio"] = PREC["label"] = "bf16"
            j = model.joint
            stt = j.exp_shift_state(dev)
            if not stt.valid:
                stt.set(0.0)
            costs = _JointLossFn.apply(enc_s, dec_s, j.forward_layer.weight, j.forward_layer.bias, j.project_layer.weight, j.project_layer.bias,
                                       y.int().contiguous(), il, tl, 1, n, "none", stt, False)
            torch.cuda.synchronize()
            z, _ = O.joint_fwd(enc_s.double().cpu().numpy(), dec_s.double().cpu().numpy(), sd64)
        c_enc = rnnt_loss_c(z.astype(np.float32), y.cpu().numpy(), il.cpu().numpy(), tl.cpu().numpy(), want_grad=False)[1].astype(np.float64)
        e = np.abs(costs.double().cpu().numpy() - want) / want
        ee = np.abs(c_enc - want) / want
        row["opt%d" % o13] = {"timed_worst": float(e.max()), "timed_mean": float(e.mean()), "enc_only_worst": float(ee.max()), "enc_only_mean": float(ee.mean())}
    model.train()
    print("step %3d  cost %8.1f | both bf16: timed worst %.2e mean %.2e (encoders alone %.2e) | label encoder bf16x3: %.2e / %.2e (%.2e) | audio encoder bf16x3: %.2e / %.2e (%.2e)"
          % (step, row["cost"], row["opt0"]["timed_worst"], row["opt0"]["timed_mean"], row["opt0"]["enc_only_worst"],
             row["opt1"]["timed_worst"], row["opt1"]["timed_mean"], row["opt1"]["enc_only_worst"],
             row["opt2"]["timed_worst"], row["opt2"]["timed_mean"], row["opt2"]["enc_only_worst"]), flush=True)
    return row


states = sorted(int(v) for v in args.states.split(","))
rows = []
done = 0
for s in states:
    for _ in range(s - done):
        flat.zero_grad()
        sync.start_step()
        loss = model.loss(inputs, ilen, targets, tlen, exp_domain=True)
        loss.backward()
        sync.finish()
        opt.step()
    done = s
    rows.append(evaluate(s))
print("worst over %d states x %d utterances: both bf16 %.2e, label encoder bf16x3 %.2e, audio encoder bf16x3 %.2e"
      % (len(rows), args.n, *(max(r["opt%d" % k]["timed_worst"] for r in rows) for k in range(3))))
