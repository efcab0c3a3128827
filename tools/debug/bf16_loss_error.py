"""Where does the bf16 step's loss error against the oracle come from?  C2 model, `--steps` SGD steps as bench.py
runs them, then per-utterance costs of a B=2 sample (eval mode) through every combination of {fp32, bf16} encoders x {fp32, bf16
two-call, bf16 exp-domain} joint + loss, each against the float64 oracle.

    python tools/debug/bf16_loss_error.py --steps 50
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

np.set_printoptions(formatter={"float_kind": lambda v: "%.3g" % v})      # (array2string would print 1.6e-04 as 0.)

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--n", type=int, default=2)
args = ap.parse_args()

os.environ["TTMI_PRECISION"] = "bf16"
import bench
from conftest import rel_err
from oracle import tt_oracle as O
from oracle.rnnt_c import rnnt_loss_c
from tt.model import Transducer, _JointLossFn
from ttmi import ops
from ttmi.train import FlatModel, FusedOptimizer, GradSync
from warprnnt_pytorch import RNNTLoss

OPT13 = 0
for kv in filter(None, os.environ.get("TTMI_OPTIONS", "").split(",")):
    if kv.split("=")[0] == "13":
        OPT13 = int(kv.split("=")[1])
dev = torch.device("cuda", 0)
cfg = bench.c2_config()
torch.manual_seed(1)
model = Transducer(cfg).to(dev).train()
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


def report(tag):
    n = args.n
    model.eval()
    x, y, il, tl = inputs[:n], targets[:n], ilen[:n], tlen[:n]
    sd64 = {k: (v.detach().cpu().numpy().astype(np.float64) if v.dtype == torch.float32 else v.detach().cpu().numpy())
            for k, v in model.state_dict().items()}
    z64, cache = O.transducer_fwd(x.cpu().numpy().astype(np.float64), y.cpu().numpy(), sd64)
    enc64, dec64 = cache[2]["enc"], cache[2]["dec"]
    _, want, _ = rnnt_loss_c(z64.astype(np.float32), y.cpu().numpy(), il.cpu().numpy(), tl.cpu().numpy(), want_grad=False)
    want = want.astype(np.float64)
    print("== %s: oracle costs %s   logits: max %.2f  blank mean %.2f  lse-ish std %.3f" % (tag, want, z64.max(), z64[..., 0].mean(), z64.std()))
    states = {}
    with torch.no_grad():
        for prec in ("fp32", "bf16"):
            os.environ["TTMI_PRECISION"] = prec
            states[prec] = model._encode(x, y)
            torch.cuda.synchronize()
            e, dcd = states[prec]
            print("   %s encoders: enc_state rel %.2e  dec_state rel %.2e" % (prec, rel_err(e.cpu().numpy(), enc64), rel_err(dcd.cpu().numpy(), dec64)))
        j = model.joint
        w = (j.forward_layer.weight, j.forward_layer.bias, j.project_layer.weight, j.project_layer.bias)
        lab = y.int().contiguous()
        for ep in ("fp32", "bf16"):
            enc_s, dec_s = states[ep]
            for jp, exp in (("fp32", False), ("bf16", False), ("bf16", True)):
                os.environ["TTMI_PRECISION"] = jp
                ops.weights_fresh()
                chunk = n
                stt = None
                if exp:
                    stt = j.exp_shift_state(enc_s.device)
                    if not stt.valid:
                        stt.set(0.0)
                costs = _JointLossFn.apply(enc_s, dec_s, *w, lab, il, tl, 1 if jp == "bf16" else 0, chunk, "none", stt, False)
                torch.cuda.synchronize()
                c = costs.double().cpu().numpy()
                print("   enc %s / joint+loss %s%s: rel err per utt %s  (signed abs %s)" %
                      (ep, jp, " exp" if exp else "", np.array2string(np.abs(c - want) / want, precision=2), np.array2string(c - want, precision=3)))
        # the oracle's joint + lattice on the bf16 encoders' states: how much of the error the encoder states carry
        for ep in ("fp32", "bf16"):
            enc_s, dec_s = states[ep]
            z, _ = O.joint_fwd(enc_s.double().cpu().numpy(), dec_s.double().cpu().numpy(), sd64)
            c = rnnt_loss_c(z.astype(np.float32), y.cpu().numpy(), il.cpu().numpy(), tl.cpu().numpy(), want_grad=False)[1].astype(np.float64)
            print("   enc %s / ORACLE joint+loss: rel err %s (signed abs %s)" % (ep, np.array2string(np.abs(c - want) / want, precision=2), np.array2string(c - want, precision=3)))
        # the same state through the two-term weight forms of the encoders' forward GEMMs (option 13: 2 = o_net / CoreNet.3, 1 = all four)
        os.environ["TTMI_PRECISION"] = "bf16"
        for v in (2, 1):
            ops.set_option(13, v)
            enc_s, dec_s = model._encode(x, y)
            torch.cuda.synchronize()
            ops.set_option(13, OPT13)
            z, _ = O.joint_fwd(enc_s.double().cpu().numpy(), dec_s.double().cpu().numpy(), sd64)
            c = rnnt_loss_c(z.astype(np.float32), y.cpu().numpy(), il.cpu().numpy(), tl.cpu().numpy(), want_grad=False)[1].astype(np.float64)
            print("   enc bf16 option 13=%d / ORACLE joint+loss: rel err %s (signed abs %s)  enc_state rel %.2e  dec_state rel %.2e" %
                  (v, np.array2string(np.abs(c - want) / want, precision=2), np.array2string(c - want, precision=3),
                   rel_err(enc_s.cpu().numpy(), enc64), rel_err(dec_s.cpu().numpy(), dec64)))
    os.environ["TTMI_PRECISION"] = "bf16"
    model.train()


report("step 0")
done = 0
for stop in sorted({10, 25, 50, 80, 120, args.steps}):
    if stop > args.steps:
        break
    while done < stop:
        flat.zero_grad()
        sync.start_step()
        loss = model.loss(inputs, ilen, targets, tlen, exp_domain=True)
        loss.backward()
        sync.finish()
        opt.step()
        done += 1
    torch.cuda.synchronize()
    print("after %d steps: training loss %.3f" % (done, float(loss)))
    report("step %d" % done)
