#!/usr/bin/env python3
"""Round 6: what BASELINE asks of the timed bf16 mode is the LOSS - train.py:53's batch mean - within 1e-4 of the reference
arithmetic.  Earlier rounds bounded the worst single utterance of a B = 2 ... 6 sample (3e-4).  This tool measures the batch mean itself: the C2 model,
bench.py's own SGD loop at B = 32 (bf16 mode, exp-domain loss, dropout on), and at the states after `--states` steps, in eval mode on the SAME weights, the
loss of the whole batch in the timed form against TTMI_PRECISION=fp32 (held to the float64 oracle at <= 1e-6 by tests/test_configs_gpu.py): signed relative
error of the batch mean, worst and mean |error| per utterance, for `--seeds` model / data seeds.

    python tools/debug/loss_error_batch_mean.py --states 0,2,5,8,10,12,15,20,25,40 --seeds 1,2
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--states", default="0,2,5,8,10,12,15,20,25,40")
ap.add_argument("--seeds", default="1,2")
ap.add_argument("--modes", default="bf16", help="comma list of timed modes to score against fp32 (bf16, bf16x3)")
ap.add_argument("--workload", default="c2", choices=["c2", "c4-band", "c4-chunk", "c5"], help="BASELINE configuration: model and batch shape as bench.py --workload")
args = ap.parse_args()
os.environ["TTMI_PRECISION"] = "bf16"
import bench
from tt.model import Transducer
from ttmi.train import FlatModel, FusedOptimizer, GradSync

dev = torch.device("cuda", 0)
B, T, U, V, d = 32, 500, 50, 4334, 512
if args.workload.startswith("c4"):
    V = 6485
elif args.workload == "c5":
    B, T, U = 8, 2000, 200
states = sorted(int(v) for v in args.states.split(","))
out = []
for seed in (int(v) for v in args.seeds.split(",")):
    os.environ["TTMI_PRECISION"] = "bf16"
    torch.manual_seed(seed)
    model = Transducer(bench.c4_config(args.workload[3:]) if args.workload.startswith("c4") else bench.c2_config()).to(dev).train()
    flat = FlatModel(model)
    flat.enable_grouped_wgrads()
    flat.enable_shadows()
    sync = GradSync(flat)
    opt = FusedOptimizer(flat, kind="sgd", lr=0.00025, momentum=0.9, max_grad_norm=200.0)
    g = torch.Generator(device=dev).manual_seed(1233 + seed)
    feats = torch.randn(B, T, 80, device=dev, generator=g)
    proj = torch.randn(80, d, device=dev, generator=torch.Generator(device=dev).manual_seed(7)) / 80 ** 0.5
    targets = torch.randint(1, V, (B, U), device=dev, generator=g)
    ilen = torch.full((B,), T, dtype=torch.int32, device=dev)
    tlen = torch.full((B,), U, dtype=torch.int32, device=dev)
    inputs = (feats.reshape(-1, 80) @ proj).reshape(B, T, d).contiguous()

    def costs_in(mode):
        os.environ["TTMI_PRECISION"] = mode
        with torch.no_grad():
            c = model.loss(inputs, ilen, targets, tlen, reduction="none", exp_domain=(mode == "bf16"), check_lengths=False)
        torch.cuda.synchronize()
        os.environ["TTMI_PRECISION"] = "bf16"
        return c.double().cpu()

    done = 0
    for s in states:
        while done < s:
            flat.zero_grad()
            sync.start_step()
            loss = model.loss(inputs, ilen, targets, tlen, exp_domain=True)
            loss.backward()
            sync.finish()
            opt.step()
            done += 1
        torch.cuda.synchronize()
        model.eval()
        ref = costs_in("fp32")
        for mode in args.modes.split(","):
            got = costs_in(mode)
            rel = (got - ref) / ref
            row = dict(seed=seed, step=s, mode=mode, loss_fp32=float(ref.mean()), batch_mean_rel=float((got.mean() - ref.mean()) / ref.mean()),
                       utt_worst=float(rel.abs().max()), utt_mean_abs=float(rel.abs().mean()), positive=int((rel > 0).sum()))
            out.append(row)
            print("seed %d step %3d %-6s loss %9.2f | batch mean %+.2e | per utterance: worst %.2e, mean |e| %.2e, %2d of %d above" %
                  (seed, s, mode, row["loss_fp32"], row["batch_mean_rel"], row["utt_worst"], row["utt_mean_abs"], row["positive"], B), flush=True)
        model.train()
    del model, flat, sync, opt
    torch.cuda.empty_cache()
for mode in args.modes.split(","):
    rows = [r for r in out if r["mode"] == mode]
    print("%s: worst |batch mean error| over %d states x seeds: %.2e; worst single utterance %.2e" %
          (mode, len(rows), max(abs(r["batch_mean_rel"]) for r in rows), max(r["utt_worst"] for r in rows)))
print(json.dumps(out))
