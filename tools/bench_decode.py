#!/usr/bin/env python3
"""Greedy decode (tt/model.py:70-108) of the C2 model on synthetic utterances: utterances/s, frames/s, host syncs,
through the product path only.  `python bench.py --mode decode` runs this and times the oracle's frame-by-frame loop on the host
beside it (token check + CPU time); the full-size token parity test is tests/test_decode_gpu.py.

A random-init model emits a symbol on (nearly) every frame, the worst case for the reference's full-history label-encoder
recompute; `--emit-rate r` raises the joint's blank bias so that about r of the frames emit (r = 0.1 ~ U = 50 symbols per T = 500).

    python tools/bench_decode.py [--utts 8] [--T 500] [--emit-rate 0.1] [--precision fp32]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))


def run(utts=8, T=500, emit_rate=0.1, precision="fp32", block=None):
    """returns (result dict, model, inputs [utts,T,d], lengths, hypotheses)"""
    args = argparse.Namespace(utts=utts, T=T, emit_rate=emit_rate, precision=precision, block=block)
    os.environ["TTMI_PRECISION"] = args.precision
    from bench import c2_config
    from tt.model import Transducer

    dev = torch.device("cuda", 0)
    cfg = c2_config()
    torch.manual_seed(1)
    model = Transducer(cfg).to(dev).eval()
    if os.environ.get("TTMI_DECODE_BATCH_GRAPHS") in ("0", "1"):
        model.config["decode_batch_graphs"] = os.environ["TTMI_DECODE_BATCH_GRAPHS"] == "1"      # (A/B: label-encoder graphs while the batch is complete - the default - or eager launches throughout)
    if os.environ.get("TTMI_DECODE_GRAPHS") == "0":
        model.config["decode_graphs"] = False                        # (debugging: eager label-encoder launches)
    d, V = cfg["enc"]["d_model"], cfg["vocab_size"]
    g = torch.Generator(device=dev).manual_seed(1234)
    feats = torch.randn(args.utts, args.T, 80, device=dev, generator=g)
    proj = torch.randn(80, d, device=dev, generator=torch.Generator(device=dev).manual_seed(7)) / 80 ** 0.5
    inputs = feats @ proj
    import hashlib
    inputs_sha = hashlib.sha256(inputs.float().cpu().numpy().tobytes()).hexdigest()[:12]      # (torch's GEMM: not bit-reproducible from process to process)
    lens = [args.T] * args.utts

    with torch.no_grad():
        if args.emit_rate < 1.0:
            # blank bias such that the blank wins on about (1 - emit_rate) of the frames against the start-token label state
            enc = model.encoder(inputs[:1], None)
            dec = model.decoder(torch.zeros(1, 1, dtype=torch.long, device=dev))
            z = model.joint(enc, dec)[0, :, 0, :].float()
            margin = z[:, 1:].max(dim=1).values - z[:, 0]
            model.joint.project_layer.bias[0] += torch.quantile(margin, 1.0 - args.emit_rate)

        model.recognize(inputs, lens)                               # warm-up (also captures the label-encoder graphs these histories need)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        enc_states = model.encoder(inputs, None)
        torch.cuda.synchronize()
        t_enc = time.perf_counter() - t0
        t0 = time.perf_counter()
        hyps1 = [model.decode(enc_states[b], lens[b], block=args.block or 64) for b in range(args.utts)]
        torch.cuda.synchronize()
        t_dec1 = time.perf_counter() - t0
        # the batched form (what recognize() runs): every utterance in lockstep over symbol steps, one label-encoder call per step
        blocks = [0]
        from ttmi import ops
        orig = ops.greedy_advance
        ops.greedy_advance = lambda *a, **k: (blocks.__setitem__(0, blocks[0] + 1), orig(*a, **k))[1]
        t0 = time.perf_counter()
        hyps = model.decode_batch(enc_states, lens, block=args.block)
        torch.cuda.synchronize()
        t_dec = time.perf_counter() - t0
        ops.greedy_advance = orig
    nsym = sum(len(h) for h in hyps)
    out = {"workload": "greedy decode, C2 model (12/6 layers, V=4334), %d utt x T=%d, %s, emit rate target %.2f"
                       % (args.utts, args.T, args.precision, args.emit_rate),
           "inputs_sha": inputs_sha, "debug_sha": {"enc_states": hashlib.sha256(enc_states.float().cpu().numpy().tobytes()).hexdigest()[:8],
                                                   "blank_bias": float(model.joint.project_layer.bias[0].detach()),
                                                   "one_at_a_time": hashlib.sha256(repr(hyps1).encode()).hexdigest()[:8],
                                                   "batched": hashlib.sha256(repr(hyps).encode()).hexdigest()[:8]},
           "utt_per_s": round(args.utts / (t_enc + t_dec), 3), "frames_per_s": round(args.utts * args.T / (t_enc + t_dec), 1),
           "encoder_ms": round(1e3 * t_enc, 2), "decode_ms_per_utt": round(1e3 * t_dec / args.utts, 2),
           "symbols_per_utt": round(nsym / args.utts, 1), "ms_per_symbol_step": round(1e3 * t_dec / max(max(len(h) for h in hyps), 1), 3),
           "host_syncs_per_batch": blocks[0], "host_syncs_per_utt": round(blocks[0] / args.utts, 1),
           "decode": "Transducer.decode_batch: the batch in lockstep over symbol steps (one joint call per scanned block, one label-encoder call per step: a graph replay while the batch is complete, eager launches on the shrinking rows after)",
           "one_utterance_at_a_time": {"utt_per_s": round(args.utts / (t_enc + t_dec1), 3), "decode_ms_per_utt": round(1e3 * t_dec1 / args.utts, 2),
                                       "tokens_identical_to_batched": hyps1 == hyps,
                                       "host_syncs_per_utt_approx": round((nsym + args.utts * -(-args.T // (args.block or 64))) / args.utts, 1)}}

    return out, model, inputs, lens, hyps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=8)
    ap.add_argument("--T", type=int, default=500)
    ap.add_argument("--emit-rate", type=float, default=0.1)
    ap.add_argument("--precision", default="fp32", choices=["bf16", "fp32"])
    ap.add_argument("--block", type=int, default=None, help="frames per joint call (default: 64, or what fits one round of projection tiles)")
    a = ap.parse_args()
    print(json.dumps(run(a.utts, a.T, a.emit_rate, a.precision, a.block)[0]), flush=True)


if __name__ == "__main__":
    main()
