#!/usr/bin/env python3
"""Is Transducer.decode_batch deterministic, and where does it differ from decode() per utterance?  (fp32, C2 model, 32 utterances)"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))
import torch
import bench_decode
out, model, inputs, lens, hyps = bench_decode.run(int(os.environ.get("UTTS", 32)), 500, 0.1, "fp32")
print("run(): utt/s", out["utt_per_s"], "symbols", out["symbols_per_utt"], "batched == single inside run():", out["one_utterance_at_a_time"]["tokens_identical_to_batched"])
print("inputs sha", hashlib.sha256(inputs.float().cpu().numpy().tobytes()).hexdigest()[:12], "bias0", float(model.joint.project_layer.bias[0]))
with torch.no_grad():
    enc = model.encoder(inputs, None)
    print("enc sha", hashlib.sha256(enc.float().cpu().numpy().tobytes()).hexdigest()[:12])
    singles = [[model.decode(enc[b], lens[b]) for b in range(len(lens))] for _ in range(2)]
    print("single deterministic:", singles[0] == singles[1])
    batched = [model.decode_batch(enc, lens) for _ in range(3)]
    print("batched now == run()'s batched:", batched[0] == hyps, " single now == batched from run():", singles[0] == hyps)
    print("batched deterministic:", batched[0] == batched[1] == batched[2], " batched == single:", batched[0] == singles[0])
    for b in range(len(lens)):
        if batched[0][b] != singles[0][b]:
            x, y = batched[0][b], singles[0][b]
            k = next((i for i in range(min(len(x), len(y))) if x[i] != y[i]), min(len(x), len(y)))
            print("  utt %d: first difference at symbol %d (%s vs %s), lengths %d / %d" % (b, k, x[k:k + 2], y[k:k + 2], len(x), len(y)))
