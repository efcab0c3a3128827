#!/usr/bin/env python3
"""hashes of the token lists of decode() per utterance and of decode_batch(), per process: which of the two changes from run to run?"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))
import torch
import bench_decode
out, model, inputs, lens, hyps = bench_decode.run(32, 500, 0.1, "fp32")
h = lambda x: hashlib.sha256(repr(x).encode()).hexdigest()[:8]
with torch.no_grad():
    enc = model.encoder(inputs, None)
    single = [model.decode(enc[b], lens[b]) for b in range(32)]
    model.config["decode_graphs"] = False
    single_eager = [model.decode(enc[b], lens[b]) for b in range(32)]
    batched_eager = model.decode_batch(enc, lens)
    model.config["decode_graphs"] = True
    batched = model.decode_batch(enc, lens)
print("run() batched", h(hyps), "| batched", h(batched), "| batched eager", h(batched_eager), "| single", h(single), "| single eager", h(single_eager),
      "| symbols", sum(map(len, hyps)), sum(map(len, batched)), sum(map(len, batched_eager)), sum(map(len, single)), sum(map(len, single_eager)))
bad = [b for b in range(32) if batched[b] != single_eager[b]]
print("   utterances where batched != single eager:", bad, " batched eager != single eager:", [b for b in range(32) if batched_eager[b] != single_eager[b]],
      " single != single eager:", [b for b in range(32) if single[b] != single_eager[b]])
for b in range(32):
    if hyps[b] != single_eager[b]:
        x, y = hyps[b], single_eager[b]
        k = next((i for i in range(min(len(x), len(y))) if x[i] != y[i]), min(len(x), len(y)))
        print("   run() batched, utt %d: first difference at symbol %d of %d / %d: %s vs %s" % (b, k, len(x), len(y), x[k:k + 3], y[k:k + 3]))
