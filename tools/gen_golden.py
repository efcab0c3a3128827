#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE model on CPU.

Runs only in the build container (needs /root/reference, which does not travel
to the GPU box).  Nothing from the reference is copied: the script imports
`tt.*` from /root/reference, pushes seeded inputs through it and stores the
tensors (state_dict, inputs, outputs, gradients, greedy tokens) as fixtures.

`tt/utils.py` imports librosa and editdistance at module top (tt/utils.py:5,7);
both are absent here and unused on the hot path, so two empty placeholder
modules are registered before the import.

The RNN-T loss module (`warprnnt_pytorch`) is not installed, so the loss and the
gradients flowing from it are produced by an independent float64 *autograd*
lattice written below (plain recursion + torch.logaddexp; no hand-written
backward), applied to the reference model's own logits.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REF = "/root/reference"
sys.path.insert(0, REF)
for _m in ("librosa", "editdistance"):
    sys.modules.setdefault(_m, types.ModuleType(_m))

import numpy as np
import torch
import yaml

from tt.utils import AttrDict, context_mask, look_ahead_mask   # noqa: E402
from tt.model import Transducer                                  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def tiny_config(k_enc, k_dec):
    cfg = yaml.load(open(os.path.join(REF, "config", "aishell.yaml")), Loader=yaml.FullLoader)
    m = cfg["model"]
    for side, k in (("enc", k_enc), ("dec", k_dec)):
        m[side].update(n_layer=2, d_model=96, n_head=4, d_head=24, d_inner=160)
    m["enc"]["max_input_length"] = k_enc
    m["dec"]["max_target_length"] = k_dec
    m["joint"].update(input_size=192, inner_size=80)
    m["vocab_size"] = 48
    m["dropout"] = 0.0
    return AttrDict(m)


def lattice_cost(logits, labels, T, U):
    """float64 autograd RNN-T negative log-likelihood of one utterance."""
    lp = torch.log_softmax(logits[:T, :U + 1].double(), -1)
    alpha = [[None] * (U + 1) for _ in range(T)]
    alpha[0][0] = lp.new_zeros(())
    for t in range(T):
        for u in range(U + 1):
            if t == 0 and u == 0:
                continue
            terms = []
            if t > 0:
                terms.append(alpha[t - 1][u] + lp[t - 1, u, 0])
            if u > 0:
                terms.append(alpha[t][u - 1] + lp[t, u - 1, labels[u - 1]])
            alpha[t][u] = terms[0] if len(terms) == 1 else torch.logaddexp(terms[0], terms[1])
    return -(alpha[T - 1][U] + lp[T - 1, U, 0])


def chunk_mask(L, chunk, left):
    i = torch.arange(L)[:, None]
    j = torch.arange(L)[None, :]
    lo = (i // chunk) * chunk - left
    hi = (i // chunk + 1) * chunk - 1
    return ((j < lo) | (j > hi)).float()


def run_case(name, k_enc, k_dec, B=2, T=40, U=6, seed=1234):
    cfg = tiny_config(k_enc, k_dec)
    torch.manual_seed(seed)
    model = Transducer(cfg).eval()
    gen = torch.Generator().manual_seed(seed + 1)
    inputs = torch.randn(B, T, 96, generator=gen)
    targets = torch.randint(1, 48, (B, U), generator=gen)
    out = {}
    for k, v in model.encoder.state_dict().items():
        out["sd/encoder." + k] = v.numpy()
    for k, v in model.decoder.state_dict().items():
        out["sd/decoder." + k] = v.numpy()
    for k, v in model.joint.state_dict().items():
        out["sd/joint." + k] = v.numpy()
    out["inputs"] = inputs.numpy()
    out["targets"] = targets.numpy()

    # ---- per-layer encoder outputs, no mask
    x = inputs.transpose(0, 1)
    for i, layer in enumerate(model.encoder.layers):
        x = layer(x, None)
        out["enc_layer%d" % i] = x.transpose(0, 1).detach().numpy()

    # ---- full forward + loss (full lengths and ragged lengths) + all grads
    for tag, act_lens, lab_lens in (("full", [T] * B, [U] * B), ("ragged", [T, T - 7], [U, U - 2])):
        model.zero_grad()
        inp = inputs.clone().requires_grad_(True)
        logits = model(inp, targets)
        costs = torch.stack([lattice_cost(logits[b], targets[b].tolist(), act_lens[b], lab_lens[b]) for b in range(B)])
        loss = costs.sum() / B
        loss.backward()
        out[tag + "/act_lens"] = np.array(act_lens, dtype=np.int32)
        out[tag + "/label_lens"] = np.array(lab_lens, dtype=np.int32)
        out[tag + "/costs"] = costs.detach().numpy()
        out[tag + "/loss"] = loss.detach().numpy()
        out[tag + "/dinputs"] = inp.grad.numpy()
        if tag == "full":
            out["logits"] = logits.detach().numpy()
        for pre, mod in (("encoder.", model.encoder), ("decoder.", model.decoder), ("joint.", model.joint)):
            for k, p in mod.named_parameters():
                out["%s/grad/%s%s" % (tag, pre, k)] = p.grad.numpy().copy()

    # ---- encoder under streaming masks (forward + input/param grads through a fixed cotangent)
    cot = torch.randn(B, T, 96, generator=gen)
    out["enc_cotangent"] = cot.numpy()
    masks = {
        "band_10_2": context_mask(inputs, 10, 2),
        "left_8_0": context_mask(inputs, 8, 0),
        "chunk_8_16": chunk_mask(T, 8, 16),
    }
    for mname, m2 in masks.items():
        model.zero_grad()
        inp = inputs.clone().requires_grad_(True)
        y = model.encoder(inp, m2[:, :, None])
        (y * cot).sum().backward()
        out["mask/%s/mask" % mname] = m2.numpy().astype(np.uint8)
        out["mask/%s/enc_out" % mname] = y.detach().numpy()
        out["mask/%s/dinputs" % mname] = inp.grad.numpy()
        for k, p in model.encoder.named_parameters():
            if k.startswith("layers.0."):      # masks only touch attention; one layer's grads suffice
                out["mask/%s/grad/encoder.%s" % (mname, k)] = p.grad.numpy().copy()

    # ---- label encoder alone with the look-ahead mask and WITHOUT it (decode path)
    tg = torch.nn.functional.pad(targets, [1, 0, 0, 0], value=0)
    out["dec_masked"] = model.decoder(tg, look_ahead_mask(tg)[:, :, None]).detach().numpy()
    out["dec_unmasked"] = model.decoder(tg).detach().numpy()

    # ---- greedy decode
    with torch.no_grad():
        lens = torch.tensor([T, T - 7])
        hyp = model.recognize(inputs, lens)
    out["greedy/lens"] = lens.numpy()
    for b, h in enumerate(hyp):
        out["greedy/tokens%d" % b] = np.array(h, dtype=np.int64)

    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print("%s: %d arrays, %.1f KB; loss(full)=%.8f logits.sum=%.5f greedy=%s" % (
        path, len(out), os.path.getsize(path) / 1024, float(out["full/loss"]), float(out["logits"].sum()),
        [len(h) for h in hyp]))


if __name__ == "__main__":
    torch.set_num_threads(4)
    run_case("tiny_klong", k_enc=64, k_dec=12)     # L <= K branch (tt/transformer.py:133-135)
    run_case("tiny_kshort", k_enc=16, k_dec=4)     # L >  K branch (tt/transformer.py:128-132)
