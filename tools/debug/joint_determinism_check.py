#!/usr/bin/env python3
"""Round 6: is the joint forward (bf16 mode, ops.joint_fwd: H = tanh(PE + PD + b), logits = H Wp^T + bp) a pure function of its inputs with the label states' second bf16 term
(option 19) on?  Calls it repeatedly on the same inputs, with the scratch arena poisoned in between, and compares logits and the PD rows the call left in the arena."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))
os.environ["TTMI_PRECISION"] = "bf16"
import torch
from ttmi import ops
dev = torch.device("cuda", 0)
B, T, U1, d, J, V = 8, 500, 51, 512, 1024, 4334
g = torch.Generator(device=dev).manual_seed(3)
enc = torch.randn(B, T, d, device=dev, generator=g); dec = torch.randn(B, U1, d, device=dev, generator=g)
wf = torch.randn(J, 2 * d, device=dev, generator=g) * 0.03; bf = torch.randn(J, device=dev, generator=g) * 0.1
wp = torch.randn(V, J, device=dev, generator=g) * 0.03; bp = torch.randn(V, device=dev, generator=g) * 0.1
al4 = lambda n: (n + 3) & ~3
M = B * T * U1
for opt in (1, 0):
    ops.set_option(19, opt)
    outs, pds = [], []
    for rep in range(4):
        lg, ctx = ops.joint_fwd(enc, dec, wf, bf, wp, bp, 1)
        torch.cuda.synchronize()
        arena = next(iter(ops._ws_cache.values()))
        off = al4(M * J) + al4(B * T * J)
        pds.append(arena[off:off + B * U1 * J].clone())
        outs.append(lg.float().clone())
        if rep == 1:
            arena.fill_(float("nan"))          # poison: anything read before it is written shows
    want = dec.reshape(-1, d).double() @ wf[:, d:].double().t()
    e = [float(((p.double().reshape(B * U1, J) - want).abs().max() / want.abs().max())) for p in pds]
    print("option 19 = %d: logits equal across calls %s; PD equal across calls %s; PD max err vs float64 %s; NaN in logits %s"
          % (opt, [bool(torch.equal(outs[0], o)) for o in outs[1:]], [bool(torch.equal(pds[0], p)) for p in pds[1:]], ["%.2e" % x for x in e], [bool(torch.isnan(o).any()) for o in outs]))
