"""alpha / beta lattice alone: ttmi_rnnt_loss_fwd on logits with a tiny vocabulary, so that the
log-sum-exp pass is negligible and the launch is the T + U dependent diagonals.  Times both kernels (ttmi_set_option(9, v)):

    python tools/bench_lattice.py            # C2 (B=32, T=500, U=50), C5 chunk (B=4, T=2000, U=200), C5 batch (B=8)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))
import torch

from ttmi import ops


def run(B, T, U, reps=20, V=8):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(B, T, U + 1, V, device="cuda", generator=g)
    y = torch.randint(1, V, (B, U), device="cuda", generator=g, dtype=torch.int32)
    tl = torch.full((B,), T, dtype=torch.int32, device="cuda")
    ul = torch.full((B,), U, dtype=torch.int32, device="cuda")
    ws = ops.rnnt_workspace(B, T, U + 1, x.device)
    out = {}
    for ver, name in ((1, "one wave per utterance"), (0, "workgroup + LDS frontier")):
        ops.set_option(9, ver)
        for _ in range(3):
            c = ops.rnnt_loss_fwd(x, y, tl, ul, 0, ws)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            c = ops.rnnt_loss_fwd(x, y, tl, ul, 0, ws)
        e1.record()
        torch.cuda.synchronize()
        out[name] = (e0.elapsed_time(e1) / reps, c.clone())
    ops.set_option(9, 0)
    (t_old, c_old), (t_new, c_new) = out["one wave per utterance"], out["workgroup + LDS frontier"]
    same = bool(torch.equal(c_old, c_new))
    D = T + U
    print("B=%d T=%d U=%d: one-wave kernel %.3f ms (%.2f us per diagonal), LDS-frontier kernel %.3f ms (%.2f us per diagonal), costs bit-identical: %s"
          % (B, T, U, t_old, 1e3 * t_old / D, t_new, 1e3 * t_new / D, same))


if __name__ == "__main__":
    run(32, 500, 50)
    run(4, 2000, 200)
    run(8, 2000, 200)
    run(2, 300, 600)
