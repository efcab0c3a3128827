#!/usr/bin/env python3
"""In-kernel cycle stamps of attn_dqde_kernel (library built with `make EXTRA=-DTTMI_STAMPS`, TTMI_PG_DEBUG=64): wave 0 of workgroup 5."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))
os.environ.setdefault("TTMI_PRECISION", "bf16")
os.environ["TTMI_PG_DEBUG"] = "64"
import torch
from tt.encoder import BaseEncoder
from ttmi import ops
import ttmi
B, L = int(os.environ.get("B", 32)), int(os.environ.get("L", 500))
torch.manual_seed(0)
layer = BaseEncoder(k_len=410, n_head=8, d_model=512, d_head=64, d_inner=1024, dropout=0.0).cuda()
x = torch.randn(B, L, 512, device="cuda", requires_grad=True)
cot = torch.randn(B, L, 512, device="cuda")
for it in range(3):
    y = layer.forward_bm(x, ops.MaskSpec(0))
    (y * cot).sum().backward()
torch.cuda.synchronize()
lib = ttmi.lib()
buf = (ctypes.c_ulonglong * 1024)()
assert lib.ttmi_debug_bwd_stamps(buf, 1024) == 0
st = [[buf[16 * n + k] for k in range(16)] for n in range(64)]
names = ["wait slab", "park dG + frags", "issue fetch", "dq mfma + slots", "barrier", "reduce + store", "dE mfma", "dc", "park q + barrier"]
print("kernel start -> first block: %d cycles" % (st[1][0] - st[0][0]))
tot = [0] * 9
for n in range(1, 17):
    seg = [st[n][k + 1] - st[n][k] for k in range(9)]
    print("block %2d: " % n + "  ".join("%s %5d" % (names[k], seg[k]) for k in range(9)) + "  | total %6d" % (st[n][9] - st[n][0]))
    for k in range(9): tot[k] += seg[k]
print("sum: " + "  ".join("%s %d" % (names[k], tot[k]) for k in range(9)) + "  = %d" % sum(tot))
print("kernel start -> loop end: %d cycles" % (st[20][0] - st[0][0]))
