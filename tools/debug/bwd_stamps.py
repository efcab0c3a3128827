#!/usr/bin/env python3
"""In-kernel cycle stamps of flash_bwd_rel_kernel (library built with `make EXTRA=-DTTMI_STAMPS`): one wave's s_memtime at the segment
boundaries of every step of one workgroup (key block 1 of head 5), printed as cycles per segment.  Timing experiments only."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))
os.environ.setdefault("TTMI_PRECISION", "bf16")
import torch
from tt.encoder import BaseEncoder
from ttmi import ops
import ttmi

ops.set_option(2, int(os.environ.get("TTMI_FLASH_DEBUG", "0")))
B, L = int(os.environ.get("B", 32)), int(os.environ.get("L", 500))
torch.manual_seed(0)
layer = BaseEncoder(k_len=410, n_head=8, d_model=512, d_head=64, d_inner=1024, dropout=0.0).cuda()
x = torch.randn(B, L, 512, device="cuda", requires_grad=True)
cot = torch.randn(B, L, 512, device="cuda")
for it in range(4):
    y = layer.forward_bm(x, ops.MaskSpec(0))
    (y * cot).sum().backward()
torch.cuda.synchronize()
lib = ttmi.lib()
buf = (ctypes.c_ulonglong * 1024)()
rc = lib.ttmi_debug_bwd_stamps(buf, 1024)
assert rc == 0, rc
st = [[buf[16 * n + k] for k in range(16)] for n in range(64)]
names = ["barrier1", "wait+park", "barrier2", "prefetch issue", "read_bias", "S/dP mfma", "elements+stores", "dV/dK mfma"]
t0 = st[0][0]
print("kernel start -> first step: %d cycles" % (st[1][0] - t0))
tot = [0] * 8
for n in range(1, 17):
    seg = [st[n][k + 1] - st[n][k] for k in range(8)]
    gap = (st[n + 1][0] - st[n][8]) if n < 16 else 0
    print("step %2d: " % n + "  ".join("%s %5d" % (names[k], seg[k]) for k in range(8)) + "  | total %6d  gap %d" % (st[n][8] - st[n][0], gap))
    for k in range(8):
        tot[k] += seg[k]
print("sum over 16 steps: " + "  ".join("%s %d" % (names[k], tot[k]) for k in range(8)) + "  = %d" % sum(tot))
print("loop end -> stores drained: %d cycles;  kernel start -> end: %d cycles" % (st[17][1] - st[17][0], st[17][1] - t0))
