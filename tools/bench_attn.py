#!/usr/bin/env python3
"""One encoder layer at the BASELINE C2 shape (B=32, T=500, d=512, H=8x64), forward+backward, for rocprofv3 runs.
TTMI_FLASH_DEBUG bits: 1 = skip bias read, 2 = skip dS write (timing experiments only; results are wrong)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))
os.environ.setdefault("TTMI_PRECISION", "bf16")
import torch
from tt.encoder import BaseEncoder
from ttmi import ops
from ttmi.ops import MaskSpec

ops.set_option(2, int(os.environ.get("TTMI_FLASH_DEBUG", "0")))
ops.set_option(0, int(os.environ.get("TTMI_NO_FLASH", "0")))
ops.set_option(4, int(os.environ.get("TTMI_TN_TARGET", "512")))
ops.set_option(3, int(os.environ.get("TTMI_FORK", "1")))
if os.environ.get("TTMI_LN_GRID"):
    ops.set_option(12, int(os.environ["TTMI_LN_GRID"]))
ops.set_option(10, int(os.environ.get("TTMI_SLICES", "1")))   # attention backward in n batch slices (slabs stay in the Infinity Cache)
ops.set_option(5, int(os.environ.get("TTMI_SLAB", "0")))      # 1 = slab by GEMM; 2 / 4 = skip the slab kernel's MFMA part / stores
B, L = int(os.environ.get("B", 32)), int(os.environ.get("L", 500))
torch.manual_seed(0)
layer = BaseEncoder(k_len=int(os.environ.get("K", 410)), n_head=8, d_model=512, d_head=64, d_inner=1024, dropout=float(os.environ.get("DROPOUT", 0.0))).cuda()
MASK = MaskSpec(int(os.environ.get("MASK", 0)))      # 1 = causal (the label encoder)
x = torch.randn(B, L, 512, device="cuda", requires_grad=True)
cot = torch.randn(B, L, 512, device="cuda")
for it in range(6):
    y = layer.forward_bm(x, MASK)
    (y * cot).sum().backward()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for it in range(10):
    y = layer.forward_bm(x, MASK)
    (y * cot).sum().backward()
e.record()
torch.cuda.synchronize()
print("layer fwd+bwd %.3f ms" % (s.elapsed_time(e) / 10))
