"""bf16 NT products the persistent kernels leave (label-encoder sized): the 64 x 64-tile LDS-DMA kernel (default) against the 128 x 128 kernel
(option 1 = 16), same box.    python3 tools/micro/bf16_mid_ab.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))
from ttmi import ops

for M, N, K in [(1632, 1536, 512), (1632, 512, 512), (1632, 1024, 512), (1632, 512, 1024), (1632, 512, 1536), (816, 1536, 512), (408, 1536, 512), (102, 1536, 512),
                (51, 512, 512), (3264, 1536, 512), (3264, 512, 1024), (6400, 512, 512)]:
    A = torch.randn(M, K, device="cuda").bfloat16()
    B = torch.randn(N, K, device="cuda").bfloat16()
    row = []
    for cdt in (torch.bfloat16, torch.float32):
        C = torch.empty(M, N, device="cuda", dtype=cdt)
        for v in (16 + 32, 16):
            ops.set_option(1, v)
            for _ in range(3):
                ops.gemm_nt_bf16(A, B, C)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.gemm_nt_bf16(A, B, C)
            e1.record()
            torch.cuda.synchronize()
            row.append(e0.elapsed_time(e1) * 50)
    ops.set_option(1, 16 + 32)
    print("M %5d N %5d K %5d | bf16 out: 64x64 %6.1f us, 128x128 %6.1f us | f32 out: 64x64 %6.1f us, 128x128 %6.1f us" % (M, N, K, row[0], row[1], row[2], row[3]))
