"""exact-f32 NT products of decode / fp32-mode sizes: the persistent 256x128 kernel with f32 operands (default) against the generic 128x128
kernel (option 17 = 0), same box.    python3 tools/micro/f32_gemm_ab.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))
from ttmi import ops

f = ops.GEMM_A_KMAJOR | ops.GEMM_B_KMAJOR
for M, N, K in [(2048, 4334, 1024), (1792, 4334, 1024), (2048, 1024, 512), (3360, 1536, 512), (3360, 1024, 512), (3360, 512, 1024), (1600, 1536, 512),
                (1024, 1536, 512), (1024, 512, 1024), (2560, 1536, 512), (2560, 512, 512), (16000, 1536, 512), (16000, 512, 1024)]:
    A = torch.randn(M, K, device="cuda")
    B = torch.randn(N, K, device="cuda")
    C = torch.empty(M, N, device="cuda")
    row = []
    for fast in (1, 0, 2):
        ops.set_option(17, fast & 1)
        ops.set_option(7, 8192 if fast == 2 else 128)                # third column: the 32x32-tile kernel of the decoder-sized products, forced
        for _ in range(3):
            ops.gemm(A, B, C, M, N, K, K, K, N, f)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.gemm(A, B, C, M, N, K, K, K, N, f)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        row.append((us, 2.0 * M * N * K / us / 1e6))
    ops.set_option(17, 1)
    ops.set_option(7, 128)
    print("M %6d N %5d K %5d | persistent %8.1f us %6.1f TFLOP/s | generic %8.1f us %6.1f TFLOP/s | 32x32 tiles %8.1f us %6.1f TFLOP/s" %
          (M, N, K, row[0][0], row[0][1], row[1][0], row[1][1], row[2][0], row[2][1]))
