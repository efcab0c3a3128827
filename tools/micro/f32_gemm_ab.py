"""exact-f32 NT products of decode / fp32-mode sizes: the persistent 256x128 kernel with f32 operands (default) against the generic 128x128
kernel (option 17 = 0), same box.    python3 tools/micro/f32_gemm_ab.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))
from ttmi import ops

f = ops.GEMM_A_KMAJOR | ops.GEMM_B_KMAJOR
NAMES = {3: "persistent 256x128", 2: "64x64 tiles", 0: "gemm.hip (32x32 / 128x128)", 1: "default rule"}
for M, N, K in [(64, 1536, 512), (128, 1536, 512), (192, 1536, 512), (256, 1536, 512), (256, 512, 512), (256, 512, 1024), (384, 1536, 512), (512, 1536, 512), (512, 512, 1024),
                (800, 1536, 512), (800, 512, 512), (800, 1024, 512), (800, 512, 1024), (1024, 1536, 512), (1200, 1536, 512), (1600, 1536, 512), (1600, 512, 1024),
                (2048, 1536, 512), (2048, 1024, 512), (2560, 512, 512), (3360, 1536, 512), (3360, 512, 1024), (1792, 4334, 1024), (2048, 4334, 1024), (16000, 1536, 512)]:
    A = torch.randn(M, K, device="cuda")
    B = torch.randn(N, K, device="cuda")
    C = torch.empty(M, N, device="cuda")
    row = []
    for mode in (3, 2, 0, 1):
        ops.set_option(17, mode)
        for _ in range(3):
            ops.gemm(A, B, C, M, N, K, K, K, N, f)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gemm(A, B, C, M, N, K, K, K, N, f)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50
        row.append("%s %7.1f us %5.1f TF" % (NAMES[mode], us, 2.0 * M * N * K / us / 1e6))
    ops.set_option(17, 1)
    print("M %5d N %5d K %5d | " % (M, N, K) + " | ".join(row))
