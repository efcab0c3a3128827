"""audio-encoder-sized bf16 NT products (16000 rows): the persistent kernels (default) against the 64 x 64-tile kernel (option 1 = 5: nothing persistent)
and the 128 x 128 kernel (1 = 5 then 1 = 16), same box.   python3 tools/micro/bf16_enc_gemm_ab.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))
from ttmi import ops

for M, N, K in [(16000, 512, 512), (16000, 1536, 512), (16000, 1024, 512), (16000, 512, 1024), (16000, 512, 1536), (6400, 512, 512), (6400, 1536, 512)]:
    A = torch.randn(M, K, device="cuda").bfloat16()
    B = torch.randn(N, K, device="cuda").bfloat16()
    out = []
    for cdt in (torch.bfloat16, torch.float32):
        C = torch.empty(M, N, device="cuda", dtype=cdt)
        for name, opts in (("persistent", [(1, 4), (1, 48)]), ("64x64", [(1, 5), (1, 48)]), ("128x128", [(1, 5), (1, 16)])):
            for k, v in opts:
                ops.set_option(k, v)
            for _ in range(3):
                ops.gemm_nt_bf16(A, B, C)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                ops.gemm_nt_bf16(A, B, C)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1000 / 30
            out.append("%s %5.1f us %4.0f TF" % (name, us, 2.0 * M * N * K / us / 1e6))
    ops.set_option(1, 4)
    ops.set_option(1, 48)
    print("M %5d N %4d K %4d | bf16 out: %s | f32 out: %s" % (M, N, K, ", ".join(out[:3]), ", ".join(out[3:])))
