#!/usr/bin/env python3
"""TFLOP/s of the throughput GEMMs at the encoder shapes of the C2 workload (rows = B*T = 16000).  GPU box only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))
import torch
from ttmi import ops


def timeit(fn, n=20):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    vers = [int(v) for v in os.environ.get("GEMM_VERSIONS", "4").split(",")]
    rows = 16000
    ops.set_option(4, int(os.environ.get("TTMI_TN_TARGET", "512")))
    for ver in vers:
        ops.set_option(1, ver)
        tot = 0.0
        for (m, n, k, what) in [(rows, 1536, 512, "qkv fwd"), (rows, 512, 512, "o fwd / o dgrad"), (rows, 2048, 512, "ffn1 fwd / ffn2 dgrad"),
                                (rows, 512, 2048, "ffn2 fwd / ffn1 dgrad"), (rows, 512, 1536, "qkv dgrad")]:
            A = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16)
            B = torch.randn(n, k, device="cuda", generator=g).to(torch.bfloat16)
            for cdt in (torch.bfloat16, torch.float32):
                C = torch.empty(m, n, device="cuda", dtype=cdt)
                ms = timeit(lambda: ops.gemm_nt_bf16(A, B, C))
                print("v%d nt %-24s %5dx%4dx%4d -> %-8s %7.1f us %7.1f TFLOP/s" % (ver, what, m, n, k, str(cdt).split(".")[1], ms * 1e3, 2.0 * m * n * k / ms / 1e9), flush=True)
        for (m, n, k, what) in [(512, 512, rows, "o wgrad"), (1536, 512, rows, "qkv wgrad"), (2048, 512, rows, "ffn1 wgrad"), (512, 2048, rows, "ffn2 wgrad")]:
            A = torch.randn(k, m, device="cuda", generator=g).to(torch.bfloat16)
            B = torch.randn(k, n, device="cuda", generator=g).to(torch.bfloat16)
            C = torch.zeros(m, n, device="cuda")
            ms = timeit(lambda: ops.gemm_tn_bf16(A, B, C, accumulate=True))
            print("v%d tn %-24s %5dx%4dx%5d             %7.1f us %7.1f TFLOP/s" % (ver, what, m, n, k, ms * 1e3, 2.0 * m * n * k / ms / 1e9), flush=True)
    ops.set_option(1, 4)


if __name__ == "__main__":
    main()
