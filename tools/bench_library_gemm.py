#!/usr/bin/env python3
"""Yardstick only (never on the product path): the vendor library's bf16 GEMM (torch.matmul -> hipBLASLt / rocBLAS) on the joint and
encoder shapes next to this repo's kernels.  Run under rocprofv3 --kernel-trace to see which library tile the heuristics pick."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))
import torch
import torch.nn.functional as F
from ttmi import ops


def timeit(fn, n=5):
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
    M, V, J = 32 * 500 * 51, 4334, 1024
    Vp = 4352
    g = torch.Generator(device="cuda").manual_seed(0)
    H = torch.randn(M, J, device="cuda", generator=g).to(torch.bfloat16)
    Wp = torch.randn(V, J, device="cuda", generator=g).to(torch.bfloat16)
    bias16 = torch.randn(V, device="cuda").to(torch.bfloat16)
    bias = bias16.float()
    Z = torch.empty(M, Vp, device="cuda", dtype=torch.bfloat16)
    Zl = torch.empty(M, V, device="cuda", dtype=torch.bfloat16)
    dH = torch.empty(M, J, device="cuda", dtype=torch.bfloat16)
    gW = torch.zeros(V, J, device="cuda")
    gW16 = torch.zeros(V, J, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * M * V * J
    rows = [
        ("ours    fwd  Z = H Wp^T + b", lambda: ops.gemm_nt_bf16(H, Wp, Z[:, :V], bias), fl),
        ("library fwd  F.linear(H, Wp, b)", lambda: F.linear(H, Wp, bias16, ) if False else torch.addmm(bias16, H, Wp.t(), out=Zl), fl),
        ("library fwd  no bias", lambda: torch.mm(H, Wp.t(), out=Zl), fl),
        ("library dgrad dH = Z Wp", lambda: torch.mm(Zl, Wp, out=dH), fl),
        ("ours    wgrad gW += Z^T H", lambda: ops.gemm_tn_bf16(Z[:, :V], H, gW, True), fl),
        ("library wgrad Z^T H (bf16 out)", lambda: torch.mm(Zl.t(), H, out=gW16), fl),
    ]
    for name, fn, f in rows:
        ms = timeit(fn)
        print("%-40s %8.3f ms  %7.1f TFLOP/s" % (name, ms, f / ms / 1e9), flush=True)
    for (m, n, k) in [(16000, 1536, 512), (16000, 512, 512), (16000, 1024, 512), (16000, 512, 1024), (8192, 8192, 8192)]:
        A = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16)
        B = torch.randn(n, k, device="cuda", generator=g).to(torch.bfloat16)
        C = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
        ms = timeit(lambda: ops.gemm_nt_bf16(A, B, C), 20)
        ml = timeit(lambda: torch.mm(A, B.t(), out=C), 20)
        print("nt %dx%dx%d ours %8.3f ms %7.1f TF | library %8.3f ms %7.1f TF" % (m, n, k, ms, 2.0 * m * n * k / ms / 1e9, ml, 2.0 * m * n * k / ml / 1e9), flush=True)


if __name__ == "__main__":
    main()
