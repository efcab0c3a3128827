#!/usr/bin/env python3
"""Micro-benchmark of the GEMM kernels at the joint-network shapes (GPU box only).  Prints TFLOP/s."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))
import torch
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
    for ver in [int(v) for v in os.environ.get("GEMM_VERSIONS", "1,2,3,4").split(",")]:
        ops.set_option(1, ver)
        print("---- throughput-GEMM version", ver, flush=True)
        run()


def run():
    M, V, Vp, J = 32 * 500 * 51, 4334, 4352, 1024
    g = torch.Generator(device="cuda").manual_seed(0)
    H = torch.randn(M, J, device="cuda", generator=g).to(torch.bfloat16)
    Wp = torch.randn(V, J, device="cuda", generator=g).to(torch.bfloat16)
    WpT = torch.zeros(J, Vp, device="cuda", dtype=torch.bfloat16)
    WpT[:, :V] = Wp.t()
    bias = torch.randn(V, device="cuda")
    Z = torch.empty(M, Vp, device="cuda", dtype=torch.bfloat16)
    Zf = torch.empty(M, Vp, device="cuda", dtype=torch.float32)
    dH = torch.empty(M, J, device="cuda", dtype=torch.bfloat16)
    gW = torch.zeros(V, J, device="cuda")
    fl = 2.0 * M * V * J
    for name, fn, f in [
        ("nt fwd  Z(bf16)=H.Wp^T  M=%d N=%d K=%d" % (M, V, J), lambda: ops.gemm_nt_bf16(H, Wp, Z[:, :V], bias), fl),
        ("nt fwd  Z(f32) =H.Wp^T", lambda: ops.gemm_nt_bf16(H, Wp, Zf[:, :V], bias), fl),
        ("nt dgrad dH=dZ.WpT^T   M=%d N=%d K=%d" % (M, J, Vp), lambda: ops.gemm_nt_bf16(Z, WpT, dH), 2.0 * M * Vp * J),
        ("tn wgrad gW+=dZ^T.H    M=%d N=%d K=%d" % (V, J, M), lambda: ops.gemm_tn_bf16(Z[:, :V], H, gW, True), fl),
    ]:
        ms = timeit(fn)
        print("%-55s %8.3f ms  %7.1f TFLOP/s" % (name, ms, f / ms / 1e9), flush=True)
    # encoder-sized
    for (m, n, k) in [(16000, 1536, 512), (16000, 512, 512), (16000, 1024, 512), (16000, 512, 1024)]:
        A = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16)
        B = torch.randn(n, k, device="cuda", generator=g).to(torch.bfloat16)
        C = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
        ms = timeit(lambda: ops.gemm_nt_bf16(A, B, C), 20)
        print("nt %dx%dx%d %8.3f ms %7.1f TFLOP/s" % (m, n, k, ms, 2.0 * m * n * k / ms / 1e9), flush=True)


if __name__ == "__main__":
    main()
