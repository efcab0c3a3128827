"""Times the generic batched GEMM (csrc/gemm.hip) on the attention-core shapes of an audio layer (B=32, H=8, L=500, Dh=64) in the exact-f32 and the
three-term bf16 (GEMM_BF16X3) forms.  usage (GPU box): python tools/debug/bench_generic_gemm.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "transformer-transducer_amd"))
from ttmi import ops  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    B, H, L, D = 32, 8, 500, 64
    Z = B * H
    g = torch.Generator(device="cuda").manual_seed(0)
    q = torch.randn(Z, L, D, device="cuda", generator=g)
    k = torch.randn(Z, L, D, device="cuda", generator=g)
    s = torch.empty(Z, L, L, device="cuda")
    o = torch.empty(Z, L, D, device="cuda")
    p = torch.randn(Z, L, L, device="cuda", generator=g)
    AK, BK = ops.GEMM_A_KMAJOR, ops.GEMM_B_KMAJOR
    qkv = torch.randn(B, L, 3 * H * D, device="cuda", generator=g)
    pp = torch.randn(Z * L * (L + 1) + 8, device="cuda", generator=g)[1:]      # the pitch-(L+1) view starts one float into its slab
    cases = [
        ("q.k^T   [500x500x64]  A k-major, B k-major", lambda f: ops.gemm(q, k, s, L, L, D, D, D, L, AK | BK | f, nz1=B, nz2=H, sA=(H * L * D, L * D), sB=(H * L * D, L * D), sC=(H * L * L, L * L)), 4.0 * Z * L * L),
        ("qu.k^T  [500x500x64]  beta=1, slices of a [B,L,3HD] qkv", lambda f: ops.gemm(qkv, qkv[0, 0, H * D:], s, L, L, D, 3 * H * D, 3 * H * D, L, AK | BK | f, beta=1.0, nz1=B, nz2=H, sA=(L * 3 * H * D, D), sB=(L * 3 * H * D, D), sC=(H * L * L, L * L)), 4.0 * Z * L * L),
        ("p.v     [500x64x500]  A k-major, B n-major", lambda f: ops.gemm(p, k, o, L, D, L, L, D, D, AK | f, nz1=B, nz2=H, sA=(H * L * L, L * L), sB=(H * L * D, L * D), sC=(H * L * D, L * D)), 4.0 * Z * L * L),
        ("p^T.do  [500x64x500]  A m-major, B n-major", lambda f: ops.gemm(p, k, o, L, D, L, L, D, D, f, nz1=B, nz2=H, sA=(H * L * L, L * L), sB=(H * L * D, L * D), sC=(H * L * D, L * D)), 4.0 * Z * L * L),
        ("ds.k    [500x64x500]  beta=1", lambda f: ops.gemm(p, k, o, L, D, L, L, D, D, AK | f, beta=1.0, nz1=B, nz2=H, sA=(H * L * L, L * L), sB=(H * L * D, L * D), sC=(H * L * D, L * D)), 4.0 * Z * L * L),
        ("dg.e    [500x64x500]  pitch L+1 view, beta=1", lambda f: ops.gemm(pp, k, o, L, D, L, L + 1, D, D, AK | f, beta=1.0, nz1=B, nz2=H, sA=(H * L * (L + 1), L * (L + 1)), sB=(H * L * D, L * D), sC=(H * L * D, L * D)), 4.0 * Z * L * L),
    ]
    for name, fn, big_bytes in cases:
        for label, f in (("f32", 0), ("x3", ops.GEMM_BF16X3)):
            us = timed(lambda: fn(f))
            print("%-50s %-4s %8.1f us   %.2f TB/s on the [Z,L,L] operand" % (name, label, us, big_bytes / us / 1e6), flush=True)


if __name__ == "__main__":
    main()


def extra():
    """context: the library's batched f32 product on the same operands, and the aligned L = 512 variant of q.k^T / p.v"""
    B, H, D = 32, 8, 64
    Z = B * H
    g = torch.Generator(device="cuda").manual_seed(0)
    for L in (500, 512):
        q = torch.randn(Z, L, D, device="cuda", generator=g)
        k = torch.randn(Z, L, D, device="cuda", generator=g)
        p = torch.randn(Z, L, L, device="cuda", generator=g)
        s = torch.empty(Z, L, L, device="cuda")
        o = torch.empty(Z, L, D, device="cuda")
        AK, BK = ops.GEMM_A_KMAJOR, ops.GEMM_B_KMAJOR
        f = ops.GEMM_BF16X3
        print("L=%d  torch.bmm q.k^T f32 %8.1f us" % (L, timed(lambda: torch.bmm(q, k.transpose(1, 2), out=s))), flush=True)
        print("L=%d  torch.bmm p.v   f32 %8.1f us" % (L, timed(lambda: torch.bmm(p, k, out=o))), flush=True)
        print("L=%d  torch.bmm p^T.v f32 %8.1f us" % (L, timed(lambda: torch.bmm(p.transpose(1, 2), k, out=o))), flush=True)
        print("L=%d  x3 q.k^T            %8.1f us" % (L, timed(lambda: ops.gemm(q, k, s, L, L, D, D, D, L, AK | BK | f, nz1=B, nz2=H, sA=(H * L * D, L * D), sB=(H * L * D, L * D), sC=(H * L * L, L * L)))), flush=True)
        print("L=%d  x3 p.v              %8.1f us" % (L, timed(lambda: ops.gemm(p, k, o, L, D, L, L, D, D, AK | f, nz1=B, nz2=H, sA=(H * L * L, L * L), sB=(H * L * D, L * D), sC=(H * L * D, L * D)))), flush=True)
        print("L=%d  copy [Z,L,L]        %8.1f us" % (L, timed(lambda: s.copy_(p))), flush=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "extra":
    extra()
