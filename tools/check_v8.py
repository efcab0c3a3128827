#!/usr/bin/env python3
"""Bring-up check of one throughput-GEMM generation (GEMM_V, default 8): exact-integer parity on awkward shapes, then TFLOP/s
at the joint shapes next to the default generation.  GPU box only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))
import torch
from ttmi import ops

V = int(os.environ.get("GEMM_V", "8"))


def ints(shape, g):
    return torch.randint(-4, 5, shape, device="cuda", generator=g).to(torch.bfloat16)


def check():
    g = torch.Generator(device="cuda").manual_seed(5)
    bad = 0
    for (M, N, K, pad) in [(1024, 256, 64, 0), (1024, 256, 128, 0), (1024, 256, 192, 0), (1030, 700, 72, 3), (2048, 260, 320, 4),
                           (5000, 4334, 1024, 18), (3000, 1024, 4352, 0), (1025, 257, 1000, 0)]:
        A, B = ints((M, K), g), ints((N, K), g)
        bias = torch.randint(-3, 4, (N,), device="cuda", generator=g).float()
        for cdt in (torch.float32, torch.bfloat16):
            Cfull = torch.full((M, N + pad), 5.0, device="cuda", dtype=cdt)
            C = Cfull[:, :N]
            ops.set_option(1, V)
            ops.gemm_nt_bf16(A, B, C, bias)
            ops.set_option(1, 4)
            want = (A.float() @ B.float().t() + bias).to(cdt)
            ok = torch.equal(C, want) and (pad == 0 or bool((Cfull[:, N:] == 5.0).all()))
            if not ok:
                bad += 1
                d = (C.float() - want.float()).abs()
                idx = (d > 0).nonzero()
                print("MISMATCH", M, N, K, pad, cdt, "n_bad", idx.shape[0], "first", idx[:4].tolist(), flush=True)
            else:
                print("ok", M, N, K, pad, cdt, flush=True)
    return bad


def check_v9():
    g = torch.Generator(device="cuda").manual_seed(11)
    bad = 0
    for (M, N, K, pad) in [(1024, 128, 64, 0), (1024, 256, 128, 0), (1024, 384, 192, 4), (16000, 512, 512, 0), (16000, 1536, 512, 0),
                           (3001, 700, 72, 3), (16000, 2048, 2048, 0), (1025, 129, 1000, 0), (50000, 512, 256, 0)]:
        A, B = ints((M, K), g), ints((N, K), g)
        bias = torch.randint(-3, 4, (N,), device="cuda", generator=g).float()
        for cdt in (torch.float32, torch.bfloat16):
            Cfull = torch.full((M, N + pad), 5.0, device="cuda", dtype=cdt)
            ops.set_option(1, 9)
            ops.gemm_nt_bf16(A, B, Cfull[:, :N], bias)
            ops.set_option(1, 4)
            want = (A.float() @ B.float().t() + bias).to(cdt)
            ok = torch.equal(Cfull[:, :N], want) and (pad == 0 or bool((Cfull[:, N:] == 5.0).all()))
            if not ok:
                bad += 1
                idx = ((Cfull[:, :N].float() - want.float()).abs() > 0).nonzero()
                print("V9 MISMATCH", M, N, K, pad, cdt, "n_bad", idx.shape[0], "first", idx[:4].tolist(), flush=True)
            else:
                print("v9 ok", M, N, K, pad, cdt, flush=True)
    return bad


def check_tn():
    g = torch.Generator(device="cuda").manual_seed(7)
    bad = 0
    for (M, N, K) in [(1024, 1024, 32768), (1124, 1024, 40000), (2048, 1024, 33001), (4334, 1024, 65536 + 64)]:
        lda = (M + 7) // 8 * 8 + 8
        A, B = ints((K, lda), g), ints((K, N), g)
        want = A[:, :M].float().t() @ B.float()
        wcs = A[:, :M].float().sum(0)
        for ver in (4, 5, 8):
            ops.set_option(1, ver)
            C = torch.ones(M, N, device="cuda")
            cs = torch.full((M,), 3.0, device="cuda")
            ops.gemm_tn_bf16(A[:, :M], B, C, accumulate=True, colsum_a=cs)
            C2 = torch.ones(M, N, device="cuda")
            ops.gemm_tn_bf16(A[:, :M], B, C2, accumulate=True)
            ops.set_option(1, 4)
            ok = torch.equal(C, want + 1) and torch.equal(cs, wcs + 3) and torch.equal(C2, want + 1)
            if not ok:
                bad += 1
                d = (C - want - 1).abs()
                print("TN MISMATCH v%d" % ver, M, N, K, "C bad", int((d > 0).sum()), "first", (d > 0).nonzero()[:4].tolist(),
                      "cs bad", int(((cs - wcs - 3).abs() > 0).sum()), "C2 bad", int(((C2 - want - 1).abs() > 0).sum()), flush=True)
            else:
                print("tn ok v%d" % ver, M, N, K, flush=True)
    return bad


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


def bench():
    M, Vv, Vp, J = 32 * 500 * 51, 4334, 4352, 1024
    g = torch.Generator(device="cuda").manual_seed(0)
    H = torch.randn(M, J, device="cuda", generator=g).to(torch.bfloat16)
    Wp = torch.randn(Vv, J, device="cuda", generator=g).to(torch.bfloat16)
    WpT = torch.zeros(J, Vp, device="cuda", dtype=torch.bfloat16)
    WpT[:, :Vv] = Wp.t()
    bias = torch.randn(Vv, device="cuda")
    Z = torch.empty(M, Vp, device="cuda", dtype=torch.bfloat16)
    dH = torch.empty(M, J, device="cuda", dtype=torch.bfloat16)
    A4 = torch.randn(4096, 4096, device="cuda", generator=g).to(torch.bfloat16)
    C4 = torch.empty(4096, 4096, device="cuda", dtype=torch.bfloat16)
    A8 = torch.randn(8192, 8192, device="cuda", generator=g).to(torch.bfloat16)
    C8 = torch.empty(8192, 8192, device="cuda", dtype=torch.bfloat16)
    for ver in [int(x) for x in os.environ.get('GEMM_BENCH', '4,%d' % V).split(',')]:
        ops.set_option(1, ver)
        for name, fn, f in [
            ("joint fwd  M=%d N=%d K=%d" % (M, Vv, J), lambda: ops.gemm_nt_bf16(H, Wp, Z[:, :Vv], bias), 2.0 * M * Vv * J),
            ("joint dgrad M=%d N=%d K=%d" % (M, J, Vp), lambda: ops.gemm_nt_bf16(Z, WpT, dH), 2.0 * M * Vp * J),
            ("4096^3", lambda: ops.gemm_nt_bf16(A4, A4, C4), 2.0 * 4096 ** 3),
            ("8192^3", lambda: ops.gemm_nt_bf16(A8, A8, C8), 2.0 * 8192 ** 3),
        ]:
            ms = timeit(fn)
            print("v%d %-40s %8.3f ms  %7.1f TFLOP/s" % (ver, name, ms, f / ms / 1e9), flush=True)
    gW = torch.zeros(Vv, J, device="cuda")
    gb = torch.zeros(Vv, device="cuda")
    for ver in (5, 4, 8):
        ops.set_option(1, ver)
        ms = timeit(lambda: ops.gemm_tn_bf16(Z[:, :Vv], H, gW, accumulate=True, colsum_a=gb))
        print("v%d joint wgrad + colsum M=%d N=%d K=%d %8.3f ms  %7.1f TFLOP/s" % (ver, Vv, J, M, ms, 2.0 * M * Vv * J / ms / 1e9), flush=True)
        ms = timeit(lambda: ops.gemm_tn_bf16(Z[:, :Vv], H, gW, accumulate=True))
        print("v%d joint wgrad          %8.3f ms  %7.1f TFLOP/s" % (ver, ms, 2.0 * M * Vv * J / ms / 1e9), flush=True)
    ops.set_option(1, 4)


if __name__ == "__main__":
    bad = check() + check_tn() + check_v9()
    if bad:
        sys.exit(1)
    bench()
