#!/usr/bin/env python3
"""Separates the persistent 256x256 GEMM's per-tile fixed cost (pipeline fill + un-overlapped epilogue) from its K-loop rate at the joint
projection's output shape: time(K) = tiles_per_CU * (a * K/64 + b), fitted over K = 256 .. 4096.  GPU box only."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))
import numpy as np
import torch
import ttmi
if os.environ.get("TTMI_AB_LIB"):          # same-box A/B: another build of the library (tools/ab_bench.sh keeps them under ab/)
    ttmi.LIB_PATH = os.path.abspath(os.environ["TTMI_AB_LIB"])
from ttmi import lib, check, ops
if os.environ.get("STAGGER"):
    ops.set_option(1, 1000 + int(os.environ["STAGGER"]))      # start stagger of the persistent NT kernel, 10 ns ticks over the grid
KS = tuple(int(k) for k in os.environ.get("KS", "256,512,1024,2048,4096").split(","))
M, N, ld = int(os.environ.get("M", 408000)), 4334, 4352
L = lib()
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr())
g = torch.Generator(device="cuda").manual_seed(0)
C = torch.empty(M, ld, dtype=torch.bfloat16, device="cuda")
nparts = 4 * ((N + 255) // 256)
rs = torch.zeros(nparts, M, device="cuda")
bias = torch.randn(N, device="cuda", generator=g) * 0.1
res = {}
for K in KS:
    A = (torch.randn(M, K, device="cuda", generator=g) * 0.5).tanh().to(torch.bfloat16)
    B = (torch.randn(N, K, device="cuda", generator=g) * (1.0 / K ** 0.5)).to(torch.bfloat16)
    def plain():
        check(L.ttmi_gemm_nt_bf16(p(A), p(B), p(C), 1, p(bias), M, N, K, ctypes.c_long(K), ctypes.c_long(K), ctypes.c_long(ld), st), "nt")
    def expo():
        check(L.ttmi_gemm_nt_bf16_exp(p(A), p(B), p(C), p(bias), p(rs), nparts, ctypes.c_void_p(0), M, N, K, ctypes.c_long(K), ctypes.c_long(K), ctypes.c_long(ld), st), "exp")
    for f in (plain, expo):
        for _ in range(2): f()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): f()
        e.record(); torch.cuda.synchronize()
        res[(f.__name__, K)] = s.elapsed_time(e) / 5
        print("%-5s K=%4d  %.3f ms  %.0f TFLOP/s" % (f.__name__, K, res[(f.__name__, K)], 2.0 * M * N * K / res[(f.__name__, K)] / 1e9), flush=True)
    del A, B
if len(KS) < 2:
    sys.exit(0)
tiles = ((M + 255) // 256) * ((N + 255) // 256) / 256.0
for name in ("plain", "expo"):
    Ks = np.array(KS, dtype=float)
    t = np.array([res[(name, int(k))] for k in Ks]) * 1e3 / tiles              # us per tile
    a, b = np.polyfit(Ks / 64.0, t, 1)
    print("%-5s per tile: %.3f us per K-tile (%.0f TFLOP/s in the loop) + %.2f us fixed; at K=1024 the fixed part is %.0f %% of the tile" %
          (name, a, 2.0 * 256 * 256 * 64 / a / 1e6 * 256, b, 100 * b / (16 * a + b)))
