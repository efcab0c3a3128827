#!/usr/bin/env python3
"""Standalone timing of the encoder weight-gradient GEMMs (C[M,N] += A[K,M]^T B[K,N], K = B*T = 16000) at the three shapes of one audio
layer, with the share of bf16 peak; checks the result against torch.  GPU box only.  TTMI_AB_LIB selects another build (ab/)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))
import torch
import ttmi
if os.environ.get("TTMI_AB_LIB"):
    ttmi.LIB_PATH = os.path.abspath(os.environ["TTMI_AB_LIB"])
from ttmi import lib, check
L = lib(); st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream); p = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
g = torch.Generator(device="cuda").manual_seed(0)
K = int(os.environ.get("K", 16000))
for M, N, cs in ((1536, 512, False), (512, 512, False), (1024, 512, True), (512, 1024, False)):
    A = torch.randn(K, M, device="cuda", generator=g).to(torch.bfloat16)
    B = torch.randn(K, N, device="cuda", generator=g).to(torch.bfloat16)
    C = torch.zeros(M, N, device="cuda")
    col = torch.zeros(M, device="cuda") if cs else None
    def run():
        check(L.ttmi_gemm_tn_bf16(p(A), p(B), p(C), M, N, K, ctypes.c_long(M), ctypes.c_long(N), ctypes.c_long(N), 1, p(col), st), "tn")
    run(); torch.cuda.synchronize()
    want = A.float().t() @ B.float()
    err = float((C - want).abs().max() / want.abs().max())
    cerr = float((col - A.float().sum(0)).abs().max() / A.float().sum(0).abs().max()) if cs else 0.0
    for _ in range(5): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50): run()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / 50
    C2 = torch.zeros_like(C); C3 = torch.zeros_like(C)
    check(L.ttmi_gemm_tn_bf16(p(A), p(B), p(C2), M, N, K, ctypes.c_long(M), ctypes.c_long(N), ctypes.c_long(N), 1, p(None), st), "tn")
    check(L.ttmi_gemm_tn_bf16(p(A), p(B), p(C3), M, N, K, ctypes.c_long(M), ctypes.c_long(N), ctypes.c_long(N), 1, p(None), st), "tn")
    print("M=%4d N=%4d K=%d colsum=%d: %.1f us  %.0f TFLOP/s (%.2f of 2500)  err %.1e colsum err %.1e  bit-identical reruns: %s" %
          (M, N, K, cs, us, 2.0 * M * N * K / us / 1e6, 2.0 * M * N * K / us / 1e6 / 2500, err, cerr, bool(torch.equal(C2, C3))), flush=True)

# ---- grouped launch: the 16 problems of four audio layers in one kernel (ttmi_wgrad_group)
from ttmi import ops
shapes = ((1536, 512, False), (512, 512, False), (1024, 512, True), (512, 1024, False)) * 4
ten = [(torch.randn(K, M, device="cuda", generator=g).to(torch.bfloat16), torch.randn(K, N, device="cuda", generator=g).to(torch.bfloat16),
        torch.zeros(M, N, device="cuda"), torch.zeros(M, device="cuda") if cs else None) for M, N, cs in shapes]
descs = (ops.WgradDesc * len(ten))()
for d, (A, B, C, col) in zip(descs, ten):
    d.A, d.B, d.C, d.colsum = A.data_ptr(), B.data_ptr(), C.data_ptr(), (col.data_ptr() if col is not None else 0)
    d.M, d.N, d.K, d.lda, d.ldb, d.ldc = A.shape[1], B.shape[1], K, A.stride(0), B.stride(0), C.stride(0)
def grp():
    check(L.ttmi_wgrad_group(descs, len(ten), st), "group")
for _ in range(3): grp()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): grp()
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) * 1e3 / 20
fl = sum(2.0 * M * N * K for M, N, _ in shapes)
by = sum((M + N) * K * 2.0 for M, N, _ in shapes)
print("grouped, 4 layers (16 problems, 256 tiles): %.1f us  %.0f TFLOP/s (%.2f of 2500), operands read once = %.0f MB -> %.2f TB/s" %
      (us, fl / us / 1e6, fl / us / 1e6 / 2500, by / 1e6, by / us / 1e6))
