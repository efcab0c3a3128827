#!/usr/bin/env python3
"""Same-process alternating A/B of the direct-store instances of the persistent 256x256 NT kernel (option 19, csrc/gemm_fast.hip: v8d) against v8 at
the joint's shapes: exp store (forward, M=816000 N=4334 K=1024), bias-only, row factor (dgrad, N=1024 K=4352).  GPU box only."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))
import torch
from ttmi import lib, check, ops
M, N, K, ld = int(os.environ.get("M", 816000)), 4334, 1024, 4352
ROUNDS, REPS = int(os.environ.get("ROUNDS", 4)), int(os.environ.get("REPS", 10))
g = torch.Generator(device="cuda").manual_seed(0)
A = (torch.randn(M, K, device="cuda", generator=g) * 0.5).tanh().to(torch.bfloat16)
B = (torch.randn(N, K, device="cuda", generator=g) * 0.03).to(torch.bfloat16)
bias = torch.randn(N, device="cuda", generator=g) * 0.1
P = torch.empty(M, ld, dtype=torch.bfloat16, device="cuda")
nparts = 4 * ((N + 255) // 256)
rs = torch.zeros(nparts, M, device="cuda")
Wt = torch.zeros(K, ld, device="cuda", dtype=torch.bfloat16)
Wt[:, :N] = B.t()
dH = torch.empty(M, K, dtype=torch.bfloat16, device="cuda")
sr = torch.rand(M, device="cuda", generator=g) + 0.5
L = lib()
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr())
cl = ctypes.c_long
def expo():
    check(L.ttmi_gemm_nt_bf16_exp(p(A), p(B), p(P), p(bias), p(rs), nparts, ctypes.c_void_p(0), M, N, K, cl(K), cl(K), cl(ld), st()), "exp")
def plain():
    check(L.ttmi_gemm_nt_bf16(p(A), p(B), p(P), 1, p(bias), M, N, K, cl(K), cl(K), cl(ld), st()), "nt")
def rowf():      # dH = (P . Wp) * (1 - H^2) * s_r; the mask operand (A: the hidden rows) is rewritten scaled - values drift, timing does not
    check(L.ttmi_gemm_nt_bf16_rowscale(p(P), p(Wt), p(dH), p(A), p(sr), M, K, ld, cl(ld), cl(ld), cl(K), st()), "rowscale")
def timeit(f):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(REPS): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / REPS
fl = {"expo": 2.0 * M * N * K, "plain": 2.0 * M * N * K, "rowf": 2.0 * M * K * ld}
CASES = {"expo": (expo, 1), "plain": (plain, 2), "rowf": (rowf, 4), "expo11": (expo, 8), "plain11": (plain, 16), "expob": (expo, 32), "plainb": (plain, 32), "rowfb": (rowf, 32)}
fl.update(expo11=fl["expo"], plain11=fl["plain"], expob=fl["expo"], plainb=fl["plain"], rowfb=fl["rowf"])
if os.environ.get("CHECK"):       # v11 against v8: stored values within one bf16 step, row sums to rounding
    ops.set_option(19, 0); expo(); P0, r0 = P.clone(), rs.sum(0)
    ops.set_option(19, 8); P.fill_(7.0); rs.zero_(); expo(); torch.cuda.synchronize()
    d = (P.float() - P0.float()).abs() / P0.float().abs().clamp_min(1e-30)
    print("v11 exp store vs v8: max rel diff %.3e, differing entries %.2e, pad zero %s, row sums rel %.3e" % (float(d.max()), float((d > 0).float().mean()),
          bool((P[:, N:] == 0).all()), float(((rs.sum(0) - r0).abs() / r0).max())), flush=True)
    ops.set_option(19, 0)
for name in os.environ.get("WHICH", "expo,plain,rowf").split(","):
    f, bit = CASES[name]
    if name.startswith("rowf"): sr.fill_(1.0)
    t = {0: [], bit: []}
    for r in range(ROUNDS):
        for b in (0, bit):
            ops.set_option(19, b)
            t[b].append(timeit(f))
    ops.set_option(19, 0)
    a, d = sorted(t[0]), sorted(t[bit])
    print("%-7s v8  %s  median %.3f ms = %.0f TFLOP/s" % (name, " ".join("%.3f" % x for x in t[0]), a[len(a) // 2], fl[name] / a[len(a) // 2] / 1e9))
    print("%-7s var %s  median %.3f ms = %.0f TFLOP/s  (%+.1f %%)" % (name, " ".join("%.3f" % x for x in t[bit]), d[len(d) // 2], fl[name] / d[len(d) // 2] / 1e9,
                                                                   100 * (d[len(d) // 2] / a[len(a) // 2] - 1)), flush=True)
