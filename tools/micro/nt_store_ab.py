"""streaming (nontemporal) output stores of the persistent GEMMs off (option 1 = 14) / on (15): exactness on integers, joint forward and dgrad timing."""
import os, sys
sys.path.insert(0, "transformer-transducer_amd")
import torch
from ttmi import ops
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
g = torch.Generator(device="cuda").manual_seed(0)
for (M, N, K, pad) in [(300000, 4334, 1024, 18), (270000, 1100, 128, 4), (262144 + 77, 1024, 256, 0)]:
    A = torch.randint(-4, 5, (M, K), device="cuda", generator=g).to(torch.bfloat16)
    B = torch.randint(-4, 5, (N, K), device="cuda", generator=g).to(torch.bfloat16)
    bias = torch.randint(-3, 4, (N,), device="cuda", generator=g).float()
    want = None
    for mode in (14, 15):
        ops.set_option(1, mode)
        C = torch.full((M, N + pad), 5.0, device="cuda", dtype=torch.bfloat16)
        ops.gemm_nt_bf16(A, B, C[:, :N], bias)
        if want is None:
            want = (A[:70000].float() @ B.float().t() + bias).to(torch.bfloat16)
            ref = C.clone()
        ok = torch.equal(C[:70000, :N], want) and torch.equal(C, ref)
        print("exact", (M, N, K, pad), "nt" if mode == 15 else "plain", ok, flush=True)
M, V, J = 32 * 500 * 51, 4334, 1024
H = torch.randn(M, J, device="cuda", generator=g).to(torch.bfloat16)
Wp = torch.randn(V, J, device="cuda", generator=g).to(torch.bfloat16)
bias = torch.randn(V, device="cuda")
Z = torch.empty(M, 4352, device="cuda", dtype=torch.bfloat16)
WpT = torch.zeros(J, 4352, device="cuda", dtype=torch.bfloat16)
dH = torch.empty(M, J, device="cuda", dtype=torch.bfloat16)
for mode in (14, 15, 14, 15):
    ops.set_option(1, mode)
    ms = timeit(lambda: ops.gemm_nt_bf16(H, Wp, Z[:, :V], bias), 5)
    m2 = timeit(lambda: ops.gemm_nt_bf16(Z, WpT, dH), 5)
    print("%s joint fwd %.3f ms  %.1f TFLOP/s | dgrad %.3f ms" % (["plain", "nt"][mode - 14], ms, 2.0 * M * V * J / ms / 1e9, m2), flush=True)
ops.set_option(1, 15)
