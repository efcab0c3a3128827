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
A8 = torch.randn(8192, 8192, device="cuda", generator=g).to(torch.bfloat16)
C8 = torch.empty(8192, 8192, device="cuda", dtype=torch.bfloat16)
for ver in (8, 10, 11, 12, 13, 8, 10):
    ops.set_option(1, ver)
    ms = timeit(lambda: ops.gemm_nt_bf16(A8, A8, C8), 10)
    print("v%d 8192^3 %.3f ms %.1f TF" % (ver, ms, 2.0 * 8192 ** 3 / ms / 1e9), flush=True)
