import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "transformer-transducer_amd"))
import torch
from ttmi import ops
def timeit(fn, n=50):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
g = torch.Generator(device="cuda").manual_seed(0)
rows = 16000
for (m, n, k, what) in [(rows, 1536, 512, "qkv fwd"), (rows, 512, 512, "o fwd/dgrad"), (rows, 1024, 512, "ffn1 fwd / ffn2 dgrad"), (rows, 512, 1024, "ffn2 fwd / ffn1 dgrad"), (rows, 512, 1536, "qkv dgrad")]:
    A = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16)
    B = torch.randn(n, k, device="cuda", generator=g).to(torch.bfloat16)
    for cdt in (torch.bfloat16, torch.float32):
        C = torch.empty(m, n, device="cuda", dtype=cdt)
        ms = timeit(lambda: ops.gemm_nt_bf16(A, B, C))
        print("nt %-24s %5dx%4dx%4d -> %-8s %7.1f us %7.1f TFLOP/s" % (what, m, n, k, str(cdt).split(".")[1], ms * 1e3, 2.0 * m * n * k / ms / 1e9), flush=True)
