import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))
import torch
from ttmi import lib, check
M, N, K, ld = 408000, 4334, int(os.environ.get("K", 1024)), 4352
L = lib(); st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream); p = lambda t: ctypes.c_void_p(t.data_ptr())
g = torch.Generator(device="cuda").manual_seed(0)
A = (torch.randn(M, K, device="cuda", generator=g) * 0.5).tanh().to(torch.bfloat16)
B = (torch.randn(N, K, device="cuda", generator=g) * 0.03).to(torch.bfloat16)
bias = torch.randn(N, device="cuda", generator=g) * 0.1
C = torch.empty(M, ld, dtype=torch.bfloat16, device="cuda")
nparts = 68
rs = torch.zeros(nparts, M, device="cuda")
stamps = torch.zeros(128, device="cuda")
for _ in range(3):
    check(L.ttmi_gemm_nt_bf16_dbg(p(A), p(B), p(C), p(bias), p(stamps), p(rs), nparts, M, N, K, ctypes.c_long(K), ctypes.c_long(K), ctypes.c_long(ld), st), "dbg")
torch.cuda.synchronize()
for name, o in (("wave 0", 0), ("wave 4", 64)):
    s = stamps[o:o + 64].cpu()
    nk = K // 64
    its = [float(s[1 + t + 1] - s[1 + t]) for t in range(min(nk, 39) - 1)]
    print(name, "K-iterations (10 ns ticks):", " ".join("%.0f" % v for v in its))
    print(name, "loop start -> end %.0f | realign %.0f | coords+prologue issue %.0f | epilogue %.0f" % (s[48] - s[0], s[49] - s[48], s[50] - s[49], s[51] - s[50]))
