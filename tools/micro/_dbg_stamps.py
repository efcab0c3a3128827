import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))
import torch
from ttmi import lib, check
M, N, K, ld = 408000, 4334, 1024, 4352
L = lib(); st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream); p = lambda t: ctypes.c_void_p(t.data_ptr())
g = torch.Generator(device="cuda").manual_seed(0)
A = (torch.randn(M, K, device="cuda", generator=g) * 0.5).tanh().to(torch.bfloat16)
B = (torch.randn(N, K, device="cuda", generator=g) * 0.03).to(torch.bfloat16)
bias = torch.randn(N, device="cuda", generator=g) * 0.1
C = torch.empty(M, ld, dtype=torch.bfloat16, device="cuda")
stamps = torch.zeros(16 * 8, device="cuda")
for _ in range(3):
    check(L.ttmi_gemm_nt_bf16_dbg(p(A), p(B), p(C), p(bias), p(stamps), M, N, K, ctypes.c_long(K), ctypes.c_long(K), ctypes.c_long(ld), st), "dbg")
torch.cuda.synchronize()
s = stamps.cpu().view(16, 8)[:, :5]
print("per tile (10 ns ticks): wait+bar | K loop | realign | prologue issue | epilogue | total")
for i in range(15):
    a = s[i]; nxt = s[i + 1][0]
    print("%2d: %5.0f %5.0f %5.0f %5.0f %5.0f   total %5.0f" % (i, a[1] - a[0], a[2] - a[1], 0, a[3] - a[2], a[4] - a[3], nxt - a[0]))
