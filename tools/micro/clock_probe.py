"""Sample the GPU clock / power (rocm-smi) while the persistent GEMM runs at 8192^3 on all CUs: is the MFMA loop clock-limited?"""
import os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))
import torch
from ttmi import ops

A = torch.randn(8192, 8192, device="cuda").to(torch.bfloat16)
C = torch.empty(8192, 8192, device="cuda", dtype=torch.bfloat16)
ops.set_option(1, 8)
stop = False
def smi():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            for ln in out.splitlines():
                if "sclk" in ln or "Power" in ln or "mclk" in ln:
                    print("   ", ln.strip(), flush=True)
        except Exception as e:
            print("rocm-smi failed:", e, flush=True)
            return
        time.sleep(0.5)
print("idle:", flush=True)
out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
print("\n".join(l for l in out.splitlines() if "sclk" in l or "Power" in l), flush=True)
th = threading.Thread(target=smi); th.start()
t0 = time.time()
n = 0
while time.time() - t0 < 4.0:
    for _ in range(50):
        ops.gemm_nt_bf16(A, A, C)
    torch.cuda.synchronize(); n += 50
dt = time.time() - t0
stop = True; th.join()
print("8192^3 x %d in %.2f s: %.1f TFLOP/s sustained" % (n, dt, 2.0 * 8192 ** 3 * n / dt / 1e12))
