"""Host cost of the torch calls the op wrappers make per launch (GPU box): python tools/debug/host_call_costs.py"""
import time
import torch

torch.cuda.init()
x = torch.empty(8, device="cuda")
idx = torch.cuda.current_device()


def t(name, fn, n=20000):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    print("%-70s %7.2f us" % (name, (time.perf_counter() - t0) / n * 1e6), flush=True)


t("torch.cuda.current_stream().cuda_stream", lambda: torch.cuda.current_stream().cuda_stream)
t("torch.cuda.current_stream(x.device).cuda_stream", lambda: torch.cuda.current_stream(x.device).cuda_stream)
t("torch._C._cuda_getCurrentRawStream(idx)", lambda: torch._C._cuda_getCurrentRawStream(idx))
t("torch.cuda.current_device()", lambda: torch.cuda.current_device())
t("torch._C._cuda_getDevice()", lambda: torch._C._cuda_getDevice())
t("torch.empty(1024, device='cuda')", lambda: torch.empty(1024, device="cuda"))
t("torch.empty(1024, device=x.device)", lambda: torch.empty(1024, device=x.device))
t("torch.empty_like(x)", lambda: torch.empty_like(x))
t("x.data_ptr()", lambda: x.data_ptr())
t("torch.cuda.is_current_stream_capturing()", lambda: torch.cuda.is_current_stream_capturing())
s = torch.cuda.Stream()
ev = torch.cuda.Event()
t("ev.record()", lambda: ev.record())
t("torch.cuda.current_stream().wait_event(ev)", lambda: torch.cuda.current_stream().wait_event(ev))
t("with torch.cuda.stream(s): pass", lambda: torch.cuda.stream(s).__enter__() or torch.cuda.stream(s).__exit__(None, None, None), 2000)
