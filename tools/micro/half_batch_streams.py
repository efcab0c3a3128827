"""Would two half-batches of the audio encoder on two streams overlap one half's memory-bound phases (GEMM epilogue bursts, LayerNorm, attention) with the other
half's K loops?  Forward only, no autograd: full batch on one stream against 2 x 16 utterances on two streams (same kernels, half the tiles per launch).
usage: python tools/micro/half_batch_streams.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))
os.environ.setdefault("TTMI_PRECISION", "bf16")
import torch
import bench
from tt.model import Transducer
from ttmi.train import FlatModel
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = Transducer(bench.c2_config()).to(dev).train()
flat = FlatModel(model); flat.enable_shadows()
x = torch.randn(32, 500, 512, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def full():
    with torch.no_grad():
        return model.encoder(x)
def halves(n=2):
    outs = []
    with torch.no_grad():
        main = torch.cuda.current_stream()
        for i, st in enumerate((sa, sb)[:n]):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                outs.append(model.encoder(x[i * 32 // n:(i + 1) * 32 // n]))
        for st in (sa, sb)[:n]:
            main.wait_stream(st)
    return outs
def seq_halves():
    with torch.no_grad():
        return [model.encoder(x[:16]), model.encoder(x[16:])]
def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
a = full(); b = torch.cat(halves()); torch.cuda.synchronize()
print("max |full - halves| = %.3e (dropout masks differ by position only if p > 0; eval of the same rows)" % float((a - b).abs().max()))
for r in range(3):
    print("round %d: full batch %.3f ms | two halves on two streams %.3f ms | two halves one after the other %.3f ms" % (r, timeit(full), timeit(halves), timeit(seq_halves)))
