"""Which part of the training step does not survive HIP graph capture?  Each case runs in its own process (a crash in hipStreamEndCapture kills it).
    python tools/debug/graph_bisect.py            # runs every case as a subprocess
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = ["enc_fwd", "enc_fwdbwd", "dec_fwdbwd", "joint_exp", "joint_plain", "opt", "both_fwd", "both_enc", "full_nooverlap", "full"]

if len(sys.argv) == 1:
    for c in CASES:
        r = subprocess.run([sys.executable, __file__, c], capture_output=True, text=True, timeout=600)
        tail = (r.stdout + r.stderr).strip().splitlines()[-1:] or [""]
        print("%-22s rc=%d  %s" % (c, r.returncode, tail[0][:200]), flush=True)
    sys.exit(0)

case = sys.argv[1]
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["TTMI_PRECISION"] = "bf16"
import torch
from test_dp_nccl_gpu import _bench_cfg, _bench_data
from tt.model import Transducer
from ttmi import ops
from ttmi.ops import MaskSpec
from ttmi.train import FlatModel, FusedOptimizer, GradSync

dev = torch.device("cuda", 0)
cfg = _bench_cfg()
cfg["dropout"] = 0.1
if case == "full_nooverlap":
    cfg["overlap_label_encoder"] = False
torch.manual_seed(1)
model = Transducer(cfg).to(dev).train()
flat = FlatModel(model)
if case not in ("enc_fwdbwd_nodefer", "both_enc_nodefer"):
    flat.enable_grouped_wgrads()
if case == "both_enc_nofork":
    ops.set_option(3, 0)
flat.enable_shadows()
sync = GradSync(flat)
opt = FusedOptimizer(flat, kind="sgd", lr=0.00025, momentum=0.9, max_grad_norm=200.0)
x, y = _bench_data(0, 0)
x, y = x.to(dev), y.to(dev)
il = torch.full((8,), 512, dtype=torch.int32, device=dev)
tl = torch.full((8,), 7, dtype=torch.int32, device=dev)
ypad = torch.nn.functional.pad(y, [1, 0, 0, 0], value=0)
enc_s = torch.randn(8, 512, 512, device=dev)
dec_s = torch.randn(8, 8, 512, device=dev)


def body():
    if case == "enc_fwd":
        with torch.no_grad():
            return model.encoder(x, None).sum()
    if case in ("enc_fwdbwd", "enc_fwdbwd_nodefer"):
        flat.zero_grad()
        model.encoder(x, None).sum().backward()
        ops.wgrad_flush(every_stream=True)
        return flat.grad.sum()
    if case == "dec_fwdbwd":
        flat.zero_grad()
        model.decoder(ypad, MaskSpec(1)).sum().backward()
        return flat.grad.sum()
    if case in ("joint_exp", "joint_plain"):
        from tt.model import _JointLossFn
        j = model.joint
        flat.zero_grad()
        e, d = enc_s.clone().requires_grad_(True), dec_s.clone().requires_grad_(True)
        st = j.exp_shift_state(dev) if case == "joint_exp" else None
        loss = _JointLossFn.apply(e, d, j.forward_layer.weight, j.forward_layer.bias, j.project_layer.weight, j.project_layer.bias,
                                  y.int().contiguous(), il, tl, 1, 8, "mean", st, True)
        loss.backward()
        return loss.detach()
    if case == "opt":
        opt.step()
        return flat.flat.sum()
    if case == "zero_grad":
        flat.zero_grad()
        return flat.grad.sum()
    if case == "both_fwd":
        with torch.no_grad():
            e, d = model._encode(x, y)
        return e.sum() + d.sum()
    if case in ("both_enc", "both_enc_nodefer", "both_enc_nofork"):
        flat.zero_grad()
        e, d = model._encode(x, y)
        (e.sum() + d.sum()).backward()
        ops.wgrad_flush(every_stream=True)
        ops.join_side_streams()
        return flat.grad.sum()
    flat.zero_grad()
    sync.start_step()
    loss = model.loss(x, il, y, tl, exp_domain=True)
    loss.backward()
    sync.finish()
    opt.step()
    return loss.detach()


s = torch.cuda.Stream(dev)
s.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(s):
    for _ in range(3):
        out = body()
torch.cuda.current_stream(dev).wait_stream(s)
torch.cuda.synchronize()
print("eager ok %s" % float(out), flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    out = body()
print("captured", flush=True)
g.replay()
torch.cuda.synchronize()
print("replayed ok %s" % float(out), flush=True)
