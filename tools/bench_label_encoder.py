#!/usr/bin/env python3
"""Label-encoder forward on one history of L tokens (the per-symbol cost of greedy decode): eager launches vs a captured graph.
    python tools/bench_label_encoder.py [--precision fp32]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    os.environ["TTMI_PRECISION"] = a.precision
    from bench import c2_config
    from tt.model import Transducer
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    model = Transducer(c2_config()).to(dev).eval()
    toks = torch.randint(1, 4334, (1, 128), device=dev)
    enc = torch.randn(64, 512, device=dev)
    res = {}
    with torch.no_grad():
        for L in (1, 8, 32, 64, 128):
            x = toks[:, :L]
            for _ in range(3):
                y = model.decoder(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.iters):
                y = model.decoder(x)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(2):
                    yg = model.decoder(x)
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                yg = model.decoder(x)
            g.replay()
            torch.cuda.synchronize()
            ok = bool(torch.equal(yg, y))
            t3 = time.perf_counter()
            for _ in range(a.iters):
                g.replay()
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            # joint block + scan (eager)
            from ttmi import ops
            d = y[:, -1:, :]
            for _ in range(3):
                lg = model.joint(enc.unsqueeze(0), d)
                ops.greedy_scan(lg[0, :, 0, :])
            torch.cuda.synchronize()
            t5 = time.perf_counter()
            for _ in range(a.iters):
                lg = model.joint(enc.unsqueeze(0), d)
                ops.greedy_scan(lg[0, :, 0, :])
            t6 = time.perf_counter()
            res[L] = {"eager_enqueue_ms": round(1e3 * (t1 - t0) / a.iters, 3), "eager_total_ms": round(1e3 * (t2 - t0) / a.iters, 3),
                      "graph_ms": round(1e3 * (t4 - t3) / a.iters, 3), "graph_equal": ok,
                      "joint_scan_sync_ms": round(1e3 * (t6 - t5) / a.iters, 3)}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
