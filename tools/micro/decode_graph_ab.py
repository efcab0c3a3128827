import sys, json
sys.path.insert(0, "tools")
import bench_decode
for graphs in (True, False):
    import tt.model
    orig = tt.model.Transducer._label_state_graphs
    if not graphs:
        tt.model.Transducer._label_state_graphs = lambda self, dev: None
    out = bench_decode.run(8, 500, 0.1, "fp32")[0]
    tt.model.Transducer._label_state_graphs = orig
    print("graphs" if graphs else "eager ", json.dumps({k: out[k] for k in ("utt_per_s", "decode_ms_per_utt", "ms_per_symbol")}), flush=True)
