#!/usr/bin/env python3
"""Markdown summary of a round's committed logs (profiles/<tag>_*): the table DESIGN.md section 7 quotes.   usage: tools/round_table.py r03 [dir]"""
import json
import os
import re
import sys

tag = sys.argv[1]
D = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def line(name):
    p = os.path.join(D, "%s_%s.log" % (tag, name))
    if not os.path.exists(p):
        return None
    for l in reversed(open(p).read().splitlines()):
        if l.startswith("{"):
            return json.loads(l)
    return None


def fmt(v, nd=3):
    return "n/a" if v is None else ("%." + str(nd) + "g") % v


c2 = line("bench_c2")
if c2:
    r = c2["roofline"]
    print("| C2 (headline) | %.2f ms = %.0f utt/s; graph replay %s ms (host %s ms, 1 launch); two-call form %s ms; fp32 mode %s ms; host issue %.1f ms |"
          % (c2["ms_per_step"], c2["value"], c2.get("graph_replay_form", {}).get("ms_per_step"), c2.get("graph_replay_form", {}).get("host_issue_ms_per_step"),
             c2.get("two_call_form", {}).get("ms_per_step"), c2.get("fp32_form", {}).get("ms_per_step"), c2["host_issue_ms_per_step"]))
    print("| joint projection (`roofline`) | %.3f ms = %.0f TFLOP/s = %.3f of 2.5 PF; traffic %s GB vs 9.0 algorithmic; MFMA-busy %s |"
          % (r["kernel_ms"], r["achieved"], r["frac"], fmt(r["traffic"] and r["traffic"] / 1e9), r.get("mfma_busy")))
    for k in ("roofline_loss", "roofline_attn", "roofline_wgrad"):
        q = c2[k]
        print("| `%s` | %s ms, %s %s = %s |" % (k, q["kernel_ms"], fmt(q["achieved"], 4), q["unit"], q["frac"]))
    print("| loss error (B=2 sample, worst utterance) | timed form %s, two-call %s, encoder states alone %s |"
          % (c2.get("loss_rel_err_vs_oracle_timed_form"), c2.get("loss_rel_err_vs_oracle"), c2.get("loss_rel_err_encoder_states_only")))
    cb = c2.get("cpu_baseline")
    if cb:
        print("| `cpu_baseline` | %s %s on %s cores (%s) |" % (cb["value"], cb["unit"], cb["cores"], cb["kind"]))
for w in ("c4-band", "c4-chunk", "c5"):
    d = line("bench_" + w)
    if d:
        extra = ""
        if w == "c5":
            extra = "; lattice op (`roofline`) %s ms, joint projection %s ms" % (d["roofline"]["kernel_ms"], d.get("roofline_joint", {}).get("kernel_ms"))
        print("| %s | %.2f ms = %.1f utt/s (two-call form %s ms; host issue %.1f ms)%s |"
              % (w, d["ms_per_step"], d["value"], d.get("two_call_form", {}).get("ms_per_step"), d["host_issue_ms_per_step"], extra))
p = os.path.join(D, "%s_decode.log" % tag)
if os.path.exists(p):
    m = re.findall(r"\{.*\}", open(p).read())
    if m:
        d = json.loads(m[-1])
        print("| greedy decode | %s utt/s, %s ms per symbol, tokens identical to the oracle: %s |" % (d.get("utt_per_s"), d.get("ms_per_symbol"), d.get("tokens_identical_to_oracle")))
