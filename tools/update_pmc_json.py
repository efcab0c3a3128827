#!/usr/bin/env python3
"""Rebuild profiles/pmc_joint_projection.json from the three rocprofv3 --pmc passes of tools/profile_round.sh (GPU box).
usage: update_pmc_json.py <fetch dir> <write dir> <sq dir> <commit> <round tag> [loss form: exp (default) | two-call]"""
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FORM = sys.argv[6] if len(sys.argv) > 6 else "exp"
# the projection launch of the default bench.py step: exp-store epilogue (LEAN 3) in the exp-domain loss form, bias epilogue (LEAN 1) otherwise
KERNEL = "gemm_nt_bf16_v8_kernel<unsigned short, 3>" if FORM == "exp" else "gemm_nt_bf16_v8_kernel<unsigned short, 1>"
NPARTS = 4 * ((4334 + 255) // 256)


def values(d, counter):
    out = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == counter:
                out.append(float(r["Counter_Value"]))
    return out


fetch, write = values(sys.argv[1], "FETCH_SIZE"), values(sys.argv[2], "WRITE_SIZE")
busy, gui = values(sys.argv[3], "SQ_VALU_MFMA_BUSY_CYCLES"), values(sys.argv[3], "GRBM_GUI_ACTIVE")
src = os.path.join(ROOT, "transformer-transducer_amd", "csrc", "gemm_fast.hip")
mean = lambda v: sum(v) / len(v)
j = {
    "kernel": KERNEL + " joint vocabulary projection, M=816000 N=4334 K=1024 (one launch per step, persistent 256x256 tiles)",
    "source": "profiles/%s_pmc_joint_kernels.txt: rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE and the SQ / GRBM counters in three separate passes of "
              "the same bench.py command (tools/profile_round.sh); means over %d / %d launches" % (sys.argv[5], len(fetch), len(write)),
    "commit": sys.argv[4],
    "loss_form": FORM,
    "gemm_fast_sha16": hashlib.sha256(open(src, "rb").read()).hexdigest()[:16],
    "fetch_size_kb": mean(fetch), "write_size_kb": mean(write),
    "correction": "gfx950 FETCH_SIZE reads 1/2 of wide (16 B/lane) streaming reads incl. global_load ... lds (MI355X_MICROARCH.md, HBM): traffic = "
                  "2*FETCH + WRITE; FETCH counts L2 misses served by the Infinity Cache too (re-read A panels)",
    "algorithmic_bytes": 816000.0 * 1024 * 2 + 4334.0 * 1024 * 2 + 816000.0 * 4352 * 2 + (816000.0 * NPARTS * 4 if FORM == "exp" else 0.0),
    "mfma_busy": round(mean(busy) / (mean(gui) / 8.0 * 1024.0), 3),
    "mfma_busy_source": "SQ_VALU_MFMA_BUSY_CYCLES %.4e / (GRBM_GUI_ACTIVE %.4e / 8 XCDs * 1024 SIMDs)" % (mean(busy), mean(gui)),
}
json.dump(j, open(os.path.join(ROOT, "profiles", "pmc_joint_projection.json"), "w"), indent=1)
print(json.dumps(j, indent=1))


# ---- the secondary roofline lines of bench.py (roofline_attn / roofline_loss / roofline_wgrad): per-launch HBM-side traffic of their kernels from
# the same three passes (the counters existed in profiles/ but the bench line said null)
def kernel_values(d, counter, pattern):
    out = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pattern in r["Kernel_Name"] and r["Counter_Name"] == counter:
                out.append(float(r["Counter_Value"]))
    return out


def sha16(rel):
    return hashlib.sha256(open(os.path.join(ROOT, "transformer-transducer_amd", "csrc", rel), "rb").read()).hexdigest()[:16]


def traffic_of(patterns):
    """sum over the kernels of one line: mean over launches of 2 * FETCH_SIZE + WRITE_SIZE (KB -> bytes), None when a kernel did not run"""
    total, parts = 0.0, {}
    for pat in patterns:
        f, w = kernel_values(sys.argv[1], "FETCH_SIZE", pat), kernel_values(sys.argv[2], "WRITE_SIZE", pat)
        if not f or not w:
            return None, parts
        parts[pat] = {"fetch_size_kb": mean(f), "write_size_kb": mean(w), "launches": len(f)}
        total += (2.0 * mean(f) + mean(w)) * 1024.0
    return total, parts


sec = {"commit": sys.argv[4], "round": sys.argv[5], "loss_form": FORM,
       "source": "profiles/%s_pmc_joint_kernels.txt (the same three rocprofv3 --pmc passes); traffic = 2 * FETCH_SIZE + WRITE_SIZE per launch, "
                 "mean over the launches of the profiled steps (audio-sized launches only for the attention kernel: L = 500)" % sys.argv[5],
       "lines": {}}
for line, pats, srcs in (("attn", ["flash_bwd_rel2_kernel<0>"], ["attn_flash.hip"]),
                         ("loss", ["rnnt_prep_exp_kernel", "rnnt_lattice_lds_kernel", "rnnt_scale_exp_kernel"], ["rnnt.hip"]),
                         ("wgrad", ["gemm_tn_bf16_group_kernel"], ["gemm_fast.hip"])):
    t, parts = traffic_of(pats)
    sec["lines"][line] = {"traffic": t, "kernels": parts, "sources": {f: sha16(f) for f in srcs}}
json.dump(sec, open(os.path.join(ROOT, "profiles", "pmc_secondary_kernels.json"), "w"), indent=1)
print(json.dumps(sec, indent=1))
