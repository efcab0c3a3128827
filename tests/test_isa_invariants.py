"""A counted wait on the vector-memory counter is only as good as the count behind it.

`s_waitcnt vmcnt(n)` returns when at most the n YOUNGEST vector-memory operations of the wave are outstanding (loads, stores, atomics and
LDS-DMA count together, in issue order).  A hand-placed wait is right iff on EVERY path at least n such operations are issued between the
last operation that must have completed and the wait.  The kernels mark both ends in their asm (csrc/common.h): `TTMI_VM_GUARD(id)` =
"everything before this point must be complete when the matching wait returns", `TTMI_VM_WAIT(id, n)` = the wait.  This test compiles
the kernel files for gfx950, rebuilds every kernel's control-flow graph from the ISA (labels, s_branch / s_cbranch, s_endpgm) and proves,
by a shortest-path search with vector-memory instructions as unit edges, that

    min over all paths from any GUARD(id) to a WAIT(id, n)  of  #vector-memory instructions on the path  >=  n,

(and that every kernel's in-loop waits meet their count exactly: they wait for nothing they need not).  It also refuses hand-placed `s_waitcnt vmcnt(n>0)`
without a tag.  Round 3's `vmcnt(63)` guarded nothing - a step issued 32 operations, not 64 - and showed only as a rare illegal access
in the full C5 step; with this test the count cannot drift from the code.  No GPU needed; skipped where hipcc is absent."""
import os
import re
import shutil
import subprocess
from collections import deque

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "transformer-transducer_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
VMEM = re.compile(r"^(global_(load|store|atomic)|buffer_(load|store|atomic)|scratch_(load|store))")


def compile_isa(name, tmp_path):
    out = tmp_path / (name + ".s")
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-I" + os.path.join(ROOT, "include"),
                        "-x", "hip", "--cuda-device-only", "-S", os.path.join(CSRC, name + ".hip"), "-o", str(out)],
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    return out.read_text()


def kernels_of(asm):
    """-> {kernel name: body text}"""
    out = {}
    for m in re.finditer(r"^(_Z\S+):\s*; @\1\s*$", asm, re.M):
        nxt = asm.find("\n.Lfunc_end", m.end())
        if nxt < 0:
            continue
        out[m.group(1)] = asm[m.end():nxt]
    return out


def parse(body):
    """-> instructions: list of dicts {op, text, guard, wait: (id, n) or None, untagged_wait: n or None, label}; labels -> index"""
    ins, labels = [], {}
    in_asm = False
    pending = []                                    # lines of the current asm block
    for raw in body.splitlines():
        line = raw.strip()
        if not line:
            continue
        if line.startswith(";;#ASMSTART"):
            in_asm, pending = True, []
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            for ln in pending:
                g = re.match(r"^; TTMI_GUARD (\S+)", ln)
                if g:
                    ins.append(dict(op="<guard>", guard=g.group(1)))
                    continue
                w = re.match(r"^s_waitcnt vmcnt\((\d+)\)\s*(?:; TTMI_WAIT (\S+))?\s*$", ln)
                if w:
                    ins.append(dict(op="s_waitcnt", wait=(w.group(2), int(w.group(1))) if w.group(2) else None,
                                    untagged=int(w.group(1)) if not w.group(2) else None))
                    continue
                if ln.startswith(";"):
                    continue
                ins.append(dict(op=ln.split()[0], text=ln))
            continue
        if in_asm:
            pending.append(line)
            continue
        if line.startswith(";") or line.startswith("."):
            m = re.match(r"^(\.LBB\S+):", line)
            if m:
                labels[m.group(1)] = len(ins)
            continue
        m = re.match(r"^(\.LBB\S+):", line)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        ins.append(dict(op=line.split()[0], text=line))
    return ins, labels


def successors(ins, labels, i):
    op = ins[i]["op"]
    if op == "s_endpgm":
        return []
    assert not op.startswith("s_setpc") and not op.startswith("s_swappc"), "indirect branch: the path analysis does not cover it"
    if op == "s_branch":
        return [labels[ins[i]["text"].split()[1]]]
    nxt = [i + 1] if i + 1 < len(ins) else []
    if op.startswith("s_cbranch"):
        return nxt + [labels[ins[i]["text"].split()[1]]]
    return nxt


def min_vmem_between(ins, labels, gid, wait_index):
    """fewest vector-memory instructions on any path from a GUARD(gid) to instruction `wait_index` (None: no guard reaches it)"""
    INF = 1 << 30
    dist = [INF] * len(ins)
    dq = deque()
    for i, x in enumerate(ins):
        if x.get("guard") == gid:
            dist[i] = 0
            dq.append(i)
    while dq:                                       # 0-1 breadth-first search
        i = dq.popleft()
        if i == wait_index:
            continue                                # paths THROUGH the wait are another wait's business
        for j in successors(ins, labels, i):
            if ins[j].get("guard") == gid:
                continue                            # a later guard supersedes this one (it is a source itself)
            w = 1 if VMEM.match(ins[j]["op"]) else 0
            if dist[i] + w < dist[j]:
                dist[j] = dist[i] + w
                (dq.append if w else dq.appendleft)(j)
    return None if dist[wait_index] >= INF else dist[wait_index]


def check_file(name, tmp_path, expect):
    """expect: {guard id: (minimum number of kernels that must carry it, n)}"""
    asm = compile_isa(name, tmp_path)
    seen = {k: 0 for k in expect}
    exact = {}
    for kname, body in kernels_of(asm).items():
        ins, labels = parse(body)
        ids = set()
        for i, x in enumerate(ins):
            assert not x.get("untagged"), "%s: hand-placed s_waitcnt vmcnt(%d) without a TTMI_WAIT tag" % (kname, x.get("untagged") or 0)
            if x.get("wait"):
                gid, n = x["wait"]
                assert gid in expect, (kname, gid)
                assert n == expect[gid][1], (kname, gid, n)
                assert not any(VMEM.match(y["op"]) is None and y["op"].startswith("flat_") for y in ins), kname
                d = min_vmem_between(ins, labels, gid, i)
                assert d is not None, "%s: WAIT %s has no GUARD on any path" % (kname, gid)
                assert d >= n, "%s: vmcnt(%d) tagged %s, but a path issues only %d vector-memory operations after the guard" % (kname, n, gid, d)
                exact[gid] = exact.get(gid, 0) + (d == n)           # (d > n is safe: such a wait also covers a few younger operations)
                ids.add(gid)
        for gid in ids:
            seen[gid] += 1
    for gid, (kmin, _) in expect.items():
        assert seen[gid] >= kmin, (name, gid, seen[gid])
        assert exact.get(gid, 0) >= seen[gid], (name, gid, exact)     # every kernel has waits whose count is met exactly (the in-loop ones)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_gemm_fast_counted_waits(tmp_path):
    """the persistent GEMMs' LDS-DMA pipelines: v8 (NT and TN, 256x256: three half-tiles = 6 DMA instructions stay in flight), the
    3-stage 256x128 kernels (NT v9 with bf16 and with f32 operands, TN v9, the grouped wgrad: one stage = 6), the 64x64-tile
    exact-f32 kernel of the decoder-sized products (3 stages of 4)"""
    check_file("gemm_fast", tmp_path, {"v8": (4, 6), "v9": (3, 6), "mid": (1, 4)})


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_attention_counted_waits(tmp_path):
    """flash_bwd_rel_kernel's asm-issued tile prefetch is waited for with vmcnt(32): everything but a step's 32 slab store instructions
    (16 accumulator elements x 2 slabs per wave, in both the interior and the edge path).  The path search above cannot prove this one:
    the compiler joins the two element loops through a flag register (`interior` computed into an SGPR pair, tested again after the first
    loop), so the graph holds a path that runs neither - infeasible, but only by the flag's value.  What is checked instead is the count
    itself: every instance holds exactly 2 unrolled steps x (interior + edge) x 32 slab stores, every hand-placed wait carries the tag and
    the count of ONE step's stores, and a guard precedes it on a path."""
    asm = compile_isa("attn_flash", tmp_path)
    n_kernels = 0
    for kname, body in kernels_of(asm).items():
        if "flash_bwd_rel_kernel" not in kname:
            continue
        ins, labels = parse(body)
        waits = [(i, x["wait"]) for i, x in enumerate(ins) if x.get("wait")]
        assert waits and all(w == ("bwdrel", 32) for _, w in waits), (kname, waits)
        assert not any(x.get("untagged") for x in ins), kname
        stores = sum(1 for x in ins if x["op"].startswith("buffer_store_short"))
        assert stores == 128, (kname, stores)
        for i, (gid, n) in waits:
            assert min_vmem_between(ins, labels, gid, i) is not None, kname
        n_kernels += 1
    assert n_kernels >= 8                            # head dims 32 / 64 x the mask kinds
    # the second-generation kernel (LDS-DMA staging): the same count under its own tag, and every SGPR-based asm access opens with s_nop 4
    n2 = 0
    for kname, body in kernels_of(asm).items():
        if "flash_bwd_rel2_kernel" not in kname:
            continue
        ins, labels = parse(body)
        waits = [(i, x["wait"]) for i, x in enumerate(ins) if x.get("wait")]
        assert waits and all(w == ("bwdrel2", 32) for _, w in waits), (kname, waits)
        assert not any(x.get("untagged") for x in ins), kname
        stores = sum(1 for x in ins if x["op"].startswith("buffer_store_short"))
        assert stores == 128, (kname, stores)         # 2 unrolled steps x (interior + diagonal / edge) x 32
        for i, (gid, n) in waits:
            assert min_vmem_between(ins, labels, gid, i) is not None, kname
        for k, x in enumerate(ins):                  # hazard: VALU write of an SGPR (v_readlane) -> vector-memory read of it needs 5 wait states
            t = x.get("text", "")
            if (x["op"].startswith("global_load") and re.search(r"\bs\[\d+:\d+\]", t)) and x["op"] != "global_load_lds_dwordx4":
                assert ins[k - 1]["op"] == "s_nop" and ins[k - 1]["text"].split()[1] == "4", (kname, t)
            if x["op"] == "global_load_lds_dwordx4" and re.search(r"\bs\[\d+:\d+\]", t):
                assert ins[k - 1]["op"] == "s_nop" and ins[k - 1]["text"].split()[1] == "4" and ins[k - 2]["op"] == "s_mov_b32", (kname, t)
        n2 += 1
    assert n2 == 5                                   # the five mask kinds
    # attn_dqde_kernel: its slab / q prefetch is issued from asm too (raw-buffer form): same 5-wait-state rule, and no hand-placed counted wait
    nd = 0
    for kname, body in kernels_of(asm).items():
        if "attn_dqde_kernel" not in kname:
            continue
        ins, labels = parse(body)
        assert not any(x.get("untagged") for x in ins) and not any(x.get("wait") for x in ins), kname
        loads = [k for k, x in enumerate(ins) if x["op"].startswith("buffer_load_dwordx4")]
        assert len(loads) >= 9, (kname, len(loads))   # 4 quarters x (dS to LDS + dG to registers) + the q rows, per block
        for k in loads:
            assert ins[k - 1]["op"] == "s_nop" and ins[k - 1]["text"].split()[1] == "4", (kname, ins[k].get("text"))
        nd += 1
    assert nd == 1


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_no_untagged_counted_waits_elsewhere(tmp_path):
    """rnnt.hip, rowops.hip and layers.hip place no counted vector-memory waits by hand (only full drains, vmcnt(0))"""
    for name in ("rnnt", "rowops"):
        asm = compile_isa(name, tmp_path)
        for kname, body in kernels_of(asm).items():
            ins, _ = parse(body)
            assert not any(x.get("untagged") for x in ins), kname


def _alone_loads(body):
    """vector loads whose NEXT vm wait is vmcnt(0) with no other load issued in between -> (alone, loads)"""
    lines = [l.strip() for l in body.splitlines()]
    loads = [i for i, l in enumerate(lines) if re.match(r"(global|buffer|flat)_load", l)]
    alone = 0
    for k, i in enumerate(loads):
        nxt = loads[k + 1] if k + 1 < len(loads) else len(lines)
        if any(l.startswith("s_waitcnt") and "vmcnt(0)" in l for l in lines[i + 1:nxt]):
            alone += 1
    return alone, len(loads)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_frame_loads_of_the_joint_sums_stay_in_flight(tmp_path):
    """round 6: written as `if (t < T && act) { load; add }` in an unrolled loop, each of a label position's sixteen frame loads compiled into a block of its own with
    s_waitcnt vmcnt(0) behind it - one load in flight per lane, 3.0 TB/s for three rounds.  The unconditional form (clamped address, selected value) must keep
    compiling to loads issued back to back: at most the last load of a batch may be followed by a full drain.  Same for the fp32 / bf16x3 forms and the loss
    gradient that overwrites its logits with bf16 planes (its ONE vmcnt(0) is deliberate: the whole row must be in registers before the first store)"""
    asm = compile_isa("rowops", tmp_path)
    seen = 0
    for kname, body in kernels_of(asm).items():
        if "joint_sum_bwd_bf16x4_part_kernel" in kname:
            alone, loads = _alone_loads(body)
            assert loads == 16 and alone <= 1, (kname, alone, loads)
            seen += 1
        elif "joint_tanh_bwd_kernel" in kname or "joint_tanh_bwd_x3_kernel" in kname:
            alone, loads = _alone_loads(body)
            assert loads >= 32 and alone <= 2, (kname, alone, loads)
            seen += 1
    assert seen == 4
    asm = compile_isa("rnnt", tmp_path)
    seen = 0
    for kname, body in kernels_of(asm).items():
        if "rnnt_grad_split_kernel" in kname:
            alone, loads = _alone_loads(body)
            assert alone <= 6, (kname, alone, loads)          # (the row's vectors in one batch; the scalars of the row - lse, alpha, beta, labels - are few)
            seen += 1
    assert seen == 4
