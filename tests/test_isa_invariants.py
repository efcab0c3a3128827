"""A counted wait on the vector-memory counter is only as good as the count behind it.  flash_bwd_rel_kernel waits for its asm-issued
prefetch with `s_waitcnt vmcnt(32)`: "everything but the 32 youngest operations", which are exactly the step's 32 slab store instructions
(16 accumulator elements x 2 slabs per wave, in both the interior and the edge path).  A first version waited for vmcnt(63) believing the
step had 64 stores: harmless in every test, a rare illegal access in the C5 step.  This test compiles the kernel file
for gfx950 and checks the count in the ISA itself, so that a change to the store loop (vectorised stores, another tile shape) cannot pass
without the wait being re-derived.  No GPU needed; skipped where hipcc is absent."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "transformer-transducer_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_flash_bwd_prefetch_wait_matches_the_store_count(tmp_path):
    out = tmp_path / "attn_flash.s"
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-I" + os.path.join(ROOT, "include"),
                        "-x", "hip", "--cuda-device-only", "-S", os.path.join(CSRC, "attn_flash.hip"), "-o", str(out)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = out.read_text()
    kernels = re.findall(r"^(_ZN\S*flash_bwd_rel_kernelILi(?:32|64)ELi\dE\S*):", asm, re.M)
    assert len(kernels) >= 8                                         # head dims 32 / 64 x the mask kinds
    for name in kernels:
        body = asm[asm.index(name + ":"):]
        body = body[:body.index(".end_amdhsa_kernel")]
        stores = len(re.findall(r"^\s*buffer_store_short", body, re.M))
        waits = re.findall(r"s_waitcnt vmcnt\((\d+)\)\s*$", body, re.M)
        # two unrolled steps x (interior path + edge path) x 32 slab store instructions each
        assert stores == 128, (name, stores)
        assert "32" in waits, (name, waits)                          # the hand-placed wait, with the count of ONE step's stores
        assert "63" not in waits, (name, waits)                      # (the compiler's own "nothing to wait for" forms carry the other counters too)
