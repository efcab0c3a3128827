"""Scan the compiled ISA of every kernel for loads that are waited for one at a time: a vector load whose NEXT vm wait is `s_waitcnt vmcnt(0)` with no other load
issued in between.  That is what a load written inside `if (...)` inside an unrolled loop compiles to (round 6: the joint's sums over frames ran sixteen such loads
per label position - DESIGN.md section 4k).  usage: python tools/isa_serialized_loads.py [file stems, default: every csrc/*.hip]   (hipcc -S, ~1 min per large file)"""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "transformer-transducer_amd", "csrc")
stems = sys.argv[1:] or sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(CSRC, "*.hip")))
print("# kernels with >= 4 loads waited for alone and >= 50 % of their loads alone (global_load_lds counts as a load)")
with tempfile.TemporaryDirectory() as tmp:
    for stem in stems:
        out = os.path.join(tmp, stem + ".s")
        r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-I" + os.path.join(ROOT, "include"),
                            "-x", "hip", "--cuda-device-only", "-S", os.path.join(CSRC, stem + ".hip"), "-o", out], capture_output=True, text=True)
        if r.returncode:
            print("%s: compile failed\n%s" % (stem, r.stderr[-500:]))
            continue
        txt = open(out).read()
        for m in re.finditer(r"^(_Z\w+):\s*;? ?@.*?\n(.*?)\n\s*\.amdhsa_kernel", txt, re.S | re.M):
            name, lines = m.group(1), [l.strip() for l in m.group(2).split("\n")]
            loads = [i for i, l in enumerate(lines) if re.match(r"(global|buffer|flat)_load", l)]
            alone = 0
            for k, i in enumerate(loads):
                nxt = loads[k + 1] if k + 1 < len(loads) else len(lines)
                if any(l.startswith("s_waitcnt") and "vmcnt(0)" in l for l in lines[i + 1:nxt]):
                    alone += 1
            if alone >= 4 and 2 * alone >= len(loads):
                print("%-12s loads %3d alone %3d  %s" % (stem, len(loads), alone, name[:120]))
