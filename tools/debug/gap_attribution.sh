#!/bin/bash
# for every idle gap > 20 us between consecutive kernels of the default bench: was the NEXT kernel's hipLaunchKernel issued after the
# previous kernel had already ended (host late) or before (device-side wait)?  GPU box.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/gaps; rm -rf $O; mkdir -p $O
(cd $R && rocprofv3 --hip-trace --kernel-trace --output-format csv -d $O -o t -- python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-two-call --no-fp32-form --no-graph-form --no-sync-form > $O/run.log 2>&1)
python3 - "$O" <<'PY'
import csv,glob,sys,collections
d=sys.argv[1]
api={}
for r in csv.DictReader(open(glob.glob(d+"/**/*hip_api_trace.csv",recursive=True)[0])):
    if r["Function"]=="hipLaunchKernel" or "Launch" in r["Function"]:
        api[r["Correlation_Id"]]=(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Function"])
ks=[r for r in csv.DictReader(open(glob.glob(d+"/**/*kernel_trace.csv",recursive=True)[0]))]
ks.sort(key=lambda r:int(r["Start_Timestamp"]))
ks=ks[len(ks)//2:]          # the timed steps
late=collections.Counter(); dev=collections.Counter(); lt=0; dt=0; n=0
end=0; prev=None
for r in ks:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    if prev is not None and s-end>20000:
        a=api.get(r["Correlation_Id"])
        name=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","")[:40]
        pname=prev["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","")[:40]
        if a and a[0]>end-2000: late[(pname,name)]+=1; lt+=s-end
        else: dev[(pname,name)]+=1; dt+=s-end
        n+=1
    if e>end: end=e; prev=r
span=(int(ks[-1]["End_Timestamp"])-int(ks[0]["Start_Timestamp"]))/1e6
print("kernels %d span %.1f ms; gaps > 20 us: %d; host-late %.2f ms, device-side %.2f ms"%(len(ks),span,n,lt/1e6,dt/1e6))
print("host-late (launch issued after the previous kernel ended):")
for k,v in late.most_common(12): print("  %3d  %s -> %s"%(v,k[0],k[1]))
print("device-side:")
for k,v in dev.most_common(8): print("  %3d  %s -> %s"%(v,k[0],k[1]))
PY
