cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_r03a
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-two-call --no-fp32-form --no-graph-form > $OUT/trace.log 2>&1
python3 tools/kernel_exclusive.py $OUT/trace > $OUT/attribution.txt
python3 tools/gpu_busy.py $OUT/trace >> $OUT/attribution.txt
head -60 $OUT/attribution.txt
