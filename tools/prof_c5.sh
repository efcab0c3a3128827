cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ulimit -c 0; export HSA_ENABLE_COREDUMP=0
O=gpurun_out/prof_c5; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --workload ${W:-c5} --steps 3 --warmup 1 --no-cpu-baseline --no-two-call --no-fp32-form --no-graph-form > $O/trace.log 2>&1
python3 tools/kernel_exclusive.py $O/trace > $O/attribution.txt
head -32 $O/attribution.txt | cut -c1-125
