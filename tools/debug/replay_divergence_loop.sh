# usage: replay_divergence_loop.sh <n> <out-file> [ENV=VALUE ...]   - one fresh process per iteration
cd $GRAFT_REPO_ROOT
N=$1; OUT=$2; shift; shift
mkdir -p gpurun_out
for i in $(seq 1 $N); do env "$@" timeout -k 10 300 python3 tools/debug/replay_divergence.py 2>&1 | tail -1 >> $OUT; echo "run $i done" ; done
