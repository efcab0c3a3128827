# how often does the first timed decode_batch of tools/bench_decode.py differ from decode() per utterance?  usage: decode_loop.sh <n> [ENV=VALUE ...]
cd $GRAFT_REPO_ROOT
N=$1; shift
for i in $(seq 1 $N); do env "$@" python3 tools/bench_decode.py --utts 32 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['debug_sha'], d['utt_per_s'], d['symbols_per_utt'], d['host_syncs_per_batch'], d['one_utterance_at_a_time']['tokens_identical_to_batched'])"; done
