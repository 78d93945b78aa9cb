cd $GRAFT_REPO_ROOT
run() { label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-compare --no-latency $EXTRA > gpurun_out/ab_$label.json 2>gpurun_out/ab_$label.err || return 1
  python -c "import json; d=json.loads(open('gpurun_out/ab_$label.json').read().strip().splitlines()[-1]); print('$label', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['chains'], d['roofline'].get('stages'))"
}
export RTMODT_TUNE_CACHE=/tmp/tc.txt
EXTRA="--host-frames --stages 3"
run h3main A=1 && run h3copy RTMODT_COPY_ON_MAIN=0 || exit 1
EXTRA="--host-frames --stages 2"
run h2 A=1 || exit 1
EXTRA="--host-frames --stages 3"
run h3mainb A=1 || exit 1
EXTRA="--host-frames --stages 3 --pageable"
run p3main A=1 || exit 1
EXTRA="--host-frames --stages 2 --pageable"
run p2 A=1 || exit 1
