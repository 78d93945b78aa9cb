cd $GRAFT_REPO_ROOT
run() { label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-compare --no-latency $EXTRA > gpurun_out/ab_$label.json 2>gpurun_out/ab_$label.err || return 1
  python -c "import json; d=json.loads(open('gpurun_out/ab_$label.json').read().strip().splitlines()[-1]); print('$label', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['chains'], d['roofline'].get('stages'))"
}
EXTRA=""
export RTMODT_TUNE_CACHE=/tmp/tc.txt
run base A=1 && run lop RTMODT_LAST_ON_POST=1 && run base2 A=1 && run lop2 RTMODT_LAST_ON_POST=1 || exit 1
EXTRA="--host-frames --stages 3"
run hlop RTMODT_LAST_ON_POST=1 || exit 1
EXTRA="--host-frames --stages 2"
run h2 A=1 || exit 1
