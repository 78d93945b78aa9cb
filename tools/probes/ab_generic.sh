cd $GRAFT_REPO_ROOT
run() { label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-compare --no-latency $EXTRA > gpurun_out/ab_$label.json 2>gpurun_out/ab_$label.err || return 1
  python -c "import json; d=json.loads(open('gpurun_out/ab_$label.json').read().strip().splitlines()[-1]); print('$label', d['value'], d['ms_per_step'], d['roofline']['achieved'], [x['op'][-20:] for x in d['roofline']['slowest_launches'][:1]])"
}
EXTRA=""
run no192 RTMODT_NO_N192=1 && run n192 A=1 && run no192b RTMODT_NO_N192=1 && run n192b A=1 && run no192c RTMODT_NO_N192=1 && run n192c A=1 || exit 1
