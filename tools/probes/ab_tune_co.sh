cd $GRAFT_REPO_ROOT
run() { label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-latency --no-compare $EXTRA > gpurun_out/ab_$label.json 2>gpurun_out/ab_$label.err || return 1
  python -c "import json; d=json.loads(open('gpurun_out/ab_$label.json').read().strip().splitlines()[-1]); print('$label', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
}
EXTRA=""
run co0 RTMODT_TUNE_CO=0 && run co1 RTMODT_TUNE_CO=1 && run co0b RTMODT_TUNE_CO=0 && run co1b RTMODT_TUNE_CO=1 || exit 1
for n in 1 2 3 4 5; do run pad$n RTMODT_TUNE_CO=0 RTMODT_PAD_STREAMS=$n || exit 1; done
EXTRA="--frames-per-stream 1"
run f1co0 RTMODT_TUNE_CO=0 && run f1co1 RTMODT_TUNE_CO=1 || exit 1
EXTRA="--frames-per-stream 4"
run f4co0 RTMODT_TUNE_CO=0 && run f4co1 RTMODT_TUNE_CO=1 || exit 1
