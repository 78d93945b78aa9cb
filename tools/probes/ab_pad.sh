cd $GRAFT_REPO_ROOT
run() { label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-latency --no-compare $EXTRA > gpurun_out/ab_$label.json 2>gpurun_out/ab_$label.err || return 1
  python -c "import json; d=json.loads(open('gpurun_out/ab_$label.json').read().strip().splitlines()[-1]); print('$label', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['chains'])"
}
EXTRA=""
for n in 0 1 2 3 4 5; do run pad$n RTMODT_TUNE_LOG=1 RTMODT_PAD_STREAMS=$n || exit 1; grep streams gpurun_out/ab_pad$n.err | head -12; done
EXTRA="--host-frames"
for n in 0 1 2; do run hpad$n RTMODT_PAD_STREAMS=$n || exit 1; done
