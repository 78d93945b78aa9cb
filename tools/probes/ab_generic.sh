cd $GRAFT_REPO_ROOT
run() { label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-compare $EXTRA > gpurun_out/ab_$label.json 2>gpurun_out/ab_$label.err || return 1
  python -c "import json; d=json.loads(open('gpurun_out/ab_$label.json').read().strip().splitlines()[-1]); print('$label', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['chains'], d['roofline'].get('stages'), d.get('latency_single_stream_ms',{}).get('p50'))"
}
for S in 1 2 4; do
EXTRA="--streams $S --frames-per-stream 1"
run s${S}plain RTMODT_PIPE=0 && run s${S}pipe RTMODT_PIPE=1 || exit 1
done
